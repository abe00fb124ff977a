// k22 — HBM-bound kernels of the UNet step: GroupNorm32 (+SiLU, +FiLM, +fused 2x resample, zero
// border for the consuming 3x3 conv), raw resample, stem conv, timestep embedding, skinny
// (M<=8) linear, LayerNorm, K / V^T packing for attention and small casts.
//
// Reference behaviour restated (file:line relative to /root/reference):
//   GroupNorm32.forward            kandinsky2/model/nn.py:26-37      (fp32 stats, 32 groups, eps 1e-5)
//   ResBlock scale-shift norm      kandinsky2/model/unet.py:212-216  (h = GN(h)*(1+scale)+shift ; SiLU)
//   Upsample / Downsample no-conv  kandinsky2/model/unet.py:67-77, 105-107 (nearest x2 / AvgPool2d(2))
//   timestep_embedding             kandinsky2/model/nn.py:101-121
//   QKVAttention K/V concat        kandinsky2/model/unet.py:296-302  (encoder K/V prepended)
#include "kernels.h"
#include "elementwise.h"

// ------------------------------------------------------------------------------------------
// 4-element loads (one GroupNorm group never splits a 4-channel vector: C % 128 == 0)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load4(const bf16_t* p, float* f) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
  f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const f16_t* p, float* f) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  unpack2_f16(r.x, f[0], f[1]); unpack2_f16(r.y, f[2], f[3]);
}
__device__ __forceinline__ void load4(const float* p, float* f) {
  const float4 r = *reinterpret_cast<const float4*>(p);
  f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
}

// ---- GroupNorm pass 1 (stand-alone form): per (batch, pixel-range) partial sums of x and x^2 per CHANNEL ----
// grid (nsplit, B), 256 threads. Threads own fixed 4-channel vectors so the sums stay in registers; pixels of
// the block's range are walked with every wave reading whole contiguous NHWC rows.  (Tensors produced by the
// 3x3 convolutions arrive with these sums already computed by the conv epilogue; this kernel serves the rest.)
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(GnStatsParams p) {
  __shared__ float red[1024 * 2 * 2];  // [pl][C][2] with PL*C <= 2048
  const int tid = threadIdx.x, b = blockIdx.y, s = blockIdx.x;
  const int C = p.C0 + p.C1, VP = C / 4;
  const int per = (p.HW + p.nsplit - 1) / p.nsplit;
  const int pix0 = s * per, pix1 = min(p.HW, pix0 + per);
  const T* x0 = reinterpret_cast<const T*>(p.x0) + (int64_t)b * p.HW * p.C0;
  const T* x1 = p.x1 ? reinterpret_cast<const T*>(p.x1) + (int64_t)b * p.HW * p.C1 : nullptr;
  int PL, pl;
  if (VP >= 256) { PL = 1; pl = 0; } else { PL = 256 / VP; pl = tid / VP; }
  float sum[3][4], sq[3][4];
  int vj[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int v = (VP >= 256) ? tid + 256 * j : ((j == 0 && pl < PL) ? tid - pl * VP : VP);
    vj[j] = v < VP ? v : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sum[j][k] = 0.f; sq[j][k] = 0.f; }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (vj[j] >= 0) {
      const int c = vj[j] * 4;
      const T* src = c < p.C0 ? x0 + c : x1 + (c - p.C0);
      const int64_t ld = c < p.C0 ? p.C0 : p.C1;
      int pix = pix0 + pl;
      // 8 independent loads in flight per thread (a block's pixel range is short: latency-bound otherwise)
      for (; pix + 7 * PL < pix1; pix += 8 * PL) {
        float f[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) load4(src + (int64_t)(pix + u * PL) * ld, f[u]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sum[j][k] += ((f[0][k] + f[1][k]) + (f[2][k] + f[3][k])) + ((f[4][k] + f[5][k]) + (f[6][k] + f[7][k]));
          sq[j][k] += ((f[0][k] * f[0][k] + f[1][k] * f[1][k]) + (f[2][k] * f[2][k] + f[3][k] * f[3][k])) +
                      ((f[4][k] * f[4][k] + f[5][k] * f[5][k]) + (f[6][k] * f[6][k] + f[7][k] * f[7][k]));
        }
      }
      for (; pix < pix1; pix += PL) {
        float f[4];
        load4(src + (int64_t)pix * ld, f);
#pragma unroll
        for (int k = 0; k < 4; ++k) { sum[j][k] += f[k]; sq[j][k] += f[k] * f[k]; }
      }
    }
  }
  float* out = p.partial + ((int64_t)b * p.nsplit + s) * C * 2;
  if (PL == 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (vj[j] >= 0) {
        float* o = out + (int64_t)vj[j] * 8;
        *reinterpret_cast<float4*>(o) = make_float4(sum[j][0], sq[j][0], sum[j][1], sq[j][1]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(sum[j][2], sq[j][2], sum[j][3], sq[j][3]);
      }
    return;
  }
  // deterministic reduction over the PL pixel lanes (only slot j == 0 is in use when VP < 256)
  if (vj[0] >= 0) {
    float* o = red + ((int64_t)pl * C + vj[0] * 4) * 2;
    *reinterpret_cast<float4*>(o) = make_float4(sum[0][0], sq[0][0], sum[0][1], sq[0][1]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(sum[0][2], sq[0][2], sum[0][3], sq[0][3]);
  }
  __syncthreads();
  for (int i = tid; i < C * 2; i += 256) {
    float a = 0.f;
    for (int q = 0; q < PL; ++q) a += red[(int64_t)q * C * 2 + i];
    out[i] = a;
  }
}

// ---- GroupNorm pass 2: fold mean/rstd, gamma/beta and FiLM into y = x*A[c] + Bc[c] -------------
// grid (groups, B), 256 threads per (batch, group): threads sum the per-channel partials of the group's channels
// over all row blocks in a fixed strided order (fp64, 4 loads in flight per thread), fixed-order block reduction
// (deterministic), then write coeff[b][c] = (A, Bc).  The input may be a virtual concat of two tensors (a group
// may straddle them).
__global__ __launch_bounds__(256) void gn_coeff_kernel(GnCoeffParams p) {
  __shared__ double red[4][2];
  __shared__ float ms[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x, b = blockIdx.y;
  const int cg = p.C / p.groups;
  // the affine / FiLM operands do not depend on the statistics: fetch them first so that their latency overlaps
  // the partial-sum reduction (this kernel is a chain of dependent round trips, not a bandwidth problem)
  float pg = 0.f, pb = 0.f, psc = 1.f, psh = 0.f;
  if (tid < cg) {
    const int c = g * cg + tid;
    pg = p.gamma[c]; pb = p.beta[c];
    if (p.film != nullptr) {
      psc = 1.f + p.film[(int64_t)b * p.film_ld + c];
      psh = p.film[(int64_t)b * p.film_ld + p.C + c];
    }
  }
  double s = 0.0, q = 0.0;
  int off = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const GnSrc sr = p.src[k];
    if (sr.C > 0) {
      const int c_lo = max(g * cg, off), c_hi = min((g + 1) * cg, off + sr.C);
      const int nc = c_hi - c_lo;
      if (nc > 0) {
        const float* base = sr.st + ((int64_t)b * sr.rpi * sr.C + (c_lo - off)) * 2;
        const int items = sr.rpi * nc;
        int i = tid;
        for (; i + 768 < items; i += 1024) {
          float2 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ii = i + 256 * u, r = ii / nc, cc = ii - r * nc;
            v[u] = *reinterpret_cast<const float2*>(base + ((int64_t)r * sr.C + cc) * 2);
          }
          s += ((double)v[0].x + (double)v[1].x) + ((double)v[2].x + (double)v[3].x);
          q += ((double)v[0].y + (double)v[1].y) + ((double)v[2].y + (double)v[3].y);
        }
        for (; i < items; i += 256) {
          const int r = i / nc, cc = i - r * nc;
          const float2 v = *reinterpret_cast<const float2*>(base + ((int64_t)r * sr.C + cc) * 2);
          s += (double)v.x;
          q += (double)v.y;
        }
      }
    }
    off += sr.C;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) { red[wave][0] = s; red[wave][1] = q; }
  __syncthreads();
  if (tid == 0) {
    const double st = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    const double qt = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const double n = (double)p.HW * (double)cg;
    const double mean = st / n;
    double var = qt / n - mean * mean;
    if (var < 0.0) var = 0.0;
    ms[0] = (float)mean;
    ms[1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  const float mean_f = ms[0], rstd_f = ms[1];
  if (tid < cg) {  // cg <= 96 (C <= 3072)
    const int c = g * cg + tid;
    float A = rstd_f * pg;
    float Bc = pb - mean_f * A;
    if (p.film != nullptr) {
      A *= psc;
      Bc = Bc * psc + psh;
    }
    *reinterpret_cast<float2*>(p.coeff + ((int64_t)b * p.C + c) * 2) = make_float2(A, Bc);
  }
}

// ---- GroupNorm pass 3: apply (+act) (+avgpool2 / nearest-up2) and write the (optionally
// zero-bordered) NHWC tensor the next conv / GEMM consumes.  One thread = 16 bytes of output;
// grid (chunks of one output row, padded output rows, batch): one 32-bit division per thread.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyParams p) {
  constexpr int EPV = Vec16<T>::N;
  // split-precision store: SiLU through the native exp2 / rcp (common.h: silu_fast; the same choice as the fused form of this pass in
  // conv3_common.h: gn_rewrite16 - the two give the same bits).  The 16-bit engines keep the IEEE form: see common.h
  const bool fast = p.out_x3 != 0;
  const int C = p.C0 + p.C1, CV = C / EPV;
  const int Ho = p.mode == 1 ? p.H / 2 : (p.mode == 2 ? p.H * 2 : p.H);
  const int Wo = p.mode == 1 ? p.W / 2 : (p.mode == 2 ? p.W * 2 : p.W);
  const int pad = p.pad, Hp = Ho + 2 * pad, Wp = Wo + 2 * pad;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wp * CV) return;
  const int xp = idx / CV, cv = idx - xp * CV;
  const int xo = xp - pad, yo = (int)blockIdx.y - pad, b = blockIdx.z;
  T* out = reinterpret_cast<T*>(p.out) + (((int64_t)b * Hp + blockIdx.y) * Wp + xp) * C + cv * EPV;
  Vec16<T> o;
  if (xo < 0 || yo < 0 || xo >= Wo || yo >= Ho) {
#pragma unroll
    for (int k = 0; k < EPV / 2; ++k) o.set2(k, 0.f, 0.f);
  } else {
    const int c = cv * EPV;
    const T* src; int cs, ld;
    if (c < p.C0) { src = reinterpret_cast<const T*>(p.x0); cs = c; ld = p.C0; }
    else { src = reinterpret_cast<const T*>(p.x1); cs = c - p.C0; ld = p.C1; }
    float A[EPV], Bc[EPV];
    const float* cf = p.coeff + ((int64_t)b * C + c) * 2;
#pragma unroll
    for (int k = 0; k < EPV; k += 2) {
      const float4 q = *reinterpret_cast<const float4*>(cf + 2 * k);
      A[k] = q.x; Bc[k] = q.y; A[k + 1] = q.z; Bc[k + 1] = q.w;
    }
    float r[EPV];
    if (p.mode == 1) {
      Vec16<T> v[4];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
          v[dy * 2 + dx].raw = *reinterpret_cast<const decltype(v[0].raw)*>(src + ((int64_t)(b * p.H + 2 * yo + dy) * p.W + 2 * xo + dx) * ld + cs);
#pragma unroll
      for (int k = 0; k < EPV; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += fast ? apply_act_sel<true>(v[q].get(k) * A[k] + Bc[k], p.act) : apply_act_sel<false>(v[q].get(k) * A[k] + Bc[k], p.act);
        r[k] = acc * 0.25f;
      }
    } else {
      const int yi = p.mode == 2 ? yo >> 1 : yo, xi = p.mode == 2 ? xo >> 1 : xo;
      Vec16<T> v;
      v.raw = *reinterpret_cast<const decltype(v.raw)*>(src + ((int64_t)(b * p.H + yi) * p.W + xi) * ld + cs);
#pragma unroll
      for (int k = 0; k < EPV; ++k) r[k] = fast ? apply_act_sel<true>(v.get(k) * A[k] + Bc[k], p.act) : apply_act_sel<false>(v.get(k) * A[k] + Bc[k], p.act);
    }
#pragma unroll
    for (int k = 0; k < EPV / 2; ++k) o.set2(k, r[2 * k], r[2 * k + 1]);
  }
  if constexpr (sizeof(T) == 4) {
    // split-precision engine: the consumer is an MFMA kernel that reads its A operand in x3 chunks (common.h); a zero stays zero
    if (p.out_x3) { x3_store4(out, x3_split4(o.raw)); return; }
  }
  *reinterpret_cast<decltype(o.raw)*>(out) = o.raw;
}

// ---- raw 2x resample of the residual branch (x_upd), unpadded NHWC -> unpadded NHWC ------------
template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const void* xin, void* yout, int B, int H, int W, int C, int mode) {
  constexpr int EPV = Vec16<T>::N;
  const int CV = C / EPV;
  const int Ho = mode == 1 ? H / 2 : H * 2, Wo = mode == 1 ? W / 2 : W * 2;
  const int64_t total = (int64_t)B * Ho * Wo * CV;
  const T* x = reinterpret_cast<const T*>(xin);
  T* y = reinterpret_cast<T*>(yout);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    int64_t t = i / CV;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    Vec16<T> o;
    if (mode == 1) {
      float r[EPV];
#pragma unroll
      for (int k = 0; k < EPV; ++k) r[k] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          Vec16<T> v;
          v.raw = *reinterpret_cast<const decltype(v.raw)*>(x + ((int64_t)(b * H + 2 * yo + dy) * W + 2 * xo + dx) * C + cv * EPV);
#pragma unroll
          for (int k = 0; k < EPV; ++k) r[k] += v.get(k);
        }
#pragma unroll
      for (int k = 0; k < EPV / 2; ++k) o.set2(k, r[2 * k] * 0.25f, r[2 * k + 1] * 0.25f);
    } else {
      o.raw = *reinterpret_cast<const decltype(o.raw)*>(x + ((int64_t)(b * H + (yo >> 1)) * W + (xo >> 1)) * C + cv * EPV);
    }
    *reinterpret_cast<decltype(o.raw)*>(y + i * EPV) = o.raw;
  }
}

// ---- stem: conv3x3 over the fp32 NCHW latent (4 ch, or 9 = [x, img*mask, mask] for inpainting)
// (unet.py:426 input_blocks[0]; text2im_model2_1.py:146-155).  One block = PT pixels of one row (8: at 96x96, B = 2 that is
// 2304 workgroups; with 32-pixel tiles the 576 workgroups left most of the 256 CUs with a single 128-thread block).
constexpr int CONV_IN_PT = 8;
template <typename T, int CIN>
__global__ __launch_bounds__(128) void conv_in_kernel(ConvInParams p) {
  constexpr int PT = CONV_IN_PT, PW = PT + 2;
  __shared__ float patch[CIN][3][PW];
  const int tid = threadIdx.x;
  const int xt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
  const int x0 = xt * PT;
  const int hw = p.H * p.W;
  for (int i = tid; i < CIN * 3 * PW; i += 128) {
    const int c = i / (3 * PW), r = (i / PW) % 3, xx = i % PW;
    const int yy = y + r - 1, xs = x0 + xx - 1;
    float v = 0.f;
    if (yy >= 0 && yy < p.H && xs >= 0 && xs < p.W) {
      const int64_t o = (int64_t)yy * p.W + xs;
      if (c < 4) v = p.x[((int64_t)b * 4 + c) * hw + o];
      else if (c < 8) v = p.img[((int64_t)b * 4 + (c - 4)) * hw + o] * (p.img_premul ? 1.f : p.mask[(int64_t)b * hw + o]);
      else v = p.mask[(int64_t)b * hw + o];
    }
    patch[c][r][xx] = v;
  }
  __syncthreads();
  T* out = reinterpret_cast<T*>(p.out);
  for (int n = tid; n < p.Cout; n += 128) {
    float w[CIN * 9];
#pragma unroll
    for (int k = 0; k < CIN * 9; ++k) w[k] = p.w[(int64_t)k * p.Cout + n];   // packed [Cin*9][Cout]: coalesced over n
    const float bias = p.bias[n];
    for (int px = 0; px < PT && x0 + px < p.W; ++px) {
      float acc = bias;
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int q = 0; q < 3; ++q) acc += w[c * 9 + r * 3 + q] * patch[c][r][px + q];
      out[((int64_t)(b * p.H + y) * p.W + x0 + px) * p.Cout + n] = from_f32<T>(acc);
    }
  }
}

// ---- direct fp32 3x3 convolution (hint stack of the 2.2 ControlNet-depth UNet; once per generation) -----------------
// thread = one output pixel x OCB output channels; the weights of a workgroup's channel group are wave-uniform loads.
template <int OCB>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(ConvDirectParams p) {
  const int Ho = (p.Hin - 1) / p.stride + 1, Wo = (p.Win - 1) / p.stride + 1;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int oc0 = blockIdx.y * OCB, b = blockIdx.z;
  if (pix >= Ho * Wo) return;
  const int yo = pix / Wo, xo = pix - yo * Wo;
  const int yi = yo * p.stride - 1, xi = xo * p.stride - 1;
  float acc[OCB];
#pragma unroll
  for (int o = 0; o < OCB; ++o) acc[o] = (oc0 + o < p.Cout) ? p.bias[oc0 + o] : 0.f;
  const int64_t plane = (int64_t)p.Hin * p.Win;
  for (int c = 0; c < p.Cin; ++c) {
    const float* xp = p.x + ((int64_t)b * p.Cin + c) * plane;
    float v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int yy = yi + r, xx = xi + q;
        v[r * 3 + q] = (yy >= 0 && yy < p.Hin && xx >= 0 && xx < p.Win) ? xp[(int64_t)yy * p.Win + xx] : 0.f;
      }
#pragma unroll
    for (int o = 0; o < OCB; ++o) {
      if (oc0 + o < p.Cout) {
        const float* wp = p.w + ((int64_t)(oc0 + o) * p.Cin + c) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[o] += wp[k] * v[k];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < OCB; ++o)
    if (oc0 + o < p.Cout) p.y[(((int64_t)b * p.Cout + oc0 + o) * Ho + yo) * Wo + xo] = apply_act(acc[o], p.act);
}

// ---- sinusoidal timestep embedding with a host-supplied frequency table (nn.py:101-121) --------
__global__ void timestep_embedding_kernel(const float* t, const float* freqs, float* out, int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float a = t[b] * freqs[k];
  out[(int64_t)b * 2 * half + k] = cosf(a);
  out[(int64_t)b * 2 * half + half + k] = sinf(a);
}

// ---- skinny linear: out[m][n] = act_out( sum_k act_in(x[m][k]) * W[n][k] + bias[n] ) + add[m][n]
// M <= 8 rows.  HBM-bound GEMV: the weights are streamed exactly once (16 B per lane per load, one wave
// per output feature, several features per wave); act_in(x) is staged ONCE per workgroup in LDS as fp32.
template <typename TW, int MT>
__global__ __launch_bounds__(256) void linear_smallm_kernel(LinearSmallParams p) {
  constexpr int EPC = 16 / sizeof(TW);
  extern __shared__ __attribute__((aligned(16))) char smem_lin[];
  float* xs = reinterpret_cast<float*>(smem_lin);  // [M][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < MT * p.K; i += 256) {
    const int m = i / p.K, k = i - m * p.K;
    float xv = m < p.M ? p.x[(int64_t)m * p.ldx + k] : 0.f;  // template rows beyond p.M are zero
    if (p.act_in == K22_ACT_SILU) xv = silu_f(xv);
    xs[i] = xv;
  }
  __syncthreads();
  const int per_wave = p.rows_per_wave;
  const int n_begin = (blockIdx.x * 4 + wave) * per_wave;
  for (int r = 0; r < per_wave; ++r) {
    const int n = n_begin + r;
    if (n >= p.N) return;
    const TW* w = reinterpret_cast<const TW*>(p.W) + (int64_t)n * p.K;
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    for (int k = lane * EPC; k < p.K; k += 64 * EPC) {
      Vec16<TW> wv;
      wv.raw = *reinterpret_cast<const decltype(wv.raw)*>(w + k);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float* xr = xs + m * p.K + k;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const float4 xv = *reinterpret_cast<const float4*>(xr + e);
          acc[m] += xv.x * wv.get(e) + xv.y * wv.get(e + 1) + xv.z * wv.get(e + 2) + xv.w * wv.get(e + 3);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v = wave_sum(acc[m]);
      if (lane == 0 && m < p.M) {
        if (p.bias) v += p.bias[n];
        v = apply_act(v, p.act_out);
        if (p.add) v += p.add[(int64_t)(p.add_mod > 0 ? m % p.add_mod : m) * p.ld_add + n];
        p.out[(int64_t)m * p.ldo + n] = v;
      }
    }
  }
}

// ---- LayerNorm over the last dim of fp32 rows (text2im_model2_1.py:71 ln_model_n; prior.py) ----
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* x, const float* g, const float* bta, float* y, int D, float eps) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)row * D;
  float s = 0.f;
  for (int i = tid; i < D; i += 256) s += xr[i];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / D;
  float q = 0.f;
  for (int i = tid; i < D; i += 256) { const float d = xr[i] - mean; q += d * d; }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / D + eps);
  for (int i = tid; i < D; i += 256) y[(int64_t)row * D + i] = (xr[i] - mean) * rstd * g[i] + bta[i];
}

// ---- fp32 -> T cast with row re-striding (ctx assembly) -----------------------------------
template <typename T>
__global__ void cast_rows_kernel(const float* x, void* yout, int rows, int cols, int64_t ldx, int64_t ldy) {
  T* y = reinterpret_cast<T*>(yout);
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    y[r * ldy + c] = from_f32<T>(x[r * ldx + c]);
  }
}

// ---- attention operand packing ----------------------------------------------------------------
// K_all[b][h][j][64]  and  VT_all[b][h][d][Tkp]  with keys j = [ctx(0..S-1) | self(S..S+T-1) | 0-pad].
// qkv  : [B*T][3C] columns [q | k | v], each [H][64];   ctxkv : [B*S][2C] columns [k | v].
// grid (Tkp/64, H, B), 256 threads; V tile transposed through LDS.
template <typename T>
__global__ __launch_bounds__(256) void kv_pack_kernel(KvPackParams p) {
  __shared__ T vt[64][66];
  const int tid = threadIdx.x, jt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int C = p.H * 64, Tk = p.S + p.T;
  const T* qkv = reinterpret_cast<const T*>(p.qkv);
  const T* ckv = reinterpret_cast<const T*>(p.ctxkv);
  T* Kall = reinterpret_cast<T*>(p.kall) + ((int64_t)(b * p.H + h) * p.Tkp) * 64;
  T* VTall = reinterpret_cast<T*>(p.vtall) + ((int64_t)(b * p.H + h) * 64) * p.Tkp;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int jj = i >> 6, d = i & 63;
    const int j = jt * 64 + jj;
    T kv = from_f32<T>(0.f), vv = from_f32<T>(0.f);
    if (j < p.S) {
      const T* r = ckv + (int64_t)(b * p.S + j) * 2 * C + h * 64 + d;
      kv = r[0]; vv = r[C];
    } else if (j < Tk) {
      const T* r = qkv + (int64_t)(b * p.T + (j - p.S)) * 3 * C + C + h * 64 + d;
      kv = r[0]; vv = r[C];
    }
    Kall[(int64_t)j * 64 + d] = kv;
    vt[d][jj] = vv;
  }
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) {
    const int d = i >> 6, jj = i & 63;
    VTall[(int64_t)d * p.Tkp + jt * 64 + jj] = vt[d][jj];
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// fp32 -> x3 chunks (common.h): the operand format of the split-precision arithmetic.  scale = K22_X3_WSCALE for weights, 1 for activations.
__global__ __launch_bounds__(256) void x3_pack_kernel(const float4* __restrict__ in, u32x4_t* __restrict__ out, int64_t n4, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = in[i];
    x3_store4(out + i, x3_split4(v.x * scale, v.y * scale, v.z * scale, v.w * scale));
  }
}
int launch_x3_pack(const float* in, void* out, int64_t n, float scale, hipStream_t s) {
  if (n % 8 || (uintptr_t)in % 16 || (uintptr_t)out % 32) return k22_set_error(K22_EINVAL, "x3_pack: n % 8 == 0, 16-byte aligned input, 32-byte aligned output (groups of eight)");
  const int64_t n4 = n / 4;
  int nb = (int)((n4 + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (nb < 1) return K22_OK;
  hipLaunchKernelGGL(x3_pack_kernel, dim3(nb), dim3(256), 0, s, reinterpret_cast<const float4*>(in), reinterpret_cast<u32x4_t*>(out), n4, scale);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

static inline int grid_for(int64_t total, int threads, int cap) {
  int64_t nb = (total + threads - 1) / threads;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

int gn_nsplit(int B, int HW) {
  // pixel ranges per image: enough workgroups to fill the chip on the big levels, but few enough that the
  // [nsplit][C][2] partials stay small next to the tensor (>= 32 pixels per range)
  int ns = 512 / (B > 0 ? B : 1);
  if (ns > 128) ns = 128;
  while (ns > 1 && HW / ns < 32) ns >>= 1;
  return ns < 1 ? 1 : ns;
}

int launch_gn_stats(const GnStatsParams& p, int dtype, hipStream_t s) {
  const int C = p.C0 + p.C1;
  if (p.groups != 32 || C % 128 != 0 || p.C0 % 4 != 0) return k22_set_error(K22_EINVAL, "gn_stats: need 32 groups, C % 128 == 0");
  if (C / 4 > 768) return k22_set_error(K22_EINVAL, "gn_stats: C too large (max 3072)");
  if (C / 4 < 256 && (256 / (C / 4)) * C > 2048) return k22_set_error(K22_EINVAL, "gn_stats: internal LDS bound");
  dim3 grid(p.nsplit, p.B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16) hipLaunchKernelGGL(gn_stats_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_gn_coeff(const GnCoeffParams& p, int B, hipStream_t s) {
  if (p.C / p.groups > 256) return k22_set_error(K22_EINVAL, "gn_coeff: more than 256 channels per group");
  hipLaunchKernelGGL(gn_coeff_kernel, dim3(p.groups, B), dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_gn_apply(const GnApplyParams& p, int dtype, hipStream_t s) {
  const int C = p.C0 + p.C1;
  const int epv = dtype == K22_F32 ? 4 : 8;
  if (C % epv || p.C0 % epv) return k22_set_error(K22_EINVAL, "gn_apply: channel alignment");
  if (p.out_x3 && dtype != K22_F32) return k22_set_error(K22_EINVAL, "gn_apply: x3-chunk output is an option of the fp32 launch");
  const int Ho = p.mode == 1 ? p.H / 2 : (p.mode == 2 ? p.H * 2 : p.H);
  const int Wo = p.mode == 1 ? p.W / 2 : (p.mode == 2 ? p.W * 2 : p.W);
  const int Hp = Ho + 2 * p.pad, Wp = Wo + 2 * p.pad;
  if (Hp > 65535 || p.B > 65535) return k22_set_error(K22_EINVAL, "gn_apply: tensor too large");
  dim3 grid((Wp * (C / epv) + 255) / 256, Hp, p.B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16) hipLaunchKernelGGL(gn_apply_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_resample(const void* x, void* y, int B, int H, int W, int C, int mode, int dtype, hipStream_t s) {
  const int epv = dtype == K22_F32 ? 4 : 8;
  const int Ho = mode == 1 ? H / 2 : H * 2, Wo = mode == 1 ? W / 2 : W * 2;
  const int nb = grid_for((int64_t)B * Ho * Wo * (C / epv), 256, 8192);
  if (dtype == K22_BF16) hipLaunchKernelGGL(resample_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, x, y, B, H, W, C, mode);
  else if (dtype == K22_F16) hipLaunchKernelGGL(resample_kernel<f16_t>, dim3(nb), dim3(256), 0, s, x, y, B, H, W, C, mode);
  else hipLaunchKernelGGL(resample_kernel<float>, dim3(nb), dim3(256), 0, s, x, y, B, H, W, C, mode);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_conv_in(const ConvInParams& p, int dtype, hipStream_t s) {
  dim3 grid((p.W + CONV_IN_PT - 1) / CONV_IN_PT, p.H, p.B);
  if (p.Cin == 4) {
    if (dtype == K22_BF16) hipLaunchKernelGGL((conv_in_kernel<bf16_t, 4>), grid, dim3(128), 0, s, p);
    else if (dtype == K22_F16) hipLaunchKernelGGL((conv_in_kernel<f16_t, 4>), grid, dim3(128), 0, s, p);
    else hipLaunchKernelGGL((conv_in_kernel<float, 4>), grid, dim3(128), 0, s, p);
  } else if (p.Cin == 9) {
    if (dtype == K22_BF16) hipLaunchKernelGGL((conv_in_kernel<bf16_t, 9>), grid, dim3(128), 0, s, p);
    else if (dtype == K22_F16) hipLaunchKernelGGL((conv_in_kernel<f16_t, 9>), grid, dim3(128), 0, s, p);
    else hipLaunchKernelGGL((conv_in_kernel<float, 9>), grid, dim3(128), 0, s, p);
  } else if (p.Cin == 8) {
    if (dtype == K22_BF16) hipLaunchKernelGGL((conv_in_kernel<bf16_t, 8>), grid, dim3(128), 0, s, p);
    else if (dtype == K22_F16) hipLaunchKernelGGL((conv_in_kernel<f16_t, 8>), grid, dim3(128), 0, s, p);
    else hipLaunchKernelGGL((conv_in_kernel<float, 8>), grid, dim3(128), 0, s, p);
  } else {
    return k22_set_error(K22_EINVAL, "conv_in: in_channels must be 4, 8 or 9");
  }
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_conv3x3_direct(const ConvDirectParams& p, hipStream_t s) {
  if (p.stride != 1 && p.stride != 2) return k22_set_error(K22_EINVAL, "conv3x3_direct: stride must be 1 or 2");
  const int Ho = (p.Hin - 1) / p.stride + 1, Wo = (p.Win - 1) / p.stride + 1;
  constexpr int OCB = 8;
  dim3 grid((Ho * Wo + 255) / 256, (p.Cout + OCB - 1) / OCB, p.B);
  hipLaunchKernelGGL((conv3x3_direct_kernel<OCB>), grid, dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_timestep_embedding(const float* t, const float* freqs, float* out, int B, int half, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * half + 255) / 256), dim3(256), 0, s, t, freqs, out, B, half);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
template <typename TW>
static void launch_linear_m(const LinearSmallParams& p, int Mt, dim3 grid, size_t smem, hipStream_t s) {
  switch (Mt) {
    case 1: hipLaunchKernelGGL((linear_smallm_kernel<TW, 1>), grid, dim3(256), smem, s, p); break;
    case 2: hipLaunchKernelGGL((linear_smallm_kernel<TW, 2>), grid, dim3(256), smem, s, p); break;
    case 4: hipLaunchKernelGGL((linear_smallm_kernel<TW, 4>), grid, dim3(256), smem, s, p); break;
    default: hipLaunchKernelGGL((linear_smallm_kernel<TW, 8>), grid, dim3(256), smem, s, p); break;
  }
}
int launch_linear_smallm(const LinearSmallParams& p0, int wdtype, hipStream_t s) {
  if (p0.M > 8 || p0.M < 1) return k22_set_error(K22_EINVAL, "linear_smallm: M must be in 1..8");
  const int epc = wdtype == K22_F32 ? 4 : 8;
  if (p0.K % epc) return k22_set_error(K22_EINVAL, "linear_smallm: K alignment");
  const int Mt = p0.M <= 2 ? p0.M : (p0.M <= 4 ? 4 : 8);  // rows the kernel is instantiated for
  const size_t smem = (size_t)Mt * p0.K * 4;
  if (smem > 64 * 1024) return k22_set_error(K22_EINVAL, "linear_smallm: M*K too large for the LDS stage");
  LinearSmallParams p = p0;
  // several features per wave when N is large (the LDS stage of x is amortised), still >= 1024 workgroups
  int rpw = 1;
  while (rpw < 8 && (p.N + 8 * rpw - 1) / (8 * rpw) >= 1024) rpw *= 2;
  p.rows_per_wave = rpw;
  dim3 grid((p.N + 4 * rpw - 1) / (4 * rpw));
  if (wdtype == K22_BF16) launch_linear_m<bf16_t>(p, Mt, grid, smem, s);
  else if (wdtype == K22_F16) launch_linear_m<f16_t>(p, Mt, grid, smem, s);
  else launch_linear_m<float>(p, Mt, grid, smem, s);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_layernorm_f32(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3(rows), dim3(256), 0, s, x, g, b, y, D, eps);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_cast_rows(const float* x, void* y, int rows, int cols, int64_t ldx, int64_t ldy, int dtype, hipStream_t s) {
  const int nb = grid_for((int64_t)rows * cols, 256, 4096);
  if (dtype == K22_BF16) hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, x, y, rows, cols, ldx, ldy);
  else if (dtype == K22_F16) hipLaunchKernelGGL(cast_rows_kernel<f16_t>, dim3(nb), dim3(256), 0, s, x, y, rows, cols, ldx, ldy);
  else hipLaunchKernelGGL(cast_rows_kernel<float>, dim3(nb), dim3(256), 0, s, x, y, rows, cols, ldx, ldy);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_kv_pack(const KvPackParams& p, int dtype, hipStream_t s) {
  if (p.Tkp % 64) return k22_set_error(K22_EINVAL, "kv_pack: Tkp % 64");
  dim3 grid(p.Tkp / 64, p.H, p.B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(kv_pack_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16) hipLaunchKernelGGL(kv_pack_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(kv_pack_kernel<float>, grid, dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
