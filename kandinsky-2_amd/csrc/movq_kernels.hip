// k22 — HBM-bound kernels of the MoVQ decoder (kandinsky2/vqgan/movq_modules.py).
//
//   SpatialNorm.forward (movq_modules.py:61-68): GroupNorm(f, 32 groups, eps 1e-6) * conv_y(zq) + conv_b(zq) with
//     zq = the raw 4-channel latent, nearest-resized to f's size; conv_y / conv_b are 1x1 (4 -> C), evaluated on the
//     fly here (8 FMAs per element) instead of materialising two [B,C,H,W] maps.  Optional SiLU (nonlinearity,
//     :29-31) and zero border for the consuming 3x3 conv.
//   Upsample.forward (:85-98): nearest x2 (+ zero border for its conv).
//   AttnBlock softmax (:218): softmax(q.k * C^-0.5) over keys, rows of a materialised [T][T] score matrix.
//   post_quant_conv (autoencoder.py:167,182-185) and the uint8 image epilogue (kandinsky2/utils.py:57-70).
#include "kernels.h"
#include "elementwise.h"

// One thread = one 16-byte channel vector of a RUN of R = min(2^shift, 8) consecutive pixels of a row: the latent zq is nearest-
// upsampled by 2^shift, so those pixels share zq and with it conv_y(zq), conv_b(zq) - the 13 parameter loads and 16 FMAs per channel
// that made the first form of this kernel load-issue bound (48 cached loads per 16 bytes of activation: 1.3 TB/s at 768 x 768) are
// paid once per run instead of once per pixel.  Per-element arithmetic and its order are unchanged (same bits).
template <typename T>
__global__ __launch_bounds__(256) void spatialnorm_apply_kernel(SpatialNormParams p) {
  constexpr int EPV = Vec16<T>::N;
  const int CV = p.C / EPV;
  const int pad = p.pad, Hp = p.H + 2 * pad, Wp = p.W + 2 * pad;
  const int R = p.shift >= 3 ? 8 : (1 << p.shift);
  const int runs = (p.W + R - 1) / R;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= runs * CV) return;
  const int run = idx / CV, cv = idx - run * CV;
  const int x0 = run * R, yo = (int)blockIdx.y - pad, b = blockIdx.z;
  const int c = cv * EPV;
  T* orow = reinterpret_cast<T*>(p.out) + ((int64_t)b * Hp + blockIdx.y) * Wp * p.C + c;
  Vec16<T> zero;
#pragma unroll
  for (int k = 0; k < EPV / 2; ++k) zero.set2(k, 0.f, 0.f);
  if (pad) {   // the zero border columns belong to the first / last run of the row
    if (run == 0) *reinterpret_cast<decltype(zero.raw)*>(orow) = zero.raw;
    if (run == runs - 1) *reinterpret_cast<decltype(zero.raw)*>(orow + (int64_t)(Wp - 1) * p.C) = zero.raw;
  }
  const int xend = x0 + R < p.W ? x0 + R : p.W;
  if (yo < 0 || yo >= p.H) {
    for (int x = x0; x < xend; ++x) *reinterpret_cast<decltype(zero.raw)*>(orow + (int64_t)(x + pad) * p.C) = zero.raw;
    return;
  }
  const float4 z = *reinterpret_cast<const float4*>(p.zq + (((int64_t)b * p.h0 + (yo >> p.shift)) * p.w0 + (x0 >> p.shift)) * 4);
  const float* cf = p.coeff + ((int64_t)b * p.C + c) * 2;
  float A[EPV], Bc[EPV], sy[EPV], sb[EPV];
#pragma unroll
  for (int k = 0; k < EPV; ++k) {
    const float2 ab = *reinterpret_cast<const float2*>(cf + 2 * k);
    const float4 wy = *reinterpret_cast<const float4*>(p.wy + (c + k) * 4);
    const float4 wb = *reinterpret_cast<const float4*>(p.wb + (c + k) * 4);
    A[k] = ab.x; Bc[k] = ab.y;
    sy[k] = ((wy.x * z.x + wy.y * z.y) + (wy.z * z.z + wy.w * z.w)) + p.by[c + k];
    sb[k] = ((wb.x * z.x + wb.y * z.y) + (wb.z * z.z + wb.w * z.w)) + p.bb[c + k];
  }
  const T* xrow = reinterpret_cast<const T*>(p.x) + ((int64_t)b * p.H + yo) * p.W * p.C + c;
  for (int x = x0; x < xend; ++x) {
    Vec16<T> v, o;
    v.raw = *reinterpret_cast<const decltype(v.raw)*>(xrow + (int64_t)x * p.C);
    float r[EPV];
#pragma unroll
    for (int k = 0; k < EPV; ++k) {
      const float norm = v.get(k) * A[k] + Bc[k];
      r[k] = apply_act(norm * sy[k] + sb[k], p.act);
    }
#pragma unroll
    for (int k = 0; k < EPV / 2; ++k) o.set2(k, r[2 * k], r[2 * k + 1]);
    *reinterpret_cast<decltype(o.raw)*>(orow + (int64_t)(x + pad) * p.C) = o.raw;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_pad_kernel(const void* xin, void* yout, int B, int H, int W, int C) {
  constexpr int EPV = Vec16<T>::N;
  const int CV = C / EPV, Hp = 2 * H + 2, Wp = 2 * W + 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wp * CV) return;
  const int xp = idx / CV, cv = idx - xp * CV;
  const int xo = xp - 1, yo = (int)blockIdx.y - 1, b = blockIdx.z;
  Vec16<T> o;
  if (xo < 0 || yo < 0 || xo >= 2 * W || yo >= 2 * H) {
#pragma unroll
    for (int k = 0; k < EPV / 2; ++k) o.set2(k, 0.f, 0.f);
  } else {
    o.raw = *reinterpret_cast<const decltype(o.raw)*>(reinterpret_cast<const T*>(xin) + (((int64_t)b * H + (yo >> 1)) * W + (xo >> 1)) * C + cv * EPV);
  }
  *reinterpret_cast<decltype(o.raw)*>(reinterpret_cast<T*>(yout) + (((int64_t)b * Hp + blockIdx.y) * Wp + xp) * C + cv * EPV) = o.raw;
}

// one workgroup per row; the row (<= 16 K elements at 1024^2) is read twice from L2, written once
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(void* xio, int L, float scale) {
  constexpr int EPV = Vec16<T>::N;
  __shared__ float red[8];
  T* x = reinterpret_cast<T*>(xio) + (int64_t)blockIdx.x * L;
  const int tid = threadIdx.x;
  const int nv = L / EPV;
  float m = -INFINITY;
  for (int i = tid; i < nv; i += 256) {
    Vec16<T> v;
    v.raw = *reinterpret_cast<const decltype(v.raw)*>(x + i * EPV);
#pragma unroll
    for (int k = 0; k < EPV; ++k) m = fmaxf(m, v.get(k));
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  float s = 0.f;
  for (int i = tid; i < nv; i += 256) {
    Vec16<T> v;
    v.raw = *reinterpret_cast<const decltype(v.raw)*>(x + i * EPV);
#pragma unroll
    for (int k = 0; k < EPV; ++k) s += expf(v.get(k) * scale - m);
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
  __syncthreads();
  const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
  for (int i = tid; i < nv; i += 256) {
    Vec16<T> v, o;
    v.raw = *reinterpret_cast<const decltype(v.raw)*>(x + i * EPV);
#pragma unroll
    for (int k = 0; k < EPV / 2; ++k) o.set2(k, expf(v.get(2 * k) * scale - m) * inv, expf(v.get(2 * k + 1) * scale - m) * inv);
    *reinterpret_cast<decltype(o.raw)*>(x + i * EPV) = o.raw;
  }
}

template <typename T>
__global__ void movq_prepare_kernel(const float* z, const float* wpq, const float* bpq, float* zq, void* xin, int B, int h, int w, int Cpad) {
  // one thread per PADDED pixel
  const int Hp = h + 2, Wp = w + 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Hp * Wp) return;
  const int xp = i % Wp, yp = (i / Wp) % Hp, b = i / (Wp * Hp);
  T* dst = reinterpret_cast<T*>(xin) + (int64_t)i * Cpad;
  const int y = yp - 1, x = xp - 1;
  float q[4] = {0.f, 0.f, 0.f, 0.f};
  const bool inside = (x >= 0 && y >= 0 && x < w && y < h);
  if (inside) {
    float zi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) zi[c] = z[(((int64_t)b * 4 + c) * h + y) * w + x];
    *reinterpret_cast<float4*>(zq + (((int64_t)b * h + y) * w + x) * 4) = make_float4(zi[0], zi[1], zi[2], zi[3]);
#pragma unroll
    for (int o = 0; o < 4; ++o) q[o] = ((wpq[o * 4] * zi[0] + wpq[o * 4 + 1] * zi[1]) + (wpq[o * 4 + 2] * zi[2] + wpq[o * 4 + 3] * zi[3])) + bpq[o];
  }
  for (int c = 0; c < Cpad; ++c) dst[c] = from_f32<T>(c < 4 ? q[c] : 0.f);
}

__global__ void to_uint8_nhwc_kernel(const float* x, unsigned char* y, int B, int C, int H, int W) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int b = (int)(pix / ((int64_t)H * W));
    const int64_t rem = pix - (int64_t)b * H * W;
    float v = (x[((int64_t)b * C + c) * H * W + rem] + 1.f) * 127.5f;
    v = rintf(v);                       // torch.round: half to even
    v = fminf(fmaxf(v, 0.f), 255.f);
    y[i] = (unsigned char)v;
  }
}

// ---- MoVQ ENCODER helpers (Encoder.forward, kandinsky2/vqgan/vqgan_blocks.py:335-367) ------------------------------------
// image fp32 NCHW [B][3][H][W] -> zero-bordered NHWC T [B][H+2][W+2][Cpad], channels 3.. zero (conv_in weights are
// zero-extended to Cpad input channels the same way): one thread per padded pixel
template <typename T>
__global__ void movq_enc_prepare_kernel(const float* img, void* xin, int B, int H, int W, int Cpad) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * Hp * Wp) return;
  const int xp = (int)(i % Wp), yp = (int)((i / Wp) % Hp), b = (int)(i / ((int64_t)Wp * Hp));
  T* dst = reinterpret_cast<T*>(xin) + i * Cpad;
  const int y = yp - 1, x = xp - 1;
  float v[3] = {0.f, 0.f, 0.f};
  if (x >= 0 && y >= 0 && x < W && y < H) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = img[(((int64_t)b * 3 + c) * H + y) * W + x];
  }
  for (int c = 0; c < Cpad; ++c) dst[c] = from_f32<T>(c < 3 ? v[c] : 0.f);
}

// NHWC [B][H][W][C] -> zero-bordered [B][H+2][W+2][C] (input of Downsample's 3x3 convolution)
template <typename T>
__global__ __launch_bounds__(256) void pad_copy_kernel(const void* xin, void* yout, int B, int H, int W, int C) {
  constexpr int EPV = Vec16<T>::N;
  const int CV = C / EPV, Hp = H + 2, Wp = W + 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wp * CV) return;
  const int xp = idx / CV, cv = idx - xp * CV;
  const int xo = xp - 1, yo = (int)blockIdx.y - 1, b = blockIdx.z;
  Vec16<T> o;
  if (xo < 0 || yo < 0 || xo >= W || yo >= H) {
#pragma unroll
    for (int k = 0; k < EPV / 2; ++k) o.set2(k, 0.f, 0.f);
  } else {
    o.raw = *reinterpret_cast<const decltype(o.raw)*>(reinterpret_cast<const T*>(xin) + (((int64_t)b * H + yo) * W + xo) * C + cv * EPV);
  }
  *reinterpret_cast<decltype(o.raw)*>(reinterpret_cast<T*>(yout) + (((int64_t)b * Hp + blockIdx.y) * Wp + xp) * C + cv * EPV) = o.raw;
}

// Downsample (vqgan_blocks.py:109-126) = F.pad(x, (0,1,0,1)) + conv 3x3 stride 2: out[y][x] = sum w[ky][kx] x[2y+ky][2x+kx],
// which is the stride-1 "same" convolution s1 (1-pixel zero border) taken at the odd positions: out[y][x] = s1[2y+1][2x+1]
// (its bottom / right border supplies the asymmetric zero padding).  This kernel is the gather.
template <typename T>
__global__ __launch_bounds__(256) void subsample_odd_kernel(const void* xin, void* yout, int B, int H, int W, int C) {
  constexpr int EPV = Vec16<T>::N;
  const int CV = C / EPV, Ho = H / 2, Wo = W / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wo * CV) return;
  const int xo = idx / CV, cv = idx - xo * CV;
  const int yo = blockIdx.y, b = blockIdx.z;
  Vec16<T> o;
  o.raw = *reinterpret_cast<const decltype(o.raw)*>(reinterpret_cast<const T*>(xin) + (((int64_t)b * H + 2 * yo + 1) * W + 2 * xo + 1) * C + cv * EPV);
  *reinterpret_cast<decltype(o.raw)*>(reinterpret_cast<T*>(yout) + (((int64_t)b * Ho + yo) * Wo + xo) * C + cv * EPV) = o.raw;
}

// quant_conv (1x1, z_channels -> embed_dim = 4 -> 4) on the fp32 NCHW encoder output (MOVQ.encode, autoencoder.py:176-180)
__global__ void movq_quant_conv_kernel(const float* h, const float* wq, const float* bq, float* out, int B, int HW) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * HW) return;
  const int b = (int)(i / HW), pix = (int)(i - (int64_t)b * HW);
  float zi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) zi[c] = h[((int64_t)b * 4 + c) * HW + pix];
#pragma unroll
  for (int o = 0; o < 4; ++o)
    out[((int64_t)b * 4 + o) * HW + pix] = ((wq[o * 4] * zi[0] + wq[o * 4 + 1] * zi[1]) + (wq[o * 4 + 2] * zi[2] + wq[o * 4 + 3] * zi[3])) + bq[o];
}

int launch_movq_enc_prepare(const float* img, void* xin, int B, int H, int W, int Cpad, int dtype, hipStream_t s) {
  const int64_t total = (int64_t)B * (H + 2) * (W + 2);
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (dtype == K22_BF16) hipLaunchKernelGGL(movq_enc_prepare_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, img, xin, B, H, W, Cpad);
  else if (dtype == K22_F16) hipLaunchKernelGGL(movq_enc_prepare_kernel<f16_t>, dim3(nb), dim3(256), 0, s, img, xin, B, H, W, Cpad);
  else hipLaunchKernelGGL(movq_enc_prepare_kernel<float>, dim3(nb), dim3(256), 0, s, img, xin, B, H, W, Cpad);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_pad_copy(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s) {
  const int epv = dtype == K22_F32 ? 4 : 8;
  if (C % epv || H + 2 > 65535) return k22_set_error(K22_EINVAL, "pad_copy: bad shape");
  dim3 grid(((W + 2) * (C / epv) + 255) / 256, H + 2, B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(pad_copy_kernel<bf16_t>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  else if (dtype == K22_F16) hipLaunchKernelGGL(pad_copy_kernel<f16_t>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  else hipLaunchKernelGGL(pad_copy_kernel<float>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_subsample_odd(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s) {
  const int epv = dtype == K22_F32 ? 4 : 8;
  if (C % epv || (H & 1) || (W & 1) || H / 2 > 65535) return k22_set_error(K22_EINVAL, "subsample_odd: bad shape");
  dim3 grid(((W / 2) * (C / epv) + 255) / 256, H / 2, B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(subsample_odd_kernel<bf16_t>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  else if (dtype == K22_F16) hipLaunchKernelGGL(subsample_odd_kernel<f16_t>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  else hipLaunchKernelGGL(subsample_odd_kernel<float>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_movq_quant_conv(const float* h, const float* wq, const float* bq, float* out, int B, int HW, hipStream_t s) {
  const int64_t total = (int64_t)B * HW;
  hipLaunchKernelGGL(movq_quant_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h, wq, bq, out, B, HW);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

int launch_spatialnorm_apply(const SpatialNormParams& p, int dtype, hipStream_t s) {
  const int epv = dtype == K22_F32 ? 4 : 8;
  if (p.C % epv) return k22_set_error(K22_EINVAL, "spatialnorm: channel alignment");
  const int Hp = p.H + 2 * p.pad;
  if (Hp > 65535 || p.B > 65535) return k22_set_error(K22_EINVAL, "spatialnorm: tensor too large");
  const int R = p.shift >= 3 ? 8 : (1 << p.shift);
  if (p.shift < 0 || (p.W >> p.shift) > p.w0 || (p.H >> p.shift) > p.h0 || (p.W & ((1 << p.shift) - 1)))
    return k22_set_error(K22_EINVAL, "spatialnorm: W must be a multiple of 2^shift (zq is the latent nearest-upsampled by 2^shift)");
  dim3 grid((((p.W + R - 1) / R) * (p.C / epv) + 255) / 256, Hp, p.B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(spatialnorm_apply_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16) hipLaunchKernelGGL(spatialnorm_apply_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(spatialnorm_apply_kernel<float>, grid, dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_upsample2_pad(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s) {
  const int epv = dtype == K22_F32 ? 4 : 8;
  if (C % epv) return k22_set_error(K22_EINVAL, "upsample2_pad: channel alignment");
  const int Hp = 2 * H + 2, Wp = 2 * W + 2;
  if (Hp > 65535) return k22_set_error(K22_EINVAL, "upsample2_pad: tensor too large");
  dim3 grid((Wp * (C / epv) + 255) / 256, Hp, B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(upsample2_pad_kernel<bf16_t>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  else if (dtype == K22_F16) hipLaunchKernelGGL(upsample2_pad_kernel<f16_t>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  else hipLaunchKernelGGL(upsample2_pad_kernel<float>, grid, dim3(256), 0, s, x, y, B, H, W, C);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_softmax_rows(void* x, int64_t rows, int L, float scale, int dtype, hipStream_t s) {
  const int epv = dtype == K22_F32 ? 4 : 8;
  if (L % epv || rows <= 0 || rows > 0x7fffffff) return k22_set_error(K22_EINVAL, "softmax_rows: bad shape");
  if (dtype == K22_BF16) hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, s, x, L, scale);
  else if (dtype == K22_F16) hipLaunchKernelGGL(softmax_rows_kernel<f16_t>, dim3((unsigned)rows), dim3(256), 0, s, x, L, scale);
  else hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, s, x, L, scale);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_movq_prepare(const float* z, const float* wpq, const float* bpq, float* zq, void* xin, int B, int h, int w,
                        int Cpad, int dtype, hipStream_t s) {
  const int total = B * (h + 2) * (w + 2);
  if (dtype == K22_BF16) hipLaunchKernelGGL(movq_prepare_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, s, z, wpq, bpq, zq, xin, B, h, w, Cpad);
  else if (dtype == K22_F16) hipLaunchKernelGGL(movq_prepare_kernel<f16_t>, dim3((total + 255) / 256), dim3(256), 0, s, z, wpq, bpq, zq, xin, B, h, w, Cpad);
  else hipLaunchKernelGGL(movq_prepare_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, z, wpq, bpq, zq, xin, B, h, w, Cpad);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
int launch_to_uint8_nhwc(const float* x, unsigned char* y, int B, int C, int H, int W, hipStream_t s) {
  const int64_t total = (int64_t)B * C * H * W;
  int nb = (int)((total + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(to_uint8_nhwc_kernel, dim3(nb), dim3(256), 0, s, x, y, B, C, H, W);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
