// k22 — skinny-M weight-streaming GEMM and the fused split-K finish + LayerNorm of the prior transformer.
//
// Replaces, for M <= a few hundred rows: nn.Linear c_qkv / c_proj / c_fc / mlp.c_proj of the prior's ResidualAttentionBlock
// (kandinsky2/model/prior.py:57-83, 105-127: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))) and the LayerNorm in front of each
// (prior.py:48-54).  At bs = 1 the prior runs 2 x 81 = 162 token rows over 2.04 GB of bf16 weights per forward: every Linear is a
// weight stream (8-34 MB) against a 0.7-2.6 MB activation, so what bounds a launch is (a) HBM for the weights and (b) the L2 -> CU path
// (~56 B/clk/CU) for the ACTIVATION re-reads of 256 workgroups.  Rounds 3-5 ran these Linears through the square-tile GEMM kernels
// (igemm / gemm8: 16-21 us each + a 5-us split-K finish + a 6-us LayerNorm launch: 117 us per layer, 0.8 TB/s); this file is the
// structure the shape wants:
//   * BOTH operands are stored FRAGMENT-MAJOR, so that one MFMA fragment (32 rows x 8 k-values per lane-half) is 1 KB contiguous
//     and a wave loads it with ONE fully coalesced global_load_dwordx4 straight into the registers the MFMA reads:
//       W:  [N / 32][K / 64][k-step 0..3][lane][8]     (launch_stream_repack, taps = 1: written once per bind)
//       A:  [K / 64][MA = ceil(M / 32)][k-step][lane][8]   (written so by its producer: the finish / LayerNorm kernel below, this
//            kernel's EPI_AFRAG epilogue, the prior's attention kernel)
//     No LDS, no barrier and no ds_read in the main loop at all;
//   * the four waves of a workgroup split K inside every 64-wide chunk (wave w = k-step w), each holding the whole
//     (32 MT) x (32 NB) accumulator tile; a register ring D chunks deep keeps D x (MT + NB) KB per wave in flight (one in-order vmcnt
//     stream: weights - HBM, nt - and activations - L2 - share the ring depth);
//   * the four partial tiles are folded through the LDS in a fixed order (wave 0 + 1 + 2 + 3), then bias / activation, and the
//     result leaves in the layout its consumer reads: row-major T (qkv, read by the attention kernel), A-fragment order (c_fc ->
//     mlp.c_proj), or fp32 split-K partials [S][M][N] (c_proj, mlp.c_proj) for finish_ln_kernel;
//   * finish_ln_kernel = split-K finish + bias + fp32 residual-stream update + the NEXT LayerNorm + the fragment-major store of its
//     output, one workgroup per token row: three launches of the old path (splitk_reduce, residual epilogue, prior_layernorm) in one.
// 16-bit storage types only (bf16 / fp16).  fp32 accumulation everywhere; LayerNorm statistics two-pass in fp32 like the reference.
#include "kernels.h"
#include "skinny.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int xcd_remap_sk(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// element offset of (row m, column k) in an A-fragment tensor with MA m-atoms
__device__ __forceinline__ int64_t afrag_off(int m, int k, int MA) {
  return ((((int64_t)(k >> 6) * MA + (m >> 5)) * 4 + ((k >> 4) & 3)) * 64 + (m & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
}

template <typename T, int MT, int NB, int D, int DBG = 0>
__global__ __launch_bounds__(256, 1) void skinny_kernel(const SkinnyParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long ts[8];
  if (DBG & 32) { for (int i = 0; i < 8; ++i) ts[i] = 0; ts[0] = __builtin_amdgcn_s_memrealtime(); }
  const int nchunks = p.K >> 6;
  const int mtiles = (p.MA + MT - 1) / MT;
  const int ntiles = p.Npad / (32 * NB);
  int L = xcd_remap_sk(blockIdx.x, gridDim.x);
  const int mt = L % mtiles;
  L /= mtiles;
  const int nt = L % ntiles, z = L / ntiles;
  const int per = (nchunks + p.splitk - 1) / p.splitk;
  const int c0 = z * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
  const int a0 = mt * MT;

  const T* __restrict__ Af = reinterpret_cast<const T*>(p.Af);
  const T* __restrict__ Wf = reinterpret_cast<const T*>(p.Wf);
  // this lane's 16 bytes of (m-atom a0 + i, chunk c, k-step w): ap + c * MA * 2048 + i * 2048   (elements)
  const T* ap = Af + ((int64_t)(a0 * 4 + w) * 64 + lane) * 8;
  const T* wp = Wf + (((int64_t)(nt * NB) * nchunks) * 4 + w) * 512 + lane * 8;
  const int64_t a_cstride = (int64_t)p.MA * 2048, w_jstride = (int64_t)nchunks * 2048;
  int aoff[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) aoff[i] = (a0 + i < p.MA ? i : p.MA - 1 - a0) * 2048;   // atoms past the operand re-read its last one (results dropped)

  f32x16_t acc[MT][NB];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // this wave's bias values (weight rows 8 w + 4 h + {0..3} of every n-atom), fetched now: in the epilogue the load would be an exposed miss
  float4 bv[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = nt * 32 * NB + 32 * j + 8 * w + 4 * h;
    bv[j] = (p.bias != nullptr && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // register ring, D chunks deep.  Every load is issued unconditionally (chunk index clamped to the last one of the range): the
  // compiler's vmcnt bookkeeping sees one straight-line stream and keeps (D - 1) x (MT + NB) loads in flight behind each wait.
  Frag<T> ar[D][MT], wr[D][NB];
  auto load = [&](int c, Frag<T> (&a)[MT], Frag<T> (&b)[NB]) __attribute__((always_inline)) {
    if (c > c1 - 1) c = c1 - 1;
    const T* apc = ap + (int64_t)c * a_cstride;
    const T* wpc = wp + (int64_t)c * 2048;
#pragma unroll
    for (int j = 0; j < NB; ++j) if (!(DBG & 4) || c == c0) b[j].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wpc + j * w_jstride));
#pragma unroll
    for (int i = 0; i < MT; ++i) if (!(DBG & 2) || c == c0) a[i].v = *reinterpret_cast<const u32x4_t*>(apc + aoff[i]);
  };
  if (c0 < c1) {
#pragma unroll
    for (int d = 0; d < D; ++d) load(c0 + d, ar[d], wr[d]);
    int c = c0;
    if (DBG & 32) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * (MT + NB)) : "memory"); ts[1] = __builtin_amdgcn_s_memrealtime(); }
    // steady state: no branch inside, and the refill of a ring slot stays BEHIND the MFMAs that read it (sched_barrier): hoisted above
    // them the compiler renames the slot and ends every pass over the ring with a full vmcnt(0) drain
    for (; c + D <= c1; c += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            if (DBG & 1) acc[i][j][0] += __builtin_bit_cast(float, wr[u][j].v[0] ^ ar[u][i].v[0]);   // measurement only: no MFMA
            else mma_atom(acc[i][j], wr[u][j], ar[u][i]);   // C^T: rows = weight rows n, columns = token rows m
          }
        __builtin_amdgcn_sched_barrier(0);
        load(c + u + D, ar[u], wr[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int rem = c1 - c;   // < D chunks left, already in the ring
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
      if (u < rem) {   // uniform
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) mma_atom(acc[i][j], wr[u][j], ar[u][i]);
      }
    }
  }

  if (DBG & 32) { asm volatile("s_nop 0" ::: "memory"); ts[2] = __builtin_amdgcn_s_memrealtime(); }
  // ---- fold the four k-steps in a fixed order: wave q ends up with accumulator registers 4q .. 4q+3 of every atom = weight rows
  // 8q + 4h + {0..3} of token column l31 -------------------------------------------------------------------------------------------
  constexpr int NAT = MT * NB;
  constexpr int GA = NAT < 8 ? NAT : (NAT % 6 == 0 ? 6 : (NAT % 8 == 0 ? 8 : NAT));   // atoms folded per pass: 16 KB of scratch each, <= 128 KB
  static_assert(NAT % GA == 0 && GA * 16 <= 160, "fold scratch");
  f32x4_t* scr = reinterpret_cast<f32x4_t*>(smem);
  const int n_atom0 = nt * 32 * NB;
#pragma unroll
  for (int g0 = 0; g0 < NAT; g0 += GA) {
    f32x4_t own[GA];
#pragma unroll
    for (int ga = 0; ga < GA; ++ga) {
      const int at = g0 + ga;
      const f32x16_t& c = acc[at / NB][at % NB];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_t v = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
        if (q != w) scr[((w * 4 + q) * GA + ga) * 64 + lane] = v;
        else own[ga] = v;
      }
    }
    __syncthreads();
    if ((DBG & 32) && g0 == 0) ts[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int ga = 0; ga < GA; ++ga) {
      const int at = g0 + ga;
      const int i = at / NB, j = at % NB;
      f32x4_t t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int src = 0; src < 4; ++src) {
        f32x4_t v;
        if (src == w) v = own[ga];
        else v = scr[((src * 4 + w) * GA + ga) * 64 + lane];
        if (src == 0) t = v;
        else t += v;
      }
      const int m = (a0 + i) * 32 + l31, n = n_atom0 + 32 * j + 8 * w + 4 * h;
      if (a0 + i >= p.MA || m >= p.M || n >= p.N) continue;
      if (p.epi == SK_EPI_PARTIAL) {
        *reinterpret_cast<f32x4_t*>(p.partial + ((int64_t)z * p.M + m) * p.N + n) = t;
        continue;
      }
      float v[4] = {t.x + bv[j].x, t.y + bv[j].y, t.z + bv[j].z, t.w + bv[j].w};
      if (p.act == K22_ACT_GELU) {   // uniform
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
      } else if (p.act != K22_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
      }
      uint2 o;
      o.x = pack2<T>(v[0], v[1]);
      o.y = pack2<T>(v[2], v[3]);
      T* out = reinterpret_cast<T*>(p.out);
      if (p.epi == SK_EPI_ROWMAJOR) *reinterpret_cast<uint2*>(out + (int64_t)m * p.ldo + n) = o;
      else *reinterpret_cast<uint2*>(out + afrag_off(m, n, p.MA)) = o;   // SK_EPI_AFRAG: the consumer's K index is this launch's n
    }
    if (g0 + GA < NAT) __syncthreads();   // the scratch is rewritten by the next pass
  }
  if ((DBG & 32) && p.trace != nullptr) {
    ts[4] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[5] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); ts[6] = xcc; for (int i = 0; i < 8; ++i) p.trace[(int64_t)blockIdx.x * 8 + i] = ts[i]; }
  }
}

// ---- split-K finish + bias + residual-stream update + LayerNorm + fragment-major store ------------------------------------------------
//   x[m][:] += bias + sum_s partial[s][m][:]          (skipped when partial == null: plain LayerNorm of x)
//   y = LayerNorm(x[m][:]) * g + b  ->  T, A-fragment order (skipped when g == null)
// One workgroup per token row; a thread owns 8 consecutive columns (one 16-byte piece of the fragment tensor).  N <= 2048, N % 8 == 0.
template <typename T>
__global__ __launch_bounds__(256) void finish_ln_kernel(const FinishLnParams p) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n = tid * 8;
  const bool on = n < p.N;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  float* xr = p.x + (int64_t)m * p.ldx;
  if (on) {
    const float4 x0 = *reinterpret_cast<const float4*>(xr + n), x1 = *reinterpret_cast<const float4*>(xr + n + 4);
    if (p.partial != nullptr) {
      const int64_t tot = (int64_t)p.M * p.N;
      const float* pp = p.partial + (int64_t)m * p.N + n;
      // every load of the row is issued before the first add (splitk <= 8; absent splits re-read the last one and are not added)
      float4 q0[8], q1[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int ss = s < p.splitk ? s : p.splitk - 1;
        q0[s] = *reinterpret_cast<const float4*>(pp + ss * tot);
        q1[s] = *reinterpret_cast<const float4*>(pp + ss * tot + 4);
      }
      float4 a0 = q0[0], a1 = q1[0];
#pragma unroll
      for (int s = 1; s < 8; ++s) {
        if (s < p.splitk) {
          a0.x += q0[s].x; a0.y += q0[s].y; a0.z += q0[s].z; a0.w += q0[s].w;
          a1.x += q1[s].x; a1.y += q1[s].y; a1.z += q1[s].z; a1.w += q1[s].w;
        }
      }
      if (p.bias != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
        a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
      }
      v[0] = a0.x + x0.x; v[1] = a0.y + x0.y; v[2] = a0.z + x0.z; v[3] = a0.w + x0.w;
      v[4] = a1.x + x1.x; v[5] = a1.y + x1.y; v[6] = a1.z + x1.z; v[7] = a1.w + x1.w;
      *reinterpret_cast<float4*>(xr + n) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(xr + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
  }
  if (p.g == nullptr) return;   // uniform
  float gg[8], bb[8];   // fetched before the reductions' barriers
  if (on) {
    const float4 g0 = *reinterpret_cast<const float4*>(p.g + n), g1 = *reinterpret_cast<const float4*>(p.g + n + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.b + n), b1 = *reinterpret_cast<const float4*>(p.b + n + 4);
    gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
    bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
  }
  float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / p.N;
  float q = 0.f;
  if (on) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / p.N + p.eps);
  if (!on) return;
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = (v[e] - mean) * rstd * gg[e] + bb[e];
  const u32x4_t o = {pack2<T>(y[0], y[1]), pack2<T>(y[2], y[3]), pack2<T>(y[4], y[5]), pack2<T>(y[6], y[7])};
  *reinterpret_cast<u32x4_t*>(reinterpret_cast<T*>(p.yfrag) + afrag_off(m, n, p.MA)) = o;
}

// row-major T [M][lda] -> A-fragment order (the unit tests' and the first layer's way in): one 16-byte piece per thread
template <typename T>
__global__ __launch_bounds__(256) void afrag_pack_kernel(const T* __restrict__ A, int64_t lda, T* __restrict__ out, int M, int K, int MA) {
  const int64_t pieces = (int64_t)M * (K >> 3);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / (K >> 3)), k = (int)(i - (int64_t)m * (K >> 3)) * 8;
    *reinterpret_cast<u32x4_t*>(out + afrag_off(m, k, MA)) = *reinterpret_cast<const u32x4_t*>(A + (int64_t)m * lda + k);
  }
}

template <typename T, int MT, int NB, int D, int DBG = 0>
int launch_skinny_cfg(const SkinnyParams& p, hipStream_t st) {
  static LdsAttrGuard guard;
  constexpr int NATL = MT * NB;
  constexpr int lds = (NATL < 8 ? NATL : (NATL % 6 == 0 ? 6 : (NATL % 8 == 0 ? 8 : NATL))) * 16 * 1024;
  if (int rc = k22_ensure_lds_attr(guard, reinterpret_cast<const void*>(&skinny_kernel<T, MT, NB, D, DBG>), lds, __FILE__, __LINE__)) return rc;
  const int mtiles = (p.MA + MT - 1) / MT, ntiles = p.Npad / (32 * NB);
  hipLaunchKernelGGL((skinny_kernel<T, MT, NB, D, DBG>), dim3(mtiles * ntiles * p.splitk), dim3(256), lds, st, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

template <typename T>
int launch_skinny_typed(const SkinnyParams& p, int mt, int nb, hipStream_t st) {
#ifdef K22_SKINNY_DEBUG   // measurement-only variants (wrong results): K22_SK_DBG bits 1 = no MFMA, 2 = A loaded once, 4 = W loaded once; 8: ring depth 12; 16: depth 4
  if (const char* e = getenv("K22_SK_DBG")) {
    const int d = atoi(e);
    if (mt == 3 && nb == 2) {
      if (d == 1) return launch_skinny_cfg<T, 3, 2, 8, 1>(p, st);
      if (d == 2) return launch_skinny_cfg<T, 3, 2, 8, 2>(p, st);
      if (d == 4) return launch_skinny_cfg<T, 3, 2, 8, 4>(p, st);
      if (d == 6) return launch_skinny_cfg<T, 3, 2, 8, 6>(p, st);
      if (d == 7) return launch_skinny_cfg<T, 3, 2, 8, 7>(p, st);
      if (d == 8) return launch_skinny_cfg<T, 3, 2, 12, 0>(p, st);
      if (d == 16) return launch_skinny_cfg<T, 3, 2, 4, 0>(p, st);
      if (d == 32) return launch_skinny_cfg<T, 3, 2, 8, 32>(p, st);
    }
  }
#endif
  if (mt == 6 && nb == 1) return launch_skinny_cfg<T, 6, 1, 6>(p, st);
  if (mt == 3 && nb == 2) return launch_skinny_cfg<T, 3, 2, 8>(p, st);
  if (mt == 3 && nb == 4) return launch_skinny_cfg<T, 3, 4, 8>(p, st);
  if (mt == 3 && nb == 1) return launch_skinny_cfg<T, 3, 1, 8>(p, st);
  if (mt == 2 && nb == 2) return launch_skinny_cfg<T, 2, 2, 8>(p, st);
  if (mt == 2 && nb == 1) return launch_skinny_cfg<T, 2, 1, 8>(p, st);
  if (mt == 1 && nb == 2) return launch_skinny_cfg<T, 1, 2, 8>(p, st);
  return k22_set_error(K22_EINVAL, "skinny: (mt, nb) must be one of (6,1) (3,4) (3,2) (3,1) (2,2) (2,1) (1,2)");
}

}  // namespace

bool skinny_supported(const SkinnyParams& p, int dtype) {
  if (dtype != K22_BF16 && dtype != K22_F16) return false;
  if (p.K % 64 || p.Npad % 64 || p.N % 4 || p.Npad < p.N || p.M < 1 || p.MA != (p.M + 31) / 32 || p.splitk < 1 || p.splitk > p.K / 64) return false;
  if (p.epi == SK_EPI_ROWMAJOR && (p.ldo % 4)) return false;
  if (p.epi == SK_EPI_AFRAG && (p.N % 64)) return false;
  if (p.epi != SK_EPI_PARTIAL && p.splitk != 1) return false;
  if (p.epi == SK_EPI_PARTIAL && p.partial == nullptr) return false;
  return true;
}

// default tile of a problem: the shapes of the prior were measured on an MI355X (profiles/r06_skinny.txt); everything else by the same rule
// of thumb (256-ish workgroups; split-K only where a finish kernel runs anyway)
void skinny_default_cfg(const SkinnyParams& p, int* mt, int* nb) {
  static const char* e = getenv("K22_SKINNY_CFG");   // "mt,nb": measurement override
  int a = 0, b = 0;
  if (e && sscanf(e, "%d,%d", &a, &b) == 2) { *mt = a; *nb = b; return; }
  *mt = p.MA >= 3 ? 3 : p.MA;   // m-tiles of <= 96 rows x 64 weight rows: the pair of workgroups of an n-tile shares its weights through the XCD's L2
  *nb = 2;
}

int launch_skinny(const SkinnyParams& p, int dtype, int mt, int nb, hipStream_t st) {
  if (!skinny_supported(p, dtype)) return k22_set_error(K22_EINVAL, "skinny: unsupported problem");
  if (mt <= 0 || nb <= 0) skinny_default_cfg(p, &mt, &nb);
  if (p.Npad % (32 * nb)) return k22_set_error(K22_EINVAL, "skinny: Npad must be a multiple of the n-tile");
  return dtype == K22_BF16 ? launch_skinny_typed<bf16_t>(p, mt, nb, st) : launch_skinny_typed<f16_t>(p, mt, nb, st);
}

int launch_finish_ln(const FinishLnParams& p, int dtype, hipStream_t st) {
  if ((dtype != K22_BF16 && dtype != K22_F16) || p.N > 2048 || p.N % 8 || p.M < 1 || (p.ldx & 3) || p.splitk > 8 || (p.g != nullptr && (p.b == nullptr || p.yfrag == nullptr)))
    return k22_set_error(K22_EINVAL, "finish_ln: unsupported problem (16-bit types, N <= 2048, N % 8 == 0)");
  if (dtype == K22_BF16) hipLaunchKernelGGL(finish_ln_kernel<bf16_t>, dim3(p.M), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(finish_ln_kernel<f16_t>, dim3(p.M), dim3(256), 0, st, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

int launch_afrag_pack(const void* A, int64_t lda, void* out, int M, int K, int dtype, hipStream_t st) {
  if ((dtype != K22_BF16 && dtype != K22_F16) || K % 64 || lda % 8 || M < 1) return k22_set_error(K22_EINVAL, "afrag_pack: bad arguments");
  const int64_t pieces = (int64_t)M * (K / 8);
  const int nb = (int)((pieces + 255) / 256 < 2048 ? (pieces + 255) / 256 : 2048);
  hipLaunchKernelGGL(afrag_pack_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(A), lda, reinterpret_cast<bf16_t*>(out), M, K, (M + 31) / 32);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

size_t afrag_bytes(int M, int K) { return (size_t)((M + 31) / 32) * 32 * K * 2; }
