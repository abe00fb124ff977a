// k22 — implicit-GEMM MFMA kernel: 3x3 convolution (zero-bordered NHWC input) and plain GEMM
// (1x1 conv, qkv / proj / linear) through ONE kernel body.
//
// Replaces, on the reference hot path: nn.Conv2d 3x3 (unet.py:152,180,188,426,562), Conv2d 1x1
// skip (unet.py:191), Conv1d k=1 qkv/encoder_kv/proj_out (unet.py:251,257,258) and nn.Linear.
//
// Structure (per 256-thread workgroup = 4 waves as 2(M) x 2(N)):
//   * block tile BM x BN, K tile = one 128-byte row per operand row (64 bf16 / 32 fp32), never
//     straddling a filter tap (Cin % 64 == 0), so a conv K tile is a contiguous channel slice of
//     one shifted pixel: address = pixel_base[m] + tap_offset + c0   (no bounds checks: the
//     producer wrote a zero border).
//   * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4: no staging VGPRs, no ds_write
//     pass), STAGES buffers deep with a COUNTED vmcnt so STAGES-2 tiles stay in flight across the
//     single raw s_barrier of each K tile:
//       iteration k:  s_waitcnt vmcnt((STAGES-2)*CH)  -> this wave's part of tile k has landed
//                     s_barrier                        -> everybody's part landed; tile k-1's buffer is free
//                     issue tile k+STAGES-1 into that buffer ; MFMAs on tile k
//   * the LDS image of one wave-instruction is lane-linear (base + lane*16 B): 8 rows x 128 B, so
//     the chunk-XOR swizzle (common.h) is applied to the per-lane SOURCE address and to the reads;
//     fragment ds_read_b128s are bank-conflict free.
//   * the MFMA is issued "transposed" (weights as the A operand, pixels as B): the 32x32 C tile then
//     has one PIXEL per lane column and 4 consecutive OUTPUT CHANNELS in consecutive accumulator
//     registers, so the epilogue moves 8 B (bf16) / 16 B (fp32) per store instead of one element.
//   * workgroups are renumbered so that the blocks an XCD runs concurrently share operand panels in
//     that XCD's L2 (hardware places block b on XCD b % 8): every weight byte is pulled from
//     HBM/Infinity-Fabric by one XCD, pixels by the few XCDs that hold their m-tile group.
//   * fp32 accumulate; epilogue fuses bias, residual add and activation, or writes fp32 split-K
//     partials that splitk_reduce_kernel finishes.
#include "kernels.h"
#include <stdlib.h>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 4 consecutive output channels of one pixel
template <typename T> __device__ __forceinline__ void store4(T* dst, const float* v);
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, const float* v) {
  uint2 w;
  w.x = pack2_bf16(v[0], v[1]);
  w.y = pack2_bf16(v[2], v[3]);
  *reinterpret_cast<uint2*>(dst) = w;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* dst, const float* v) {
  uint2 w;
  w.x = pack2_f16(v[0], v[1]);
  w.y = pack2_f16(v[2], v[3]);
  *reinterpret_cast<uint2*>(dst) = w;
}
template <> __device__ __forceinline__ void store4<float>(float* dst, const float* v) {
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<x3_t>(x3_t* dst, const float* v) { store4<float>(reinterpret_cast<float*>(dst), v); }
template <> __device__ __forceinline__ void store4<x2_t>(x2_t* dst, const float* v) { store4<float>(reinterpret_cast<float*>(dst), v); }
template <typename T> __device__ __forceinline__ void load4f(const T* src, float* v);
template <> __device__ __forceinline__ void load4f<bf16_t>(const bf16_t* src, float* v) {
  const uint2 r = *reinterpret_cast<const uint2*>(src);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4f<f16_t>(const f16_t* src, float* v) {
  const uint2 r = *reinterpret_cast<const uint2*>(src);
  unpack2_f16(r.x, v[0], v[1]); unpack2_f16(r.y, v[2], v[3]);
}
template <> __device__ __forceinline__ void load4f<float>(const float* src, float* v) {
  const float4 r = *reinterpret_cast<const float4*>(src);
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}

template <> __device__ __forceinline__ void load4f<x3_t>(const x3_t* src, float* v) { load4f<float>(reinterpret_cast<const float*>(src), v); }
template <> __device__ __forceinline__ void load4f<x2_t>(const x2_t* src, float* v) { load4f<float>(reinterpret_cast<const float*>(src), v); }

// Logical block index of hardware block `bid` (XCD = bid % 8): XCD x gets the contiguous range
// [start_x, start_x + count_x) of logical indices, in dispatch order.  Bijective for any nblocks.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ARAW (K22_F16X3 only): the A operand is plain fp32 rows, split into fp16 halves while its fragments are read (p.a_raw).
template <typename T, int BM, int BN, int STAGES, bool ARAW = false>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int MI = BM / 64, NI = BN / 64;     // 32x32 atoms per wave along M / N
  constexpr int A_CH = BM / 32, B_CH = BN / 32, CH = A_CH + B_CH;  // LDS-DMA instructions per wave per K tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
  constexpr int GM = 8;                         // m-tiles per rasterisation group
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> (m-tile, n-tile, k-split): groups of GM m-tiles x all n-tiles, k-split outermost ----
  const int gx = (p.M + BM - 1) / BM, gy = (p.N + BN - 1) / BN;
  int L = p.xcd_remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy;
  const int bz = L / per_z;
  L -= bz * per_z;
  const int grp = L / (GM * gy);
  const int first_m = grp * GM;
  const int gsz = gx - first_m < GM ? gx - first_m : GM;
  const int lin = L - grp * GM * gy;
  const int bx = first_m + lin % gsz, by = lin / gsz;
  const int m0 = bx * BM, n0 = by * BN;

  const bool conv = (p.taps == 9);
  const T* __restrict__ A0 = reinterpret_cast<const T*>(p.A0);
  const T* __restrict__ A1 = reinterpret_cast<const T*>(p.A1);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);

  // ---- loader geometry: thread -> (row = tid/8 + 32*i, physical chunk = tid%8) --------------
  const int lrow = tid >> 3, lpos = tid & 7;
  const int lchunk = lpos ^ ((lrow >> 1) & 7);  // logical (source) chunk that lands at this position
  int64_t aoff0[A_CH], aoff1[A_CH], boff[B_CH];
  const int ldb = p.taps * p.Kc;
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int m = m0 + lrow + 32 * i;
    if (m > p.M - 1) m = p.M - 1;
    if (conv) {
      const int hw = p.H * p.W;
      const int b = m / hw, rem = m - b * hw;
      const int y = rem / p.W, x = rem - y * p.W;
      aoff0[i] = ((int64_t)(b * (p.H + 2) + y) * (p.W + 2) + x) * p.Kc;
      aoff1[i] = 0;
    } else {
      aoff0[i] = (int64_t)m * p.lda0;
      aoff1[i] = (int64_t)m * p.lda1;
    }
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    int n = n0 + lrow + 32 * i;
    if (n > p.Npad - 1) n = p.Npad - 1;
    boff[i] = (int64_t)n * ldb;
  }
  const int kt_per_tap = p.Kc / BK;
  const int nkt = p.taps * kt_per_tap;
  int kt0 = 0, kt1 = nkt;
  if (p.splitk > 1) {
    const int per = (nkt + p.splitk - 1) / p.splitk;
    kt0 = bz * per;
    kt1 = kt0 + per < nkt ? kt0 + per : nkt;
  }

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  const int64_t lck = lchunk * EPC;
  const int wave_row0 = wave * 8;  // rows [8*wave + 32*i, +8) of a tile are filled by this wave's i-th DMA

#define K22_ISSUE(KT, BUFI)                                                                              \
  {                                                                                                      \
    int kt_ = (KT);                                                                                      \
    if (kt_ > kt1 - 1) kt_ = kt1 - 1; /* past-the-end slots re-read the last tile: uniform vmcnt counting */ \
    const int tap_ = kt_ / kt_per_tap;                                                                   \
    const int k0_ = (kt_ - tap_ * kt_per_tap) * BK;                                                      \
    const bool second_ = (!conv) && (k0_ >= p.K0);                                                       \
    const T* src_ = second_ ? A1 : A0;                                                                   \
    const int ty_ = tap_ / 3, tx_ = tap_ - ty_ * 3;                                                      \
    const int64_t add_ = (conv ? (int64_t)(ty_ * (p.W + 2) + tx_) * p.Kc + k0_ : (int64_t)(second_ ? k0_ - p.K0 : k0_)) + lck; \
    char* As_ = smem + (BUFI) * BUF + wave_row0 * 128;                                                   \
    _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + (second_ ? aoff1[i] : aoff0[i]) + add_), \
                                         (__attribute__((address_space(3))) void*)(As_ + i * 4096), 16, 0, 0); \
    const int64_t badd_ = (int64_t)tap_ * p.Kc + k0_ + lck;                                              \
    char* Bs_ = As_ + A_BYTES;                                                                           \
    _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wp + boff[i] + badd_), \
                                         (__attribute__((address_space(3))) void*)(Bs_ + i * 4096), 16, 0, 0); \
  }

  if (kt0 < kt1) {
#pragma unroll
    for (int s_ = 0; s_ < STAGES - 1; ++s_) K22_ISSUE(kt0 + s_, s_);
    int cur = 0, fill = STAGES - 1;
    for (int kt = kt0; kt < kt1; ++kt) {
      wait_vmcnt<(STAGES - 2) * CH>();
      raw_barrier();
      K22_ISSUE(kt + STAGES - 1, fill);
      {
        const char* As = smem + cur * BUF;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          FragA<T> a[MI];
          Frag<T> b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ld_frag_a<ARAW, T>(a[mi], As, wm * (BM / 2) + mi * 32 + l31, ks, h);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) ld_frag(b[ni], Bs, wn * (BN / 2) + ni * 32 + l31, ks, h);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], b[ni], a[mi]);  // C^T: rows = n, cols = m
        }
      }
      cur = (cur + 1 == STAGES) ? 0 : cur + 1;
      fill = (fill + 1 == STAGES) ? 0 : fill + 1;
    }
    wait_vmcnt<0>();  // drain the redundant tail DMAs before the LDS is released
  }
#undef K22_ISSUE

  // ---- epilogue: lane = pixel m, registers 4j..4j+3 = channels n0 + 8j + 4h + {0..3} -----------------
  const int mbase = m0 + wm * (BM / 2), nbase = n0 + wn * (BN / 2);
  const bool vec_ok = ((p.N & 3) == 0) && ((p.ldo & 3) == 0) && ((p.ldr & 3) == 0);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  float* part = p.splitk > 1 ? p.partial + (int64_t)bz * p.M * p.N : nullptr;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mbase + mi * 32 + l31;
    if (m >= p.M) continue;
    int64_t nchw_base = 0;
    if (p.out_mode == IG_OUT_NCHW_F32) {
      const int hw = p.H * p.W;
      const int b = m / hw, rem = m - b * hw;
      nchw_base = (int64_t)b * p.N * hw + rem;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nbase + ni * 32 + 8 * j + 4 * h;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc_unscale<T>(acc[mi][ni][4 * j + e]);
        if (part != nullptr) {
          if (vec_ok) {
            *reinterpret_cast<float4*>(part + (int64_t)m * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) part[(int64_t)m * p.N + n + e] = v[e];
          }
          continue;
        }
        if (vec_ok) {
          if (p.bias != nullptr) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (res != nullptr) {
            float rv[4];
            if (p.res_f32) load4f<float>(reinterpret_cast<const float*>(p.residual) + (int64_t)m * p.ldr + n, rv);
            else load4f<T>(res + (int64_t)m * p.ldr + n, rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
          if (p.out_mode == IG_OUT_ROWMAJOR) {
            store4<T>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + n, v);
          } else if (p.out_mode == IG_OUT_ROWMAJOR_F32) {
            store4<float>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n, v);
          } else if (p.out_mode == IG_OUT_QKV) {
            const int C = p.N / 3, heads = C >> 6;
            const int which = n / C, c = n - which * C;
            const int head = c >> 6, d = c & 63;
            const int b = m / p.att_T, t = m - b * p.att_T;
            if (which == 0) {
              store4<T>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + c, v);
            } else if (which == 1) {
              store4<T>(reinterpret_cast<T*>(p.kall) + ((int64_t)(b * heads + head) * p.att_Tkp + p.att_S + t) * 64 + d, v);
            } else {
              T* vt = reinterpret_cast<T*>(p.vtall) + ((int64_t)(b * heads + head) * 64 + d) * p.att_Tkp + p.att_S + t;
#pragma unroll
              for (int e = 0; e < 4; ++e) vt[(int64_t)e * p.att_Tkp] = from_f32<T>(v[e]);
            }
          } else {
            const int64_t hw = (int64_t)p.H * p.W;
#pragma unroll
            for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(p.out)[nchw_base + (int64_t)(n + e) * hw] = v[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) continue;
            float u = v[e];
            if (p.bias != nullptr) u += p.bias[n + e];
            if (res != nullptr) u += p.res_f32 ? reinterpret_cast<const float*>(p.residual)[(int64_t)m * p.ldr + n + e] : to_f32(res[(int64_t)m * p.ldr + n + e]);
            u = apply_act(u, p.act);
            if (p.out_mode == IG_OUT_ROWMAJOR) {
              reinterpret_cast<T*>(p.out)[(int64_t)m * p.ldo + n + e] = from_f32<T>(u);
            } else if (p.out_mode == IG_OUT_ROWMAJOR_F32) {
              reinterpret_cast<float*>(p.out)[(int64_t)m * p.ldo + n + e] = u;
            } else {
              reinterpret_cast<float*>(p.out)[nchw_base + (int64_t)(n + e) * p.H * p.W] = u;
            }
          }
        }
      }
  }
}

// Finishes a split-K launch: out = act(sum_s partial[s] + bias + residual).  4 channels per thread.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmParams p) {
  const int64_t total = (int64_t)p.M * p.N;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  if (((p.N & 3) == 0) && ((p.ldo & 3) == 0) && ((p.ldr & 3) == 0)) {
    const int nq = p.N >> 2;
    const int64_t quads = total >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
      const int m = (int)(i / nq), n = (int)(i - (int64_t)m * nq) * 4;
      float4 a = *reinterpret_cast<const float4*>(p.partial + i * 4);
      for (int s = 1; s < p.splitk; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(p.partial + (int64_t)s * total + i * 4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      float v[4] = {a.x, a.y, a.z, a.w};
      if (p.bias != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (p.bias2 != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias2 + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (res != nullptr) {
        float rv[4];
        if (p.res_f32) load4f<float>(reinterpret_cast<const float*>(p.residual) + (int64_t)m * p.ldr + n, rv);
        else load4f<T>(res + (int64_t)m * p.ldr + n, rv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
      if (p.out_mode == IG_OUT_ROWMAJOR) {
        store4<T>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + n, v);
      } else if (p.out_mode == IG_OUT_ROWMAJOR_F32) {
        store4<float>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n, v);
      } else if (p.out_mode == IG_OUT_QKV) {
        // same destinations as igemm_kernel's epilogue: q row-major, k / v^T behind the context keys of the attention operands
        const int C = p.N / 3, heads = C >> 6;
        const int which = n / C, c = n - which * C;
        const int head = c >> 6, d = c & 63;
        const int b = m / p.att_T, t = m - b * p.att_T;
        if (which == 0) {
          store4<T>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + c, v);
        } else if (which == 1) {
          store4<T>(reinterpret_cast<T*>(p.kall) + ((int64_t)(b * heads + head) * p.att_Tkp + p.att_S + t) * 64 + d, v);
        } else {
          T* vt = reinterpret_cast<T*>(p.vtall) + ((int64_t)(b * heads + head) * 64 + d) * p.att_Tkp + p.att_S + t;
#pragma unroll
          for (int e = 0; e < 4; ++e) vt[(int64_t)e * p.att_Tkp] = from_f32<T>(v[e]);
        }
      } else {
        const int hw = p.H * p.W;
        const int b = m / hw, rem = m - b * hw;
#pragma unroll
        for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(p.out)[((int64_t)b * p.N + n + e) * hw + rem] = v[e];
      }
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.N), n = (int)(i - (int64_t)m * p.N);
    float v = 0.f;
    for (int s = 0; s < p.splitk; ++s) v += p.partial[(int64_t)s * total + i];
    if (p.bias != nullptr) v += p.bias[n];
    if (p.bias2 != nullptr) v += p.bias2[n];
    if (res != nullptr) v += p.res_f32 ? reinterpret_cast<const float*>(p.residual)[(int64_t)m * p.ldr + n] : to_f32(res[(int64_t)m * p.ldr + n]);
    v = apply_act(v, p.act);
    if (p.out_mode == IG_OUT_ROWMAJOR) {
      reinterpret_cast<T*>(p.out)[(int64_t)m * p.ldo + n] = from_f32<T>(v);
    } else if (p.out_mode == IG_OUT_ROWMAJOR_F32) {
      reinterpret_cast<float*>(p.out)[(int64_t)m * p.ldo + n] = v;
    } else {
      const int hw = p.H * p.W;
      const int b = m / hw, rem = m - b * hw;
      reinterpret_cast<float*>(p.out)[((int64_t)b * p.N + n) * hw + rem] = v;
    }
  }
}

// Split-K finish + GroupNorm partial sums.  Workgroup = 16 consecutive output rows (never straddling an image:
// HW % 16 == 0) x 64 channels; thread = one row x 4 consecutive channels, so the splitk partial loads of a thread
// are independent and the grid is (M/16) x (N/64) workgroups.  stats[row_block][n] = (sum, sumsq) of the
// stored values, reduced over the 16 rows in a fixed order through LDS.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(const IgemmParams p) {
  __shared__ float red[16][64][2];
  const int64_t total = (int64_t)p.M * p.N;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  const int r = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const int m = blockIdx.x * 16 + r, n = blockIdx.y * 64 + cq * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool ok = m < p.M && n < p.N;
  if (ok) {
    const int64_t i = (int64_t)m * p.N + n;
    float4 a = *reinterpret_cast<const float4*>(p.partial + i);
#pragma unroll 4
    for (int s = 1; s < p.splitk; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(p.partial + (int64_t)s * total + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    if (p.bias != nullptr) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.bias2 != nullptr) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias2 + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (res != nullptr) {
      float rv[4];
      if (p.res_f32) load4f<float>(reinterpret_cast<const float*>(p.residual) + (int64_t)m * p.ldr + n, rv);
      else load4f<T>(res + (int64_t)m * p.ldr + n, rv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += rv[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
    if (p.out_mode == IG_OUT_ROWMAJOR) {
      store4<T>(reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + n, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = to_f32(from_f32<T>(v[e]));
    } else {
      store4<float>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n, v);
    }
  }
  if (p.stats == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[r][cq * 4 + e][0] = ok ? v[e] : 0.f;
    red[r][cq * 4 + e][1] = ok ? v[e] * v[e] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int ch = threadIdx.x >> 1, which = threadIdx.x & 1;
    float a = 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) a += red[rr][ch][which];
    const int nn = blockIdx.y * 64 + ch;
    if (nn < p.N) p.stats[((int64_t)blockIdx.x * p.N + nn) * 2 + which] = a;
  }
}

// ---- host side ------------------------------------------------------------------------
struct IgemmPlan { int halo, bm, bn, splitk, stages; };  // halo: conv3_halo.hip kernel (bm = 256 / 128, bn = 128)

static int g_stages_override = -1;
static int g_xcd_remap = 1;
static int g_conv_algo = 0;
void igemm_set_default_stages(int v) { g_stages_override = (v >= 2 && v <= 4) ? v : -1; }
void igemm_set_xcd_remap(int v) { g_xcd_remap = v ? 1 : 0; }
#ifdef K22_DEBUG_VARIANTS
void igemm_set_conv_algo(int v) { g_conv_algo = ((v >= 0 && v <= 9 && v != 4 && v != 5) || (v >= 11 && v <= 14) || v == 20) ? v : 0; }
#else   // 8, 9, 13, 14 are measurement-only kernels (wrong results): not reachable in a release build
void igemm_set_conv_algo(int v) { g_conv_algo = ((v >= 0 && v <= 7 && v != 4 && v != 5) || v == 11 || v == 12 || v == 20) ? v : 0; }
#endif
static int g_gemm_algo = 0;   // 0 = generic igemm_kernel, 10 = gemm8_kernel where it applies (unit tests / kernel benches)
void igemm_set_gemm_algo(int v) { g_gemm_algo = (v == 10 || v == 20) ? v : 0; }

static int g_default_stages() {
  static int v = -1;
  if (g_stages_override >= 0) return g_stages_override;
  if (v < 0) {
    const char* e = getenv("K22_IGEMM_STAGES");  // 2..4 = LDS-DMA pipeline depth
    v = e ? atoi(e) : 2;
    if (v < 2 || v > 4) v = 2;
  }
  return v;
}

static IgemmPlan igemm_plan(const IgemmParams& p, int dtype) {
  const int BK = k22_bk(dtype);
  const int nkt = p.taps * (p.Kc / BK);
  IgemmPlan pl;
  pl.halo = 0;
  pl.stages = (p.stages >= 2 && p.stages <= 4) ? p.stages : g_default_stages();
  auto blocks = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  // ---- weight-streaming small-M kernel (stream_gemm.hip): p.algo == 20 (tuner candidate) or the conv_algo / gemm_algo option --
  if (p.algo == 20 || (p.algo == 0 && (p.taps == 9 ? g_conv_algo : g_gemm_algo) == 20)) {
    const int mb = p.force_bm == 288 ? 9 : 5;
    if (stream_supported(p, dtype, mb)) {
      pl.halo = 20; pl.bm = mb * 32; pl.bn = 64;
      const int nslab = p.Kc / 64;
      if (p.splitk > 0) {
        pl.splitk = p.splitk;
      } else {
        const int nb = stream_mtiles(p, mb);
        pl.splitk = nb >= 240 ? 1 : 240 / nb;
      }
      if (pl.splitk > nslab) pl.splitk = nslab;
      if (pl.splitk < 1) pl.splitk = 1;
      return pl;
    }
  }
  // ---- 3x3 convolution: LDS-resident halo kernels when they apply (algo 2 = 128-byte rows, 3 = 64-byte rows) -----
  const int algo = (p.algo && p.algo != 20) ? p.algo : (g_conv_algo == 20 ? 0 : g_conv_algo);
  if (p.taps == 9 && algo != 1 && p.N >= 128 && !p.a_raw) {   // (the halo kernels read their input in x3 chunks only)
    IgemmParams ph = p;
    ph.algo = ((algo >= 3 && algo <= 9 && algo != 4 && algo != 5) || (algo >= 11 && algo <= 14)) ? algo : 2;
    int bm = 0;
    if (p.force_bm == 256 || p.force_bm == 128) {
      if (conv3_halo_supported(ph, dtype, p.force_bm) && (algo >= 2 || p.force_bn == 0)) bm = p.force_bm;
    } else if (p.force_bm == 0) {
      if (conv3_halo_supported(ph, dtype, 256)) bm = 256;
      else if (conv3_halo_supported(ph, dtype, 128)) bm = 128;
    }
    if (bm) {
      const int nslab = (p.Kc / BK) * (ph.algo == 3 ? 2 : 1);   // split-K granularity: slabs / half slabs
      pl.halo = ph.algo; pl.bm = bm; pl.bn = 128;
      if (p.splitk > 0) {
        pl.splitk = p.splitk;
      } else {
        const int B = p.M / (p.H * p.W);
        const int nb = B * conv3_halo_tiles_per_image(p, bm) * ((p.N + 127) / 128);
        int sk = 1;
        while (nb * sk < 200 && sk < 16 && (p.Kc / BK) / (sk * 2) >= 2) sk *= 2;
        pl.splitk = sk;
      }
      if (pl.splitk > nslab) pl.splitk = nslab;
      if (pl.splitk < 1) pl.splitk = 1;
      return pl;
    }
  }
  // ---- plain GEMM on the 8-wave frame (p.algo == 10: tuner candidate; or the "gemm_algo" option) --------------
  if (p.taps == 1 && (p.algo == 10 || (p.algo == 0 && g_gemm_algo == 10))) {
    int bm = 0;
    if (p.force_bm == 256 || p.force_bm == 128) { if (gemm8_supported(p, dtype, p.force_bm)) bm = p.force_bm; }
    else if (gemm8_supported(p, dtype, 256)) {
      const int hw = p.H > 0 ? p.H * p.W : p.M;
      bm = ((p.M / hw) * ((hw + 255) / 256) * ((p.N + 127) / 128) >= 400 || hw % 256 == 0) ? 256 : 128;
    }
    if (bm) {
      pl.halo = 10; pl.bm = bm; pl.bn = 128;
      pl.splitk = p.splitk > 0 ? p.splitk : 1;
      if (p.out_mode == IG_OUT_QKV) pl.splitk = 1;
      if (pl.splitk > p.Kc / BK) pl.splitk = p.Kc / BK;
      return pl;
    }
  }
  if (p.force_bm && p.force_bn) {
    pl.bm = p.force_bm;
    pl.bn = p.force_bn;
  } else if (p.N <= 64) {
    pl.bm = (p.M >= 128) ? 128 : 64;
    pl.bn = 64;
  } else if (blocks(128, 128) >= 160) {
    pl.bm = 128; pl.bn = 128;
  } else if (p.M > 64) {
    pl.bm = 128; pl.bn = 64;
  } else {
    pl.bm = 64; pl.bn = 64;
  }
  if (p.splitk > 0) {
    pl.splitk = p.splitk;
  } else {
    const int nb = blocks(pl.bm, pl.bn);
    int sk = 1;
    while (nb * sk < 200 && sk < 16 && nkt / (sk * 2) >= 6) sk *= 2;
    pl.splitk = sk;
  }
  if (pl.splitk > nkt) pl.splitk = nkt > 0 ? nkt : 1;
  return pl;
}

int igemm_choose_splitk(const IgemmParams& p, int dtype) {
  IgemmParams q = p;
  q.splitk = 0;
  return igemm_plan(q, dtype).splitk;
}

static bool reduce_rows_ok(const IgemmParams& p) {
  // 3x3 convolutions always finish through the row-tiled reduction; plain GEMMs only when they owe GroupNorm sums
  const int hw = (p.taps == 9 || p.stats != nullptr) ? p.H * p.W : 0;
  return hw > 0 && hw % 16 == 0 && (p.N & 3) == 0 && (p.ldo & 3) == 0 && (p.ldr & 3) == 0 &&
         (p.out_mode == IG_OUT_ROWMAJOR || p.out_mode == IG_OUT_ROWMAJOR_F32);
}

int igemm_stats_rows_per_image(const IgemmParams& p, int dtype) {
  const IgemmPlan pl = igemm_plan(p, dtype);
  if (pl.halo == 20) {   // always finished by the row-tiled reduction
    if (p.H <= 0 || p.out_mode == IG_OUT_QKV) return 0;
    IgemmParams q = p;
    q.stats = reinterpret_cast<float*>(1);
    return reduce_rows_ok(q) ? p.H * p.W / 16 : 0;
  }
  if (p.taps == 1) {
    if (pl.halo != 10 || p.H <= 0 || p.out_mode == IG_OUT_QKV) return 0;
    if (pl.splitk > 1) {
      IgemmParams q = p;
      q.stats = reinterpret_cast<float*>(1);   // "owes GroupNorm sums": the row-tiled reduction
      return reduce_rows_ok(q) ? p.H * p.W / 16 : 0;
    }
    return gemm8_tiles_per_image(p, pl.bm);
  }
  if (pl.splitk > 1) return reduce_rows_ok(p) ? p.H * p.W / 16 : 0;
  if (pl.halo) return conv3_halo_tiles_per_image(p, pl.bm);
  return 0;
}

template <typename T>
static int launch_reduce(const IgemmParams& q, hipStream_t stream) {
  if (q.stats != nullptr || reduce_rows_ok(q)) {
    if (!reduce_rows_ok(q)) return k22_set_error(K22_EINVAL, "igemm: GroupNorm partial sums are not available for this split-K problem");
    hipLaunchKernelGGL((splitk_reduce_rows_kernel<T>), dim3((q.M + 15) / 16, (q.N + 63) / 64), dim3(256), 0, stream, q);
    K22_CHECK_LAUNCH();
    return K22_OK;
  }
  const int64_t total = (int64_t)q.M * q.N;
  int nb = (int)((total / 4 + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(nb), dim3(256), 0, stream, q);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

template <typename T, int BM, int BN, int STAGES, bool ARAW = false>
static int launch_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  constexpr int smem0 = STAGES * (BM + BN) * 128;
  static const int dbg_pad = getenv("K22_DBG_LDS_PAD") ? atoi(getenv("K22_DBG_LDS_PAD")) : 0;   // debug (tools/lds_victim_probe.py)
  const int smem = dbg_pad == 1 ? (smem0 + 1279) / 1280 * 1280 : (dbg_pad == 2 ? smem0 + 4096 : smem0);
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, STAGES, ARAW>), smem, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  q.xcd_remap = g_xcd_remap;
  const int nblocks = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * splitk;
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, STAGES, ARAW>), dim3(nblocks), dim3(256), smem, stream, q);
  K22_CHECK_LAUNCH();
  if (splitk > 1) {
    if constexpr (is_x3<T>::value) return launch_reduce<float>(q, stream);   // an x3 epilogue stores fp32
    else return launch_reduce<T>(q, stream);
  }
  return K22_OK;
}

template <typename T, int BM, int BN>
static int launch_stages(const IgemmParams& p, const IgemmPlan& pl, hipStream_t stream) {
  if constexpr (is_x3<T>::value) {
    if (p.a_raw) {
      if (pl.stages == 2) return launch_cfg<T, BM, BN, 2, true>(p, pl.splitk, stream);
      if (pl.stages == 3) return launch_cfg<T, BM, BN, 3, true>(p, pl.splitk, stream);
      return launch_cfg<T, BM, BN, 4, true>(p, pl.splitk, stream);
    }
  }
  if (pl.stages == 2) return launch_cfg<T, BM, BN, 2>(p, pl.splitk, stream);
  if (pl.stages == 3) return launch_cfg<T, BM, BN, 3>(p, pl.splitk, stream);
  return launch_cfg<T, BM, BN, 4>(p, pl.splitk, stream);
}

template <typename T>
static int launch_typed(const IgemmParams& p, const IgemmPlan& pl, hipStream_t stream) {
  if (pl.bm == 128 && pl.bn == 128) return launch_stages<T, 128, 128>(p, pl, stream);
  if (pl.bm == 128 && pl.bn == 64) return launch_stages<T, 128, 64>(p, pl, stream);
  if (pl.bm == 64 && pl.bn == 128) return launch_stages<T, 64, 128>(p, pl, stream);
  if (pl.bm == 64 && pl.bn == 64) return launch_stages<T, 64, 64>(p, pl, stream);
  return k22_set_error(K22_EINVAL, "igemm: unsupported tile configuration");
}

// split-K finish of an engine of arithmetic type `dtype`: the partials are fp32 and so is what an x3 epilogue stores
static int launch_reduce_dt(const IgemmParams& q, int dtype, hipStream_t stream) {
  return dtype == K22_BF16 ? launch_reduce<bf16_t>(q, stream) : (dtype == K22_F16 ? launch_reduce<f16_t>(q, stream) : launch_reduce<float>(q, stream));
}

int launch_igemm(const IgemmParams& p, int dtype, hipStream_t stream) {
  const int BK = k22_bk(dtype);
  if (p.a_raw && !k22_is_split(dtype)) return k22_set_error(K22_EINVAL, "igemm: a_raw is an option of the split-precision arithmetics only");
  if (p.M <= 0 || p.N <= 0) return K22_OK;
  if (p.taps != 1 && p.taps != 9) return k22_set_error(K22_EINVAL, "igemm: taps must be 1 or 9");
  if (p.Kc % BK != 0 || p.K0 % BK != 0) return k22_set_error(K22_EINVAL, "igemm: K per tap must be a multiple of 64 (bf16) / 32 (fp32)");
  if (p.Npad % 64 != 0 || p.Npad < p.N) return k22_set_error(K22_EINVAL, "igemm: Npad must be roundup(N,64)");
  if (p.K0 < p.Kc && p.A1 == nullptr) return k22_set_error(K22_EINVAL, "igemm: A1 missing for concat operand");
  if (p.out_mode == IG_OUT_QKV) {
    if (p.taps != 1 || p.N % 192 || (p.ldo & 3) || !p.kall || !p.vtall || p.att_T <= 0 || p.M % p.att_T || p.residual)
      return k22_set_error(K22_EINVAL, "igemm: bad qkv-projection problem (N = 3*heads*64, no residual)");
  }
  IgemmPlan pl = igemm_plan(p, dtype);
  if (p.out_mode == IG_OUT_QKV && pl.halo != 20) {
    if (p.splitk > 1) return k22_set_error(K22_EINVAL, "igemm: the qkv projection splits K only on the streaming kernel");
    pl.splitk = 1;
  }
  if (pl.halo == 20) {
    // weight-streaming kernel: fp32 partial tiles, then the common split-K finish (also for splitk == 1)
    if (p.partial == nullptr) return k22_set_error(K22_EINVAL, "igemm: the streaming kernel needs the fp32 partial buffer");
    IgemmParams q = p;
    q.splitk = pl.splitk;
    q.xcd_remap = g_xcd_remap;
    int rc = launch_stream(q, dtype, pl.bm / 32, pl.splitk, stream);
    if (rc) return rc;
    return launch_reduce_dt(q, dtype, stream);
  }
  if (pl.splitk > 1 && p.partial == nullptr) pl.splitk = 1;
  if (p.res_f32 && pl.halo) return k22_set_error(K22_EINVAL, "igemm: fp32 residual is not supported by the halo kernel");
  if (p.gn_coeff != nullptr && !conv3_algo_fuses_gn(pl.halo))
    return k22_set_error(K22_EINVAL, "igemm: the fused GroupNorm-apply input needs the specialised halo kernel (algo 11 / 12)");
  if (p.S0 != nullptr && !pl.halo) return k22_set_error(K22_EINVAL, "igemm: the fused 1x1 skip connection needs the halo kernel");
  if (p.stats != nullptr && pl.splitk == 1 && !pl.halo)
    return k22_set_error(K22_EINVAL, "igemm: GroupNorm partial sums requested from a configuration that cannot produce them");
  if (pl.halo == 10) {
    IgemmParams q = p;
    q.splitk = pl.splitk;
    q.xcd_remap = g_xcd_remap;
    if (q.stages < 2 && g_stages_override >= 0) q.stages = g_stages_override;   // "igemm_stages" option (benches / tests)
    int rc = launch_gemm8(q, dtype, pl.bm, pl.splitk, stream);
    if (rc || pl.splitk == 1) return rc;
    return launch_reduce_dt(q, dtype, stream);
  }
  if (pl.halo) {
    IgemmParams q = p;
    q.splitk = pl.splitk;
    q.xcd_remap = g_xcd_remap;
    q.algo = pl.halo;
    if (q.stages < 2 && g_stages_override >= 0) q.stages = g_stages_override;   // "igemm_stages" option (benches / tests)
    int rc = launch_conv3_halo(q, dtype, pl.bm, pl.splitk, stream);
    if (rc || pl.splitk == 1) return rc;
    return launch_reduce_dt(q, dtype, stream);
  }
  if (dtype == K22_BF16) return launch_typed<bf16_t>(p, pl, stream);
  else if (dtype == K22_F16) return launch_typed<f16_t>(p, pl, stream);
  if (dtype == K22_F32) return launch_typed<float>(p, pl, stream);
  if (dtype == K22_F16X3) return launch_typed<x3_t>(p, pl, stream);
  if (dtype == K22_F16X2) return launch_typed<x2_t>(p, pl, stream);
  return k22_set_error(K22_EINVAL, "igemm: bad dtype");
}
