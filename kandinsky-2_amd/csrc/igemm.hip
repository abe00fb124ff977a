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
//   * global -> registers (16 B/lane, issued before the MFMAs of the current tile) -> LDS
//     double buffer (written after them), ONE barrier per K tile.
//   * LDS image is chunk-XOR-swizzled (common.h) so fragment ds_read_b128s are conflict-free.
//   * 32x32x16 atoms, fp32 accumulate; epilogue fuses bias, residual add and activation, or
//     writes fp32 split-K partials that splitk_reduce_kernel finishes.
#include "kernels.h"
#include <stdlib.h>

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int MI = BM / 64, NI = BN / 64;     // 32x32 atoms per wave along M / N
  constexpr int A_CH = BM / 32, B_CH = BN / 32;  // 16-byte chunks per thread per K tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool conv = (p.taps == 9);
  const T* __restrict__ A0 = reinterpret_cast<const T*>(p.A0);
  const T* __restrict__ A1 = reinterpret_cast<const T*>(p.A1);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);

  // ---- loader geometry: thread -> (row = tid/8 + 32*i, physical chunk = tid%8) --------------
  const int lrow = tid >> 3, lpos = tid & 7;
  const int lchunk = lpos ^ (lrow & 7);  // logical (source) chunk that lands at this position
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
    kt0 = blockIdx.z * per;
    kt1 = kt0 + per < nkt ? kt0 + per : nkt;
  }

  // Staging registers are native vectors and every load below is unconditional (the prefetch of the
  // last iteration re-reads the last tile): conditional staging made hipcc keep them in scratch memory.
  u32x4_t areg[A_CH], breg[B_CH];
  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;
  const int64_t lck = lchunk * EPC;

#define K22_GLOAD(KT)                                                                                   \
  {                                                                                                     \
    const int kt_ = (KT);                                                                               \
    const int tap_ = kt_ / kt_per_tap;                                                                  \
    const int k0_ = (kt_ - tap_ * kt_per_tap) * BK;                                                     \
    const bool second_ = (!conv) && (k0_ >= p.K0);                                                      \
    const T* src_ = second_ ? A1 : A0;                                                                  \
    const int ty_ = tap_ / 3, tx_ = tap_ - ty_ * 3;                                                     \
    const int64_t add_ = (conv ? (int64_t)(ty_ * (p.W + 2) + tx_) * p.Kc + k0_ : (int64_t)(second_ ? k0_ - p.K0 : k0_)) + lck; \
    _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                                    \
        areg[i] = *reinterpret_cast<const u32x4_t*>(src_ + (second_ ? aoff1[i] : aoff0[i]) + add_);     \
    const int64_t badd_ = (int64_t)tap_ * p.Kc + k0_ + lck;                                             \
    _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                    \
        breg[i] = *reinterpret_cast<const u32x4_t*>(Wp + boff[i] + badd_);                              \
  }
#define K22_LSTORE(BUFI)                                                                                \
  {                                                                                                     \
    char* As_ = smem + (BUFI) * BUF;                                                                    \
    char* Bs_ = As_ + A_BYTES;                                                                          \
    _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                                    \
        *reinterpret_cast<u32x4_t*>(As_ + (lrow + 32 * i) * 128 + lpos * 16) = areg[i];                 \
    _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                                                    \
        *reinterpret_cast<u32x4_t*>(Bs_ + (lrow + 32 * i) * 128 + lpos * 16) = breg[i];                 \
  }

  if (kt0 < kt1) {
    K22_GLOAD(kt0);
    K22_LSTORE(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const int nxt = kt + 1 < kt1 ? kt + 1 : kt1 - 1;
      K22_GLOAD(nxt);
      {
        const char* As = smem + cur * BUF;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          Frag<T> a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ld_frag(a[mi], As, wm * (BM / 2) + mi * 32 + l31, ks, h);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) ld_frag(b[ni], Bs, wn * (BN / 2) + ni * 32 + l31, ks, h);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], a[mi], b[ni]);
        }
      }
      K22_LSTORE(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
#undef K22_GLOAD
#undef K22_LSTORE

  // ---- epilogue ------------------------------------------------------------------------
  const int mbase = m0 + wm * (BM / 2), nbase = n0 + wn * (BN / 2);
  if (p.splitk > 1) {
    float* part = p.partial + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = nbase + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + mi * 32 + c_row(r, lane);
          if (m < p.M && n < p.N) part[(int64_t)m * p.N + n] = acc[mi][ni][r];
        }
      }
    return;
  }
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = nbase + ni * 32 + l31;
      const float bv = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + mi * 32 + c_row(r, lane);
        if (m < p.M && n < p.N) {
          float v = acc[mi][ni][r] + bv;
          if (res != nullptr) v += to_f32(res[(int64_t)m * p.ldr + n]);
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
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no staging VGPRs,
// no ds_write pass), STAGES buffers deep with a COUNTED vmcnt so STAGES-2 tiles stay in flight across
// the single raw s_barrier of each K tile.  Same tile geometry / swizzle / epilogue as igemm_kernel.
//   iteration k:  s_waitcnt vmcnt((STAGES-2)*CH)  -> this wave's part of tile k has landed
//                 s_barrier                        -> everybody's part landed; tile k-1's buffer is free
//                 issue tile k+STAGES-1 into that buffer ; MFMAs on tile k
// The LDS image of one wave-instruction is lane-linear (base + lane*16 B): 8 rows x 128 B, so the XOR
// swizzle is applied to the per-lane SOURCE address (logical chunk = pos ^ (row & 7)) and to the reads.
// ---------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int BM, int BN, int STAGES>
__global__ __launch_bounds__(256) void igemm_glds_kernel(const IgemmParams p) {
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int MI = BM / 64, NI = BN / 64;
  constexpr int A_CH = BM / 32, B_CH = BN / 32, CH = A_CH + B_CH;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool conv = (p.taps == 9);
  const T* __restrict__ A0 = reinterpret_cast<const T*>(p.A0);
  const T* __restrict__ A1 = reinterpret_cast<const T*>(p.A1);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);

  const int lrow = tid >> 3, lpos = tid & 7;
  const int lchunk = lpos ^ (lrow & 7);
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
    kt0 = blockIdx.z * per;
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
      __builtin_amdgcn_s_barrier();
      K22_ISSUE(kt + STAGES - 1, fill);
      {
        const char* As = smem + cur * BUF;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          Frag<T> a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ld_frag(a[mi], As, wm * (BM / 2) + mi * 32 + l31, ks, h);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) ld_frag(b[ni], Bs, wn * (BN / 2) + ni * 32 + l31, ks, h);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], a[mi], b[ni]);
        }
      }
      cur = (cur + 1 == STAGES) ? 0 : cur + 1;
      fill = (fill + 1 == STAGES) ? 0 : fill + 1;
    }
    wait_vmcnt<0>();  // drain the redundant tail DMAs before the LDS is released
  }
#undef K22_ISSUE

  const int mbase = m0 + wm * (BM / 2), nbase = n0 + wn * (BN / 2);
  if (p.splitk > 1) {
    float* part = p.partial + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = nbase + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + mi * 32 + c_row(r, lane);
          if (m < p.M && n < p.N) part[(int64_t)m * p.N + n] = acc[mi][ni][r];
        }
      }
    return;
  }
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = nbase + ni * 32 + l31;
      const float bv = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + mi * 32 + c_row(r, lane);
        if (m < p.M && n < p.N) {
          float v = acc[mi][ni][r] + bv;
          if (res != nullptr) v += to_f32(res[(int64_t)m * p.ldr + n]);
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
    }
}

// Finishes a split-K launch: out = act(sum_s partial[s] + bias + residual).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IgemmParams p) {
  const int64_t total = (int64_t)p.M * p.N;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.N), n = (int)(i - (int64_t)m * p.N);
    float v = 0.f;
    for (int s = 0; s < p.splitk; ++s) v += p.partial[(int64_t)s * total + i];
    if (p.bias != nullptr) v += p.bias[n];
    if (res != nullptr) v += to_f32(res[(int64_t)m * p.ldr + n]);
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

// ---- host side ------------------------------------------------------------------------
struct IgemmPlan { int bm, bn, splitk, stages; };  // stages == 0 -> register-staged igemm_kernel

static int g_stages_override = -1;
void igemm_set_default_stages(int v) { g_stages_override = (v == 0 || (v >= 2 && v <= 4)) ? v : -1; }

static int g_default_stages() {
  static int v = -1;
  if (g_stages_override >= 0) return g_stages_override;
  if (v < 0) {
    const char* e = getenv("K22_IGEMM_STAGES");  // 0 = register staging, 2..4 = LDS-DMA pipeline depth
    v = e ? atoi(e) : 2;
    if (v == 1 || v > 4) v = 0;
  }
  return v;
}

static IgemmPlan igemm_plan(const IgemmParams& p, int dtype) {
  const int BK = (dtype == K22_BF16) ? 64 : 32;
  const int nkt = p.taps * (p.Kc / BK);
  IgemmPlan pl;
  auto blocks = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  if (p.force_bm && p.force_bn) {
    pl.bm = p.force_bm;
    pl.bn = p.force_bn;
  } else if (p.N <= 64) {
    pl.bm = (p.M >= 128) ? 128 : 64;
    pl.bn = 64;
  } else if (blocks(128, 128) >= 160) {
    pl.bm = 128; pl.bn = 128;
  } else if (p.M > 64 && blocks(128, 64) >= 96) {
    pl.bm = 128; pl.bn = 64;
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
  pl.stages = p.stages >= 0 ? p.stages : g_default_stages();
  return pl;
}

int igemm_choose_splitk(const IgemmParams& p, int dtype) {
  IgemmParams q = p;
  q.splitk = 0;
  return igemm_plan(q, dtype).splitk;
}

template <typename T, int BM, int BN>
static int launch_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  IgemmParams q = p;
  q.splitk = splitk;
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, splitk);
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN>), grid, dim3(256), smem, stream, q);
  K22_CHECK_LAUNCH();
  if (splitk > 1) {
    const int64_t total = (int64_t)p.M * p.N;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(nb), dim3(256), 0, stream, q);
    K22_CHECK_LAUNCH();
  }
  return K22_OK;
}

template <typename T, int BM, int BN, int STAGES>
static int launch_glds(const IgemmParams& p, int splitk, hipStream_t stream) {
  constexpr int smem = STAGES * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_glds_kernel<T, BM, BN, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  IgemmParams q = p;
  q.splitk = splitk;
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, splitk);
  hipLaunchKernelGGL((igemm_glds_kernel<T, BM, BN, STAGES>), grid, dim3(256), smem, stream, q);
  K22_CHECK_LAUNCH();
  if (splitk > 1) {
    const int64_t total = (int64_t)p.M * p.N;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(nb), dim3(256), 0, stream, q);
    K22_CHECK_LAUNCH();
  }
  return K22_OK;
}

template <typename T, int BM, int BN>
static int launch_glds_stages(const IgemmParams& p, const IgemmPlan& pl, hipStream_t stream) {
  if (pl.stages == 2) return launch_glds<T, BM, BN, 2>(p, pl.splitk, stream);
  if (pl.stages == 3) return launch_glds<T, BM, BN, 3>(p, pl.splitk, stream);
  return launch_glds<T, BM, BN, 4>(p, pl.splitk, stream);
}

template <typename T>
static int launch_typed(const IgemmParams& p, const IgemmPlan& pl, hipStream_t stream) {
  if (pl.stages >= 2) {
    if (pl.bm == 128 && pl.bn == 128) return launch_glds_stages<T, 128, 128>(p, pl, stream);
    if (pl.bm == 128 && pl.bn == 64) return launch_glds_stages<T, 128, 64>(p, pl, stream);
    if (pl.bm == 64 && pl.bn == 128) return launch_glds_stages<T, 64, 128>(p, pl, stream);
    if (pl.bm == 64 && pl.bn == 64) return launch_glds_stages<T, 64, 64>(p, pl, stream);
  }
  if (pl.bm == 128 && pl.bn == 128) return launch_cfg<T, 128, 128>(p, pl.splitk, stream);
  if (pl.bm == 128 && pl.bn == 64) return launch_cfg<T, 128, 64>(p, pl.splitk, stream);
  if (pl.bm == 64 && pl.bn == 128) return launch_cfg<T, 64, 128>(p, pl.splitk, stream);
  if (pl.bm == 64 && pl.bn == 64) return launch_cfg<T, 64, 64>(p, pl.splitk, stream);
  if (pl.bm == 256 && pl.bn == 128) return launch_cfg<T, 256, 128>(p, pl.splitk, stream);
  return k22_set_error(K22_EINVAL, "igemm: unsupported tile configuration");
}

int launch_igemm(const IgemmParams& p, int dtype, hipStream_t stream) {
  const int BK = (dtype == K22_BF16) ? 64 : 32;
  if (p.M <= 0 || p.N <= 0) return K22_OK;
  if (p.taps != 1 && p.taps != 9) return k22_set_error(K22_EINVAL, "igemm: taps must be 1 or 9");
  if (p.Kc % BK != 0 || p.K0 % BK != 0) return k22_set_error(K22_EINVAL, "igemm: K per tap must be a multiple of 64 (bf16) / 32 (fp32)");
  if (p.Npad % 64 != 0 || p.Npad < p.N) return k22_set_error(K22_EINVAL, "igemm: Npad must be roundup(N,64)");
  if (p.K0 < p.Kc && p.A1 == nullptr) return k22_set_error(K22_EINVAL, "igemm: A1 missing for concat operand");
  IgemmPlan pl = igemm_plan(p, dtype);
  if (pl.splitk > 1 && p.partial == nullptr) pl.splitk = 1;
  if (dtype == K22_BF16) return launch_typed<bf16_t>(p, pl, stream);
  if (dtype == K22_F32) return launch_typed<float>(p, pl, stream);
  return k22_set_error(K22_EINVAL, "igemm: bad dtype");
}
