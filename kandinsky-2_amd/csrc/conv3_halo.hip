// k22 — 3x3 convolution with the input tile (+halo) RESIDENT IN LDS across the nine filter taps.
//
// Replaces nn.Conv2d 3x3 of the ResBlocks (kandinsky2/model/unet.py:152,180; ~83 % of the FLOPs of a
// denoise step).  The generic implicit GEMM (igemm.hip) re-fetches the A operand for every tap: 9 x BM rows
// per 64-channel slab, and at 128x128 tiles the L2->LDS path (~56-64 B/clk/CU) is as busy as the MFMA pipes.
// Here the pixels are indexed in the PADDED row-major plane, v = y*(W+2) + x, so that tap (ky,kx) of output v
// is input v + ky*(W+2) + kx: a pure 1-D shift.  A block owns BM consecutive v of one image and keeps the
// BM + 2*(W+2) + 2 input pixels it needs (x 64 channels = one 128-byte row each) in LDS; every tap reads its A
// fragments from the same LDS image at a different row offset.  Outputs with x >= W (two "junk" columns per
// row, 2 % at 96x96) are computed and dropped.  Per slab a block moves (BM + 2W + 6) + 9*BN rows instead of
// 9*(BM + BN): 188 FLOP/B at 256x128 against 64 FLOP/B before, and 3 LDS-DMA instructions per wave per tap
// instead of 8.
//
//   * 8 waves (4 along M x 2 along N), BM in {256, 128}, BN = 128, K slab = 128 bytes of channels
//   * halo buffer double-buffered across slabs (the next slab's halo trickles in, one 8-row piece per wave per
//     tap), weight tiles in an NBST-deep ring (2, 3, 4 or 6: as many as the LDS holds next to the halo) with a COUNTED
//     vmcnt: NBST-1 taps of weights are in flight across the ONE raw s_barrier per tap (at the low-resolution
//     levels the weights stream from HBM, ~2 us away, and a tap is ~0.4 us of MFMA work)
//   * LDS images are lane-linear per LDS-DMA instruction with the chunk-XOR swizzle of common.h on the source
//     address and on the reads (rows of an atom are consecutive -> conflict-free ds_read_b128)
//   * transposed MFMA (lane = pixel, registers = 4 consecutive channels) and an epilogue that goes through LDS:
//     fp32 tile -> 32 B per thread row segments: bias, residual, one rounding, 16-byte stores, and the per-channel
//     (sum, sum of squares) of the STORED values for the next GroupNorm (deterministic, no atomics)
//   * split-K over channel slabs for the low-resolution levels (fp32 partials, finished by splitk_reduce)
//   * variants selected by p.algo (measured in DESIGN.md section 9): 2 = this kernel; 7 = its LDS-DMA issued from inline asm
//     (exact lgkmcnt for the fragment reads; the tuner's usual pick); 6 = 7 + explicit fragment pipeline; 3 = conv3_halo3 below;
//     8 / 9 = measurement only.  gemm8_kernel (plain GEMM on the
//     same frame, qkv / proj_out) lives in this file too because it shares the epilogue (halo_tail)
//   * optional fused 1x1 skip_connection of the ResBlock (out += x . Ws^T): a second, plain-GEMM K loop over the
//     block input's channels (A rows = the tile's own pixels, no halo) that accumulates into the same registers,
//     so the skip tensor is never written, re-read or launched separately
#include "conv3_common.h"

// conv3_spec.hip
int launch_conv3_halo_spec(const IgemmParams& p, int dtype, int bm, int nbst, int splitk, hipStream_t stream);
#ifdef K22_DEBUG_VARIANTS
int launch_conv3_halo_spec_debug(const IgemmParams& p, int nbst, int splitk, hipStream_t stream);
#endif

// LW = number of waves that issue the LDS-DMA: 8 (every wave loads its share right after the barrier: 3 pieces per wave per tap).  (A
// 4-loader form - p.algo 5 - measured 2-5 % slower in round 1 and was deleted in round 5; the parameter stays for the index arithmetic.)
// PIPE: explicit fragment pipeline across the per-tap barrier.  The fragments of k-step ks+1 are read from LDS BEFORE
// the MFMAs of k-step ks are issued (two register sets), and the last k-step of a tap is multiplied after the NEXT
// tap's barrier, where it covers the LDS-DMA issue and the first fragment reads of that tap: a wave never parks on
// lgkmcnt with an empty matrix pipe behind it (the compiler's own schedule reads one MFMA ahead of the use).
template <typename T, int BM, int NBST, bool TRACE = false, int LW = 8, int MODE = 0>
__global__ __launch_bounds__(512) void conv3_halo_kernel(const IgemmParams p) {
  constexpr bool PIPE = MODE == 1;      // explicit fragment pipeline (above)
  constexpr bool GASM = MODE == 1 || MODE == 2;  // LDS-DMA from inline asm (glds16_asm): exact lgkmcnt(N) for the fragment reads
  constexpr bool DBG_NOLOAD = MODE == 3;  // measurement only (wrong results): no LDS-DMA inside the tap loop
  constexpr bool DBG_NOMMA = MODE == 4;   // measurement only (wrong results): fragments are read but not multiplied
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int BN = HALO_BN, NW = LW, WM = 4, WN = 2;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int B_SLOTS = BN / 8 / NW;  // weight LDS-DMA instructions per loader wave per tap
  constexpr int B_BYTES = BN * 128;
  constexpr int APT = HALO_NW / LW;     // halo pieces per loader wave per tap
  constexpr int A_SLOTS = (10 - NBST) * APT;  // taps 0 .. 9-NBST issue APT halo pieces per loader wave
  constexpr int GM = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const bool loader = LW == HALO_NW || wave < LW;   // wave-uniform

  const int W2 = p.W + 2;
  const int VR = p.H * W2;                     // virtual output rows per image
  const int TPI = (VR + BM - 1) / BM;          // m-tiles per image
  const int HRp = (BM + 2 * W2 + 2 + 7) & ~7;  // halo rows (padded to whole 8-row LDS-DMA pieces)
  const int NP = HRp >> 3;
  const int A_BYTES = HRp * 128;
  const int PR_MAX = (p.H + 2) * W2 - 1;       // last pixel of one padded plane
  const int B = p.M / (p.H * p.W);

  // ---- block -> (m-tile, n-tile, k-split) ------------------------------------------------------
  const int gx = B * TPI, gy = (p.N + BN - 1) / BN;
  int L = p.xcd_remap ? xcd_remap_h(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy;
  const int bz = L / per_z;
  L -= bz * per_z;
  const int grp = L / (GM * gy);
  const int first_m = grp * GM;
  const int gsz = gx - first_m < GM ? gx - first_m : GM;
  const int lin = L - grp * GM * gy;
  const int bx = first_m + lin % gsz, by = lin / gsz;
  const int img = bx / TPI, v0 = (bx - img * TPI) * BM;
  const int n0 = by * BN;

  const T* __restrict__ Aimg = reinterpret_cast<const T*>(p.A0) + (int64_t)img * (p.H + 2) * W2 * p.Kc;
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);

  // ---- loader geometry ---------------------------------------------------------------------------
  // halo piece j (8 rows) is issued by wave j % NW in its slot j / NW; lane -> (row 8j + lane/8, position lane%8)
  int aoff[A_SLOTS];
#pragma unroll
  for (int q = 0; q < A_SLOTS; ++q) {
    int j = q * NW + wave;
    if (j > NP - 1) j = NP - 1;  // surplus slots re-load the last piece (same bytes, same place): uniform counting
    const int hr = 8 * j + (lane >> 3);
    int pr = v0 + hr;
    if (pr > PR_MAX) pr = PR_MAX;
    const int chunk = (lane & 7) ^ ((hr >> 1) & 7);
    aoff[q] = pr * p.Kc + chunk * EPC;
  }
  int boff[B_SLOTS];
#pragma unroll
  for (int i = 0; i < B_SLOTS; ++i) {
    const int row = 8 * (wave + NW * i) + (lane >> 3);
    int n = n0 + row;
    if (n > p.Npad - 1) n = p.Npad - 1;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    boff[i] = n * 9 * p.Kc + chunk * EPC;
  }
  const int nslab = p.Kc / BK;
  int s0 = 0, s1 = nslab;
  if (p.splitk > 1) {
    const int per = (nslab + p.splitk - 1) / p.splitk;
    s0 = bz * per;
    s1 = s0 + per < nslab ? s0 + per : nslab;
  }

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  FragA<T> pa[MI];          // PIPE: fragments read but not yet multiplied (zero = a no-op group before the first tap)
  Frag<T> pb[NI];
  if constexpr (PIPE) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) pa[mi] = FragA<T>{};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) pb[ni] = Frag<T>{};
  }
  char* const Bst = smem + 2 * A_BYTES;
  // fragment rows: A row = wm*(BM/4) + mi*32 + l31 + tap shift ; B row = wn*(BN/2) + ni*32 + l31
  const int abase = wm * (BM / WM) + l31;
  int brow[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) brow[ni] = (wn * (BN / WN) + ni * 32 + l31) * 128;
  const int bsw = (l31 >> 1) & 7;  // (row >> 1) & 7 with row = multiple of 32 + l31

  // LDS byte address of the dynamic shared segment (wave-uniform), for the asm LDS-DMA path
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
#define K22_ISSUE_A(Q, SLAB, DST)                                                                          \
  {                                                                                                        \
    int j_ = (Q) * NW + wave;                                                                              \
    if (j_ > NP - 1) j_ = NP - 1;                                                                          \
    if constexpr (GASM)                                                                                    \
      glds16_asm(Aimg + aoff[Q] + (SLAB) * BK, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((DST) - smem) + j_ * 1024)); \
    else                                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Aimg + aoff[Q] + (SLAB) * BK), \
                                     (__attribute__((address_space(3))) void*)((DST) + j_ * 1024), 16, 0, 0); \
  }
#define K22_ISSUE_B(SLAB, TAP, STAGE)                                                                      \
  {                                                                                                        \
    const int kofs_ = (TAP) * p.Kc + (SLAB) * BK;                                                          \
    char* dst_ = Bst + (STAGE) * B_BYTES + wave * 1024;                                                    \
    _Pragma("unroll") for (int i = 0; i < B_SLOTS; ++i) {                                                  \
      if constexpr (GASM)                                                                                  \
        glds16w_asm(Wp + boff[i] + kofs_, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(dst_ - smem) + i * NW * 1024)); \
      else                                                                                                 \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wp + boff[i] + kofs_), \
                                         (__attribute__((address_space(3))) void*)(dst_ + i * NW * 1024), 16, 0, 0); \
    }                                                                                                      \
  }

  if (s0 < s1) {
    // prologue: the whole halo of the first slab, then the weight tiles of taps 0 .. NBST-2
    if (loader) {
#pragma unroll
      for (int q = 0; q < A_SLOTS; ++q) K22_ISSUE_A(q, s0, smem);
#pragma unroll
      for (int t = 0; t < NBST - 1; ++t) K22_ISSUE_B(s0, t, t);
    }
    int trace_it = 0;
    (void)trace_it;
    int cur = 0;               // ring slot of the current tap's weights
    int fill = NBST - 1;       // ring slot the tile NBST-1 taps ahead goes into
    for (int s = s0; s < s1; ++s) {
      char* const Acur = smem + ((s - s0) & 1) * A_BYTES;
      char* const Anext = smem + (((s - s0) & 1) ^ 1) * A_BYTES;
      const int sn = s + 1 < s1 ? s + 1 : s1 - 1;  // past-the-end loads re-read the last slab (uniform counting)
      // the nine taps are expanded with literal tap numbers: the vmcnt immediates and the (slab, tap) of the
      // prefetched weight tile are compile-time functions of the tap
#define K22_TAP(TAP)                                                                                       \
      {                                                                                                    \
        unsigned long long tr0_ = 0, tr1_ = 0, tr2_ = 0;                                                   \
        if constexpr (TRACE) tr0_ = __builtin_amdgcn_s_memtime();                                          \
        if (loader && !DBG_NOLOAD) wait_vmcnt<B_SLOTS * (NBST - 2) + APT * halo_count_a<NBST>(TAP)>();     \
        if constexpr (TRACE) tr1_ = __builtin_amdgcn_s_memtime();                                          \
        raw_barrier();                                                                      \
        if constexpr (TRACE) tr2_ = __builtin_amdgcn_s_memtime();                                          \
        /* the weight tile NBST-1 taps ahead, then the next piece(s) of the next slab's halo */            \
        if (loader && !DBG_NOLOAD) {                                                                       \
          constexpr int ta_ = ((TAP) + NBST - 1) % 9;                                                      \
          const int sa_ = ((TAP) + NBST - 1 >= 9) ? sn : s;                                                \
          K22_ISSUE_B(sa_, ta_, fill);                                                                     \
          if constexpr ((TAP) <= 9 - NBST) {                                                               \
            _Pragma("unroll") for (int a_ = 0; a_ < APT; ++a_) {                                           \
              constexpr int qb_ = ((TAP) <= 9 - NBST ? (TAP) : 0) * APT;                                   \
              K22_ISSUE_A(qb_ + a_, sn, Anext);                                                            \
            }                                                                                              \
          }                                                                                                \
        }                                                                                                  \
        const char* Bcur = Bst + cur * B_BYTES;                                                            \
        const int shift = ((TAP) / 3) * W2 + ((TAP) % 3);                                                  \
        const char* arow[MI];                                                                              \
        int asw[MI];                                                                                       \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                \
          const int ar = abase + mi * 32 + shift;                                                          \
          arow[mi] = Acur + ar * 128;                                                                      \
          asw[mi] = (ar >> 1) & 7;                                                                         \
        }                                                                                                  \
        if constexpr (PIPE) {                                                                              \
          FragA<T> ca[MI];                                                                                 \
          Frag<T> cb[NI];                                                                                  \
          _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ks += 2) {                                       \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag_at(ca[mi], arow[mi], asw[mi], ks, h); \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag_at(cb[ni], Bcur + brow[ni], bsw, ks, h); \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                              \
              _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], pb[ni], pa[mi]);     \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag_at(pa[mi], arow[mi], asw[mi], ks + 1, h); \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag_at(pb[ni], Bcur + brow[ni], bsw, ks + 1, h); \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                              \
              _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], cb[ni], ca[mi]);     \
            __builtin_amdgcn_sched_barrier(0);                                                             \
          }                                                                                                \
          /* the reads of the last k-step are complete before the slot can be refilled (after the next barrier) */ \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        } else {                                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                            \
          FragA<T> a[MI];                                                                                  \
          Frag<T> b[NI];                                                                                   \
          _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag_at(a[mi], arow[mi], asw[mi], ks, h);   \
          _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag_at(b[ni], Bcur + brow[ni], bsw, ks, h); \
          if constexpr (DBG_NOMMA) {                                                                       \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(a[mi].v));             \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(b[ni].v));             \
          } else {                                                                                         \
          _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], b[ni], a[mi]);         \
          }                                                                                                \
        }                                                                                                  \
        /* no fragment read of this tap's weight slot in flight when the slot is refilled after the next barrier */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        }                                                                                                  \
        if constexpr (TRACE) {                                                                             \
          /* stamp after the last MFMA has been ISSUED (issue blocks while the pipe is busy) */            \
          asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[MI - 1][NI - 1][15]));                             \
          const unsigned long long tr3_ = __builtin_amdgcn_s_memtime();                                    \
          if (blockIdx.x == 0 && (wave == 0 || wave == 5) && lane == 0 && p.trace != nullptr) {            \
            unsigned long long* o_ = p.trace + ((size_t)(wave ? 1 : 0) * 4096 + (size_t)trace_it * 4);     \
            if (trace_it < 1024) { o_[0] = tr0_; o_[1] = tr1_; o_[2] = tr2_; o_[3] = tr3_; }                \
          }                                                                                                \
          ++trace_it;                                                                                      \
        }                                                                                                  \
        cur = (cur + 1 == NBST) ? 0 : cur + 1;                                                             \
        fill = (fill + 1 == NBST) ? 0 : fill + 1;                                                          \
      }
      K22_TAP(0) K22_TAP(1) K22_TAP(2) K22_TAP(3) K22_TAP(4) K22_TAP(5) K22_TAP(6) K22_TAP(7) K22_TAP(8)
#undef K22_TAP
    }
  }
#undef K22_ISSUE_A
#undef K22_ISSUE_B
  if constexpr (PIPE) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], pb[ni], pa[mi]);
  }
  halo_tail<T, BM>(p, acc, smem, bx, bz, img, v0, n0);
}

// ================================================================================================================
// gemm8_kernel: plain GEMM  out[m][n] = sum_k A[m][k] W[n][k]  (1x1 convolutions: qkv / proj_out of the AttentionBlocks,
// kandinsky2/model/unet.py:244-268) on the frame of the halo kernel: 8 waves (4 x 2), BM x 128 tile, BM in {256, 128},
// both operands through an NST-deep LDS-DMA ring (one 128-byte-row K slab of A and of W per stage, counted vmcnt, one
// raw barrier per slab) and the same epilogue through LDS (halo_tail): 16-byte stores, bias + residual, GroupNorm
// partial sums of the stored values, or the qkv-projection layout.  Against igemm_kernel (4 waves, <= 128 x 128):
// twice the FLOPs per L2->LDS byte at 256 x 128 and two waves per SIMD; m-tiles never straddle an image (rows of a
// tile past the image are masked), so the per-tile statistics are per-image statistics.
// (Round 2: a variant that staged both operands through registers - plain global_load two slabs ahead, ds_write_b128 into a
// two-stage LDS ring, no LDS-DMA - was built, parity-green, and measured equal: 1.29 ms against 1.26-1.31 ms for the GEMMs of one
// step.  The K loop of these GEMMs is not bound by the LDS-DMA issue cost; removed.)
// ================================================================================================================
template <typename T, int BM, int NST, bool ARAW = false>
__global__ __launch_bounds__(512) void gemm8_kernel(const IgemmParams p) {
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int BN = HALO_BN, NW = HALO_NW, WM = 4, WN = 2;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int A_SLOTS = BM / 8 / NW, B_SLOTS = BN / 8 / NW, CH = A_SLOTS + B_SLOTS;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
  constexpr int GM = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;

  const int HW = p.H > 0 ? p.H * p.W : p.M;   // rows per image
  const int TPI = (HW + BM - 1) / BM;
  const int B = p.M / HW;
  const int gx = B * TPI, gy = (p.N + BN - 1) / BN;
  int L = p.xcd_remap ? xcd_remap_h(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy;
  const int bz = L / per_z;
  L -= bz * per_z;
  const int grp = L / (GM * gy);
  const int first_m = grp * GM;
  const int gsz = gx - first_m < GM ? gx - first_m : GM;
  const int lin = L - grp * GM * gy;
  const int bx = first_m + lin % gsz, by = lin / gsz;
  const int img = bx / TPI, v0 = (bx - img * TPI) * BM;
  const int n0 = by * BN;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A0) + (int64_t)img * HW * p.lda0;
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);
  int aoff[A_SLOTS], boff[B_SLOTS];
#pragma unroll
  for (int i = 0; i < A_SLOTS; ++i) {
    const int row = 8 * (wave + NW * i) + (lane >> 3);
    int v = v0 + row;
    if (v > HW - 1) v = HW - 1;                 // rows past the image re-read its last pixel; they are never stored
    aoff[i] = v * (int)p.lda0 + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
  }
#pragma unroll
  for (int i = 0; i < B_SLOTS; ++i) {
    const int row = 8 * (wave + NW * i) + (lane >> 3);
    int n = n0 + row;
    if (n > p.Npad - 1) n = p.Npad - 1;
    boff[i] = n * p.Kc + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
  }
  const int nslab = p.Kc / BK;
  int s0 = 0, s1 = nslab;
  if (p.splitk > 1) {
    const int per = (nslab + p.splitk - 1) / p.splitk;
    s0 = bz * per;
    s1 = s0 + per < nslab ? s0 + per : nslab;
  }

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
#define K22_ISSUE_G(SLAB, STAGE)                                                                           \
  {                                                                                                        \
    int sl_ = (SLAB);                                                                                      \
    if (sl_ > s1 - 1) sl_ = s1 - 1;   /* past-the-end stages re-read the last slab: uniform counting */    \
    const unsigned d_ = lds0 + (STAGE) * BUF + wave * 1024;                                                \
    _Pragma("unroll") for (int i = 0; i < A_SLOTS; ++i)                                                    \
        glds16_asm(A + aoff[i] + sl_ * BK, __builtin_amdgcn_readfirstlane(d_ + i * NW * 1024));            \
    _Pragma("unroll") for (int i = 0; i < B_SLOTS; ++i)                                                    \
        glds16w_asm(Wp + boff[i] + sl_ * BK, __builtin_amdgcn_readfirstlane(d_ + A_BYTES + i * NW * 1024)); \
  }
  int arow[MI], brow[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) arow[mi] = (wm * (BM / WM) + mi * 32 + l31) * 128;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) brow[ni] = A_BYTES + (wn * (BN / WN) + ni * 32 + l31) * 128;
  const int sw = (l31 >> 1) & 7;

  if (s0 < s1) {
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) K22_ISSUE_G(s0 + t, t);
    int cur = 0, fill = NST - 1;
    for (int s = s0; s < s1; ++s) {
      wait_vmcnt<(NST - 2) * CH>();
      raw_barrier();
      K22_ISSUE_G(s + NST - 1, fill);
      const char* St = smem + cur * BUF;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        FragA<T> a[MI];
        Frag<T> b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ld_frag_at_a<ARAW, T>(a[mi], St + arow[mi], sw, ks, h);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ld_frag_at(b[ni], St + brow[ni], sw, ks, h);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], b[ni], a[mi]);
      }
      cur = (cur + 1 == NST) ? 0 : cur + 1;
      fill = (fill + 1 == NST) ? 0 : fill + 1;
    }
  }
#undef K22_ISSUE_G
  halo_tail<T, BM, true>(p, acc, smem, bx, bz, img, v0, n0);
}

// ================================================================================================================
// gemm8_spec_kernel (round 6, p.stages == 3 / 4; 16-bit types): gemm8_kernel's tile, ring and epilogue with the eight waves SPECIALISED the way
// conv3_halo_spec_kernel's are - what the tuning report of the C3 shape asked for: at M = 32 768 the lock-step GEMM ran at 0.18-0.29 of the
// bf16 peak (qkv 64x64: 161 us for 116 GFLOP) beside 3x3 convolutions of the same tile at 0.57.
//   waves 0-3 (consumers, one per SIMD): 2 x 2 over the BM x 128 tile, (BM/2) x 64 per wave; per slab they read fragments and issue MFMAs
//              through the explicit two-set pipeline (reads of k-step ks + 1 interleaved one behind every MFMA of k-step ks; the last k-step
//              of a slab is multiplied behind the next slab's barrier) - no LDS-DMA, no vmcnt wait;
//   waves 4-7 (producers): all the LDS-DMA of a slab (BM/32 + 4 pieces each) right behind its barrier, then the counted vmcnt wait.
// One raw barrier per slab for all eight waves; slot reuse as in gemm8_kernel (iteration s refills the stage iteration s - 1 read; the
// consumers drain lgkmcnt before the next barrier).  Every accumulator sees the same MFMAs in the same k order as in gemm8_kernel: same bits.
// ================================================================================================================
template <typename T, int BM, int NST>
__global__ __launch_bounds__(512) void gemm8_spec_kernel(const IgemmParams p) {
  using TR = TT<T>;
  static_assert(sizeof(T) == 2, "gemm8_spec_kernel: 16-bit operands");
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  static_assert(KSTEPS % 2 == 0, "two-set fragment pipeline");
  constexpr int BN = HALO_BN, NWL = 4, WM = 2, WN = 2;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int A_SLOTS = BM / 8 / NWL, B_SLOTS = BN / 8 / NWL, CH = A_SLOTS + B_SLOTS;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
  constexpr int GM = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;   // wave-uniform
  const int h = lane >> 5, l31 = lane & 31;

  const int HW = p.H > 0 ? p.H * p.W : p.M;   // rows per image
  const int TPI = (HW + BM - 1) / BM;
  const int B = p.M / HW;
  const int gx = B * TPI, gy = (p.N + BN - 1) / BN;
  int L = p.xcd_remap ? xcd_remap_h(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy;
  const int bz = L / per_z;
  L -= bz * per_z;
  const int grp = L / (GM * gy);
  const int first_m = grp * GM;
  const int gsz = gx - first_m < GM ? gx - first_m : GM;
  const int lin = L - grp * GM * gy;
  const int bx = first_m + lin % gsz, by = lin / gsz;
  const int img = bx / TPI, v0 = (bx - img * TPI) * BM;
  const int n0 = by * BN;

  const int nslab = p.Kc / BK;
  int s0 = 0, s1 = nslab;
  if (p.splitk > 1) {
    const int per = (nslab + p.splitk - 1) / p.splitk;
    s0 = bz * per;
    s1 = s0 + per < nslab ? s0 + per : nslab;
  }

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  if (s0 < s1) {
    if (producer) {
      const int lw = wave - 4;
      const T* __restrict__ A = reinterpret_cast<const T*>(p.A0) + (int64_t)img * HW * p.lda0;
      const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);
      int aoff[A_SLOTS], boff[B_SLOTS];
#pragma unroll
      for (int i = 0; i < A_SLOTS; ++i) {
        const int row = 8 * (lw + NWL * i) + (lane >> 3);
        int v = v0 + row;
        if (v > HW - 1) v = HW - 1;                 // rows past the image re-read its last pixel; they are never stored
        aoff[i] = v * (int)p.lda0 + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
      }
#pragma unroll
      for (int i = 0; i < B_SLOTS; ++i) {
        const int row = 8 * (lw + NWL * i) + (lane >> 3);
        int n = n0 + row;
        if (n > p.Npad - 1) n = p.Npad - 1;
        boff[i] = n * p.Kc + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
      }
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
#define K22_ISSUE_GS(SLAB, STAGE)                                                                          \
      {                                                                                                    \
        int sl_ = (SLAB);                                                                                  \
        if (sl_ > s1 - 1) sl_ = s1 - 1;   /* past-the-end stages re-read the last slab: uniform counting */ \
        const unsigned d_ = lds0 + (STAGE) * BUF + lw * 1024;                                              \
        _Pragma("unroll") for (int i = 0; i < A_SLOTS; ++i)                                                \
            glds16_asm(A + aoff[i] + sl_ * BK, __builtin_amdgcn_readfirstlane(d_ + i * NWL * 1024));       \
        _Pragma("unroll") for (int i = 0; i < B_SLOTS; ++i)                                                \
            glds16w_asm(Wp + boff[i] + sl_ * BK, __builtin_amdgcn_readfirstlane(d_ + A_BYTES + i * NWL * 1024)); \
      }
#pragma unroll
      for (int t = 0; t < NST - 1; ++t) K22_ISSUE_GS(s0 + t, t);
      int fill = NST - 1;
      for (int s = s0; s < s1; ++s) {
        wait_vmcnt<(NST - 2) * CH>();
        raw_barrier();
        K22_ISSUE_GS(s + NST - 1, fill);
        fill = (fill + 1 == NST) ? 0 : fill + 1;
      }
#undef K22_ISSUE_GS
    } else {
      const int wm = wave >> 1, wn = wave & 1;
      int arow[MI], brow[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) arow[mi] = (wm * (BM / WM) + mi * 32 + l31) * 128;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) brow[ni] = A_BYTES + (wn * (BN / WN) + ni * 32 + l31) * 128;
      const int sw = (l31 >> 1) & 7;
      Frag<T> pa[MI], pb[NI];          // fragments read but not yet multiplied (zero = a no-op group before the first slab)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) pa[mi] = Frag<T>{};
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) pb[ni] = Frag<T>{};
      constexpr int NRD = MI + NI, NMF = MI * NI, MPR = NMF / NRD;
#define K22_GS_INTERLEAVE()                                                                                \
      {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NRD; ++i_) {                                               \
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
        }                                                                                                  \
        if constexpr (NMF - MPR * NRD > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - MPR * NRD, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
      }
      int cur = 0;
      for (int s = s0; s < s1; ++s) {
        raw_barrier();
        const char* St = smem + cur * BUF;
        Frag<T> ca[MI], cb[NI];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks += 2) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ld_frag_at(ca[mi], St + arow[mi], sw, ks, h);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) ld_frag_at(cb[ni], St + brow[ni], sw, ks, h);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], pb[ni], pa[mi]);
          K22_GS_INTERLEAVE();
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ld_frag_at(pa[mi], St + arow[mi], sw, ks + 1, h);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) ld_frag_at(pb[ni], St + brow[ni], sw, ks + 1, h);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], cb[ni], ca[mi]);
          K22_GS_INTERLEAVE();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // no fragment read of this stage in flight when the producers refill it after the next barrier
        cur = (cur + 1 == NST) ? 0 : cur + 1;
      }
#undef K22_GS_INTERLEAVE
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], pb[ni], pa[mi]);
    }
  }
  halo_tail<T, BM, true, true>(p, acc, smem, bx, bz, img, v0, n0);
}

// ================================================================================================================
// conv3_halo3_kernel: the same LDS-resident halo scheme on 64-BYTE rows.  The K loop walks HALF slabs (32 bf16 / 16 fp32
// channels); one iteration = the three taps of one filter row (ky) of one half slab = 24 MFMAs per wave between
// barriers (16 before), its weight tiles (3 x 128 rows x 64 B = 24 KB) sit in an RB-deep ring (RB-1 iterations in
// flight: 1-3 us of prefetch even where the weights stream from HBM) and the halo buffers shrink to half
// (2 x 29 KB at 96x96 instead of 2 x 57 KB), which is what makes room for the deeper ring at the wide levels.
//   per wave per iteration: <= 3 halo LDS-DMA pieces (16 rows x 64 B each, first two iterations of a half slab) + 3 weight
//   pieces, 24 ds_read_b128 (swizzle keyed on row >> 2: conflict-free for 64-byte rows), 24 MFMAs.
// ================================================================================================================
namespace {
__device__ __forceinline__ void ld_frag64(Frag<bf16_t>& f, const char* rowp, int sw, int ks, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(rowp + (((2 * ks + h) ^ sw) << 4));
}
__device__ __forceinline__ void ld_frag64(Frag<f16_t>& f, const char* rowp, int sw, int ks, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(rowp + (((2 * ks + h) ^ sw) << 4));
}
__device__ __forceinline__ void ld_frag64(Frag<float>& f, const char* rowp, int sw, int /*ks == 0*/, int h) {
  const float4 a = *reinterpret_cast<const float4*>(rowp + (((2 * h) ^ sw) << 4));
  const float4 b = *reinterpret_cast<const float4*>(rowp + (((2 * h + 1) ^ sw) << 4));
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// loads a wave may leave in flight at the top of iteration KY (see the derivation next to K22_IT3 below)
template <int RB> constexpr int halo3_wait(int ky) { return RB == 2 ? 0 : (RB == 3 ? (ky == 0 ? 3 : 6) : (ky == 0 ? 6 : (ky == 1 ? 9 : 12))); }
constexpr int HALO3_ASLOTS = 6;   // halo LDS-DMA slots per wave per half slab (3 in each of its first two iterations)
}  // namespace

template <typename T, int BM, int RB>
__global__ __launch_bounds__(512) void conv3_halo3_kernel(const IgemmParams p) {
  constexpr int EPH = 64 / (int)sizeof(T);   // channels per 64-byte half-slab row
  constexpr int EPC = TT<T>::EPC;            // channels per 16-byte chunk
  constexpr int KS_U = EPH / 16;             // MFMA k-steps per tap of a half slab (2 bf16, 1 fp32)
  constexpr int BN = HALO_BN, NW = HALO_NW, WM = 4, WN = 2;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int BT_BYTES = 3 * BN * 64;      // weight tiles of one iteration (three taps)
  constexpr int GM = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;

  const int W2 = p.W + 2;
  const int VR = p.H * W2;
  const int TPI = (VR + BM - 1) / BM;
  const int HR16 = (BM + 2 * W2 + 2 + 15) & ~15;   // halo rows, whole 16-row LDS-DMA pieces
  const int NPA = HR16 >> 4;
  const int A_BYTES = HR16 * 64;
  const int PR_MAX = (p.H + 2) * W2 - 1;
  const int B = p.M / (p.H * p.W);

  const int gx = B * TPI, gy = (p.N + BN - 1) / BN;
  int L = p.xcd_remap ? xcd_remap_h(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy;
  const int bz = L / per_z;
  L -= bz * per_z;
  const int grp = L / (GM * gy);
  const int first_m = grp * GM;
  const int gsz = gx - first_m < GM ? gx - first_m : GM;
  const int lin = L - grp * GM * gy;
  const int bx = first_m + lin % gsz, by = lin / gsz;
  const int img = bx / TPI, v0 = (bx - img * TPI) * BM;
  const int n0 = by * BN;

  const T* __restrict__ Aimg = reinterpret_cast<const T*>(p.A0) + (int64_t)img * (p.H + 2) * W2 * p.Kc;
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);

  // loader geometry: a piece = 16 rows x 64 B; lane -> (row 16*piece + lane/4, position lane%4)
  int aoff[HALO3_ASLOTS];
#pragma unroll
  for (int q = 0; q < HALO3_ASLOTS; ++q) {
    int j = q * NW + wave;
    if (j > NPA - 1) j = NPA - 1;   // surplus slots re-load the last piece (same bytes, same place): uniform counting
    const int hr = 16 * j + (lane >> 2);
    int pr = v0 + hr;
    if (pr > PR_MAX) pr = PR_MAX;
    aoff[q] = pr * p.Kc + ((lane & 3) ^ ((hr >> 2) & 3)) * EPC;
  }
  int boff;
  {
    const int row = 16 * wave + (lane >> 2);
    int n = n0 + row;
    if (n > p.Npad - 1) n = p.Npad - 1;
    boff = n * 9 * p.Kc + ((lane & 3) ^ ((row >> 2) & 3)) * EPC;
  }
  const int nhs = p.Kc / EPH;
  int h0 = 0, h1 = nhs;
  if (p.splitk > 1) {
    const int per = (nhs + p.splitk - 1) / p.splitk;
    h0 = bz * per;
    h1 = h0 + per < nhs ? h0 + per : nhs;
  }

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  char* const Bring = smem + 2 * A_BYTES;
  const int abase = wm * (BM / WM) + l31;
  int brow[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) brow[ni] = (wn * (BN / WN) + ni * 32 + l31) * 64;
  const int bsw = (l31 >> 2) & 3;

#define K22_ISSUE_A3(Q, HS, DST)                                                                           \
  {                                                                                                        \
    int j_ = (Q) * NW + wave;                                                                              \
    if (j_ > NPA - 1) j_ = NPA - 1;                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Aimg + aoff[Q] + (HS) * EPH), \
                                     (__attribute__((address_space(3))) void*)((DST) + j_ * 1024), 16, 0, 0); \
  }
  // weight tiles of iteration (HS, KY): taps KY*3 + {0,1,2}; this wave's piece = rows [16*wave, +16) of each
#define K22_ISSUE_B3(HS, KY, SLOT)                                                                         \
  {                                                                                                        \
    const int kofs_ = (KY) * 3 * p.Kc + (HS) * EPH;                                                        \
    char* dst_ = Bring + (SLOT) * BT_BYTES + wave * 1024;                                                  \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                       \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wp + boff + kofs_ + kx * p.Kc), \
                                         (__attribute__((address_space(3))) void*)(dst_ + kx * BN * 64), 16, 0, 0); \
  }

  if (h0 < h1) {
    // prologue: the whole halo of the first half slab, then the weight tiles of iterations 0 .. RB-2
#pragma unroll
    for (int q = 0; q < HALO3_ASLOTS; ++q) K22_ISSUE_A3(q, h0, smem);
#pragma unroll
    for (int j = 0; j < RB - 1; ++j) {
      const int hs_ = h0 + j / 3 < h1 ? h0 + j / 3 : h1 - 1;
      K22_ISSUE_B3(hs_, j % 3, j);
    }
    int cur = 0, fill = RB - 1;
    for (int hs = h0; hs < h1; ++hs) {
      char* const Acur = smem + ((hs - h0) & 1) * A_BYTES;
      char* const Anext = smem + (((hs - h0) & 1) ^ 1) * A_BYTES;
      const int hn = hs + 1 < h1 ? hs + 1 : h1 - 1;   // past-the-end loads re-read the last half slab (uniform counting)
      // VMEM queue of a wave, per iteration: [3 halo pieces (KY < 2)] then [3 weight pieces of iteration it+RB-1].
      // At the top of iteration KY the weight tiles issued RB-1 iterations ago must have landed, and at KY == 0 also the
      // halo issued in the previous half slab's iterations 0 and 1; everything younger may stay in flight:
      //   RB = 2: 0 / 0 / 0     RB = 3: 3 / 6 / 6     RB = 4: 6 (halo!) / 9 / 12        (KY = 0 / 1 / 2)
#define K22_IT3(KY)                                                                                        \
      {                                                                                                    \
        wait_vmcnt<halo3_wait<RB>(KY)>();                                                                  \
        raw_barrier();                                                                      \
        if constexpr ((KY) < 2) {                                                                          \
          K22_ISSUE_A3(3 * ((KY) < 2 ? (KY) : 0) + 0, hn, Anext);                                          \
          K22_ISSUE_A3(3 * ((KY) < 2 ? (KY) : 0) + 1, hn, Anext);                                          \
          K22_ISSUE_A3(3 * ((KY) < 2 ? (KY) : 0) + 2, hn, Anext);                                          \
        }                                                                                                  \
        {                                                                                                  \
          constexpr int kya_ = ((KY) + RB - 1) % 3;                                                        \
          int hsa_ = hs + ((KY) + RB - 1) / 3;                                                             \
          if (hsa_ > h1 - 1) hsa_ = h1 - 1;                                                                \
          K22_ISSUE_B3(hsa_, kya_, fill);                                                                  \
        }                                                                                                  \
        const char* Bcur = Bring + cur * BT_BYTES;                                                         \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                 \
          const int shift = (KY) * W2 + kx;                                                                \
          const char* arow[MI];                                                                            \
          int asw[MI];                                                                                     \
          _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                              \
            const int ar = abase + mi * 32 + shift;                                                        \
            arow[mi] = Acur + ar * 64;                                                                     \
            asw[mi] = (ar >> 2) & 3;                                                                       \
          }                                                                                                \
          _Pragma("unroll") for (int ks = 0; ks < KS_U; ++ks) {                                            \
            Frag<T> a[MI], b[NI];                                                                          \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag64(a[mi], arow[mi], asw[mi], ks, h);  \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag64(b[ni], Bcur + kx * BN * 64 + brow[ni], bsw, ks, h); \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                              \
              _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], b[ni], a[mi]);       \
          }                                                                                                \
        }                                                                                                  \
        cur = (cur + 1 == RB) ? 0 : cur + 1;                                                               \
        fill = (fill + 1 == RB) ? 0 : fill + 1;                                                            \
      }
      K22_IT3(0) K22_IT3(1) K22_IT3(2)
#undef K22_IT3
    }
  }
#undef K22_ISSUE_A3
#undef K22_ISSUE_B3
  halo_tail<T, BM>(p, acc, smem, bx, bz, img, v0, n0);
}

// ---- host side ---------------------------------------------------------------------------------
static int halo3_rows(const IgemmParams& p, int bm) { return (bm + 2 * (p.W + 2) + 2 + 15) & ~15; }
static size_t halo3_smem_bytes(const IgemmParams& p, int bm, int rb) {
  const size_t main_loop = (size_t)2 * halo3_rows(p, bm) * 64 + (size_t)rb * 3 * HALO_BN * 64;
  const size_t epi = (size_t)bm * (HALO_BN * 4 + 16);
  const size_t skip = p.S0 ? (size_t)(bm == 256 ? 3 : 4) * (bm * 128 + HALO_BN * 128) : 0;   // halo_tail: NSK stages
  size_t m = main_loop > epi ? main_loop : epi;
  return m > skip ? m : skip;
}
static int halo3_pick_rb(const IgemmParams& p, int bm) {
  if (halo3_rows(p, bm) / 16 > HALO3_ASLOTS * HALO_NW) return 0;
  for (int rb : {4, 3, 2})
    if (halo3_smem_bytes(p, bm, rb) <= 160 * 1024) return rb;
  return 0;
}

// ---- gemm8_kernel ---------------------------------------------------------------------------------------------------
// ring depth: 3 stages of 48 KB at BM = 256; at BM = 128 (32 KB per stage) either 4 stages (one workgroup per CU) or,
// on request (p.stages == 2), 2 stages = 68 KB with the epilogue tile, so that TWO workgroups share a CU and one's
// prologue / epilogue overlaps the other's K loop (short-K GEMMs: K = 768 is 12 slabs).
static int gemm8_nst(int bm, int stages) { return bm == 256 ? 3 : (stages == 2 ? 2 : 4); }
static size_t gemm8_smem_bytes(int bm, int nst) {
  const size_t main_loop = (size_t)nst * (bm + HALO_BN) * 128;
  const size_t epi = (size_t)bm * (HALO_BN * 4 + 16);
  return main_loop > epi ? main_loop : epi;
}
int gemm8_tiles_per_image(const IgemmParams& p, int bm) { return ((p.H > 0 ? p.H * p.W : p.M) + bm - 1) / bm; }

bool gemm8_supported(const IgemmParams& p, int dtype, int bm) {
  const int BK = k22_bk(dtype);
  if (p.taps != 1 || (bm != 256 && bm != 128) || p.N < 128) return false;
  if (p.K0 != p.Kc || p.S0 != nullptr || p.res_f32) return false;          // one A operand, no fused skip, T residual
  if (p.out_mode != IG_OUT_ROWMAJOR && p.out_mode != IG_OUT_ROWMAJOR_F32 && p.out_mode != IG_OUT_QKV) return false;
  if (p.N % 8 || p.ldo % 8 || (p.residual && p.ldr % 8) || p.Kc % BK || p.lda0 % 8) return false;
  const int hw = p.H > 0 ? p.H * p.W : p.M;
  if (hw <= 0 || p.M % hw) return false;
  if (p.out_mode == IG_OUT_QKV && (p.att_T != hw || p.N % 384)) return false;  // an n-tile stays inside q, k or v
  if ((int64_t)p.M * p.lda0 >= (1ll << 31) || (int64_t)p.Npad * p.Kc >= (1ll << 31)) return false;
  return true;
}

template <typename T, int BM, int NST, bool ARAW = false>
static int launch_gemm8_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  if constexpr (is_x3<T>::value && !ARAW) { if (p.a_raw) return launch_gemm8_cfg<T, BM, NST, true>(p, splitk, stream); }
  const size_t smem = gemm8_smem_bytes(BM, NST);
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&gemm8_kernel<T, BM, NST, ARAW>), 160 * 1024, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  const int hw = p.H > 0 ? p.H * p.W : p.M;
  const int nblocks = (p.M / hw) * gemm8_tiles_per_image(p, BM) * ((p.N + HALO_BN - 1) / HALO_BN) * splitk;
  hipLaunchKernelGGL((gemm8_kernel<T, BM, NST, ARAW>), dim3(nblocks), dim3(512), smem, stream, q);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

template <typename T, int BM, int NST>
static int launch_gemm8_spec_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  const size_t smem = gemm8_smem_bytes(BM, NST);
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&gemm8_spec_kernel<T, BM, NST>), 160 * 1024, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  const int hw = p.H > 0 ? p.H * p.W : p.M;
  const int nblocks = (p.M / hw) * gemm8_tiles_per_image(p, BM) * ((p.N + HALO_BN - 1) / HALO_BN) * splitk;
  hipLaunchKernelGGL((gemm8_spec_kernel<T, BM, NST>), dim3(nblocks), dim3(512), smem, stream, q);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

// p.stages == 3 selects the specialised, pipelined form (16-bit types; the split types keep the lock-step kernel)
bool gemm8_spec_supported(int dtype) { return dtype == K22_BF16 || dtype == K22_F16; }

// Launches gemm8_kernel only (a split-K reduction, if any, is the caller's: launch_igemm).
int launch_gemm8(const IgemmParams& p, int dtype, int bm, int splitk, hipStream_t stream) {
  if (!gemm8_supported(p, dtype, bm)) return k22_set_error(K22_EINVAL, "gemm8: unsupported problem");
  if ((p.stages == 3 || p.stages == 4) && gemm8_spec_supported(dtype)) {
    // 4 (BM = 128 only): 2-deep ring = 68 KB with the epilogue tile, so that TWO workgroups share a CU and one's prologue / epilogue (a burst of
    // output traffic that every CU of a lock-step round issues at the same time) overlaps the other's K loop - the p.stages == 2 idea of gemm8_kernel
    if (p.stages == 4 && bm == 128) return dtype == K22_BF16 ? launch_gemm8_spec_cfg<bf16_t, 128, 2>(p, splitk, stream) : launch_gemm8_spec_cfg<f16_t, 128, 2>(p, splitk, stream);
    if (dtype == K22_BF16) return bm == 256 ? launch_gemm8_spec_cfg<bf16_t, 256, 3>(p, splitk, stream) : launch_gemm8_spec_cfg<bf16_t, 128, 4>(p, splitk, stream);
    return bm == 256 ? launch_gemm8_spec_cfg<f16_t, 256, 3>(p, splitk, stream) : launch_gemm8_spec_cfg<f16_t, 128, 4>(p, splitk, stream);
  }
  const int nst = gemm8_nst(bm, p.stages);
  if (dtype == K22_BF16) {
    if (bm == 256) return launch_gemm8_cfg<bf16_t, 256, 3>(p, splitk, stream);
    return nst == 2 ? launch_gemm8_cfg<bf16_t, 128, 2>(p, splitk, stream) : launch_gemm8_cfg<bf16_t, 128, 4>(p, splitk, stream);
  }
  if (dtype == K22_F16) {
    if (bm == 256) return launch_gemm8_cfg<f16_t, 256, 3>(p, splitk, stream);
    return nst == 2 ? launch_gemm8_cfg<f16_t, 128, 2>(p, splitk, stream) : launch_gemm8_cfg<f16_t, 128, 4>(p, splitk, stream);
  }
  if (dtype == K22_F16X3) {
    if (bm == 256) return launch_gemm8_cfg<x3_t, 256, 3>(p, splitk, stream);
    return nst == 2 ? launch_gemm8_cfg<x3_t, 128, 2>(p, splitk, stream) : launch_gemm8_cfg<x3_t, 128, 4>(p, splitk, stream);
  }
  if (dtype == K22_F16X2) {
    if (bm == 256) return launch_gemm8_cfg<x2_t, 256, 3>(p, splitk, stream);
    return nst == 2 ? launch_gemm8_cfg<x2_t, 128, 2>(p, splitk, stream) : launch_gemm8_cfg<x2_t, 128, 4>(p, splitk, stream);
  }
  if (bm == 256) return launch_gemm8_cfg<float, 256, 3>(p, splitk, stream);
  return nst == 2 ? launch_gemm8_cfg<float, 128, 2>(p, splitk, stream) : launch_gemm8_cfg<float, 128, 4>(p, splitk, stream);
}

bool conv3_halo_supported(const IgemmParams& p, int dtype, int bm) {
  const int BK = k22_bk(dtype);
  if (p.taps != 9 || (bm != 256 && bm != 128)) return false;
  // split precision: the input is read in x3 chunks; instantiated forms = the lock-step kernel with asm LDS-DMA (algo 2 / 5 / 6 / 7 all
  // run it) and the specialised kernel (11 / 12)
  if (k22_is_split(dtype) && (p.a_raw || p.algo == 3)) return false;
  if (p.gn_coeff != nullptr) {   // fused GroupNorm-apply: the specialised kernels only; slabs never straddle the two raw sources
    if (!conv3_algo_fuses_gn(p.algo) || !p.gn_x0 || p.gn_C0 <= 0 || p.gn_C0 > p.Kc || p.gn_C0 % BK || (p.gn_C0 < p.Kc && !p.gn_x1)) return false;
    if ((int64_t)p.H * p.W * p.Kc >= (1ll << 31)) return false;
  }
  if (p.out_mode != IG_OUT_ROWMAJOR && p.out_mode != IG_OUT_ROWMAJOR_F32) return false;
  if (p.N % 8 || p.ldo % 8 || (p.residual && p.ldr % 8) || p.Kc % BK) return false;
  if (p.H <= 0 || p.W <= 0 || p.M % (p.H * p.W)) return false;
  if (p.S0 != nullptr) {
    if (!p.Ws || p.SK0 % BK || p.SK1 % BK || (p.SK1 > 0 && !p.S1) || p.SK0 <= 0) return false;
    if ((int64_t)p.M * (p.SK0 > p.SK1 ? p.SK0 : p.SK1) >= (1ll << 31) || (int64_t)p.Npad * (p.SK0 + p.SK1) >= (1ll << 31)) return false;
  }
  if ((p.algo == 3 ? halo3_pick_rb(p, bm) : halo_pick_nbst(p, bm)) == 0) return false;
  if ((int64_t)(p.H + 2) * (p.W + 2) * p.Kc >= (1ll << 31) || (int64_t)p.Npad * 9 * p.Kc >= (1ll << 31)) return false;
  return true;
}

int conv3_halo_tiles_per_image(const IgemmParams& p, int bm) { return (p.H * (p.W + 2) + bm - 1) / bm; }

template <typename T, int BM, int NBST, int LW, int MODE>
static int launch_halo_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  const size_t smem = halo_smem_bytes(p, BM, NBST);
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&conv3_halo_kernel<T, BM, NBST, false, LW, MODE>), 160 * 1024, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  const int B = p.M / (p.H * p.W);
  const int nblocks = B * conv3_halo_tiles_per_image(p, BM) * ((p.N + HALO_BN - 1) / HALO_BN) * splitk;
  hipLaunchKernelGGL((conv3_halo_kernel<T, BM, NBST, false, LW, MODE>), dim3(nblocks), dim3(512), smem, stream, q);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

template <typename T, int BM, int LW, int MODE>
static int launch_halo_nbst(const IgemmParams& p, int nbst, int splitk, hipStream_t stream) {
  if (nbst == 2) return launch_halo_cfg<T, BM, 2, LW, MODE>(p, splitk, stream);
  if (nbst == 3) return launch_halo_cfg<T, BM, 3, LW, MODE>(p, splitk, stream);
  if (nbst <= 5) return launch_halo_cfg<T, BM, 4, LW, MODE>(p, splitk, stream);
  return launch_halo_cfg<T, BM, 6, LW, MODE>(p, splitk, stream);
}

template <typename T, int BM, int RB>
static int launch_halo3_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  const size_t smem = halo3_smem_bytes(p, BM, RB);
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&conv3_halo3_kernel<T, BM, RB>), 160 * 1024, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  const int B = p.M / (p.H * p.W);
  const int nblocks = B * conv3_halo_tiles_per_image(p, BM) * ((p.N + HALO_BN - 1) / HALO_BN) * splitk;
  hipLaunchKernelGGL((conv3_halo3_kernel<T, BM, RB>), dim3(nblocks), dim3(512), smem, stream, q);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
template <typename T, int BM>
static int launch_halo3_rb(const IgemmParams& p, int rb, int splitk, hipStream_t stream) {
  if (rb == 2) return launch_halo3_cfg<T, BM, 2>(p, splitk, stream);
  if (rb == 3) return launch_halo3_cfg<T, BM, 3>(p, splitk, stream);
  return launch_halo3_cfg<T, BM, 4>(p, splitk, stream);
}

// developer tool: conv3_halo_kernel<bf16, 256, NBST> with per-tap s_memtime stamps (wave 0 and wave 5 of block 0):
// trace[w][tap][4] = (before the counted vmcnt wait, after it, after the barrier, after the last MFMA was issued)
int launch_conv3_halo_trace(const IgemmParams& p, int dtype, hipStream_t stream) {
  if (dtype != K22_BF16 || !conv3_halo_supported(p, dtype, 256) || p.trace == nullptr) return k22_set_error(K22_EINVAL, "conv3_halo_trace: bf16, BM = 256 only");
  const int nbst = halo_pick_nbst(p, 256) >= 4 ? 4 : 2;
  const size_t smem = halo_smem_bytes(p, 256, nbst);
  IgemmParams q = p;
  q.splitk = 1; q.xcd_remap = 1;
  const int B = p.M / (p.H * p.W);
  const int nblocks = B * conv3_halo_tiles_per_image(p, 256) * ((p.N + HALO_BN - 1) / HALO_BN);
  if (nbst == 4) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<bf16_t, 256, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv3_halo_kernel<bf16_t, 256, 4, true>), dim3(nblocks), dim3(512), smem, stream, q);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<bf16_t, 256, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv3_halo_kernel<bf16_t, 256, 2, true>), dim3(nblocks), dim3(512), smem, stream, q);
  }
  K22_CHECK_LAUNCH();
  return K22_OK;
}

// Launches the halo kernel only (the split-K reduction, if any, is the caller's: launch_igemm).
// p.algo == 3 selects the 64-byte-row kernel (conv3_halo3_kernel), anything else the 128-byte-row one (6 / 7: its pipelined / asm-DMA
// forms; 11 / 12: the specialised kernel of conv3_spec.hip).  (Algos 4 - two wave groups in opposite phases - and 5 - loader-wave
// specialisation - measured 12-20 % / 2-5 % slower in round 1, were never tuner candidates, and were deleted in round 5.)
int launch_conv3_halo(const IgemmParams& p, int dtype, int bm, int splitk, hipStream_t stream) {
  if (!conv3_halo_supported(p, dtype, bm)) return k22_set_error(K22_EINVAL, "conv3_halo: unsupported problem");
  if (k22_is_split(dtype)) {
    int nb = halo_pick_nbst(p, bm);
    if (p.stages >= 2 && p.stages < nb) nb = p.stages == 5 ? 4 : p.stages;
    if (p.algo == 11 || p.algo == 12) return launch_conv3_halo_spec(p, dtype, bm, nb, splitk, stream);
    if (p.algo == 13 || p.algo == 14 || p.algo == 8 || p.algo == 9) return k22_set_error(K22_EINVAL, "conv3_halo: no measurement-only variants in split precision");
    if (dtype == K22_F16X2) return bm == 256 ? launch_halo_nbst<x2_t, 256, 8, 2>(p, nb, splitk, stream) : launch_halo_nbst<x2_t, 128, 8, 2>(p, nb, splitk, stream);
    return bm == 256 ? launch_halo_nbst<x3_t, 256, 8, 2>(p, nb, splitk, stream) : launch_halo_nbst<x3_t, 128, 8, 2>(p, nb, splitk, stream);
  }
  if (p.algo == 3) {
    int rb = halo3_pick_rb(p, bm);
    if (p.stages >= 2 && p.stages < rb) rb = p.stages;
    if (dtype == K22_BF16) return bm == 256 ? launch_halo3_rb<bf16_t, 256>(p, rb, splitk, stream) : launch_halo3_rb<bf16_t, 128>(p, rb, splitk, stream);
    else if (dtype == K22_F16) return bm == 256 ? launch_halo3_rb<f16_t, 256>(p, rb, splitk, stream) : launch_halo3_rb<f16_t, 128>(p, rb, splitk, stream);
    return bm == 256 ? launch_halo3_rb<float, 256>(p, rb, splitk, stream) : launch_halo3_rb<float, 128>(p, rb, splitk, stream);
  }
  int nbst = halo_pick_nbst(p, bm);
  if (p.stages >= 2 && p.stages < nbst) nbst = p.stages == 5 ? 4 : p.stages;  // tuning knob: shallower ring on request
  // producer / consumer wave specialisation (conv3_spec.hip): 11 = compiler-scheduled consumers, 12 = explicit, interleaved fragment pipeline
  if (p.algo == 11 || p.algo == 12) return launch_conv3_halo_spec(p, dtype, bm, nbst, splitk, stream);
#ifdef K22_DEBUG_VARIANTS   // measurement-only kernels (wrong results) are compiled only into a developer build: make CXXFLAGS+=-DK22_DEBUG_VARIANTS
  if (p.algo == 13 || p.algo == 14) {  // measurement-only forms of algo 12 (wrong results): 13 = no LDS-DMA inside the tap loop, 14 = LDS-DMA issued but never waited for
    if (dtype != K22_BF16 || bm != 256) return k22_set_error(K22_EINVAL, "conv3_halo: the debug variants are bf16, BM = 256 only");
    return launch_conv3_halo_spec_debug(p, nbst, splitk, stream);
  }
#else
  if (p.algo == 13 || p.algo == 14 || p.algo == 8 || p.algo == 9) return k22_set_error(K22_EINVAL, "conv3_halo: measurement-only variants need a -DK22_DEBUG_VARIANTS build");
#endif
  if (p.algo == 6) {  // explicit fragment pipeline across the barrier + asm LDS-DMA
    if (dtype == K22_BF16) return bm == 256 ? launch_halo_nbst<bf16_t, 256, 8, 1>(p, nbst, splitk, stream) : launch_halo_nbst<bf16_t, 128, 8, 1>(p, nbst, splitk, stream);
    else if (dtype == K22_F16) return bm == 256 ? launch_halo_nbst<f16_t, 256, 8, 1>(p, nbst, splitk, stream) : launch_halo_nbst<f16_t, 128, 8, 1>(p, nbst, splitk, stream);
    return bm == 256 ? launch_halo_nbst<float, 256, 8, 1>(p, nbst, splitk, stream) : launch_halo_nbst<float, 128, 8, 1>(p, nbst, splitk, stream);
  }
#ifdef K22_DEBUG_VARIANTS
  if (p.algo == 8 || p.algo == 9) {  // measurement-only variants (wrong results): 8 = no LDS-DMA in the loop, 9 = no MFMA
    if (dtype != K22_BF16 || bm != 256) return k22_set_error(K22_EINVAL, "conv3_halo: debug variants are bf16, BM = 256 only");
    return p.algo == 8 ? launch_halo_nbst<bf16_t, 256, 8, 3>(p, nbst, splitk, stream) : launch_halo_nbst<bf16_t, 256, 8, 4>(p, nbst, splitk, stream);
  }
#endif
  if (p.algo == 7) {  // compiler schedule + asm LDS-DMA
    if (dtype == K22_BF16) return bm == 256 ? launch_halo_nbst<bf16_t, 256, 8, 2>(p, nbst, splitk, stream) : launch_halo_nbst<bf16_t, 128, 8, 2>(p, nbst, splitk, stream);
    else if (dtype == K22_F16) return bm == 256 ? launch_halo_nbst<f16_t, 256, 8, 2>(p, nbst, splitk, stream) : launch_halo_nbst<f16_t, 128, 8, 2>(p, nbst, splitk, stream);
    return bm == 256 ? launch_halo_nbst<float, 256, 8, 2>(p, nbst, splitk, stream) : launch_halo_nbst<float, 128, 8, 2>(p, nbst, splitk, stream);
  }
  if (dtype == K22_BF16) return bm == 256 ? launch_halo_nbst<bf16_t, 256, 8, 0>(p, nbst, splitk, stream) : launch_halo_nbst<bf16_t, 128, 8, 0>(p, nbst, splitk, stream);
  else if (dtype == K22_F16) return bm == 256 ? launch_halo_nbst<f16_t, 256, 8, 0>(p, nbst, splitk, stream) : launch_halo_nbst<f16_t, 128, 8, 0>(p, nbst, splitk, stream);
  return bm == 256 ? launch_halo_nbst<float, 256, 8, 0>(p, nbst, splitk, stream) : launch_halo_nbst<float, 128, 8, 0>(p, nbst, splitk, stream);
}
