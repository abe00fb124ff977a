// k22 - conv3_halo_spec_kernel: the LDS-resident-halo 3x3 convolution with producer / consumer wave specialisation (the tile
// table's pick for most convolutions of a step) and, in its producers, the fused GroupNorm-apply (nn.py:26-37, unet.py:150-152,
// 174-180, 212-216).  Split from conv3_halo.hip (frame, LDS images and epilogue described there) so that it builds on its own.
#include "conv3_common.h"

// ================================================================================================================
// conv3_halo_spec_kernel (p.algo == 11 / 12): the same tile, LDS images and epilogue with the eight waves SPECIALISED.
//   waves 0-3 ("consumers", one per SIMD): 2 x 2 over the BM x 128 tile, (BM/2) x 64 per wave (BM = 256: 128 x 64 = 8
//              accumulators, 6 fragment reads per 8 MFMAs instead of 4 per 4); per tap they do nothing but read fragments and
//              issue MFMAs - no LDS-DMA instruction, no vmcnt wait;
//   waves 4-7 ("producers", the second wave of every SIMD): issue all the LDS-DMA of a tap (4 weight pieces + 2 halo pieces
//              each) right after its barrier and sit in the counted vmcnt wait for the next tap's tile.
// Producers and consumers execute the same barriers (one per tap): the ring-slot / halo-buffer reuse argument of
// conv3_halo_kernel holds unchanged.
// PIPE (p.algo == 12): explicit fragment pipeline in the consumers.  The fragments of k-step ks+1 are read while the MFMAs of
// k-step ks are issued (two register sets), and the last k-step of a tap is multiplied after the next tap's barrier, where
// its eight MFMAs cover the latency of that tap's first fragment reads.  Within a block the reads and the MFMAs are
// INTERLEAVED one read behind every MFMA (sched_group_barrier): issued as a read burst and an MFMA burst, the matrix pipe
// idles while the one wave that owns it spends its issue slots on ds_read_b128s.  Measured (bench_kernels, all 3x3
// convolutions of one step): 4.09 ms against 4.51 ms for the best lock-step variant at BM = 256, 4.43 against 4.77 at
// BM = 128; the specialisation alone (algo 11, compiler-scheduled consumers) is +-0 - it is the interleaved pipeline that the
// one-owner matrix pipe makes worthwhile.
// DBG (measurement only, wrong results): 1 (p.algo == 13) = the producers issue nothing inside the tap loop - what is left is
// the consumers' speed limit under the same barriers; 2 (p.algo == 14) = the LDS-DMA is issued but never waited for.  At 96x96
// 768->768, same box: 1.08 PFLOP/s complete, 1.22 without the waits, 1.39 without the loads - half of what the loads cost
// there is the ONE tap a tile has to land in (2-slot ring: the LDS holds the double-buffered halo), half is contention;
// where four slots fit (48x48) the waits cost nothing and the contention is the same 12-14 %.  Staging the weight tiles
// through producer registers (global_load two taps ahead, ds_write_b128 into the 2-slot ring) was built and measured equal
// to the LDS-DMA form at 96x96 and slower elsewhere; removed.  So was a four-block form of the consumer pipeline that reads the
// weight fragments of the last k-step one block early (only halo reads in flight at the barrier, no full read wait with the
// 2-slot ring): same-box A/B/A/B 1.718-1.740 ms for all three forms on the 96x96 / 48x48 shapes.
// ================================================================================================================
template <typename T, int BM, int NBST, bool PIPE = false, int DBG = 0>
__global__ __launch_bounds__(512) void conv3_halo_spec_kernel(const IgemmParams p) {
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int BN = HALO_BN, NWL = 4, WM = 2, WN = 2;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int B_SLOTS = BN / 8 / NWL;        // 4 weight LDS-DMA instructions per producer per tap
  constexpr int B_BYTES = BN * 128;
  constexpr int APT = 2;                       // halo pieces per producer per tap (taps 0 .. 9-NBST)
  constexpr int A_SLOTS = (10 - NBST) * APT;
  constexpr int GM = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;             // wave-uniform
  const int h = lane >> 5, l31 = lane & 31;

  const int W2 = p.W + 2;
  const int VR = p.H * W2;
  const int TPI = (VR + BM - 1) / BM;
  const int HRp = (BM + 2 * W2 + 2 + 7) & ~7;
  const int NP = HRp >> 3;
  const int A_BYTES = HRp * 128;
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

  char* const Bst = smem + 2 * A_BYTES;
  if (s0 < s1) {
    if (producer && p.gn_coeff != nullptr) {
      // ------------------------- producers, fused GroupNorm-apply: LDS-DMA of the RAW tensor, then rewrite in place ---------------------
      // The halo piece a producer wave loaded is rewritten by the SAME wave one tap later (its own counted vmcnt orders the ds_read behind
      // the DMA; no other wave touches the buffer until the barrier that opens the next slab): y = act(x * A[c] + Bc[c]), zero at the
      // border positions of the padded plane - which the raw tensor does not have: the per-lane source is the unpadded pixel (clamped at
      // the border; those lanes' values are discarded).  A lane's 16 bytes are the same channel chunk in every piece of its wave
      // (chunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7): the piece index has the wave's parity), so its 2 * EPC coefficients
      // live in registers, re-loaded once per slab.  Surplus slots (pieces past the halo; the lock-step counting wants every wave to issue the
      // same number of DMAs) do not re-load the last piece - that would land raw bytes on rewritten ones - but 1 KB of the weights into a
      // scratch piece behind the weight ring that nobody reads (a plain load with a register destination would do for the counting, but a
      // register written by a load the compiler does not know about is a register it re-uses meanwhile).
      const int lw = wave - 4;
      constexpr int NCF = EPC / 2;                   // 16-byte registers of coefficients per lane: (A, Bc) x EPC channels
      const T* __restrict__ X0 = reinterpret_cast<const T*>(p.gn_x0) + (int64_t)img * p.H * p.W * p.gn_C0;
      const int C1 = p.Kc - p.gn_C0;
      const T* __restrict__ X1 = p.gn_x1 ? reinterpret_cast<const T*>(p.gn_x1) + (int64_t)img * p.H * p.W * C1 : nullptr;
      const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);
      const int gchunk = (lane & 7) ^ ((4 * (lw & 1) + (lane >> 4)) & 7);
      const float* __restrict__ cfbase = p.gn_coeff + ((int64_t)img * p.Kc + gchunk * EPC) * 2;
      int pix[A_SLOTS];
      unsigned bmask = 0;
#pragma unroll
      for (int q = 0; q < A_SLOTS; ++q) {
        int j = q * NWL + lw;
        if (j > NP - 1) j = NP - 1;
        const int hr = 8 * j + (lane >> 3);
        int pr = v0 + hr;
        if (pr > PR_MAX) pr = PR_MAX;
        const int y = pr / W2, x = pr - y * W2;          // padded coordinates
        const bool border = y < 1 || y > p.H || x < 1 || x > p.W;
        const int yy = y < 1 ? 0 : (y > p.H ? p.H - 1 : y - 1), xx = x < 1 ? 0 : (x > p.W ? p.W - 1 : x - 1);
        pix[q] = yy * p.W + xx;
        bmask |= (border ? 1u : 0u) << q;
      }
      int boff[B_SLOTS];
#pragma unroll
      for (int i = 0; i < B_SLOTS; ++i) {
        const int row = 8 * (lw + NWL * i) + (lane >> 3);
        int n = n0 + row;
        if (n > p.Npad - 1) n = p.Npad - 1;
        boff[i] = n * 9 * p.Kc + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
      }
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
      const unsigned lds_scratch = lds0 + 2 * A_BYTES + NBST * B_BYTES;
      u32x4_t cfr[NCF];
#pragma unroll
      for (int k = 0; k < NCF; ++k) cfr[k] = u32x4_t{0u, 0u, 0u, 0u};
#define K22_GN_A(Q, SLAB, DSTOFF)                                                                          \
      {                                                                                                    \
        const int jn_ = (Q) * NWL + lw;                                                                    \
        if (jn_ > NP - 1) {                                                                                \
          glds16_asm(reinterpret_cast<const char*>(p.Wp) + lane * 16, __builtin_amdgcn_readfirstlane(lds_scratch)); \
        } else {                                                                                           \
          const int k0_ = (SLAB) * BK;                                                                     \
          const bool second_ = k0_ >= p.gn_C0;                                                             \
          const T* sp_ = (second_ ? X1 + (int64_t)pix[Q] * C1 + (k0_ - p.gn_C0) : X0 + (int64_t)pix[Q] * p.gn_C0 + k0_) + gchunk * EPC; \
          glds16_asm(sp_, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(DSTOFF) + jn_ * 1024));         \
        }                                                                                                  \
      }
#define K22_GN_B(SLAB, TAP, STAGE)                                                                         \
      {                                                                                                    \
        const int kofs_ = (TAP) * p.Kc + (SLAB) * BK;                                                      \
        const unsigned d_ = lds0 + 2 * A_BYTES + (STAGE) * B_BYTES + lw * 1024;                            \
        _Pragma("unroll") for (int i = 0; i < B_SLOTS; ++i)                                                \
            glds16w_asm(Wp + boff[i] + kofs_, __builtin_amdgcn_readfirstlane(d_ + i * NWL * 1024));         \
      }
#define K22_GN_COEF(SLAB)                                                                                  \
      {                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < NCF; ++k) gload16_asm(cfr[k], cfbase + (int64_t)(SLAB) * BK * 2 + 4 * k); \
      }
#define K22_GN_REWRITE(Q, DSTOFF)                                                                          \
      {                                                                                                    \
        const int jn_ = (Q) * NWL + lw;                                                                    \
        if (jn_ <= NP - 1) {                                                                               \
          u32x4_t* a_ = reinterpret_cast<u32x4_t*>(smem + (DSTOFF) + jn_ * 1024 + lane * 16);              \
          float cf_[2 * EPC];                                                                              \
          _Pragma("unroll") for (int k = 0; k < NCF; ++k) {                                                \
            cf_[4 * k] = __uint_as_float(cfr[k].x); cf_[4 * k + 1] = __uint_as_float(cfr[k].y);            \
            cf_[4 * k + 2] = __uint_as_float(cfr[k].z); cf_[4 * k + 3] = __uint_as_float(cfr[k].w);        \
          }                                                                                                \
          if constexpr (is_x3<T>::value) {   /* groups of eight: this lane's four values own one half of the hi piece and of the lo piece   \
                                                 (logical chunks 2g, 2g + 1 = this lane's and its neighbour's 16 bytes; both lanes read before they write) */ \
            const u32x4_t s_ = gn_rewrite16(T{}, *a_, cf_, p.gn_act, ((bmask >> (Q)) & 1u) != 0u);        \
            const int od_ = gchunk & 1, o_ = (DSTOFF) + jn_ * 1024 + lane * 16;                            \
            *reinterpret_cast<u32x2_t*>(smem + (o_ ^ (od_ << 4)) + 8 * od_) = u32x2_t{s_.x, s_.y};         \
            *reinterpret_cast<u32x2_t*>(smem + (o_ ^ (od_ << 4) ^ 16) + 8 * od_) = u32x2_t{s_.z, s_.w};    \
          } else                                                                                           \
          *a_ = gn_rewrite16(T{}, *a_, cf_, p.gn_act, ((bmask >> (Q)) & 1u) != 0u);                        \
        }                                                                                                  \
      }
      // prologue: coefficients of the first slab, its whole halo, the weight tiles of taps 0 .. NBST-2; then the halo is rewritten
      K22_GN_COEF(s0);
#pragma unroll
      for (int q = 0; q < A_SLOTS; ++q) K22_GN_A(q, s0, 0);
#pragma unroll
      for (int t = 0; t < NBST - 1; ++t) K22_GN_B(s0, t, t);
      gn_wait<(NBST - 1) * B_SLOTS>(cfr);
#pragma unroll
      for (int q = 0; q < A_SLOTS; ++q) K22_GN_REWRITE(q, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      int fill = NBST - 1;
      for (int s = s0; s < s1; ++s) {
        const int anext_off = (((s - s0) & 1) ^ 1) * A_BYTES;
        const int sn = s + 1 < s1 ? s + 1 : s1 - 1;   // past-the-end loads re-read the last slab (uniform counting)
        const bool more = s + 1 < s1;                 // a next slab exists: its halo is rewritten as it lands
#define K22_GN_PTAP(TAP)                                                                                   \
        {                                                                                                  \
          wait_vmcnt<B_SLOTS * (NBST - 2) + APT * halo_count_a<NBST>(TAP)>();                              \
          raw_barrier();                                                                                   \
          if constexpr ((TAP) == 0) { if (more) K22_GN_COEF(sn); }                                         \
          constexpr int ta_ = ((TAP) + NBST - 1) % 9;                                                      \
          const int sa_ = ((TAP) + NBST - 1 >= 9) ? sn : s;                                                \
          K22_GN_B(sa_, ta_, fill);                                                                        \
          if constexpr ((TAP) <= 9 - NBST) {                                                               \
            _Pragma("unroll") for (int a_i = 0; a_i < APT; ++a_i) {                                        \
              constexpr int qb_ = ((TAP) <= 9 - NBST ? (TAP) : 0) * APT;                                   \
              K22_GN_A(qb_ + a_i, sn, anext_off);                                                          \
            }                                                                                              \
          }                                                                                                \
          /* the pieces issued one tap ago (and the coefficients issued at tap 0) have had a whole tap to land: wait for everything   \
             older than THIS tap's DMA, rewrite them, and have the stores complete before the next barrier */                        \
          if constexpr ((TAP) >= 1 && (TAP) - 1 <= 9 - NBST) {                                             \
            if (more) {                                                                                    \
              gn_wait<B_SLOTS + ((TAP) <= 9 - NBST ? APT : 0)>(cfr);                                       \
              _Pragma("unroll") for (int a_i = 0; a_i < APT; ++a_i) {                                      \
                constexpr int qr_ = ((TAP) >= 1 ? (TAP) - 1 : 0) * APT;                                    \
                K22_GN_REWRITE(qr_ + a_i, anext_off);                                                      \
              }                                                                                            \
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
            }                                                                                              \
          }                                                                                                \
          fill = (fill + 1 == NBST) ? 0 : fill + 1;                                                        \
        }
        K22_GN_PTAP(0) K22_GN_PTAP(1) K22_GN_PTAP(2) K22_GN_PTAP(3) K22_GN_PTAP(4) K22_GN_PTAP(5) K22_GN_PTAP(6) K22_GN_PTAP(7) K22_GN_PTAP(8)
#undef K22_GN_PTAP
      }
#undef K22_GN_A
#undef K22_GN_B
#undef K22_GN_COEF
#undef K22_GN_REWRITE
    } else if (producer) {
      // ---------------------------------------- producers: LDS-DMA only -----------------------------------------------
      const int lw = wave - 4;
      const T* __restrict__ Aimg = reinterpret_cast<const T*>(p.A0) + (int64_t)img * (p.H + 2) * W2 * p.Kc;
      const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);
      int aoff[A_SLOTS];
#pragma unroll
      for (int q = 0; q < A_SLOTS; ++q) {
        int j = q * NWL + lw;
        if (j > NP - 1) j = NP - 1;   // surplus slots re-load the last piece (same bytes, same place): uniform counting
        const int hr = 8 * j + (lane >> 3);
        int pr = v0 + hr;
        if (pr > PR_MAX) pr = PR_MAX;
        aoff[q] = pr * p.Kc + ((lane & 7) ^ ((hr >> 1) & 7)) * EPC;
      }
      int boff[B_SLOTS];
#pragma unroll
      for (int i = 0; i < B_SLOTS; ++i) {
        const int row = 8 * (lw + NWL * i) + (lane >> 3);
        int n = n0 + row;
        if (n > p.Npad - 1) n = p.Npad - 1;
        boff[i] = n * 9 * p.Kc + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
      }
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
#define K22_SP_A(Q, SLAB, DSTOFF)                                                                          \
      {                                                                                                    \
        int j_ = (Q) * NWL + lw;                                                                           \
        if (j_ > NP - 1) j_ = NP - 1;                                                                      \
        glds16_asm(Aimg + aoff[Q] + (SLAB) * BK, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(DSTOFF) + j_ * 1024)); \
      }
#define K22_SP_B(SLAB, TAP, STAGE)                                                                         \
      {                                                                                                    \
        const int kofs_ = (TAP) * p.Kc + (SLAB) * BK;                                                      \
        const unsigned d_ = lds0 + 2 * A_BYTES + (STAGE) * B_BYTES + lw * 1024;                            \
        _Pragma("unroll") for (int i = 0; i < B_SLOTS; ++i)                                                \
            glds16w_asm(Wp + boff[i] + kofs_, __builtin_amdgcn_readfirstlane(d_ + i * NWL * 1024));         \
      }
#pragma unroll
      for (int q = 0; q < A_SLOTS; ++q) K22_SP_A(q, s0, 0);
#pragma unroll
      for (int t = 0; t < NBST - 1; ++t) K22_SP_B(s0, t, t);
      int fill = NBST - 1;
      for (int s = s0; s < s1; ++s) {
        const int anext_off = (((s - s0) & 1) ^ 1) * A_BYTES;
        const int sn = s + 1 < s1 ? s + 1 : s1 - 1;   // past-the-end loads re-read the last slab (uniform counting)
#define K22_SP_PTAP(TAP)                                                                                   \
        {                                                                                                  \
          if constexpr (DBG == 0) wait_vmcnt<B_SLOTS * (NBST - 2) + APT * halo_count_a<NBST>(TAP)>();      \
          raw_barrier();                                                                                   \
          constexpr int ta_ = ((TAP) + NBST - 1) % 9;                                                      \
          const int sa_ = ((TAP) + NBST - 1 >= 9) ? sn : s;                                                \
          if constexpr (DBG != 1) K22_SP_B(sa_, ta_, fill);                                                \
          if constexpr ((TAP) <= 9 - NBST && DBG != 1) {                                                   \
            _Pragma("unroll") for (int a_ = 0; a_ < APT; ++a_) {                                           \
              constexpr int qb_ = ((TAP) <= 9 - NBST ? (TAP) : 0) * APT;                                   \
              K22_SP_A(qb_ + a_, sn, anext_off);                                                           \
            }                                                                                              \
          }                                                                                                \
          fill = (fill + 1 == NBST) ? 0 : fill + 1;                                                        \
        }
        K22_SP_PTAP(0) K22_SP_PTAP(1) K22_SP_PTAP(2) K22_SP_PTAP(3) K22_SP_PTAP(4) K22_SP_PTAP(5) K22_SP_PTAP(6) K22_SP_PTAP(7) K22_SP_PTAP(8)
#undef K22_SP_PTAP
      }
#undef K22_SP_A
#undef K22_SP_B
    } else {
      // ---------------------------------------- consumers: fragments + MFMA only --------------------------------------
      const int wm = wave >> 1, wn = wave & 1;
      const int abase = wm * (BM / WM) + l31;
      int brow[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) brow[ni] = (wn * (BN / WN) + ni * 32 + l31) * 128;
      const int bsw = (l31 >> 1) & 7;
      int cur = 0;
      // (s_setprio 3 in the consumers, so that they win issue arbitration against the producer on their SIMD: measured +-1 %)
      FragA<T> pa[MI];          // PIPE: fragments read but not yet multiplied (zero = a no-op group before the first tap)
      Frag<T> pb[NI];
      if constexpr (PIPE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) pa[mi] = FragA<T>{};
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) pb[ni] = Frag<T>{};
      }
      // One block of the pipeline = NRD fragment reads (of the NEXT group) + NMF MFMAs (of the group read one block ago), which
      // are independent of each other: one read is scheduled behind every MPR MFMAs (0x008 = MFMA, 0x100 = DS read), so that
      // each ds_read_b128 issues under the 32 cycles the MFMA before it occupies the pipe.
      constexpr int NRD = is_x2<T>::value ? MI + 2 * NI : (MI + NI) * FragCost<T>::READS;   // x2: the activation fragment is ONE read (hi piece only)
      constexpr int NMF = MI * NI * FragCost<T>::MFMAS;
      constexpr int MPR = NMF / NRD;
#define K22_SP_INTERLEAVE()                                                                                \
      {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NRD; ++i_) {                                               \
          __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
        }                                                                                                  \
        if constexpr (NMF - MPR * NRD > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - MPR * NRD, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
      }
      for (int s = s0; s < s1; ++s) {
        const char* const Acur = smem + ((s - s0) & 1) * A_BYTES;
#define K22_SP_CTAP(TAP)                                                                                   \
        {                                                                                                  \
          raw_barrier();                                                                                   \
          const char* Bcur = Bst + cur * B_BYTES;                                                          \
          const int shift = ((TAP) / 3) * W2 + ((TAP) % 3);                                                \
          const char* arow[MI];                                                                            \
          int asw[MI];                                                                                     \
          _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                              \
            const int ar = abase + mi * 32 + shift;                                                        \
            arow[mi] = Acur + ar * 128;                                                                    \
            asw[mi] = (ar >> 1) & 7;                                                                       \
          }                                                                                                \
          if constexpr (PIPE) {                                                                            \
            FragA<T> ca[MI];                                                                               \
            Frag<T> cb[NI];                                                                                \
            _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ks += 2) {                                     \
              _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag_at(ca[mi], arow[mi], asw[mi], ks, h); \
              _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag_at(cb[ni], Bcur + brow[ni], bsw, ks, h); \
              _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                            \
                _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], pb[ni], pa[mi]);   \
              K22_SP_INTERLEAVE();                                                                         \
              _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag_at(pa[mi], arow[mi], asw[mi], ks + 1, h); \
              _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag_at(pb[ni], Bcur + brow[ni], bsw, ks + 1, h); \
              _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                            \
                _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], cb[ni], ca[mi]);   \
              K22_SP_INTERLEAVE();                                                                         \
            }                                                                                              \
          } else {                                                                                         \
          _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                          \
            FragA<T> a[MI];                                                                                \
            Frag<T> b[NI];                                                                                 \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) ld_frag_at(a[mi], arow[mi], asw[mi], ks, h); \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) ld_frag_at(b[ni], Bcur + brow[ni], bsw, ks, h); \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                              \
              _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], b[ni], a[mi]);       \
          }                                                                                                \
          }                                                                                                \
          /* Fragment reads may not stay in flight across the next barrier: right after barrier TAP+1 the producers refill    \
             slot (TAP + NBST) % NBST = the weight slot THIS tap read, whatever the ring depth (and after tap 8 the halo buffer \
             of the slab before).  Only the DMA's latency kept the NBST >= 3 forms correct in round 2; now nothing is left to   \
             timing.  (The MFMAs that use the fragments can still sink below the barrier - registers only - in both forms.) */  \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
          cur = (cur + 1 == NBST) ? 0 : cur + 1;                                                           \
        }
        K22_SP_CTAP(0) K22_SP_CTAP(1) K22_SP_CTAP(2) K22_SP_CTAP(3) K22_SP_CTAP(4) K22_SP_CTAP(5) K22_SP_CTAP(6) K22_SP_CTAP(7) K22_SP_CTAP(8)
#undef K22_SP_CTAP
      }
#undef K22_SP_INTERLEAVE
      if constexpr (PIPE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], pb[ni], pa[mi]);
      }
    }
  }
  halo_tail<T, BM, false, true>(p, acc, smem, bx, bz, img, v0, n0);
}


template <typename T, int BM, int NBST, bool PIPE, int DBG = 0>
static int launch_halo_spec_cfg(const IgemmParams& p, int splitk, hipStream_t stream) {
  const size_t smem = halo_smem_bytes(p, BM, NBST);
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&conv3_halo_spec_kernel<T, BM, NBST, PIPE, DBG>), 160 * 1024, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  const int B = p.M / (p.H * p.W);
  const int nblocks = B * conv3_halo_tiles_per_image(p, BM) * ((p.N + HALO_BN - 1) / HALO_BN) * splitk;
  hipLaunchKernelGGL((conv3_halo_spec_kernel<T, BM, NBST, PIPE, DBG>), dim3(nblocks), dim3(512), smem, stream, q);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
template <typename T, int BM, bool PIPE>
static int launch_halo_spec_nbst(const IgemmParams& p, int nbst, int splitk, hipStream_t stream) {
  if (nbst == 2) return launch_halo_spec_cfg<T, BM, 2, PIPE>(p, splitk, stream);
  if (nbst == 3) return launch_halo_spec_cfg<T, BM, 3, PIPE>(p, splitk, stream);
  if (nbst <= 5) return launch_halo_spec_cfg<T, BM, 4, PIPE>(p, splitk, stream);
  return launch_halo_spec_cfg<T, BM, 6, PIPE>(p, splitk, stream);
}

// p.algo == 11 (compiler-scheduled consumers) / 12 (explicit, interleaved fragment pipeline); nbst = ring depth already chosen by the caller
int launch_conv3_halo_spec(const IgemmParams& p, int dtype, int bm, int nbst, int splitk, hipStream_t stream) {
  const bool pipe = p.algo == 12;
  if (dtype == K22_F16X2) {
    // asymmetric split: activation fragments are 4 registers (hi halves only), so the two-set pipeline fits at BM = 256 too
    if (!pipe) return bm == 256 ? launch_halo_spec_nbst<x2_t, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<x2_t, 128, false>(p, nbst, splitk, stream);
    return bm == 256 ? launch_halo_spec_nbst<x2_t, 256, true>(p, nbst, splitk, stream) : launch_halo_spec_nbst<x2_t, 128, true>(p, nbst, splitk, stream);
  }
  if (dtype == K22_F16X3) {
    // BM = 256: two fragment sets of 8 registers per fragment do not fit beside 128 accumulators (as for fp32): compiler-scheduled consumers
    if (!pipe) return bm == 256 ? launch_halo_spec_nbst<x3_t, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<x3_t, 128, false>(p, nbst, splitk, stream);
    return bm == 256 ? launch_halo_spec_nbst<x3_t, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<x3_t, 128, true>(p, nbst, splitk, stream);
  }
  if (!pipe) {
    if (dtype == K22_BF16) return bm == 256 ? launch_halo_spec_nbst<bf16_t, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<bf16_t, 128, false>(p, nbst, splitk, stream);
    else if (dtype == K22_F16) return bm == 256 ? launch_halo_spec_nbst<f16_t, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<f16_t, 128, false>(p, nbst, splitk, stream);
    return bm == 256 ? launch_halo_spec_nbst<float, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<float, 128, false>(p, nbst, splitk, stream);
  }
  if (dtype == K22_BF16) return bm == 256 ? launch_halo_spec_nbst<bf16_t, 256, true>(p, nbst, splitk, stream) : launch_halo_spec_nbst<bf16_t, 128, true>(p, nbst, splitk, stream);
  else if (dtype == K22_F16) return bm == 256 ? launch_halo_spec_nbst<f16_t, 256, true>(p, nbst, splitk, stream) : launch_halo_spec_nbst<f16_t, 128, true>(p, nbst, splitk, stream);
  // fp32, BM = 256: two fragment sets of 8 registers per fragment do not fit beside 128 accumulators (the pipelined form spills):
  // the compiler-scheduled consumer is used there
  return bm == 256 ? launch_halo_spec_nbst<float, 256, false>(p, nbst, splitk, stream) : launch_halo_spec_nbst<float, 128, true>(p, nbst, splitk, stream);
}
#ifdef K22_DEBUG_VARIANTS   // measurement-only forms of algo 12 (wrong results): 13 = no LDS-DMA inside the tap loop, 14 = LDS-DMA issued but never waited for
int launch_conv3_halo_spec_debug(const IgemmParams& p, int nbst, int splitk, hipStream_t stream) {
  if (p.algo == 13) return nbst == 2 ? launch_halo_spec_cfg<bf16_t, 256, 2, true, 1>(p, splitk, stream) : launch_halo_spec_cfg<bf16_t, 256, 4, true, 1>(p, splitk, stream);
  return nbst == 2 ? launch_halo_spec_cfg<bf16_t, 256, 2, true, 2>(p, splitk, stream) : launch_halo_spec_cfg<bf16_t, 256, 4, true, 2>(p, splitk, stream);
}
#endif
