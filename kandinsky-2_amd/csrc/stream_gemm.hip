// k22 — weight-streaming kernel for SMALL-M 3x3 convolutions and GEMMs (the 12x12 / 24x24 levels of the UNet, the qkv /
// proj_out projections there, the prior's and the conditioning towers' Linears: M = a few hundred rows).
//
// Replaces, for those shapes: nn.Conv2d 3x3 of the ResBlocks (kandinsky2/model/unet.py:152,180) + the fused 1x1
// skip_connection (:191), Conv1d qkv / proj_out (:251,258), nn.Linear of PriorTransformer (prior.py:93-120).
//
// Why another kernel: at M = 288 (2 x 12 x 12) a 1536 -> 1536 convolution is 12 GFLOP over 42 MB of weights - 5 us of MFMA
// and 5-8 us of HBM - but the LDS-resident halo kernel (conv3_halo.hip) spends 33 us on it: its 256-row tile is 44 %
// padding, both m-tiles pull every weight tile, and split-K 8-10 over 128-wide n-tiles writes and re-reads 18 MB of fp32
// partials.  Here the roles are swapped:
//   * the WEIGHTS never touch the LDS: every wave reads its own B fragments straight from global memory into registers
//     (lane = weight row, 16 B = its 8 k-values), through a register ring 8-9 (slab, tap) items deep, so ~18 KB per wave are
//     in flight against the HBM latency; each weight byte is fetched by exactly one wave of one workgroup per m-tile;
//   * the ACTIVATIONS of the whole m-tile (a band of image rows + halo, <= 288 pixels) sit in the LDS as a zero-bordered
//     plane, one 64-channel slab at a time (double-buffered, filled through registers: ~25 KB per slab), and are re-used by
//     the nine taps through per-lane plane-row offsets: compact rows, no padding rows in the MFMA tile;
//   * the four waves of a workgroup split K: wave w multiplies channels [16w, 16w+16) of every slab for the whole
//     (<= MB*32) x 64 tile, so one A fragment read feeds two MFMAs and the n-tile is only 64 wide: 24 n-tiles x 2 m-tiles
//     at 12x12 fill the chip with split-K 5 instead of 10.  The four partial tiles are folded through the LDS in a fixed
//     order (wave 0 + 1 + 2 + 3) before anything leaves the CU;
//   * the workgroup writes ONE fp32 partial tile per (m-tile, n-tile, k-range); the existing split-K finish
//     (splitk_reduce_rows_kernel / splitk_reduce_kernel, igemm.hip) adds bias / residual / activation, rounds once, and
//     emits the GroupNorm partial sums - the same epilogue arithmetic, in the same order, as every other split-K launch.
// bf16 / fp16 storage only (the K split maps the four 16-wide MFMA k-steps of a 128-byte row to the four waves).
#include "kernels.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ int xcd_remap_stream(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int MB, int TAPS> struct StreamCfg {
  static constexpr int G = (TAPS == 1 && MB <= 5) ? 2 : 1;   // 64-channel slabs per LDS stage
  static constexpr int SITEMS = G * TAPS;                      // (slab, tap) items per stage
  static constexpr int D = TAPS == 9 ? 9 : 8;                  // depth of the weight-fragment register ring (items)
  static constexpr int SPI = D / SITEMS;                       // stages per unrolled outer iteration
  static constexpr int NP = TAPS == 9 ? (MB <= 5 ? 8 : 12) : MB;   // 32-row fill passes per slab (plane rows <= 32 NP; rows past the plane re-read its last row)
};

// One K phase: acc += A(plane rows, channels of slabs [s0, s1)) x W^T for the workgroup's 64 weight rows.
//   plane row r, channel c  =  c < K0 ? a0[r * ld0 + c] : a1[r * ld1 + c - K0]      (r < PR; rows are clamped to PR - 1)
//   item (slab s, tap t): lane's A fragment of m-block i = plane row prow[i] + (t / 3) * W2 + t % 3, channels 64 s + 16 w + 8 h ..
//                         lane's B fragment of n-block j = wl{j}[t * wtap + 64 s .. +8]
template <typename T, int MB, int NB, int TAPS, int DBG, bool FRAG>
__device__ __forceinline__ void stream_phase(f32x16_t (&acc)[MB][NB], char* smem, const int buf_bytes,
                                             const T* __restrict__ a0, const int64_t ld0, const int K0,
                                             const T* __restrict__ a1, const int64_t ld1, const int PR,
                                             const int (&prow)[MB], const int W2,
                                             const T* __restrict__ wl0, const T* __restrict__ wl1, const int64_t wtap,
                                             const int s0, const int s1, const int w, const int tid,
                                             unsigned long long* ts = nullptr) {
  using C = StreamCfg<MB, TAPS>;
  int nts = 2;
  const int nslab = s1 - s0;
  if (nslab <= 0) return;   // uniform over the workgroup
  const int nstage = (nslab + C::G - 1) / C::G;
  const int h = (tid & 63) >> 5;
  const int frow = tid >> 3, fpos = tid & 7;
  constexpr int sub_bytes = C::NP * 32 * 128;

  // Every global load below is issued UNCONDITIONALLY (indices clamped to the last stage / item): the compiler's vmcnt
  // bookkeeping then sees one straight-line sequence and keeps the whole weight ring in flight; only LDS reads and MFMAs sit
  // behind the (workgroup-uniform) tail guards.
  u32x4_t stg[C::G][C::NP];
  auto fill_load = [&](int stage) __attribute__((always_inline)) {
    if (stage > nstage - 1) stage = nstage - 1;
#pragma unroll
    for (int g = 0; g < C::G; ++g) {
      const int sreq = s0 + stage * C::G + g;
      const int s = sreq > s1 - 1 ? s1 - 1 : sreq;
#pragma unroll
      for (int q = 0; q < C::NP; ++q) {
        const int r = frow + 32 * q;
        const int rr = r < PR ? r : PR - 1;
        const int c = s * 64 + ((fpos ^ ((r >> 1) & 7)) << 3);
        const T* src = c < K0 ? a0 + (int64_t)rr * ld0 + c : a1 + (int64_t)rr * ld1 + (c - K0);
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(src);
        if (C::G > 1 && sreq > s1 - 1) v = (u32x4_t){0u, 0u, 0u, 0u};   // odd tail of a two-slab stage: a zero slab (multiplied, adds 0)
        stg[g][q] = v;
      }
    }
  };
  auto fill_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < C::G; ++g)
#pragma unroll
      for (int q = 0; q < C::NP; ++q)
        *reinterpret_cast<u32x4_t*>(smem + buf * buf_bytes + g * sub_bytes + (frow + 32 * q) * 128 + fpos * 16) = stg[g][q];
  };
  Frag<T> breg[C::D][NB];
  const int last_item = nstage * C::SITEMS - 1;   // items of a zero slab included: their weights are the last real slab's
  auto load_b = [&](int it, Frag<T> (&f)[NB]) __attribute__((always_inline)) {
    if (it > last_item) it = last_item;
    int sl = it / TAPS;
    const int tap = it - sl * TAPS;
    if (sl > nslab - 1) sl = nslab - 1;
    // FRAG: wl{j} point at this lane's 16 bytes of the fragment of (n-block j, item 0 of the op, k quarter w); items are 4 x 1 KB apart
    const int64_t off = FRAG ? (int64_t)((s0 + sl) * TAPS + tap) * 2048 : (int64_t)tap * wtap + (int64_t)(s0 + sl) * 64;
#ifdef K22_W_NT   // measurement build: non-temporal weight stream
    f[0].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wl0 + off));
    if (NB > 1) f[NB - 1].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wl1 + off));
#else
    f[0] = *reinterpret_cast<const Frag<T>*>(wl0 + off);
    if (NB > 1) f[NB - 1] = *reinterpret_cast<const Frag<T>*>(wl1 + off);
#endif
  };

  fill_load(0);
#pragma unroll
  for (int d = 0; d < C::D; ++d) load_b(d, breg[d]);
  fill_store(0);
  __syncthreads();
  if (DBG & 8) ts[1] = __builtin_readcyclecounter();   // first slab in the LDS
  int cur = 0;
  for (int st = 0; st < nstage; st += C::SPI) {
#pragma unroll
    for (int u = 0; u < C::SPI; ++u) {
      const int stage = st + u;
      fill_load(stage + 1);
      if (stage < nstage) {   // uniform over the workgroup; no global loads inside
        const char* abuf = smem + cur * buf_bytes;
        // explicit two-set fragment pipeline: the A fragments of item k+1 are read while the MFMAs of item k run
        Frag<T> a[2][MB];
        auto read_a = [&](int k, Frag<T> (&f)[MB]) __attribute__((always_inline)) {
          const int g = k / TAPS, tap = k - g * TAPS;
          const int shift = TAPS == 9 ? (tap / 3) * W2 + (tap % 3) : 0;
          const char* sub = abuf + g * sub_bytes;
#pragma unroll
          for (int i = 0; i < MB; ++i) {
            if ((DBG & 4) && (k | stage | i) != 0) f[i] = f[0];   // measurement only: one LDS read per phase
            else ld_frag(f[i], sub, prow[i] + shift, w, h);
          }
        };
        read_a(0, a[0]);
        __builtin_amdgcn_sched_barrier(0);     // the first item's reads issue back to back (their own region), not one per MFMA pair
#pragma unroll
        for (int k = 0; k < C::SITEMS; ++k) {
          const int d = u * C::SITEMS + k;   // compile-time after unrolling: the ring slot
          if (k + 1 < C::SITEMS) read_a(k + 1, a[(k + 1) & 1]);
#pragma unroll
          for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              if (DBG & 1) acc[i][j][0] += __builtin_bit_cast(float, breg[d][j].v[0] ^ a[k & 1][i].v[0]);   // measurement only: no MFMA
              else mma_atom(acc[i][j], breg[d][j], a[k & 1][i]);   // C^T: rows = weight rows (channels), columns = pixels
            }
          // one fragment read of the NEXT item behind every NB MFMAs of this one (0x008 = MFMA, 0x100 = DS read): left to itself the
          // scheduler keeps two fragment registers and puts a full LDS round trip in front of every pair of MFMAs
          if (k + 1 < C::SITEMS && (DBG & 7) == 0) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          // TAPS == 9 (one stage = the whole ring): slot k is refilled as soon as item k's MFMAs are issued, so the next stage's first
          // fragments have a whole stage of MFMA time to arrive; refilled in bulk after the stage they were an exposed HBM round trip
          // per slab (stage < nstage always holds here: SPI == 1)
          if (C::SPI == 1 && !(DBG & 2)) load_b(stage * C::SITEMS + k + C::D, breg[d]);
          __builtin_amdgcn_sched_barrier(0);   // one item = one scheduling region: nothing of item k+1's MFMAs moves up into this one
        }
      }
      if (C::SPI != 1) {
#pragma unroll
        for (int k = 0; k < C::SITEMS; ++k)
          if (!(DBG & 2)) load_b(stage * C::SITEMS + k + C::D, breg[u * C::SITEMS + k]);   // DBG 2 (measurement only): the ring is never refilled
      }
      fill_store(cur ^ 1);
      __syncthreads();
      if ((DBG & 8) && nts < 11) ts[nts++] = __builtin_readcyclecounter();
      cur ^= 1;
    }
  }
}

template <typename T, int MB, int NB, int DBG, bool FRAG>
__global__ __launch_bounds__(256) void stream_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long ts[16];
  if (DBG & 8) {
#pragma unroll
    for (int i = 0; i < 16; ++i) ts[i] = 0;
    ts[0] = __builtin_readcyclecounter();
    ts[11] = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock: places the workgroups of a launch on one time axis
  }

  constexpr int NT = 32 * NB;
  const int n_tiles = p.Npad / NT;
  int L = p.xcd_remap ? xcd_remap_stream(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int mt = L % p.st_mtiles;
  L /= p.st_mtiles;
  const int nt = L % n_tiles, bz = L / n_tiles;
  const int n0 = nt * NT;
  const int TM = p.st_tm;
  const int m0 = mt * TM;
  const int rows = p.M - m0 < TM ? p.M - m0 : TM;
  const int mb_used = (rows + 31) >> 5;
  const bool conv = p.taps == 9;
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.Wp);

  f32x16_t acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main phase --------------------------------------------------------------------------------------------
  {
    const int nslab = p.Kc >> 6;
    const int per = (nslab + p.splitk - 1) / p.splitk;
    const int s0 = bz * per, s1 = s0 + per < nslab ? s0 + per : nslab;
    const int64_t ldb = (int64_t)p.taps * p.Kc;
    const T* wl0 = Wp + (int64_t)(n0 + l31) * ldb + (2 * w + h) * 8;
    const T* wl1 = wl0 + 32 * ldb;
    if (FRAG) {   // [n-block][item][k quarter][lane][8]: n-block stride = items * 4 KB
      const int64_t nbs = (int64_t)nslab * p.taps * 2048;
      wl0 = reinterpret_cast<const T*>(p.Wfrag) + (int64_t)(n0 >> 5) * nbs + w * 512 + lane * 8;
      wl1 = wl0 + nbs;
    }
    int prow[MB];
    if (conv) {
      const int hw = p.H * p.W, W2 = p.W + 2;
      const int b = m0 / hw, y0 = (m0 - b * hw) / p.W;
      const T* plane = reinterpret_cast<const T*>(p.A0) + ((int64_t)(b * (p.H + 2) + y0) * W2) * p.Kc;
      const int PR = (p.st_rb + 2) * W2;
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        int px = 32 * i + l31;
        if (px > rows - 1) px = rows - 1;
        const int y = px / p.W, x = px - y * p.W;
        prow[i] = y * W2 + x;
      }
      stream_phase<T, MB, NB, 9, DBG, FRAG>(acc, smem, p.st_buf, plane, p.Kc, p.Kc, plane, p.Kc, PR, prow, W2, wl0, wl1, p.Kc, s0, s1, w, tid, ts);
    } else {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int px = 32 * i + l31;
        prow[i] = px > rows - 1 ? rows - 1 : px;
      }
      const T* a0 = reinterpret_cast<const T*>(p.A0) + (int64_t)m0 * p.lda0;
      const T* a1 = p.A1 ? reinterpret_cast<const T*>(p.A1) + (int64_t)m0 * p.lda1 : a0;
      stream_phase<T, MB, NB, 1, DBG, FRAG>(acc, smem, p.st_buf, a0, p.lda0, p.K0, a1, p.lda1, rows, prow, 0, wl0, wl1, 0, s0, s1, w, tid);
    }
  }
  // ---- fused 1x1 skip_connection: a second K phase over the unpadded rows of [S0 | S1] --------------------------------
  if (p.S0 != nullptr) {
    const int SK = p.SK0 + p.SK1, nslab = SK >> 6;
    const int per = (nslab + p.splitk - 1) / p.splitk;
    const int s0 = bz * per, s1 = s0 + per < nslab ? s0 + per : nslab;
    const T* Ws = reinterpret_cast<const T*>(p.Ws);
    const T* wl0 = Ws + (int64_t)(n0 + l31) * SK + (2 * w + h) * 8;
    const T* wl1 = wl0 + 32 * (int64_t)SK;
    if (FRAG) {
      const int64_t nbs = (int64_t)nslab * 2048;
      wl0 = reinterpret_cast<const T*>(p.Wsfrag) + (int64_t)(n0 >> 5) * nbs + w * 512 + lane * 8;
      wl1 = wl0 + nbs;
    }
    int prow[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int px = 32 * i + l31;
      prow[i] = px > rows - 1 ? rows - 1 : px;
    }
    const T* a0 = reinterpret_cast<const T*>(p.S0) + (int64_t)m0 * p.SK0;
    const T* a1 = p.S1 ? reinterpret_cast<const T*>(p.S1) + (int64_t)m0 * p.SK1 : a0;
    stream_phase<T, MB, NB, 1, DBG, FRAG>(acc, smem, p.st_buf, a0, p.SK0, p.SK0, a1, p.SK1, rows, prow, 0, wl0, wl1, 0, s0, s1, w, tid);
  }

  // ---- fold the four K quarters (fixed order: wave 0 + 1 + 2 + 3) and write the workgroup's fp32 partial tile ------------
  // accumulator registers 4q .. 4q+3 of a 32x32 block = channels 8q + 4h + {0..3} of pixel l31: wave q keeps those, the
  // other three park theirs in the LDS (one float4 per lane: conflict-free), one n-block per pass.
  if (DBG & 8) ts[12] = __builtin_readcyclecounter();
  f32x4_t* scr = reinterpret_cast<f32x4_t*>(smem);
  float* part = p.partial + (int64_t)bz * p.M * p.N;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    f32x4_t own[MB];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q != w) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
          if (i < mb_used) {
            const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            scr[((w * 4 + q) * MB + i) * 64 + lane] = v;
          }
      } else {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          own[i] = v;
        }
      }
    }
    __syncthreads();
    const int n = n0 + 32 * j + 8 * w + 4 * h;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      if (i < mb_used) {
        f32x4_t t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int src = 0; src < 4; ++src) {
          f32x4_t v;
          if (src == w) v = own[i];
          else v = scr[((src * 4 + w) * MB + i) * 64 + lane];
          if (src == 0) t = v;
          else t += v;
        }
        const int px = 32 * i + l31;
        if (px < rows && n < p.N) *reinterpret_cast<f32x4_t*>(part + (int64_t)(m0 + px) * p.N + n) = t;
      }
    }
    if (j + 1 < NB) __syncthreads();
  }
  if ((DBG & 8) && p.st_trace != nullptr) {
    __builtin_amdgcn_s_waitcnt(0);   // partial stores issued and (vmcnt) acknowledged
    ts[13] = __builtin_readcyclecounter();
    if (tid == 0) {
      unsigned long long* o = p.st_trace + (int64_t)blockIdx.x * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = ts[i];
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[14] = xcc; o[15] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

int stream_max_rows(int mb) { return mb * 32; }

}  // namespace

// Geometry of a launch: rows per m-tile (conv: a band of st_rb image rows), m-tiles, bytes of one LDS A buffer.
struct StreamGeom { int tm, rb, mtiles, buf, smem, nt; };

static bool stream_geom(const IgemmParams& p, int mb, StreamGeom* g) {
  const int cap = stream_max_rows(mb);
  const int np_conv = mb <= 5 ? 8 : 12, gslabs = mb <= 5 ? 2 : 1;   // StreamCfg::NP / ::G
  if (p.taps == 9) {
    if (p.H <= 0 || p.W <= 0 || p.M % (p.H * p.W) || p.W > cap) return false;
    int rb = 0;
    for (int r = 1; r <= p.H; ++r) if (p.H % r == 0 && r * p.W <= cap && (r + 2) * (p.W + 2) <= np_conv * 32) rb = r;
    if (rb == 0) return false;
    g->rb = rb; g->tm = rb * p.W; g->mtiles = p.M / g->tm;
    g->buf = np_conv * 32 * 128;                       // one slab per stage
  } else {
    g->rb = 0; g->mtiles = (p.M + cap - 1) / cap; g->tm = (p.M + g->mtiles - 1) / g->mtiles;
    g->buf = mb * 32 * 128 * gslabs;
  }
  if (p.S0 != nullptr && mb * 32 * 128 * gslabs > g->buf) g->buf = mb * 32 * 128 * gslabs;
  g->nt = mb <= 5 ? 64 : 32;                          // 9 m-blocks leave registers for one n-block per wave
  const int scratch = mb * 16 * 1024;
  g->smem = 2 * g->buf > scratch ? 2 * g->buf : scratch;
  return g->smem <= 160 * 1024;
}

bool stream_supported(const IgemmParams& p, int dtype, int mb) {
  if ((dtype != K22_BF16 && dtype != K22_F16) || (mb != 5 && mb != 9)) return false;
  if (p.taps != 1 && p.taps != 9) return false;
  if (p.Kc % 64 || p.K0 % 64 || p.N % 4 || p.Npad % 64 || p.Npad < p.N) return false;
  if (p.taps == 1 && (p.lda0 % 8 || (p.K0 < p.Kc && p.lda1 % 8))) return false;
  if (p.out_mode == IG_OUT_QKV && (p.ldo % 4 || p.N % 192)) return false;
  if ((p.ldo & 3) || (p.ldr & 3)) return false;
  if (p.S0 != nullptr && (p.taps != 9 || !p.Ws || p.SK0 % 64 || p.SK1 % 64 || p.SK0 <= 0 || (p.SK1 > 0 && !p.S1))) return false;
  if ((int64_t)p.Npad * p.taps * p.Kc >= (1ll << 31)) return false;
  StreamGeom g;
  return stream_geom(p, mb, &g);
}

// workgroups of a launch at split-K 1: m-tiles x n-tiles
int stream_mtiles(const IgemmParams& p, int mb) {
  StreamGeom g;
  return stream_geom(p, mb, &g) ? g.mtiles * (p.Npad / g.nt) : 0;
}

static std::atomic<long> g_stream_launches{0};
long stream_launch_count() { return g_stream_launches.load(); }

template <typename T, int MB, int NB, int DBG, bool FRAG>
static int launch_stream_cfg(const IgemmParams& p, const StreamGeom& g, int splitk, hipStream_t stream) {
  static LdsAttrGuard attr_guard;
  if (int rc_ = k22_ensure_lds_attr(attr_guard, reinterpret_cast<const void*>(&stream_kernel<T, MB, NB, DBG, FRAG>), 160 * 1024, __FILE__, __LINE__)) return rc_;
  IgemmParams q = p;
  q.splitk = splitk;
  q.st_tm = g.tm; q.st_rb = g.rb; q.st_mtiles = g.mtiles; q.st_buf = g.buf;
  const int nblocks = g.mtiles * (p.Npad / g.nt) * splitk;
  hipLaunchKernelGGL((stream_kernel<T, MB, NB, DBG, FRAG>), dim3(nblocks), dim3(256), g.smem, stream, q);
  K22_CHECK_LAUNCH();
  g_stream_launches.fetch_add(1, std::memory_order_relaxed);
  return K22_OK;
}

template <typename T>
static int launch_stream_typed(const IgemmParams& p, const StreamGeom& g, int mb, int splitk, hipStream_t stream) {
  const bool frag = p.Wfrag != nullptr;
  if (frag && p.S0 != nullptr && p.Wsfrag == nullptr) return k22_set_error(K22_EINVAL, "stream: fragment-major weights given without the skip weights' copy");
#ifdef K22_STREAM_DEBUG
  // measurement-only variants: K22_STREAM_DBG bits 1 = no MFMA, 2 = weight ring never refilled, 4 = one LDS read per phase (all three: wrong
  // results), 8 = cycle stamps of every workgroup into p.st_trace (results intact)
  if (const char* e = getenv("K22_STREAM_DBG")) {
    const int dbg = atoi(e);
    if (frag && mb == 5) {
      if (dbg == 1) return launch_stream_cfg<T, 5, 2, 1, true>(p, g, splitk, stream);
      if (dbg == 2) return launch_stream_cfg<T, 5, 2, 2, true>(p, g, splitk, stream);
      if (dbg == 4) return launch_stream_cfg<T, 5, 2, 4, true>(p, g, splitk, stream);
      if (dbg == 8) return launch_stream_cfg<T, 5, 2, 8, true>(p, g, splitk, stream);
      if (dbg == 9) return launch_stream_cfg<T, 5, 2, 9, true>(p, g, splitk, stream);
      if (dbg == 10) return launch_stream_cfg<T, 5, 2, 10, true>(p, g, splitk, stream);
      if (dbg == 12) return launch_stream_cfg<T, 5, 2, 12, true>(p, g, splitk, stream);
    }
    if (frag && mb == 9 && dbg == 8) return launch_stream_cfg<T, 9, 1, 8, true>(p, g, splitk, stream);
  }
#endif
  if (mb == 5) return frag ? launch_stream_cfg<T, 5, 2, 0, true>(p, g, splitk, stream) : launch_stream_cfg<T, 5, 2, 0, false>(p, g, splitk, stream);
  return frag ? launch_stream_cfg<T, 9, 1, 0, true>(p, g, splitk, stream) : launch_stream_cfg<T, 9, 1, 0, false>(p, g, splitk, stream);
}

// Launches stream_kernel only: it always leaves fp32 partial tiles [splitk][M][N] in p.partial; the finish (bias, residual,
// activation, rounding, GroupNorm partial sums) is the caller's (launch_igemm -> launch_reduce), also for splitk == 1.
int launch_stream(const IgemmParams& p, int dtype, int mb, int splitk, hipStream_t stream) {
  if (!stream_supported(p, dtype, mb)) return k22_set_error(K22_EINVAL, "stream: unsupported problem");
  if (p.partial == nullptr) return k22_set_error(K22_EINVAL, "stream: needs the fp32 partial buffer [splitk][M][N]");
  StreamGeom g;
  (void)stream_geom(p, mb, &g);
  if (dtype == K22_BF16) return launch_stream_typed<bf16_t>(p, g, mb, splitk, stream);
  if (dtype == K22_F16) return launch_stream_typed<f16_t>(p, g, mb, splitk, stream);
  return k22_set_error(K22_EINVAL, "stream: bad dtype");
}

// ---- one-time repack of a row-major weight matrix [Npad][taps * Kc] into fragment-major order ----------------------------------------
//   out[(((nb * n_items + it) * 4 + w) * 64 + lane) * 8 + e] = W[nb * 32 + (lane & 31)][tap * Kc + slab * 64 + 16 w + 8 (lane >> 5) + e],
//   it = slab * taps + tap: exactly the 16 bytes lane `lane` of wave (k quarter) w feeds to the MFMA for n-block nb at item it, so one
//   B fragment is 1 KB contiguous and the four waves of a workgroup read 4 KB contiguous per item.  16-bit types (the data is moved as is).
__global__ __launch_bounds__(256) void stream_repack_kernel(const uint4* __restrict__ W, uint4* __restrict__ out, int Npad, int taps, int Kc) {
  const int nslab = Kc >> 6;
  const int64_t n_items = (int64_t)nslab * taps;
  const int64_t total = (int64_t)(Npad >> 5) * n_items * 256;        // 16-byte pieces
  const int64_t ldw = (int64_t)taps * Kc / 8;                       // row stride in 16-byte pieces
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), w = (int)((i >> 6) & 3);
    const int64_t ni = i >> 8;
    const int64_t nb = ni / n_items, it = ni - nb * n_items;
    const int slab = (int)(it / taps), tap = (int)(it - (int64_t)slab * taps);
    const int64_t row = nb * 32 + (lane & 31);
    const int64_t kpiece = ((int64_t)tap * Kc + slab * 64 + 16 * w + 8 * (lane >> 5)) >> 3;
    out[i] = W[row * ldw + kpiece];
  }
}

size_t stream_frag_bytes(int Npad, int taps, int Kc, int dtype) { return dtype == K22_F32 ? 0 : (size_t)Npad * taps * Kc * 2; }

int launch_stream_repack(const void* W, void* out, int Npad, int taps, int Kc, int dtype, hipStream_t stream) {
  if (dtype == K22_F32 || !W || !out || Npad % 64 || Kc % 64 || (taps != 1 && taps != 9)) return k22_set_error(K22_EINVAL, "stream_repack: bad arguments");
  const int64_t total = (int64_t)(Npad / 32) * (Kc / 64) * taps * 256;
  int nb = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(stream_repack_kernel, dim3(nb), dim3(256), 0, stream, reinterpret_cast<const uint4*>(W), reinterpret_cast<uint4*>(out), Npad, taps, Kc);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
