// k22 - shared by the LDS-resident-halo convolution kernels (conv3_halo.hip: lock-step kernels, gemm8; conv3_spec.hip: the
// producer / consumer specialised kernel): fragment loaders, the LDS-DMA helper, the fused GroupNorm-apply rewrite, the common
// epilogue halo_tail, and the LDS budget every launcher and kernel agrees on.
#pragma once
#include "kernels.h"
#include <stdlib.h>

namespace {

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void ld_frag_at(Frag<bf16_t>& f, const char* rowp, int sw, int ks, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(rowp + (((2 * ks + h) ^ sw) << 4));
}
__device__ __forceinline__ void ld_frag_at(Frag<f16_t>& f, const char* rowp, int sw, int ks, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(rowp + (((2 * ks + h) ^ sw) << 4));
}
__device__ __forceinline__ void ld_frag_at(Frag<float>& f, const char* rowp, int sw, int ks, int h) {
  const float4 a = *reinterpret_cast<const float4*>(rowp + (((4 * ks + 2 * h) ^ sw) << 4));
  const float4 b = *reinterpret_cast<const float4*>(rowp + (((4 * ks + 2 * h + 1) ^ sw) << 4));
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

// split precision: the operand is in x3 chunks (weights; activations written by gn_apply / the attention epilogue)
__device__ __forceinline__ void ld_frag_at(Frag<x3_t>& f, const char* rowp, int sw, int ks, int h) {
  x3_frag_from_chunks(f, *reinterpret_cast<const u32x4_t*>(rowp + (((4 * ks + 2 * h) ^ sw) << 4)),
                      *reinterpret_cast<const u32x4_t*>(rowp + (((4 * ks + 2 * h + 1) ^ sw) << 4)));
}
__device__ __forceinline__ void ld_frag_at(Frag<x2_t>& f, const char* rowp, int sw, int ks, int h) {
  x3_frag_from_chunks(f, *reinterpret_cast<const u32x4_t*>(rowp + (((4 * ks + 2 * h) ^ sw) << 4)),
                      *reinterpret_cast<const u32x4_t*>(rowp + (((4 * ks + 2 * h + 1) ^ sw) << 4)));
}
// asymmetric split, activation operand: the hi piece of the group (one conflict-free ds_read_b128)
__device__ __forceinline__ void ld_frag_at(FragHi& f, const char* rowp, int sw, int ks, int h) {
  f.hi = *reinterpret_cast<const u32x4_t*>(rowp + (((4 * ks + 2 * h) ^ sw) << 4));
}
// RAW operands (split arithmetics only): plain fp32 rows in the LDS, split while they are read; every other type = ld_frag_at
template <bool RAW, typename T, typename F> __device__ __forceinline__ void ld_frag_at_a(F& f, const char* rowp, int sw, int ks, int h) {
  if constexpr (RAW && is_x3<T>::value) {
    x3_frag_from_f32(f, *reinterpret_cast<const float4*>(rowp + (((4 * ks + 2 * h) ^ sw) << 4)),
                     *reinterpret_cast<const float4*>(rowp + (((4 * ks + 2 * h + 1) ^ sw) << 4)));
  } else {
    ld_frag_at(f, rowp, sw, ks, h);
  }
}

template <typename T> __device__ __forceinline__ void store8(T* dst, const float* v);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* dst, const float* v) {
  uint4 w;
  w.x = pack2_bf16(v[0], v[1]);
  w.y = pack2_bf16(v[2], v[3]);
  w.z = pack2_bf16(v[4], v[5]);
  w.w = pack2_bf16(v[6], v[7]);
  *reinterpret_cast<uint4*>(dst) = w;
}
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* dst, const float* v) {
  uint4 w;
  w.x = pack2_f16(v[0], v[1]);
  w.y = pack2_f16(v[2], v[3]);
  w.z = pack2_f16(v[4], v[5]);
  w.w = pack2_f16(v[6], v[7]);
  *reinterpret_cast<uint4*>(dst) = w;
}
template <> __device__ __forceinline__ void store8<float>(float* dst, const float* v) {
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<x3_t>(x3_t* dst, const float* v) { store8<float>(reinterpret_cast<float*>(dst), v); }
template <> __device__ __forceinline__ void store8<x2_t>(x2_t* dst, const float* v) { store8<float>(reinterpret_cast<float*>(dst), v); }
template <typename T> __device__ __forceinline__ void load8f(const T* src, float* v);
template <> __device__ __forceinline__ void load8f<bf16_t>(const bf16_t* src, float* v) {
  const uint4 r = *reinterpret_cast<const uint4*>(src);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void load8f<f16_t>(const f16_t* src, float* v) {
  const uint4 r = *reinterpret_cast<const uint4*>(src);
  unpack2_f16(r.x, v[0], v[1]); unpack2_f16(r.y, v[2], v[3]); unpack2_f16(r.z, v[4], v[5]); unpack2_f16(r.w, v[6], v[7]);
}
template <> __device__ __forceinline__ void load8f<float>(const float* src, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8f<x3_t>(const x3_t* src, float* v) { load8f<float>(reinterpret_cast<const float*>(src), v); }
template <> __device__ __forceinline__ void load8f<x2_t>(const x2_t* src, float* v) { load8f<float>(reinterpret_cast<const float*>(src), v); }
// value as it will be read back from memory (the GroupNorm statistics are those of the stored tensor)
template <typename T> __device__ __forceinline__ float stored(float v);
template <> __device__ __forceinline__ float stored<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <> __device__ __forceinline__ float stored<f16_t>(float v) { return (float)(f16_t)v; }
template <> __device__ __forceinline__ float stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored<x3_t>(float v) { return v; }
template <> __device__ __forceinline__ float stored<x2_t>(float v) { return v; }

__device__ __forceinline__ int xcd_remap_h(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// LDS-DMA issued from inline asm: 16 bytes per lane from `g` to LDS byte address `lds_dst` (wave-uniform) + 16*lane.
// hipcc books a __builtin_amdgcn_global_load_lds as a FLAT access pending on BOTH counters, and because the counted
// vmcnt waits of this file are invisible to it, every later wait for a ds_read becomes lgkmcnt(0) - also for fragments
// read a whole MFMA group ago, with younger reads still in flight.  An asm statement is absent from its bookkeeping:
// the compiler then emits exact lgkmcnt(N) for the fragment reads; completion of the DMA is counted by hand anyway.
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}


// ---- fused GroupNorm-apply (conv3_halo_spec_kernel producers; IgemmParams::gn_coeff) -------------------------------------------
// 16 bytes of raw T values as the LDS-DMA left them (EPC consecutive channels of one pixel) -> act(x * A[c] + Bc[c]) in the operand
// format of T, zero at a border position.  cf = this lane's coefficients (A, Bc) x EPC as gn_coeff_kernel wrote them.  Same
// expression and rounding as gn_apply_kernel (elementwise.hip): the fused and the stand-alone path give the same bits.
__device__ __forceinline__ u32x4_t gn_rewrite16(bf16_t, u32x4_t raw, const float* cf, int act, bool border) {
  float v[8];
  unpack2_bf16(raw.x, v[0], v[1]); unpack2_bf16(raw.y, v[2], v[3]); unpack2_bf16(raw.z, v[4], v[5]); unpack2_bf16(raw.w, v[6], v[7]);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k] * cf[2 * k] + cf[2 * k + 1], act);
  const u32x4_t o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
  return border ? u32x4_t{0u, 0u, 0u, 0u} : o;
}
__device__ __forceinline__ u32x4_t gn_rewrite16(f16_t, u32x4_t raw, const float* cf, int act, bool border) {
  float v[8];
  unpack2_f16(raw.x, v[0], v[1]); unpack2_f16(raw.y, v[2], v[3]); unpack2_f16(raw.z, v[4], v[5]); unpack2_f16(raw.w, v[6], v[7]);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k] * cf[2 * k] + cf[2 * k + 1], act);
  const u32x4_t o = {pack2_f16(v[0], v[1]), pack2_f16(v[2], v[3]), pack2_f16(v[4], v[5]), pack2_f16(v[6], v[7])};
  return border ? u32x4_t{0u, 0u, 0u, 0u} : o;
}
__device__ __forceinline__ u32x4_t gn_rewrite16(float, u32x4_t raw, const float* cf, int act, bool border) {
  float v[4] = {__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)};
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = apply_act_sel<false>(v[k] * cf[2 * k] + cf[2 * k + 1], act);
  const u32x4_t o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  return border ? u32x4_t{0u, 0u, 0u, 0u} : o;
}
// split precision: fp32 in, x3 chunk out (what gn_apply_kernel's out_x3 store writes)
__device__ __forceinline__ u32x4_t gn_rewrite16(x3_t, u32x4_t raw, const float* cf, int act, bool border) {
  float v[4] = {__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)};
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = apply_act_sel<true>(v[k] * cf[2 * k] + cf[2 * k + 1], act);
  const u32x4_t o = x3_split4(v[0], v[1], v[2], v[3]);
  return border ? u32x4_t{0u, 0u, 0u, 0u} : o;
}
__device__ __forceinline__ u32x4_t gn_rewrite16(x2_t, u32x4_t raw, const float* cf, int act, bool border) { return gn_rewrite16(x3_t{}, raw, cf, act, border); }
// plain 16-byte global load from inline asm: one more entry of the producers' hand-counted VMEM queue (a compiler-issued load would
// make hipcc wait vmcnt(0) - draining every LDS-DMA in flight - before its first use); the destination is valid only behind a
// gn_wait_* statement that names it
__device__ __forceinline__ void gload16_asm(u32x4_t& dst, const void* g) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(g) : "memory");
}
template <int N> __device__ __forceinline__ void gn_wait(u32x4_t (&c)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void gn_wait(u32x4_t (&c)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(c[0]), "+v"(c[1]) : "n"(N) : "memory");
}

// the same for a WEIGHT piece.  K22_W_NT (measurement build): non-temporal hint - each weight byte is read by the few workgroups of one
// n-tile column, once (MI355X_MICROARCH.md "nt-weights": issued -> landed -18 % for read-once streams, -6 % end to end when every CU re-reads)
__device__ __forceinline__ void glds16w_asm(const void* g, unsigned lds_dst) {
#ifdef K22_W_NT
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
#else
  glds16_asm(g, lds_dst);
#endif
}

constexpr int HALO_BN = 128;
constexpr int HALO_NW = 8;        // waves per workgroup
constexpr int HALO_MAXA = 8;      // halo LDS-DMA slots per wave per slab: taps 0 .. 9-NBST carry one each

// Each iteration issues [weight tile NBST-1 taps ahead (2 loads)] THEN [one halo piece]: the halo piece is the
// youngest entry of the VMEM queue, so the counted wait for the weight tile can leave it in flight — it gets two
// taps (~1.6 us at 96x96) to come back from the Infinity Cache / HBM instead of one.  This is the number of halo
// pieces issued in the NBST-1 iterations before tap T, i.e. younger than the weight tile tap T needs; taps are
// unrolled, so it is a compile-time constant.
template <int NBST> constexpr int halo_count_a(int t) {
  int c = 0;
  for (int k = 1; k <= NBST - 1; ++k) {
    const int u = t - k;
    if (u >= 0 && u <= 9 - NBST) ++c;
  }
  return c;
}

}  // namespace

// Everything after the 3x3 K loop, shared by the halo kernels: optional fused 1x1 skip connection (a plain-GEMM K loop
// on 128-byte rows), then the epilogue through LDS (bias, residual, one rounding, 16-byte stores, GroupNorm partials).
// PLAIN = false: rows are positions v of the padded plane (3x3 convolution); PLAIN = true: rows are the pixels
// v < H*W of image `img` themselves (gemm8_kernel), and the qkv-projection output mode is available.
// SPEC = true (conv3_halo_spec_kernel): only waves 0-3 hold accumulators, 2 x 2 over the tile with (BM/2) x 64 each; waves 4-7
// were the producers of the main loop.  All eight waves issue the skip loop's LDS-DMA and run the store / statistics epilogue.
template <typename T, int BM, bool PLAIN = false, bool SPEC = false>
__device__ __forceinline__ void halo_tail(const IgemmParams& p, f32x16_t (&acc)[SPEC ? BM / 64 : BM / 128][2], char* smem, int bx, int bz, int img, int v0, int n0) {
  using TR = TT<T>;
  constexpr int BK = TR::BK, EPC = TR::EPC, KSTEPS = TR::KSTEPS;
  constexpr int BN = HALO_BN, NW = HALO_NW, WM = SPEC ? 2 : 4, WN = 2;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int B_SLOTS = BN / 8 / NW;
  constexpr int B_BYTES = BN * 128;
  constexpr int TS = BN * 4 + 16;       // epilogue tile row stride (bytes): conflict-free float4 writes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool cw = !SPEC || wave < 4;      // this wave holds accumulators (wave-uniform)
  const int h = lane >> 5, l31 = lane & 31;
  const int W2 = PLAIN ? 1 : p.W + 2;
  const int VR = PLAIN ? (p.H > 0 ? p.H * p.W : p.M) : p.H * W2;
  const int abase = wm * (BM / WM) + l31;
  int brow[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) brow[ni] = (wn * (BN / WN) + ni * 32 + l31) * 128;
  const int bsw = (l31 >> 1) & 7;

  // ---- fused 1x1 skip connection: acc += X[tile pixels][SK] . Ws[n][SK]^T, 3- / 4-stage LDS-DMA ring ----------
  if (p.S0 != nullptr) {
    const int SK = p.SK0 + p.SK1;
    const int nss = SK / BK;
    const int sp = p.splitk > 1 ? p.splitk : 1;
    const int q0 = nss * bz / sp, q1 = nss * (bz + 1) / sp;
    if (q0 < q1) {
      constexpr int SA_SLOTS = BM / 8 / NW;             // input-tile LDS-DMA instructions per wave per slab
      constexpr int SBUF = BM * 128 + B_BYTES;
      wait_vmcnt<0>();
      __syncthreads();                                   // main-loop buffers are free
      int spix[SA_SLOTS], schunk[SA_SLOTS];
#pragma unroll
      for (int i = 0; i < SA_SLOTS; ++i) {
        const int row = 8 * (wave + NW * i) + (lane >> 3);
        int v = v0 + row;
        if (v > VR - 1) v = VR - 1;
        const int y = v / W2;
        int x = v - y * W2;
        if (x > p.W - 1) x = p.W - 1;                    // junk columns read a valid pixel; their rows are dropped
        spix[i] = (img * p.H + y) * p.W + x;
        schunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
      }
      int wsoff[B_SLOTS];
#pragma unroll
      for (int i = 0; i < B_SLOTS; ++i) {
        const int row = 8 * (wave + NW * i) + (lane >> 3);
        int n = n0 + row;
        if (n > p.Npad - 1) n = p.Npad - 1;
        wsoff[i] = n * SK + ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
      }
      const T* __restrict__ X0 = reinterpret_cast<const T*>(p.S0);
      const T* __restrict__ X1 = reinterpret_cast<const T*>(p.S1);
      const T* __restrict__ Ws = reinterpret_cast<const T*>(p.Ws);
      // NSK-deep LDS-DMA ring with a COUNTED vmcnt (round 5).  The 2-stage form this replaces waited vmcnt(0) at every slab: one DMA round
      // trip (1-2 us from L2 / HBM) per 0.25-0.5 us of MFMA work, i.e. +13...25 us on every convolution with a fused skip - the reason the
      // tile table gave seven of the 96x96 convolutions to the lock-step kernel.  Now NSK - 1 slabs are in flight across the one raw
      // barrier per slab; the DMA is issued from inline asm (glds16_asm) so that hipcc neither drains it nor waits lgkmcnt(0) for it.
      // Slot reuse: iteration q refills the stage iteration q - 1 read, after the barrier every wave reaches only behind its
      // lgkmcnt(0); past-the-end issues re-load the last slab (uniform counting), drained by the vmcnt(0) below the loop.
      constexpr int NSK = BM == 256 ? 3 : 4;               // 3 x 48 KB / 4 x 32 KB of the 160 KB
      constexpr int SCH = SA_SLOTS + B_SLOTS;              // LDS-DMA instructions per wave per slab
      const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
#define K22_ISSUE_SKIP(Q, BUFI)                                                                            \
      {                                                                                                    \
        int q_ = (Q);                                                                                      \
        if (q_ > q1 - 1) q_ = q1 - 1;                                                                      \
        const int k0_ = q_ * BK;                                                                           \
        const bool second_ = k0_ >= p.SK0;                                                                 \
        const T* xs_ = second_ ? X1 : X0;                                                                  \
        const int ldx_ = second_ ? p.SK1 : p.SK0;                                                          \
        const int kk_ = second_ ? k0_ - p.SK0 : k0_;                                                       \
        const unsigned dA_ = lds0 + (BUFI) * SBUF + wave * 1024;                                           \
        _Pragma("unroll") for (int i = 0; i < SA_SLOTS; ++i)                                               \
            glds16_asm(xs_ + (int64_t)spix[i] * ldx_ + kk_ + schunk[i], __builtin_amdgcn_readfirstlane(dA_ + i * NW * 1024)); \
        _Pragma("unroll") for (int i = 0; i < B_SLOTS; ++i)                                                \
            glds16w_asm(Ws + wsoff[i] + k0_, __builtin_amdgcn_readfirstlane(dA_ + BM * 128 + i * NW * 1024)); \
      }
#pragma unroll
      for (int t = 0; t < NSK - 1; ++t) K22_ISSUE_SKIP(q0 + t, t);
      int buf = 0, fill = NSK - 1;
      for (int q = q0; q < q1; ++q) {
        wait_vmcnt<(NSK - 2) * SCH>();
        raw_barrier();
        K22_ISSUE_SKIP(q + NSK - 1, fill);
        const char* sA = smem + buf * SBUF;
        const char* sB = sA + BM * 128;
        if (cw) {
        // (asymmetric split: the skip input is the un-normalised residual stream - its operand rounding alone costs as much as that of
        // every GroupNorm output together, tests/golden/drift_ablation_x2.json - so this K loop keeps all three MFMAs)
        using TSK = typename SkipT<T>::type;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          Frag<TSK> a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ld_frag_at_a<true, TSK>(a[mi], sA + (abase + mi * 32) * 128, bsw, ks, h);   // S0 / S1: plain T rows
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) ld_frag_at(b[ni], sB + brow[ni], bsw, ks, h);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_atom(acc[mi][ni], b[ni], a[mi]);
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // no fragment read of this stage in flight when it is refilled after the next barrier
        buf = (buf + 1 == NSK) ? 0 : buf + 1;
        fill = (fill + 1 == NSK) ? 0 : fill + 1;
      }
#undef K22_ISSUE_SKIP
    }
  }
  wait_vmcnt<0>();
  __syncthreads();  // every wave is done with the operand buffers: the LDS becomes the fp32 output tile

  // ---- epilogue 1: accumulators -> LDS tile [BM][BN] fp32 (lane = pixel, 4 consecutive channels per quad) ----
  if (cw)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * (BM / WM) + mi * 32 + l31;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wn * (BN / WN) + ni * 32 + 8 * j + 4 * h;
        *reinterpret_cast<float4*>(smem + row * TS + col * 4) =
            make_float4(acc_unscale<T>(acc[mi][ni][4 * j]), acc_unscale<T>(acc[mi][ni][4 * j + 1]), acc_unscale<T>(acc[mi][ni][4 * j + 2]),
                        acc_unscale<T>(acc[mi][ni][4 * j + 3]));
      }
  }
  __syncthreads();

  // ---- qkv projection, n-tile inside v (an n-tile never straddles q / k / v): V^T_all wants the tile transposed.
  // lane = token (consecutive lanes -> consecutive addresses of one V^T row), 4 channels per LDS read
  if constexpr (PLAIN) {
    if (p.out_mode == IG_OUT_QKV && n0 >= 2 * (p.N / 3)) {
      const int C = p.N / 3, heads = C >> 6;
      constexpr int CPG = BN / (512 / BM);   // channels per thread group
      const int r = tid % BM, cg = tid / BM;
      const int v = v0 + r;
      if (v < VR) {
#pragma unroll 4
        for (int c4 = 0; c4 < CPG; c4 += 4) {
          const int col = cg * CPG + c4, n = n0 + col;
          if (n >= p.N) break;
          const float4 t = *reinterpret_cast<const float4*>(smem + r * TS + col * 4);
          float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.bias != nullptr) b = *reinterpret_cast<const float4*>(p.bias + n);
          const int c = n - 2 * C, head = c >> 6, d = c & 63;
          T* vt = reinterpret_cast<T*>(p.vtall) + ((int64_t)(img * heads + head) * 64 + d) * p.att_Tkp + p.att_S + v;
          vt[0] = from_f32<T>(t.x + b.x);
          vt[(int64_t)p.att_Tkp] = from_f32<T>(t.y + b.y);
          vt[(int64_t)2 * p.att_Tkp] = from_f32<T>(t.z + b.z);
          vt[(int64_t)3 * p.att_Tkp] = from_f32<T>(t.w + b.w);
        }
      }
      return;
    }
  }

  // ---- epilogue 2: thread = 8 channels x RPT consecutive rows -------------------------------------------
  constexpr int SEGS = BN / 8;            // 16 column segments
  constexpr int RGS = 512 / SEGS;         // 32 row groups
  constexpr int RPT = BM / RGS;           // rows per thread
  const int cs = tid % SEGS, rg = tid / SEGS;
  const int n = n0 + cs * 8;
  const bool n_ok = n < p.N;              // N % 8 == 0 (checked on the host)
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  const bool finish = (p.splitk <= 1);
  if (finish && n_ok && p.bias != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
    bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  if (finish && n_ok && p.bias2 != nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias2 + n), b1 = *reinterpret_cast<const float4*>(p.bias2 + n + 4);
    bias8[0] += b0.x; bias8[1] += b0.y; bias8[2] += b0.z; bias8[3] += b0.w;
    bias8[4] += b1.x; bias8[5] += b1.y; bias8[6] += b1.z; bias8[7] += b1.w;
  }
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  float* part = finish ? nullptr : p.partial + (int64_t)bz * p.M * p.N;
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int row = rg * RPT + k;
    const int v = v0 + row;
    int64_t m;
    if constexpr (PLAIN) {
      if (!n_ok || v >= VR) continue;
      m = (int64_t)img * VR + v;
    } else {
      const int y = v / W2, x = v - y * W2;
      if (!n_ok || v >= VR || x >= p.W) continue;
      m = ((int64_t)img * p.H + y) * p.W + x;
    }
    const float4 t0 = *reinterpret_cast<const float4*>(smem + row * TS + cs * 32);
    const float4 t1 = *reinterpret_cast<const float4*>(smem + row * TS + cs * 32 + 16);
    float val[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    if (!finish) {
      *reinterpret_cast<float4*>(part + m * p.N + n) = t0;
      *reinterpret_cast<float4*>(part + m * p.N + n + 4) = t1;
      continue;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] += bias8[e];
    if (res != nullptr) {
      float rv[8];
      load8f<T>(res + m * p.ldr + n, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] += rv[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] = apply_act(val[e], p.act);
    if (PLAIN && p.out_mode == IG_OUT_QKV) {
      // columns [q | k | v] x [heads][64]: q row-major, k behind the context keys of K_all, v transposed into V^T_all
      const int C = p.N / 3, heads = C >> 6;
      const int which = n / C, c = n - which * C;
      const int head = c >> 6, d = c & 63;
      if (which == 0) {
        store8<T>(reinterpret_cast<T*>(p.out) + m * p.ldo + c, val);
      } else if (which == 1) {
        store8<T>(reinterpret_cast<T*>(p.kall) + ((int64_t)(img * heads + head) * p.att_Tkp + p.att_S + v) * 64 + d, val);
      } else {
        T* vt = reinterpret_cast<T*>(p.vtall) + ((int64_t)(img * heads + head) * 64 + d) * p.att_Tkp + p.att_S + v;
#pragma unroll
        for (int e = 0; e < 8; ++e) vt[(int64_t)e * p.att_Tkp] = from_f32<T>(val[e]);
      }
      continue;
    }
    if (p.out_mode == IG_OUT_ROWMAJOR) store8<T>(reinterpret_cast<T*>(p.out) + m * p.ldo + n, val);
    else store8<float>(reinterpret_cast<float*>(p.out) + m * p.ldo + n, val);
    if (p.stats != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sv = (p.out_mode == IG_OUT_ROWMAJOR) ? stored<T>(val[e]) : val[e];
        ssum[e] += sv;
        ssq[e] += sv * sv;
      }
    }
  }
  if (!finish || p.stats == nullptr) return;

  // ---- epilogue 3: per-channel (sum, sumsq) of this tile's stored values, fixed-order reduction ---------
  __syncthreads();  // tile fully consumed
  float* red = reinterpret_cast<float*>(smem);  // [RGS][BN][2]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    *reinterpret_cast<float2*>(red + ((rg * BN) + cs * 8 + e) * 2) = make_float2(ssum[e], ssq[e]);
  }
  __syncthreads();
  if (tid < BN * 2) {
    const int ch = tid >> 1, which = tid & 1;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < RGS; ++r) a += red[((r * BN) + ch) * 2 + which];
    if (n0 + ch < p.N) p.stats[((int64_t)bx * p.N + n0 + ch) * 2 + which] = a;
  }
}

// ---- host side ---------------------------------------------------------------------------------
inline int halo_rows(const IgemmParams& p, int bm) { return (bm + 2 * (p.W + 2) + 2 + 7) & ~7; }

inline size_t halo_smem_bytes(const IgemmParams& p, int bm, int nbst) {
  // (+ 1 KB behind the weight ring for the fused GroupNorm-apply form: the landing place of its surplus DMA slots)
  const size_t main_loop = (size_t)2 * halo_rows(p, bm) * 128 + (size_t)nbst * HALO_BN * 128 + (p.gn_coeff ? 1024 : 0);
  const size_t epi = (size_t)bm * (HALO_BN * 4 + 16);
  const size_t red = (size_t)32 * HALO_BN * 2 * 4;
  const size_t skip = p.S0 ? (size_t)(bm == 256 ? 3 : 4) * (bm * 128 + HALO_BN * 128) : 0;   // halo_tail: NSK stages
  size_t m = main_loop > epi ? main_loop : epi;
  m = m > skip ? m : skip;
  return m > red ? m : red;
}

// deepest weight ring (2, 3, 4 or 6 tiles) that fits the LDS next to the double-buffered halo and still leaves
// enough halo slots (taps 0 .. 9-NBST, one 8-row piece per wave each); 0 = the problem does not fit at all.
// Depth matters at the low-resolution levels: their weights stream from HBM (~2 us away) while a tap is
// 0.2-0.4 us of MFMA work, so a workgroup must keep ~64+ KB of weight tiles in flight (Little's law).
inline int halo_pick_nbst(const IgemmParams& p, int bm) {
  const int np = halo_rows(p, bm) / 8;
  for (int nbst : {6, 4, 3, 2}) {
    if (np > (10 - nbst) * HALO_NW) continue;
    if (halo_smem_bytes(p, bm, nbst) > 160 * 1024) continue;
    return nbst;
  }
  return 0;
}

