// k22 — MI355X (gfx950 / CDNA4) native Kandinsky-2 sampling engine: shared device helpers.
//
// Element types: the torso of the UNet runs in bf16 (product path; MFMA v_mfma_f32_32x32x16_bf16), in fp16 (the reference's own
// reduced-precision mode, use_fp16 / convert_to_fp16 - kandinsky2/model/unet.py:409,566-572; v_mfma_f32_32x32x16_f16: the same
// rate and the same bytes as bf16 with 3 more mantissa bits) or in fp32 (parity path; v_mfma_f32_32x32x2_f32, exact
// fp32 FMA chain).  All paths share one kernel structure: every LDS tile row is 128
// bytes = 8 chunks of 16 B, and an "atom" is a 32x32x16 matrix product whose A/B
// fragments are 8 consecutive K elements per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // bf16 storage type
typedef _Float16 f16_t;         // fp16 storage type (IEEE binary16)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// K22_F16X3 (round 4, UNet engine only) = the SPLIT-PRECISION arithmetic: fp32 tensors, every MFMA operand x carried as the fp16 pair
// hi = rne(x), lo = rne(x - hi) and every product as THREE v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi, one fp32 accumulator):
// ~23 significand bits per operand at 3/16 of the exact-fp32 MFMA cost.  See x3_t below.
// K22_F16X2 (round 5) = the ASYMMETRIC split: the same tensors and operand formats as K22_F16X3 (so one op of a plan can run either), but
// a conv / GEMM product is TWO MFMAs - w_hi.a_hi + w_lo.a_hi: the weights keep ~22 bits (their rounding is the SYSTEMATIC error of a
// 16-bit engine: the same delta at every pixel and step), the activation operand is taken at fp16 precision (its rounding is random
// per pixel and step).  See x2_t below and DESIGN.md (precision plan).
enum K22DType { K22_BF16 = 0, K22_F32 = 1, K22_F16 = 2, K22_F16X3 = 3, K22_F16X2 = 4 };
// the two split-precision arithmetics: fp32 tensors in HBM, x3-chunk MFMA operands
inline bool k22_is_split(int dtype) { return dtype == K22_F16X3 || dtype == K22_F16X2; }
// bytes per element / elements per 128-byte LDS row of a storage type code
inline int k22_esz(int dtype) { return (dtype == K22_F32 || k22_is_split(dtype)) ? 4 : 2; }
inline int k22_bk(int dtype) { return (dtype == K22_F32 || k22_is_split(dtype)) ? 32 : 64; }
inline bool k22_dtype_ok(int dtype) { return dtype == K22_BF16 || dtype == K22_F32 || dtype == K22_F16; }
// what the kernels that only move / normalise data are instantiated on for an engine of arithmetic type `dtype`
inline int k22_storage_dtype(int dtype) { return k22_is_split(dtype) ? (int)K22_F32 : dtype; }
enum K22Act { K22_ACT_NONE = 0, K22_ACT_SILU = 1, K22_ACT_GELU = 2 };

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even: one v_cvt_pk_bf16_f32 per PAIR on gfx950 (the integer
// add-and-shift emulation costs ~5 VALU per element and made the elementwise kernels ALU-bound)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu); }
// fp32 pair -> packed fp16 pair, round-to-nearest-even (v_cvt_f16_f32 x2 + pack; NOT v_cvt_pkrtz, which truncates)
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ void unpack2_f16(uint32_t w, float& lo, float& hi) {
  const f16x2_t v = __builtin_bit_cast(f16x2_t, w);
  lo = (float)v[0]; hi = (float)v[1];
}
__device__ __forceinline__ void unpack2_bf16(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
// packed pair of a 16-bit storage type <-> two floats
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack2_bf16(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack2_f16(lo, hi); }
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t w, float& lo, float& hi) { unpack2_bf16(w, lo, hi); }
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t w, float& lo, float& hi) { unpack2_f16(w, lo, hi); }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) { return f32_to_bf16(f); }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float f) { return (f16_t)f; }
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// SiLU from the two native transcendentals, x * rcp(1 + exp2(-x * log2 e)): a few ulp of fp32 in 6 instructions, where the IEEE form
// above (precise expf + a true division) compiles to ~55.  Used ONLY by the split-precision engine's GroupNorm store (gn_apply's
// out_x3, gn_rewrite16<x3_t>), whose parity gates were measured with it (C2 final latent 3.6e-6).  The bf16 / fp16 engines keep the
// IEEE form: round 4 tried silu_fast there and measured (a) no speed-up of the bf16 step (GroupNorm class 1.16 -> 1.18 ms: those
// launches are latency-, not VALU-bound) and (b) a WORSE fp16 / bf16 MoVQ decode against the reference goldens in 9 of 10 fp16
// cases (768 px fp16: uint8 max diff 3 -> 6, 11.5 % -> 12.6 % of the bytes differ; bf16: 24 -> 36 grey levels), reproduced as an
// A/B of two builds on one box (tools/gpu_silu_ab.sh, profiles/r04_silu_ab.txt); the IEEE form restores round 3's bits exactly.
#ifdef K22_SILU_IEEE_EVERYWHERE   // measurement-only build: tools/gpu_silu_ab.sh
__device__ __forceinline__ float silu_fast(float x) { return silu_f(x); }
#else
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
#endif
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// GELU with erfc from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute) on the native rcp / exp2: ~15 instructions where erff costs
// ~90 with divergent branches.  GELU's error is <= 0.75e-7 |x| - three orders below the fp16 / bf16 rounding of the value it is used for
// (the 16-bit epilogues of the skinny GEMM, skinny.hip); 1 + erf is formed without cancellation on the negative side.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float q = poly * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);   // erfc(z)
  return x >= 0.f ? x * fmaf(-0.5f, q, 1.0f) : x * (0.5f * q);
}
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == K22_ACT_SILU) return silu_f(x);
  if (act == K22_ACT_GELU) return gelu_f(x);
  return x;
}
// FAST: SiLU through silu_fast (see above); everything else as apply_act
template <bool FAST> __device__ __forceinline__ float apply_act_sel(float x, int act) {
  if constexpr (FAST) { if (act == K22_ACT_SILU) return silu_fast(x); }
  return apply_act(x, act);
}

// ---- split-precision operand format ("x3 chunk") ------------------------------------------------------------------------------
// An MFMA operand tensor of the K22_F16X3 arithmetic occupies 4 bytes per element like fp32, but every aligned group of EIGHT
// consecutive K elements (32 bytes) is stored as [hi0 .. hi7 | lo0 .. lo7] in fp16, hi = rne_f16(x), lo = rne_f16(x - hi)
// (x - hi is exact in fp32).  fp16 subnormals are kept (the kernel mode never flushes them), so |x| >= 2^-2 is carried to 2^-23
// relative and anything smaller to 2^-25 absolute.  A fragment of a 32x32x16 atom (8 consecutive K of one row) is ONE group: its hi
// halves are one 16-byte piece, its lo halves the next - two conflict-free ds_read_b128 straight into the registers the MFMAs read,
// and the activation fragment of the asymmetric split (hi halves only) is ONE.  (Rounds 4-5 used groups of four, [hi x4 | lo x4]: a
// fragment then had to be gathered out of two pieces with register moves - 1.5 v_mov per MFMA in the x2 / x3 tap loops,
// profiles/r05_isa_mix.txt - and the x2 activation reads were 8-byte halves with a 3.4x LDS bank-conflict ratio.)
// WEIGHTS are packed in this format once (pack.py), pre-multiplied by the exact power of two K22_X3_WSCALE so that their lo halves
// stay normal; the epilogues multiply the accumulators by 1 / K22_X3_WSCALE.  ACTIVATIONS are written in it by their producers
// (GroupNorm-apply, the attention epilogue: x3_store4 below) or converted from fp32 rows at fragment-read time where the producer is
// not one of ours (`a_raw` operands: 1x1 skip connections, the conditioning GEMMs).  Tensors are 32-byte aligned with K % 8 == 0.
struct x3_t { float f; };   // storage tag: sizeof == 4; as a STORED tensor type it behaves as float (epilogue outputs are fp32)
constexpr float K22_X3_WSCALE = 256.0f;
constexpr float K22_X3_WSCALE_INV = 1.0f / 256.0f;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
// four consecutive K elements -> their (hi, lo) fp16 halves: .x .y = hi0..hi3, .z .w = lo0..lo3
__device__ __forceinline__ u32x4_t x3_split4(float x0, float x1, float x2, float x3) {
  const uint32_t h01 = pack2_f16(x0, x1), h23 = pack2_f16(x2, x3);
  const f16x2_t a = __builtin_bit_cast(f16x2_t, h01), b = __builtin_bit_cast(f16x2_t, h23);
  return u32x4_t{h01, h23, pack2_f16(x0 - (float)a[0], x1 - (float)a[1]), pack2_f16(x2 - (float)b[0], x3 - (float)b[1])};
}
__device__ __forceinline__ u32x4_t x3_split4(const float4 v) { return x3_split4(v.x, v.y, v.z, v.w); }
// stores four consecutive K elements (element index % 4 == 0) of an x3-chunk tensor: `at` = the address the fp32 values would have
// (generic or LDS pointer).  Their group of eight starts at the enclosing 32-byte boundary; elements 0-3 / 4-7 of the group own bytes
// 0-7 / 8-15 of its hi piece and of its lo piece.
__device__ __forceinline__ void x3_store4(void* at, const u32x4_t s) {
  char* g = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(at) & ~(uintptr_t)31);
  const int half = (int)((reinterpret_cast<uintptr_t>(at) >> 4) & 1);
  *reinterpret_cast<u32x2_t*>(g + 8 * half) = u32x2_t{s.x, s.y};
  *reinterpret_cast<u32x2_t*>(g + 16 + 8 * half) = u32x2_t{s.z, s.w};
}
__device__ __forceinline__ float to_f32(x3_t v) { return v.f; }
template <> __device__ __forceinline__ x3_t from_f32<x3_t>(float f) { return x3_t{f}; }
// K22_F16X2: the same storage and operand formats (is_x3 is true for it: everything that concerns the FORMAT is shared), two MFMAs per
// product - the activation fragment carries only its hi halves (FragA below: two 8-byte LDS reads, 4 registers).
struct x2_t { float f; };
__device__ __forceinline__ float to_f32(x2_t v) { return v.f; }
template <> __device__ __forceinline__ x2_t from_f32<x2_t>(float f) { return x2_t{f}; }
template <typename T> struct is_x3 { static constexpr bool value = false; };
template <> struct is_x3<x3_t> { static constexpr bool value = true; };
template <> struct is_x3<x2_t> { static constexpr bool value = true; };
// arithmetic of the fused 1x1 skip connection inside a conv kernel of type T (the full split for the asymmetric one: see halo_tail)
template <typename T> struct SkipT { using type = T; };
template <> struct SkipT<x2_t> { using type = x3_t; };
template <typename T> struct is_x2 { static constexpr bool value = false; };
template <> struct is_x2<x2_t> { static constexpr bool value = true; };

// ---- per-type tile traits ------------------------------------------------------------
template <typename T> struct TT;
template <> struct TT<bf16_t> {
  static constexpr int BK = 64;      // K elements per 128-byte LDS row
  static constexpr int EPC = 8;      // elements per 16-byte chunk
  static constexpr int KSTEPS = 4;   // 32x32x16 atoms per LDS row
};
template <> struct TT<f16_t> : TT<bf16_t> {};
template <> struct TT<float> {
  static constexpr int BK = 32;
  static constexpr int EPC = 4;
  static constexpr int KSTEPS = 2;
};
template <> struct TT<x3_t> : TT<float> {};
template <> struct TT<x2_t> : TT<float> {};
// LDS reads per fragment / MFMA instructions per 32x32x16 atom (the consumers' interleave of conv3_halo_spec_kernel)
template <typename T> struct FragCost { static constexpr int READS = sizeof(T) == 2 ? 1 : 2, MFMAS = sizeof(T) == 2 ? 1 : 8; };
template <> struct FragCost<x3_t> { static constexpr int READS = 2, MFMAS = 3; };
template <> struct FragCost<x2_t> { static constexpr int READS = 2, MFMAS = 2; };   // (reads per A + B fragment pair average 1.5: priced at the weight fragment's two)

// A/B fragment of one 32x32x16 atom: 8 consecutive K elements of one row.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { u32x4_t v; };
template <> struct Frag<f16_t> { u32x4_t v; };
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<x3_t> { u32x4_t hi, lo; };
template <> struct Frag<x2_t> { u32x4_t hi, lo; };   // the WEIGHT fragment of the asymmetric split
struct FragHi { u32x4_t hi; };                       // its ACTIVATION fragment: hi halves only
// FragA<T>: the fragment type of the activation operand of a conv / GEMM atom (Frag<T> itself except for x2_t)
template <typename T> struct FragAT { using type = Frag<T>; };
template <> struct FragAT<x2_t> { using type = FragHi; };
template <typename T> using FragA = typename FragAT<T>::type;
// the two 16-byte pieces of a group (hi x8, lo x8) ARE the fragment
__device__ __forceinline__ void x3_frag_from_chunks(Frag<x3_t>& f, const u32x4_t c0, const u32x4_t c1) {
  f.hi = c0;
  f.lo = c1;
}
// eight fp32 values -> fragment (the `a_raw` operands: 20 VALU per fragment)
__device__ __forceinline__ void x3_frag_from_f32(Frag<x3_t>& f, const float4 a, const float4 b) {
  const u32x4_t s0 = x3_split4(a), s1 = x3_split4(b);
  f.hi = u32x4_t{s0.x, s0.y, s1.x, s1.y};
  f.lo = u32x4_t{s0.z, s0.w, s1.z, s1.w};
}
__device__ __forceinline__ void x3_frag_from_chunks(Frag<x2_t>& f, const u32x4_t c0, const u32x4_t c1) {
  f.hi = c0;
  f.lo = c1;
}
// activation fragment of the asymmetric split from plain fp32 values: four conversions
__device__ __forceinline__ void x3_frag_from_f32(FragHi& f, const float4 a, const float4 b) {
  f.hi = u32x4_t{pack2_f16(a.x, a.y), pack2_f16(a.z, a.w), pack2_f16(b.x, b.y), pack2_f16(b.z, b.w)};
}

// Swizzled LDS tile: row r (128 B), logical chunk c (16 B) lives at physical chunk c ^ ((r >> 1) & 7).
// ds_read_b128 is serviced in 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) over 64 banks
// (256 B = two 128-B rows): the 16-B slot of (r, chunk) is (r & 1) * 8 + chunk, so rows of equal parity in a
// group need distinct swizzles -> key on r >> 1 (keying on r & 7 leaves every group 2-way conflicted).
// With this image a ds_read_b128 of "row = lane&31, chunk = const" is bank-conflict free.
__device__ __forceinline__ int lds_chunk_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// lane-half h (= lane>>5) reads K elements [16*ks + 8*h, +8) of row r.
__device__ __forceinline__ void ld_frag(Frag<bf16_t>& f, const char* tile, int r, int ks, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 2 * ks + h));
}
__device__ __forceinline__ void ld_frag(Frag<f16_t>& f, const char* tile, int r, int ks, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 2 * ks + h));
}
__device__ __forceinline__ void ld_frag(Frag<float>& f, const char* tile, int r, int ks, int h) {
  const float4 a = *reinterpret_cast<const float4*>(tile + lds_chunk_off(r, 4 * ks + 2 * h));
  const float4 b = *reinterpret_cast<const float4*>(tile + lds_chunk_off(r, 4 * ks + 2 * h + 1));
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// split-precision operand already in x3 chunks (weights; activations written by our producers): same two chunks as the fp32 form
__device__ __forceinline__ void ld_frag(Frag<x3_t>& f, const char* tile, int r, int ks, int h) {
  x3_frag_from_chunks(f, *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 4 * ks + 2 * h)),
                      *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 4 * ks + 2 * h + 1)));
}
__device__ __forceinline__ void ld_frag(Frag<x2_t>& f, const char* tile, int r, int ks, int h) {
  x3_frag_from_chunks(f, *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 4 * ks + 2 * h)),
                      *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 4 * ks + 2 * h + 1)));
}
// asymmetric split, activation operand in x3 chunks: the hi piece of the group, one ds_read_b128
__device__ __forceinline__ void ld_frag(FragHi& f, const char* tile, int r, int ks, int h) {
  f.hi = *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 4 * ks + 2 * h));
}
// RAW = the LDS image holds plain fp32 rows (split arithmetics only): convert while reading.  Every other type: ld_frag.
template <bool RAW, typename T, typename F> __device__ __forceinline__ void ld_frag_a(F& f, const char* tile, int r, int ks, int h) {
  if constexpr (RAW && is_x3<T>::value) {
    x3_frag_from_f32(f, *reinterpret_cast<const float4*>(tile + lds_chunk_off(r, 4 * ks + 2 * h)),
                     *reinterpret_cast<const float4*>(tile + lds_chunk_off(r, 4 * ks + 2 * h + 1)));
  } else {
    ld_frag(f, tile, r, ks, h);
  }
}

// acc(32x32, C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) += A(32x16) * B(16x32)
// A fragment: row = lane&31; B fragment: col = lane&31; both hold K = 8*(lane>>5) + j.
__device__ __forceinline__ void mma_atom(f32x16_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v),
                                                acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_atom(f32x16_t& acc, const Frag<f16_t>& a, const Frag<f16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.v), __builtin_bit_cast(f16x8_t, b.v), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_atom(f32x16_t& acc, const Frag<float>& a, const Frag<float>& b) {
  // 8 exact-fp32 MFMAs (K=2 each): call j pairs element j of lane-half 0 with element j of lane-half 1.
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// split precision: hi.hi + hi.lo + lo.hi into ONE fp32 accumulator (the dropped lo.lo term is <= 2^-24 of the product)
__device__ __forceinline__ void mma_atom(f32x16_t& acc, const Frag<x3_t>& a, const Frag<x3_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.hi), __builtin_bit_cast(f16x8_t, b.hi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.hi), __builtin_bit_cast(f16x8_t, b.lo), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a.lo), __builtin_bit_cast(f16x8_t, b.hi), acc, 0, 0, 0);
}
// asymmetric split: first operand = the WEIGHT fragment (hi, lo), second = the activation's hi halves: w_hi.a_hi + w_lo.a_hi
__device__ __forceinline__ void mma_atom(f32x16_t& acc, const Frag<x2_t>& w, const FragHi& a) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, w.hi), __builtin_bit_cast(f16x8_t, a.hi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, w.lo), __builtin_bit_cast(f16x8_t, a.hi), acc, 0, 0, 0);
}
// factor the epilogues apply to an accumulator before anything else (undoes the weights' power-of-two pre-scale)
template <typename T> __device__ __forceinline__ float acc_unscale(float v) {
  if constexpr (is_x3<T>::value) return v * K22_X3_WSCALE_INV; else return v;
}

__device__ __forceinline__ int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Raw workgroup barrier (no implied vmcnt/lgkmcnt drain: LDS-DMA stays in flight across it) that the COMPILER may not
// move LDS accesses across: the builtin alone is not a memory barrier for the optimiser.
__device__ __forceinline__ void raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- 64-lane reductions --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- vector load/store of 8 (bf16) / 4 (f32) elements = 16 bytes -----------------------
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ float get(int i) const {
    const uint32_t w = (&raw.x)[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  __device__ __forceinline__ void set2(int pair, float lo, float hi) {
    (&raw.x)[pair] = pack2_bf16(lo, hi);
  }
};
template <> struct Vec16<f16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ float get(int i) const {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, (&raw.x)[i >> 1]);
    return (float)v[i & 1];
  }
  __device__ __forceinline__ void set2(int pair, float lo, float hi) { (&raw.x)[pair] = pack2_f16(lo, hi); }
};
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ float get(int i) const { return (&raw.x)[i]; }
  __device__ __forceinline__ void set2(int pair, float lo, float hi) {
    (&raw.x)[2 * pair] = lo;
    (&raw.x)[2 * pair + 1] = hi;
  }
};
template <> struct Vec16<x3_t> : Vec16<float> {};
template <> struct Vec16<x2_t> : Vec16<float> {};

#define K22_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return k22_set_error_hip(e__, __FILE__, __LINE__); \
  } while (0)

int k22_set_error(int code, const char* msg);
int k22_set_error_hip(hipError_t e, const char* file, int line);
