// k22 — fused attention with prepended context keys (flash-style, online softmax in fp32).
//
// Restates QKVAttention.forward's einsum path (kandinsky2/model/unet.py:286-340): per (batch, head)
// softmax((q*s)(k*s)^T) v with s = ch^-0.25, keys = [encoder K | self K], fp32 softmax.  The
// [B*H, T, S] score tensor of the reference is never materialised.
//
// Work split: workgroup = 4 waves = 128 queries of one (b, head); wave = 32 queries.
//   S^T[key][q] = K[key][:] . Q[q][:]   (A = K tile rows from LDS, B = Q fragments in registers)
//     -> C layout puts ONE query per lane column (lane&31) and 32 keys of the 64-key tile in the
//        lane's 32 accumulator registers: row max / row sum are lane-local plus one xor-32 shuffle.
//   O^T[d][q] += V^T[d][key] . P^T[key][q]   (A = V^T tile rows from LDS, B = P straight from the
//        S^T accumulator registers: no LDS round trip, the rescale factor is lane-local).
// K_all / V^T_all are produced by kv_pack_kernel (elementwise.hip), zero padded to 64 keys.
#include "kernels.h"
#include "elementwise.h"
#include <type_traits>
#include <stdlib.h>

// V^T fragment whose K (= key) order matches the S^T accumulator registers:
// element j<4 -> key 16a + 4h + j ; j>=4 -> key 16a + 8 + 4h + (j-4).
// 16-bit tiles (round 6): the V^T tile is stored KEY-PERMUTED within every block of 16 keys - piece 2a + h' of a row holds
// [keys 16a + 4h' .. + 3 | keys 16a + 8 + 4h' .. + 3] (vt_store16 below) - so that the fragment is ONE conflict-free ds_read_b128.  With the
// natural key order it was two 8-byte reads gathered by six register moves per PV atom: 24 v_mov per 64-key tile, and the LDS array - not the
// matrix pipe and not VALU issue - was what the tile loop waited for (ds_read2st64_b64: 8+ LDS cycles per wave-instruction against 4;
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 1.64 on this kernel, profiles/r05_pmc_mfma_bf16.txt).
__device__ __forceinline__ void ld_frag_split(Frag<bf16_t>& f, const char* tile, int r, int a, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 2 * a + h));
}
__device__ __forceinline__ void ld_frag_split(Frag<f16_t>& f, const char* tile, int r, int a, int h) {
  f.v = *reinterpret_cast<const u32x4_t*>(tile + lds_chunk_off(r, 2 * a + h));
}
// stores keys 8 cc .. 8 cc + 7 (one 16-byte register, natural order) of V^T row `row` into the key-permuted 16-bit tile: keys +0..3 go to
// piece 2 (cc >> 1), keys +4..7 to piece 2 (cc >> 1) + 1, both at byte 8 (cc & 1)
__device__ __forceinline__ void vt_store16(char* tile, int row, int cc, const u32x4_t v) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
  *reinterpret_cast<u32x2_t*>(tile + lds_chunk_off(row, cc & ~1) + 8 * (cc & 1)) = u32x2_t{v.x, v.y};
  *reinterpret_cast<u32x2_t*>(tile + lds_chunk_off(row, cc | 1) + 8 * (cc & 1)) = u32x2_t{v.z, v.w};
}
__device__ __forceinline__ void ld_frag_split(Frag<float>& f, const char* tile, int r, int a, int h) {
  const char* sub = tile + (a >> 1) * 8192;
  const int al = a & 1;
  const float4 lo = *reinterpret_cast<const float4*>(sub + lds_chunk_off(r, 4 * al + h));
  const float4 hi = *reinterpret_cast<const float4*>(sub + lds_chunk_off(r, 4 * al + 2 + h));
  f.v[0] = lo.x; f.v[1] = lo.y; f.v[2] = lo.z; f.v[3] = lo.w;
  f.v[4] = hi.x; f.v[5] = hi.y; f.v[6] = hi.z; f.v[7] = hi.w;
}
// split precision: the LDS tiles hold x3 chunks (converted once per workgroup while staging: K22_ATT_LSTORE), P and Q are split in registers
__device__ __forceinline__ void ld_frag_split(Frag<x3_t>& f, const char* tile, int r, int a, int h) {
  const char* sub = tile + (a >> 1) * 8192;
  const int al = a & 1;
  // keys 16 a + 4 h + {0..3} = elements 4 h .. of group 2 al, keys 16 a + 8 + 4 h + {0..3} = the same elements of group 2 al + 1: 8-byte
  // halves of the groups' hi pieces (chunks 4 al, 4 al + 2) and lo pieces (4 al + 1, 4 al + 3)
  const u32x2_t h0 = *reinterpret_cast<const u32x2_t*>(sub + lds_chunk_off(r, 4 * al) + 8 * h);
  const u32x2_t l0 = *reinterpret_cast<const u32x2_t*>(sub + lds_chunk_off(r, 4 * al + 1) + 8 * h);
  const u32x2_t h1 = *reinterpret_cast<const u32x2_t*>(sub + lds_chunk_off(r, 4 * al + 2) + 8 * h);
  const u32x2_t l1 = *reinterpret_cast<const u32x2_t*>(sub + lds_chunk_off(r, 4 * al + 3) + 8 * h);
  f.hi = u32x4_t{h0.x, h0.y, h1.x, h1.y};
  f.lo = u32x4_t{l0.x, l0.y, l1.x, l1.y};
}
__device__ __forceinline__ void make_pfrag(Frag<x3_t>& f, const float* p) {
  x3_frag_from_f32(f, make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7]));
}
__device__ __forceinline__ void ld_qfrag(Frag<x3_t>& f, const x3_t* p) {
  x3_frag_from_f32(f, *reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4));
}
__device__ __forceinline__ void make_pfrag(Frag<bf16_t>& f, const float* p) {
  f.v = u32x4_t{pack2_bf16(p[0], p[1]),
                pack2_bf16(p[2], p[3]),
                pack2_bf16(p[4], p[5]),
                pack2_bf16(p[6], p[7])};
}
__device__ __forceinline__ void make_pfrag(Frag<f16_t>& f, const float* p) {
  f.v = u32x4_t{pack2_f16(p[0], p[1]), pack2_f16(p[2], p[3]), pack2_f16(p[4], p[5]), pack2_f16(p[6], p[7])};
}
__device__ __forceinline__ void make_pfrag(Frag<float>& f, const float* p) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = p[j];
}
__device__ __forceinline__ void ld_qfrag(Frag<bf16_t>& f, const bf16_t* p) { f.v = *reinterpret_cast<const u32x4_t*>(p); }
__device__ __forceinline__ void ld_qfrag(Frag<f16_t>& f, const f16_t* p) { f.v = *reinterpret_cast<const u32x4_t*>(p); }
__device__ __forceinline__ void ld_qfrag(Frag<float>& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// xh_t (round 5): the attention of the ASYMMETRIC split engine (K22_F16X2).  q / K_all / V^T_all are the fp32 tensors the split engines'
// qkv GEMM writes; they are rounded to fp16 while the tiles are staged (Q: at fragment load) and the kernel runs the fp16 engine's body:
// ONE MFMA per product, fp32 online softmax, native exp2.  The operand rounding of the attention is 1.5 % of the error budget of that
// engine (tests/golden/drift_ablation_x2.json, kind `a`); the output is written as x3 chunks for the proj_out GEMM, which keeps all
// three MFMAs.
struct xh_t { float f; };
template <typename T> struct AttC { using type = T; static constexpr bool MIX = false; };
template <> struct AttC<xh_t> { using type = f16_t; static constexpr bool MIX = true; };
__device__ __forceinline__ u32x4_t f16x8_from_f32(const u32x4_t a, const u32x4_t b) {
  const float4 x = __builtin_bit_cast(float4, a), y = __builtin_bit_cast(float4, b);
  return u32x4_t{pack2_f16(x.x, x.y), pack2_f16(x.z, x.w), pack2_f16(y.x, y.y), pack2_f16(y.z, y.w)};
}
__device__ __forceinline__ void ld_qfrag(Frag<f16_t>& f, const float* p) {
  f.v = f16x8_from_f32(*reinterpret_cast<const u32x4_t*>(p), *reinterpret_cast<const u32x4_t*>(p + 4));
}
// 2^x: the bf16 path uses the raw v_exp_f32 (inputs here are <= 0; results below 2^-126 flush, which is
// far below bf16 resolution of the probabilities), the fp32 parity path the accurate exp2f.
template <typename T> __device__ __forceinline__ float exp2_t(float x);
template <> __device__ __forceinline__ float exp2_t<bf16_t>(float x) { return __builtin_amdgcn_exp2f(x); }
template <> __device__ __forceinline__ float exp2_t<f16_t>(float x) { return __builtin_amdgcn_exp2f(x); }
template <> __device__ __forceinline__ float exp2_t<float>(float x) { return exp2f(x); }
template <> __device__ __forceinline__ float exp2_t<x3_t>(float x) { return exp2f(x); }

// The online softmax is the issue-bound part of this kernel (measured at T = 2304: waves ISSUING 49 % of their cycles,
// matrix pipe 18 % busy), so its instruction count is what matters:
//   * v_max3_f32 from inline asm: fmaxf() costs an extra canonicalising v_max per operand and never fuses to max3;
//   * the row sum as packed adds (v_pk_add_f32: two scores per instruction);
//   * __launch_bounds__(256, 2): with two workgroups per CU guaranteed the compiler keeps the S^T accumulators in VGPRs
//     (VGPR-form MFMA); at the default bound it parks them in AGPRs and every score pays a v_accvgpr_read.
// Together: 330 -> ~150 VALU instructions per 64-key tile per wave, 96 -> 82 us at T = 2304 (12 heads, batch 2).
// (Double-buffering the K / V^T tiles for one barrier per tile was measured too: no gain, the kernel is issue-bound.)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max2f(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// DBG (K22_ATT_PROBE builds only, wrong results; 32 = attention_pipe_kernel without its sched_group_barrier pins): 1 = the K / V^T tiles are staged once (no barriers, stores or global loads inside the loop),
// 2 = no v_exp (p = the fma result), 4 = no S MFMAs, 8 = no PV MFMAs, 16 = no max / rescale.  tools/micro/attn_probe.hip
template <typename TS, int DBG = 0>
__global__ __launch_bounds__(256, 2) void attention_kernel(AttentionParams p) {
  using T = typename AttC<TS>::type;          // arithmetic / LDS type
  constexpr bool MIX = AttC<TS>::MIX;         // fp32 tensors in memory, fp16 tiles and MFMAs (xh_t)
  using TG = typename std::conditional<MIX, float, T>::type;   // element type of q / K_all / V^T_all / out in memory
  using TR = TT<T>;
  constexpr int EPC = TR::EPC, BK = TR::BK, KSTEPS = TR::KSTEPS;
  constexpr int NSUB = 64 / BK;   // 128-byte sub-tiles per 64-element row (1 bf16, 2 fp32)
  constexpr int CPR = 64 / EPC;   // 16-byte chunks per 64-element row
  constexpr int LCH = CPR / 4;    // chunks per thread per 64x64 tile
  constexpr int GCH = MIX ? 2 * LCH : LCH;   // 16-byte global loads per thread per tile and tensor
  __shared__ __attribute__((aligned(16))) char smem[2 * NSUB * 8192];
  char* Ks = smem;
  char* Vs = smem + NSUB * 8192;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int t = blockIdx.x * 128 + wave * 32 + l31;
  const int tq = t < p.T ? t : p.T - 1;

  const TG* qrow = reinterpret_cast<const TG*>(p.q) + (int64_t)(b * p.T + tq) * p.ldq + hd * 64;
  Frag<T> qf[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) ld_qfrag(qf[a], qrow + 16 * a + 8 * h);

  const TG* Kg = reinterpret_cast<const TG*>(p.kall) + (int64_t)(b * p.H + hd) * p.Tkp * 64;
  const TG* Vg = reinterpret_cast<const TG*>(p.vtall) + (int64_t)(b * p.H + hd) * 64 * p.Tkp;
  const int nkt = (p.Tk + 63) / 64;

  // unconditional, native-vector staging (conditional staging ends up in scratch memory)
  u32x4_t kreg[GCH], vreg[GCH];
#define K22_ATT_GLOAD(KT)                                                                                \
  {                                                                                                      \
    const int kt_ = (KT);                                                                                \
    _Pragma("unroll") for (int i = 0; i < LCH; ++i) {                                                    \
      const int q = tid + i * 256, row = q / CPR, cc = q - row * CPR;                                    \
      if constexpr (MIX) {   /* one fp16 chunk of the tile = two consecutive fp32 chunks in memory */    \
        const float* kp_ = Kg + (int64_t)(kt_ * 64 + row) * 64 + cc * EPC;                               \
        const float* vp_ = Vg + (int64_t)row * p.Tkp + kt_ * 64 + cc * EPC;                              \
        kreg[2 * i] = *reinterpret_cast<const u32x4_t*>(kp_); kreg[2 * i + 1] = *reinterpret_cast<const u32x4_t*>(kp_ + 4); \
        vreg[2 * i] = *reinterpret_cast<const u32x4_t*>(vp_); vreg[2 * i + 1] = *reinterpret_cast<const u32x4_t*>(vp_ + 4); \
      } else {                                                                                           \
      kreg[i] = *reinterpret_cast<const u32x4_t*>(Kg + (int64_t)(kt_ * 64 + row) * 64 + cc * EPC);       \
      vreg[i] = *reinterpret_cast<const u32x4_t*>(Vg + (int64_t)row * p.Tkp + kt_ * 64 + cc * EPC);      \
      }                                                                                                  \
    }                                                                                                    \
  }
#define K22_ATT_LSTORE()                                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < LCH; ++i) {                                                    \
      const int q = tid + i * 256, row = q / CPR, cc = q - row * CPR;                                    \
      const int off = (cc >> 3) * 8192 + lds_chunk_off(row, cc & 7);                                     \
      if constexpr (MIX) {                                                                               \
        *reinterpret_cast<u32x4_t*>(Ks + off) = f16x8_from_f32(kreg[2 * i], kreg[2 * i + 1]);            \
        vt_store16(Vs, row, cc, f16x8_from_f32(vreg[2 * i], vreg[2 * i + 1]));                           \
      } else if constexpr (is_x3<T>::value) {   /* fp32 K / V^T rows -> x3 chunks, once per workgroup */    \
        /* groups of eight: logical chunks 2g / 2g + 1 = hi / lo pieces; this thread's four values own bytes 8 (cc & 1) .. of each */ \
        const u32x4_t ks_ = x3_split4(__builtin_bit_cast(float4, kreg[i])), vs_ = x3_split4(__builtin_bit_cast(float4, vreg[i])); \
        const int ohi_ = (cc >> 3) * 8192 + lds_chunk_off(row, (cc & 7) & ~1) + 8 * (cc & 1), olo_ = (cc >> 3) * 8192 + lds_chunk_off(row, (cc & 7) | 1) + 8 * (cc & 1); \
        *reinterpret_cast<u32x2_t*>(Ks + ohi_) = u32x2_t{ks_.x, ks_.y}; *reinterpret_cast<u32x2_t*>(Ks + olo_) = u32x2_t{ks_.z, ks_.w}; \
        *reinterpret_cast<u32x2_t*>(Vs + ohi_) = u32x2_t{vs_.x, vs_.y}; *reinterpret_cast<u32x2_t*>(Vs + olo_) = u32x2_t{vs_.z, vs_.w}; \
      } else if constexpr (sizeof(T) == 2) {                                                             \
        *reinterpret_cast<u32x4_t*>(Ks + off) = kreg[i];                                                 \
        vt_store16(Vs, row, cc, vreg[i]);                                                                \
      } else {                                                                                           \
      *reinterpret_cast<u32x4_t*>(Ks + off) = kreg[i];                                                   \
      *reinterpret_cast<u32x4_t*>(Vs + off) = vreg[i];                                                   \
      }                                                                                                  \
    }                                                                                                    \
  }

  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const float cexp = p.scale * 1.4426950408889634f;  // exp(s*scale - m*scale) = exp2((s - m) * cexp)

  K22_ATT_GLOAD(0);
  for (int kt = 0; kt < nkt; ++kt) {
    if (!(DBG & 1) || kt == 0) {
    __syncthreads();  // every wave finished reading the previous tile
    K22_ATT_LSTORE();
    __syncthreads();
    K22_ATT_GLOAD(kt + 1 < nkt ? kt + 1 : nkt - 1);
    }

    f32x16_t s[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        Frag<T> kf;
        ld_frag(kf, Ks + (a / KSTEPS) * 8192, kb * 32 + l31, a % KSTEPS, h);
        if constexpr (DBG & 4) { if constexpr (sizeof(T) == 2) { s[kb][a] += __uint_as_float(kf.v.x); s[kb][a + 4] += __uint_as_float(qf[a].v.y); } }
        else mma_atom(s[kb], kf, qf[a]);
      }

    // ---- online softmax, lane-local: p = exp2(s*c - m*c) with c = scale*log2(e) folded into ONE fma per
    // score; the running maximum is kept on the raw scores (c > 0), and the accumulator rescale is skipped
    // (wave-uniform test) on the tiles that do not raise any query's maximum — the common case.
    float pv[2][16];
    const bool tail = (kt * 64 + 64 > p.Tk);
    if (tail || p.causal || p.key_valid != nullptr) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + kb * 32 + c_row(r, lane);
          bool dead = key >= p.Tk || (p.causal && key > t);
          if (p.key_valid != nullptr && key < p.kv_n) dead = dead || p.key_valid[(int64_t)b * p.kv_ld + key] == 0.f;
          if (dead) s[kb][r] = -INFINITY;
        }
    }
    float mq[4];   // four independent max3 chains (a single chain is 16 dependent instructions)
#pragma unroll
    for (int c = 0; c < 4; ++c) mq[c] = max2f(s[0][c], s[1][c]);
#pragma unroll
    for (int r = 4; r < 16; ++r) mq[r & 3] = max3f(mq[r & 3], s[0][r], s[1][r]);
    float mloc = max2f(max3f(mq[0], mq[1], mq[2]), mq[3]);
    mloc = max2f(mloc, __shfl_xor(mloc, 32, 64));
    if constexpr (DBG & 16) mloc = s[0][0];
    if (__any(mloc > m_run)) {
      const float m_new = max2f(m_run, mloc);
      const float alpha = exp2_t<T>((m_run - m_new) * cexp);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    }
    const float mc = m_run * cexp;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t lsum2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float e0 = (DBG & 2) ? fmaf(s[kb][r], cexp, -mc) : exp2_t<T>(fmaf(s[kb][r], cexp, -mc));
        const float e1 = (DBG & 2) ? fmaf(s[kb][r + 1], cexp, -mc) : exp2_t<T>(fmaf(s[kb][r + 1], cexp, -mc));
        pv[kb][r] = e0;
        pv[kb][r + 1] = e1;
        lsum2 += f32x2_t{e0, e1};
      }
    l_run += lsum2.x + lsum2.y;

#pragma unroll
    for (int a = 0; a < 4; ++a) {
      Frag<T> pf;
      make_pfrag(pf, &pv[a >> 1][8 * (a & 1)]);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        Frag<T> vf;
        ld_frag_split(vf, Vs, db * 32 + l31, a, h);
        if constexpr (DBG & 8) { if constexpr (sizeof(T) == 2) { o[db][a] += __uint_as_float(vf.v.x); o[db][a + 4] += __uint_as_float(pf.v.y); } }
        else mma_atom(o[db], vf, pf);
      }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (t < p.T) {
    TG* orow = reinterpret_cast<TG*>(p.out) + (int64_t)(b * p.T + t) * p.ldo + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * h;
        if constexpr (MIX) {
          if (p.out_x3) x3_store4(orow + d, x3_split4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
          else *reinterpret_cast<float4*>(orow + d) = make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        } else if constexpr (sizeof(T) == 2) {
          uint2 w;
          w.x = pack2<T>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          w.y = pack2<T>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(orow + d) = w;
        } else if (is_x3<T>::value && p.out_x3) {   // the proj_out GEMM reads this tensor as its A operand: x3 chunks
          x3_store4(orow + d, x3_split4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
        } else {
          *reinterpret_cast<float4*>(orow + d) =
              make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        }
      }
  }
}

// ---- attention_pipe_kernel (round 6): the same arithmetic for the UNet's unmasked 16-bit attention, SOFTWARE-PIPELINED ----------------------
// What tools/micro/attn_probe.hip measured on attention_kernel at T = 2304 (12 heads, batch 2; profiles/r06_attention.txt): with ONE
// workgroup per CU (one wave per SIMD) a 64-key tile takes ~2600 cycles of which the matrix pipe is busy 512 - 27 % of it is the two barriers
// + tile stores of the single-buffered staging, 15 % the 32 v_exp_f32 (~12 cycles each), 15 % the max chain with its ds_bpermute round trip
// and rescale branch, and the MFMAs of a tile are strictly serial with its softmax (S -> max -> exp -> P V); a second workgroup on the CU
// recovers only a quarter of that.  This kernel removes the serialisation instead of adding waves:
//   * K and V^T tiles DOUBLE-BUFFERED in the LDS (32 KB): ONE barrier per tile; the tile stores of iteration kt (K of tile kt + 2, V^T of
//     tile kt + 1, prefetched into registers one iteration earlier) are issued right behind the barrier and have the whole iteration to land;
//   * the S MFMAs of tile kt + 1 are issued in the same instruction stream as the softmax of tile kt (two score register sets, the loop is
//     unrolled by two so that they swap roles without moves): the matrix pipe works under the VALU-bound part of the tile;
//     (the issue order is pinned with sched_group_barrier: one MFMA, eight VALU, ... - the maxima are plain fmaxf here, not the inline-asm
//     v_max3 of attention_kernel: inline asm is invisible to the instruction-group scheduler; and no DS-read groups: with them the group
//     solver drops the whole pipeline - the address adds of the reads are VALU that would have to sit in a later group);
//   * the cross-half maximum is ONE v_permlane32_swap (no LDS round trip), the accumulator rescale stays a wave-uniform branch but sits
//     between the two MFMA phases;
//   * only the last tile can be partial: its key mask is applied once, outside the common path (no causal / key_valid forms here -
//     launch_attention keeps attention_kernel for those and for the 4-byte types).
// Bit-identical to attention_kernel: the same operands in the same MFMA order, the same online-softmax decisions.
__device__ __forceinline__ float xhalf_max(float v) {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
  return __builtin_fmaxf(v, __shfl_xor(v, 32, 64));
#endif
}

template <typename TS, int DBG = 0>
__global__ __launch_bounds__(256, 2) void attention_pipe_kernel(AttentionParams p) {
  using T = typename AttC<TS>::type;
  constexpr bool MIX = AttC<TS>::MIX;
  using TG = typename std::conditional<MIX, float, T>::type;
  static_assert(sizeof(T) == 2, "attention_pipe_kernel: 16-bit tiles");
  constexpr int EPC = 8, CPR = 8, LCH = 2;
  constexpr int GCH = MIX ? 2 * LCH : LCH;
  __shared__ __attribute__((aligned(16))) char smem[4 * 8192];
  char* const Kb = smem;           // K tile i at Kb + (i & 1) * 8192
  char* const Vb = smem + 16384;   // V^T tile i (key-permuted, vt_store16) at Vb + (i & 1) * 8192

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  // 1-D grid, XCD-aware (workgroup i runs on XCD i % 8; per-XCD L2s): every XCD gets a CONTIGUOUS range of (image, head, query block)
  // triples, query blocks fastest - the K / V^T of one (image, head) are then fetched into ONE L2 (two at a range boundary) instead of
  // all eight (FETCH_SIZE of the launch at T = 2304: 8 x 14.7 MB of K / V^T before; profiles/r06_attention.txt)
  const int nqb = (p.T + 127) / 128;
  const int nblk = (int)gridDim.x;
  int L = (int)blockIdx.x;
  if (!(DBG & 64)) { const int q8 = nblk >> 3, r8 = nblk & 7, x8 = L & 7; L = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (L >> 3); }
  const int pair = L / nqb, qb = L - pair * nqb;
  const int b = pair / p.H, hd = pair - b * p.H;
  const int t = qb * 128 + wave * 32 + l31;
  const int tq = t < p.T ? t : p.T - 1;
  const TG* qrow = reinterpret_cast<const TG*>(p.q) + (int64_t)(b * p.T + tq) * p.ldq + hd * 64;
  Frag<T> qf[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) ld_qfrag(qf[a], qrow + 16 * a + 8 * h);
  const TG* Kg = reinterpret_cast<const TG*>(p.kall) + (int64_t)(b * p.H + hd) * p.Tkp * 64;
  const TG* Vg = reinterpret_cast<const TG*>(p.vtall) + (int64_t)(b * p.H + hd) * 64 * p.Tkp;
  const int nkt = (p.Tk + 63) / 64;

  u32x4_t kreg[GCH], vreg[GCH];
#define K22_AP_KLOAD(KT)                                                                                 \
  {                                                                                                      \
    const int kt_ = (KT) < nkt ? (KT) : nkt - 1;                                                         \
    _Pragma("unroll") for (int i = 0; i < LCH; ++i) {                                                    \
      const int q = tid + i * 256, row = q / CPR, cc = q - row * CPR;                                    \
      if constexpr (MIX) {                                                                               \
        const float* kp_ = reinterpret_cast<const float*>(Kg) + (int64_t)(kt_ * 64 + row) * 64 + cc * EPC; \
        kreg[2 * i] = *reinterpret_cast<const u32x4_t*>(kp_); kreg[2 * i + 1] = *reinterpret_cast<const u32x4_t*>(kp_ + 4); \
      } else {                                                                                           \
        kreg[i] = *reinterpret_cast<const u32x4_t*>(Kg + (int64_t)(kt_ * 64 + row) * 64 + cc * EPC);     \
      }                                                                                                  \
    }                                                                                                    \
  }
#define K22_AP_VLOAD(KT)                                                                                 \
  {                                                                                                      \
    const int kt_ = (KT) < nkt ? (KT) : nkt - 1;                                                         \
    _Pragma("unroll") for (int i = 0; i < LCH; ++i) {                                                    \
      const int q = tid + i * 256, row = q / CPR, cc = q - row * CPR;                                    \
      if constexpr (MIX) {                                                                               \
        const float* vp_ = reinterpret_cast<const float*>(Vg) + (int64_t)row * p.Tkp + kt_ * 64 + cc * EPC; \
        vreg[2 * i] = *reinterpret_cast<const u32x4_t*>(vp_); vreg[2 * i + 1] = *reinterpret_cast<const u32x4_t*>(vp_ + 4); \
      } else {                                                                                           \
        vreg[i] = *reinterpret_cast<const u32x4_t*>(Vg + (int64_t)row * p.Tkp + kt_ * 64 + cc * EPC);    \
      }                                                                                                  \
    }                                                                                                    \
  }
#define K22_AP_KSTORE(DST)                                                                               \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < LCH; ++i) {                                                    \
      const int q = tid + i * 256, row = q / CPR, cc = q - row * CPR;                                    \
      if constexpr (MIX) *reinterpret_cast<u32x4_t*>((DST) + lds_chunk_off(row, cc)) = f16x8_from_f32(kreg[2 * i], kreg[2 * i + 1]); \
      else *reinterpret_cast<u32x4_t*>((DST) + lds_chunk_off(row, cc)) = kreg[i];                        \
    }                                                                                                    \
  }
#define K22_AP_VSTORE(DST)                                                                               \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < LCH; ++i) {                                                    \
      const int q = tid + i * 256, row = q / CPR, cc = q - row * CPR;                                    \
      if constexpr (MIX) vt_store16((DST), row, cc, f16x8_from_f32(vreg[2 * i], vreg[2 * i + 1]));       \
      else vt_store16((DST), row, cc, vreg[i]);                                                          \
    }                                                                                                    \
  }
  // the eight K fragments of a tile (KF[2 * 4]: key block kb, k-step a) / its eight V^T fragments (VF[4 * 2]: 16-key block a, d block db)
#define K22_AP_KREAD(KF, KT_BUF)                                                                         \
  {                                                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) ld_frag(KF[kb * 4 + a], (KT_BUF), kb * 32 + l31, a, h); \
  }
#define K22_AP_VREAD(VF, VT_BUF)                                                                         \
  {                                                                                                      \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                        \
      _Pragma("unroll") for (int db = 0; db < 2; ++db) ld_frag_split(VF[a * 2 + db], (VT_BUF), db * 32 + l31, a, h); \
  }
  // S^T tile = K tile . Q^T into SC (zeroed here)
#define K22_AP_S(SC, KF)                                                                                 \
  {                                                                                                      \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                   \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) SC[kb][r] = 0.f;                                    \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                    \
        if constexpr (DBG & 4) { SC[kb][a] += __uint_as_float(KF[kb * 4 + a].v.x); SC[kb][a + 4] += __uint_as_float(qf[a].v.y); } \
        else mma_atom(SC[kb], KF[kb * 4 + a], qf[a]);                                                    \
      }                                                                                                  \
    }                                                                                                    \
  }

  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const float cexp = p.scale * 1.4426950408889634f;

  // prologue: K tiles 0 and 1 and V^T tile 0 in the LDS, K tile 2 and V^T tile 1 on their way into the registers, S of tile 0
  K22_AP_KLOAD(0);
  K22_AP_VLOAD(0);
  K22_AP_KSTORE(Kb);
  K22_AP_VSTORE(Vb);
  K22_AP_KLOAD(1);
  K22_AP_KSTORE(Kb + 8192);
  K22_AP_KLOAD(2);
  K22_AP_VLOAD(1);
  __syncthreads();
  f32x16_t s0[2], s1[2];
  {
    Frag<T> kf0[8];
    K22_AP_KREAD(kf0, Kb);
    K22_AP_S(s0, kf0);
  }

  // one tile: SC = its scores (complete), SN = the next tile's (issued here); PAR = kt & 1 (compile time)
#define K22_AP_TILE(SC, SN, PAR, KT)                                                                     \
  {                                                                                                      \
    if constexpr (!(DBG & 1)) {                                                                          \
    __syncthreads();   /* K tile kt + 1 / V^T tile kt visible; every wave is done with K tile kt and V^T tile kt - 1 */ \
    K22_AP_KSTORE(Kb + (PAR) * 8192);          /* K tile kt + 2 */                                       \
    K22_AP_VSTORE(Vb + (1 - (PAR)) * 8192);    /* V^T tile kt + 1 */                                     \
    K22_AP_KLOAD((KT) + 3);                                                                              \
    K22_AP_VLOAD((KT) + 2);                                                                              \
    }                                                                                                    \
    if ((KT) == nkt - 1 && (p.Tk & 63)) {      /* the one partial tile */                                \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                   \
          if ((KT) * 64 + kb * 32 + c_row(r, lane) >= p.Tk) SC[kb][r] = -INFINITY;                       \
    }                                                                                                    \
    /* every fragment read of the tile is issued HERE, ahead of the arithmetic (sched_barrier: nothing moves across): behind the MFMA that   \
       uses it a ds_read_b128 costs its full LDS latency per MFMA - the first form of this kernel waited ~100 cycles in front of each */    \
    Frag<T> kf_[8], vf_[8];                                                                              \
    K22_AP_KREAD(kf_, Kb + (1 - (PAR)) * 8192);                                                          \
    K22_AP_VREAD(vf_, Vb + (PAR) * 8192);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    /* phase A: S of the next tile (matrix pipe) under the softmax of this one (VALU).  One basic block, issue order pinned:        \
       four K-fragment reads and the max chain first (the chain covers the read latency), then one MFMA + one read + a share of the    \
       fma / v_exp pairs, eight times - an MFMA occupies the pipe for 32 cycles, during which the wave issues the VALU behind it. */   \
    K22_AP_S(SN, kf_);                                                                                   \
    float mq[4];                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) mq[c] = __builtin_fmaxf(SC[0][c], SC[1][c]);           \
    _Pragma("unroll") for (int r = 4; r < 16; ++r) mq[r & 3] = __builtin_fmaxf(__builtin_fmaxf(mq[r & 3], SC[0][r]), SC[1][r]); \
    const float mloc = xhalf_max(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mq[0], mq[1]), mq[2]), mq[3])); \
    const float m_old = m_run;                                                                           \
    m_run = (DBG & 16) ? SC[0][0] : __builtin_fmaxf(m_run, mloc);                                        \
    const float mc = m_run * cexp;                                                                       \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                     \
        SC[kb][r] = (DBG & 2) ? fmaf(SC[kb][r], cexp, -mc) : __builtin_amdgcn_exp2f(fmaf(SC[kb][r], cexp, -mc)); \
    if constexpr (!(DBG & 32)) {                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x002, 30, 0);                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                                 \
    }                                                                                                    \
    }                                                                                                    \
    /* everything above belongs in front of the rescale branch (the compiler would otherwise sink the v_exp block behind it) */        \
    asm volatile("" : "+v"(SC[0]), "+v"(SC[1]), "+v"(SN[0]), "+v"(SN[1]));                               \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    /* rescale where a maximum rose (wave-uniform; rare after the first tiles) */                        \
    if (__any(m_run > m_old)) {                                                                          \
      const float alpha = __builtin_amdgcn_exp2f((m_old - m_run) * cexp);                                \
      l_run *= alpha;                                                                                    \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }             \
    }                                                                                                    \
    /* phase B: P V (matrix pipe) under the row sums and the P conversions (VALU) */                     \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                      \
      Frag<T> pf;                                                                                        \
      float pv_[8];                                                                                      \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) pv_[j] = SC[a >> 1][8 * (a & 1) + j];                \
      make_pfrag(pf, pv_);                                                                               \
      _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                                 \
        if constexpr (DBG & 8) { o[db][a] += __uint_as_float(vf_[a * 2 + db].v.x); o[db][a + 4] += __uint_as_float(pf.v.y); } \
        else mma_atom(o[db], vf_[a * 2 + db], pf);                                                       \
      }                                                                                                  \
    }                                                                                                    \
    float lsa = 0.f, lsb = 0.f;                                                                          \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
      _Pragma("unroll") for (int r = 0; r < 16; r += 2) { lsa += SC[kb][r]; lsb += SC[kb][r + 1]; }      \
    l_run += lsa + lsb;                                                                                  \
    if constexpr (!(DBG & 32)) {                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                                 \
    }                                                                                                    \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  }
  for (int kt = 0; kt < nkt; kt += 2) {
    K22_AP_TILE(s0, s1, 0, kt);
    if (kt + 1 < nkt) K22_AP_TILE(s1, s0, 1, kt + 1);
  }
#undef K22_AP_TILE
#undef K22_AP_S
#undef K22_AP_KREAD
#undef K22_AP_VREAD
#undef K22_AP_KLOAD
#undef K22_AP_VLOAD
#undef K22_AP_KSTORE
#undef K22_AP_VSTORE

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (t < p.T) {
    TG* orow = reinterpret_cast<TG*>(p.out) + (int64_t)(b * p.T + t) * p.ldo + hd * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * h;
        if constexpr (MIX) {
          if (p.out_x3) x3_store4(orow + d, x3_split4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
          else *reinterpret_cast<float4*>(orow + d) = make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        } else {
          uint2 w;
          w.x = pack2<T>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          w.y = pack2<T>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(orow + d) = w;
        }
      }
  }
}

// ---- small-T attention of the prior transformer (QKVMultiheadAttention, kandinsky2/model/prior.py:86-102) ---------------------------
// T <= 128 tokens (the prior: 81), one workgroup per (head, image), every key in ONE tile: no online rescale, no kv_pack pass and no
// per-score global mask loads.  q / k / v are read straight from the c_qkv output  qkv[B * T][ldq] = [Q | K | V] x [head][64]  (row-major,
// written by the skinny GEMM); K goes to the LDS as rows, V transposed, the additive mask of prior.py:262-263 (-inf for padding keys and
// for keys after the query) as one float per key.  Wave w owns queries 32 w .. 32 w + 31; the arithmetic is attention_kernel's (S^T = K Q^T
// so a lane owns one query, lane-local fp32 softmax, P from the accumulator registers).  The output is written row-major or in the
// A-fragment order of the skinny GEMM that reads it (c_proj), selected by out_frag.  16-bit storage types.
template <typename T>
__global__ __launch_bounds__(256) void small_attention_kernel(SmallAttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + 512];
  char* Ks = smem;            // [128 keys][128 B]
  char* Vs = smem + 16384;    // V^T: 2 sub-tiles of [64 d][64 keys]
  float* kdead = reinterpret_cast<float*>(smem + 32768);   // 1 = padding key
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int hd = blockIdx.x, b = blockIdx.y;
  const int C = p.H * 64;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.T * p.ldq + hd * 64;
  // 8 consecutive values of token `tok` (of image b) at column `col` of this head's Q / K / V plane `which`: the stored T row, or the
  // finish of c_qkv's split-K partials done here (bias + partials in split order, ONE rounding to T - what a finish launch would store)
  auto fetch8 = [&](int tok, int which, int col) __attribute__((always_inline)) -> u32x4_t {
    if (p.part == nullptr) return *reinterpret_cast<const u32x4_t*>(base + (int64_t)tok * p.ldq + which * C + col);
    const int64_t e = ((int64_t)b * p.T + tok) * p.ldq + which * C + hd * 64 + col, tot = (int64_t)p.B * p.T * p.ldq;
    float4 q0[4], q1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ss = s < p.nsplit ? s : p.nsplit - 1;
      q0[s] = *reinterpret_cast<const float4*>(p.part + ss * tot + e);
      q1[s] = *reinterpret_cast<const float4*>(p.part + ss * tot + e + 4);
    }
    float4 a0 = q0[0], a1 = q1[0];
#pragma unroll
    for (int s = 1; s < 4; ++s)
      if (s < p.nsplit) {
        a0.x += q0[s].x; a0.y += q0[s].y; a0.z += q0[s].z; a0.w += q0[s].w;
        a1.x += q1[s].x; a1.y += q1[s].y; a1.z += q1[s].z; a1.w += q1[s].w;
      }
    if (p.bias != nullptr) {
      const float* bp = p.bias + which * C + hd * 64 + col;
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
      a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    }
    return u32x4_t{pack2<T>(a0.x, a0.y), pack2<T>(a0.z, a0.w), pack2<T>(a1.x, a1.y), pack2<T>(a1.z, a1.w)};
  };
  // this wave's queries: loaded before the staging barrier so that their latency hides behind it
  const int t = wave * 32 + l31;
  const int tq = t < p.T ? t : p.T - 1;
  Frag<T> qf[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) qf[a].v = fetch8(tq, 0, 16 * a + 8 * h);
  // K rows / V^T columns: 128 keys x 8 chunks of 16 bytes; keys >= T are zero (their scores are masked; zero keeps 0 * V finite)
  for (int i = tid; i < 128 * 8; i += 256) {
    const int key = i >> 3, c = i & 7;
    u32x4_t kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (key < p.T) {
      kv = fetch8(key, 1, c * 8);
      vv = fetch8(key, 2, c * 8);
    }
    *reinterpret_cast<u32x4_t*>(Ks + lds_chunk_off(key, c)) = kv;
    const unsigned short* ve = reinterpret_cast<const unsigned short*>(&vv);
    char* vsub = Vs + (key >> 6) * 8192;
    const int kk = key & 63;
#pragma unroll
    for (int e = 0; e < 8; ++e)   // key-permuted V^T tile (vt_store16's layout): key kk -> piece 2 (kk >> 4) + ((kk >> 2) & 1), byte 8 ((kk >> 3) & 1) + 2 (kk & 3)
      *reinterpret_cast<unsigned short*>(vsub + lds_chunk_off(c * 8 + e, 2 * (kk >> 4) + ((kk >> 2) & 1)) + 8 * ((kk >> 3) & 1) + 2 * (kk & 3)) = ve[e];
  }
  if (tid < 128) kdead[tid] = (p.key_valid != nullptr && tid < p.kv_n && p.key_valid[(int64_t)b * p.kv_ld + tid] == 0.f) ? 1.f : 0.f;
  __syncthreads();
  if (wave * 32 >= p.T) return;   // uniform per wave; no barrier below
  const int nkb = (p.T + 31) >> 5;   // 32-key blocks (<= 4)
  f32x16_t s[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if (kb < nkb) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        Frag<T> kf;
        ld_frag(kf, Ks, kb * 32 + l31, a, h);
        mma_atom(s[kb], kf, qf[a]);
      }
    }
  }
  float mloc = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + c_row(r, lane);
      const bool dead = key >= p.T || (p.causal && key > t) || kdead[key & 127] != 0.f;
      s[kb][r] = dead ? -INFINITY : s[kb][r];
      mloc = fmaxf(mloc, s[kb][r]);
    }
  mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
  const float cexp = p.scale * 1.4426950408889634f;
  const float mc = mloc * cexp;
  float pv[4][16];
  float l = 0.f;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = exp2_t<T>(fmaf(s[kb][r], cexp, -mc));
      pv[kb][r] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    if (a < 2 * nkb) {
      Frag<T> pf;
      make_pfrag(pf, &pv[a >> 1][8 * (a & 1)]);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        Frag<T> vf;
        ld_frag_split(vf, Vs + (a >> 2) * 8192, db * 32 + l31, a & 3, h);
        mma_atom(o[db], vf, pf);
      }
    }
  }
  if (t >= p.T) return;
  const float inv = 1.f / l;
  const int m = b * p.T + t;
  T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = db * 32 + 8 * g + 4 * h;
      uint2 w2;
      w2.x = pack2<T>(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
      w2.y = pack2<T>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
      if (p.out_frag) {
        const int k = hd * 64 + d;   // K index of the consumer GEMM: [K / 64][MA][4][64][8]
        const int64_t off = ((((int64_t)(k >> 6) * p.MA + (m >> 5)) * 4 + ((k >> 4) & 3)) * 64 + (m & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
        *reinterpret_cast<uint2*>(out + off) = w2;
      } else {
        *reinterpret_cast<uint2*>(out + (int64_t)m * p.ldo + hd * 64 + d) = w2;
      }
    }
}

int launch_small_attention(const SmallAttnParams& p, int dtype, hipStream_t s) {
  if (p.T < 1 || p.T > 128 || p.H < 1 || p.B < 1 || (p.ldq & 7) || (!p.out_frag && (p.ldo & 3)) || (p.part != nullptr && (p.nsplit < 1 || p.nsplit > 4)))
    return k22_set_error(K22_EINVAL, "small_attention: 1 <= T <= 128, 64 channels per head, 16-byte aligned rows");
  dim3 grid(p.H, p.B);
  if (dtype == K22_BF16) hipLaunchKernelGGL(small_attention_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16) hipLaunchKernelGGL(small_attention_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else return k22_set_error(K22_EINVAL, "small_attention: 16-bit storage types only");
  K22_CHECK_LAUNCH();
  return K22_OK;
}

int launch_attention(const AttentionParams& p, int dtype, hipStream_t s) {
  if (p.Tkp % 64 || p.Tkp < p.Tk) return k22_set_error(K22_EINVAL, "attention: Tkp must be roundup(Tk,64)");
  dim3 grid((p.T + 127) / 128, p.H, p.B);
  // the UNet's / MoVQ-free unmasked attention on 16-bit tiles: the software-pipelined kernel (K22_ATT_PIPE=0: attention_kernel, for A/B runs)
  static const bool pipe = [] { const char* e = getenv("K22_ATT_PIPE"); return !(e && e[0] == '0'); }();
  if (pipe && !p.causal && p.key_valid == nullptr && (dtype == K22_BF16 || dtype == K22_F16 || dtype == K22_F16X2)) {
    const dim3 g1(grid.x * grid.y * grid.z);   // 1-D: the kernel maps workgroups to (image, head, query block) itself (XCD-aware)
    if (dtype == K22_BF16) hipLaunchKernelGGL(attention_pipe_kernel<bf16_t>, g1, dim3(256), 0, s, p);
    else if (dtype == K22_F16) hipLaunchKernelGGL(attention_pipe_kernel<f16_t>, g1, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attention_pipe_kernel<xh_t>, g1, dim3(256), 0, s, p);
    K22_CHECK_LAUNCH();
    return K22_OK;
  }
  if (dtype == K22_BF16) hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16) hipLaunchKernelGGL(attention_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F32) hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16X3) hipLaunchKernelGGL(attention_kernel<x3_t>, grid, dim3(256), 0, s, p);
  else if (dtype == K22_F16X2) hipLaunchKernelGGL(attention_kernel<xh_t>, grid, dim3(256), 0, s, p);
  else return k22_set_error(K22_EINVAL, "attention: bad dtype");
  K22_CHECK_LAUNCH();
  return K22_OK;
}
