// k22 — conditioning-encoder engine: the three transformer towers that turn a prompt / an image into the embeddings the prior and
// the UNet consume.  One config-driven engine (K22Encoder) covers
//
//   K22_ENC_CLIP_TEXT    OpenAI CLIP ViT-L/14 text tower as Kandinsky2_1.generate_clip_emb walks it
//                        (kandinsky2/kandinsky2_1_model.py:159-168): token + positional embedding, 12 pre-LN causal blocks with
//                        QuickGELU MLPs, ln_final -> txt_feat_seq [B,77,768]; row argmax(tokens) @ text_projection -> txt_feat
//   K22_ENC_CLIP_VISION  clip_model.encode_image (kandinsky2_1_model.py:177-181): 14x14 patch embedding, class token,
//                        positional embedding, ln_pre, 24 pre-LN blocks, ln_post(class token) @ proj -> [B,768]
//   K22_ENC_XLMR         MultilingualCLIP.forward (kandinsky2/model/text_encoders.py:108-122): transformers' XLMRobertaModel
//                        (embeddings + LayerNorm, 24 post-LN blocks, erf-GELU, key-padding mask) -> embs [B,77,1024];
//                        LinearTransformation(masked mean over tokens) -> [B,768]
//
// They run once per prompt, not per denoising step (a "next" row, outside the per-step roofline accounting).
// MI355X mapping: the same frame as the prior engine (prior.hip) - every Linear is one launch_igemm over the [N][K] weight, the
// fp32 residual stream is updated in place by the GEMM epilogue, LayerNorm writes the next GEMM's operand, attention is the UNet's
// flash kernel with a causal / key-validity mask.  QuickGELU is applied by a separate pass on the fp32 accumulator output of c_fc
// (one rounding, like a fused epilogue) so that the hot GEMM kernels' epilogue is left exactly as validated.
// Tile configurations of the transformer Linears come from the process-wide tile table like the other engines' (tuning.h): the
// production shapes (2/4/8 sequences, 1/2 images) are in the shipped table, so nothing is timed for them and the bits are the same
// on every box; other shapes are measured at their first pass (K22_AUTOTUNE=0: the fixed heuristic).
#include "kernels.h"
#include "elementwise.h"
#include "../../include/k22.h"
#include "tuning.h"

#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// ---- LayerNorm over the last dim of fp32 rows; optional fp32 output (may alias the input row: post-LN blocks normalise the
// residual stream in place) and optional T output (the next GEMM's operand).  D <= 2048. -------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void enc_layernorm_kernel(const float* x, int64_t ldx, const float* g, const float* bta, float* yf,
                                                            int64_t ldyf, TO* yt, int64_t ldyt, int D, float eps) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)row * ldx;
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int i = tid + k * 256; v[k] = i < D ? xr[i] : 0.f; s += v[k]; }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int i = tid + k * 256; if (i < D) { const float d = v[k] - mean; q += d * d; } }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / D + eps);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = tid + k * 256;
    if (i < D) {
      const float y = (v[k] - mean) * rstd * g[i] + bta[i];
      if (yf != nullptr) yf[(int64_t)row * ldyf + i] = y;
      if (yt != nullptr) yt[(int64_t)row * ldyt + i] = from_f32<TO>(y);
    }
  }
}

// ---- token (+ position (+ token-type)) embedding -> fp32 sequence.  xlmr: position ids as transformers'
// create_position_ids_from_input_ids: (running count of non-padding tokens) * (token != pad) + pad_id -------------------------------
__global__ __launch_bounds__(256) void enc_embed_kernel(const int* tok, const float* tok_emb, const float* pos_emb, const float* type_emb,
                                                        float* x, int n_ctx, int D, int vocab, int xlmr, int pad_id, int max_pos) {
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int* tb = tok + (int64_t)b * n_ctx;
  int id = tb[t];
  int pos = t;
  if (xlmr) {
    int cnt = 0;
    for (int j = 0; j <= t; ++j) cnt += tb[j] != pad_id;
    pos = (id != pad_id ? cnt : 0) + pad_id;
    pos = pos < max_pos ? pos : max_pos - 1;
  }
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* te = tok_emb + (int64_t)id * D;
  const float* pe = pos_emb + (int64_t)pos * D;
  float* xr = x + ((int64_t)b * n_ctx + t) * D;
  for (int i = tid; i < D; i += 256) xr[i] = te[i] + pe[i] + (type_emb != nullptr ? type_emb[i] : 0.f);
}

// ---- CLIP text: the row of every sequence at its highest token id (the end-of-text token), torch.argmax semantics (first) --------
__global__ __launch_bounds__(256) void enc_gather_eot_kernel(const int* tok, const float* x, float* out, int n_ctx, int D) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* tb = tok + (int64_t)b * n_ctx;
  int best = 0, bv = tb[0];
  for (int j = 1; j < n_ctx; ++j) if (tb[j] > bv) { bv = tb[j]; best = j; }
  const float* xr = x + ((int64_t)b * n_ctx + best) * D;
  for (int i = tid; i < D; i += 256) out[(int64_t)b * D + i] = xr[i];
}

// ---- XLM-R: (embs * mask).sum(1) / mask.sum(1) ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_masked_mean_kernel(const float* x, const float* mask, float* out, int n_ctx, int D) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* mb = mask + (int64_t)b * n_ctx;
  float den = 0.f;
  for (int t = 0; t < n_ctx; ++t) den += mb[t];
  for (int i = tid; i < D; i += 256) {
    float s = 0.f;
    for (int t = 0; t < n_ctx; ++t) s += x[((int64_t)b * n_ctx + t) * D + i] * mb[t];
    out[(int64_t)b * D + i] = s / den;
  }
}

// ---- vision: [B][3][S][S] fp32 -> patch rows [B*P][Kp] T (column c*p*p + i*p + j as conv1.weight flattens; zero padding columns) ---
template <typename T>
__global__ __launch_bounds__(256) void enc_patchify_kernel(const float* img, T* out, int S, int patch, int Kp) {
  const int g = S / patch, P = g * g;
  const int row = blockIdx.x, b = row / P, pi = row % P, py = pi / g, px = pi % g;
  const int K = 3 * patch * patch;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float v = 0.f;
    if (k < K) {
      const int c = k / (patch * patch), r = k % (patch * patch), i = r / patch, j = r % patch;
      v = img[(((int64_t)b * 3 + c) * S + py * patch + i) * S + px * patch + j];
    }
    out[(int64_t)row * Kp + k] = from_f32<T>(v);
  }
}
// x[b][0] = class_embedding + pos[0];  x[b][1+p] = patch_out[b*P+p] + pos[1+p]
__global__ __launch_bounds__(256) void enc_vision_assemble_kernel(const float* patch_out, const float* cls, const float* pos, float* x, int P, int D) {
  const int b = blockIdx.y, t = blockIdx.x;
  const float* src = t == 0 ? cls : patch_out + ((int64_t)b * P + (t - 1)) * D;
  float* xr = x + ((int64_t)b * (P + 1) + t) * D;
  for (int i = threadIdx.x; i < D; i += 256) xr[i] = src[i] + pos[(int64_t)t * D + i];
}

// ---- MLP activation on fp32 rows -> T rows: QuickGELU (x * sigmoid(1.702 x), clip/model.py QuickGELU) or, exact != 0, the erf GELU
// of transformers' CLIP with hidden_act "gelu" (the open_clip ViT-bigG/14 image encoder of Kandinsky 2.2) ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void enc_quickgelu_kernel(const float* x, T* y, int64_t n, int exact) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    y[i] = from_f32<T>(exact ? gelu_f(v) : v / (1.0f + __expf(-1.702f * v)));
  }
}

// ---- attention for head widths other than 64 (CLIP ViT-bigG/14: 1664 / 16 = 104 per head), short sequences (n <= 512), no mask:
// CLIPAttention.forward of transformers (softmax((q * hd^-0.5) k^T) v per head).  One workgroup = 16 queries of one (image, head):
// the head's K rows sit in the LDS as fp32 (row stride hd + 1 words: conflict-free across keys), every wave walks its 4 queries one
// at a time - lane = key for the scores (5 keys per lane at n = 257), lane = channel for P.V (V rows straight from L2, coalesced).
// Plain fp32 FMAs: 48 layers x 27 MFLOP per head is ~20 GFLOP per image, a few ms once per generation - not worth an MFMA tiling.
template <typename T>
__global__ __launch_bounds__(256) void enc_attention_generic_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n, int D, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) char esm[];
  // K and V rows of this (image, head) in the LDS, in the storage type (row stride hd + 2 elements: odd word stride for the 16-bit types,
  // so that 64 keys read one channel without bank conflicts).  fp32 storage (parity path): K only - V rows come from L2 (K + V would
  // need 216 KB); that path is for 1e-6 parity checks, not for speed.
  constexpr bool V_IN_LDS = sizeof(T) == 2;
  const int rs = hd + 2;
  T* Ks = reinterpret_cast<T*>(esm);                               // [n][rs]
  T* Vs = Ks + (size_t)n * rs;                                     // [n][rs] (16-bit types)
  float* qs = reinterpret_cast<float*>(esm + (((size_t)n * rs * sizeof(T) * (V_IN_LDS ? 2 : 1)) + 15) / 16 * 16);   // [4 waves][128]
  float* ps = qs + 4 * 128;                                        // [4 waves][512]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 16;   // 16 queries per workgroup (4 per wave): 17 x 16 heads = 272 workgroups per image at n = 257
  const int64_t ld = 3 * (int64_t)D;
  const T* base = qkv + (int64_t)b * n * ld + head * hd;
  for (int i = tid; i < n * hd; i += 256) {
    const int key = i / hd, d = i - key * hd;
    Ks[key * rs + d] = base[(int64_t)key * ld + D + d];
    if (V_IN_LDS) Vs[key * rs + d] = base[(int64_t)key * ld + 2 * D + d];
  }
  __syncthreads();
  float* qw = qs + wave * 128;
  float* pw = ps + wave * 512;
  for (int qq = 0; qq < 4; ++qq) {
    const int qi = q0 + wave * 4 + qq;
    if (qi >= n) break;                                            // wave-uniform
    for (int d = lane; d < hd; d += 64) qw[d] = to_f32(base[(int64_t)qi * ld + d]) * scale;
    __builtin_amdgcn_wave_barrier();
    float sc[8];
    float m = -3.0e38f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int key = kk * 64 + lane;
      float a = -3.0e38f;
      if (kk * 64 < n) {                                           // wave-uniform: whole key blocks past the sequence cost nothing
        a = 0.f;
        const T* kr = Ks + (key < n ? key : n - 1) * rs;
        for (int d = 0; d < hd; ++d) a += qw[d] * to_f32(kr[d]);
        if (key >= n) a = -3.0e38f;
      }
      sc[kk] = a;
      m = fmaxf(m, a);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int key = kk * 64 + lane;
      if (key < n) {
        const float e = __expf(sc[kk] - m);
        pw[key] = e;
        l += e;
      }
    }
    l = wave_sum(l);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.f / l;
    for (int d = lane; d < hd; d += 64) {
      float acc = 0.f;
      if (V_IN_LDS) {
        const T* vp = Vs + d;
        for (int key = 0; key < n; ++key) acc += pw[key] * to_f32(vp[key * rs]);
      } else {
        const T* vp = base + 2 * D + d;
        for (int key = 0; key < n; ++key) acc += pw[key] * to_f32(vp[(int64_t)key * ld]);
      }
      out[((int64_t)b * n + qi) * D + head * hd + d] = from_f32<T>(acc * inv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

struct ESlot { size_t bytes = 0, off = 0; };
typedef std::function<int(hipStream_t)> EOp;
}  // namespace

struct K22Encoder {
  K22EncoderConfig cfg;
  int dtype; size_t esz;
  std::unordered_map<std::string, const void*> w;
  int B = 0;
  std::deque<ESlot> slots;
  std::vector<EOp> ops;
  std::deque<Tuned> tuned;   // every transformer Linear: its tile configuration (table / measurement at the first pass)
  bool tuned_done = false;
  int autotune = 1;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  std::string err;
  hipGraphExec_t graph_exec = nullptr;   // the 150-300 launches of one tower pass, replayed as one graph
  hipStream_t cap_stream = nullptr;
  ~K22Encoder() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
  }
  ESlot *s_tok, *s_valid, *s_img, *s_patch, *s_pout, *s_inp, *s_ln, *s_qkv, *s_att, *s_fc, *s_fc32, *s_seq, *s_pool_in, *s_pooled, *s_splitk,
      *s_kall, *s_vtall, *s_flush;

  ESlot* new_slot(size_t bytes = 0) { slots.emplace_back(); slots.back().bytes = bytes; return &slots.back(); }
  static void need(ESlot* s, size_t bytes) { if (bytes > s->bytes) s->bytes = bytes; }
  template <typename T = char> T* ptr(const ESlot* s) const { return reinterpret_cast<T*>(ws + s->off); }
  const void* W_(const std::string& name) {
    auto it = w.find(name);
    if (it == w.end()) { if (err.empty()) err = "missing weight: " + name; return nullptr; }
    return it->second;
  }
  const float* Wf(const std::string& name) { return reinterpret_cast<const float*>(W_(name)); }

  // out = A[M][K] . W[N][K]^T + bias:  mode 0 -> T rows; 1 -> fp32 rows; 2 -> fp32 rows += (in-place residual stream)
  void op_linear(ESlot* a, int M, int N, int K, const std::string& pfx, int act, ESlot* dst, int ldo, int mode) {
    tuned.emplace_back();
    Tuned* t = &tuned.back();
    IgemmParams& p = t->p;
    p.stages = -1;
    p.M = M; p.N = N; p.Npad = (N + 63) / 64 * 64; p.Kc = K; p.K0 = K; p.taps = 1; p.lda0 = K; p.ldo = ldo; p.ldr = ldo;
    p.out_mode = mode == 0 ? IG_OUT_ROWMAJOR : IG_OUT_ROWMAJOR_F32; p.act = act; p.res_f32 = mode == 2 ? 1 : 0;
    p.Wp = W_(pfx + ".weight"); p.bias = Wf(pfx + ".bias");
    tuned_make_candidates(*t, dtype);
    tuned_default_cfg(*t, dtype);
    need(s_splitk, tuned_max_splitk_bytes(*t, autotune != 0));
    const int dt = dtype;
    t->run = [=](hipStream_t st) {
      IgemmParams q = t->p;
      tuned_apply_cfg(q, t->cfg);
      q.A0 = ptr(a); q.out = ptr(dst); q.partial = ptr<float>(s_splitk);
      q.residual = mode == 2 ? ptr(dst) : nullptr;
      return launch_igemm(q, dt, st);
    };
    ops.push_back([=](hipStream_t st) { return t->run(st); });
  }
  // LayerNorm of `rows` fp32 rows of x (stride ldx): fp32 copy into yf (may be x itself) and / or T copy into yt
  void op_ln(ESlot* x, size_t x_off, int64_t ldx, int rows, const std::string& pfx, ESlot* yf, int64_t ldyf, ESlot* yt) {
    const float* g = Wf(pfx + ".weight"); const float* b = Wf(pfx + ".bias");
    const int D = cfg.width, dt = dtype;
    const float eps = cfg.ln_eps;
    ops.push_back([=](hipStream_t st) {
      const float* xp = reinterpret_cast<const float*>(ptr(x) + x_off);
      float* f = yf ? ptr<float>(yf) : nullptr;
      if (yt != nullptr && dt == K22_BF16)
        hipLaunchKernelGGL(enc_layernorm_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, f, ldyf, ptr<bf16_t>(yt), (int64_t)D, D, eps);
      else if (yt != nullptr && dt == K22_F16)
        hipLaunchKernelGGL(enc_layernorm_kernel<f16_t>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, f, ldyf, ptr<f16_t>(yt), (int64_t)D, D, eps);
      else
        hipLaunchKernelGGL(enc_layernorm_kernel<float>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, f, ldyf, yt ? ptr<float>(yt) : nullptr, (int64_t)D, D, eps);
      K22_CHECK_LAUNCH();
      return K22_OK;
    });
  }

  int plan(int nB) {
    const int D = cfg.width, n = cfg.n_ctx, M = nB * n, heads = cfg.heads, od = cfg.out_dim, kind = cfg.kind;
    if (nB < 1 || nB > 8) return k22_set_error(K22_EINVAL, "encoder: 1..8 sequences / images per call");
    if (D < 64 || D % 64 || D > 2048 || heads < 1 || D % heads) return k22_set_error(K22_EINVAL, "encoder: width % 64 == 0, width <= 2048, width % heads == 0");
    const int hd = D / heads;
    const bool flash = hd == 64;       // the UNet's flash attention kernel; anything else: enc_attention_generic_kernel (vision tower only)
    const size_t ga_smem = ((size_t)n * (hd + 2) * esz * (esz == 2 ? 2 : 1) + 15) / 16 * 16 + 4 * 640 * 4;   // enc_attention_generic_kernel
    if (!flash && (cfg.kind != K22_ENC_CLIP_VISION || hd > 128 || n > 512 || ga_smem > 160 * 1024))
      return k22_set_error(K22_EINVAL, "encoder: head widths other than 64 are built for the vision tower (<= 128 per head, <= 512 tokens)");
    const int F = cfg.mlp_dim > 0 ? cfg.mlp_dim : 4 * D;
    if (F % 64) return k22_set_error(K22_EINVAL, "encoder: mlp_dim % 64");
    if (kind < K22_ENC_CLIP_TEXT || kind > K22_ENC_XLMR) return k22_set_error(K22_EINVAL, "encoder: kind");
    if (n < 1 || cfg.layers < 1 || od < 1) return k22_set_error(K22_EINVAL, "encoder: n_ctx, layers and out_dim must be positive");
    const bool vision = kind == K22_ENC_CLIP_VISION, xlmr = kind == K22_ENC_XLMR;
    if (vision && (cfg.patch < 1 || cfg.image_size < cfg.patch)) return k22_set_error(K22_EINVAL, "encoder: patch / image_size");
    if (xlmr && (cfg.max_pos < 2 || cfg.pad_id < 0 || cfg.pad_id >= cfg.max_pos)) return k22_set_error(K22_EINVAL, "encoder: max_pos / pad_id");
    const int g = vision ? cfg.image_size / cfg.patch : 0, P = g * g;
    const int Kraw = 3 * cfg.patch * cfg.patch, Kp = (Kraw + 63) / 64 * 64;
    if (vision && (cfg.image_size % cfg.patch || P + 1 != n)) return k22_set_error(K22_EINVAL, "encoder: n_ctx must be (image_size/patch)^2 + 1");
    if (!vision && cfg.vocab < 1) return k22_set_error(K22_EINVAL, "encoder: vocab");
    // validated: only now drop the previous plan
    B = nB;
    slots.clear(); ops.clear(); err.clear(); ws = nullptr; tuned.clear(); tuned_done = false;
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    s_tok = new_slot((size_t)M * 4); s_valid = new_slot((size_t)M * 4);
    s_img = new_slot(vision ? (size_t)B * 3 * cfg.image_size * cfg.image_size * 4 : 0);
    s_patch = new_slot(vision ? (size_t)B * P * Kp * esz : 0); s_pout = new_slot(vision ? (size_t)B * P * D * 4 : 0);
    s_inp = new_slot((size_t)M * D * 4); s_ln = new_slot((size_t)M * D * esz); s_qkv = new_slot((size_t)M * 3 * D * esz);
    s_att = new_slot((size_t)M * D * esz); s_fc = new_slot((size_t)M * F * esz); s_fc32 = new_slot(xlmr ? 0 : (size_t)M * F * 4);
    s_seq = new_slot((size_t)M * D * 4); s_pool_in = new_slot((size_t)B * D * 4); s_pooled = new_slot((size_t)B * od * 4);
    s_splitk = new_slot(256); s_kall = new_slot(); s_vtall = new_slot();
    s_flush = new_slot(autotune ? ((size_t)320 << 20) : 0);
    const int Bn = B, dt = dtype;

    // ---- input sequence ------------------------------------------------------------------------------------------------------
    if (vision) {
      // x = conv1(image) as a GEMM over 14x14 patches (clip/model.py VisionTransformer.forward)
      const void* wp = W_("patch.weight");
      const float* cls = Wf("class_embedding"); const float* pos = Wf("positional_embedding");
      const int S = cfg.image_size, patch = cfg.patch;
      ops.push_back([=](hipStream_t st) {
        if (dt == K22_BF16) hipLaunchKernelGGL(enc_patchify_kernel<bf16_t>, dim3(Bn * P), dim3(256), 0, st, ptr<float>(s_img), ptr<bf16_t>(s_patch), S, patch, Kp);
        else if (dt == K22_F16) hipLaunchKernelGGL(enc_patchify_kernel<f16_t>, dim3(Bn * P), dim3(256), 0, st, ptr<float>(s_img), ptr<f16_t>(s_patch), S, patch, Kp);
        else hipLaunchKernelGGL(enc_patchify_kernel<float>, dim3(Bn * P), dim3(256), 0, st, ptr<float>(s_img), ptr<float>(s_patch), S, patch, Kp);
        K22_CHECK_LAUNCH();
        return K22_OK;
      });
      {
        IgemmParams p = {};
        p.stages = -1;
        p.M = B * P; p.N = D; p.Npad = D; p.Kc = Kp; p.K0 = Kp; p.taps = 1; p.lda0 = Kp; p.ldo = D; p.ldr = D;
        p.out_mode = IG_OUT_ROWMAJOR_F32; p.act = K22_ACT_NONE; p.splitk = 1; p.Wp = wp; p.bias = nullptr;
        ops.push_back([=](hipStream_t st) {
          IgemmParams q = p;
          q.A0 = ptr(s_patch); q.out = ptr(s_pout); q.partial = ptr<float>(s_splitk);
          return launch_igemm(q, dt, st);
        });
      }
      ops.push_back([=](hipStream_t st) {
        hipLaunchKernelGGL(enc_vision_assemble_kernel, dim3(P + 1, Bn), dim3(256), 0, st, ptr<float>(s_pout), cls, pos, ptr<float>(s_inp), P, D);
        K22_CHECK_LAUNCH();
        return K22_OK;
      });
      op_ln(s_inp, 0, D, M, "ln_pre", s_inp, D, nullptr);
    } else {
      const float* te = Wf("token_embedding"); const float* pe = Wf("positional_embedding");
      const float* ty = xlmr ? Wf("token_type_embedding") : nullptr;
      const int vocab = cfg.vocab, pad = cfg.pad_id, maxp = cfg.max_pos;
      ops.push_back([=](hipStream_t st) {
        hipLaunchKernelGGL(enc_embed_kernel, dim3(n, Bn), dim3(256), 0, st, ptr<int>(s_tok), te, pe, ty, ptr<float>(s_inp), n, D, vocab, xlmr ? 1 : 0, pad, maxp);
        K22_CHECK_LAUNCH();
        return K22_OK;
      });
      if (xlmr) op_ln(s_inp, 0, D, M, "embeddings_ln", s_inp, D, s_ln);
    }
    // ---- transformer -----------------------------------------------------------------------------------------------------------
    const int Tkp = (n + 63) / 64 * 64;
    if (flash) {
      need(s_kall, (size_t)B * heads * Tkp * 64 * esz);
      need(s_vtall, (size_t)B * heads * Tkp * 64 * esz);
    }
    for (int l = 0; l < cfg.layers; ++l) {
      const std::string pfx = "layers." + std::to_string(l);
      if (!xlmr) op_ln(s_inp, 0, D, M, pfx + ".ln_1", nullptr, 0, s_ln);
      op_linear(s_ln, M, 3 * D, D, pfx + ".qkv", K22_ACT_NONE, s_qkv, 3 * D, 0);
      const int causal = kind == K22_ENC_CLIP_TEXT ? 1 : 0;
      if (!flash) {
        const size_t smem = ga_smem;
        const float sc = 1.0f / sqrtf((float)hd);
        ops.push_back([=](hipStream_t st) {
          dim3 grid((n + 15) / 16, heads, Bn);
#define K22_GA(TT_)                                                                                                              \
          {                                                                                                                      \
            static LdsAttrGuard guard;                                                                                           \
            if (int rc_ = k22_ensure_lds_attr(guard, reinterpret_cast<const void*>(&enc_attention_generic_kernel<TT_>), 160 * 1024, __FILE__, __LINE__)) return rc_; \
            hipLaunchKernelGGL(enc_attention_generic_kernel<TT_>, grid, dim3(256), smem, st, ptr<TT_>(s_qkv), ptr<TT_>(s_att), n, D, hd, sc);  \
          }
          if (dt == K22_BF16) K22_GA(bf16_t) else if (dt == K22_F16) K22_GA(f16_t) else K22_GA(float)
#undef K22_GA
          K22_CHECK_LAUNCH();
          return K22_OK;
        });
      } else
      ops.push_back([=](hipStream_t st) {
        KvPackParams kp;
        kp.qkv = ptr(s_qkv); kp.ctxkv = nullptr; kp.kall = ptr(s_kall); kp.vtall = ptr(s_vtall);
        kp.B = Bn; kp.H = heads; kp.T = n; kp.S = 0; kp.Tkp = Tkp;
        int rc = launch_kv_pack(kp, dt, st);
        if (rc) return rc;
        AttentionParams ap = {};
        ap.q = ptr(s_qkv); ap.ldq = 3 * D; ap.kall = ptr(s_kall); ap.vtall = ptr(s_vtall); ap.out = ptr(s_att); ap.ldo = D;
        ap.B = Bn; ap.H = heads; ap.T = n; ap.Tk = n; ap.Tkp = Tkp; ap.scale = 0.125f;
        ap.causal = causal;
        if (xlmr) { ap.key_valid = ptr<float>(s_valid); ap.kv_ld = n; ap.kv_n = n; }
        return launch_attention(ap, dt, st);
      });
      op_linear(s_att, M, D, D, pfx + ".proj", K22_ACT_NONE, s_inp, D, 2);
      if (xlmr) {
        // BertSelfOutput: LayerNorm(dense(attn) + x);  BertIntermediate: GELU(dense);  BertOutput: LayerNorm(dense(h) + x)
        op_ln(s_inp, 0, D, M, pfx + ".ln_1", s_inp, D, s_ln);
        op_linear(s_ln, M, F, D, pfx + ".fc", K22_ACT_GELU, s_fc, F, 0);
        op_linear(s_fc, M, D, F, pfx + ".out", K22_ACT_NONE, s_inp, D, 2);
        op_ln(s_inp, 0, D, M, pfx + ".ln_2", s_inp, D, l + 1 < cfg.layers ? s_ln : nullptr);
      } else {
        op_ln(s_inp, 0, D, M, pfx + ".ln_2", nullptr, 0, s_ln);
        op_linear(s_ln, M, F, D, pfx + ".fc", K22_ACT_NONE, s_fc32, F, 1);
        const int64_t nel = (int64_t)M * F;
        const int exact = cfg.hidden_act == 1 ? 1 : 0;
        ops.push_back([=](hipStream_t st) {
          const int nb = (int)((nel + 255) / 256 < 4096 ? (nel + 255) / 256 : 4096);
          if (dt == K22_BF16) hipLaunchKernelGGL(enc_quickgelu_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, ptr<float>(s_fc32), ptr<bf16_t>(s_fc), nel, exact);
          else if (dt == K22_F16) hipLaunchKernelGGL(enc_quickgelu_kernel<f16_t>, dim3(nb), dim3(256), 0, st, ptr<float>(s_fc32), ptr<f16_t>(s_fc), nel, exact);
          else hipLaunchKernelGGL(enc_quickgelu_kernel<float>, dim3(nb), dim3(256), 0, st, ptr<float>(s_fc32), ptr<float>(s_fc), nel, exact);
          K22_CHECK_LAUNCH();
          return K22_OK;
        });
        op_linear(s_fc, M, D, F, pfx + ".out", K22_ACT_NONE, s_inp, D, 2);
      }
    }
    // ---- heads ---------------------------------------------------------------------------------------------------------------
    const void* hw = W_("head.weight");
    const float* hb = xlmr ? Wf("head.bias") : nullptr;
    if (kind == K22_ENC_CLIP_TEXT) {
      op_ln(s_inp, 0, D, M, "ln_final", s_seq, D, nullptr);
      ops.push_back([=](hipStream_t st) {
        hipLaunchKernelGGL(enc_gather_eot_kernel, dim3(Bn), dim3(256), 0, st, ptr<int>(s_tok), ptr<float>(s_seq), ptr<float>(s_pool_in), n, D);
        K22_CHECK_LAUNCH();
        return K22_OK;
      });
    } else if (vision) {
      op_ln(s_inp, 0, (int64_t)n * D, B, "ln_post", s_pool_in, D, nullptr);      // class-token rows only
    } else {
      ops.push_back([=](hipStream_t st) {
        hipError_t e = hipMemcpyAsync(ptr(s_seq), ptr(s_inp), (size_t)M * D * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
        hipLaunchKernelGGL(enc_masked_mean_kernel, dim3(Bn), dim3(256), 0, st, ptr<float>(s_inp), ptr<float>(s_valid), ptr<float>(s_pool_in), n, D);
        K22_CHECK_LAUNCH();
        return K22_OK;
      });
    }
    ops.push_back([=](hipStream_t st) {
      LinearSmallParams lp = {};
      lp.x = ptr<float>(s_pool_in); lp.ldx = D; lp.W = hw; lp.bias = hb; lp.out = ptr<float>(s_pooled); lp.ldo = od;
      lp.M = Bn; lp.N = od; lp.K = D; lp.act_in = K22_ACT_NONE; lp.act_out = K22_ACT_NONE;
      return launch_linear_smallm(lp, K22_F32, st);
    });
    if (!err.empty()) return k22_set_error(K22_EINVAL, err.c_str());
    size_t off = 0;
    for (auto& s : slots) { s.off = off; off += (s.bytes + 255) / 256 * 256; }
    ws_bytes = off + 256;
    return K22_OK;
  }
};

extern "C" {

int k22_encoder_create(const K22EncoderConfig* cfg, const K22Weight* weights, int n_weights, K22Encoder** out) {
  if (!cfg || !out || (!weights && n_weights > 0)) return k22_set_error(K22_EINVAL, "encoder_create: null argument");
  if (!k22_dtype_ok(cfg->dtype)) return k22_set_error(K22_EINVAL, "encoder_create: dtype");
  K22Encoder* m = new K22Encoder();
  m->cfg = *cfg; m->dtype = cfg->dtype; m->esz = cfg->dtype == K22_F32 ? 4 : 2;
  for (int i = 0; i < n_weights; ++i) m->w[weights[i].name] = weights[i].ptr;
  {
    const char* e = getenv("K22_AUTOTUNE");
    m->autotune = e ? (atoi(e) != 0) : 1;
  }
  *out = m;
  return K22_OK;
}
void k22_encoder_destroy(K22Encoder* m) { delete m; }

int k22_encoder_plan(K22Encoder* m, int B, size_t* workspace_bytes) {
  if (!m || !workspace_bytes) return k22_set_error(K22_EINVAL, "encoder_plan: null argument");
  int rc = m->plan(B);
  if (rc) { if (!m->err.empty()) { m->ops.clear(); m->ws = nullptr; } return rc; }   // a missing weight is found after the old plan was dropped
  *workspace_bytes = m->ws_bytes;
  return K22_OK;
}
int k22_encoder_bind(K22Encoder* m, void* workspace, size_t workspace_bytes) {
  if (!m || !workspace) return k22_set_error(K22_EINVAL, "encoder_bind: null argument");
  if (m->ops.empty()) return k22_set_error(K22_EINVAL, "encoder_bind: plan first");
  if (workspace_bytes < m->ws_bytes) return k22_set_error(K22_ENOMEM, "encoder_bind: workspace too small");
  if ((uintptr_t)workspace % 256) return k22_set_error(K22_EINVAL, "encoder_bind: workspace must be 256-byte aligned");
  m->ws = reinterpret_cast<char*>(workspace);
  if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
  return K22_OK;
}

int k22_encoder_forward(K22Encoder* m, const int* tokens, const float* key_valid, const float* image, float* seq_out, float* pooled_out,
                        void* stream) {
  if (!m || !m->ws) return k22_set_error(K22_EINVAL, "encoder_forward: bind a workspace first");
  const K22EncoderConfig& c = m->cfg;
  const bool vision = c.kind == K22_ENC_CLIP_VISION, xlmr = c.kind == K22_ENC_XLMR;
  if (vision ? !image : !tokens) return k22_set_error(K22_EINVAL, "encoder_forward: tokens (text towers) / image (vision tower) is null");
  if (xlmr && !key_valid) return k22_set_error(K22_EINVAL, "encoder_forward: the XLM-R tower needs the attention mask");
  if (vision && seq_out) return k22_set_error(K22_EINVAL, "encoder_forward: the vision tower has no sequence output");
  if (!pooled_out) return k22_set_error(K22_EINVAL, "encoder_forward: pooled_out is null");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e;
#define K22_CPY(dst, src, bytes)                                                   \
  e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);                \
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  const size_t M = (size_t)m->B * c.n_ctx;
  if (vision) { K22_CPY(m->ptr(m->s_img), image, (size_t)m->B * 3 * c.image_size * c.image_size * 4); }
  else { K22_CPY(m->ptr(m->s_tok), tokens, M * 4); }
  if (xlmr) { K22_CPY(m->ptr(m->s_valid), key_valid, M * 4); }
  if (m->autotune && !m->tuned_done) {
    // problems the tile table does not know are measured here (the in-place residual GEMMs accumulate garbage into the sequence
    // buffer meanwhile: the real pass below rebuilds it from the inputs)
    int rc = tune_igemm_ops(m->tuned, m->dtype, m->s_flush->bytes ? m->ptr(m->s_flush) : nullptr, m->s_flush->bytes, st);
    if (rc) return rc;
    m->tuned_done = true;
  }
  if (!m->graph_exec) {
    // first pass on this plan: run eagerly once (function attributes, code load), then capture the launch list
    for (auto& op : m->ops) { int rc = op(st); if (rc) return rc; }
    if (!m->cap_stream) {
      e = hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    }
    hipGraph_t g = nullptr;
    e = hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    int rc = K22_OK;
    for (auto& op : m->ops) { rc = op(m->cap_stream); if (rc) break; }
    e = hipStreamEndCapture(m->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    e = hipGraphInstantiate(&m->graph_exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { m->graph_exec = nullptr; return k22_set_error_hip(e, __FILE__, __LINE__); }
  }
  e = hipGraphLaunch(m->graph_exec, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  if (seq_out) { K22_CPY(seq_out, m->ptr(m->s_seq), M * c.width * 4); }
  K22_CPY(pooled_out, m->ptr(m->s_pooled), (size_t)m->B * c.out_dim * 4);
#undef K22_CPY
  return K22_OK;
}

}  // extern "C"
