// k22 — diffusion-prior transformer engine + its sampler step.
//
// Replaces PriorTransformer.forward (kandinsky2/model/prior.py:226-270) with its blocks (LayerNorm :48-54,
// MultiheadAttention / QKVMultiheadAttention :57-102, MLP :74-83, ResidualAttentionBlock :105-127) and the per-step
// arithmetic of PriorDiffusionModel.forward's sampling loop (:336-384: guided_model_fn CFG with per-sample scales,
// START_X mean, FIXED_SMALL variance, denoised_fn clamp, gaussian_diffusion.py:223-322, 352-382).
//
// MI355X mapping: the transformer is weight-streaming bound (2.0 GB of bf16 weights per forward for 162 token rows at
// bs = 1), so every Linear is one launch_igemm over the [N][K] weight exactly as the reference stores it, the fp32
// residual stream is updated in place by the GEMM epilogue (res_f32), LayerNorm writes the GEMM's T operand, and
// the 81-token attention runs one workgroup per (batch, head) out of LDS.  c_qkv rows are packed as Q | K | V planes
// (the reference keeps per-head [q|k|v] interleaved, prior.py:93-95).
#include "kernels.h"
#include "elementwise.h"
#include "../../include/k22.h"
#include "tuning.h"
#include "skinny.h"

#include <algorithm>
#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

// ---- LayerNorm over the last dim: fp32 rows (stride ldx) -> T or fp32 rows -----------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void prior_layernorm_kernel(const float* x, int64_t ldx, const float* g, const float* bta,
                                                              TO* y, int64_t ldy, int D, float eps) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)row * ldx;
  float s = 0.f;
  for (int i = tid; i < D; i += 256) s += xr[i];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / D;
  float q = 0.f;
  for (int i = tid; i < D; i += 256) { const float d = xr[i] - mean; q += d * d; }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / D + eps);
  for (int i = tid; i < D; i += 256) y[(int64_t)row * ldy + i] = from_f32<TO>((xr[i] - mean) * rstd * g[i] + bta[i]);
}

// ---- input sequence: rows 0..n_text-1 / n_text / n_text+1 / n_text+2 were written by the projections; add the
// positional embedding everywhere and put prd_emb in the last row (prior.py:248-256) -----------------------------
__global__ void prior_finish_input_kernel(float* inp, const float* pos, const float* prd, int B, int n_ctx, int D) {
  const int64_t total = (int64_t)B * n_ctx * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D), t = (int)((i / D) % n_ctx);
    const float base = (t == n_ctx - 1) ? prd[d] : inp[i];
    inp[i] = base + pos[(int64_t)t * D + d];
  }
}

// ---- one ancestral step of the prior (START_X mean, FIXED_SMALL variance) with classifier-free guidance -------------
// x, model_out, noise, x_out: [2*bs][D] with halves [cond | uncond]; scales [bs]; tab = (coef1, coef2, log_var, nonzero)
__global__ void prior_sampler_step_kernel(const float* x, const float* model_out, const float* noise, const float* scales,
                                          const float* tab, float clamp, float* x_out, int bs, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * bs * D) return;
  const int d = i % D, n = i / D, j = n % bs;
  const float c = model_out[(int64_t)j * D + d], u = model_out[(int64_t)(j + bs) * D + d];
  float x0 = u + scales[j] * (c - u);
  x0 = fminf(fmaxf(x0, -clamp), clamp);
  const float mean = __fadd_rn(__fmul_rn(tab[0], x0), __fmul_rn(tab[1], x[i]));
  x_out[i] = mean + tab[3] * expf(0.5f * tab[2]) * noise[i];
}

namespace {
struct PSlot { size_t bytes = 0, off = 0; };
typedef std::function<int(hipStream_t)> POp;
}  // namespace

struct K22Prior {
  K22PriorConfig cfg;
  int dtype; size_t esz;
  std::unordered_map<std::string, const void*> w;
  int B = 0;
  std::deque<PSlot> slots;
  std::vector<POp> ops;
  std::deque<Tuned> tuned;   // every transformer Linear: tile configuration picked by measurement at the first forward
  bool tuned_done = false;
  int autotune = 1;
  hipGraphExec_t graph_exec = nullptr;   // the ~250 launches of one forward, replayed as one graph
  hipStream_t cap_stream = nullptr;
  ~K22Prior() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
  }
  size_t ws_bytes = 0;
  char* ws = nullptr;
  std::string err;
  PSlot *s_x, *s_t, *s_temb, *s_te1, *s_txtemb, *s_txtenc, *s_txtencT, *s_valid, *s_mask, *s_inp, *s_ln, *s_qkv, *s_att, *s_fc,
      *s_lnlast, *s_out, *s_splitk, *s_flush, *s_kall, *s_vtall;
  // skinny path (round 6; skinny.hip): 16-bit engines run the transformer through the fragment-major weight-streaming GEMM, the fused
  // split-K finish + LayerNorm and the small-T attention: 7 launches per layer instead of 12.  K22_PRIOR_SKINNY=0 keeps the old path.
  bool skinny = false;
  bool wfrag_done = false;
  struct WFrag { const void* src; PSlot* dst; int Npad, K; };
  std::vector<WFrag> wfrags;   // fragment-major copies of the transformer weights, written into the workspace once per bind

  PSlot* new_slot(size_t bytes = 0) { slots.emplace_back(); slots.back().bytes = bytes; return &slots.back(); }
  static void need(PSlot* s, size_t bytes) { if (bytes > s->bytes) s->bytes = bytes; }
  template <typename T = char> T* ptr(const PSlot* s) const { return reinterpret_cast<T*>(ws + s->off); }
  const void* W_(const std::string& name) {
    auto it = w.find(name);
    if (it == w.end()) { if (err.empty()) err = "missing weight: " + name; return nullptr; }
    return it->second;
  }
  const float* Wf(const std::string& name) { return reinterpret_cast<const float*>(W_(name)); }

  // out (+)= A[M][K] . W[N][K]^T + bias; A is T; out T, or fp32 with an fp32 residual (in-place residual stream)
  void op_linear(PSlot* a, size_t a_off, int M, int N, int K, const std::string& pfx, int act, PSlot* dst, size_t dst_off, int ldo,
                 bool f32_out_residual) {
    tuned.emplace_back();
    Tuned* t = &tuned.back();
    IgemmParams& p = t->p;
    p.stages = -1;
    p.M = M; p.N = N; p.Npad = (N + 63) / 64 * 64; p.Kc = K; p.K0 = K; p.taps = 1; p.lda0 = K; p.ldo = ldo; p.ldr = ldo;
    p.out_mode = f32_out_residual ? IG_OUT_ROWMAJOR_F32 : IG_OUT_ROWMAJOR; p.act = act; p.res_f32 = f32_out_residual ? 1 : 0;
    p.Wp = W_(pfx + ".weight"); p.bias = Wf(pfx + ".bias");
    tuned_make_candidates(*t, dtype);
    tuned_default_cfg(*t, dtype);
    need(s_splitk, tuned_max_splitk_bytes(*t, autotune != 0));
    const int dt = dtype;
    t->run = [=](hipStream_t st) {
      IgemmParams q = t->p;
      tuned_apply_cfg(q, t->cfg);
      q.A0 = ptr(a) + a_off; q.out = ptr(dst) + dst_off; q.partial = ptr<float>(s_splitk);
      q.residual = f32_out_residual ? (ptr(dst) + dst_off) : nullptr;
      return launch_igemm(q, dt, st);
    };
    ops.push_back([=](hipStream_t st) { return t->run(st); });
  }
  void op_ln(PSlot* x, size_t x_off, int64_t ldx, int rows, const std::string& pfx, PSlot* y, bool to_f32) {
    const float* g = Wf(pfx + ".weight"); const float* b = Wf(pfx + ".bias");
    const int D = cfg.xf_width, dt = dtype;
    ops.push_back([=](hipStream_t st) {
      const float* xp = reinterpret_cast<const float*>(ptr(x) + x_off);
      if (to_f32) hipLaunchKernelGGL(prior_layernorm_kernel<float>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, ptr<float>(y), (int64_t)D, D, 1e-5f);
      else if (dt == K22_BF16) hipLaunchKernelGGL(prior_layernorm_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, ptr<bf16_t>(y), (int64_t)D, D, 1e-5f);
      else if (dt == K22_F16) hipLaunchKernelGGL(prior_layernorm_kernel<f16_t>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, ptr<f16_t>(y), (int64_t)D, D, 1e-5f);
      else hipLaunchKernelGGL(prior_layernorm_kernel<float>, dim3(rows), dim3(256), 0, st, xp, ldx, g, b, ptr<float>(y), (int64_t)D, D, 1e-5f);
      K22_CHECK_LAUNCH();
      return K22_OK;
    });
  }

  // fragment-major copy of a transformer weight [Npad][K] (slot in the workspace, repacked at the first forward after a bind)
  PSlot* wfrag_of(const std::string& name, int N, int K) {
    const int Npad = (N + 63) / 64 * 64;
    PSlot* s = new_slot((size_t)Npad * K * esz);
    wfrags.push_back(WFrag{W_(name), s, Npad, K});
    return s;
  }
  // out = act(A . W^T + bias) through the skinny kernel; epi: SkinnyEpi; partial launches leave [splitk][M][N] in s_splitk
  void op_skinny(PSlot* a, int M, int N, int K, const std::string& pfx, int act, int epi, int splitk, PSlot* dst, int ldo) {
    PSlot* wf = wfrag_of(pfx + ".weight", N, K);
    const float* bias = Wf(pfx + ".bias");
    if (epi == SK_EPI_PARTIAL) need(s_splitk, (size_t)splitk * M * N * 4);
    const int dt = dtype;
    ops.push_back([=](hipStream_t st) {
      SkinnyParams q = {};
      q.Af = ptr(a); q.Wf = ptr(wf); q.bias = epi == SK_EPI_PARTIAL ? nullptr : bias; q.out = dst ? ptr(dst) : nullptr; q.partial = ptr<float>(s_splitk);
      q.M = M; q.N = N; q.Npad = (N + 63) / 64 * 64; q.K = K; q.MA = (M + 31) / 32; q.splitk = splitk; q.epi = epi; q.act = act; q.ldo = ldo;
      return launch_skinny(q, dt, 0, 0, st);
    });
  }
  // x (+)= bias + sum of the split-K partials in s_splitk (splitk > 0); then LayerNorm `ln` of x into s_ln in A-fragment order (ln non-empty)
  void op_finish_ln(int M, int N, int splitk, const std::string& bias_name, const std::string& ln) {
    const float* bias = bias_name.empty() ? nullptr : Wf(bias_name);
    const float* g = ln.empty() ? nullptr : Wf(ln + ".weight");
    const float* bb = ln.empty() ? nullptr : Wf(ln + ".bias");
    const int dt = dtype;
    ops.push_back([=](hipStream_t st) {
      FinishLnParams q = {};
      q.partial = splitk > 0 ? ptr<float>(s_splitk) : nullptr; q.splitk = splitk; q.bias = bias; q.x = ptr<float>(s_inp); q.ldx = N;
      q.g = g; q.b = bb; q.yfrag = ptr(s_ln); q.M = M; q.N = N; q.MA = (M + 31) / 32; q.eps = 1e-5f;
      return launch_finish_ln(q, dt, st);
    });
  }

  int plan(int nB) {
    B = nB;
    slots.clear(); ops.clear(); err.clear(); ws = nullptr; tuned.clear(); tuned_done = false; wfrags.clear(); wfrag_done = false;
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    const int D = cfg.xf_width, nt = cfg.text_ctx, nc = nt + 4, cd = cfg.clip_dim, cw = cfg.clip_xf_width, M = B * nc;
    if (B < 1 || B > 8) return k22_set_error(K22_EINVAL, "prior: batch (2*bs) must be in 1..8 per engine call");
    if (D % 64 || D / cfg.xf_heads != 64) return k22_set_error(K22_EINVAL, "prior: 64 channels per head");
    s_x = new_slot((size_t)B * cd * 4); s_t = new_slot((size_t)B * 4 + 64);
    s_temb = new_slot((size_t)B * D * 4); s_te1 = new_slot((size_t)B * D * 4);
    s_txtemb = new_slot((size_t)B * cd * 4); s_txtenc = new_slot((size_t)B * nt * cw * 4); s_txtencT = new_slot((size_t)B * nt * cw * esz);
    s_valid = new_slot((size_t)B * nt * 4); s_mask = new_slot((size_t)B * nc * nc * 4);
    {
      const char* e = getenv("K22_PRIOR_SKINNY");
      skinny = (!e || atoi(e) != 0) && esz == 2 && D <= 2048 && nc <= 128;
    }
    const size_t Mp = (size_t)(M + 31) / 32 * 32;   // the fragment-major tensors hold whole 32-row atoms
    s_inp = new_slot((size_t)M * D * 4); s_ln = new_slot(Mp * D * esz); s_qkv = new_slot((size_t)M * 3 * D * esz);
    s_att = new_slot(Mp * D * esz); s_fc = new_slot(Mp * 4 * D * esz);
    s_lnlast = new_slot((size_t)B * D * 4); s_out = new_slot((size_t)B * cd * 4); s_splitk = new_slot(256);
    s_flush = new_slot(autotune ? ((size_t)320 << 20) : 0);
    s_kall = new_slot(); s_vtall = new_slot();
    const int Bn = B, dt = dtype;
    const size_t es = esz;

    // ---- input sequence (prior.py:238-256) ----------------------------------------------------------------
    {
      const float* freqs = Wf("time_freqs");
      const float* w0 = Wf("time_embed.0.weight"); const float* b0 = Wf("time_embed.0.bias");
      const float* w2 = Wf("time_embed.2.weight"); const float* b2 = Wf("time_embed.2.bias");
      const float* wte = Wf("text_emb_proj.weight"); const float* bte = Wf("text_emb_proj.bias");
      const float* wci = Wf("clip_img_proj.weight"); const float* bci = Wf("clip_img_proj.bias");
      ops.push_back([=](hipStream_t st) {
        int rc = launch_timestep_embedding(ptr<float>(s_t), freqs, ptr<float>(s_temb), Bn, D / 2, st);
        if (rc) return rc;
        LinearSmallParams lp = {};
        lp.x = ptr<float>(s_temb); lp.ldx = D; lp.W = w0; lp.bias = b0; lp.out = ptr<float>(s_te1); lp.ldo = D;
        lp.M = Bn; lp.N = D; lp.K = D; lp.act_in = K22_ACT_NONE; lp.act_out = K22_ACT_SILU;
        rc = launch_linear_smallm(lp, K22_F32, st);
        if (rc) return rc;
        float* inp = ptr<float>(s_inp);
        lp.x = ptr<float>(s_te1); lp.W = w2; lp.bias = b2; lp.out = inp + (size_t)(nt + 1) * D; lp.ldo = (int64_t)nc * D; lp.act_out = K22_ACT_NONE;
        rc = launch_linear_smallm(lp, K22_F32, st);                                   // t_emb -> row n_text+1
        if (rc) return rc;
        lp.x = ptr<float>(s_txtemb); lp.ldx = cd; lp.K = cd; lp.W = wte; lp.bias = bte; lp.out = inp + (size_t)nt * D;
        rc = launch_linear_smallm(lp, K22_F32, st);                                   // text_emb -> row n_text
        if (rc) return rc;
        lp.x = ptr<float>(s_x); lp.W = wci; lp.bias = bci; lp.out = inp + (size_t)(nt + 2) * D;
        rc = launch_linear_smallm(lp, K22_F32, st);                                   // clip_img_proj(x) -> row n_text+2
        if (rc) return rc;
        return launch_cast_rows(ptr<float>(s_txtenc), ptr(s_txtencT), Bn * nt, cw, cw, cw, dt, st);
      });
      // text_enc_proj: one GEMM per batch element straight into rows 0..n_text-1 of the sequence (fp32 out)
      for (int b = 0; b < B; ++b) {
        IgemmParams p = {};
        p.stages = -1;
        p.M = nt; p.N = D; p.Npad = D; p.Kc = cw; p.K0 = cw; p.taps = 1; p.lda0 = cw; p.ldo = D; p.ldr = D;
        p.out_mode = IG_OUT_ROWMAJOR_F32; p.act = K22_ACT_NONE; p.splitk = 1;
        p.Wp = W_("text_enc_proj.weight"); p.bias = Wf("text_enc_proj.bias");
        ops.push_back([=](hipStream_t st) {
          IgemmParams q = p;
          q.A0 = ptr(s_txtencT) + (size_t)b * nt * cw * es; q.out = ptr<float>(s_inp) + (size_t)b * nc * D;
          return launch_igemm(q, dt, st);
        });
      }
      const float* pos = Wf("positional_embedding"); const float* prd = Wf("prd_emb");
      ops.push_back([=](hipStream_t st) {
        hipLaunchKernelGGL(prior_finish_input_kernel, dim3(256), dim3(256), 0, st, ptr<float>(s_inp), pos, prd, Bn, nc, D);
        K22_CHECK_LAUNCH();
        return K22_OK;
      });
    }
    // ---- transformer (prior.py:105-155) ------------------------------------------------------------------
    if (skinny) {
      // per layer: c_qkv -> attention -> c_proj (split-K partials) -> [finish + residual + ln_2] -> c_fc (+ GELU) -> mlp.c_proj (partials)
      // -> [finish + residual + next ln_1].  Split-K only where a finish launch runs anyway (N = D: 64-wide n-tiles alone would leave
      // three quarters of the CUs idle).
      const int heads = cfg.xf_heads;
      const int sk_proj = std::max(1, std::min(4, D / 512)), sk_fc2 = std::max(1, std::min(4, 4 * D / 512));
      static const int sk_qkv_env = [] { const char* e = getenv("K22_PRIOR_QKV_SPLIT"); return e ? atoi(e) : 1; }();   // measured: 1 (profiles/r06_skinny.txt: split 2 of c_qkv 22.4 us against 18.4)
      const int sk_qkv = std::max(1, std::min(std::min(4, sk_qkv_env), D / 512));
      op_finish_ln(M, D, 0, "", "transformer.resblocks.0.ln_1");
      for (int l = 0; l < cfg.xf_layers; ++l) {
        const std::string pfx = "transformer.resblocks." + std::to_string(l);
        // c_qkv: split-K partials whose finish (bias, one rounding) rides on the attention's staging loads (sk_qkv = 1: row-major T output)
        const float* qkv_bias = Wf(pfx + ".attn.c_qkv.bias");
        if (sk_qkv > 1) op_skinny(s_ln, M, 3 * D, D, pfx + ".attn.c_qkv", K22_ACT_NONE, SK_EPI_PARTIAL, sk_qkv, nullptr, 3 * D);
        else op_skinny(s_ln, M, 3 * D, D, pfx + ".attn.c_qkv", K22_ACT_NONE, SK_EPI_ROWMAJOR, 1, s_qkv, 3 * D);
        ops.push_back([=](hipStream_t st) {
          SmallAttnParams ap = {};
          ap.qkv = ptr(s_qkv); ap.ldq = 3 * D;
          if (sk_qkv > 1) { ap.part = ptr<float>(s_splitk); ap.nsplit = sk_qkv; ap.bias = qkv_bias; } ap.out = ptr(s_att); ap.ldo = D; ap.out_frag = 1; ap.MA = (M + 31) / 32;
          ap.B = Bn; ap.H = heads; ap.T = nc; ap.scale = 0.125f; ap.causal = 1; ap.key_valid = ptr<float>(s_valid); ap.kv_ld = nt; ap.kv_n = nt;
          return launch_small_attention(ap, dt, st);
        });
        op_skinny(s_att, M, D, D, pfx + ".attn.c_proj", K22_ACT_NONE, SK_EPI_PARTIAL, sk_proj, nullptr, D);
        op_finish_ln(M, D, sk_proj, pfx + ".attn.c_proj.bias", pfx + ".ln_2");
        op_skinny(s_ln, M, 4 * D, D, pfx + ".mlp.c_fc", K22_ACT_GELU, SK_EPI_AFRAG, 1, s_fc, 4 * D);
        op_skinny(s_fc, M, D, 4 * D, pfx + ".mlp.c_proj", K22_ACT_NONE, SK_EPI_PARTIAL, sk_fc2, nullptr, D);
        op_finish_ln(M, D, sk_fc2, pfx + ".mlp.c_proj.bias", l + 1 < cfg.xf_layers ? "transformer.resblocks." + std::to_string(l + 1) + ".ln_1" : std::string());
      }
    } else
    for (int l = 0; l < cfg.xf_layers; ++l) {
      const std::string pfx = "transformer.resblocks." + std::to_string(l);
      op_ln(s_inp, 0, D, M, pfx + ".ln_1", s_ln, false);
      op_linear(s_ln, 0, M, 3 * D, D, pfx + ".attn.c_qkv", K22_ACT_NONE, s_qkv, 0, 3 * D, false);
      // attention: the UNet's flash kernel (attention.hip) with the prior's mask (prior.py:262-263) — the additive
      // mask is -inf exactly for padding keys and for keys after the query, the 4 extra positions are always valid
      const int heads = cfg.xf_heads, Tkp = (nc + 63) / 64 * 64;
      need(s_kall, (size_t)B * heads * Tkp * 64 * esz);
      need(s_vtall, (size_t)B * heads * Tkp * 64 * esz);
      ops.push_back([=](hipStream_t st) {
        KvPackParams kp;
        kp.qkv = ptr(s_qkv); kp.ctxkv = nullptr; kp.kall = ptr(s_kall); kp.vtall = ptr(s_vtall);
        kp.B = Bn; kp.H = heads; kp.T = nc; kp.S = 0; kp.Tkp = Tkp;
        int rc = launch_kv_pack(kp, dt, st);
        if (rc) return rc;
        AttentionParams ap = {};
        ap.q = ptr(s_qkv); ap.ldq = 3 * D; ap.kall = ptr(s_kall); ap.vtall = ptr(s_vtall); ap.out = ptr(s_att); ap.ldo = D;
        ap.B = Bn; ap.H = heads; ap.T = nc; ap.Tk = nc; ap.Tkp = Tkp; ap.scale = 0.125f;
        ap.causal = 1; ap.key_valid = ptr<float>(s_valid); ap.kv_ld = nt; ap.kv_n = nt;
        return launch_attention(ap, dt, st);
      });
      op_linear(s_att, 0, M, D, D, pfx + ".attn.c_proj", K22_ACT_NONE, s_inp, 0, D, true);
      op_ln(s_inp, 0, D, M, pfx + ".ln_2", s_ln, false);
      op_linear(s_ln, 0, M, 4 * D, D, pfx + ".mlp.c_fc", K22_ACT_GELU, s_fc, 0, 4 * D, false);
      op_linear(s_fc, 0, M, D, 4 * D, pfx + ".mlp.c_proj", K22_ACT_NONE, s_inp, 0, D, true);
    }
    // ---- final_ln on the last token + out_proj (prior.py:266-269) -------------------------------------------
    if (cfg.xf_final_ln) {
      op_ln(s_inp, (size_t)(nc - 1) * D * 4, (int64_t)nc * D, B, "final_ln", s_lnlast, true);
    } else {
      ops.push_back([=](hipStream_t st) {
        hipError_t e = hipMemcpy2DAsync(ptr(s_lnlast), (size_t)D * 4, ptr(s_inp) + (size_t)(nc - 1) * D * 4, (size_t)nc * D * 4, (size_t)D * 4, Bn,
                                        hipMemcpyDeviceToDevice, st);
        return e == hipSuccess ? K22_OK : k22_set_error_hip(e, __FILE__, __LINE__);
      });
    }
    {
      const float* wo = Wf("out_proj.weight"); const float* bo = Wf("out_proj.bias");
      ops.push_back([=](hipStream_t st) {
        LinearSmallParams lp = {};
        lp.x = ptr<float>(s_lnlast); lp.ldx = D; lp.W = wo; lp.bias = bo; lp.out = ptr<float>(s_out); lp.ldo = cd;
        lp.M = Bn; lp.N = cd; lp.K = D; lp.act_in = K22_ACT_NONE; lp.act_out = K22_ACT_NONE;
        return launch_linear_smallm(lp, K22_F32, st);
      });
    }
    if (!err.empty()) return k22_set_error(K22_EINVAL, err.c_str());
    size_t off = 0;
    for (auto& s : slots) { s.off = off; off += (s.bytes + 255) / 256 * 256; }
    ws_bytes = off + 256;
    return K22_OK;
  }
};

extern "C" {

int k22_prior_create(const K22PriorConfig* cfg, const K22Weight* weights, int n_weights, K22Prior** out) {
  if (!cfg || !out) return k22_set_error(K22_EINVAL, "prior_create: null argument");
  if (!k22_dtype_ok(cfg->dtype)) return k22_set_error(K22_EINVAL, "prior_create: dtype");
  K22Prior* m = new K22Prior();
  m->cfg = *cfg; m->dtype = cfg->dtype; m->esz = cfg->dtype == K22_F32 ? 4 : 2;
  for (int i = 0; i < n_weights; ++i) m->w[weights[i].name] = weights[i].ptr;
  {
    const char* e = getenv("K22_AUTOTUNE");
    m->autotune = e ? (atoi(e) != 0) : 1;
  }
  *out = m;
  return K22_OK;
}
int k22_prior_tuning_report(const K22Prior* m, char* buf, size_t cap) {
  if (!m || !buf || cap == 0) return k22_set_error(K22_EINVAL, "prior_tuning_report: null argument");
  snprintf(buf, cap, "%s", tuning_report_text(m->tuned).c_str());
  return K22_OK;
}
void k22_prior_destroy(K22Prior* m) { delete m; }

int k22_prior_plan(K22Prior* m, int B, size_t* workspace_bytes) {
  if (!m || !workspace_bytes) return k22_set_error(K22_EINVAL, "prior_plan: null argument");
  int rc = m->plan(B);
  if (rc) return rc;
  *workspace_bytes = m->ws_bytes;
  return K22_OK;
}
int k22_prior_bind(K22Prior* m, void* workspace, size_t workspace_bytes) {
  if (!m || !workspace) return k22_set_error(K22_EINVAL, "prior_bind: null argument");
  if (m->ops.empty()) return k22_set_error(K22_EINVAL, "prior_bind: plan first");
  if (workspace_bytes < m->ws_bytes) return k22_set_error(K22_ENOMEM, "prior_bind: workspace too small");
  if ((uintptr_t)workspace % 256) return k22_set_error(K22_EINVAL, "prior_bind: workspace must be 256-byte aligned");
  m->ws = reinterpret_cast<char*>(workspace);
  m->wfrag_done = false;
  if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
  return K22_OK;
}
int k22_prior_forward(K22Prior* m, const float* x, const float* timesteps, const float* text_emb, const float* text_enc,
                      const float* key_valid, float* out, void* stream) {
  if (!m || !m->ws) return k22_set_error(K22_EINVAL, "prior_forward: bind a workspace first");
  if (!x || !timesteps || !text_emb || !text_enc || !key_valid || !out) return k22_set_error(K22_EINVAL, "prior_forward: null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const K22PriorConfig& c = m->cfg;
  hipError_t e;
#define K22_CPY(dst, src, bytes)                                                   \
  e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);                \
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  K22_CPY(m->ptr(m->s_x), x, (size_t)m->B * c.clip_dim * 4);
  K22_CPY(m->ptr(m->s_t), timesteps, (size_t)m->B * 4);
  K22_CPY(m->ptr(m->s_txtemb), text_emb, (size_t)m->B * c.clip_dim * 4);
  K22_CPY(m->ptr(m->s_txtenc), text_enc, (size_t)m->B * c.text_ctx * c.clip_xf_width * 4);
  K22_CPY(m->ptr(m->s_valid), key_valid, (size_t)m->B * c.text_ctx * 4);
  if (m->skinny && !m->wfrag_done) {
    for (auto& wf : m->wfrags) {
      int rc = launch_stream_repack(wf.src, m->ptr(wf.dst), wf.Npad, 1, wf.K, m->dtype, st);
      if (rc) return rc;
    }
    m->wfrag_done = true;
  }
  if (m->autotune && !m->tuned_done) {
    // the in-place residual GEMMs make the tuning runs accumulate garbage into the sequence buffer: harmless, the
    // real forward below rebuilds it from the inputs
    int rc = tune_igemm_ops(m->tuned, m->dtype, m->s_flush->bytes ? m->ptr(m->s_flush) : nullptr, m->s_flush->bytes, st);
    if (rc) return rc;
    m->tuned_done = true;
  }
  if (!m->graph_exec) {
    // first forward: run eagerly once (function attributes, code load), then capture the launch list
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
  K22_CPY(out, m->ptr(m->s_out), (size_t)m->B * c.clip_dim * 4);
#undef K22_CPY
  return K22_OK;
}

int k22_prior_sampler_step(const float* x, const float* model_out, const float* noise, const float* scales, const float* table_row,
                           float clamp, float* x_out, int bs, int D, void* stream) {
  if (!x || !model_out || !noise || !scales || !table_row || !x_out || bs < 1 || D < 1) return k22_set_error(K22_EINVAL, "prior_sampler_step: bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(prior_sampler_step_kernel, dim3((2 * bs * D + 255) / 256), dim3(256), 0, st, x, model_out, noise, scales, table_row, clamp, x_out, bs, D);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

}  // extern "C"
