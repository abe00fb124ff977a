// k22 — UNet engine: builds the launch list of one Text2ImUNet forward from the reference's
// hyper-parameters and replays it (eagerly or as a captured hipGraph) on a caller-owned workspace.
//
// Structure mirrors UNetModel.__init__/forward (kandinsky2/model/unet.py:371-611) and
// Text2ImUNet.forward/get_text_emb (kandinsky2/model/text2im_model2_1.py:57-103):
//   emb = time_embed(timestep_embedding(t)) + xf_proj ; 16 input blocks ; middle ; 16 output blocks
//   (skip concat) ; out = GN+SiLU+conv3x3.
// MI355X mapping: activations live as NHWC in the workspace; every 3x3 conv input is written by the
// GroupNorm-apply kernel with a zero border; skip concats are never materialised (GroupNorm and the
// 1x1 skip GEMM take two base pointers); all 36 emb_layers run as ONE skinny GEMV per step;
// encoder_kv is hoisted into set_condition (its input, the cached xf_out, is step-invariant).
#include "kernels.h"
#include "elementwise.h"
#include "../../include/k22.h"
#include "tuning.h"

#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int launch_step_advance(int* step, int delta, hipStream_t s);

namespace {

struct Slot { size_t bytes = 0, off = 0; };

enum OpKind { OP_CONV3 = 0, OP_GEMM = 1, OP_GN = 2, OP_ATTN = 3, OP_MISC = 4, OP_NKINDS = 5 };

struct Op {
  std::function<int(hipStream_t)> fn;
  int kind = OP_MISC;
  double flops = 0.0;   // algorithmic FLOPs (2*M*N*K) for MFMA ops
  double flops2 = 0.0;  // FLOPs of a fused 1x1 skip connection riding on a conv op
  double bytes = 0.0;   // algorithmic HBM bytes (read + write) for HBM-bound ops
  int kernels = 1;      // kernel launches issued by fn
  Op() {}
  template <typename F> Op(F f, int k = OP_MISC, double fl = 0.0, double by = 0.0, int nk = 1) : fn(f), kind(k), flops(fl), bytes(by), kernels(nk) {}
  int operator()(hipStream_t st) const { return fn(st); }
};
typedef std::vector<Op> OpList;

struct Act {  // unpadded NHWC activation, optionally a virtual channel concat of two tensors
  Slot* s0 = nullptr; int C0 = 0;
  Slot* s1 = nullptr; int C1 = 0;
  int H = 0, W = 0;
  // GroupNorm partial sums delivered by the producing convolution (null = none: run gn_stats_kernel)
  const Tuned* p0 = nullptr; Slot* st0 = nullptr;
  const Tuned* p1 = nullptr; Slot* st1 = nullptr;
  int C() const { return C0 + C1; }
};

}  // namespace

struct K22UNet {
  K22UNetConfig cfg;
  int dtype;   // arithmetic type of the MFMA kernels (K22DType; K22_F16X3 = split precision)
  int sdt;     // type of everything the data-movement kernels touch: dtype, except fp32 for the split-precision arithmetic
  size_t esz;
  std::unordered_map<std::string, const void*> w;

  // ---- plan state ----
  int B = 0, H = 0, W = 0;
  std::deque<Slot> slots;
  OpList ops;       // one UNet forward
  OpList cond_ops;  // conditioning head
  OpList hint_ops;  // 2.2 ControlNet-depth: input_hint_block over the hint image (once per generation)
  bool hint_set = false;
  std::deque<Tuned> tuned;  // stable addresses: op closures and Act descriptors point into it
  bool tuned_done = false;
  int autotune = 1;
  int fuse_skip = 1;
  // GroupNorm-apply (+FiLM, +SiLU, zero border) fused into the consuming 3x3 convolution's halo fill (conv3_halo_spec_kernel producers)
  // wherever the convolution's tile configuration is one of the specialised kernels; the stand-alone gn_apply stays for the rest.
  // OFF by default (K22_FUSE_GN=1 turns it on): parity-green and bit-identical to the two-kernel path, but measured SLOWER on MI355X -
  // bf16 C2 step 143.5 -> 112.9 steps/s (conv class 4.30 -> 6.48 ms for 0.26 ms of gn_apply removed), f16x3 66.5 -> 59.4: the matrix
  // pipe's owner shares its SIMD's issue port with the producer wave, every n-tile of an m-tile re-normalises the same pixels (6-12x
  // the work of one stand-alone pass), and the rewrite pins each halo piece to land within ONE tap (profiles/r04_gn_fused_negative.txt).
  int fuse_gn = 0;
  struct GnLink { const Tuned* consumer = nullptr; };   // GroupNorm op -> the convolution that reads its output (set after both exist)
  std::deque<GnLink> gn_links;
  // weight-streaming kernel (stream_gemm.hip) for the small-M 3x3 convolutions: fragment-major copies of their weights live in the
  // workspace (one more copy of those weights: ~1.2 GB of the 2.1 UNet in bf16), written once per bind before the first forward
  int stream_frag = 1;
  struct FragJob { const void* W; Slot* slot; int Npad, taps, Kc; };
  std::vector<FragJob> frag_jobs;
  bool frag_done = false;
  int repack_frags(hipStream_t st) {
    for (auto& j : frag_jobs) {
      int rc = launch_stream_repack(j.W, ptr(j.slot), j.Npad, j.taps, j.Kc, dtype, st);
      if (rc) return rc;
    }
    frag_done = true;
    return K22_OK;
  }
  size_t ws_bytes = 0;
  char* ws = nullptr;
  bool cond_set = false;
  bool warmed = false;   // one eager pass of the op list has run on this plan (function attributes set, code loaded): capture may start
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t cap_stream = nullptr;  // private stream used only to CAPTURE (the caller's may be the legacy default stream)
  // the whole denoising loop as ONE graph (k22_unet_sample_loop): valid for exactly the buffers / scalars it was captured with
  hipGraphExec_t loop_exec = nullptr;
  std::vector<unsigned long long> loop_key;
  std::string err;

  // persistent slots
  Slot *s_xin, *s_img, *s_mask, *s_t, *s_out;
  Slot *s_temb, *s_e1, *s_emb, *s_film, *s_xfproj, *s_ctx;
  Slot *s_full, *s_pool, *s_imgemb, *s_tmpf, *s_tmpf2, *s_fullT;
  Slot *s_part, *s_coeff, *s_P1, *s_U1, *s_P2, *s_S, *s_N, *s_QKV, *s_KALL, *s_VT, *s_ATT, *s_splitk, *s_flush;
  Slot *s_U1st;
  Slot *s_ctxf = nullptr, *s_hint = nullptr, *s_hintin = nullptr, *s_hbuf[2] = {nullptr, nullptr};
  Slot* s_h[3];
  Slot* s_hst[3];
  std::vector<Slot*> s_ctxkv;  // one per attention block
  int n_attn = 0;
  int64_t film_total = 0;
  // k22_unet_sample_loop: the time embedding and ALL FiLM vectors of every step of the loop are computed by batched launches before the first
  // step (the timesteps of a loop are known up front and nothing else feeds them): rows = steps x B, up to 8 rows per launch, so the 231 MB
  // emb_layers weight stream is read once per 8 / B steps instead of once per step.  Every output row is the same arithmetic as in the per-step
  // launch (one accumulator per row, same order): the loop stays bit-identical to the stepwise calls (asserted by the GPU tests).
  float* film_all = nullptr;        // [rows][film_total] | temb [rows][mc] | e1 [rows][ted] | emb [rows][ted]
  size_t film_all_bytes = 0;
  int hoist_time = 1;               // K22_HOIST_TIME=0 (measurement only): the loop keeps the per-step time / FiLM launches
  const float* film_cur = nullptr;  // != null while a hoisted loop runs: this step's [B][film_total]
  const float* film_base() const { return film_cur ? film_cur : ptr<float>(s_film); }
  struct TimeW { const float* freqs; const float* w0; const float* b0; const float* w2; const float* b2; const void* we; const float* be; int mc, ted, dt; } tw = {};
  // rows of (timestep -> sinusoid -> time_embed MLP + conditioning row (m % B) -> SiLU -> emb_layers of every ResBlock): nn.py:101-121, unet.py:159-170
  int time_rows(const float* t, float* temb, float* e1, float* emb, float* film, int rows, hipStream_t st) {
    int rc = launch_timestep_embedding(t, tw.freqs, temb, rows, tw.mc / 2, st);
    if (rc) return rc;
    LinearSmallParams lp = {};
    lp.x = temb; lp.ldx = tw.mc; lp.W = tw.w0; lp.bias = tw.b0; lp.out = e1; lp.ldo = tw.ted;
    lp.M = rows; lp.N = tw.ted; lp.K = tw.mc; lp.act_in = K22_ACT_NONE; lp.act_out = K22_ACT_SILU;
    rc = launch_linear_smallm(lp, K22_F32, st);
    if (rc) return rc;
    lp.x = e1; lp.ldx = tw.ted; lp.W = tw.w2; lp.bias = tw.b2; lp.add = ptr<float>(s_xfproj); lp.ld_add = tw.ted; lp.add_mod = B;
    lp.out = emb; lp.K = tw.ted; lp.act_out = K22_ACT_NONE;
    rc = launch_linear_smallm(lp, K22_F32, st);
    if (rc) return rc;
    LinearSmallParams le = {};
    le.x = emb; le.ldx = tw.ted; le.W = tw.we; le.bias = tw.be; le.out = film; le.ldo = film_total;
    le.M = rows; le.N = (int)film_total; le.K = tw.ted; le.act_in = K22_ACT_SILU; le.act_out = K22_ACT_NONE;
    return launch_linear_smallm(le, tw.dt, st);
  }

  Slot* new_slot(size_t bytes = 0) { slots.emplace_back(); slots.back().bytes = bytes; return &slots.back(); }
  static void need(Slot* s, size_t bytes) { if (bytes > s->bytes) s->bytes = bytes; }
  template <typename T = char> T* ptr(const Slot* s) const { return reinterpret_cast<T*>(ws + s->off); }

  const void* W_(const std::string& name) {
    auto it = w.find(name);
    if (it == w.end()) { if (err.empty()) err = "missing weight: " + name; return nullptr; }
    return it->second;
  }
  const float* Wf(const std::string& name) { return reinterpret_cast<const float*>(W_(name)); }

  // (Round 4's "two half-batch chains" mode - the CFG pair as two engines on two streams, +3.9 % - is gone: it rested on a work-around for
  // a kernel pair whose co-resident victim returned wrong elements.  Round 5 narrowed that to the victim's packed-fp32 instructions
  // (profiles/r05_two_stream_probe.txt; `make NOPK=1` builds the library without them): INTEGRATION.md G.)
  // the [B][out_channels][HW] model output
  float* model_out() { return ptr<float>(s_out); }
  int exec(hipStream_t st) { return run_ops(st); }
  int exec_eager(hipStream_t st) { const int rc = exec(st); if (rc == K22_OK) warmed = true; return rc; }
  // first forward of a plan / binding: fragment-major weight copies, tile configurations the table does not know
  int prepare_run(hipStream_t st) {
    if (!frag_done) { int rc = repack_frags(st); if (rc) return rc; }
    if (autotune && !tuned_done) {
      int rc = tune_all(st);
      if (rc) return rc;
      tuned_done = true;
    }
    return K22_OK;
  }
  void drop_graphs() {
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (loop_exec) { (void)hipGraphExecDestroy(loop_exec); loop_exec = nullptr; }
  }

  ~K22UNet() {
    if (loop_exec) (void)hipGraphExecDestroy(loop_exec);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    if (film_all) (void)hipFree(film_all);
  }

  // One forward on `st`: the op list in order.  (A variant that forked the time-embedding / FiLM GEMV onto a second stream,
  // i.e. a parallel branch of the captured graph joined by the first FiLM consumer, was measured on one box, alternating
  // runs: 116.9 / 117.2 steps/s with the branch against 119.4 / 119.1 without - the 231 MB weight stream slows the
  // convolutions it runs beside by more than the 80 us it hides.  Removed.)
  int run_ops_range(hipStream_t st, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi && i < ops.size(); ++i) { int rc = ops[i](st); if (rc) return rc; }
    return K22_OK;
  }
  int run_ops(hipStream_t st) { return run_ops_range(st, 0, ops.size()); }
  int run_ops_eager(hipStream_t st) { const int rc = run_ops(st); if (rc == K22_OK) warmed = true; return rc; }

  // ------------------------------------------------------------------------------------------
  void op_gn(OpList& L, const Act& in, const std::string& pfx, int64_t film_off,
             int act, int mode, int pad, Slot* dst, const GnLink* link = nullptr) {
    const int Bn = B, C = in.C(), HW = in.H * in.W;
    const int nsplit = gn_nsplit(Bn, HW);
    // producer-side partial sums for every part of the input?  Otherwise the stand-alone stats pass.
    const bool fused = in.st0 != nullptr && (in.s1 == nullptr || in.st1 != nullptr);
    if (!fused) need(s_part, (size_t)Bn * nsplit * C * 2 * sizeof(float));
    need(s_coeff, (size_t)Bn * C * 2 * sizeof(float));
    const int Ho = mode == 1 ? in.H / 2 : (mode == 2 ? in.H * 2 : in.H);
    const int Wo = mode == 1 ? in.W / 2 : (mode == 2 ? in.W * 2 : in.W);
    need(dst, (size_t)Bn * (Ho + 2 * pad) * (Wo + 2 * pad) * C * esz);
    const float* gamma = Wf(pfx + ".weight");
    const float* beta = Wf(pfx + ".bias");
    const Act a = in;
    const int dt = sdt;
    const int x3 = k22_is_split(dtype) ? 1 : 0;   // every GroupNorm output feeds a convolution or the qkv GEMM: written in x3 chunks
    const double gn_bytes = (double)Bn * HW * C * esz * (fused ? 1.0 : 2.0) + (double)Bn * (Ho + 2 * pad) * (Wo + 2 * pad) * C * esz;
    L.push_back(Op([=](hipStream_t st) {
      const void* x0 = ptr(a.s0);
      const void* x1 = a.s1 ? ptr(a.s1) : nullptr;
      GnCoeffParams cp = {};
      if (fused) {
        cp.src[0].st = ptr<float>(a.st0); cp.src[0].rpi = a.p0->rpi; cp.src[0].C = a.C0;
        if (a.s1) { cp.src[1].st = ptr<float>(a.st1); cp.src[1].rpi = a.p1->rpi; cp.src[1].C = a.C1; }
        if (cp.src[0].rpi <= 0 || (a.s1 && cp.src[1].rpi <= 0)) return k22_set_error(K22_EINVAL, "unet: producer delivered no GroupNorm partial sums");
      } else {
        GnStatsParams sp;
        sp.x0 = x0; sp.x1 = x1; sp.C0 = a.C0; sp.C1 = a.C1;
        sp.HW = HW; sp.B = Bn; sp.groups = 32; sp.nsplit = nsplit; sp.partial = ptr<float>(s_part);
        int rc = launch_gn_stats(sp, dt, st);
        if (rc) return rc;
        cp.src[0].st = ptr<float>(s_part); cp.src[0].rpi = nsplit; cp.src[0].C = C;
      }
      cp.HW = HW; cp.C = C; cp.groups = 32; cp.eps = 1e-5f;
      cp.gamma = gamma; cp.beta = beta;
      cp.film = film_off >= 0 ? film_base() + film_off : nullptr; cp.film_ld = film_total;
      cp.coeff = ptr<float>(s_coeff);
      GnApplyParams ap = {};
      ap.x0 = x0; ap.x1 = x1; ap.C0 = a.C0; ap.C1 = a.C1; ap.B = Bn; ap.H = a.H; ap.W = a.W;
      ap.mode = mode; ap.pad = pad; ap.act = act; ap.coeff = ptr<float>(s_coeff); ap.out = ptr(dst); ap.out_x3 = x3;
      int rc = launch_gn_coeff(cp, Bn, st);
      if (rc) return rc;
      // the consuming convolution applies the coefficients itself while it fills its LDS halo (fused GroupNorm-apply): nothing to write
      if (link != nullptr && link->consumer != nullptr && conv3_algo_fuses_gn(link->consumer->cfg.algo)) return K22_OK;
      return launch_gn_apply(ap, dt, st);
    }, OP_GN, 0.0, gn_bytes, fused ? 2 : 3));
  }

  static void apply_cfg(IgemmParams& q, const Cfg& c) { tuned_apply_cfg(q, c); }
  int op_dt(const Tuned& t) const { return t.dt >= 0 ? t.dt : dtype; }
  void make_candidates(Tuned& t) { tuned_make_candidates(t, op_dt(t)); }
  void default_cfg(Tuned& t) { tuned_default_cfg(t, op_dt(t)); }
  void finish_cfg(Tuned& t) { tuned_finish_cfg(t, op_dt(t)); }
  size_t max_splitk_bytes(const Tuned& t) const { return tuned_max_splitk_bytes(t, autotune != 0); }
  int max_rpi(const Tuned& t) const { return tuned_max_rpi(t, op_dt(t), autotune != 0); }

  // ---- precision plan of the asymmetric split engine (K22_F16X2; round 5) -----------------------------------------------------------
  // Which MFMA ops take the activation operand at fp16 precision (two MFMAs per product) and which keep the full split (three) is a
  // property of the PLAN, read off the operand-rounding ablation of the oracle (oracle/drift_ablation.py -> tests/golden/
  // drift_ablation_x2.json; share of the first-forward error variance of "every activation operand at fp16"):
  //   1x1 skip_connection inputs (the un-normalised residual stream)           51 %   -> always three MFMAs (3 % of the FLOPs)
  //   the out head's 3x3 convolution (384 -> 8 channels)                       14 %   -> always three MFMAs (0.03 % of the FLOPs)
  //   3x3 convolutions of the top level: in_layers 12 %, out_layers 14 %              -> x2_plan bits 0 / 1 (40 % of the FLOPs)
  //   3x3 convolutions one level down 6 %, two 0.7 %, three 0.08 %; qkv 0.02 %        -> two MFMAs
  //   attention operands (q, k, v, P)                                          1.5 %  -> ONE MFMA (fp16 tiles, attention_kernel<xh_t>)
  //   proj_out, encoder_kv, to_model_dim_n (small; inputs not normalised)             -> three MFMAs
  // x2_plan: bit 0 = the top level's in_layers convolutions run x2, bit 1 = its out_layers convolutions run x2 (K22_X2_PLAN).  Default 0:
  // measured at C2 on MI355X (profiles/r05_x2_plans.txt; same box, x3 = 59.5 steps/s at 3.6e-6): plan 0 69.4 steps/s at 2.7e-4 max-abs /
  // 5.4e-5 rms, plan 1 71.9 at 5.3e-4 / 8.5e-5, plan 2 71.3 at 4.6e-4 / 9.3e-5, plan 3 73.9 at 5.8e-4 / 1.14e-4 - the rms values are the
  // ablation's predictions to 4 %; the last 6 % of speed cost the margin to the 5e-4 bar.
  int x2_plan = 0;
  enum ConvRole { CONV_IN = 0, CONV_OUT = 1, CONV_HEAD = 2 };
  int conv_dt(int role, int Hc) const {
    if (dtype != K22_F16X2) return dtype;
    if (role == CONV_HEAD) return K22_F16X3;
    if (Hc >= H) return (x2_plan >> role) & 1 ? K22_F16X2 : K22_F16X3;   // top level (the latent's own resolution)
    return K22_F16X2;
  }
  // GEMMs: only the qkv projection (its input is a GroupNorm output) runs x2
  int gemm_dt(bool qkv) const { return dtype == K22_F16X2 ? (qkv ? K22_F16X2 : K22_F16X3) : dtype; }

  // conv3x3 over a zero-bordered slot `src` [B][Hc+2][Wc+2][Cin].  `stats` (optional) receives the GroupNorm
  // partial sums of the output; returns the launch descriptor (null when the output is not a tunable T tensor).
  // gn_in (optional): the raw tensor(s) whose GroupNorm (coefficients in s_coeff, written by the op_gn just before) `src` holds: with a
  // specialised-kernel configuration the convolution reads gn_in and applies the coefficients in its halo fill instead of reading `src`
  Tuned* op_conv(OpList& L, int role, Slot* src, int Hc, int Wc, int Cin, int Cout,
                 const std::string& pfx, const Act* residual, Slot* dst, int out_mode, Slot* stats = nullptr,
                 const Act* skip_in = nullptr, const std::string& skip_pfx = std::string(), const Act* gn_in = nullptr, int gn_act = K22_ACT_NONE) {
    tuned.emplace_back();
    Tuned* t = &tuned.back();
    t->dt = conv_dt(role, Hc);
    IgemmParams& p = t->p;
    p.stages = -1;
    p.M = B * Hc * Wc; p.N = Cout; p.Npad = (Cout + 63) / 64 * 64; p.Kc = Cin; p.K0 = Cin; p.taps = 9;
    p.H = Hc; p.W = Wc; p.ldo = Cout; p.ldr = Cout; p.out_mode = out_mode; p.act = K22_ACT_NONE;
    p.Wp = W_(pfx + ".weight"); p.bias = Wf(pfx + ".bias");
    Act sk;
    if (skip_in) {
      // fused 1x1 skip_connection: S0 is only a non-null marker here, the device pointers are set at launch
      sk = *skip_in;
      p.S0 = reinterpret_cast<const void*>(1); p.S1 = sk.s1 ? reinterpret_cast<const void*>(1) : nullptr;
      p.SK0 = sk.C0; p.SK1 = sk.C1;
      p.Ws = W_(skip_pfx + ".weight"); p.bias2 = Wf(skip_pfx + ".bias");
    }
    t->want_stats = stats != nullptr;
    // small M (the 24x24 / 12x12 levels): the weight-streaming kernel becomes a candidate, fed from a fragment-major copy of the weights
    Slot *wf = nullptr, *wsf = nullptr;
    if (stream_frag && k22_esz(dtype) == 2 && p.M <= 1152 && (stream_supported(p, dtype, 5) || stream_supported(p, dtype, 9))) {
      wf = new_slot();
      need(wf, stream_frag_bytes(p.Npad, 9, Cin, dtype));
      frag_jobs.push_back({p.Wp, wf, p.Npad, 9, Cin});
      p.Wfrag = reinterpret_cast<const void*>(1);          // marker for the candidate list; the device pointer is set at launch
      if (skip_in) {
        wsf = new_slot();
        need(wsf, stream_frag_bytes(p.Npad, 1, p.SK0 + p.SK1, dtype));
        frag_jobs.push_back({p.Ws, wsf, p.Npad, 1, p.SK0 + p.SK1});
        p.Wsfrag = reinterpret_cast<const void*>(1);
      }
    }
    make_candidates(*t);
    default_cfg(*t);
    // The fragment-major copy costs one more copy of this layer's weights in every plan's workspace and a repack per bind:
    // keep it only where the weight-streaming kernel can actually run - the tile table (or the fixed heuristic, autotune off) names it
    // for this problem, or the problem is unknown and will be measured with it as a candidate.
    if (wf && t->cfg.algo != 20 && (t->from_table || !autotune)) {
      wf->bytes = 0; wf = nullptr;
      if (wsf) { wsf->bytes = 0; wsf = nullptr; frag_jobs.pop_back(); }
      frag_jobs.pop_back();
      p.Wfrag = nullptr; p.Wsfrag = nullptr;
      make_candidates(*t);
      default_cfg(*t);
    }
    need(s_splitk, max_splitk_bytes(*t));
    if (t->want_stats) need(stats, (size_t)B * max_rpi(*t) * Cout * 2 * sizeof(float));
    need(dst, out_mode == IG_OUT_ROWMAJOR ? (size_t)p.M * Cout * esz : (size_t)p.M * Cout * sizeof(float));
    Slot* rs = residual ? residual->s0 : nullptr;
    const int dt = t->dt;
    const bool gnf = gn_in != nullptr && fuse_gn;
    Act ga;
    if (gnf) ga = *gn_in;
    t->run = [=](hipStream_t st) {
      IgemmParams q = t->p;
      apply_cfg(q, t->cfg);
      q.A0 = ptr(src); q.residual = rs ? ptr(rs) : nullptr; q.out = ptr(dst); q.partial = ptr<float>(s_splitk);
      if (gnf && conv3_algo_fuses_gn(t->cfg.algo)) {
        q.gn_coeff = ptr<float>(s_coeff); q.gn_x0 = ptr(ga.s0); q.gn_x1 = ga.s1 ? ptr(ga.s1) : nullptr; q.gn_C0 = ga.C0; q.gn_act = gn_act;
      }
      q.stats = t->want_stats ? ptr<float>(stats) : nullptr;
      if (q.S0) { q.S0 = ptr(sk.s0); q.S1 = sk.s1 ? ptr(sk.s1) : nullptr; }
      q.Wfrag = wf ? ptr(wf) : nullptr; q.Wsfrag = wsf ? ptr(wsf) : nullptr;
      return launch_igemm(q, dt, st);
    };
    const double fl = 2.0 * p.M * (double)p.N * 9.0 * p.Kc;
    L.push_back(Op([=](hipStream_t st) { return t->run(st); }, OP_CONV3, fl, 0.0, 1));
    if (skip_in) L.back().flops2 = 2.0 * p.M * (double)p.N * (p.SK0 + p.SK1);  // counted in the GEMM class
    return t;
  }

  // GEMM over unpadded rows (1x1 conv / linear); `in` may be a virtual concat.
  // `stats` (optional) receives the GroupNorm partial sums of the output (image-shaped inputs only: in.H * in.W rows
  // per image); whether a configuration that can deliver them exists is reported by the returned want_stats.
  // a_raw: `in` holds plain T rows that no producer of ours wrote in x3 chunks (split-precision arithmetic only; ignored otherwise)
  Tuned* op_gemm(OpList& L, const Act& in, int M, int N, const std::string& pfx,
                 const Act* residual, Slot* dst, int ldo = 0, int out_mode = IG_OUT_ROWMAJOR, Slot* stats = nullptr, bool a_raw = false) {
    tuned.emplace_back();
    Tuned* t = &tuned.back();
    t->dt = gemm_dt(out_mode == IG_OUT_QKV);
    IgemmParams& p = t->p;
    p.stages = -1;
    p.M = M; p.N = N; p.Npad = (N + 63) / 64 * 64; p.Kc = in.C(); p.K0 = in.C0; p.taps = 1;
    p.lda0 = in.C0; p.lda1 = in.C1; p.ldo = ldo ? ldo : N; p.ldr = N; p.out_mode = out_mode;
    p.act = K22_ACT_NONE;
    if (in.H > 0 && in.W > 0 && M % (in.H * in.W) == 0 && M / (in.H * in.W) == B) { p.H = in.H; p.W = in.W; }  // rows per image
    if (out_mode == IG_OUT_QKV) p.att_T = in.H * in.W;
    p.Wp = W_(pfx + ".weight"); p.bias = Wf(pfx + ".bias");
    p.a_raw = (a_raw && k22_is_split(dtype)) ? 1 : 0;
    t->want_stats = stats != nullptr && p.H > 0;
    make_candidates(*t);
    if (t->want_stats && t->cands.empty()) { t->want_stats = false; make_candidates(*t); }
    default_cfg(*t);
    need(s_splitk, max_splitk_bytes(*t));
    if (t->want_stats) need(stats, (size_t)B * max_rpi(*t) * N * 2 * sizeof(float));
    need(dst, (size_t)M * p.ldo * esz);
    const Act a = in;
    Slot* rs = residual ? residual->s0 : nullptr;
    const int dt = t->dt;
    t->run = [=](hipStream_t st) {
      IgemmParams q = t->p;
      apply_cfg(q, t->cfg);
      q.A0 = ptr(a.s0); q.A1 = a.s1 ? ptr(a.s1) : nullptr;
      q.residual = rs ? ptr(rs) : nullptr; q.out = ptr(dst); q.partial = ptr<float>(s_splitk);
      if (t->aux0) { q.kall = ptr(reinterpret_cast<Slot*>(t->aux0)); q.vtall = ptr(reinterpret_cast<Slot*>(t->aux1)); }
      q.stats = t->want_stats ? ptr<float>(stats) : nullptr;
      return launch_igemm(q, dt, st);
    };
    L.push_back(Op([=](hipStream_t st) { return t->run(st); }, OP_GEMM, 2.0 * p.M * (double)p.N * p.Kc, 0.0, 1));
    return t;
  }

  int tune_all(hipStream_t st) { return tune_igemm_ops(tuned, dtype, s_flush->bytes ? ptr(s_flush) : nullptr, s_flush->bytes, st); }

  // ResBlock (unet.py:110-220) with use_scale_shift_norm=True; updown: 0 none, 1 down, 2 up.
  Act resblock(const std::string& pfx, const Act& in, int Cout, int updown, int64_t& film_cursor, Slot* dst, Slot* dst_stats) {
    const int Cin = in.C();
    const int Ho = updown == 1 ? in.H / 2 : (updown == 2 ? in.H * 2 : in.H);
    const int Wo = updown == 1 ? in.W / 2 : (updown == 2 ? in.W * 2 : in.W);
    // in_layers: GN + SiLU (+ resample) -> conv3x3
    GnLink* l1 = nullptr;
    if (fuse_gn && updown == 0) { gn_links.emplace_back(); l1 = &gn_links.back(); }   // (the resampling GroupNorms keep gn_apply)
    op_gn(ops, in, pfx + ".in_layers.0", -1, K22_ACT_SILU, updown, 1, s_P1, l1);
    Tuned* t1 = op_conv(ops, CONV_IN, s_P1, Ho, Wo, Cin, Cout, pfx + ".in_layers.2", nullptr, s_U1, IG_OUT_ROWMAJOR, s_U1st, nullptr, std::string(),
                        l1 ? &in : nullptr, K22_ACT_SILU);
    if (l1) l1->consumer = t1;
    // out_layers: GN * (1+scale) + shift -> SiLU -> conv3x3 (+ skip)
    Act u1; u1.s0 = s_U1; u1.C0 = Cout; u1.H = Ho; u1.W = Wo;
    if (t1->want_stats) { u1.p0 = t1; u1.st0 = s_U1st; }
    const int64_t film_off = film_cursor;
    film_cursor += 2 * Cout;
    GnLink* l2 = nullptr;
    if (fuse_gn) { gn_links.emplace_back(); l2 = &gn_links.back(); }
    op_gn(ops, u1, pfx + ".out_layers.0", film_off, K22_ACT_SILU, 0, 1, s_P2, l2);
    Act skip;
    skip.H = Ho; skip.W = Wo; skip.C0 = Cout;
    if (updown) {
      if (Cin != Cout || in.s1) { if (err.empty()) err = "updown ResBlock with channel change is not supported"; }
      need(s_S, (size_t)B * Ho * Wo * Cin * esz);
      const Act a = in; const int Bn = B, dt = sdt;
      ops.push_back([=](hipStream_t st) { return launch_resample(ptr(a.s0), ptr(s_S), Bn, a.H, a.W, Cin, updown, dt, st); });
      skip.s0 = s_S;
    } else if (Cin != Cout) {
      // 1x1 skip_connection: fused into the second conv (halo kernel) when it applies, else its own GEMM
      IgemmParams probe = {};
      probe.M = B * Ho * Wo; probe.N = Cout; probe.Npad = (Cout + 63) / 64 * 64; probe.Kc = Cout; probe.K0 = Cout; probe.taps = 9;
      probe.H = Ho; probe.W = Wo; probe.ldo = Cout; probe.ldr = Cout; probe.out_mode = IG_OUT_ROWMAJOR;
      probe.S0 = reinterpret_cast<const void*>(1); probe.S1 = in.s1 ? reinterpret_cast<const void*>(1) : nullptr;
      probe.SK0 = in.C0; probe.SK1 = in.C1; probe.Ws = reinterpret_cast<const void*>(1);
      if (fuse_skip && Cout >= 128 && (conv3_halo_supported(probe, dtype, 256) || conv3_halo_supported(probe, dtype, 128))) {
        Tuned* t2 = op_conv(ops, CONV_OUT, s_P2, Ho, Wo, Cout, Cout, pfx + ".out_layers.3", nullptr, dst, IG_OUT_ROWMAJOR, dst_stats,
                            &in, pfx + ".skip_connection", l2 ? &u1 : nullptr, K22_ACT_SILU);
        if (l2) l2->consumer = t2;
        Act out; out.s0 = dst; out.C0 = Cout; out.H = Ho; out.W = Wo;
        if (t2->want_stats) { out.p0 = t2; out.st0 = dst_stats; }
        return out;
      }
      op_gemm(ops, in, B * Ho * Wo, Cout, pfx + ".skip_connection", nullptr, s_S, 0, IG_OUT_ROWMAJOR, nullptr, /*a_raw=*/true);
      skip.s0 = s_S;
    } else {
      if (in.s1) { if (err.empty()) err = "identity skip over a concat input"; }
      skip.s0 = in.s0;
    }
    Tuned* t2 = op_conv(ops, CONV_OUT, s_P2, Ho, Wo, Cout, Cout, pfx + ".out_layers.3", &skip, dst, IG_OUT_ROWMAJOR, dst_stats, nullptr, std::string(),
                        l2 ? &u1 : nullptr, K22_ACT_SILU);
    if (l2) l2->consumer = t2;
    Act out; out.s0 = dst; out.C0 = Cout; out.H = Ho; out.W = Wo;
    if (t2->want_stats) { out.p0 = t2; out.st0 = dst_stats; }
    return out;
  }

  // AttentionBlock (unet.py:223-269) + QKVAttention (:272-340).  The qkv projection writes q row-major and k / v
  // straight into this block's attention operands (K_all, V^T_all) behind the context keys, which were projected
  // (encoder_kv) and packed once per conditioning: no per-step packing kernel.
  Act attnblock(const std::string& pfx, const Act& in, Slot* dst, Slot* dst_stats = nullptr) {
    const int C = in.C0, T = in.H * in.W, Hh = C / 64, S = cfg.ctx_len;
    const int Tk = S + T, Tkp = (Tk + 63) / 64 * 64;
    n_attn++;
    op_gn(ops, in, pfx + ".norm", -1, K22_ACT_NONE, 0, 0, s_N);
    Act n; n.s0 = s_N; n.C0 = C; n.H = in.H; n.W = in.W;
    Slot* kall = new_slot((size_t)B * Hh * Tkp * 64 * esz);
    Slot* vtall = new_slot((size_t)B * Hh * Tkp * 64 * esz);
    // encoder_kv (step-invariant) + context part of K_all / V^T_all (and zero padding) -> cond_ops
    Slot* ckv = new_slot((size_t)B * S * 2 * C * esz);
    s_ctxkv.push_back(ckv);
    const int Bn = B, dt = dtype, sd = sdt;
    {
      Act c; c.s0 = s_ctx; c.C0 = cfg.ctx_dim; c.H = 1; c.W = S;
      op_gemm(cond_ops, c, B * S, 2 * C, pfx + ".encoder_kv", nullptr, ckv, 0, IG_OUT_ROWMAJOR, nullptr, /*a_raw=*/true);
      cond_ops.push_back(Op([=](hipStream_t st) {
        KvPackParams kp;
        kp.qkv = nullptr; kp.ctxkv = ptr(ckv); kp.kall = ptr(kall); kp.vtall = ptr(vtall);
        kp.B = Bn; kp.H = Hh; kp.T = 0; kp.S = S; kp.Tkp = Tkp;
        return launch_kv_pack(kp, sd, st);
      }));
    }
    // qkv projection in IG_OUT_QKV mode
    need(s_QKV, (size_t)B * T * C * esz);  // q only
    {
      Tuned* t = op_gemm(ops, n, B * T, 3 * C, pfx + ".qkv", nullptr, s_QKV, C, IG_OUT_QKV);
      t->p.att_T = T; t->p.att_S = S; t->p.att_Tkp = Tkp;
      t->aux0 = kall; t->aux1 = vtall;
    }
    need(s_ATT, (size_t)B * T * C * esz);
    ops.push_back(Op([=](hipStream_t st) {
      AttentionParams ap = {};
      ap.q = ptr(s_QKV); ap.ldq = C; ap.kall = ptr(kall); ap.vtall = ptr(vtall); ap.out = ptr(s_ATT); ap.ldo = C;
      ap.B = Bn; ap.H = Hh; ap.T = T; ap.Tk = Tk; ap.Tkp = Tkp; ap.scale = 0.125f;
      ap.out_x3 = k22_is_split(dt) ? 1 : 0;   // proj_out reads it as an x3-chunk operand
      return launch_attention(ap, dt, st);
    }, OP_ATTN, 4.0 * Bn * Hh * (double)T * Tk * 64.0, 0.0, 1));
    Act a; a.s0 = s_ATT; a.C0 = C; a.H = in.H; a.W = in.W;
    // proj_out + residual; its epilogue also delivers the GroupNorm partial sums the next ResBlock needs
    Tuned* tp = op_gemm(ops, a, B * T, C, pfx + ".proj_out", &in, dst, 0, IG_OUT_ROWMAJOR, dst_stats);
    Act out; out.s0 = dst; out.C0 = C; out.H = in.H; out.W = in.W;
    if (tp->want_stats) { out.p0 = tp; out.st0 = dst_stats; }
    return out;
  }

  bool has_attn(int ds) const {
    for (int i = 0; i < cfg.n_attention_ds; ++i) if (cfg.attention_ds[i] == ds) return true;
    return false;
  }

  int plan(int nB, int nH, int nW) {
    // validate before touching the current plan: a rejected shape leaves the engine usable at the old one
    const int n_down = cfg.n_levels - 1;
    if (nB < 1 || nH < 1 || nW < 1) return k22_set_error(K22_EINVAL, "unet: B, H, W must be positive");
    if (nH % (1 << n_down) || nW % (1 << n_down)) return k22_set_error(K22_EINVAL, "unet: H, W must be divisible by 2^(levels-1)");
    if (nB > 8) return k22_set_error(K22_EINVAL, "unet: batch (2*bs) must be <= 8 per engine call");
    B = nB; H = nH; W = nW;
    gn_links.clear();
    slots.clear(); ops.clear(); cond_ops.clear(); hint_ops.clear(); s_ctxkv.clear(); frag_jobs.clear(); frag_done = false; n_attn = 0; err.clear();
    tuned.clear(); tuned_done = false; warmed = false;
    ws = nullptr; cond_set = false; hint_set = false;
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (loop_exec) { (void)hipGraphExecDestroy(loop_exec); loop_exec = nullptr; }
    const int mc = cfg.model_channels, ted = 4 * mc;

    // total FiLM width = sum over ResBlocks of 2*Cout, in module order
    film_total = 0;
    {
      int ch = mc * cfg.channel_mult[0];
      for (int l = 0; l < cfg.n_levels; ++l) {
        for (int i = 0; i < cfg.num_res_blocks; ++i) { ch = mc * cfg.channel_mult[l]; film_total += 2 * ch; }
        if (l != cfg.n_levels - 1) film_total += 2 * ch;
      }
      film_total += 2 * 2 * ch;  // middle
      for (int l = cfg.n_levels - 1; l >= 0; --l)
        for (int i = 0; i <= cfg.num_res_blocks; ++i) {
          ch = mc * cfg.channel_mult[l];
          film_total += 2 * ch;
          if (l && i == cfg.num_res_blocks) film_total += 2 * ch;
        }
    }

    s_xin = new_slot((size_t)B * 4 * H * W * 4); s_img = new_slot((size_t)B * 4 * H * W * 4);
    s_mask = new_slot((size_t)B * H * W * 4); s_t = new_slot((size_t)B * 4 + 64);
    s_out = new_slot((size_t)B * cfg.out_channels * H * W * 4);
    s_temb = new_slot((size_t)B * mc * 4); s_e1 = new_slot((size_t)B * ted * 4); s_emb = new_slot((size_t)B * ted * 4);
    s_film = new_slot((size_t)B * film_total * 4); s_xfproj = new_slot((size_t)B * ted * 4);
    s_ctx = new_slot((size_t)B * cfg.ctx_len * cfg.ctx_dim * esz);
    const int ntext_p = cfg.ctx_len - cfg.n_image_embs;   // text tokens (0 for the 2.2 head)
    s_full = new_slot((size_t)B * (ntext_p > 0 ? ntext_p : 1) * cfg.text_dim1 * 4); s_pool = new_slot((size_t)B * cfg.text_dim2 * 4);
    s_imgemb = new_slot((size_t)B * cfg.image_dim * 4);
    s_tmpf = new_slot((size_t)B * cfg.n_image_embs * cfg.ctx_dim * 4 + (size_t)B * ted * 4);
    s_tmpf2 = new_slot((size_t)B * ted * 4);
    s_fullT = new_slot((size_t)B * (ntext_p > 0 ? ntext_p : 1) * cfg.text_dim1 * esz);
    s_ctxf = new_slot(cfg.head_type == 1 ? (size_t)B * cfg.ctx_len * cfg.ctx_dim * 4 : 0);
    if (cfg.hint_channels) {
      s_hint = new_slot((size_t)B * 4 * H * W * 4);
      s_hintin = new_slot((size_t)B * cfg.hint_channels * 64 * H * W * 4);
      for (int i = 0; i < 2; ++i) s_hbuf[i] = new_slot((size_t)B * 16 * 64 * H * W * 4);   // widest stage: 16 ch at 8H x 8W
    }
    s_part = new_slot(); s_coeff = new_slot(); s_P1 = new_slot(); s_U1 = new_slot(); s_P2 = new_slot(); s_S = new_slot();
    s_N = new_slot(); s_QKV = new_slot(); s_KALL = new_slot(); s_VT = new_slot(); s_ATT = new_slot();
    s_splitk = new_slot(256);
    s_U1st = new_slot();
    s_flush = new_slot(autotune ? ((size_t)320 << 20) : 0);  // evicts the Infinity Cache between tuning runs
    for (int i = 0; i < 3; ++i) { s_h[i] = new_slot(); s_hst[i] = new_slot(); }
    int hrot = 0, hcur = 0;
    auto next_h = [&]() { hcur = hrot; hrot = (hrot + 1) % 3; return s_h[hcur]; };
    auto cur_hst = [&]() { return s_hst[hcur]; };

    if (cfg.head_type == 1) build_cond_ops_22(); else build_cond_ops();
    if (cfg.hint_channels) build_hint_ops();

    // ---- time embedding + all FiLM vectors -------------------------------------------------
    {
      tw.freqs = Wf("time_freqs");
      tw.w0 = Wf("time_embed.0.weight"); tw.b0 = Wf("time_embed.0.bias");
      tw.w2 = Wf("time_embed.2.weight"); tw.b2 = Wf("time_embed.2.bias");
      tw.we = W_("emb_layers.weight"); tw.be = Wf("emb_layers.bias");
      tw.mc = mc; tw.ted = ted; tw.dt = sdt;
      const int Bn = B;
      ops.push_back([=](hipStream_t st) {
        if (film_cur) return (int)K22_OK;   // a hoisted loop computed this step's rows before its first step
        return time_rows(ptr<float>(s_t), ptr<float>(s_temb), ptr<float>(s_e1), ptr<float>(s_emb), ptr<float>(s_film), Bn, st);
      });
    }

    // ---- input blocks ------------------------------------------------------------------------
    int64_t film_cursor = 0;
    std::vector<Act> hs;
    int ch = mc * cfg.channel_mult[0];
    {
      Slot* d = new_slot((size_t)B * H * W * ch * esz);
      ConvInParams cp = {};
      cp.w = Wf("input_blocks.0.0.weight"); cp.bias = Wf("input_blocks.0.0.bias");
      cp.B = B; cp.H = H; cp.W = W; cp.Cin = cfg.in_channels; cp.Cout = ch;
      const int dt = sdt; const bool inpaint = cfg.in_channels == 9, hinted = cfg.in_channels == 8;
      const int premul = (hinted || cfg.head_type == 1) ? 1 : 0;   // 2.2: channels 4-7 arrive as they are (hint latent / masked image)
      ops.push_back([=](hipStream_t st) {
        ConvInParams q = cp;
        q.x = ptr<float>(s_xin); q.img = inpaint ? ptr<float>(s_img) : (hinted ? ptr<float>(s_hint) : nullptr);
        q.mask = inpaint ? ptr<float>(s_mask) : nullptr; q.img_premul = premul;
        q.out = ptr(d);
        return launch_conv_in(q, dt, st);
      });
      Act a; a.s0 = d; a.C0 = ch; a.H = H; a.W = W;
      hs.push_back(a);
    }
    Act h = hs.back();
    int ds = 1, blk = 1;
    for (int l = 0; l < cfg.n_levels; ++l) {
      for (int i = 0; i < cfg.num_res_blocks; ++i) {
        const int co = mc * cfg.channel_mult[l];
        const std::string pfx = "input_blocks." + std::to_string(blk);
        Slot* d = new_slot();
        if (has_attn(ds)) {
          Slot* rd = next_h();
          Act r = resblock(pfx + ".0", h, co, 0, film_cursor, rd, cur_hst());
          h = attnblock(pfx + ".1", r, d, new_slot());
        } else {
          h = resblock(pfx + ".0", h, co, 0, film_cursor, d, new_slot());
        }
        ch = co; hs.push_back(h); ++blk;
      }
      if (l != cfg.n_levels - 1) {
        const std::string pfx = "input_blocks." + std::to_string(blk);
        Slot* d = new_slot();
        h = resblock(pfx + ".0", h, ch, 1, film_cursor, d, new_slot());
        hs.push_back(h); ++blk; ds *= 2;
      }
    }
    // ---- middle ------------------------------------------------------------------------------
    {
      Slot* d0 = next_h(); Slot* st0 = cur_hst();
      Act r = resblock("middle_block.0", h, ch, 0, film_cursor, d0, st0);
      Slot* da = next_h(); Slot* sta = cur_hst();
      Act a = attnblock("middle_block.1", r, da, sta);
      Slot* d2 = next_h(); Slot* st2 = cur_hst();
      h = resblock("middle_block.2", a, ch, 0, film_cursor, d2, st2);
    }
    // ---- output blocks ---------------------------------------------------------------------
    blk = 0;
    for (int l = cfg.n_levels - 1; l >= 0; --l) {
      for (int i = 0; i <= cfg.num_res_blocks; ++i) {
        const Act skip = hs.back(); hs.pop_back();
        Act cat; cat.s0 = h.s0; cat.C0 = h.C0; cat.s1 = skip.s0; cat.C1 = skip.C0; cat.H = h.H; cat.W = h.W;
        cat.p0 = h.p0; cat.st0 = h.st0; cat.p1 = skip.p0; cat.st1 = skip.st0;
        if (skip.H != h.H || skip.W != h.W) return k22_set_error(K22_EINVAL, "unet: skip shape mismatch");
        const int co = mc * cfg.channel_mult[l];
        const std::string pfx = "output_blocks." + std::to_string(blk);
        { Slot* d = next_h(); Slot* dst_st = cur_hst(); h = resblock(pfx + ".0", cat, co, 0, film_cursor, d, dst_st); }
        int sub = 1;
        if (has_attn(ds)) { Slot* da = next_h(); Slot* sta = cur_hst(); h = attnblock(pfx + "." + std::to_string(sub), h, da, sta); ++sub; }
        if (l && i == cfg.num_res_blocks) {
          { Slot* d = next_h(); Slot* dst_st = cur_hst(); h = resblock(pfx + "." + std::to_string(sub), h, co, 2, film_cursor, d, dst_st); }
          ds /= 2;
        }
        ch = co; ++blk;
      }
    }
    if (film_cursor != film_total) return k22_set_error(K22_EINVAL, "unet: internal FiLM width mismatch");
    // ---- out: GN + SiLU + conv3x3 -> fp32 NCHW ---------------------------------------------------
    op_gn(ops, h, "out.0", -1, K22_ACT_SILU, 0, 1, s_P1);
    op_conv(ops, CONV_HEAD, s_P1, H, W, ch, cfg.out_channels, "out.2", nullptr, s_out, IG_OUT_NCHW_F32);

    if (!err.empty()) return k22_set_error(K22_EINVAL, err.c_str());
    // ---- lay the slots out -----------------------------------------------------------------------
    size_t off = 0;
    for (auto& s : slots) { s.off = off; off += (s.bytes + 255) / 256 * 256; }
    ws_bytes = off + 256;
    return K22_OK;
  }

  // Kandinsky 2.2 conditioning head (the UNet2DConditionModel injected at kandinsky2_2_model.py:26-41; arithmetic of diffusers'
  // ImageProjection / ImageTimeEmbedding, restated in oracle/unet22_ref.py): image-only.
  //   ctx[b][0:S] = LayerNorm_768( Linear(image_dim -> S*768)(image_emb[b]) viewed [S][768] )      (encoder_hid_proj)
  //   emb        += LayerNorm_1536( Linear(image_dim -> 1536)(image_emb[b]) )                       (add_embedding)
  void build_cond_ops_22() {
    const int mc = cfg.model_channels, ted = 4 * mc, Bn = B, dt = sdt;
    const int cd = cfg.ctx_dim, S = cfg.ctx_len, di = cfg.image_dim;
    const float* w_cp = Wf("head22.ctx_proj.weight"); const float* b_cp = Wf("head22.ctx_proj.bias");
    const float* g_cn = Wf("head22.ctx_norm.weight"); const float* b_cn = Wf("head22.ctx_norm.bias");
    const float* w_ep = Wf("head22.emb_proj.weight"); const float* b_ep = Wf("head22.emb_proj.bias");
    const float* g_en = Wf("head22.emb_norm.weight"); const float* b_en = Wf("head22.emb_norm.bias");
    cond_ops.push_back([=](hipStream_t st) {
      float* proj = ptr<float>(s_tmpf);                       // [B][S*cd]
      float* embp = ptr<float>(s_tmpf) + (size_t)Bn * S * cd;  // [B][ted]
      LinearSmallParams lp = {};
      lp.x = ptr<float>(s_imgemb); lp.ldx = di; lp.W = w_cp; lp.bias = b_cp; lp.out = proj; lp.ldo = S * cd;
      lp.M = Bn; lp.N = S * cd; lp.K = di;
      int rc = launch_linear_smallm(lp, K22_F32, st);
      if (rc) return rc;
      rc = launch_layernorm_f32(proj, g_cn, b_cn, ptr<float>(s_ctxf), Bn * S, cd, 1e-5f, st);
      if (rc) return rc;
      rc = launch_cast_rows(ptr<float>(s_ctxf), ptr(s_ctx), Bn * S, cd, cd, cd, dt, st);
      if (rc) return rc;
      lp.W = w_ep; lp.bias = b_ep; lp.out = embp; lp.ldo = ted; lp.N = ted;
      rc = launch_linear_smallm(lp, K22_F32, st);
      if (rc) return rc;
      return launch_layernorm_f32(embp, g_en, b_en, ptr<float>(s_xfproj), Bn, ted, 1e-5f, st);
    });
  }

  // input_hint_block of diffusers' ImageHintTimeEmbedding (ControlNet-depth UNet of Kandinsky 2.2): eight 3x3 convolutions,
  // SiLU between them, three of them stride 2: [B,3,8H,8W] -> [B,4,H,W].
  void build_hint_ops() {
    static const int chans[9] = {0, 16, 16, 32, 32, 96, 96, 256, 4};
    static const int strides[8] = {1, 1, 2, 1, 2, 1, 2, 1};
    int hin = 8 * H, win = 8 * W, cin = cfg.hint_channels;
    for (int k = 0; k < 8; ++k) {
      const float* w = Wf("hint." + std::to_string(k) + ".weight");
      const float* b = Wf("hint." + std::to_string(k) + ".bias");
      const int cout = chans[k + 1], stride = strides[k], Bn = B, ci = cin, hi = hin, wi = win;
      Slot* src = k == 0 ? s_hintin : s_hbuf[(k - 1) & 1];
      Slot* dst = k == 7 ? s_hint : s_hbuf[k & 1];
      hint_ops.push_back([=](hipStream_t st) {
        ConvDirectParams q = {};
        q.x = ptr<float>(src); q.w = w; q.bias = b; q.y = ptr<float>(dst);
        q.B = Bn; q.Cin = ci; q.Cout = cout; q.Hin = hi; q.Win = wi; q.stride = stride; q.act = k == 7 ? K22_ACT_NONE : K22_ACT_SILU;
        return launch_conv3x3_direct(q, st);
      });
      cin = cout;
      if (stride == 2) { hin = (hin - 1) / 2 + 1; win = (win - 1) / 2 + 1; }
    }
  }

  // Text2ImUNet.get_text_emb (text2im_model2_1.py:57-80), pooling_type == "from_model".
  void build_cond_ops() {
    const int mc = cfg.model_channels, ted = 4 * mc, Bn = B, dt = sdt;
    const int nie = cfg.n_image_embs, cd = cfg.ctx_dim, S = cfg.ctx_len, ntext = S - nie;
    const float* w_cs = Wf("clip_to_seq.weight"); const float* b_cs = Wf("clip_to_seq.bias");
    const float* w_pn = Wf("proj_n.weight"); const float* b_pn = Wf("proj_n.bias");
    const float* g_ln = Wf("ln_model_n.weight"); const float* b_ln = Wf("ln_model_n.bias");
    const float* w_il = Wf("img_layer.weight"); const float* b_il = Wf("img_layer.bias");
    const int d1 = cfg.text_dim1, d2 = cfg.text_dim2, di = cfg.image_dim;
    cond_ops.push_back([=](hipStream_t st) {
      float* clipseq = ptr<float>(s_tmpf);                  // [B][nie*cd]
      float* proj = ptr<float>(s_tmpf) + (size_t)Bn * nie * cd;  // [B][ted]
      LinearSmallParams lp = {};
      lp.x = ptr<float>(s_imgemb); lp.ldx = di; lp.W = w_cs; lp.bias = b_cs; lp.out = clipseq; lp.ldo = nie * cd;
      lp.M = Bn; lp.N = nie * cd; lp.K = di;
      int rc = launch_linear_smallm(lp, K22_F32, st);
      if (rc) return rc;
      // ctx[b][0:nie] = clip_seq
      for (int b = 0; b < Bn; ++b) {
        rc = launch_cast_rows(clipseq + (size_t)b * nie * cd, ptr(s_ctx) + (size_t)b * S * cd * esz, nie, cd, cd, cd, dt, st);
        if (rc) return rc;
      }
      // xf_proj = LN(proj_n(pooled)) + img_layer(image_emb)
      lp.x = ptr<float>(s_pool); lp.ldx = d2; lp.W = w_pn; lp.bias = b_pn; lp.out = proj; lp.ldo = ted; lp.N = ted; lp.K = d2;
      rc = launch_linear_smallm(lp, K22_F32, st);
      if (rc) return rc;
      rc = launch_layernorm_f32(proj, g_ln, b_ln, ptr<float>(s_tmpf2), Bn, ted, 1e-5f, st);
      if (rc) return rc;
      lp.x = ptr<float>(s_imgemb); lp.ldx = di; lp.W = w_il; lp.bias = b_il; lp.add = ptr<float>(s_tmpf2); lp.ld_add = ted;
      lp.out = ptr<float>(s_xfproj); lp.ldo = ted; lp.N = ted; lp.K = di;
      rc = launch_linear_smallm(lp, K22_F32, st);
      if (rc) return rc;
      // full_emb -> T
      return launch_cast_rows(ptr<float>(s_full), ptr(s_fullT), Bn * ntext, d1, d1, d1, dt, st);
    });
    // ctx[b][nie:S] = to_model_dim_n(full_emb[b])   (one GEMM per batch element: rows re-strided)
    {
      IgemmParams p = {};
    p.stages = -1;
      p.M = ntext; p.N = cd; p.Npad = (cd + 63) / 64 * 64; p.Kc = d1; p.K0 = d1; p.taps = 1; p.lda0 = d1; p.ldo = cd;
      p.out_mode = IG_OUT_ROWMAJOR; p.splitk = 1;
      p.a_raw = k22_is_split(dtype) ? 1 : 0;   // s_fullT: plain rows written by cast_rows
      p.Wp = W_("to_model_dim_n.weight"); p.bias = Wf("to_model_dim_n.bias");
      const size_t es = esz;
      const int gdt = gemm_dt(false);
      cond_ops.push_back([=](hipStream_t st) {
        for (int b = 0; b < Bn; ++b) {
          IgemmParams q = p;
          q.A0 = ptr(s_fullT) + (size_t)b * ntext * d1 * es;
          q.out = ptr(s_ctx) + ((size_t)b * S + nie) * cd * es;
          int rc = launch_igemm(q, gdt, st);
          if (rc) return rc;
        }
        return K22_OK;
      });
    }
  }
};

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int k22_unet_create(const K22UNetConfig* cfg, const K22Weight* weights, int n_weights, K22UNet** out) {
  if (!cfg || !out) return k22_set_error(K22_EINVAL, "unet_create: null argument");
  if (!k22_dtype_ok(cfg->dtype) && !k22_is_split(cfg->dtype)) return k22_set_error(K22_EINVAL, "unet_create: dtype");
  if (cfg->num_head_channels != 64) return k22_set_error(K22_EINVAL, "unet_create: only num_head_channels == 64");
  if (cfg->model_channels % 128) return k22_set_error(K22_EINVAL, "unet_create: model_channels % 128");
  if (cfg->n_levels < 1 || cfg->n_levels > 8) return k22_set_error(K22_EINVAL, "unet_create: n_levels");
  if (cfg->in_channels != 4 && cfg->in_channels != 9 && cfg->in_channels != 8) return k22_set_error(K22_EINVAL, "unet_create: in_channels must be 4, 8 or 9");
  if (cfg->head_type != 0 && cfg->head_type != 1) return k22_set_error(K22_EINVAL, "unet_create: head_type must be 0 (2.1) or 1 (2.2)");
  if ((cfg->hint_channels != 0) != (cfg->in_channels == 8) || (cfg->hint_channels != 0 && cfg->hint_channels != 3))
    return k22_set_error(K22_EINVAL, "unet_create: hint_channels = 3 goes with in_channels = 8 (latent + hint latent), else 0");
  if (cfg->head_type == 1 && cfg->n_image_embs != cfg->ctx_len) return k22_set_error(K22_EINVAL, "unet_create: the 2.2 head has image tokens only (n_image_embs == ctx_len)");
  K22UNet* u = new K22UNet();
  u->cfg = *cfg; u->dtype = cfg->dtype; u->sdt = k22_storage_dtype(cfg->dtype); u->esz = k22_esz(cfg->dtype);
  {
    const char* e = getenv("K22_AUTOTUNE");  // 0 = heuristics only (no measurement at the first forward)
    u->autotune = e ? (atoi(e) != 0) : 1;
    const char* sf = getenv("K22_STREAM");   // 0 = no fragment-major weight copies, no weight-streaming kernel
    u->stream_frag = sf ? (atoi(sf) != 0) : 1;
    const char* fg = getenv("K22_FUSE_GN");   // 0 = every GroupNorm through the stand-alone gn_apply kernel
    u->fuse_gn = fg ? (atoi(fg) != 0) : 0;
    const char* f = getenv("K22_FUSE_SKIP");  // 0 = 1x1 skip connections as separate GEMMs
    u->fuse_skip = f ? (atoi(f) != 0) : 1;
    const char* xp = getenv("K22_X2_PLAN");   // K22_F16X2 only: which of the top level's convolutions run with two MFMAs (K22UNet::x2_plan)
    u->x2_plan = xp ? (atoi(xp) & 3) : 0;
    const char* ht = getenv("K22_HOIST_TIME");   // 0 (measurement only) = k22_unet_sample_loop keeps the per-step time-embedding / FiLM launches
    u->hoist_time = ht ? (atoi(ht) != 0) : 1;
  }
  for (int i = 0; i < n_weights; ++i) u->w[weights[i].name] = weights[i].ptr;
  *out = u;
  return K22_OK;
}

void k22_unet_destroy(K22UNet* u) { delete u; }

int k22_unet_plan(K22UNet* u, int B, int H, int W, size_t* workspace_bytes) {
  if (!u || !workspace_bytes) return k22_set_error(K22_EINVAL, "unet_plan: null argument");
  int rc = u->plan(B, H, W);
  if (rc) return rc;
  *workspace_bytes = u->ws_bytes;
  return K22_OK;
}

int k22_unet_bind(K22UNet* u, void* workspace, size_t workspace_bytes) {
  if (!u || !workspace) return k22_set_error(K22_EINVAL, "unet_bind: null argument");
  if (u->ops.empty()) return k22_set_error(K22_EINVAL, "unet_bind: plan first");
  if (workspace_bytes < u->ws_bytes) return k22_set_error(K22_ENOMEM, "unet_bind: workspace too small");
  if ((uintptr_t)workspace % 256) return k22_set_error(K22_EINVAL, "unet_bind: workspace must be 256-byte aligned");
  u->ws = reinterpret_cast<char*>(workspace);
  u->cond_set = false; u->hint_set = false; u->frag_done = false;
  u->drop_graphs();
  return K22_OK;
}

int k22_unet_set_condition(K22UNet* u, const float* full_emb, const float* pooled_emb, const float* image_emb, void* stream) {
  if (!u || !u->ws) return k22_set_error(K22_EINVAL, "unet_set_condition: bind a workspace first");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const K22UNetConfig& c = u->cfg;
  const int ntext = c.ctx_len - c.n_image_embs;
  if (!image_emb) return k22_set_error(K22_EINVAL, "unet_set_condition: image_emb is required");
  if (c.head_type == 0 && (!full_emb || !pooled_emb)) return k22_set_error(K22_EINVAL, "unet_set_condition: the 2.1 head needs full_emb and pooled_emb");
  hipError_t e;
  const size_t pb = (size_t)u->B;
  if (c.head_type == 0) {
    e = hipMemcpyAsync(u->ptr(u->s_full), full_emb, pb * ntext * c.text_dim1 * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    e = hipMemcpyAsync(u->ptr(u->s_pool), pooled_emb, pb * c.text_dim2 * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  }
  e = hipMemcpyAsync(u->ptr(u->s_imgemb), image_emb, pb * c.image_dim * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  for (auto& op : u->cond_ops) { int rc = op(st); if (rc) return rc; }
  u->cond_set = true;
  return K22_OK;
}

int k22_unet_set_hint(K22UNet* u, const float* hint, void* stream) {
  if (!u || !u->ws) return k22_set_error(K22_EINVAL, "unet_set_hint: bind a workspace first");
  if (!u->cfg.hint_channels) return k22_set_error(K22_EINVAL, "unet_set_hint: this UNet has no hint input (hint_channels == 0)");
  if (!hint) return k22_set_error(K22_EINVAL, "unet_set_hint: null hint");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t per = (size_t)u->B * u->cfg.hint_channels * 64 * u->H * u->W;
  hipError_t e = hipMemcpyAsync(u->ptr(u->s_hintin), hint, per * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  for (auto& op : u->hint_ops) { int rc = op(st); if (rc) return rc; }
  u->hint_set = true;
  return K22_OK;
}

int k22_unet_forward(K22UNet* u, const float* x, const float* timesteps, const float* inpaint_image,
                     const float* inpaint_mask, float* out, int use_graph, void* stream) {
  if (!u || !u->ws) return k22_set_error(K22_EINVAL, "unet_forward: bind a workspace first");
  if (!u->cond_set) return k22_set_error(K22_EINVAL, "unet_forward: call k22_unet_set_condition first");
  if (u->cfg.in_channels == 9 && (!inpaint_image || !inpaint_mask)) return k22_set_error(K22_EINVAL, "unet_forward: inpainting UNet needs inpaint_image and inpaint_mask");
  if (u->cfg.hint_channels && !u->hint_set) return k22_set_error(K22_EINVAL, "unet_forward: call k22_unet_set_hint first");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t hw = (size_t)u->H * u->W;
  hipError_t e;
#define K22_CPY(dst, src, bytes)                                                   \
  e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);                \
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  const size_t pb = (size_t)u->B;
  K22_CPY(u->ptr(u->s_xin), x, pb * 4 * hw * 4);
  K22_CPY(u->ptr(u->s_t), timesteps, pb * 4);
  if (u->cfg.in_channels == 9) {
    K22_CPY(u->ptr(u->s_img), inpaint_image, pb * 4 * hw * 4);
    K22_CPY(u->ptr(u->s_mask), inpaint_mask, pb * hw * 4);
  }
  // first forward on this plan: fragment-major weight copies; conv / GEMM problems the tile table does not know are measured on the device
  { int rc = u->prepare_run(st); if (rc) return rc; }
  if (use_graph) {
    if (!u->graph_exec) {
      // warm-up eagerly once (sets function attributes), then capture
      if (!u->warmed) { int rc = u->exec_eager(st); if (rc) return rc; }
      hipGraph_t g = nullptr;
      if (!u->cap_stream) {
        e = hipStreamCreateWithFlags(&u->cap_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
      }
      e = hipStreamBeginCapture(u->cap_stream, hipStreamCaptureModeThreadLocal);
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
      const int rc = u->exec(u->cap_stream);
      e = hipStreamEndCapture(u->cap_stream, &g);
      if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
      e = hipGraphInstantiate(&u->graph_exec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e != hipSuccess) { u->graph_exec = nullptr; return k22_set_error_hip(e, __FILE__, __LINE__); }
    }
    e = hipGraphLaunch(u->graph_exec, st);
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  } else {
    int rc = u->exec_eager(st);
    if (rc) return rc;
  }
  K22_CPY(out, u->ptr(u->s_out), pb * u->cfg.out_channels * hw * 4);
#undef K22_CPY
  return K22_OK;
}

// The whole classifier-free-guided p_sampler loop of Kandinsky2_1.generate_img (kandinsky2_1_model.py:222-257 ->
// gaussian_diffusion.py:384-475) as ONE hipGraph: for every step  UNet([x_half | x_half], t_k) -> k22_sampler_step  with no host work
// between steps (the reference returns to Python - and to the CPU for np.percentile - at every step).  All inputs are device buffers
// the caller fills BEFORE the call: timesteps [n_steps][B] (the values the UNet receives, in execution order), noise_seq
// [n_steps][B][4][HW] (the ancestral noise of every step, drawn up front), table [T][8] + table_rows[k] (host array: the schedule row
// of step k).  x [B][4][HW] is updated in place (x_tmp: scratch of the same size).  The captured graph is replayed as long as the
// same buffers and scalars are passed (a generation service re-uses its buffers); anything else re-captures.
int k22_unet_sample_loop(K22UNet* u, float* x, float* x_tmp, const float* timesteps, const float* noise_seq, const float* init_img,
                         const float* mask, const float* inpaint_image, const float* inpaint_mask, const float* table,
                         const int* table_rows, int n_steps, float guidance, float clamp_lo, float clamp_hi, int pct_index,
                         double pct_gamma, void* scratch, int use_graph, void* stream) {
  if (!u || !u->ws) return k22_set_error(K22_EINVAL, "unet_sample_loop: bind a workspace first");
  if (!u->cond_set) return k22_set_error(K22_EINVAL, "unet_sample_loop: call k22_unet_set_condition first");
  if (!x || !x_tmp || !timesteps || !noise_seq || !table || !table_rows || !scratch || n_steps < 1)
    return k22_set_error(K22_EINVAL, "unet_sample_loop: null argument");
  if (u->B % 2) return k22_set_error(K22_EINVAL, "unet_sample_loop: the batch is the CFG batch [cond | uncond] (even)");
  if (u->cfg.out_channels != 8) return k22_set_error(K22_EINVAL, "unet_sample_loop: the UNet must predict eps and variance (8 channels)");
  if (u->cfg.in_channels == 9 && (!inpaint_image || !inpaint_mask)) return k22_set_error(K22_EINVAL, "unet_sample_loop: inpainting UNet needs inpaint_image and inpaint_mask");
  if (u->cfg.hint_channels && !u->hint_set) return k22_set_error(K22_EINVAL, "unet_sample_loop: call k22_unet_set_hint first");
  const int B = u->B, HW = u->H * u->W;
  if (pct_index >= 4 * HW) return k22_set_error(K22_EINVAL, "unet_sample_loop: percentile index out of range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e;
  { int rc = u->prepare_run(st); if (rc) return rc; }
  const size_t half = (size_t)(B / 2) * 4 * HW * sizeof(float);
  const size_t pb = (size_t)B;
  // the time / FiLM rows of all steps up front (K22UNet::film_all): 8 / B steps per batched launch; B > 8 keeps the per-step launches
  const bool hoist = B <= 8 && u->hoist_time;
  const int64_t rows_all = (int64_t)n_steps * B;
  float *fa_film = nullptr, *fa_temb = nullptr, *fa_e1 = nullptr, *fa_emb = nullptr;
  if (hoist) {
    const size_t need = (size_t)rows_all * (size_t)(u->film_total + u->tw.mc + 2 * u->tw.ted) * sizeof(float);
    if (need > u->film_all_bytes) {
      u->drop_graphs();   // captured loops hold the old buffer's addresses
      if (u->film_all) { (void)hipStreamSynchronize(st); (void)hipFree(u->film_all); u->film_all = nullptr; u->film_all_bytes = 0; }
      e = hipMalloc(reinterpret_cast<void**>(&u->film_all), need);
      if (e != hipSuccess) { u->film_all = nullptr; return k22_set_error_hip(e, __FILE__, __LINE__); }
      u->film_all_bytes = need;
    }
    fa_film = u->film_all;
    fa_temb = fa_film + rows_all * u->film_total;
    fa_e1 = fa_temb + rows_all * u->tw.mc;
    fa_emb = fa_e1 + rows_all * u->tw.ted;
  }
  // one pass over the loop on `s`: what is captured is exactly what an eager call runs
  auto run_loop_body = [&](hipStream_t s) -> int {
    hipError_t er;
    if (hoist) {
      const int spl = 8 / B;   // steps per batched launch
      for (int k0 = 0; k0 < n_steps; k0 += spl) {
        const int rows = (n_steps - k0 < spl ? n_steps - k0 : spl) * B;
        const int64_t r0 = (int64_t)k0 * B;
        const int rc = u->time_rows(timesteps + r0, fa_temb + r0 * u->tw.mc, fa_e1 + r0 * u->tw.ted, fa_emb + r0 * u->tw.ted, fa_film + r0 * u->film_total, rows, s);
        if (rc) return rc;
      }
    }
    if (u->cfg.in_channels == 9) {
      er = hipMemcpyAsync(u->ptr(u->s_img), inpaint_image, pb * 4 * HW * 4, hipMemcpyDeviceToDevice, s);
      if (er != hipSuccess) return k22_set_error_hip(er, __FILE__, __LINE__);
      er = hipMemcpyAsync(u->ptr(u->s_mask), inpaint_mask, pb * HW * 4, hipMemcpyDeviceToDevice, s);
      if (er != hipSuccess) return k22_set_error_hip(er, __FILE__, __LINE__);
    }
    float* cur = x; float* nxt = x_tmp;
    for (int k = 0; k < n_steps; ++k) {
      // model_fn: the UNet sees the first half twice (kandinsky2_1_model.py:223-225)
      er = hipMemcpyAsync(u->ptr(u->s_xin), cur, half, hipMemcpyDeviceToDevice, s);
      if (er != hipSuccess) return k22_set_error_hip(er, __FILE__, __LINE__);
      er = hipMemcpyAsync(u->ptr(u->s_xin) + half, cur, half, hipMemcpyDeviceToDevice, s);
      if (er != hipSuccess) return k22_set_error_hip(er, __FILE__, __LINE__);
      if (hoist) {
        u->film_cur = fa_film + (int64_t)k * B * u->film_total;
      } else {
        er = hipMemcpyAsync(u->ptr(u->s_t), timesteps + (size_t)k * B, (size_t)B * 4, hipMemcpyDeviceToDevice, s);
        if (er != hipSuccess) return k22_set_error_hip(er, __FILE__, __LINE__);
      }
      int rc = u->exec(s);
      if (rc) return rc;
      SamplerParams p = {};
      p.x = cur; p.model_out = u->model_out(); p.noise = noise_seq + (size_t)k * B * 4 * HW; p.init_img = init_img; p.mask = mask;
      p.table = table; p.step = nullptr; p.step_host = table_rows[k]; p.guidance = guidance; p.clamp_lo = clamp_lo; p.clamp_hi = clamp_hi;
      p.use_cfg = 1; p.n_lo = pct_index; p.gamma = pct_gamma;
      p.s_buf = reinterpret_cast<float*>(scratch); p.x0_buf = reinterpret_cast<float*>(scratch) + 64;
      p.x_out = nxt; p.x0_out = nullptr; p.N = B; p.HW = HW;
      rc = launch_sampler_step(p, s);
      if (rc) return rc;
      float* t_ = cur; cur = nxt; nxt = t_;
    }
    if (cur != x) {
      er = hipMemcpyAsync(x, cur, (size_t)B * 4 * HW * sizeof(float), hipMemcpyDeviceToDevice, s);
      if (er != hipSuccess) return k22_set_error_hip(er, __FILE__, __LINE__);
    }
    return K22_OK;
  };
  auto run_loop = [&](hipStream_t s) -> int { const int rc = run_loop_body(s); u->film_cur = nullptr; return rc; };
  if (!use_graph) return run_loop(st);
  // key of the captured loop: every pointer and scalar baked into its nodes
  std::vector<unsigned long long> key = {(unsigned long long)(uintptr_t)x, (unsigned long long)(uintptr_t)x_tmp, (unsigned long long)(uintptr_t)timesteps,
      (unsigned long long)(uintptr_t)noise_seq, (unsigned long long)(uintptr_t)init_img, (unsigned long long)(uintptr_t)mask,
      (unsigned long long)(uintptr_t)inpaint_image, (unsigned long long)(uintptr_t)inpaint_mask, (unsigned long long)(uintptr_t)table,
      (unsigned long long)(uintptr_t)scratch, (unsigned long long)n_steps, (unsigned long long)pct_index, (unsigned long long)(hoist ? 1 : 0)};
  auto bits = [](double v) { unsigned long long b; memcpy(&b, &v, 8); return b; };
  key.push_back(bits(guidance)); key.push_back(bits(clamp_lo)); key.push_back(bits(clamp_hi)); key.push_back(bits(pct_gamma));
  for (int k = 0; k < n_steps; ++k) key.push_back((unsigned long long)table_rows[k]);
  if (!u->loop_exec || key != u->loop_key) {
    if (u->loop_exec) { (void)hipGraphExecDestroy(u->loop_exec); u->loop_exec = nullptr; }
    if (!u->warmed) {   // first forward of this plan: run one step's ops eagerly (function attributes, code load) on a scratch input; a re-capture
                        // for other scalars / buffers (guidance, step count: they are baked into the graph's nodes) does not repeat it
      e = hipMemsetAsync(u->ptr(u->s_xin), 0, pb * 4 * HW * 4, st);
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
      e = hipMemcpyAsync(u->ptr(u->s_t), timesteps, pb * 4, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
      if (u->cfg.in_channels == 9) {
        e = hipMemcpyAsync(u->ptr(u->s_img), inpaint_image, pb * 4 * HW * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
        e = hipMemcpyAsync(u->ptr(u->s_mask), inpaint_mask, pb * HW * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
      }
      int rc = u->exec_eager(st);
      if (rc) return rc;
      e = hipStreamSynchronize(st);
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    }
    if (!u->cap_stream) {
      e = hipStreamCreateWithFlags(&u->cap_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    }
    hipGraph_t g = nullptr;
    e = hipStreamBeginCapture(u->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    const int rc = run_loop(u->cap_stream);
    e = hipStreamEndCapture(u->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
    e = hipGraphInstantiate(&u->loop_exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { u->loop_exec = nullptr; return k22_set_error_hip(e, __FILE__, __LINE__); }
    u->loop_key = key;
  }
  e = hipGraphLaunch(u->loop_exec, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  return K22_OK;
}

int k22_unet_num_ops(const K22UNet* u) {
  return u ? (int)u->ops.size() : 0;
}

int k22_unet_set_autotune(K22UNet* u, int on) {
  if (!u) return k22_set_error(K22_EINVAL, "unet_set_autotune: null handle");
  if (!u->ops.empty() && (on != 0) != (u->autotune != 0)) return k22_set_error(K22_EINVAL, "unet_set_autotune: call before k22_unet_plan");
  u->autotune = on ? 1 : 0;
  return K22_OK;
}

// Text table of the chosen tile configurations (one line per distinct conv / GEMM problem of the plan).
int k22_unet_tuning_report(const K22UNet* u, char* buf, size_t cap) {
  if (!u || !buf || cap == 0) return k22_set_error(K22_EINVAL, "unet_tuning_report: null argument");
  const std::string out = tuning_report_text(u->tuned);
  snprintf(buf, cap, "%s", out.c_str());
  return K22_OK;
}

int k22_unet_profile(K22UNet* u, int reps, double* ms, double* flops, double* bytes, int* launches, void* stream) {
  if (!u || !u->ws || !u->cond_set) return k22_set_error(K22_EINVAL, "unet_profile: run k22_unet_forward once first");
  if (reps < 1) reps = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int k = 0; k < OP_NKINDS; ++k) { ms[k] = 0.0; flops[k] = 0.0; bytes[k] = 0.0; launches[k] = 0; }
  int rc = K22_OK;
  {
    K22UNet* q = u;
    if (!q->frag_done) { int rc0 = q->repack_frags(st); if (rc0) return rc0; }   // a profile before the first forward of this binding
    const size_t n = q->ops.size();
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) { hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return k22_set_error_hip(r, __FILE__, __LINE__); }
    for (size_t i = 0; i < n; ++i) {
      flops[q->ops[i].kind] += q->ops[i].flops + q->ops[i].flops2;  // a fused 1x1 skip is work of the conv launch
      bytes[q->ops[i].kind] += q->ops[i].bytes;
      launches[q->ops[i].kind] += q->ops[i].kernels;
    }
    for (int r = 0; r < reps && rc == K22_OK; ++r) {
      for (size_t i = 0; i < n; ++i) {
        (void)hipEventRecord(ev[2 * i], st);
        rc = q->ops[i](st);
        (void)hipEventRecord(ev[2 * i + 1], st);
        if (rc) break;
      }
      hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) { rc = k22_set_error_hip(e, __FILE__, __LINE__); break; }
      for (size_t i = 0; i < n; ++i) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]);
        ms[q->ops[i].kind] += (double)t / reps;
      }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  return rc;
}

}  // extern "C"
