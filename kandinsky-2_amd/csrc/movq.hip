// k22 — MoVQ decoder engine: latent [B,4,h,w] -> image [B,3,8h,8w]  (MOVQ.decode, kandinsky2/vqgan/autoencoder.py:182-185;
// MOVQDecoder.forward, kandinsky2/vqgan/movq_modules.py:326-357), and MoVQ ENCODER engine: image [B,3,H,W] -> latent
// [B,4,H/8,W/8] (MOVQ.encode, autoencoder.py:176-180; Encoder.forward, kandinsky2/vqgan/vqgan_blocks.py:335-367: the
// img2img / inpainting pre-step, kandinsky2_1_model.py:458-469, 519-534).  One K22MoVQ handle carries one plan of either.
//
// Same building blocks as the UNet engine: NHWC activations in a caller-owned workspace, every 3x3 conv input
// written zero-bordered by the (Spatial)Norm-apply kernel, convolutions / 1x1 projections through launch_igemm
// (LDS-resident halo kernel where the row fits the LDS, generic implicit GEMM at the wide levels), GroupNorm
// statistics by gn_stats + gn_coeff.  MoVQ specifics:
//   * SpatialNorm: GN(f) * conv_y(zq) + conv_b(zq), zq = raw latent nearest-resized — evaluated inside the apply
//     kernel from the 4-channel latent (movq_kernels.hip), nothing is materialised;
//   * AttnBlock (single head, C = 512, T = h*w keys): q / k projections as GEMMs, v projected TRANSPOSED by swapping
//     the GEMM operands (V^T = Wv . h^T, so it is directly the [N][K] operand of the P.V GEMM), scores = Q K^T as a
//     GEMM per image into a [T][T] buffer (the reference materialises it too, movq_modules.py:214-218), row softmax,
//     O = P V^T^T + bv (softmax rows sum to 1, so the v bias moves behind the product), proj_out + residual.
#include "kernels.h"
#include "elementwise.h"
#include "../../include/k22.h"

#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
struct MSlot { size_t bytes = 0, off = 0; };
typedef std::function<int(hipStream_t)> MOp;
struct MAct { MSlot* s = nullptr; int C = 0, H = 0, W = 0; };
}  // namespace

struct K22MoVQ {
  K22MoVQConfig cfg;
  int dtype; size_t esz;
  std::unordered_map<std::string, const void*> w;
  int B = 0, h0 = 0, w0 = 0;
  std::deque<MSlot> slots;
  std::vector<MOp> ops;
  size_t ws_bytes = 0;
  char* ws = nullptr;
  std::string err;
  MSlot *s_zq, *s_xin, *s_part, *s_coeff, *s_P, *s_U, *s_S, *s_N, *s_Q, *s_K, *s_VT, *s_SC, *s_O, *s_splitk, *s_out, *s_z;
  MSlot *s_img = nullptr, *s_lat = nullptr;   // encoder plan: fp32 NCHW image in, fp32 NCHW latent out
  bool enc = false;                           // the current plan is the encoder (plain GroupNorm instead of SpatialNorm)
  int encH = 0, encW = 0;
  MSlot* s_h[3];
  int hrot = 0;

  MSlot* new_slot(size_t bytes = 0) { slots.emplace_back(); slots.back().bytes = bytes; return &slots.back(); }
  static void need(MSlot* s, size_t bytes) { if (bytes > s->bytes) s->bytes = bytes; }
  template <typename T = char> T* ptr(const MSlot* s) const { return reinterpret_cast<T*>(ws + s->off); }
  MSlot* next_h() { MSlot* s = s_h[hrot]; hrot = (hrot + 1) % 3; return s; }
  const void* W_(const std::string& name) {
    auto it = w.find(name);
    if (it == w.end()) { if (err.empty()) err = "missing weight: " + name; return nullptr; }
    return it->second;
  }
  const float* Wf(const std::string& name) { return reinterpret_cast<const float*>(W_(name)); }
  static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

  // generic launch of one igemm problem with the library's heuristics
  void push_igemm(IgemmParams p, std::function<void(IgemmParams&)> bind) {
    p.stages = -1;
    p.splitk = igemm_choose_splitk(p, dtype);
    if (p.splitk > 1) need(s_splitk, (size_t)p.splitk * p.M * p.N * sizeof(float));
    const int dt = dtype;
    ops.push_back([=](hipStream_t st) {
      IgemmParams q = p;
      bind(q);
      q.partial = ptr<float>(s_splitk);
      return launch_igemm(q, dt, st);
    });
  }

  // SpatialNorm (+act) -> dst (zero-bordered when pad)
  void op_sn(const MAct& in, const std::string& pfx, int act, int pad, MSlot* dst) {
    const int Bn = B, C = in.C, HW = in.H * in.W;
    const int nsplit = gn_nsplit(Bn, HW);
    need(s_part, (size_t)Bn * nsplit * C * 2 * sizeof(float));
    need(s_coeff, (size_t)Bn * C * 2 * sizeof(float));
    need(dst, (size_t)Bn * (in.H + 2 * pad) * (in.W + 2 * pad) * C * esz);
    const float* gamma = Wf(pfx + ".norm_layer.weight"); const float* beta = Wf(pfx + ".norm_layer.bias");
    const float* wy = Wf(pfx + ".conv_y.weight"); const float* by = Wf(pfx + ".conv_y.bias");
    const float* wb = Wf(pfx + ".conv_b.weight"); const float* bb = Wf(pfx + ".conv_b.bias");
    const MAct a = in;
    const int dt = dtype, shift = ilog2(in.H / h0), hh = h0, ww = w0;
    ops.push_back([=](hipStream_t st) {
      GnStatsParams sp;
      sp.x0 = ptr(a.s); sp.x1 = nullptr; sp.C0 = C; sp.C1 = 0; sp.HW = HW; sp.B = Bn; sp.groups = 32; sp.nsplit = nsplit;
      sp.partial = ptr<float>(s_part);
      int rc = launch_gn_stats(sp, dt, st);
      if (rc) return rc;
      GnCoeffParams cp = {};
      cp.src[0].st = ptr<float>(s_part); cp.src[0].rpi = nsplit; cp.src[0].C = C;
      cp.HW = HW; cp.C = C; cp.groups = 32; cp.eps = 1e-6f; cp.gamma = gamma; cp.beta = beta; cp.film = nullptr; cp.film_ld = 0;
      cp.coeff = ptr<float>(s_coeff);
      rc = launch_gn_coeff(cp, Bn, st);
      if (rc) return rc;
      SpatialNormParams np;
      np.x = ptr(a.s); np.coeff = ptr<float>(s_coeff); np.zq = ptr<float>(s_zq); np.wy = wy; np.by = by; np.wb = wb; np.bb = bb;
      np.out = ptr(dst); np.B = Bn; np.H = a.H; np.W = a.W; np.C = C; np.h0 = hh; np.w0 = ww; np.shift = shift; np.act = act; np.pad = pad;
      return launch_spatialnorm_apply(np, dt, st);
    });
  }

  // plain GroupNorm (vqgan_blocks.Normalize: 32 groups, eps 1e-6, affine) (+act) -> dst (zero-bordered when pad)
  void op_gn(const MAct& in, const std::string& pfx, int act, int pad, MSlot* dst) {
    const int Bn = B, C = in.C, HW = in.H * in.W;
    const int nsplit = gn_nsplit(Bn, HW);
    need(s_part, (size_t)Bn * nsplit * C * 2 * sizeof(float));
    need(s_coeff, (size_t)Bn * C * 2 * sizeof(float));
    need(dst, (size_t)Bn * (in.H + 2 * pad) * (in.W + 2 * pad) * C * esz);
    const float* gamma = Wf(pfx + ".weight"); const float* beta = Wf(pfx + ".bias");
    const MAct a = in;
    const int dt = dtype;
    ops.push_back([=](hipStream_t st) {
      GnStatsParams sp = {};
      sp.x0 = ptr(a.s); sp.x1 = nullptr; sp.C0 = C; sp.C1 = 0; sp.HW = HW; sp.B = Bn; sp.groups = 32; sp.nsplit = nsplit;
      sp.partial = ptr<float>(s_part);
      int rc = launch_gn_stats(sp, dt, st);
      if (rc) return rc;
      GnCoeffParams cp = {};
      cp.src[0].st = ptr<float>(s_part); cp.src[0].rpi = nsplit; cp.src[0].C = C;
      cp.HW = HW; cp.C = C; cp.groups = 32; cp.eps = 1e-6f; cp.gamma = gamma; cp.beta = beta; cp.film = nullptr; cp.film_ld = 0;
      cp.coeff = ptr<float>(s_coeff);
      rc = launch_gn_coeff(cp, Bn, st);
      if (rc) return rc;
      GnApplyParams ap = {};
      ap.x0 = ptr(a.s); ap.x1 = nullptr; ap.C0 = C; ap.C1 = 0; ap.B = Bn; ap.H = a.H; ap.W = a.W; ap.mode = 0; ap.pad = pad; ap.act = act;
      ap.coeff = ptr<float>(s_coeff); ap.out = ptr(dst);
      return launch_gn_apply(ap, dt, st);
    });
  }
  void op_norm(const MAct& in, const std::string& pfx, int act, int pad, MSlot* dst) {
    if (enc) op_gn(in, pfx, act, pad, dst);
    else op_sn(in, pfx, act, pad, dst);
  }

  void op_conv3(MSlot* src, int H, int W, int Cin, int Cout, const std::string& pfx, MSlot* residual, MSlot* dst, int out_mode) {
    IgemmParams p = {};
    p.M = B * H * W; p.N = Cout; p.Npad = (Cout + 63) / 64 * 64; p.Kc = Cin; p.K0 = Cin; p.taps = 9; p.H = H; p.W = W;
    p.ldo = Cout; p.ldr = Cout; p.out_mode = out_mode; p.act = K22_ACT_NONE;
    p.Wp = W_(pfx + ".weight"); p.bias = Wf(pfx + ".bias");
    need(dst, out_mode == IG_OUT_ROWMAJOR ? (size_t)p.M * Cout * esz : (size_t)p.M * Cout * sizeof(float));
    push_igemm(p, [=](IgemmParams& q) { q.A0 = ptr(src); q.residual = residual ? ptr(residual) : nullptr; q.out = ptr(dst); });
  }

  void op_gemm(MSlot* a, int M, int N, int K, const std::string& pfx, MSlot* residual, MSlot* dst) {
    IgemmParams p = {};
    p.M = M; p.N = N; p.Npad = (N + 63) / 64 * 64; p.Kc = K; p.K0 = K; p.taps = 1; p.lda0 = K; p.ldo = N; p.ldr = N;
    p.out_mode = IG_OUT_ROWMAJOR; p.act = K22_ACT_NONE;
    p.Wp = W_(pfx + ".weight"); p.bias = Wf(pfx + ".bias");
    need(dst, (size_t)M * N * esz);
    push_igemm(p, [=](IgemmParams& q) { q.A0 = ptr(a); q.residual = residual ? ptr(residual) : nullptr; q.out = ptr(dst); });
  }

  // ResnetBlock (movq_modules.py:120-182), temb is None in the decoder
  MAct resblock(const std::string& pfx, const MAct& in, int Cout) {
    const int Cin = in.C, H = in.H, W = in.W;
    op_norm(in, pfx + ".norm1", K22_ACT_SILU, 1, s_P);
    op_conv3(s_P, H, W, Cin, Cout, pfx + ".conv1", nullptr, s_U, IG_OUT_ROWMAJOR);
    MAct u; u.s = s_U; u.C = Cout; u.H = H; u.W = W;
    op_norm(u, pfx + ".norm2", K22_ACT_SILU, 1, s_P);
    MSlot* skip = in.s;
    if (Cin != Cout) {
      op_gemm(in.s, B * H * W, Cout, Cin, pfx + ".nin_shortcut", nullptr, s_S);
      skip = s_S;
    }
    MSlot* d = next_h();
    if (d == in.s) d = next_h();
    op_conv3(s_P, H, W, Cout, Cout, pfx + ".conv2", skip, d, IG_OUT_ROWMAJOR);
    MAct out; out.s = d; out.C = Cout; out.H = H; out.W = W;
    return out;
  }

  // AttnBlock (movq_modules.py:185-225)
  MAct attnblock(const std::string& pfx, const MAct& in) {
    const int C = in.C, T = in.H * in.W, Bn = B, dt = dtype;
    const size_t es = esz;
    if (T % 64 || C % 64) { if (err.empty()) err = "movq attention: h*w and C must be multiples of 64"; }
    op_norm(in, pfx + ".norm", K22_ACT_NONE, 0, s_N);
    op_gemm(s_N, B * T, C, C, pfx + ".q", nullptr, s_Q);
    op_gemm(s_N, B * T, C, C, pfx + ".k", nullptr, s_K);
    need(s_VT, (size_t)B * C * T * esz);
    need(s_SC, (size_t)B * T * T * esz);
    need(s_O, (size_t)B * T * C * esz);
    const void* wv = W_(pfx + ".v.weight"); const float* bv = Wf(pfx + ".v.bias");
    for (int b = 0; b < Bn; ++b) {
      // V^T_b [C][T] = Wv [C][C] . N_b [T][C]^T
      {
        IgemmParams p = {};
        p.M = C; p.N = T; p.Npad = T; p.Kc = C; p.K0 = C; p.taps = 1; p.lda0 = C; p.ldo = T; p.ldr = T; p.out_mode = IG_OUT_ROWMAJOR;
        push_igemm(p, [=](IgemmParams& q) { q.A0 = wv; q.Wp = ptr(s_N) + (size_t)b * T * C * es; q.bias = nullptr; q.out = ptr(s_VT) + (size_t)b * C * T * es; });
      }
      // scores_b [T][T] = Q_b . K_b^T
      {
        IgemmParams p = {};
        p.M = T; p.N = T; p.Npad = T; p.Kc = C; p.K0 = C; p.taps = 1; p.lda0 = C; p.ldo = T; p.ldr = T; p.out_mode = IG_OUT_ROWMAJOR;
        push_igemm(p, [=](IgemmParams& q) { q.A0 = ptr(s_Q) + (size_t)b * T * C * es; q.Wp = ptr(s_K) + (size_t)b * T * C * es; q.bias = nullptr;
                                            q.out = ptr(s_SC) + (size_t)b * T * T * es; });
      }
    }
    const float scale = 1.0f / sqrtf((float)C);
    ops.push_back([=](hipStream_t st) { return launch_softmax_rows(ptr(s_SC), (int64_t)Bn * T, T, scale, dt, st); });
    for (int b = 0; b < Bn; ++b) {
      IgemmParams p = {};
      p.M = T; p.N = C; p.Npad = C; p.Kc = T; p.K0 = T; p.taps = 1; p.lda0 = T; p.ldo = C; p.ldr = C; p.out_mode = IG_OUT_ROWMAJOR;
      push_igemm(p, [=](IgemmParams& q) { q.A0 = ptr(s_SC) + (size_t)b * T * T * es; q.Wp = ptr(s_VT) + (size_t)b * C * T * es; q.bias = bv;
                                          q.out = ptr(s_O) + (size_t)b * T * C * es; });
    }
    MSlot* d = next_h();
    if (d == in.s) d = next_h();
    op_gemm(s_O, B * T, C, C, pfx + ".proj_out", in.s, d);
    MAct out; out.s = d; out.C = C; out.H = in.H; out.W = in.W;
    return out;
  }

  int plan(int nB, int nh, int nw) {
    B = nB; h0 = nh; w0 = nw; enc = false;
    slots.clear(); ops.clear(); err.clear(); ws = nullptr; hrot = 0;
    const int nres = cfg.n_levels;
    if (B < 1 || h0 < 1 || w0 < 1) return k22_set_error(K22_EINVAL, "movq: empty input");
    s_z = new_slot((size_t)B * 4 * h0 * w0 * 4);
    s_zq = new_slot((size_t)B * h0 * w0 * 4 * 4);
    s_xin = new_slot((size_t)B * (h0 + 2) * (w0 + 2) * 64 * esz);
    s_part = new_slot(); s_coeff = new_slot(); s_P = new_slot(); s_U = new_slot(); s_S = new_slot(); s_N = new_slot();
    s_Q = new_slot(); s_K = new_slot(); s_VT = new_slot(); s_SC = new_slot(); s_O = new_slot(); s_splitk = new_slot(256);
    for (int i = 0; i < 3; ++i) s_h[i] = new_slot();
    const int H8 = h0 << (nres - 1), W8 = w0 << (nres - 1);
    s_out = new_slot((size_t)B * cfg.out_ch * H8 * W8 * 4);

    // post_quant_conv + layouts
    {
      const float* wpq = Wf("post_quant_conv.weight"); const float* bpq = Wf("post_quant_conv.bias");
      const int Bn = B, hh = h0, ww = w0, dt = dtype;
      ops.push_back([=](hipStream_t st) { return launch_movq_prepare(ptr<float>(s_z), wpq, bpq, ptr<float>(s_zq), ptr(s_xin), Bn, hh, ww, 64, dt, st); });
    }
    int block_in = cfg.ch * cfg.ch_mult[nres - 1];
    MAct hcur;
    {
      MSlot* d = next_h();
      op_conv3(s_xin, h0, w0, 64, block_in, "decoder.conv_in", nullptr, d, IG_OUT_ROWMAJOR);
      hcur.s = d; hcur.C = block_in; hcur.H = h0; hcur.W = w0;
    }
    hcur = resblock("decoder.mid.block_1", hcur, block_in);
    hcur = attnblock("decoder.mid.attn_1", hcur);
    hcur = resblock("decoder.mid.block_2", hcur, block_in);
    for (int lvl = nres - 1; lvl >= 0; --lvl) {
      const int block_out = cfg.ch * cfg.ch_mult[lvl];
      const bool attn = (cfg.attn_levels >> lvl) & 1;
      for (int i = 0; i <= cfg.num_res_blocks; ++i) {
        const std::string pfx = "decoder.up." + std::to_string(lvl);
        hcur = resblock(pfx + ".block." + std::to_string(i), hcur, block_out);
        block_in = block_out;
        if (attn) hcur = attnblock(pfx + ".attn." + std::to_string(i), hcur);
      }
      if (lvl != 0) {
        // Upsample: nearest x2 + conv3x3
        const MAct a = hcur;
        const int Bn = B, dt = dtype, C = hcur.C;
        need(s_P, (size_t)B * (2 * a.H + 2) * (2 * a.W + 2) * C * esz);
        ops.push_back([=](hipStream_t st) { return launch_upsample2_pad(ptr(a.s), ptr(s_P), Bn, a.H, a.W, C, dt, st); });
        MSlot* d = next_h();
        if (d == a.s) d = next_h();
        op_conv3(s_P, 2 * a.H, 2 * a.W, C, C, "decoder.up." + std::to_string(lvl) + ".upsample.conv", nullptr, d, IG_OUT_ROWMAJOR);
        hcur.s = d; hcur.H = 2 * a.H; hcur.W = 2 * a.W;
      }
    }
    op_sn(hcur, "decoder.norm_out", K22_ACT_SILU, 1, s_P);
    op_conv3(s_P, hcur.H, hcur.W, hcur.C, cfg.out_ch, "decoder.conv_out", nullptr, s_out, IG_OUT_NCHW_F32);
    if (!err.empty()) return k22_set_error(K22_EINVAL, err.c_str());
    size_t off = 0;
    for (auto& s : slots) { s.off = off; off += (s.bytes + 255) / 256 * 256; }
    ws_bytes = off + 256;
    return K22_OK;
  }

  // Encoder.forward + quant_conv.  H, W: image size (multiples of 2^(levels-1); (H/8)*(W/8) a multiple of 64 for the
  // attention GEMMs).
  int plan_enc(int nB, int H, int W) {
    const int nres = cfg.n_levels;
    B = nB; enc = true; encH = H; encW = W;
    slots.clear(); ops.clear(); err.clear(); ws = nullptr; hrot = 0;
    if (B < 1 || H < 1 || W < 1 || H % (1 << (nres - 1)) || W % (1 << (nres - 1))) return k22_set_error(K22_EINVAL, "movq encoder: H, W must be positive multiples of 2^(levels-1)");
    h0 = H >> (nres - 1); w0 = W >> (nres - 1);
    s_img = new_slot((size_t)B * 3 * H * W * 4);
    s_xin = new_slot((size_t)B * (H + 2) * (W + 2) * 64 * esz);
    s_part = new_slot(); s_coeff = new_slot(); s_P = new_slot(); s_U = new_slot(); s_S = new_slot(); s_N = new_slot();
    s_Q = new_slot(); s_K = new_slot(); s_VT = new_slot(); s_SC = new_slot(); s_O = new_slot(); s_splitk = new_slot(256);
    for (int i = 0; i < 3; ++i) s_h[i] = new_slot();
    s_out = new_slot((size_t)B * cfg.z_channels * h0 * w0 * 4);
    s_lat = new_slot((size_t)B * 4 * h0 * w0 * 4);
    s_z = s_zq = nullptr;
    const int Bn = B, dt = dtype;
    ops.push_back([=](hipStream_t st) { return launch_movq_enc_prepare(ptr<float>(s_img), ptr(s_xin), Bn, H, W, 64, dt, st); });
    MAct hcur;
    {
      MSlot* d = next_h();
      op_conv3(s_xin, H, W, 64, cfg.ch, "encoder.conv_in", nullptr, d, IG_OUT_ROWMAJOR);
      hcur.s = d; hcur.C = cfg.ch; hcur.H = H; hcur.W = W;
    }
    for (int lvl = 0; lvl < nres; ++lvl) {
      const int block_out = cfg.ch * cfg.ch_mult[lvl];
      const bool attn = (cfg.attn_levels >> lvl) & 1;
      const std::string pfx = "encoder.down." + std::to_string(lvl);
      for (int i = 0; i < cfg.num_res_blocks; ++i) {
        hcur = resblock(pfx + ".block." + std::to_string(i), hcur, block_out);
        if (attn) hcur = attnblock(pfx + ".attn." + std::to_string(i), hcur);
      }
      if (lvl != nres - 1) {
        // Downsample: F.pad (0,1,0,1) + conv3x3 stride 2 = the stride-1 "same" conv taken at the odd positions
        const MAct a = hcur;
        const int C = hcur.C;
        need(s_P, (size_t)B * (a.H + 2) * (a.W + 2) * C * esz);
        ops.push_back([=](hipStream_t st) { return launch_pad_copy(ptr(a.s), ptr(s_P), Bn, a.H, a.W, C, dt, st); });
        op_conv3(s_P, a.H, a.W, C, C, pfx + ".downsample.conv", nullptr, s_U, IG_OUT_ROWMAJOR);
        MSlot* d = next_h();
        if (d == a.s) d = next_h();
        need(d, (size_t)B * (a.H / 2) * (a.W / 2) * C * esz);
        ops.push_back([=](hipStream_t st) { return launch_subsample_odd(ptr(s_U), ptr(d), Bn, a.H, a.W, C, dt, st); });
        hcur.s = d; hcur.H = a.H / 2; hcur.W = a.W / 2;
      }
    }
    hcur = resblock("encoder.mid.block_1", hcur, hcur.C);
    hcur = attnblock("encoder.mid.attn_1", hcur);
    hcur = resblock("encoder.mid.block_2", hcur, hcur.C);
    op_gn(hcur, "encoder.norm_out", K22_ACT_SILU, 1, s_P);
    op_conv3(s_P, hcur.H, hcur.W, hcur.C, cfg.z_channels, "encoder.conv_out", nullptr, s_out, IG_OUT_NCHW_F32);
    {
      const float* wq = Wf("quant_conv.weight"); const float* bq = Wf("quant_conv.bias");
      const int hw = h0 * w0;
      ops.push_back([=](hipStream_t st) { return launch_movq_quant_conv(ptr<float>(s_out), wq, bq, ptr<float>(s_lat), Bn, hw, st); });
    }
    if (!err.empty()) return k22_set_error(K22_EINVAL, err.c_str());
    size_t off = 0;
    for (auto& s : slots) { s.off = off; off += (s.bytes + 255) / 256 * 256; }
    ws_bytes = off + 256;
    return K22_OK;
  }
};

extern "C" {

int k22_movq_create(const K22MoVQConfig* cfg, const K22Weight* weights, int n_weights, K22MoVQ** out) {
  if (!cfg || !out) return k22_set_error(K22_EINVAL, "movq_create: null argument");
  if (!k22_dtype_ok(cfg->dtype)) return k22_set_error(K22_EINVAL, "movq_create: dtype");
  if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->z_channels != 4 || cfg->ch % 128) return k22_set_error(K22_EINVAL, "movq_create: unsupported configuration (z_channels == 4, ch % 128 == 0)");
  K22MoVQ* m = new K22MoVQ();
  m->cfg = *cfg; m->dtype = cfg->dtype; m->esz = cfg->dtype == K22_F32 ? 4 : 2;
  for (int i = 0; i < n_weights; ++i) m->w[weights[i].name] = weights[i].ptr;
  *out = m;
  return K22_OK;
}
void k22_movq_destroy(K22MoVQ* m) { delete m; }

int k22_movq_plan(K22MoVQ* m, int B, int h, int w, size_t* workspace_bytes) {
  if (!m || !workspace_bytes) return k22_set_error(K22_EINVAL, "movq_plan: null argument");
  int rc = m->plan(B, h, w);
  if (rc) return rc;
  *workspace_bytes = m->ws_bytes;
  return K22_OK;
}
int k22_movq_bind(K22MoVQ* m, void* workspace, size_t workspace_bytes) {
  if (!m || !workspace) return k22_set_error(K22_EINVAL, "movq_bind: null argument");
  if (m->ops.empty()) return k22_set_error(K22_EINVAL, "movq_bind: plan first");
  if (workspace_bytes < m->ws_bytes) return k22_set_error(K22_ENOMEM, "movq_bind: workspace too small");
  if ((uintptr_t)workspace % 256) return k22_set_error(K22_EINVAL, "movq_bind: workspace must be 256-byte aligned");
  m->ws = reinterpret_cast<char*>(workspace);
  return K22_OK;
}
int k22_movq_decode(K22MoVQ* m, const float* z, float* out, unsigned char* out_u8, void* stream) {
  if (!m || !m->ws || m->enc) return k22_set_error(K22_EINVAL, "movq_decode: plan the decoder and bind a workspace first");
  if (!z || (!out && !out_u8)) return k22_set_error(K22_EINVAL, "movq_decode: null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemcpyAsync(m->ptr(m->s_z), z, (size_t)m->B * 4 * m->h0 * m->w0 * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  for (auto& op : m->ops) { int rc = op(st); if (rc) return rc; }
  const int H8 = m->h0 << (m->cfg.n_levels - 1), W8 = m->w0 << (m->cfg.n_levels - 1);
  if (out) {
    e = hipMemcpyAsync(out, m->ptr(m->s_out), (size_t)m->B * m->cfg.out_ch * H8 * W8 * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  }
  if (out_u8) return launch_to_uint8_nhwc(m->ptr<float>(m->s_out), out_u8, m->B, m->cfg.out_ch, H8, W8, st);
  return K22_OK;
}
int k22_movq_plan_encoder(K22MoVQ* m, int B, int H, int W, size_t* workspace_bytes) {
  if (!m || !workspace_bytes) return k22_set_error(K22_EINVAL, "movq_plan_encoder: null argument");
  int rc = m->plan_enc(B, H, W);
  if (rc) return rc;
  *workspace_bytes = m->ws_bytes;
  return K22_OK;
}
int k22_movq_encode(K22MoVQ* m, const float* image, float* latent, void* stream) {
  if (!m || !m->ws || !m->enc) return k22_set_error(K22_EINVAL, "movq_encode: plan the encoder and bind a workspace first");
  if (!image || !latent) return k22_set_error(K22_EINVAL, "movq_encode: null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipMemcpyAsync(m->ptr(m->s_img), image, (size_t)m->B * 3 * m->encH * m->encW * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  for (auto& op : m->ops) { int rc = op(st); if (rc) return rc; }
  e = hipMemcpyAsync(latent, m->ptr(m->s_lat), (size_t)m->B * 4 * m->h0 * m->w0 * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return k22_set_error_hip(e, __FILE__, __LINE__);
  return K22_OK;
}
int k22_movq_num_ops(const K22MoVQ* m) { return m ? (int)m->ops.size() : 0; }

}  // extern "C"
