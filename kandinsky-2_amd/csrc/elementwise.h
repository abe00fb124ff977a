// k22 — parameter blocks + launchers of the HBM-bound kernels (elementwise.hip), attention
// (attention.hip) and the sampler step (sampler.hip).
#pragma once
#include "common.h"

struct GnStatsParams {
  const void* x0; const void* x1;   // NHWC [B][HW][C0] (+ [B][HW][C1] virtual concat)
  int C0, C1, HW, B, groups, nsplit;
  float* partial;                    // [B][nsplit][C][2] = per-channel (sum, sumsq) of each pixel range
};
// Per-channel partial sums of one tensor: image b owns rows [b*rpi, (b+1)*rpi) of st[rows][C][2].  Written by
// gn_stats_kernel or by the producing convolution's epilogue (conv3_halo.hip / splitk_reduce_rows_kernel).
struct GnSrc { const float* st; int rpi, C; };
struct GnCoeffParams {
  GnSrc src[2];                            // src[1].C == 0 when the input is not a virtual concat
  int HW, C, groups; float eps;
  const float* gamma; const float* beta;  // [C]
  const float* film; int64_t film_ld;      // optional [B][film_ld]: scale at +c, shift at +C+c
  float* coeff;                            // [B][C][2] = (A, Bc):  y = x*A + Bc
};
struct GnApplyParams {
  const void* x0; const void* x1; int C0, C1;
  int B, H, W;          // input spatial dims
  int mode;             // 0 same, 1 avgpool2 after activation, 2 nearest-up2 after activation
  int pad;              // 1 -> write [B][Ho+2][Wo+2][C] with a zero border
  int act;              // K22Act
  const float* coeff;
  void* out;
  int out_x3;           // fp32 launches only (the K22_F16X3 engine): write the output in x3 chunks (common.h) - what the consuming
                        // convolution / GEMM of the split-precision arithmetic reads as its A operand
};
struct ConvInParams {
  const float* x; const float* img; const float* mask;  // NCHW fp32; img/mask only for Cin == 9
  const float* w; const float* bias;                    // [Cin*3*3][Cout] (pack.py transposes the reference's [Cout][Cin][3][3]), [Cout] fp32
  void* out;                                            // NHWC T [B][H][W][Cout]
  int B, H, W, Cin, Cout;
  int img_premul;                                       // channels 4-7 = img as is (2.2: hint latent / already masked image); else img*mask
};
// Direct fp32 3x3 convolution, pad 1, stride 1 or 2, NCHW in / out, optional SiLU: the ControlNet-depth hint stack of the 2.2 UNet
// (3 -> 16 -> 16 -> 32/2 -> 32 -> 96/2 -> 96 -> 256/2 -> 4 on the 8h x 8w hint image), run ONCE per generation.
struct ConvDirectParams {
  const float* x; const float* w; const float* bias; float* y;   // x [B][Cin][Hin][Win], w [Cout][Cin][3][3], y [B][Cout][Ho][Wo]
  int B, Cin, Cout, Hin, Win, stride, act;
};
int launch_conv3x3_direct(const ConvDirectParams& p, hipStream_t s);
struct LinearSmallParams {
  const float* x; int64_t ldx;  // [M][K] fp32
  const void* W;                // [N][K] (dtype given at launch)
  const float* bias;            // [N] or null
  const float* add; int64_t ld_add;  // optional [M][N] added after the output activation
  int add_mod;                       // > 0: row m adds row m % add_mod of `add` (the rows of several sampler steps batched in one call)
  float* out; int64_t ldo;
  int M, N, K, act_in, act_out;
  int rows_per_wave;  // set by the launcher
};
struct KvPackParams {
  const void* qkv; const void* ctxkv; void* kall; void* vtall;
  int B, H, T, S, Tkp;
};
struct AttentionParams {
  const void* q; int64_t ldq;   // query rows [B*T] with stride ldq; head h at column h*64
  const void* kall; const void* vtall;  // [B][H][Tkp][64], [B][H][64][Tkp]
  void* out; int64_t ldo;       // [B*T][ldo], head h at column h*64
  int B, H, T, Tk, Tkp;
  float scale;                  // applied to q.k (reference: ch^-0.25 on both q and k => 1/8)
  int causal;                   // 1: key j attends only to queries t >= j (prior transformer, prior.py:326-334)
  const float* key_valid;       // optional [B][kv_ld]: 0 = padding key (masked for every query); keys >= kv_n are valid
  int kv_ld, kv_n;
  int out_x3;                   // K22_F16X3 only: write `out` in x3 chunks (common.h) - the operand format of the GEMM that reads it
};
// small-T attention of the prior transformer (attention.hip: small_attention_kernel)
struct SmallAttnParams {
  const void* qkv; int64_t ldq;   // [B * T][ldq] = [Q | K | V] x [H][64] (row-major, c_qkv output with Q | K | V planes)
  const float* part; int nsplit; const float* bias;   // part != null: qkv = T(bias + sum_s part[s][B * T][ldq]) instead (c_qkv's fp32 split-K partials, nsplit <= 4)
  void* out; int64_t ldo;         // row-major [B * T][ldo] (out_frag == 0) or the A-fragment order of a skinny GEMM with K = H * 64
  int out_frag, MA;               // MA = ceil(B * T / 32) m-atoms of the fragment tensor
  int B, H, T;
  float scale;
  int causal;
  const float* key_valid; int kv_ld, kv_n;   // as AttentionParams
};
struct SamplerParams {
  const float* x;          // [N][4][HW] current latent (fp32 NCHW)
  const float* model_out;  // [N][8][HW] raw UNet output (before classifier-free guidance)
  const float* noise;      // [N][4][HW]
  const float* init_img;   // inpainting: [N][4][HW] or null
  const float* mask;       // inpainting: [N][1][HW] or null
  const float* table;      // [steps][8] per-step scalars (see sampler.hip)
  const int* step;         // device step counter (index into table) or null -> step_host
  int step_host;
  float guidance, clamp_lo, clamp_hi;
  int use_cfg;             // 1: eps = u + g (c - u) with halves [cond | uncond]; 0: eps as is
  int n_lo; double gamma;  // percentile order statistic index and interpolation weight; n_lo < 0: off
  float* x0_buf;           // scratch [N][4][HW]
  float* s_buf;            // scratch [1]
  float* x_out;            // [N][4][HW]
  float* x0_out;           // pred_xstart [N][4][HW] or null
  int N, HW;
};

int launch_x3_pack(const float* in, void* out, int64_t n, float scale, hipStream_t s);
int gn_nsplit(int B, int HW);
int launch_gn_stats(const GnStatsParams& p, int dtype, hipStream_t s);
int launch_gn_coeff(const GnCoeffParams& p, int B, hipStream_t s);
int launch_gn_apply(const GnApplyParams& p, int dtype, hipStream_t s);
int launch_resample(const void* x, void* y, int B, int H, int W, int C, int mode, int dtype, hipStream_t s);
int launch_conv_in(const ConvInParams& p, int dtype, hipStream_t s);
int launch_timestep_embedding(const float* t, const float* freqs, float* out, int B, int half, hipStream_t s);
int launch_linear_smallm(const LinearSmallParams& p, int wdtype, hipStream_t s);
int launch_layernorm_f32(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, hipStream_t s);
int launch_cast_rows(const float* x, void* y, int rows, int cols, int64_t ldx, int64_t ldy, int dtype, hipStream_t s);
int launch_kv_pack(const KvPackParams& p, int dtype, hipStream_t s);
int launch_attention(const AttentionParams& p, int dtype, hipStream_t s);
int launch_small_attention(const SmallAttnParams& p, int dtype, hipStream_t s);
int launch_sampler_step(const SamplerParams& p, hipStream_t s);
int launch_plms_step(const float* x, const float* model_out, const float* h1, const float* h2, const float* h3, int order, const float* tab,
                     float guidance, int use_cfg, float* x_out, float* e_store, float* x0_out, int N, int HW, hipStream_t s);
int launch_prepare_mask(const float* old_mask, float* out, int C, int H, int W, hipStream_t s);
int launch_ddim_step(const float* x, const float* model_out, const float* noise, const float* tab, float guidance, int use_cfg,
                     float* x_out, float* x0_out, int N, int HW, hipStream_t s);

// ---- MoVQ decoder (movq.hip / movq_kernels.hip) -----------------------------------------------------------
struct SpatialNormParams {
  const void* x;          // NHWC T [B][H][W][C]
  const float* coeff;     // [B][C][2] GroupNorm (A, Bc) incl. gamma / beta
  const float* zq;        // NHWC fp32 [B][h0][w0][4] raw latent; nearest-resized: (y, x) -> (y >> shift, x >> shift)
  const float* wy; const float* by;   // conv_y 1x1: [C][4], [C]
  const float* wb; const float* bb;   // conv_b 1x1: [C][4], [C]
  void* out;              // NHWC T, optionally zero-bordered
  int B, H, W, C, h0, w0, shift, act, pad;
};
int launch_spatialnorm_apply(const SpatialNormParams& p, int dtype, hipStream_t s);
// nearest x2 upsample of an unpadded NHWC tensor into a zero-bordered one (input of Upsample.conv)
int launch_upsample2_pad(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s);
// in-place softmax over rows of length L (scaled logits), T storage, fp32 math
int launch_softmax_rows(void* x, int64_t rows, int L, float scale, int dtype, hipStream_t s);
// z [B][4][h][w] fp32 NCHW -> zq NHWC fp32 [B][h][w][4] and post_quant_conv(z) as zero-bordered NHWC T with the 4
// channels zero-extended to Cpad (the K slab of the implicit GEMM)
int launch_movq_prepare(const float* z, const float* wpq, const float* bpq, float* zq, void* xin, int B, int h, int w,
                        int Cpad, int dtype, hipStream_t s);
// fp32 NCHW [-1,1] image -> uint8 NHWC:  ((x + 1) * 127.5).round().clamp(0, 255)   (kandinsky2/utils.py:57-70)
int launch_movq_enc_prepare(const float* img, void* xin, int B, int H, int W, int Cpad, int dtype, hipStream_t s);
int launch_pad_copy(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s);
int launch_subsample_odd(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s);
int launch_movq_quant_conv(const float* h, const float* wq, const float* bq, float* out, int B, int HW, hipStream_t s);
int launch_to_uint8_nhwc(const float* x, unsigned char* y, int B, int C, int H, int W, hipStream_t s);
