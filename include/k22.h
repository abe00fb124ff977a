/* k22.h — C ABI of libk22hip.so, the MI355X (gfx950) native Kandinsky-2 sampling engine.
 *
 * The reference (ai-forever/Kandinsky-2, pure PyTorch) has no FFI; its seams on the sampling hot path
 * are Python objects.  Each entry point below names the reference call it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - extern "C"; every function returns 0 (K22_OK) or a negative K22_E* code; k22_last_error()
 *     returns a thread-local message.  No C++ exception crosses the boundary.
 *   - The CALLER owns every buffer (inputs, outputs, weight arena, workspace).  All pointers are
 *     device pointers valid on the current HIP device unless a parameter says "host".
 *   - All work is ENQUEUED on the hipStream_t passed as `stream` (void* here so the header needs no
 *     HIP include); nothing synchronises, nothing allocates device memory.
 *   - Activations inside the engine are NHWC (channels last); the public tensors keep the reference's
 *     NCHW fp32 layout so the Python modules stay drop-in.
 */
#ifndef K22_H
#define K22_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define K22_OK 0
#define K22_EINVAL (-1)
#define K22_EHIP (-2)
#define K22_ENOMEM (-3)

#define K22_BF16 0 /* product path: bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulate */
#define K22_F32 1  /* parity path : fp32 storage, v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) */
#define K22_F16 2  /* the reference's own reduced-precision mode (use_fp16 / convert_to_fp16, kandinsky2/model/unet.py:409, 566-572):
                      fp16 storage, v_mfma_f32_32x32x16_f16, fp32 accumulate - bf16's speed and bytes, 3 more mantissa bits */
#define K22_F16X3 3 /* UNet engine only - SPLIT PRECISION, the arithmetic that holds BASELINE.json's 1e-3 final-latent gate against the
                      reference p_sampler (kandinsky2/kandinsky2_1_model.py:245-257 -> kandinsky2/model/gaussian_diffusion.py:384-475) at
                      16-bit MFMA rate: tensors stay fp32, every MFMA operand x is carried as the fp16 pair hi = rne(x), lo = rne(x - hi)
                      and every product as three v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi) into one fp32 accumulator: ~23
                      significand bits per operand at 3/16 of the exact-fp32 MFMA cost.  Operand format ("x3 chunks"): 4 bytes per
                      element, every aligned group of 4 consecutive K elements stored as [hi x4 | lo x4] fp16 (csrc/common.h);
                      weights are packed so (pre-multiplied by 2^8) by pack.py / k22_x3_pack, activations by their producers. */
#define K22_F16X2 4 /* UNet engine only - the ASYMMETRIC SPLIT (round 5): the tensors, the arena and the operand formats of K22_F16X3, with
                      the precision of every MFMA op a property of the PLAN: weights always as (hi, lo) pairs (their rounding is the
                      systematic error of a 16-bit engine), the activation operand at fp16 precision - two MFMAs per product,
                      w_hi.a_hi + w_lo.a_hi - where the operand-rounding ablation (oracle/drift_ablation.py, tests/golden/
                      drift_ablation_x2.json) prices it as cheap: the 3x3 convolutions below the top resolution level and the qkv
                      projections; ONE MFMA (fp16 tiles, fp32 softmax) in the attention; all three where it is not: 1x1 skip
                      connections, the out head, proj_out / encoder_kv, and (default plan; env K22_X2_PLAN bits 0 / 1 release its
                      in_layers / out_layers convolutions) the top level.  C2 final latent 2.7e-4 from the reference p_sampler
                      (gate 1e-3) at 1.17x the K22_F16X3 engine's speed.  The kernel-level entries accept it: k22_conv3x3* / k22_gemm* /
                      k22_qkv_project = two MFMAs, activation operand rounded to fp16 (a fused skip keeps three); k22_attention = fp16
                      tiles on fp32 tensors. */

int k22_version(void);
/* Build flavour of this library: bit mask of K22_BUILD_*.  0 = the shipped form. */
#define K22_BUILD_PACKED_FP32 1     /* built with NOPK=0: packed-fp32 VALU instructions present - do NOT overlap two handles on one device */
#define K22_BUILD_DEBUG_VARIANTS 2  /* measurement-only kernel variants compiled in (K22_DEBUG_VARIANTS / K22_STREAM_DEBUG / K22_SKINNY_DEBUG) */
int k22_build_flags(void);
const char* k22_last_error(void);
/* Tuning knobs.  PROCESS-WIDE MUTABLE STATE, NOT RE-ENTRANT: these exist for the kernel-level test / measurement surface (k22_gemm,
 * k22_conv3x3*, k22_groupnorm) and must not be changed while another thread is inside any k22_* call.  Leave them at their defaults
 * in a process that runs engines (k22_unet_*, k22_prior_*, k22_movq_*, k22_encoder_*): engine launches carry the configuration of their
 * tile-table line, but a line that says "generic kernel" (algo 0) is dispatched through the same switch these knobs override.
 * Everything else in this header is re-entrant per handle (one handle = one stream at a time), and handles may run CONCURRENTLY on
 * different streams of one device: since round 6 the library is built without packed-fp32 VALU instructions (csrc/Makefile NOPK = 1), the
 * instruction class behind the one wrong-result kernel pair rounds 4-5 found under concurrency (0 of 900 victim launches wrong with every
 * offender, against 148-227 of 900 in a packed build on the same box: profiles/r06_two_stream_probe.txt).  k22_build_flags() lets a
 * host assert it: a library built with NOPK=0 reports K22_BUILD_PACKED_FP32 and must be driven one stream at a time.
 * Knobs: "igemm_stages" = 2..4 LDS-DMA pipeline depth (-1 default; with "gemm_algo" = 10: 2 = two workgroups per CU at BM = 128, 3 / 4 = the
 * specialised, pipelined gemm8_spec_kernel for the 16-bit types and its two-per-CU form - same bits as the lock-step kernel);
 * "igemm_xcd_remap" = 0/1 XCD-aware workgroup renumbering; "conv_algo" = 0 auto, 1 generic implicit GEMM,
 * 2 LDS-resident halo kernel for the 3x3 convolutions (3-7: its variants, see conv3_halo.hip; 8-9: measurement only);
 * "gemm_algo" = 0 generic implicit-GEMM kernel, 10 = 8-wave BM x 128 tile kernel where it applies;
 * "conv_algo" / "gemm_algo" = 20: the weight-streaming small-M kernel (stream_gemm.hip; bm = 160 / 288 picks 5 / 9 m-blocks
 * per workgroup, needs the fp32 partial buffer also for splitk == 1). */
int k22_set_option(const char* name, int value);
/* Launch counters for tests ("stream_launches": launches of the weight-streaming kernel since the library was loaded);
 * -1 for an unknown name. */
long k22_debug_counter(const char* name);
/* Fragment-major copy of a 16-bit weight matrix [Npad][taps * Kc] for the weight-streaming kernel (1 KB contiguous per MFMA B
 * fragment; layout in stream_gemm.hip).  k22_stream_frag_bytes = size of the copy (0 for fp32: the kernel is 16-bit only).
 * k22_debug_set_stream_frag hands the copies (main weights, fused-skip weights or NULL) to the kernel-level test entries below
 * until reset with NULLs; the engines keep their own copies and never read it. */
size_t k22_stream_frag_bytes(int Npad, int taps, int Kc, int dtype);
int k22_stream_repack(const void* W, void* out, int Npad, int taps, int Kc, int dtype, void* stream);
int k22_debug_set_stream_frag(const void* wfrag, const void* wsfrag);
/* Parity-test mode: with a scratch buffer set (and no explicit copies), every kernel-level test entry repacks its weights into the
 * scratch on `stream` before the launch, so the tests' own packing helpers exercise the fragment-major path unchanged.  NULL resets. */
int k22_debug_set_stream_scratch(void* scratch, size_t bytes);

/* ---- UNet engine --------------------------------------------------------------------------
 * Replaces Text2ImUNet.forward / InpaintText2ImUNet.forward (kandinsky2/model/text2im_model2_1.py:
 * 85-103, 146-155) and everything below it: UNetModel blocks (kandinsky2/model/unet.py:343-611),
 * ResBlock (:110-220), AttentionBlock/QKVAttention (:223-340), GroupNorm32 (kandinsky2/model/nn.py:
 * 26-37), timestep_embedding (nn.py:101-121).  Hyper-parameters mirror create_model()
 * (kandinsky2/model/model_creation.py:9-83) / CONFIG_2_1["model_config"] (kandinsky2/configs.py:125-149).
 */
typedef struct K22UNetConfig {
  int dtype;               /* K22_BF16 | K22_F32 | K22_F16 | K22_F16X3 | K22_F16X2 */
  int in_channels;         /* 4; 9 for the inpainting UNet (x, image*mask, mask); 8 for the 2.2 ControlNet-depth UNet (x, hint latent) */
  int model_channels;      /* 384 */
  int out_channels;        /* 8 = eps + learned variance */
  int num_res_blocks;      /* 3 */
  int n_levels;            /* len(channel_mult) */
  int channel_mult[8];     /* (1,2,3,4) */
  int n_attention_ds;      /* attention at these downsample rates */
  int attention_ds[8];     /* (2,4,8) */
  int num_head_channels;   /* 64 (only 64 is implemented) */
  int ctx_dim;             /* model_dim / encoder_channels, 768 */
  int ctx_len;             /* num_image_embs + text_ctx = 10 + 77 */
  int n_image_embs;        /* 10 */
  int text_dim1;           /* text_encoder_in_dim1, 1024 */
  int text_dim2;           /* text_encoder_in_dim2, 768 (pooled) */
  int image_dim;           /* image_encoder_in_dim, 768 (2.2: CLIP-bigG image embedding, 1280) */
  int head_type;           /* 0: Kandinsky 2.1 head (Text2ImUNet.get_text_emb: 10 image tokens + 77 text tokens,
                            *    text2im_model2_1.py:57-80);
                            * 1: Kandinsky 2.2 head (image-only: ctx = LayerNorm(Linear(image_emb)) as ctx_len tokens, additive
                            *    time term LayerNorm(Linear(image_emb)): the UNet2DConditionModel the reference injects,
                            *    kandinsky2_2_model.py:26-41) */
  int hint_channels;       /* 0, or 3: ControlNet-depth variant of the 2.2 UNet (hint image through the input_hint_block conv
                            *    stack, concatenated to the latent: in_channels 8; notebooks/kandinsky2_2_controlnet.ipynb:9235) */
} K22UNetConfig;

/* One packed parameter tensor inside the caller's weight arena (see kandinsky-2_amd/pack.py for the
 * layouts; names follow the reference state_dict keys so reference checkpoints load). */
typedef struct K22Weight {
  const char* name;
  const void* ptr;
} K22Weight;

typedef struct K22UNet K22UNet; /* opaque */

/* Builds the op graph description.  `weights` pointers must stay valid for the engine's lifetime. */
int k22_unet_create(const K22UNetConfig* cfg, const K22Weight* weights, int n_weights, K22UNet** out);
void k22_unet_destroy(K22UNet* u);

/* Plans a forward for batch B (= 2*bs with classifier-free guidance) and latent H x W; returns the
 * workspace size the caller must provide to k22_unet_bind().  Re-planning invalidates the binding. */
int k22_unet_plan(K22UNet* u, int B, int H, int W, size_t* workspace_bytes);
int k22_unet_bind(K22UNet* u, void* workspace, size_t workspace_bytes);

/* Conditioning head, once per generation: Text2ImUNet.get_text_emb (text2im_model2_1.py:57-80) plus
 * the step-invariant AttentionBlock.encoder_kv projections (unet.py:263-264).
 * full_emb [B,77,text_dim1], pooled_emb [B,text_dim2], image_emb [B,image_dim], all fp32 contiguous. */
int k22_unet_set_condition(K22UNet* u, const float* full_emb, const float* pooled_emb, const float* image_emb,
                           void* stream);

/* ControlNet-depth variant only (hint_channels == 3), once per generation after k22_unet_bind: hint [B,3,8H,8W] fp32 NCHW
 * (the depth map in [0,1]) -> input_hint_block -> [B,4,H,W], kept in the workspace and concatenated to x by every forward. */
int k22_unet_set_hint(K22UNet* u, const float* hint, void* stream);

/* One UNet evaluation.  x [B,4,H,W] fp32 NCHW, timesteps [B] fp32 (already mapped/rescaled exactly as
 * _WrappedModel.__call__ does, kandinsky2/model/respace.py:128-133), inpaint_image [B,4,H,W] and
 * inpaint_mask [B,1,H,W] (NULL unless in_channels == 9; the product image*mask is formed inside like
 * text2im_model2_1.py:151), out [B,out_channels,H,W] fp32 NCHW.
 * use_graph != 0 replays a captured hipGraph of the whole forward (captured on first use). */
int k22_unet_forward(K22UNet* u, const float* x, const float* timesteps, const float* inpaint_image,
                     const float* inpaint_mask, float* out, int use_graph, void* stream);

/* Number of engine ops in one planned forward (diagnostics). */
/* The whole classifier-free-guided p_sampler loop of Kandinsky2_1.generate_img (kandinsky2/kandinsky2_1_model.py:222-257 driving
 * kandinsky2/model/gaussian_diffusion.py:384-475) as ONE hipGraph replay: per step  UNet([x_half | x_half], timesteps[k]) ->
 * k22_sampler_step(guidance, clamp, percentile threshold, posterior mean, noise_seq[k])  with no host work between steps.
 * Device buffers, filled by the caller before the call: x [B][4][HW] (in: x_T, out: the final latent; x_tmp: same-size scratch),
 * timesteps [n_steps][B] in execution order, noise_seq [n_steps][B][4][HW], init_img / mask (inpainting blend of the sampler step, or
 * NULL), inpaint_image / inpaint_mask (9-channel UNet inputs, or NULL), table [T][8]; table_rows: HOST array, schedule row of step k.
 * use_graph != 0: captured on first use and replayed while the same buffers / scalars are passed.
 * The time embedding, time_embed MLP and FiLM vectors (unet.py:159-170) of ALL n_steps are computed by batched launches before the first
 * step (the handle keeps an [n_steps * B] x FiLM-width device buffer for them); every row is the arithmetic of the per-step launch, so the
 * loop equals n_steps calls of k22_unet_forward + k22_sampler_step bit for bit (env K22_HOIST_TIME=0: per-step launches, measurement only). */
int k22_unet_sample_loop(K22UNet* u, float* x, float* x_tmp, const float* timesteps, const float* noise_seq, const float* init_img,
                         const float* mask, const float* inpaint_image, const float* inpaint_mask, const float* table,
                         const int* table_rows, int n_steps, float guidance, float clamp_lo, float clamp_hi, int pct_index,
                         double pct_gamma, void* scratch, int use_graph, void* stream);
int k22_unet_num_ops(const K22UNet* u);
/* Tile configurations of the convolutions / GEMMs are chosen by measurement during the first k22_unet_forward
 * after k22_unet_plan (default on; env K22_AUTOTUNE=0 or k22_unet_set_autotune(u, 0) before planning = heuristics).
 * k22_unet_tuning_report writes a text table of the choices into buf. */
int k22_unet_set_autotune(K22UNet* u, int on);
int k22_unet_tuning_report(const K22UNet* u, char* buf, size_t cap);

/* Multi-GPU: one process per GPU, prompts sharded by rank, the packed weight arena of rank `root` broadcast ONCE
 * at start-up over the caller's RCCL communicator (ncclComm_t passed as void*), in <= 1 GiB pieces on `stream`; no collective
 * in the step loop.  Replaces nothing in the reference (it has no inference parallelism); mirrors what
 * kandinsky2_amd.parallel.broadcast_arena does through torch.distributed for hosts that own their communicator. */
int k22_comm_broadcast_weights(void* arena, size_t bytes, int root, void* nccl_comm, void* stream);

/* Tile table: the process-wide map  conv / GEMM problem -> tile configuration  that every engine consults BEFORE it
 * measures anything (csrc/tuning.h).  The package ships kandinsky-2_amd/tiles_gfx950.txt (measured on an MI355X for the
 * shapes of BASELINE.json's configs); the Python binding loads it when the library is opened, so those shapes get the
 * same configurations, and therefore the same bits, on every box.  Problems outside the table are measured on the
 * device at the first forward (K22_AUTOTUNE=0: fixed heuristic instead) and remembered for the life of the process;
 * env K22_TUNE_CACHE=<file> persists them.  load / save return the number of lines (< 0 = error), size the number of
 * entries, measured the number of entries this process timed itself. */
int k22_tile_table_load(const char* path);
int k22_tile_table_save(const char* path);
int k22_tile_table_size(void);
int k22_tile_table_measured(void);
void k22_tile_table_clear(void);

/* Measurement aid for bench.py: replays the planned forward EAGERLY `reps` times on `stream` with a HIP
 * event pair around every op, and reports per op class k (0 conv3x3, 1 GEMM, 2 GroupNorm, 3 attention,
 * 4 other): ms[k] = average summed device time per forward, flops[k]/bytes[k] = algorithmic work per
 * forward, launches[k] = kernel launches per forward.  Synchronises the stream (not a hot-path call). */
int k22_unet_profile(K22UNet* u, int reps, double* ms, double* flops, double* bytes, int* launches, void* stream);

/* ---- sampler step ---------------------------------------------------------------------------
 * Replaces one iteration of GaussianDiffusion.p_sample_loop_progressive
 * (kandinsky2/model/gaussian_diffusion.py:463-475): the CFG combine of Kandinsky2_1.generate_img.model_fn
 * (kandinsky2/kandinsky2_1_model.py:222-233) applied to the raw UNet output, p_mean_variance
 * (gaussian_diffusion.py:223-322) including the host-side np.percentile dynamic threshold
 * (:284-294, done on device here) and p_sample (:352-382).
 *   x, noise, x_out, x0_out(nullable): [N,4,H,W] fp32;  model_out: [N,8,H,W] fp32 (raw UNet output);
 *   init_img [N,4,H,W] / mask [N,1,H,W]: inpainting blend of denoised_fun (kandinsky2_1_model.py:238-240)
 *   or NULL;  table: device [steps][8] fp32 (see k22_sampler_table_columns); step_index selects the row;
 *   pct_index / pct_gamma: order-statistic index and weight of the 99.5 percentile (host computes them
 *   with numpy's own formula), pct_index < 0 disables thresholding (clip_denoised=False);
 *   scratch: device, >= k22_sampler_scratch_bytes(N,HW).
 */
size_t k22_sampler_scratch_bytes(int N, int HW);
/* DDIM step (DDIMSampler.p_sample_ddim, kandinsky2/model/samplers.py:290-331) with model_fn's guidance folded in:
 * e = u + g (c - u) on channels 0-3 of model_out [N][8][HW]; x0 = (x - sqrt(1-a_t) e) / sqrt(a_t);
 * x_out = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noise.   table_row: device fp32[4] =
 * (a_t, a_prev, sigma_t, sqrt(1 - a_t)); noise may be NULL (eta = 0); x0_out may be NULL. */
int k22_ddim_step(const float* x, const float* model_out, const float* noise, const float* table_row, float guidance, int use_cfg,
                  float* x_out, float* x0_out, int N, int HW, void* stream);
/* Inpainting mask pre-step (prepare_mask, kandinsky2/utils.py:11-31, an O(h*w) Python loop in the reference): every pixel
 * whose ORIGINAL value is not 1 zeroes its up / left / up-left / down / right / down-right neighbours.  mask, out: device fp32
 * [C][H][W], out of place (channel 0 decides, every channel is written). */
int k22_prepare_mask(const float* mask, float* out, int C, int H, int W, void* stream);
/* PLMS step (PLMSSampler.p_sample_plms, kandinsky2/model/samplers.py:566-637; eta = 0) with model_fn's guidance folded in:
 * e_t = u + g (c - u) of this model call; e' = Adams-Bashforth combination of e_t and the eps history by `order`
 * (0: e_t; 1: (3 e_t - h1)/2; 2: (23 e_t - 16 h1 + 5 h2)/12; 3: (55 e_t - 59 h1 + 37 h2 - 9 h3)/24; 4: (h1 + e_t)/2, the second
 * stage of the pseudo improved Euler start); x_out = DDIM update (sigma = 0) with e'.  eps_out (optional, [N][4][HW])
 * receives e_t for the caller's history ring; table_row as for k22_ddim_step. */
int k22_plms_step(const float* x, const float* model_out, const float* eps_hist1, const float* eps_hist2, const float* eps_hist3, int order,
                  const float* table_row, float guidance, int use_cfg, float* x_out, float* eps_out, float* x0_out, int N, int HW, void* stream);
int k22_sampler_step(const float* x, const float* model_out, const float* noise, const float* init_img,
                     const float* mask, const float* table, int step_index, float guidance, int use_cfg,
                     float clamp_lo, float clamp_hi, int pct_index, double pct_gamma, void* scratch,
                     float* x_out, float* x0_out, int N, int HW, void* stream);

/* ---- individual kernels (unit-parity surface; the engine calls the same launchers) -------------
 * dtype-typed buffers are bf16, fp16 or fp32 according to `dtype`.  Layouts: see the headers in kandinsky-2_amd/csrc. */
int k22_gemm(const void* A0, const void* A1, const void* Wp, const float* bias, const void* residual, void* out,
             void* partial, int M, int N, int Npad, int K0, int K1, long lda0, long lda1, int ldo, int ldr,
             int out_f32, int act, int splitk, int bm, int bn, int dtype, void* stream);
/* fp32 -> x3 chunks (K22_F16X3 operand format): dst[n] (4 bytes per element) from src[n] * scale, n % 4 == 0, 16-byte aligned.
 * scale = 256 for weights (what pack.py writes and the epilogues undo), 1 for activations.  With dtype == K22_F16X3 the kernel-level
 * entries take: k22_gemm / k22_gemm_gnstats / k22_qkv_project - A as plain fp32 rows (split at fragment-read time), W in x3 chunks;
 * k22_conv3x3* - x_padded AND W in x3 chunks (in the engine the GroupNorm-apply kernel writes the input so: k22_groupnorm with
 * dtype == K22_F16X3 = fp32 in, x3 chunks out), the fused skip operands as plain fp32 rows; k22_attention - fp32 in and out.
 * Outputs, residuals and biases are fp32. */
int k22_x3_pack(const float* src, void* dst, long n, float scale, void* stream);
int k22_conv3x3(const void* x_padded, const void* Wp, const float* bias, const void* residual, void* out,
                void* partial, int B, int H, int W, int Cin, int Cout, int Npad, int out_mode, int act, int splitk,
                int bm, int bn, int dtype, void* stream);
/* Tail of a channel-changing ResBlock in one launch (unet.py:191,219-220): out = conv3x3(x_padded) + bias +
 * [skip0 | skip1](unpadded NHWC, SK0 + SK1 channels) . Ws[Npad][SK0+SK1]^T + bias_s   (1x1 skip_connection fused
 * as a second K loop of the halo kernel; skip1 may be null with SK1 = 0). */
int k22_conv3x3_skip(const void* x_padded, const void* Wp, const float* bias, const void* skip0, const void* skip1,
                     int SK0, int SK1, const void* Ws, const float* bias_s, void* out, void* partial, int B, int H, int W,
                     int Cin, int Cout, int Npad, int splitk, int bm, int dtype, void* stream);
/* Same convolution (row-major T output, no activation) that also emits the per-channel partial sums the next
 * GroupNorm needs (ResBlock: conv -> GroupNorm32, unet.py:157-164/212-216), so the tensor is not re-read:
 * stats[row][c] = (sum, sum of squares) of the STORED outputs, image b owning rows [b*rpi, (b+1)*rpi);
 * *rows_per_image receives rpi.  Fails (K22_EINVAL) for configurations that cannot produce them. */
int k22_conv3x3_gnstats(const void* x_padded, const void* Wp, const float* bias, const void* residual, void* out,
                        void* partial, int B, int H, int W, int Cin, int Cout, int Npad, int splitk, int bm, int bn,
                        float* stats, int stats_capacity_rows, int* rows_per_image, int dtype, void* stream);
/* Same for a 1x1 convolution over unpadded rows [B*H*W][K] (AttentionBlock proj_out, unet.py:244-268, whose output the
 * next ResBlock's GroupNorm32 normalises): out = A W^T + bias (+ residual) through the 8-wave BM x 128 GEMM kernel
 * (bm = 256 / 128 / 0 = auto), with the per-tile (sum, sum of squares) rows as above. */
int k22_gemm_gnstats(const void* A, const void* Wp, const float* bias, const void* residual, void* out, void* partial,
                     int B, int H, int W, int N, int Npad, int K, int splitk, int bm, float* stats,
                     int stats_capacity_rows, int* rows_per_image, int dtype, void* stream);
/* Developer tool: runs the 256-row bf16 halo kernel once with s_memtime stamps; trace = device u64 [2][1024][4]
 * (wave 0 / wave 5 of workgroup 0; per tap: before the counted vmcnt wait, after it, after the barrier, after the last
 * MFMA was issued).  tools/conv_trace.py prints the per-phase cycle budget. */
int k22_debug_conv_trace(const void* x_padded, const void* Wp, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                         int Npad, unsigned long long* trace, void* stream);
/* Fused form of k22_groupnorm + k22_conv3x3 (round 4; north_star: "ResBlock conv3x3 / GroupNorm / SiLU fused with LDS-staged input tiles"):
 * out = conv3x3(act(GroupNorm32(cat(x0, x1)) [*(1 + scale) + shift])) + bias + residual, kandinsky2/model/nn.py:26-37 + unet.py:150-152,
 * 174-180, 212-216.  The convolution (conv3_halo_spec_kernel: algo 11 / 12) reads the RAW unpadded NHWC tensors; its producer waves apply
 * the per-channel coefficients (and the zero border) to the halo tile in the LDS right after the LDS-DMA lands: no normalised copy of the
 * activation is written or read.  x0 [B][H][W][C0] (+ x1 [B][H][W][C1]) of the engine's storage type (fp32 for K22_F16X3), Wp packed as
 * for k22_conv3x3, film as for k22_groupnorm; scratch >= k22_groupnorm_scratch_bytes(B, C0 + C1).  Same bits as the two-call form. */
int k22_conv3x3_gn(const void* x0, const void* x1, int C0, int C1, const float* gamma, const float* beta, const float* film, long film_ld,
                   float eps, int act, void* scratch, const void* Wp, const float* bias, const void* residual, void* out, void* partial,
                   int B, int H, int W, int Cout, int Npad, int splitk, int bm, int algo, int dtype, void* stream);
int k22_groupnorm(const void* x0, const void* x1, int C0, int C1, int B, int H, int W, const float* gamma,
                  const float* beta, const float* film, long film_ld, float eps, int act, int mode, int pad,
                  void* scratch, void* out, int dtype, void* stream);
size_t k22_groupnorm_scratch_bytes(int B, int C);
int k22_attention(const void* qkv, const void* ctxkv, void* kall, void* vtall, void* out, int B, int H, int T, int S,
                  int dtype, void* stream);
/* qkv projection of an AttentionBlock (Conv1d C -> 3C, unet.py:251) written straight into the attention operands:
 * x [B*T][K], packed weight rows ordered [q | k | v] x [H][64]; q -> q_out [B*T][C]; k -> kall[b][h][S+t][64];
 * v -> vtall[b][h][d][S+t] (both [..][Tkp = roundup(S+T,64)]; the first S keys belong to the context). */
int k22_qkv_project(const void* x, const void* Wp, const float* bias, void* q_out, void* kall, void* vtall,
                    int B, int H, int T, int S, int K, int bm, int bn, int dtype, void* stream);
/* The same projection on the weight-streaming small-M kernel (unet.py:251 at the 12x12 / 24x24 levels): partial = fp32
 * scratch [splitk][B*T][3C]; bm = 160 / 288. */
int k22_qkv_project_stream(const void* x, const void* Wp, const float* bias, void* q_out, void* kall, void* vtall, void* partial,
                           int B, int H, int T, int S, int K, int bm, int splitk, int dtype, void* stream);
int k22_linear_smallm(const float* x, const void* W, const float* bias, const float* add, float* out, int M, int N,
                      int K, int act_in, int act_out, int wdtype, void* stream);

/* ---- skinny-M weight-streaming GEMM family (round 6; csrc/skinny.hip) - unit-parity surface of the kernels the prior engine's
 * 16-bit path is made of.  Replaces nn.Linear c_qkv / c_proj / c_fc / mlp.c_proj (kandinsky2/model/prior.py:57-83), the LayerNorm in
 * front of each (prior.py:48-54, 105-127) and QKVMultiheadAttention (prior.py:86-102) for M = a few hundred token rows.  Both GEMM
 * operands are FRAGMENT-MAJOR (1 KB contiguous per 32-row x 16-k MFMA fragment): weights [Npad/32][K/64][4][64][8] written by
 * k22_stream_repack(W, out, Npad, 1, K, ...); activations [K/64][ceil(M/32)][4][64][8] written by k22_afrag_pack (row-major in), by
 * k22_finish_ln, by k22_skinny_gemm with epi = 1, or by k22_small_attention with out_frag = 1.  dtype: K22_BF16 / K22_F16.
 *   k22_skinny_gemm: epi 0 -> out[m*ldo + n] = act(A.W^T + bias) in T;  epi 1 -> the same values in the A-fragment order of a consumer
 *     whose K is this N;  epi 2 -> fp32 partial[z][m][n], z < splitk (no bias / activation);  (mt, nb) = m-atoms x n-atoms of 32 per
 *     workgroup, one of (6,1) (3,2) (3,1) (2,2) (2,1) (1,2); 0,0 = default.
 *   k22_finish_ln: x[m][:] += bias + sum_z partial[z][m][:] (skipped when partial == NULL), then yfrag = LayerNorm(x[m][:]) * gain +
 *     beta in T, A-fragment order (skipped when gain == NULL).  N <= 2048.
 *   k22_small_attention: qkv [B*T][3*H*64] = [Q | K | V] x [H][64] row-major T -> softmax(q.k / 8 + mask) v, T <= 128 tokens;
 *     mask = causal (key <= query) and key_valid [B][kv_n] (0 = padding key; keys >= kv_n are valid), as prior.py:262-263.
 *     qkv_partial != NULL: qkv is instead T(qkv_bias + sum_s qkv_partial[s][B*T][3*H*64]), s < nsplit <= 4 - the finish of a split-K
 *     c_qkv launch (k22_skinny_gemm epi 2) folded into the attention's staging loads. */
size_t k22_afrag_bytes(int M, int K);
int k22_afrag_pack(const void* A, long lda, void* out, int M, int K, int dtype, void* stream);
int k22_skinny_gemm(const void* Afrag, const void* Wfrag, const float* bias, void* out, float* partial, int M, int N, int Npad, int K,
                    int splitk, int epi, int act, int ldo, int mt, int nb, int dtype, void* stream);
int k22_finish_ln(const float* partial, int splitk, const float* bias, float* x, long ldx, const float* gain, const float* beta, void* yfrag,
                  int M, int N, float eps, int dtype, void* stream);
int k22_small_attention(const void* qkv, const float* qkv_partial, int nsplit, const float* qkv_bias, void* out, int out_frag, int B, int H, int T,
                        int causal, const float* key_valid, int kv_n, int dtype, void* stream);

/* ---- diffusion prior ----------------------------------------------------------------------------
 * Replaces PriorTransformer.forward (kandinsky2/model/prior.py:226-270) and the per-step update of
 * PriorDiffusionModel.forward's sampling loop (prior.py:336-384; gaussian_diffusion.py:223-322, 352-382 with
 * model_mean_type START_X, model_var_type FIXED_SMALL, clip_denoised False, denoised_fn clamp(+-10)).
 * Hyper-parameters mirror CONFIG_2_1["prior"]["params"]["model"]["hparams"] (kandinsky2/configs.py:101-111).
 * Weight names are the reference state_dict keys of PriorDiffusionModel.model (time_embed.0.weight ...,
 * transformer.resblocks.<l>.attn.c_qkv.weight ...); transformer Linear weights packed T [roundup(N,64)][K] with
 * c_qkv rows re-ordered to Q | K | V planes x [head][64]; everything else fp32; plus "time_freqs" [xf_width/2]. */
typedef struct K22PriorConfig {
  int dtype;          /* K22_BF16 | K22_F32 | K22_F16 */
  int text_ctx;       /* 77 */
  int xf_width;       /* 2048 */
  int xf_layers;      /* 20 */
  int xf_heads;       /* 32 (64 channels per head) */
  int xf_final_ln;    /* 1 */
  int clip_dim;       /* 768 */
  int clip_xf_width;  /* 768 */
} K22PriorConfig;
typedef struct K22Prior K22Prior;
int k22_prior_create(const K22PriorConfig* cfg, const K22Weight* weights, int n_weights, K22Prior** out);
void k22_prior_destroy(K22Prior* m);
int k22_prior_plan(K22Prior* m, int B, size_t* workspace_bytes);   /* B = 2*bs rows [cond | uncond], <= 8 */
int k22_prior_bind(K22Prior* m, void* workspace, size_t workspace_bytes);
int k22_prior_tuning_report(const K22Prior* m, char* buf, size_t cap);  /* as k22_unet_tuning_report */
/* x [B][clip_dim], timesteps [B] (as the reference passes them: original indices as floats), text_emb [B][clip_dim],
 * text_enc [B][text_ctx][clip_xf_width], key_valid [B][text_ctx] (1 = token, 0 = padding; the `mask` argument of
 * the reference as floats) -> out [B][clip_dim]; all fp32 device buffers. */
int k22_prior_forward(K22Prior* m, const float* x, const float* timesteps, const float* text_emb, const float* text_enc,
                      const float* key_valid, float* out, void* stream);
/* One ancestral step: x0 = clamp(uncond + scales[j]*(cond - uncond), +-clamp); mean = c1*x0 + c2*x;
 * x_out = mean + nonzero*exp(0.5*logvar)*noise.  table_row: device fp32[4] = (posterior_mean_coef1, coef2,
 * posterior_log_variance_clipped, t != 0); x / model_out / noise / x_out: [2*bs][D]; scales [bs]. */
int k22_prior_sampler_step(const float* x, const float* model_out, const float* noise, const float* scales, const float* table_row,
                           float clamp, float* x_out, int bs, int D, void* stream);

/* ---- MoVQ decoder ---------------------------------------------------------------------------
 * Replaces MOVQ.decode (kandinsky2/vqgan/autoencoder.py:182-185: post_quant_conv + MOVQDecoder.forward,
 * kandinsky2/vqgan/movq_modules.py:228-357: conv_in, mid ResnetBlock/AttnBlock/ResnetBlock, up levels with
 * SpatialNorm ResnetBlocks (+AttnBlocks at the lowest level), nearest-x2 Upsample convs, norm_out + swish +
 * conv_out) and the uint8 epilogue of process_images (kandinsky2/utils.py:57-70).  Hyper-parameters mirror
 * CONFIG_2_1["image_enc_params"]["params"]["ddconfig"] (kandinsky2/configs.py:75-86).
 * Weight names are the reference state_dict keys ("post_quant_conv.weight", "decoder.conv_in.weight",
 * "decoder.mid.block_1.norm1.norm_layer.weight", ".conv_y.weight" ...); 3x3 weights packed [Npad][ky][kx][Cin]
 * (conv_in zero-extended to Cin = 64), 1x1 weights [Npad][Cin], everything else fp32 as the reference. */
typedef struct K22MoVQConfig {
  int dtype;           /* K22_BF16 | K22_F32 | K22_F16 */
  int ch;              /* 128 */
  int n_levels;        /* len(ch_mult) */
  int ch_mult[8];      /* (1, 2, 2, 4) */
  int num_res_blocks;  /* 2  (the decoder runs num_res_blocks + 1 blocks per level) */
  int attn_levels;     /* bit i set = AttnBlock after every ResnetBlock of level i (resolution/2^i in attn_resolutions) */
  int z_channels;      /* 4 */
  int out_ch;          /* 3 */
} K22MoVQConfig;
typedef struct K22MoVQ K22MoVQ;
int k22_movq_create(const K22MoVQConfig* cfg, const K22Weight* weights, int n_weights, K22MoVQ** out);
void k22_movq_destroy(K22MoVQ* m);
/* latent [B][4][h][w]; the caller allocates *workspace_bytes (256-byte aligned) and binds it */
int k22_movq_plan(K22MoVQ* m, int B, int h, int w, size_t* workspace_bytes);
int k22_movq_bind(K22MoVQ* m, void* workspace, size_t workspace_bytes);
/* z: fp32 NCHW latent (un-quantised, as the sampler leaves it: decode(h, force_not_quantize) semantics);
 * out: fp32 NCHW [B][3][H][W] (may be null); out_u8: uint8 NHWC [B][H][W][3] = ((x+1)*127.5).round().clamp(0,255)
 * (may be null); H = h << (n_levels-1). */
int k22_movq_decode(K22MoVQ* m, const float* z, float* out, unsigned char* out_u8, void* stream);
/* MoVQ ENCODER on the same handle type (MOVQ.encode, kandinsky2/vqgan/autoencoder.py:176-180 = Encoder.forward,
 * vqgan_blocks.py:335-367, + quant_conv; the img2img / inpainting pre-step of kandinsky2_1_model.py:458-469, 519-534).
 * Weights: "encoder.*" (GroupNorm weight/bias fp32; 3x3 packed [Npad][ky][kx][Cin], conv_in zero-extended to Cin = 64, conv_out
 * rows padded to 64; 1x1 [Npad][Cin]) and fp32 "quant_conv.*".  image: fp32 NCHW [B][3][H][W]; latent: fp32 NCHW [B][4][H/8][W/8]
 * (not multiplied by the pipeline's latent scale).  A handle holds one plan: encoder or decoder. */
int k22_movq_plan_encoder(K22MoVQ* m, int B, int H, int W, size_t* workspace_bytes);
int k22_movq_encode(K22MoVQ* m, const float* image, float* latent, void* stream);
int k22_movq_num_ops(const K22MoVQ* m);

/* ---- conditioning encoders ---------------------------------------------------------------------
 * The transformer towers that run once per prompt / image, one engine type for the three of them:
 *   K22_ENC_CLIP_TEXT    clip_model.token_embedding / positional_embedding / transformer / ln_final / text_projection as
 *                        Kandinsky2_1.generate_clip_emb walks them (kandinsky2/kandinsky2_1_model.py:159-168; OpenAI clip
 *                        model.py CLIP.encode_text): seq_out = ln_final(x) [B][n_ctx][width], pooled_out = seq_out[b][argmax
 *                        (tokens[b])] @ text_projection [B][out_dim]
 *   K22_ENC_CLIP_VISION  clip_model.encode_image (kandinsky2_1_model.py:177-181; clip model.py VisionTransformer.forward):
 *                        image [B][3][image_size][image_size] (already preprocessed) -> pooled_out [B][out_dim]
 *   K22_ENC_XLMR         MultilingualCLIP.forward (kandinsky2/model/text_encoders.py:108-122) = transformers XLMRobertaModel +
 *                        masked mean + LinearTransformation: seq_out = last_hidden_state [B][n_ctx][width], pooled_out [B][out_dim]
 * Weights (names are the engine's; kandinsky-2_amd/encoders.py maps the reference / OpenAI-clip / transformers state_dict keys):
 *   fp32: "token_embedding" [vocab][width], "positional_embedding" [n_ctx | max_pos][width], xlmr "token_type_embedding" [width],
 *   "embeddings_ln.*"; vision "class_embedding" [width], "ln_pre.*", "ln_post.*"; text "ln_final.*"; "layers.<l>.ln_1.*",
 *   "layers.<l>.ln_2.*", every Linear bias; "head.weight" [out_dim][width] (text_projection^T / visual.proj^T /
 *   LinearTransformation.weight), xlmr "head.bias".
 *   T (engine dtype), rows padded to 64: "layers.<l>.qkv.weight" [3*width][width] (rows Q | K | V, heads x 64 inside each),
 *   ".proj.weight", ".fc.weight" [4*width][width], ".out.weight" [width][4*width]; vision "patch.weight" [width][roundup(3*patch^2,
 *   64)] (conv1.weight flattened (c, i, j), zero padded).
 * The Linears' tile configurations come from the tile table like the other engines' (shipped for the production shapes). */
enum { K22_ENC_CLIP_TEXT = 0, K22_ENC_CLIP_VISION = 1, K22_ENC_XLMR = 2 };
typedef struct K22EncoderConfig {
  int dtype;       /* K22_BF16 | K22_F32 | K22_F16 */
  int kind;        /* K22_ENC_* */
  int width;       /* 768 / 1024 / 1024 (64 channels per head: the UNet's flash attention kernel); the vision tower also takes any
                      width % heads == 0 with width / heads <= 128 - CLIP ViT-bigG/14 of Kandinsky 2.2 (kandinsky2_2_model.py:24): 1664 / 16
                      = 104 per head - on a small-sequence attention kernel of its own (encoder.hip: enc_attention_generic_kernel) */
  int layers;      /* 12 / 24 / 24 */
  int heads;       /* 12 / 16 / 16 */
  int n_ctx;       /* tokens per sequence: 77 / 257 / 77 */
  int vocab;       /* 49408 / 0 / 250002 */
  int out_dim;     /* 768 */
  int image_size;  /* vision: 224 */
  int patch;       /* vision: 14 */
  int max_pos;     /* xlmr: 514 rows of position embeddings */
  int pad_id;      /* xlmr: 1 (padding token id = position-id offset) */
  float ln_eps;    /* 1e-5 */
  int mlp_dim;     /* 0 = 4 * width (OpenAI CLIP, XLM-R); CLIP ViT-bigG/14: 8192 at width 1664 */
  int hidden_act;  /* CLIP towers: 0 = QuickGELU (OpenAI CLIP), 1 = exact erf GELU (open_clip bigG: "hidden_act": "gelu") */
} K22EncoderConfig;
typedef struct K22Encoder K22Encoder;
int k22_encoder_create(const K22EncoderConfig* cfg, const K22Weight* weights, int n_weights, K22Encoder** out);
void k22_encoder_destroy(K22Encoder* m);
int k22_encoder_plan(K22Encoder* m, int B, size_t* workspace_bytes);     /* B sequences / images per call, <= 8 */
int k22_encoder_bind(K22Encoder* m, void* workspace, size_t workspace_bytes);
/* tokens: int32 [B][n_ctx] (text towers); key_valid: fp32 [B][n_ctx], 1 = token, 0 = padding (xlmr: the attention_mask);
 * image: fp32 [B][3][S][S] (vision); seq_out: fp32 [B][n_ctx][width] or null; pooled_out: fp32 [B][out_dim].  Device buffers. */
int k22_encoder_forward(K22Encoder* m, const int* tokens, const float* key_valid, const float* image, float* seq_out, float* pooled_out,
                        void* stream);

/* ---- img2img / inpainting latent arithmetic ----------------------------------------------------
 * out = mask * (sa*init + sb*noise) + (1 - mask) * x   [N][C][HW] fp32; mask [N][1][HW] (1 = keep the known region).
 * mask == NULL (x ignored): out = sa*init + sb*noise = DDPMScheduler.add_noise / q_sample (kandinsky2/utils.py:43-54) with
 * sa = sqrt(alphas_cumprod[t]), sb = sqrt(1 - alphas_cumprod[t]); noise == NULL: the un-noised init.  With a mask it is the
 * per-step re-imposition of the known region in the Kandinsky 2.2 inpainting pipeline (kandinsky2/kandinsky2_2_model.py:150-173 ->
 * diffusers KandinskyV22InpaintPipeline).  broadcast_first != 0: init / noise / mask are those of image 0 for every n. */
int k22_blend_noised(const float* x, const float* init, const float* noise, const float* mask, float sa, float sb, float* out,
                     int N, int C, int HW, int broadcast_first, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* K22_H */
