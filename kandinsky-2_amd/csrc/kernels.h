// k22 — internal kernel launch API (host side).  Every launcher enqueues on `stream`, allocates
// nothing and returns 0 or a negative K22_E* code (message via k22_last_error()).
#pragma once
#include "common.h"

#include <atomic>

#define K22_OK 0
#define K22_EINVAL (-1)
#define K22_EHIP (-2)
#define K22_ENOMEM (-3)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: one guard per kernel instantiation holds a
// bit per device ordinal, so a process that drives several GPUs (or launches from several threads) sets it on each.
struct LdsAttrGuard { std::atomic<unsigned long long> done{0}; };
inline int k22_ensure_lds_attr(LdsAttrGuard& g, const void* fn, int bytes, const char* file, int line) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (g.done.load(std::memory_order_acquire) & bit) return K22_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return k22_set_error_hip(e, file, line);
  g.done.fetch_or(bit, std::memory_order_release);
  return K22_OK;
}

// ---------------------------------------------------------------------------------------
// Implicit GEMM:  out[m][n] = sum_k A(m,k) * Wp[n][k]  (+bias[n]) (+residual[m][n]) -> act
//   taps == 1 : plain GEMM, A row m = A0[m*lda0 + k] for k < K0, else A1[m*lda1 + (k-K0)]
//               ("virtual concat" of two row-major operands; A1 may be null when K0 == Kc).
//   taps == 9 : 3x3 convolution, stride 1, pad 1 over a ZERO-BORDERED NHWC input
//               A0 = [B][H+2][W+2][Kc]; m = (b*H + y)*W + x; k = tap*Kc + c, tap = ky*3+kx.
//   Wp is the packed weight [Npad][taps*Kc] (K contiguous), Npad = roundup(N, 64), zero rows.
// ---------------------------------------------------------------------------------------
enum IgemmOut { IG_OUT_ROWMAJOR = 0,  // T out[m*ldo + n]
                IG_OUT_ROWMAJOR_F32 = 1,  // float out[m*ldo + n]
                IG_OUT_NCHW_F32 = 2,      // float out[((b*N + n)*H + y)*W + x]   (m = (b*H+y)*W+x)
                IG_OUT_QKV = 3 };         // qkv projection (N = 3C, columns [q | k | v] x [heads][64]) written straight into
                                          // the attention operands: q -> T out[m*ldo + n];  k -> kall[b][head][S+t][d];
                                          // v -> vtall[b][head][d][S+t]   (m = b*T + t; see AttentionParams)

struct IgemmParams {
  const void* A0;
  const void* A1;
  const void* Wp;
  const float* bias;     // [N] or null
  const void* residual;  // T [M][ldr] or null (fp32 [M][ldr] when res_f32 is set: fp32 residual stream of the prior)
  void* out;
  float* partial;        // split-K scratch [splitk][M][N] fp32 (needed when splitk > 1)
  int M, N, Npad;
  int Kc;                // K elements per tap (K0 + K1 in concat mode)
  int K0;                // channels served by A0 (== Kc when A1 is null)
  int taps;              // 1 or 9
  int H, W;              // conv geometry (taps == 9) and NCHW store geometry
  long lda0, lda1;       // row strides (elements) in GEMM mode
  int ldo, ldr;
  int out_mode;          // IgemmOut
  int act;               // K22Act applied after bias+residual
  int splitk;            // >= 1 (0 = let the launcher choose)
  int force_bm, force_bn;  // 0 = heuristic
  int stages;            // 2..4 = LDS-DMA pipeline depth, anything else = default (env K22_IGEMM_STAGES, else 2)
  int xcd_remap;         // set by the launcher: XCD-aware block renumbering on/off
  int algo;              // taps == 9 only: 0 auto, 1 generic implicit GEMM, 2 LDS-resident halo kernel (conv3_halo.hip)
  // optional fused 1x1 "skip_connection" of a ResBlock (unet.py:191), halo kernel only:
  //   out += [S0 | S1](unpadded NHWC rows m, SK0 + SK1 channels) . Ws[n][SK0+SK1]^T + bias2[n]
  const void* S0; const void* S1; const void* Ws; const float* bias2;
  int SK0, SK1;
  void* kall; void* vtall;                 // IG_OUT_QKV only
  int att_T, att_S, att_Tkp;               // IG_OUT_QKV only: tokens per image, context keys, padded key count
  unsigned long long* trace;  // developer tool (k22_debug_conv_trace): per-tap s_memtime stamps of two waves of block 0
  int res_f32;           // residual is fp32 (generic kernel / split-K finish only)
  float* stats;          // optional GroupNorm side output: per-row-block, per-channel (sum, sumsq) of the STORED
                         // values, [stats_rows][N][2] fp32 (see IgemmStatsInfo); null = not wanted
  // weight-streaming kernel (stream_gemm.hip): the weights again in FRAGMENT-MAJOR order - [n-block of 32 rows][(slab, tap) item][k quarter]
  // [lane][8 elements] = 1 KB contiguous per MFMA B fragment - written once by launch_stream_repack (Wsfrag: the fused 1x1 skip's weights);
  // null = fragments are gathered from the row-major Wp / Ws (32 bytes of 32 different lines per load: measured L1-lookup bound)
  const void* Wfrag; const void* Wsfrag;
  int a_raw;             // K22_F16X3 only: 1 = the A operand (A0 / A1) is plain fp32 rows, converted to split halves at fragment-read
                         // time (generic kernel and gemm8 only); 0 = A is in x3 chunks (common.h), written so by its producer.
                         // The fused-skip operands S0 / S1 are always plain fp32 rows; weights are always x3 chunks.
  // Fused GroupNorm-apply (conv3_halo_spec_kernel, taps == 9): the input is NOT the zero-bordered, normalised tensor A0 but the raw
  // UNPADDED NHWC tensor(s) gn_x0 [B][H][W][gn_C0] (+ gn_x1 [B][H][W][Kc - gn_C0]: virtual concat); the producer waves read them
  // into the LDS halo image, and rewrite it in place as  act(x * A[c] + Bc[c])  from gn_coeff [B][Kc][2] (gn_coeff_kernel's table:
  // mean / rstd, gamma / beta and FiLM folded), zero at the border positions - GroupNorm32 + FiLM + SiLU of nn.py:26-37,
  // unet.py:150-152, 174-180, 212-216 without the stand-alone apply pass.  gn_coeff == null: off (A0 is read).
  const float* gn_coeff; const void* gn_x0; const void* gn_x1; int gn_C0, gn_act;
  unsigned long long* st_trace;          // K22_STREAM_DEBUG builds only: 16 stamps per workgroup
  int st_tm, st_rb, st_mtiles, st_buf;   // set by launch_stream (stream_gemm.hip): rows per m-tile, image rows per band, m-tiles, bytes of one LDS A buffer
};

// How a producer laid out its GroupNorm partial sums: image b owns rows [b*rows_per_image, (b+1)*rows_per_image).
struct IgemmStatsInfo { int rows_per_image; };
// Number of stats rows per image launch_igemm will write for this problem (0 = this configuration cannot
// produce stats; the caller must run the stand-alone gn_stats kernel instead).
int igemm_stats_rows_per_image(const IgemmParams& p, int dtype);

int launch_igemm(const IgemmParams& p, int dtype, hipStream_t stream);
// conv3_halo.hip: 3x3 convolution with the input tile (+halo) resident in LDS across the 9 taps.
bool conv3_halo_supported(const IgemmParams& p, int dtype, int bm);
// the kernel variants that can take the fused GroupNorm-apply input (IgemmParams::gn_coeff): the specialised halo kernels
inline bool conv3_algo_fuses_gn(int algo) { return algo == 11 || algo == 12; }
int conv3_halo_tiles_per_image(const IgemmParams& p, int bm);
int launch_conv3_halo(const IgemmParams& p, int dtype, int bm, int splitk, hipStream_t stream);
int launch_conv3_halo_trace(const IgemmParams& p, int dtype, hipStream_t stream);
// 8-wave BM x 128 GEMM on the halo kernels' frame (conv3_halo.hip: gemm8_kernel); p.algo == 10 selects it in launch_igemm
bool gemm8_supported(const IgemmParams& p, int dtype, int bm);
bool gemm8_spec_supported(int dtype);   // p.stages == 3: gemm8_spec_kernel (producer / consumer waves, explicit fragment pipeline)
int gemm8_tiles_per_image(const IgemmParams& p, int bm);
int launch_gemm8(const IgemmParams& p, int dtype, int bm, int splitk, hipStream_t stream);
// stream_gemm.hip: weight-streaming kernel for small M (p.algo == 20; bm = 160 / 288 selects 5 / 9 m-blocks per workgroup).
// It leaves fp32 partial tiles [splitk][M][N] in p.partial (also for splitk == 1); launch_igemm runs the split-K finish.
bool stream_supported(const IgemmParams& p, int dtype, int mb);
int stream_mtiles(const IgemmParams& p, int mb);
int launch_stream(const IgemmParams& p, int dtype, int mb, int splitk, hipStream_t stream);
long stream_launch_count();
// bytes of the fragment-major copy of a [Npad][taps * Kc] weight matrix, and the one-time repack (any 16-bit dtype)
size_t stream_frag_bytes(int Npad, int taps, int Kc, int dtype);
int launch_stream_repack(const void* W, void* out, int Npad, int taps, int Kc, int dtype, hipStream_t stream);
void igemm_set_gemm_algo(int v);
void igemm_set_conv_algo(int v);       // tuning knob: 0 auto, 1 generic, 2 halo
void igemm_set_default_stages(int v);  // tuning knob: 2..4 LDS-DMA stages, -1 env/default
void igemm_set_xcd_remap(int v);       // tuning knob: XCD-aware block renumbering (default on)
int igemm_choose_splitk(const IgemmParams& p, int dtype);  // split-K factor the heuristic picks (scratch = splitk*M*N*4 B)
