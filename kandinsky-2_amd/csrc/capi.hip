// k22 — C ABI glue: error reporting and the kernel-level entry points declared in include/k22.h.
#include "kernels.h"
#include "elementwise.h"
#include "../../include/k22.h"
#include "tuning.h"
#include "skinny.h"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int k22_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}
int k22_set_error_hip(hipError_t e, const char* file, int line) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
  return K22_EHIP;
}

extern "C" {

int k22_version(void) { return 100; }
int k22_build_flags(void) {
  int f = 0;
#ifndef K22_NOPK
  f |= K22_BUILD_PACKED_FP32;
#endif
#if defined(K22_DEBUG_VARIANTS) || defined(K22_STREAM_DEBUG) || defined(K22_SKINNY_DEBUG)
  f |= K22_BUILD_DEBUG_VARIANTS;
#endif
  return f;
}

int k22_set_option(const char* name, int value) {
  if (name && !strcmp(name, "igemm_stages")) { igemm_set_default_stages(value); return K22_OK; }
  if (name && !strcmp(name, "igemm_xcd_remap")) { igemm_set_xcd_remap(value); return K22_OK; }
  if (name && !strcmp(name, "conv_algo")) { igemm_set_conv_algo(value); return K22_OK; }
  if (name && !strcmp(name, "gemm_algo")) { igemm_set_gemm_algo(value); return K22_OK; }
  return k22_set_error(K22_EINVAL, "k22_set_option: unknown option");
}
const char* k22_last_error(void) { return g_err; }

// Test / bench surface of the weight-streaming kernel's fragment-major weights: k22_stream_repack writes the copy, and the pointers
// given to k22_debug_set_stream_frag ride along with every LATER kernel-level test entry of this file (k22_gemm, k22_conv3x3*,
// k22_qkv_project_stream) until they are reset to NULL.  The engines own their copies themselves (engine.hip); nothing in the product
// path reads these globals.
#ifdef K22_STREAM_DEBUG
static unsigned long long* g_dbg_trace = nullptr;
extern "C" int k22_debug_set_stream_trace(unsigned long long* t) { g_dbg_trace = t; return K22_OK; }
#define K22_DBG_TRACE(p) (p).st_trace = g_dbg_trace
#else
#define K22_DBG_TRACE(p) (void)0
#endif
static const void* g_dbg_wfrag = nullptr;
static const void* g_dbg_wsfrag = nullptr;
static void* g_dbg_scratch = nullptr;      // "repack on every call" mode for the parity tests (their helpers pack the weights themselves)
static size_t g_dbg_scratch_bytes = 0;
static int dbg_frag(IgemmParams& p, int dtype, hipStream_t st) {
  K22_DBG_TRACE(p);
  p.Wfrag = g_dbg_wfrag; p.Wsfrag = g_dbg_wsfrag;
  if (g_dbg_wfrag || !g_dbg_scratch || dtype == K22_F32 || (p.taps != 1 && p.taps != 9) || p.Kc % 64 || p.Npad % 64) return K22_OK;
  const size_t n1 = stream_frag_bytes(p.Npad, p.taps, p.Kc, dtype);
  const bool skip = p.S0 != nullptr && p.Ws != nullptr && (p.SK0 + p.SK1) % 64 == 0;
  const size_t n2 = skip ? stream_frag_bytes(p.Npad, 1, p.SK0 + p.SK1, dtype) : 0;
  if (n1 + n2 > g_dbg_scratch_bytes) return k22_set_error(K22_ENOMEM, "debug stream scratch too small");
  if (int rc = launch_stream_repack(p.Wp, g_dbg_scratch, p.Npad, p.taps, p.Kc, dtype, st)) return rc;
  p.Wfrag = g_dbg_scratch;
  if (skip) {
    void* w2 = static_cast<char*>(g_dbg_scratch) + n1;
    if (int rc = launch_stream_repack(p.Ws, w2, p.Npad, 1, p.SK0 + p.SK1, dtype, st)) return rc;
    p.Wsfrag = w2;
  }
  return K22_OK;
}
#define K22_DBG_FRAG(p) do { if (int rc_ = dbg_frag((p), dtype, reinterpret_cast<hipStream_t>(stream))) return rc_; } while (0)
int k22_debug_set_stream_frag(const void* wfrag, const void* wsfrag) { g_dbg_wfrag = wfrag; g_dbg_wsfrag = wsfrag; return K22_OK; }
int k22_debug_set_stream_scratch(void* scratch, size_t bytes) { g_dbg_scratch = scratch; g_dbg_scratch_bytes = scratch ? bytes : 0; return K22_OK; }
size_t k22_stream_frag_bytes(int Npad, int taps, int Kc, int dtype) { return stream_frag_bytes(Npad, taps, Kc, dtype); }
int k22_stream_repack(const void* W, void* out, int Npad, int taps, int Kc, int dtype, void* stream) {
  return launch_stream_repack(W, out, Npad, taps, Kc, dtype, reinterpret_cast<hipStream_t>(stream));
}

long k22_debug_counter(const char* name) {
  if (name && !strcmp(name, "stream_launches")) return stream_launch_count();
  return -1;
}

// ---- multi-GPU: the ONE collective of a job -------------------------------------------------------------
// Broadcast of the packed weight arena from `root` over an RCCL communicator the caller owns (one process per GPU; prompts are
// sharded by rank and nothing is exchanged in the step loop).  ncclBroadcast is resolved at call time from the RCCL that is
// already in the process (the one the communicator was created with - PyTorch's, or the host program's), else from
// librccl.so.1: libk22hip.so itself carries no link-time dependency on a second RCCL copy.  Sent in <= 1 GiB pieces: xGMI rings
// are per-link bound, large messages amortise the latency and keep any one call below RCCL's int-count limits.
int k22_comm_broadcast_weights(void* arena, size_t bytes, int root, void* nccl_comm, void* stream) {
  if (!arena || !nccl_comm) return k22_set_error(K22_EINVAL, "k22_comm_broadcast_weights: null arena / communicator");
  typedef int (*bcast_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, int, void*, hipStream_t);
  static bcast_fn fn = nullptr;
  if (!fn) {
    void* sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");
    if (!sym) {
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) sym = dlsym(h, "ncclBroadcast");
    }
    if (!sym) return k22_set_error(K22_EINVAL, "k22_comm_broadcast_weights: RCCL (ncclBroadcast) is not loadable in this process");
    fn = reinterpret_cast<bcast_fn>(sym);
  }
  const size_t piece = (size_t)1 << 30;
  char* base = reinterpret_cast<char*>(arena);
  for (size_t off = 0; off < bytes; off += piece) {
    const size_t n = bytes - off < piece ? bytes - off : piece;
    const int rc = fn(base + off, base + off, n, /*ncclUint8*/ 1, root, nccl_comm, reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) { char msg[96]; snprintf(msg, sizeof msg, "k22_comm_broadcast_weights: ncclBroadcast returned %d", rc); return k22_set_error(K22_EHIP, msg); }
  }
  return K22_OK;
}

// ---- tile table (tuning.h) ---------------------------------------------------------------------------------------
int k22_tile_table_load(const char* path) {
  if (!path) return k22_set_error(K22_EINVAL, "k22_tile_table_load: null path");
  const int n = tile_table_load_file(path);
  if (n < 0) return k22_set_error(K22_EINVAL, "k22_tile_table_load: cannot read the file");
  return n;
}
int k22_tile_table_save(const char* path) {
  if (!path) return k22_set_error(K22_EINVAL, "k22_tile_table_save: null path");
  const int n = tile_table_save_file(path);
  if (n < 0) return k22_set_error(K22_EINVAL, "k22_tile_table_save: cannot write the file");
  return n;
}
int k22_tile_table_size(void) {
  TileTable& tt = tile_table();
  std::lock_guard<std::mutex> lk(tt.mu);
  return (int)tt.m.size();
}
int k22_tile_table_measured(void) {
  TileTable& tt = tile_table();
  std::lock_guard<std::mutex> lk(tt.mu);
  return (int)tt.measured;
}
void k22_tile_table_clear(void) {
  TileTable& tt = tile_table();
  std::lock_guard<std::mutex> lk(tt.mu);
  tt.m.clear();
  tt.measured = 0;
}

int k22_gemm(const void* A0, const void* A1, const void* Wp, const float* bias, const void* residual, void* out,
             void* partial, int M, int N, int Npad, int K0, int K1, long lda0, long lda1, int ldo, int ldr,
             int out_f32, int act, int splitk, int bm, int bn, int dtype, void* stream) {
  IgemmParams p = {};
    p.stages = -1;
  p.A0 = A0; p.A1 = A1; p.Wp = Wp; p.bias = bias; p.residual = residual; p.out = out;
  p.partial = reinterpret_cast<float*>(partial);
  p.M = M; p.N = N; p.Npad = Npad; p.Kc = K0 + K1; p.K0 = K0; p.taps = 1; p.lda0 = lda0; p.lda1 = lda1;
  p.ldo = ldo; p.ldr = ldr; p.out_mode = out_f32 ? IG_OUT_ROWMAJOR_F32 : IG_OUT_ROWMAJOR; p.act = act;
  p.splitk = splitk; p.force_bm = bm; p.force_bn = bn;
  p.a_raw = k22_is_split(dtype) ? 1 : 0;   // unit entry: A0 / A1 are plain fp32 rows (the engine feeds x3 chunks where its own kernels produce A)
  if (p.splitk == 0) p.splitk = partial ? igemm_choose_splitk(p, dtype) : 1;
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_x3_pack(const float* src, void* dst, long n, float scale, void* stream) {
  if (!src || !dst || n < 0) return k22_set_error(K22_EINVAL, "x3_pack: null argument");
  return launch_x3_pack(src, dst, n, scale, reinterpret_cast<hipStream_t>(stream));
}

int k22_conv3x3(const void* x_padded, const void* Wp, const float* bias, const void* residual, void* out,
                void* partial, int B, int H, int W, int Cin, int Cout, int Npad, int out_mode, int act, int splitk,
                int bm, int bn, int dtype, void* stream) {
  IgemmParams p = {};
    p.stages = -1;
  p.A0 = x_padded; p.Wp = Wp; p.bias = bias; p.residual = residual; p.out = out;
  p.partial = reinterpret_cast<float*>(partial);
  p.M = B * H * W; p.N = Cout; p.Npad = Npad; p.Kc = Cin; p.K0 = Cin; p.taps = 9; p.H = H; p.W = W;
  p.ldo = Cout; p.ldr = Cout; p.out_mode = out_mode; p.act = act; p.splitk = splitk; p.force_bm = bm; p.force_bn = bn;
  if (p.splitk == 0) p.splitk = partial ? igemm_choose_splitk(p, dtype) : 1;
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_conv3x3_gnstats(const void* x_padded, const void* Wp, const float* bias, const void* residual, void* out,
                        void* partial, int B, int H, int W, int Cin, int Cout, int Npad, int splitk, int bm, int bn,
                        float* stats, int stats_capacity_rows, int* rows_per_image, int dtype, void* stream) {
  IgemmParams p = {};
  p.stages = -1;
  p.A0 = x_padded; p.Wp = Wp; p.bias = bias; p.residual = residual; p.out = out;
  p.partial = reinterpret_cast<float*>(partial);
  p.M = B * H * W; p.N = Cout; p.Npad = Npad; p.Kc = Cin; p.K0 = Cin; p.taps = 9; p.H = H; p.W = W;
  p.ldo = Cout; p.ldr = Cout; p.out_mode = IG_OUT_ROWMAJOR; p.act = K22_ACT_NONE; p.splitk = splitk; p.force_bm = bm; p.force_bn = bn;
  if (p.splitk == 0) p.splitk = partial ? igemm_choose_splitk(p, dtype) : 1;
  const int rpi = igemm_stats_rows_per_image(p, dtype);
  if (rows_per_image) *rows_per_image = rpi;
  if (rpi <= 0) return k22_set_error(K22_EINVAL, "conv3x3_gnstats: this configuration cannot produce GroupNorm partial sums");
  if (rpi * B > stats_capacity_rows) return k22_set_error(K22_ENOMEM, "conv3x3_gnstats: stats buffer too small");
  p.stats = stats;
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_gemm_gnstats(const void* A, const void* Wp, const float* bias, const void* residual, void* out, void* partial,
                     int B, int H, int W, int N, int Npad, int K, int splitk, int bm, float* stats,
                     int stats_capacity_rows, int* rows_per_image, int dtype, void* stream) {
  IgemmParams p = {};
  p.stages = -1;
  p.A0 = A; p.Wp = Wp; p.bias = bias; p.residual = residual; p.out = out; p.partial = reinterpret_cast<float*>(partial);
  p.M = B * H * W; p.N = N; p.Npad = Npad; p.Kc = K; p.K0 = K; p.taps = 1; p.H = H; p.W = W; p.lda0 = K;
  p.ldo = N; p.ldr = N; p.out_mode = IG_OUT_ROWMAJOR; p.act = K22_ACT_NONE; p.splitk = splitk > 0 ? splitk : 1;
  p.force_bm = bm; p.algo = (bm == 160 || bm == 288) ? 20 : 10;
  p.a_raw = k22_is_split(dtype) ? 1 : 0;
  const int rpi = igemm_stats_rows_per_image(p, dtype);
  if (rows_per_image) *rows_per_image = rpi;
  if (rpi <= 0) return k22_set_error(K22_EINVAL, "gemm_gnstats: this configuration cannot produce GroupNorm partial sums");
  if (rpi * B > stats_capacity_rows) return k22_set_error(K22_ENOMEM, "gemm_gnstats: stats buffer too small");
  p.stats = stats;
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_conv3x3_skip(const void* x_padded, const void* Wp, const float* bias, const void* skip0, const void* skip1,
                     int SK0, int SK1, const void* Ws, const float* bias_s, void* out, void* partial, int B, int H, int W,
                     int Cin, int Cout, int Npad, int splitk, int bm, int dtype, void* stream) {
  IgemmParams p = {};
  p.stages = -1;
  p.A0 = x_padded; p.Wp = Wp; p.bias = bias; p.out = out; p.partial = reinterpret_cast<float*>(partial);
  p.M = B * H * W; p.N = Cout; p.Npad = Npad; p.Kc = Cin; p.K0 = Cin; p.taps = 9; p.H = H; p.W = W;
  p.ldo = Cout; p.ldr = Cout; p.out_mode = IG_OUT_ROWMAJOR; p.act = K22_ACT_NONE; p.splitk = splitk; p.force_bm = bm; p.force_bn = 0;
  p.S0 = skip0; p.S1 = skip1; p.SK0 = SK0; p.SK1 = SK1; p.Ws = Ws; p.bias2 = bias_s; p.algo = 0;  /* "conv_algo" option 3 selects the 64-byte-row halo kernel */
  if (p.splitk == 0) p.splitk = partial ? igemm_choose_splitk(p, dtype) : 1;
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_debug_conv_trace(const void* x_padded, const void* Wp, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                         int Npad, unsigned long long* trace, void* stream) {
  IgemmParams p = {};
  p.stages = -1;
  p.A0 = x_padded; p.Wp = Wp; p.bias = bias; p.out = out;
  p.M = B * H * W; p.N = Cout; p.Npad = Npad; p.Kc = Cin; p.K0 = Cin; p.taps = 9; p.H = H; p.W = W;
  p.ldo = Cout; p.ldr = Cout; p.out_mode = IG_OUT_ROWMAJOR; p.act = K22_ACT_NONE; p.splitk = 1; p.trace = trace;
  return launch_conv3_halo_trace(p, K22_BF16, reinterpret_cast<hipStream_t>(stream));
}

size_t k22_groupnorm_scratch_bytes(int B, int C) {
  return (size_t)B * 128 * C * 2 * sizeof(float) + (size_t)B * C * 2 * sizeof(float) + 512;
}

int k22_groupnorm(const void* x0, const void* x1, int C0, int C1, int B, int H, int W, const float* gamma,
                  const float* beta, const float* film, long film_ld, float eps, int act, int mode, int pad,
                  void* scratch, void* out, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int C = C0 + C1, HW = H * W;
  const int out_x3 = k22_is_split(dtype) ? 1 : 0;   // split-precision engine: fp32 in, x3 chunks out (what its convolutions read)
  dtype = k22_storage_dtype(dtype);
  const int nsplit = gn_nsplit(B, HW);
  float* partial = reinterpret_cast<float*>(scratch);
  float* coeff = partial + (size_t)B * 128 * C * 2;
  GnStatsParams sp;
  sp.x0 = x0; sp.x1 = x1; sp.C0 = C0; sp.C1 = C1; sp.HW = HW; sp.B = B; sp.groups = 32; sp.nsplit = nsplit; sp.partial = partial;
  int rc = launch_gn_stats(sp, dtype, st);
  if (rc) return rc;
  GnCoeffParams cp = {};
  cp.src[0].st = partial; cp.src[0].rpi = nsplit; cp.src[0].C = C; cp.src[1].st = nullptr; cp.src[1].rpi = 0; cp.src[1].C = 0;
  cp.HW = HW; cp.C = C; cp.groups = 32; cp.eps = eps; cp.gamma = gamma; cp.beta = beta;
  cp.film = film; cp.film_ld = film_ld; cp.coeff = coeff;
  GnApplyParams ap = {};
  ap.x0 = x0; ap.x1 = x1; ap.C0 = C0; ap.C1 = C1; ap.B = B; ap.H = H; ap.W = W; ap.mode = mode; ap.pad = pad; ap.act = act;
  ap.coeff = coeff; ap.out = out; ap.out_x3 = out_x3;
  rc = launch_gn_coeff(cp, B, st);
  if (rc) return rc;
  return launch_gn_apply(ap, dtype, st);
}

// conv3x3(GroupNorm32(cat(x0, x1)) [*(1 + scale) + shift] -> act) with the GroupNorm-apply FUSED into the convolution's halo fill
// (conv3_halo_spec_kernel, IgemmParams::gn_coeff): statistics pass + coefficient kernel, then ONE convolution launch that reads the raw
// tensors.  algo = 11 / 12 (the specialised kernels).  scratch: k22_groupnorm_scratch_bytes(B, C0 + C1).
int k22_conv3x3_gn(const void* x0, const void* x1, int C0, int C1, const float* gamma, const float* beta, const float* film, long film_ld,
                   float eps, int act, void* scratch, const void* Wp, const float* bias, const void* residual, void* out, void* partial,
                   int B, int H, int W, int Cout, int Npad, int splitk, int bm, int algo, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int C = C0 + C1, HW = H * W;
  const int sdt = k22_storage_dtype(dtype);
  const int nsplit = gn_nsplit(B, HW);
  float* part = reinterpret_cast<float*>(scratch);
  float* coeff = part + (size_t)B * 128 * C * 2;
  GnStatsParams sp;
  sp.x0 = x0; sp.x1 = x1; sp.C0 = C0; sp.C1 = C1; sp.HW = HW; sp.B = B; sp.groups = 32; sp.nsplit = nsplit; sp.partial = part;
  int rc = launch_gn_stats(sp, sdt, st);
  if (rc) return rc;
  GnCoeffParams cp = {};
  cp.src[0].st = part; cp.src[0].rpi = nsplit; cp.src[0].C = C;
  cp.HW = HW; cp.C = C; cp.groups = 32; cp.eps = eps; cp.gamma = gamma; cp.beta = beta; cp.film = film; cp.film_ld = film_ld; cp.coeff = coeff;
  rc = launch_gn_coeff(cp, B, st);
  if (rc) return rc;
  IgemmParams p = {};
  p.stages = -1;
  p.Wp = Wp; p.bias = bias; p.residual = residual; p.out = out; p.partial = reinterpret_cast<float*>(partial);
  p.M = B * H * W; p.N = Cout; p.Npad = Npad; p.Kc = C; p.K0 = C; p.taps = 9; p.H = H; p.W = W;
  p.ldo = Cout; p.ldr = Cout; p.out_mode = IG_OUT_ROWMAJOR; p.act = K22_ACT_NONE; p.splitk = splitk > 0 ? splitk : 1; p.force_bm = bm; p.algo = algo;
  p.gn_coeff = coeff; p.gn_x0 = x0; p.gn_x1 = x1; p.gn_C0 = C0; p.gn_act = act;
  return launch_igemm(p, dtype, st);
}


int k22_attention(const void* qkv, const void* ctxkv, void* kall, void* vtall, void* out, int B, int H, int T, int S,
                  int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int C = H * 64, Tk = S + T, Tkp = (Tk + 63) / 64 * 64;
  KvPackParams kp;
  kp.qkv = qkv; kp.ctxkv = ctxkv; kp.kall = kall; kp.vtall = vtall; kp.B = B; kp.H = H; kp.T = T; kp.S = S; kp.Tkp = Tkp;
  int rc = launch_kv_pack(kp, k22_storage_dtype(dtype), st);
  if (rc) return rc;
  AttentionParams ap = {};
  ap.q = qkv; ap.ldq = 3 * C; ap.kall = kall; ap.vtall = vtall; ap.out = out; ap.ldo = C;
  ap.B = B; ap.H = H; ap.T = T; ap.Tk = Tk; ap.Tkp = Tkp; ap.scale = 0.125f;
  return launch_attention(ap, dtype, st);
}

int k22_qkv_project(const void* x, const void* Wp, const float* bias, void* q_out, void* kall, void* vtall,
                    int B, int H, int T, int S, int K, int bm, int bn, int dtype, void* stream) {
  const int C = H * 64, Tkp = (S + T + 63) / 64 * 64;
  IgemmParams p = {};
  p.stages = -1;
  p.A0 = x; p.Wp = Wp; p.bias = bias; p.out = q_out; p.kall = kall; p.vtall = vtall;
  p.M = B * T; p.N = 3 * C; p.Npad = 3 * C; p.Kc = K; p.K0 = K; p.taps = 1; p.lda0 = K; p.lda1 = 0;
  p.ldo = C; p.ldr = 3 * C; p.out_mode = IG_OUT_QKV; p.act = K22_ACT_NONE; p.splitk = 1; p.force_bm = bm; p.force_bn = bn;
  p.att_T = T; p.att_S = S; p.att_Tkp = Tkp; p.H = 1; p.W = T;   /* rows per image */
  p.a_raw = k22_is_split(dtype) ? 1 : 0;
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_qkv_project_stream(const void* x, const void* Wp, const float* bias, void* q_out, void* kall, void* vtall, void* partial,
                           int B, int H, int T, int S, int K, int bm, int splitk, int dtype, void* stream) {
  const int C = H * 64, Tkp = (S + T + 63) / 64 * 64;
  IgemmParams p = {};
  p.stages = -1;
  p.A0 = x; p.Wp = Wp; p.bias = bias; p.out = q_out; p.kall = kall; p.vtall = vtall; p.partial = reinterpret_cast<float*>(partial);
  p.M = B * T; p.N = 3 * C; p.Npad = 3 * C; p.Kc = K; p.K0 = K; p.taps = 1; p.lda0 = K; p.lda1 = 0;
  p.ldo = C; p.ldr = 3 * C; p.out_mode = IG_OUT_QKV; p.act = K22_ACT_NONE; p.splitk = splitk > 0 ? splitk : 1; p.force_bm = bm; p.algo = 20;
  p.att_T = T; p.att_S = S; p.att_Tkp = Tkp; p.H = 1; p.W = T;
  if (!stream_supported(p, dtype, bm == 288 ? 9 : 5)) return k22_set_error(K22_EINVAL, "qkv_project_stream: unsupported problem");
  K22_DBG_FRAG(p);
  return launch_igemm(p, dtype, reinterpret_cast<hipStream_t>(stream));
}

// ---- skinny-M weight-streaming GEMM family (skinny.hip, small_attention_kernel): unit-parity surface -------------------------------
size_t k22_afrag_bytes(int M, int K) { return afrag_bytes(M, K); }
int k22_afrag_pack(const void* A, long lda, void* out, int M, int K, int dtype, void* stream) {
  return launch_afrag_pack(A, lda, out, M, K, dtype, reinterpret_cast<hipStream_t>(stream));
}
int k22_skinny_gemm(const void* Afrag, const void* Wfrag, const float* bias, void* out, float* partial, int M, int N, int Npad, int K,
                    int splitk, int epi, int act, int ldo, int mt, int nb, int dtype, void* stream) {
  SkinnyParams q = {};
  q.Af = Afrag; q.Wf = Wfrag; q.bias = bias; q.out = out; q.partial = partial; q.M = M; q.N = N; q.Npad = Npad; q.K = K; q.MA = (M + 31) / 32;
  q.splitk = splitk > 0 ? splitk : 1; q.epi = epi; q.act = act; q.ldo = ldo;
  q.trace = reinterpret_cast<unsigned long long*>(g_dbg_scratch);   // K22_SKINNY_DEBUG builds: stamps go to the debug scratch (k22_debug_set_stream_scratch)
  return launch_skinny(q, dtype, mt, nb, reinterpret_cast<hipStream_t>(stream));
}
int k22_finish_ln(const float* partial, int splitk, const float* bias, float* x, long ldx, const float* gain, const float* beta, void* yfrag,
                  int M, int N, float eps, int dtype, void* stream) {
  FinishLnParams q = {};
  q.partial = partial; q.splitk = splitk; q.bias = bias; q.x = x; q.ldx = ldx; q.g = gain; q.b = beta; q.yfrag = yfrag; q.M = M; q.N = N;
  q.MA = (M + 31) / 32; q.eps = eps;
  return launch_finish_ln(q, dtype, reinterpret_cast<hipStream_t>(stream));
}
int k22_small_attention(const void* qkv, const float* qkv_partial, int nsplit, const float* qkv_bias, void* out, int out_frag, int B, int H, int T,
                        int causal, const float* key_valid, int kv_n, int dtype, void* stream) {
  SmallAttnParams ap = {};
  ap.qkv = qkv; ap.part = qkv_partial; ap.nsplit = nsplit; ap.bias = qkv_bias; ap.ldq = 3 * H * 64; ap.out = out; ap.ldo = H * 64; ap.out_frag = out_frag; ap.MA = (B * T + 31) / 32;
  ap.B = B; ap.H = H; ap.T = T; ap.scale = 0.125f; ap.causal = causal; ap.key_valid = key_valid; ap.kv_ld = kv_n; ap.kv_n = kv_n;
  return launch_small_attention(ap, dtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_linear_smallm(const float* x, const void* W, const float* bias, const float* add, float* out, int M, int N,
                      int K, int act_in, int act_out, int wdtype, void* stream) {
  LinearSmallParams lp = {};
  lp.x = x; lp.ldx = K; lp.W = W; lp.bias = bias; lp.add = add; lp.ld_add = N; lp.out = out; lp.ldo = N;
  lp.M = M; lp.N = N; lp.K = K; lp.act_in = act_in; lp.act_out = act_out;
  return launch_linear_smallm(lp, wdtype, reinterpret_cast<hipStream_t>(stream));
}

int k22_ddim_step(const float* x, const float* model_out, const float* noise, const float* table_row, float guidance, int use_cfg,
                  float* x_out, float* x0_out, int N, int HW, void* stream) {
  if (!x || !model_out || !table_row || !x_out) return k22_set_error(K22_EINVAL, "ddim_step: null argument");
  return launch_ddim_step(x, model_out, noise, table_row, guidance, use_cfg, x_out, x0_out, N, HW, reinterpret_cast<hipStream_t>(stream));
}

int k22_prepare_mask(const float* mask, float* out, int C, int H, int W, void* stream) {
  if (!mask || !out) return k22_set_error(K22_EINVAL, "prepare_mask: null argument");
  return launch_prepare_mask(mask, out, C, H, W, reinterpret_cast<hipStream_t>(stream));
}

int k22_plms_step(const float* x, const float* model_out, const float* eps_hist1, const float* eps_hist2, const float* eps_hist3, int order,
                  const float* table_row, float guidance, int use_cfg, float* x_out, float* eps_out, float* x0_out, int N, int HW, void* stream) {
  if (!x || !model_out || !table_row || !x_out) return k22_set_error(K22_EINVAL, "plms_step: null argument");
  return launch_plms_step(x, model_out, eps_hist1, eps_hist2, eps_hist3, order, table_row, guidance, use_cfg, x_out, eps_out, x0_out, N, HW,
                          reinterpret_cast<hipStream_t>(stream));
}

size_t k22_sampler_scratch_bytes(int N, int HW) { return (size_t)N * 4 * HW * sizeof(float) + 256; }

int k22_sampler_step(const float* x, const float* model_out, const float* noise, const float* init_img,
                     const float* mask, const float* table, int step_index, float guidance, int use_cfg,
                     float clamp_lo, float clamp_hi, int pct_index, double pct_gamma, void* scratch,
                     float* x_out, float* x0_out, int N, int HW, void* stream) {
  SamplerParams p = {};
  p.x = x; p.model_out = model_out; p.noise = noise; p.init_img = init_img; p.mask = mask; p.table = table;
  p.step = nullptr; p.step_host = step_index; p.guidance = guidance; p.clamp_lo = clamp_lo; p.clamp_hi = clamp_hi;
  p.use_cfg = use_cfg; p.n_lo = pct_index; p.gamma = pct_gamma;
  p.s_buf = reinterpret_cast<float*>(scratch);
  p.x0_buf = reinterpret_cast<float*>(scratch) + 64;
  p.x_out = x_out; p.x0_out = x0_out; p.N = N; p.HW = HW;
  if (pct_index >= 4 * HW) return k22_set_error(K22_EINVAL, "sampler: percentile index out of range");
  return launch_sampler_step(p, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- debug: LDS sentinel (tools/lds_victim_probe.py) -------------------------------------------------------------
// Every workgroup fills `bytes` of dynamic LDS with a pattern, then re-reads it `spins` times and reports words that changed under it:
// rec[0] = number of changed words seen, rec[1 + 4 * i ...] = (workgroup, byte offset, value found, spin) of the first 255 of them.
// Nothing in the product launches it; it exists to find out which kernel on ANOTHER stream writes into LDS it does not own.
__global__ __launch_bounds__(256) void lds_sentinel_kernel(int words, int spins, unsigned* rec) {
  extern __shared__ unsigned sent_lds[];
  for (int i = threadIdx.x; i < words; i += 256) sent_lds[i] = 0xA5000000u | (unsigned)i;
  __syncthreads();
  for (int s = 0; s < spins; ++s) {
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = reinterpret_cast<volatile unsigned*>(sent_lds)[i];
      if (v != (0xA5000000u | (unsigned)i)) {
        const unsigned k = atomicAdd(rec, 1u);
        if (k < 255u) { rec[1 + 4 * k] = blockIdx.x; rec[2 + 4 * k] = 4u * i; rec[3 + 4 * k] = v; rec[4 + 4 * k] = (unsigned)s; }
        sent_lds[i] = 0xA5000000u | (unsigned)i;
      }
    }
    __builtin_amdgcn_s_sleep(8);
  }
}
extern "C" int k22_debug_lds_sentinel(int bytes, int workgroups, int spins, unsigned* rec, void* stream) {
  if (bytes <= 0 || bytes > 64 * 1024 || (bytes & 3) || !rec) return k22_set_error(K22_EINVAL, "lds_sentinel: bytes in 4..65536, multiple of 4");
  hipLaunchKernelGGL(lds_sentinel_kernel, dim3(workgroups), dim3(256), (size_t)bytes, reinterpret_cast<hipStream_t>(stream), bytes / 4, spins, rec);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
