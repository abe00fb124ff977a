// k22 — skinny-M weight-streaming GEMM + fused split-K finish / LayerNorm (skinny.hip): host-side launch API.
#pragma once
#include "kernels.h"

enum SkinnyEpi { SK_EPI_ROWMAJOR = 0,   // T out[m * ldo + n]            (+ bias, activation)
                 SK_EPI_AFRAG = 1,      // T, A-fragment order of a consumer GEMM whose K is this launch's N   (+ bias, activation)
                 SK_EPI_PARTIAL = 2 };  // float partial[z][m][n], z < splitk   (finish_ln_kernel adds bias / residual)

struct SkinnyParams {
  const void* Af;      // activations, fragment-major [K / 64][MA][4][64][8] T, MA = ceil(M / 32)
  const void* Wf;      // weights, fragment-major [Npad / 32][K / 64][4][64][8] T   (launch_stream_repack with taps = 1)
  const float* bias;   // [N] or null
  void* out;
  float* partial;
  int M, N, Npad, K, MA;
  int splitk, epi, act, ldo;
  unsigned long long* trace;   // K22_SKINNY_DEBUG builds only: 8 stamps per workgroup (s_memrealtime, 100 MHz)
};

struct FinishLnParams {
  const float* partial;   // [splitk][M][N] fp32 or null (null: x is only normalised)
  int splitk;
  const float* bias;      // [N] or null
  float* x;               // fp32 residual stream [M][ldx], updated in place
  long ldx;
  const float* g; const float* b;   // LayerNorm gain / bias [N]; g == null: no LayerNorm output
  void* yfrag;            // LayerNorm output, A-fragment order [N / 64][MA][4][64][8] T
  int M, N, MA;
  float eps;
};

bool skinny_supported(const SkinnyParams& p, int dtype);
void skinny_default_cfg(const SkinnyParams& p, int* mt, int* nb);
// mt x nb = m-atoms x n-atoms (of 32) per workgroup; <= 0: skinny_default_cfg
int launch_skinny(const SkinnyParams& p, int dtype, int mt, int nb, hipStream_t st);
int launch_finish_ln(const FinishLnParams& p, int dtype, hipStream_t st);
int launch_afrag_pack(const void* A, int64_t lda, void* out, int M, int K, int dtype, hipStream_t st);
size_t afrag_bytes(int M, int K);
