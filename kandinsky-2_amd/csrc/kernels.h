// k22 — internal kernel launch API (host side).  Every launcher enqueues on `stream`, allocates
// nothing and returns 0 or a negative K22_E* code (message via k22_last_error()).
#pragma once
#include "common.h"

#define K22_OK 0
#define K22_EINVAL (-1)
#define K22_EHIP (-2)
#define K22_ENOMEM (-3)

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
                IG_OUT_NCHW_F32 = 2 };    // float out[((b*N + n)*H + y)*W + x]   (m = (b*H+y)*W+x)

struct IgemmParams {
  const void* A0;
  const void* A1;
  const void* Wp;
  const float* bias;     // [N] or null
  const void* residual;  // T [M][ldr] or null
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
  int stages;            // -1 = default (env K22_IGEMM_STAGES), 0 = register staging, 2..4 = LDS-DMA pipeline depth
};

int launch_igemm(const IgemmParams& p, int dtype, hipStream_t stream);
void igemm_set_default_stages(int v);  // tuning knob: 0 register staging, 2..4 LDS-DMA stages, -1 env/default
int igemm_choose_splitk(const IgemmParams& p, int dtype);  // split-K factor the heuristic picks (scratch = splitk*M*N*4 B)
