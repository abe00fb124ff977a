// k22 — small per-call / per-step latent arithmetic of the img2img and inpainting pipelines:
//   noised(t)  = sa * init + sb * noise                      DDPMScheduler.add_noise / q_sample (kandinsky2/utils.py:43-54)
//   out        = mask * noised(t) + (1 - mask) * x           the per-step re-imposition of the known region in the Kandinsky 2.2
//                                                            inpainting pipeline (diffusers KandinskyV22InpaintPipeline loop, used by
//                                                            kandinsky2/kandinsky2_2_model.py:150-173); mask [N][1][h][w], 1 = keep
// One elementwise kernel; x == nullptr / mask == nullptr gives plain add_noise.
#include "kernels.h"
#include "../../include/k22.h"

namespace {
__global__ __launch_bounds__(256) void blend_noised_kernel(const float* x, const float* init, const float* noise, const float* mask, float sa,
                                                           float sb, float* out, int C, int HW, int64_t total, int init_bcast) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / ((int64_t)C * HW), p = i % HW;
    const int64_t j = init_bcast ? i % ((int64_t)C * HW) : i;         // init / noise / mask of image 0 serve the whole batch
    const float v = sa * init[j] + (noise != nullptr ? sb * noise[j] : 0.f);
    if (mask == nullptr) { out[i] = v; continue; }
    const float m = mask[(init_bcast ? 0 : n) * HW + p];
    out[i] = m * v + (1.f - m) * x[i];
  }
}
}  // namespace

extern "C" int k22_blend_noised(const float* x, const float* init, const float* noise, const float* mask, float sa, float sb, float* out,
                                int N, int C, int HW, int broadcast_first, void* stream) {
  if (!init || !out || N < 1 || C < 1 || HW < 1) return k22_set_error(K22_EINVAL, "blend_noised: bad argument");
  if (mask != nullptr && x == nullptr) return k22_set_error(K22_EINVAL, "blend_noised: a mask needs the current latent x");
  const int64_t total = (int64_t)N * C * HW;
  const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(blend_noised_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, init, noise, mask, sa, sb, out, C, HW,
                     total, broadcast_first ? 1 : 0);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
