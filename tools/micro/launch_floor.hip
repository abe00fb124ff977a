// Measurement tool (not part of the library): wall time per dependent launch of trivial kernels on one stream, eager and as one hipGraph,
// for several grid sizes; and the same with every second kernel replaced by a grid barrier inside ONE kernel (barrier-counter form).
// Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void empty_kernel(float* p, int n) { if (n < 0) p[threadIdx.x] = 1.f; }
__global__ void touch_kernel(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* buf; hipMalloc(&buf, 64 << 20);
  hipStream_t st; hipStreamCreate(&st);
  const int N = 400;
  for (int wgs : {1, 64, 256, 1024}) {
    for (int mode = 0; mode < 2; ++mode) {
      auto body = [&](hipStream_t s) { for (int i = 0; i < N; ++i) { if (mode == 0) hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, s, buf, 0); else hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(256), 0, s, buf, wgs * 256); } };
      body(st); hipStreamSynchronize(st);
      double t0 = now(); body(st); hipStreamSynchronize(st); double eager = (now() - t0) / N * 1e6;
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal); body(st); hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      t0 = now(); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st); hipStreamSynchronize(st); double graph = (now() - t0) / (5 * N) * 1e6;
      printf("%s kernel, %4d workgroups: eager %.2f us / launch, hipGraph %.2f us / launch\n", mode ? "touch" : "empty", wgs, eager, graph);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
