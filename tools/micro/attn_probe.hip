// Measurement tool (not part of the library): where the UNet's attention kernel spends its time.  Includes csrc/attention.hip itself and
// times measurement-only variants of its kernels (template parameter DBG: parts of the tile loop removed, wrong results) on the C2 shapes.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -DK22_NOPK=1 -DK22_ATT_PROBE \
//         -I kandinsky-2_amd/csrc -o tools/micro/attn_probe tools/micro/attn_probe.hip
// Run: tools/micro/attn_probe
#include "../../kandinsky-2_amd/csrc/attention.hip"
#include <stdio.h>
#include <stdlib.h>
#include <cstring>
#include <vector>
int k22_set_error(int code, const char* msg) { fprintf(stderr, "k22 error %d: %s\n", code, msg); return code; }
int k22_set_error_hip(hipError_t e, const char* file, int line) { fprintf(stderr, "hip error %s at %s:%d\n", hipGetErrorString(e), file, line); return -1; }

static unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

template <typename F> static double time_us(F&& launch, int reps, hipStream_t st) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipStreamSynchronize(st);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / reps * 1e3 < best) best = ms / reps * 1e3;
  }
  return best;
}

struct Case { int B, H, T, S; };

int main(int argc, char** argv) {
  hipStream_t st; hipStreamCreate(&st);
  const Case cases[] = {{2, 12, 2304, 87}, {1, 14, 2304, 87}, {2, 18, 576, 87}, {2, 24, 144, 87}};
  for (const Case& c : cases) {
    const int C = c.H * 64, Tk = c.S + c.T, Tkp = (Tk + 63) / 64 * 64;
    const size_t nq = (size_t)c.B * c.T * 3 * C, nk = (size_t)c.B * c.H * Tkp * 64;
    std::vector<unsigned short> hq(nq), hk(nk), hv(nk);
    srand(1);
    for (auto& v : hq) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    for (auto& v : hk) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    for (auto& v : hv) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    unsigned short *q, *k, *v, *o;
    hipMalloc(&q, nq * 2); hipMalloc(&k, nk * 2); hipMalloc(&v, nk * 2); hipMalloc(&o, (size_t)c.B * c.T * C * 2);
    hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), nk * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), nk * 2, hipMemcpyHostToDevice);
    AttentionParams ap = {};
    ap.q = q; ap.ldq = 3 * C; ap.kall = k; ap.vtall = v; ap.out = o; ap.ldo = C;
    ap.B = c.B; ap.H = c.H; ap.T = c.T; ap.Tk = Tk; ap.Tkp = Tkp; ap.scale = 0.125f;
    const dim3 grid((c.T + 127) / 128, c.H, c.B);
    const dim3 grid1(grid.x * grid.y * grid.z);   // attention_pipe_kernel: 1-D, XCD-aware mapping inside
    const double fl = 4.0 * c.B * c.H * c.T * (double)Tk * 64;
    printf("B=%d H=%d T=%d Tk=%d: %d workgroups of 256\n", c.B, c.H, c.T, Tk, grid.x * grid.y * grid.z);
#define RUN(NAME, ...)                                                                      \
    {                                                                                       \
      const double us = time_us([&] { __VA_ARGS__; }, 30, st);                              \
      printf("  %-58s %7.1f us  %6.1f TFLOP/s\n", NAME, us, fl / us / 1e6);                 \
    }
    RUN("attention_kernel (shipped)", hipLaunchKernelGGL((attention_kernel<bf16_t, 0>), grid, dim3(256), 0, st, ap));
    RUN("  tiles staged once (no barriers / stores / loads in loop)", hipLaunchKernelGGL((attention_kernel<bf16_t, 1>), grid, dim3(256), 0, st, ap));
    RUN("  no v_exp", hipLaunchKernelGGL((attention_kernel<bf16_t, 2>), grid, dim3(256), 0, st, ap));
    RUN("  no S MFMAs", hipLaunchKernelGGL((attention_kernel<bf16_t, 4>), grid, dim3(256), 0, st, ap));
    RUN("  no PV MFMAs", hipLaunchKernelGGL((attention_kernel<bf16_t, 8>), grid, dim3(256), 0, st, ap));
    RUN("  no MFMAs at all", hipLaunchKernelGGL((attention_kernel<bf16_t, 12>), grid, dim3(256), 0, st, ap));
    RUN("  no max / rescale", hipLaunchKernelGGL((attention_kernel<bf16_t, 16>), grid, dim3(256), 0, st, ap));
    RUN("  staged once + no exp + no max", hipLaunchKernelGGL((attention_kernel<bf16_t, 19>), grid, dim3(256), 0, st, ap));
    RUN("  staged once + no MFMAs", hipLaunchKernelGGL((attention_kernel<bf16_t, 13>), grid, dim3(256), 0, st, ap));
    RUN("attention_pipe_kernel", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 0>), grid1, dim3(256), 0, st, ap));
    RUN("  no v_exp", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 2>), grid1, dim3(256), 0, st, ap));
    RUN("  tiles staged once", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 1>), grid1, dim3(256), 0, st, ap));
    RUN("  no MFMAs", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 12>), grid1, dim3(256), 0, st, ap));
    RUN("  no max", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 16>), grid1, dim3(256), 0, st, ap));
    RUN("  workgroups in launch order (no XCD-aware mapping)", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 64>), grid1, dim3(256), 0, st, ap));
    RUN("  no sched_group_barrier pins", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 32>), grid1, dim3(256), 0, st, ap));
    RUN("  staged once + no exp + no max", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 19>), grid1, dim3(256), 0, st, ap));
    RUN("  staged once + no MFMAs", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 13>), grid1, dim3(256), 0, st, ap));
    RUN("  staged once + no MFMAs + no exp + no max", hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 31>), grid1, dim3(256), 0, st, ap));
    {   // same bits as the shipped kernel?
      const size_t no = (size_t)c.B * c.T * C;
      std::vector<unsigned short> a(no), b2(no);
      hipLaunchKernelGGL((attention_kernel<bf16_t, 0>), grid, dim3(256), 0, st, ap); hipStreamSynchronize(st);
      hipMemcpy(a.data(), o, no * 2, hipMemcpyDeviceToHost);
      hipMemset(o, 0, no * 2);
      hipLaunchKernelGGL((attention_pipe_kernel<bf16_t, 0>), grid1, dim3(256), 0, st, ap); hipStreamSynchronize(st);
      hipMemcpy(b2.data(), o, no * 2, hipMemcpyDeviceToHost);
      size_t nd = 0; for (size_t i = 0; i < no; ++i) nd += a[i] != b2[i];
      printf("  pipe vs shipped: %zu of %zu output values differ\n", nd, no);
    }
    hipFree(q); hipFree(k); hipFree(v); hipFree(o);
  }
  return 0;
}
