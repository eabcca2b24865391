// Micro-benchmark (measurement aid): what does one dependent launch of a captured graph cost on this box, and how much of a
// small kernel's start is the scalar load of its kernel arguments?
//   * chain of N dependent launches of `hop` (every workgroup: kernarg -> index -> data -> store, ping-pong buffers) replayed
//     from ONE graph; reported: microseconds per launch;
//   * the same kernel with its arguments (a) inside one by-value struct, (b) as plain leading scalars;
//   * built twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=16 (the command processor then hands the first
//     kernarg dwords to every wave in SGPRs: no s_load / s_waitcnt in front of the first address computation);
//   * `empty` = a kernel that does nothing (the floor of a dependent launch), `write<MB>` = a kernel that also dirties MB of L2
//     (what the release at the launch boundary has to write back).
// build (tools/gpu_launch_chain.sh does both):
//   hipcc --offload-arch=gfx950 -O3 -o lc_plain tools/ubench/launch_chain.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o lc_preload tools/ubench/launch_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct HopArgs { const int* idx; const float* in; float* out; int n; float add; };

__global__ void __launch_bounds__(256) hop_struct(HopArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = a.idx[i % a.n];
  a.out[i] = a.in[j] + a.add;
}

__global__ void __launch_bounds__(256) hop_plain(const int* idx, const float* in, float* out, int n, float add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = idx[i % n];
  out[i] = in[j] + add;
}

__global__ void __launch_bounds__(256) empty_kernel(float* out) { if (out == nullptr) __builtin_trap(); }

// every workgroup also writes `per_wg` float4 (dirty L2 lines the boundary's release writes back)
__global__ void __launch_bounds__(256) dirty_kernel(float4* out, int per_wg) {
  float4* o = out + (size_t)blockIdx.x * per_wg;
  for (int k = threadIdx.x; k < per_wg; k += 256) o[k] = make_float4(1.f, 2.f, 3.f, (float)k);
}

template <typename F>
static double time_chain(hipStream_t st, int n_chain, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < n_chain; ++k) launch(k);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return (double)ms * 1e3 / ((double)reps * n_chain);
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 224;
  const int n_chain = 40, reps = 100;
  hipStream_t st; CK(hipStreamCreate(&st));
  const int n = wgs * 256;
  int* idx; float *a, *b; float4* big;
  CK(hipMalloc(&idx, n * sizeof(int))); CK(hipMalloc(&a, n * sizeof(float))); CK(hipMalloc(&b, n * sizeof(float)));
  CK(hipMalloc(&big, (size_t)64 << 20));
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919) % n);
  CK(hipMemcpy(idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
  CK(hipMemset(a, 0, n * sizeof(float))); CK(hipMemset(b, 0, n * sizeof(float)));
  printf("{\"workgroups\": %d, \"chain\": %d, \"replays\": %d", wgs, n_chain, reps);
  for (int round = 0; round < 2; ++round) {
    const double t_empty = time_chain(st, n_chain, reps, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, st, a); });
    const double t_struct = time_chain(st, n_chain, reps, [&](int k) {
      HopArgs x{idx, (k & 1) ? b : a, (k & 1) ? a : b, n, 1.f};
      hipLaunchKernelGGL(hop_struct, dim3(wgs), dim3(256), 0, st, x); });
    const double t_plain = time_chain(st, n_chain, reps, [&](int k) {
      hipLaunchKernelGGL(hop_plain, dim3(wgs), dim3(256), 0, st, (const int*)idx, (const float*)((k & 1) ? b : a), (k & 1) ? a : b, n, 1.f); });
    printf(", \"round%d\": {\"empty_us\": %.3f, \"hop_struct_us\": %.3f, \"hop_plain_us\": %.3f", round, t_empty, t_struct, t_plain);
    const int mbs[4] = {1, 4, 16, 48};
    for (int m = 0; m < 4; ++m) {
      const int per_wg = (int)(((size_t)mbs[m] << 20) / 16 / wgs);
      const double t = time_chain(st, n_chain, reps, [&](int) { hipLaunchKernelGGL(dirty_kernel, dim3(wgs), dim3(256), 0, st, big, per_wg); });
      printf(", \"dirty_%dMB_us\": %.3f", mbs[m], t);
    }
    printf("}");
  }
  printf("}\n");
  return 0;
}
