// Micro-benchmark (measurement aid): can one wave issue VALU / transcendental / LDS work in the shadow of its own
// v_mfma_f32_16x16x4_f32 stream?  Prints cycles per loop iteration for: MFMA only, VALU only, both interleaved, ...
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_coissue tools/ubench/mfma_coissue.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, int iters, float x0) {
  __shared__ float lds[4096];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = x0 + i + threadIdx.x;
  float a = x0 + threadIdx.x, b = x0 * 0.5f;
  lds[threadIdx.x] = a;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE & 1) { a0 = MFMA16(a, b, a0); a1 = MFMA16(b, a, a1); }
      if (MODE & 2) {   // 6 plain VALU ops per MFMA pair
#pragma unroll
        for (int q = 0; q < 6; ++q) v[(j + q) & 7] = v[(j + q) & 7] * 1.0001f + 0.5f;
      }
      if (MODE & 4) {   // 2 transcendentals per MFMA pair
        v[j] = __builtin_amdgcn_exp2f(v[j]);
        v[(j + 1) & 7] = __builtin_amdgcn_rcpf(v[(j + 1) & 7]);
      }
      if (MODE & 8) {   // 1 LDS write + 1 LDS read per MFMA pair
        lds[threadIdx.x + 64 * j] = v[j];
        v[(j + 3) & 7] += lds[threadIdx.x + 64 * ((j + 5) & 7)];
      }
    }
  }
  long long t1 = clock64();
  float s = a0[0] + a0[1] + a1[2] + a1[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.1f cycles per group of 2 MFMA slots\n", name, (double)h / (iters * 8.0));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256); hipMalloc(&cyc, 8);
  run<1>("2 MFMA (two chains)", out, cyc);
  run<2>("12 VALU (6 x mul+add, no contraction)", out, cyc);
  run<3>("2 MFMA + 12 VALU", out, cyc);
  run<4>("2 transcendental", out, cyc);
  run<5>("2 MFMA + 2 transcendental", out, cyc);
  run<7>("2 MFMA + 12 VALU + 2 transcendental", out, cyc);
  run<8>("LDS write + read", out, cyc);
  run<9>("2 MFMA + LDS write + read", out, cyc);
  run<15>("everything", out, cyc);
  return 0;
}
