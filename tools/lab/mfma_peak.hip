// Practical fp32 MFMA peak of the device: register-resident v_mfma_f32_32x32x2_f32 loop, no memory.
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void peak(float* out, int iters, float a, float b, int rnd) {
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  // pseudo-random operands per lane: data-dependent switching power as in a real GEMM
  unsigned h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u;
  float x = a * (static_cast<float>(h >> 8) * (1.0f / 16777216.0f) - 0.5f);
  h = h * 1664525u + 1013904223u;
  float y = b * (static_cast<float>(h >> 8) * (1.0f / 16777216.0f) - 0.5f);
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c3, 0, 0, 0);
    if (rnd) {   // new operands every step (2 VALU per 4 MFMA)
      x = -x * 0.999f;
      y = y * -1.001f;
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  const int blocks = 256 * 2, iters = 100000;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rnd = 0; rnd <= 1; ++rnd)
  for (int wpb = 2; wpb <= 2; ++wpb) {
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(peak, dim3(256 * wpb), dim3(256), 0, 0, out, iters, rnd ? 1.0f : 0.0f, rnd ? 0.5f : 0.0f, rnd);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = 4096.0 * 4 * iters * 4.0 * 256 * wpb;
      printf("rnd=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", rnd, wpb, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
