// What does s_memtime count?  A register-resident fp32 MFMA loop on every CU; wave 0 of every 64th workgroup reads
// s_memtime (the counter tools/lab/ws_trace.py stamps with) and s_memrealtime (100 MHz wall clock) before and after.
// ticks / wall time = the frequency of s_memtime under full matrix load; hipEvent time cross-checks the wall clock.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/lab/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void spin(float* out, unsigned long long* t, int iters, int mfma) {
  f16v c0 = {0}, c1 = {0};
  float x = 0.25f + threadIdx.x * 1e-3f, y = -0.5f;
  const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (mfma) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
    t[(blockIdx.x >> 6) * 2] = m1 - m0;
    t[(blockIdx.x >> 6) * 2 + 1] = r1 - r0;
  }
}

int main() {
  const int blocks = 512;
  float* out;
  unsigned long long* t;
  hipMalloc(&out, blocks * 256 * 4);
  hipHostMalloc(&t, 16 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mfma = 1; mfma >= 0; --mfma)
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = mfma ? 400000 : 40000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, out, t, iters, mfma);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double wall_us = t[1] / 100.0;
      printf("%s: event %.3f ms | wave 0: s_memtime %llu ticks, s_memrealtime %llu ticks = %.1f us -> s_memtime runs at %.1f MHz",
             mfma ? "mfma loop " : "sleep loop", ms, t[0], t[1], wall_us, t[0] / wall_us);
      if (mfma) printf(" | MFMA issue: %.1f s_memtime ticks each, %.1f TFLOP/s chip", double(t[0]) / (2.0 * iters),
                       4096.0 * 2 * iters * 4.0 * blocks / ms / 1e9);
      printf("\n");
    }
  return 0;
}
