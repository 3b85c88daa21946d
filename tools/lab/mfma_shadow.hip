// Can a wave issue its OWN independent VALU instructions while its MFMA executes (the "shadow" of a 64-cycle
// v_mfma_f32_32x32x2_f32)?  One wave per SIMD; per loop trip 4 MFMAs on independent accumulators, each followed by NV
// independent v_fma_f32 on other registers (all inline asm, fixed order).  Cycles per trip vs NV tells what a VALU
// instruction costs next to the wave's own MFMA stream.  Also: two such waves per SIMD (as the two workgroups of a CU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shadow tools/lab/mfma_shadow.hip && /tmp/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

#define VF(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
template <int NV>
__device__ __forceinline__ void valu_block(float& a0, float& a1, float& a2, float& a3, float b, float c) {
  // NV independent-ish VALU: round-robin over four chains
  if constexpr (NV >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
  if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
  if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
  if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
  if constexpr (NV > 4) valu_block<NV - 4>(a0, a1, a2, a3, b, c);
}
template <>
__device__ __forceinline__ void valu_block<0>(float&, float&, float&, float&, float, float) {}

template <int NV, int SHAPE>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* t, int iters) {
  float x = 0.25f + threadIdx.x * 1e-3f, y = -0.5f, b = 1.0001f, c = 0.5f;
  float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const unsigned long long m0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (SHAPE == 1) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(x), "v"(y));
      valu_block<NV>(a0, a1, a2, a3, b, c);
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(y), "v"(x));
      valu_block<NV>(a0, a1, a2, a3, b, c);
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c2) : "v"(x), "v"(x));
      valu_block<NV>(a0, a1, a2, a3, b, c);
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c3) : "v"(y), "v"(y));
      valu_block<NV>(a0, a1, a2, a3, b, c);
    } else {
      valu_block<NV>(a0, a1, a2, a3, b, c);
      valu_block<NV>(a0, a1, a2, a3, b, c);
      valu_block<NV>(a0, a1, a2, a3, b, c);
      valu_block<NV>(a0, a1, a2, a3, b, c);
    }
  }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) t[threadIdx.x >> 6] = m1 - m0;
}

template <int NV, int SHAPE>
void run(float* out, unsigned long long* t, int threads) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<NV, SHAPE>), dim3(256), dim3(threads), 0, 0, out, t, iters);
    hipDeviceSynchronize();
  }
  printf("%s, %d wave(s) per SIMD, %2d VALU after each MFMA slot: %7.1f cycles per trip of 4 slots (wave 0)%s\n",
         SHAPE ? "4 x v_mfma_f32_32x32x2" : "no MFMA              ", threads / 256, NV, double(t[0]) / iters,
         SHAPE ? "  [256 = MFMA-bound]" : "");
}

int main() {
  float* out;
  unsigned long long* t;
  hipMalloc(&out, 256 * 512 * 4);
  hipHostMalloc(&t, 64);
  run<0, 1>(out, t, 256);
  run<4, 1>(out, t, 256);
  run<8, 1>(out, t, 256);
  run<12, 1>(out, t, 256);
  run<16, 1>(out, t, 256);
  run<24, 1>(out, t, 256);
  run<8, 0>(out, t, 256);
  run<16, 0>(out, t, 256);
  run<0, 1>(out, t, 512);
  run<4, 1>(out, t, 512);
  run<8, 1>(out, t, 512);
  run<16, 1>(out, t, 512);
  return 0;
}
