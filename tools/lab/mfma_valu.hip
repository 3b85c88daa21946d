// How long does a VALU instruction of one wave take while ANOTHER wave of the same SIMD keeps the matrix pipe busy, and
// what does it cost the MFMA stream?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run back-to-back fp32
// MFMAs of a chosen shape on four independent accumulators; waves 4-7 (their SIMD mates) run a chain of N dependent
// v_fma_f32 (or N independent ones) and time it with s_memtime (2.4 GHz, tools/lab/clock_probe.hip).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/lab/mfma_valu.hip && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// SHAPE 0: no MFMA (waves 0-3 idle), 1: 32x32x2 (64 cycles), 2: 16x16x4 (32 cycles), 3: 4x4x1 16-block (8 cycles?)
template <int SHAPE, bool DEP, int PRIO = 0>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* t, int mfma_iters, int valu_n) {
  const int wave = threadIdx.x >> 6;
  float x = 0.25f + threadIdx.x * 1e-3f, y = -0.5f;
  if (wave < 4) {
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    float s = 0;
    if constexpr (SHAPE == 1) {
      f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < mfma_iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, c3, 0, 0, 0);
      }
      for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    } else if constexpr (SHAPE == 2) {
      f4v c[8] = {};
      for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32((k & 1) ? x : y, (k & 2) ? x : y, c[k], 0, 0, 0);
      }
      for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = m1 - m0;
  } else {
    // let the MFMA waves get going
    __builtin_amdgcn_s_sleep(64);
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    float a = x, b = 1.0001f, c = 0.5f;
    float a1 = x + 1, a2 = x + 2, a3 = x + 3;
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < valu_n; i += 4) {
      if constexpr (DEP) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                     : "+v"(a) : "v"(b), "v"(c));
      } else {
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                     : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
      }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = a + a1 + a2 + a3;
    if (threadIdx.x == 256 && blockIdx.x == 0) t[1] = m1 - m0;
  }
}

template <int SHAPE, bool DEP, int PRIO = 0>
void run(const char* name, float* out, unsigned long long* t, int valu_n) {
  const int iters = 4000;                    // SHAPE 1: 16000 MFMAs x 64 cycles; SHAPE 2: 32000 x 32 cycles = 1.02 M cycles
  for (int rep = 0; rep < 2; ++rep) {
    t[0] = t[1] = 0;
    hipLaunchKernelGGL((probe<SHAPE, DEP, PRIO>), dim3(256), dim3(512), 0, 0, out, t, iters, valu_n);
    hipDeviceSynchronize();
  }
  const double mf = SHAPE == 1 ? iters * 4.0 : SHAPE == 2 ? iters * 8.0 : 1.0;
  printf("%-34s %s VALU x %6d: %8.1f cycles each", name, DEP ? "dependent  " : "independent", valu_n, double(t[1]) / valu_n);
  if (SHAPE) printf(" | MFMA stream: %6.1f cycles each (%d-cycle shape)", double(t[0]) / mf, SHAPE == 1 ? 64 : 32);
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* t;
  hipMalloc(&out, 256 * 512 * 4);
  hipHostMalloc(&t, 64);
  // valu_n = 0: MFMA stream alone; then a VALU chain that spans the whole MFMA run (16000 x 64 cycles)
  run<0, true>("no MFMA on the SIMD", out, t, 20000);
  run<0, false>("no MFMA on the SIMD", out, t, 20000);
  run<1, true>("32x32x2 mate, VALU idle", out, t, 4);
  run<2, true>("16x16x4 mate, VALU idle", out, t, 4);
  run<1, true>("32x32x2 mate", out, t, 12000);
  run<1, false>("32x32x2 mate", out, t, 12000);
  run<2, true>("16x16x4 mate", out, t, 12000);
  run<2, false>("16x16x4 mate", out, t, 12000);
  // the VALU wave at s_setprio(3): what a staging / epilogue instruction costs and what it takes from the MFMA stream
  run<1, true, 3>("32x32x2 mate, VALU wave prio 3", out, t, 12000);
  run<1, false, 3>("32x32x2 mate, VALU wave prio 3", out, t, 12000);
  run<2, true, 3>("16x16x4 mate, VALU wave prio 3", out, t, 12000);
  run<2, false, 3>("16x16x4 mate, VALU wave prio 3", out, t, 12000);
  run<1, false, 3>("32x32x2 mate, VALU wave prio 3", out, t, 60000);
  run<2, false, 3>("16x16x4 mate, VALU wave prio 3", out, t, 60000);
  return 0;
}
