// Consumer-loop probe: ds_read fragments + 4 MFMA per k-step from an LDS-resident chunk; no global traffic.
// variants: 0 = read-then-mfma (as the layer kernel compiles), 1 = software-pipelined fragment prefetch,
//           2 = 16x16x4 MFMA shape with the same tile (more, shorter MFMAs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int KC = 32, TM = 128, TN = 128;

template <int VAR>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ float As[KC][TM + 1];
  __shared__ float Bs[KC][TN];
  for (int i = threadIdx.x; i < KC * TM; i += 256) As[i / TM][i % TM] = (i * 37 % 101) * 0.01f - 0.5f;
  for (int i = threadIdx.x; i < KC * TN; i += 256) Bs[i / TN][i % TN] = (i * 53 % 103) * 0.01f - 0.5f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave & 1, wc = wave >> 1, il = lane & 31, hi = lane >> 5;
  f16v c[2][2] = {};
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) {
      for (int kk = 0; kk < KC / 2; ++kk) {
        float a0 = As[2 * kk + hi][(wr * 2 + 0) * 32 + il], a1 = As[2 * kk + hi][(wr * 2 + 1) * 32 + il];
        float b0 = Bs[2 * kk + hi][(wc * 2 + 0) * 32 + il], b1 = Bs[2 * kk + hi][(wc * 2 + 1) * 32 + il];
        c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c[0][1], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c[1][0], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c[1][1], 0, 0, 0);
        asm volatile("" ::: "memory");
      }
    } else {
      float a0 = As[hi][(wr * 2 + 0) * 32 + il], a1 = As[hi][(wr * 2 + 1) * 32 + il];
      float b0 = Bs[hi][(wc * 2 + 0) * 32 + il], b1 = Bs[hi][(wc * 2 + 1) * 32 + il];
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        const int kn = (kk + 1) % (KC / 2);
        float na0 = As[2 * kn + hi][(wr * 2 + 0) * 32 + il], na1 = As[2 * kn + hi][(wr * 2 + 1) * 32 + il];
        float nb0 = Bs[2 * kn + hi][(wc * 2 + 0) * 32 + il], nb1 = Bs[2 * kn + hi][(wc * 2 + 1) * 32 + il];
        c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c[0][1], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c[1][0], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c[1][1], 0, 0, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += c[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
void run(float* out, int blocks_per_cu) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 4096.0 * 4 * (KC / 2) * iters * 4.0 * 256 * blocks_per_cu;
    printf("var=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", VAR, blocks_per_cu, ms, flops / ms / 1e9);
  }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 256 * 4);
  run<0>(out, 1); run<0>(out, 2); run<0>(out, 3);
  run<1>(out, 1); run<1>(out, 2); run<1>(out, 3);
  return 0;
}
