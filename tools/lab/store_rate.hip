// Lab: how fast can ONE compute unit store?  Workgroups of 512 threads write private 16-byte-per-lane streams
// (global_store_dwordx4, 1 KB per wave instruction, rows of 512 B like the layer kernels' epilogue); the number of
// workgroups (= busy CUs, one each) varies.  Prints bytes / clock / CU at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stores(float* out, long per_block_f4, int reps) {
  f4* base = reinterpret_cast<f4*>(out) + static_cast<long>(blockIdx.x) * per_block_f4;
  const f4 v = {1.0f, 2.0f, 3.0f, static_cast<float>(threadIdx.x)};
  for (int r = 0; r < reps; ++r)
    for (long i = threadIdx.x; i < per_block_f4; i += 512) base[i] = v;
}

__global__ __launch_bounds__(512) void loads(const float* in, float* sink, long per_block_f4, int reps) {
  const f4* base = reinterpret_cast<const f4*>(in) + static_cast<long>(blockIdx.x) * per_block_f4;
  f4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (long i = threadIdx.x; i < per_block_f4; i += 512) acc += base[i];
  if (acc.x == 12345.0f) sink[threadIdx.x] = acc.y;
}

int main() {
  const long per_block = 8L << 20;                 // bytes per workgroup and pass
  const int maxb = 512;
  float *buf, *sink;
  (void)hipMalloc(&buf, per_block * maxb);
  (void)hipMalloc(&sink, 4096);
  (void)hipMemset(buf, 0, per_block * maxb);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int nb : {1, 8, 32, 64, 128, 256, 512}) {
      const int reps = 2;
      for (int it = 0; it < 2; ++it) {
        (void)hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(stores, dim3(nb), dim3(512), 0, 0, buf, per_block / 16, reps);
        else hipLaunchKernelGGL(loads, dim3(nb), dim3(512), 0, 0, buf, sink, per_block / 16, reps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it == 1) {
          const double bytes = double(per_block) * nb * reps;
          const int cus = nb < 256 ? nb : 256;
          printf("%s workgroups=%3d: %8.3f ms  %7.1f GB/s  %6.2f B/clk/CU\n", kind ? "loads " : "stores", nb, ms,
                 bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.4e9 / cus);
        }
      }
    }
  return 0;
}
