// Which workgroups of a persistent 512-workgroup launch share a CU?  Every workgroup (512 threads, 72 KB of LDS: two
// per CU, like the wave-specialised layer kernels) records HW_REG_XCC_ID and the SE / SH / CU fields of HW_REG_HW_ID
// and stays resident until all have started.  Prints the histogram of id differences between CU mates.
//   hipcc --offload-arch=gfx950 tools/lab/cu_probe.hip -o /tmp/cu_probe && /tmp/cu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char lds[];
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  if (threadIdx.x == 0) {
    lds[0] = 1;
    out[blockIdx.y * gridDim.x + blockIdx.x] = ((x & 0xf) << 16) | ((h >> 8) & 0xff);   // xcc | SE SH CU
  }
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < static_cast<unsigned long long>(spin)) {}
}

static void run(dim3 grid, const char* what) {
  const int n = grid.x * grid.y;
  unsigned* d;
  hipMalloc(&d, n * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipLaunchKernelGGL(probe, grid, dim3(512), 72 * 1024, 0, d, 5000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(n);
  hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int i = 0; i < n; ++i) cu[h[i]].push_back(i);
  std::map<int, int> diff;
  int mates = 0;
  for (auto& kv : cu)
    for (size_t a = 0; a + 1 < kv.second.size(); ++a) {
      diff[kv.second[a + 1] - kv.second[a]]++;
      ++mates;
    }
  printf("%s: %d workgroups on %zu CUs; id difference between CU mates:", what, n, cu.size());
  for (auto& kv : diff) printf("  %d x%d", kv.first, kv.second);
  printf("\n  first CUs:");
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 6) break;
    printf("  [%05x:", kv.first);
    for (int id : kv.second) printf(" %d", id);
    printf("]");
  }
  printf("\n");
  hipFree(d);
}

int main() {
  run(dim3(512), "1-D 512");
  run(dim3(256, 2), "2-D (256, 2)");
  run(dim3(128, 4), "2-D (128, 4)");
  run(dim3(768), "1-D 768 (third of them queue)");
  return 0;
}
