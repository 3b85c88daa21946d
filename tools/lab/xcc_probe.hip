// Which XCD does workgroup i of a launch run on?  Reads HW_REG_XCC_ID in every workgroup of (a) a small 1-D grid that
// fits the chip at once, (b) a grid several times larger than the chip, (c) a 2-D grid -- the assumption behind
// pdr::xcd_contiguous (fused_gather.hip) and the XCD-local tile order tried in round 3 is xcc == linear id % 8.
//   hipcc --offload-arch=gfx950 tools/lab/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out, int spin) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = static_cast<int>(x & 0xf);
  // keep the workgroup resident for a while so that later workgroups queue behind a full chip
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < static_cast<unsigned long long>(spin)) {}
}

static void run(dim3 grid, int threads, int spin, const char* what) {
  const int n = grid.x * grid.y;
  int* d;
  hipMalloc(&d, n * sizeof(int));
  hipMemset(d, 0xff, n * sizeof(int));
  hipLaunchKernelGGL(probe, grid, dim3(threads), 0, 0, d, spin);
  hipDeviceSynchronize();
  std::vector<int> h(n);
  hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
  int match = 0;
  int hist[16] = {0};
  for (int i = 0; i < n; ++i) {
    match += h[i] == i % 8;
    if (h[i] >= 0 && h[i] < 16) hist[h[i]]++;
  }
  printf("%-34s %6d workgroups: xcc == id %% 8 for %6d (%.1f %%); first 16:", what, n, match, 100.0 * match / n);
  for (int i = 0; i < 16 && i < n; ++i) printf(" %d", h[i]);
  printf("; per xcc:");
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("\n");
  hipFree(d);
}

int main() {
  run(dim3(512), 512, 2000, "1-D 512 x 512 threads (fits)");
  run(dim3(4096), 512, 500, "1-D 4096 x 512 threads (8x chip)");
  run(dim3(256, 2), 512, 2000, "2-D (256, 2) x 512 threads");
  run(dim3(16384), 256, 100, "1-D 16384 x 256 threads");
  return 0;
}
