// group_gather.hip -- gather_points / group_points / three_interpolate (+ grads).
//
// Replaces the reference's one-block-per-cloud gather kernels
// (sampling_gpu.cu:8-47, group_points_gpu.cu:8-64, interpolate_gpu.cu:72-143).
// These are pure HBM-bound gathers.  Mapping here: one thread per OUTPUT position
// (j or (j,k)), consecutive lanes -> consecutive output addresses, so every store
// is a full-wave coalesced 256-B line; the index is loaded once and reused for all
// channels of the thread's channel slab; the gathered source row (N floats of one
// channel, <= 12 KiB) is L1/L2 resident.  grid = (positions/256, channel slabs, B)
// fills the 256 CUs where the reference used B blocks.
#include "pdr_common.h"

namespace {

constexpr int kSlab = 8;  // channels per thread

// out[b,c,pos] = points[b,c,idx[b,pos]],  pos in [0,P)  (P = m or np*ns)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ points,
                                                          const int* __restrict__ idx, int C,
                                                          int N, int P,
                                                          float* __restrict__ out) {
  const int b = blockIdx.z;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= P) return;
  const int c0 = blockIdx.y * kSlab;
  const int a = idx[static_cast<size_t>(b) * P + pos];
  const float* src = points + (static_cast<size_t>(b) * C + c0) * N + a;
  float* dst = out + (static_cast<size_t>(b) * C + c0) * P + pos;
  const int cn = (C - c0) < kSlab ? (C - c0) : kSlab;
  float v[kSlab];
#pragma unroll
  for (int c = 0; c < kSlab; ++c) v[c] = c < cn ? src[static_cast<size_t>(c) * N] : 0.0f;
#pragma unroll
  for (int c = 0; c < kSlab; ++c)
    if (c < cn) dst[static_cast<size_t>(c) * P] = v[c];
}

// grad_points[b,c,idx[b,pos]] += grad_out[b,c,pos]
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ grad_out,
                                                           const int* __restrict__ idx, int C,
                                                           int N, int P,
                                                           float* __restrict__ grad_points) {
  const int b = blockIdx.z;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= P) return;
  const int c0 = blockIdx.y * kSlab;
  const int a = idx[static_cast<size_t>(b) * P + pos];
  const int cn = (C - c0) < kSlab ? (C - c0) : kSlab;
  for (int c = 0; c < cn; ++c)
    atomicAdd(grad_points + (static_cast<size_t>(b) * C + c0 + c) * N + a,
              grad_out[(static_cast<size_t>(b) * C + c0 + c) * P + pos]);
}

// out[b,c,j] = fma(p[i3],w3, fma(p[i1],w1, p[i2]*w2))   (interpolate_gpu.cu:98-99
// under the nvcc contraction model)
__global__ __launch_bounds__(256) void three_interpolate_kernel(
    const float* __restrict__ points, const int* __restrict__ idx,
    const float* __restrict__ weight, int C, int m, int n, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int c0 = blockIdx.y * kSlab;
  const int* ii = idx + (static_cast<size_t>(b) * n + j) * 3;
  const float* w = weight + (static_cast<size_t>(b) * n + j) * 3;
  const int i1 = ii[0], i2 = ii[1], i3 = ii[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int cn = (C - c0) < kSlab ? (C - c0) : kSlab;
  for (int c = 0; c < cn; ++c) {
    const float* p = points + (static_cast<size_t>(b) * C + c0 + c) * m;
    out[(static_cast<size_t>(b) * C + c0 + c) * n + j] =
        __builtin_fmaf(p[i3], w3, __builtin_fmaf(p[i1], w1, p[i2] * w2));
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    const float* __restrict__ grad_out, const int* __restrict__ idx,
    const float* __restrict__ weight, int C, int n, int m, float* __restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int c0 = blockIdx.y * kSlab;
  const int* ii = idx + (static_cast<size_t>(b) * n + j) * 3;
  const float* w = weight + (static_cast<size_t>(b) * n + j) * 3;
  const int i1 = ii[0], i2 = ii[1], i3 = ii[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int cn = (C - c0) < kSlab ? (C - c0) : kSlab;
  for (int c = 0; c < cn; ++c) {
    float* g = grad_points + (static_cast<size_t>(b) * C + c0 + c) * m;
    const float go = grad_out[(static_cast<size_t>(b) * C + c0 + c) * n + j];
    atomicAdd(g + i1, go * w1);
    atomicAdd(g + i2, go * w2);
    atomicAdd(g + i3, go * w3);
  }
}

inline dim3 grid_for(int P, int C, int B) { return dim3((P + 255) / 256, (C + kSlab - 1) / kSlab, B); }

}  // namespace

extern "C" int pdr_gather_points(const float* points, const int* idx, int B, int C, int N,
                                 int m, float* out, pdr_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || m < 0) return PDR_EINVAL;
  if (B == 0 || C == 0 || m == 0) return PDR_OK;
  if (!points || !idx || !out) return PDR_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, grid_for(m, C, B), dim3(256), 0, pdr::as_stream(stream),
                     points, idx, C, N, m, out);
  return pdr::check_launch();
}

extern "C" int pdr_group_points(const float* points, const int* idx, int B, int C, int N, int np,
                                int ns, float* out, pdr_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || np < 0 || ns < 0) return PDR_EINVAL;
  if (B == 0 || C == 0 || np == 0 || ns == 0) return PDR_OK;
  if (!points || !idx || !out) return PDR_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, grid_for(np * ns, C, B), dim3(256), 0,
                     pdr::as_stream(stream), points, idx, C, N, np * ns, out);
  return pdr::check_launch();
}

static int scatter_common(const float* grad_out, const int* idx, int B, int C, int N, int P,
                          float* grad_points, hipStream_t s) {
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * N, s) !=
      hipSuccess)
    return pdr::check_launch();
  if (P == 0) return PDR_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, grid_for(P, C, B), dim3(256), 0, s, grad_out, idx, C, N,
                     P, grad_points);
  return pdr::check_launch();
}

extern "C" int pdr_gather_points_grad(const float* grad_out, const int* idx, int B, int C, int N,
                                      int m, float* grad_points, pdr_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || m < 0) return PDR_EINVAL;
  if (B == 0 || C == 0) return PDR_OK;
  if (!grad_points || (m > 0 && (!grad_out || !idx))) return PDR_EINVAL;
  return scatter_common(grad_out, idx, B, C, N, m, grad_points, pdr::as_stream(stream));
}

extern "C" int pdr_group_points_grad(const float* grad_out, const int* idx, int B, int C, int N,
                                     int np, int ns, float* grad_points, pdr_stream_t stream) {
  if (B < 0 || C < 0 || N <= 0 || np < 0 || ns < 0) return PDR_EINVAL;
  if (B == 0 || C == 0) return PDR_OK;
  if (!grad_points || (np * ns > 0 && (!grad_out || !idx))) return PDR_EINVAL;
  return scatter_common(grad_out, idx, B, C, N, np * ns, grad_points, pdr::as_stream(stream));
}

extern "C" int pdr_three_interpolate(const float* points, const int* idx, const float* weight,
                                     int B, int C, int m, int n, float* out,
                                     pdr_stream_t stream) {
  if (B < 0 || C < 0 || m <= 0 || n < 0) return PDR_EINVAL;
  if (B == 0 || C == 0 || n == 0) return PDR_OK;
  if (!points || !idx || !weight || !out) return PDR_EINVAL;
  hipLaunchKernelGGL(three_interpolate_kernel, grid_for(n, C, B), dim3(256), 0,
                     pdr::as_stream(stream), points, idx, weight, C, m, n, out);
  return pdr::check_launch();
}

extern "C" int pdr_three_interpolate_grad(const float* grad_out, const int* idx,
                                          const float* weight, int B, int C, int n, int m,
                                          float* grad_points, pdr_stream_t stream) {
  if (B < 0 || C < 0 || m <= 0 || n < 0) return PDR_EINVAL;
  if (B == 0 || C == 0) return PDR_OK;
  if (!grad_points || (n > 0 && (!grad_out || !idx || !weight))) return PDR_EINVAL;
  hipStream_t s = pdr::as_stream(stream);
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * static_cast<size_t>(B) * C * m, s) !=
      hipSuccess)
    return pdr::check_launch();
  if (n == 0) return PDR_OK;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid_for(n, C, B), dim3(256), 0, s, grad_out,
                     idx, weight, C, n, m, grad_points);
  return pdr::check_launch();
}
