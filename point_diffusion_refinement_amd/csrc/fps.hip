// fps.hip -- furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (reference sampling_gpu.cu:69-173).
// The reference runs one 512-thread block per cloud, re-reads/re-writes the
// (B,N) `temp` array from global memory every round and pays ~10 __syncthreads
// per round for its shared-memory tree.  Here a cloud is owned by one workgroup
// of T threads whose points AND running min-distances live in VGPRs for the
// whole kernel; a round is
//     LDS broadcast of the last pick -> PPT fused distance/min/arg-max updates
//     -> one DPP wave arg-max (no LDS) -> W-entry LDS exchange, ONE barrier.
// Index-exactness: the reference's winner is the maximum value with ties broken
// by (bit-reversed reference thread id, then lowest k) -- the shared-memory tree
// keeps the LEFT slot on ties, so bit 0 of tid is the most significant tie bit.
// That total order is folded into a 64-bit key so the result is independent of
// this kernel's own thread geometry:
//     key = float_bits(d2) << 32 | (0xFFFF - tierank) << 16 | k
// with tierank = bitrev(k % R) * ceil(N/R) + k / R and R = opt_n_threads(N)
// (cuda_utils.h:13-19).  Each thread pre-sorts its points by tierank so the
// in-register scan needs only the reference's strict '>'.
#include "pdr_common.h"

namespace {

constexpr int kMaxResidentN = 16384;  // 16-bit k / tierank fields

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_resident_kernel(
    const float* __restrict__ xyz, int N, int m, int R, int Rbits, int Q,
    int* __restrict__ idxs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int W = T / 64;
  float* sx = smem;
  float* sy = smem + N;
  float* sz = smem + 2 * N;
  // exchange slots: 2 parities x W waves (u64), 16-B aligned behind the cloud
  unsigned long long* slots =
      reinterpret_cast<unsigned long long*>(smem + ((3 * N + 3) & ~3));

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* p = xyz + static_cast<size_t>(b) * N * 3;
  int* out = idxs + static_cast<size_t>(b) * m;

  for (int i = tid; i < N; i += T) {
    sx[i] = p[i * 3 + 0];
    sy[i] = p[i * 3 + 1];
    sz[i] = p[i * 3 + 2];
  }

  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
  unsigned low[PPT];  // (0xFFFF - tierank) << 16 | k ; larger = preferred
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * T;
    if (k < N) {
      px[i] = p[k * 3 + 0];
      py[i] = p[k * 3 + 1];
      pz[i] = p[k * 3 + 2];
      const float mag = PDR_SUM3(px[i], py[i], pz[i]);
      // reference: `if (mag <= 1e-3) continue;` -- float promoted to double
      tmp[i] = (static_cast<double>(mag) <= 1e-3) ? -1.0f : 1e10f;
      const unsigned rank = pdr::bitrev(static_cast<unsigned>(k % R), Rbits) * Q + k / R;
      low[i] = ((0xFFFFu - rank) << 16) | static_cast<unsigned>(k);
    } else {
      px[i] = py[i] = pz[i] = 0.0f;
      tmp[i] = -1.0f;  // padding: never selectable, never updated upward
      low[i] = 0u;
    }
  }
  // sort this thread's points by descending `low` (= ascending tierank) so that
  // "first strictly greater wins" inside the thread equals the reference order.
#pragma unroll
  for (int a = 0; a < PPT - 1; ++a) {
#pragma unroll
    for (int c = 0; c < PPT - 1 - a; ++c) {
      if (low[c] < low[c + 1]) {
        float t;
        unsigned u;
        t = px[c]; px[c] = px[c + 1]; px[c + 1] = t;
        t = py[c]; py[c] = py[c + 1]; py[c + 1] = t;
        t = pz[c]; pz[c] = pz[c + 1]; pz[c + 1] = t;
        t = tmp[c]; tmp[c] = tmp[c + 1]; tmp[c + 1] = t;
        u = low[c]; low[c] = low[c + 1]; low[c + 1] = u;
      }
    }
  }

  if (tid == 0) out[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = sx[old], y1 = sy[old], z1 = sz[old];
    float best = -1.0f;
    unsigned bestlow = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
      const float d = PDR_SUM3(dx, dy, dz);
      const float d2 = fminf(d, tmp[i]);
      tmp[i] = d2;
      const bool gt = d2 > best;
      best = gt ? d2 : best;
      bestlow = gt ? low[i] : bestlow;
    }
    // best >= 0 -> monotone uint bits; "no candidate" (-1) -> key 0
    unsigned long long key =
        best < 0.0f ? 0ull : pdr::u64_from(__float_as_uint(best), bestlow);
    key = pdr::wave_max_u64(key);
    if constexpr (W > 1) {
      unsigned long long* s = slots + (j & 1) * W;
      if ((tid & 63) == 0) s[tid >> 6] = key;
      __syncthreads();
      unsigned long long r = s[0];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        const unsigned long long o = s[w];
        r = o > r ? o : r;
      }
      key = r;
    }
    old = static_cast<int>(key & 0xFFFFull);
    if (tid == 0) out[j] = old;
  }
}

// Streaming fallback for N > kMaxResidentN: same ordering rule, running distances
// in the caller's (B,N) temp (reference sampling.cpp:74-76), two-stage (value,
// rank) arg-max.  Not on the BASELINE hot path; kept simple.
__global__ __launch_bounds__(1024) void fps_stream_kernel(
    const float* __restrict__ xyz, int N, int m, int R, int Rbits, int Q,
    float* __restrict__ temp, int* __restrict__ idxs) {
  __shared__ float sval[16];
  __shared__ unsigned long long srank[16];
  __shared__ int sold;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* p = xyz + static_cast<size_t>(b) * N * 3;
  float* tp = temp + static_cast<size_t>(b) * N;
  int* out = idxs + static_cast<size_t>(b) * m;
  for (int k = tid; k < N; k += 1024) {
    const float mag = PDR_SUM3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2]);
    tp[k] = (static_cast<double>(mag) <= 1e-3) ? -1.0f : 1e10f;
  }
  if (tid == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = p[old * 3], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
    float best = -1.0f;
    unsigned long long bestrank = ~0ull;  // smaller = preferred; low 32 bits = k
    for (int k = tid; k < N; k += 1024) {
      const float dx = p[k * 3] - x1, dy = p[k * 3 + 1] - y1, dz = p[k * 3 + 2] - z1;
      const float d2 = fminf(PDR_SUM3(dx, dy, dz), tp[k]);
      tp[k] = d2;
      const unsigned long long rank =
          (static_cast<unsigned long long>(pdr::bitrev(k % R, Rbits)) * Q + k / R) << 32 |
          static_cast<unsigned>(k);
      const bool better = d2 > best || (d2 == best && d2 >= 0.0f && rank < bestrank);
      best = better ? d2 : best;
      bestrank = better ? rank : bestrank;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const unsigned long long orank = __shfl_xor(bestrank, off, 64);
      const bool better = ov > best || (ov == best && orank < bestrank);
      best = better ? ov : best;
      bestrank = better ? orank : bestrank;
    }
    if ((tid & 63) == 0) {
      sval[tid >> 6] = best;
      srank[tid >> 6] = bestrank;
    }
    __syncthreads();
    if (tid == 0) {
      float bv = sval[0];
      unsigned long long br = srank[0];
      for (int w = 1; w < 16; ++w) {
        const bool better = sval[w] > bv || (sval[w] == bv && srank[w] < br);
        bv = better ? sval[w] : bv;
        br = better ? srank[w] : br;
      }
      sold = bv < 0.0f ? 0 : static_cast<int>(br & 0xFFFFFFFFull);
      out[j] = sold;
    }
    __syncthreads();
    old = sold;
  }
}

template <int T, int PPT>
int launch_resident(const float* xyz, int B, int N, int m, int R, int Rbits, int Q,
                    int* idx, hipStream_t s) {
  const size_t lds = static_cast<size_t>((3 * N + 3) & ~3) * sizeof(float) +
                     2 * (T / 64) * sizeof(unsigned long long);
  // the attribute is per DEVICE: set it on every large launch (cheap host call, no global state), checked
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_resident_kernel<T, PPT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return PDR_ELAUNCH;
  hipLaunchKernelGGL((fps_resident_kernel<T, PPT>), dim3(B), dim3(T), lds, s, xyz, N, m, R,
                     Rbits, Q, idx);
  return pdr::check_launch();
}

}  // namespace

extern "C" int pdr_opt_n_threads(int work_size) {
  // cuda_utils.h:13-19, evaluated in double exactly like the reference host code
  if (work_size <= 0) return 1;
  const int pow_2 = static_cast<int>(log(static_cast<double>(work_size)) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

extern "C" size_t pdr_fps_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  // 3N floats of LDS must fit next to the exchange slots (160 KiB per workgroup)
  return (N <= 12288) ? 0 : static_cast<size_t>(B) * N * sizeof(float);
}

extern "C" int pdr_furthest_point_sampling(const float* xyz, int B, int N, int m,
                                           float* temp, int* idx, pdr_stream_t stream) {
  if (B < 0 || N <= 0 || m < 0) return PDR_EINVAL;
  if (B == 0 || m == 0) return PDR_OK;
  if (!xyz || !idx) return PDR_EINVAL;
  hipStream_t s = pdr::as_stream(stream);
  const int R = pdr_opt_n_threads(N);
  int Rbits = 0;
  while ((1 << Rbits) < R) ++Rbits;
  const int Q = (N + R - 1) / R;
  if (pdr_fps_workspace_bytes(B, N) > 0) {
    if (!temp) return PDR_EINVAL;
    hipLaunchKernelGGL(fps_stream_kernel, dim3(B), dim3(1024), 0, s, xyz, N, m, R, Rbits, Q,
                       temp, idx);
    return pdr::check_launch();
  }
  static_assert(kMaxResidentN >= 12288, "resident path must cover the LDS-resident range");
#define PDR_FPS_CASE(T, PPT) \
  if (N <= (T) * (PPT)) return launch_resident<T, PPT>(xyz, B, N, m, R, Rbits, Q, idx, s)
  PDR_FPS_CASE(64, 1);
  PDR_FPS_CASE(64, 2);
  PDR_FPS_CASE(256, 1);
  PDR_FPS_CASE(256, 2);
  PDR_FPS_CASE(256, 4);
  PDR_FPS_CASE(256, 8);
  PDR_FPS_CASE(256, 12);
  PDR_FPS_CASE(256, 16);
  PDR_FPS_CASE(1024, 8);
  PDR_FPS_CASE(1024, 12);
#undef PDR_FPS_CASE
  return PDR_EUNSUPPORTED;
}
