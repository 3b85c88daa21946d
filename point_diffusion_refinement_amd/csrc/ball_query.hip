// ball_query.hip -- radius neighbour search with counts for gfx950.
//
// Replaces query_ball_point_kernel (reference ball_query_gpu.cu:9-47): one CUDA
// thread per query scanning all n points serially, one block per cloud.
//
// MI355X design: the SEARCHED cloud is held in VGPRs -- lane l of a wave keeps
// points {c*64 + l} for every 64-point chunk c (n = 3072 -> 48 chunks = 144
// VGPRs) -- and the QUERIES are streamed through as wave-uniform scalars.  A
// chunk test is 8 flops + compare + one 64-bit ballot; the ballot IS the
// reference's index order, so "first nsample hits in index order" becomes a
// prefix popcount (mbcnt) and the scan stops, wave-uniformly, as soon as
// nsample hits were written.  No LDS, no divergence; grid = B x ceil(m / (4 Qw))
// workgroups of 4 waves, so a B=32, m=2048 call launches >1000 workgroups
// instead of the reference's 32 blocks.
//
// Exactness: d2 uses the same fused expression tree as the oracle (PDR_SUM3),
// the test is the strict `d2 < radius*radius` with radius^2 rounded in f32.
#include "pdr_common.h"

namespace {

template <int NCH, bool RESIDENT>
__global__ __launch_bounds__(256) void ball_query_kernel(
    const float* __restrict__ new_xyz, const float* __restrict__ xyz, int n, int m,
    float radius2, int nsample, int qpw, int* __restrict__ idx,
    int* __restrict__ counts) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* p = xyz + static_cast<size_t>(b) * n * 3;
  const float* q = new_xyz + static_cast<size_t>(b) * m * 3;
  int* oi = idx + static_cast<size_t>(b) * m * nsample;
  int* oc = counts + static_cast<size_t>(b) * m;

  const int nch = RESIDENT ? NCH : (n + 63) / 64;
  float px[NCH], py[NCH], pz[NCH];
  if constexpr (RESIDENT) {
    // unconditional loads from a clamped slot, all in flight together (a load under `k < n` is a branch with its own
    // wait: NCH dependent round trips before the first query), then the out-of-range slots: +inf -> d2 = inf (or
    // NaN), never < radius2
    if constexpr (NCH <= 32) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = min(c * 64 + lane, n - 1);
        px[c] = p[k * 3 + 0];
        py[c] = p[k * 3 + 1];
        pz[c] = p[k * 3 + 2];
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool ok = c * 64 + lane < n;
        px[c] = ok ? px[c] : __builtin_inff();
        py[c] = ok ? py[c] : __builtin_inff();
        pz[c] = ok ? pz[c] : __builtin_inff();
      }
    } else {
      // (48 / 64 slots = 144 / 192 point registers: the compiler batches these conditional loads by itself, and the
      // clamped form measured 4 % slower at 2048 x 3072)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = c * 64 + lane;
        const bool ok = k < n;
        px[c] = ok ? p[k * 3 + 0] : __builtin_inff();
        py[c] = ok ? p[k * 3 + 1] : __builtin_inff();
        pz[c] = ok ? p[k * 3 + 2] : __builtin_inff();
      }
    }
    // complete BEFORE the query loop: otherwise the vmcnt(0) of their first use sits inside the loop, where (stores
    // count in vmcnt on gfx9) it also drains the previous query's index stores
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  }

  const int j0 = (blockIdx.x * 4 + wave) * qpw;
  // query coordinates one trip ahead (scalar loads: their latency was exposed at the top of every trip)
  const int jf = min(j0, m - 1);
  float nqx = q[jf * 3 + 0], nqy = q[jf * 3 + 1], nqz = q[jf * 3 + 2];
  for (int jj = 0; jj < qpw; ++jj) {
    const int j = j0 + jj;  // wave-uniform
    if (j >= m) break;
    const float qx = nqx, qy = nqy, qz = nqz;
    {
      const int jn = min(j + 1, m - 1);
      nqx = q[jn * 3 + 0], nqy = q[jn * 3 + 1], nqz = q[jn * 3 + 2];
    }
    int* row = oi + static_cast<size_t>(j) * nsample;
    int cnt = 0;
    int first = 0;
    // body of one 64-point chunk; `x,y,z` = this lane's point of the chunk
    auto scan = [&](int c, float x, float y, float z) {
      const float dx = qx - x, dy = qy - y, dz = qz - z;
      const float d2 = PDR_SUM3(dx, dy, dz);
      const bool hit = d2 < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull) {
        if (cnt == 0) first = c * 64 + __builtin_ctzll(mask);
        const int pos = cnt + __builtin_amdgcn_mbcnt_hi(
                                  static_cast<unsigned>(mask >> 32),
                                  __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
        if (hit && pos < nsample) row[pos] = c * 64 + lane;
        cnt += __builtin_popcountll(mask);
      }
    };
    if constexpr (RESIDENT) {
      // constant trip count + wave-uniform guard (no `break`): the loop must unroll completely,
      // otherwise px/py/pz are indexed dynamically and land in scratch memory
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (cnt < nsample) scan(c, px[c], py[c], pz[c]);
      }
    } else {
      for (int c = 0; c < nch && cnt < nsample; ++c) {
        const int k = c * 64 + lane;
        const bool ok = k < n;
        const int kc = ok ? k : n - 1;
        const float x = p[kc * 3 + 0], y = p[kc * 3 + 1], z = p[kc * 3 + 2];
        scan(c, ok ? x : __builtin_inff(), ok ? y : __builtin_inff(), ok ? z : __builtin_inff());
      }
    }
    const int filled = cnt < nsample ? cnt : nsample;
    // reference: on the first hit all nsample slots are set to it; later hits
    // overwrite slots [1, cnt).  No hit: row stays zero, count 0.
    for (int l = filled + lane; l < nsample; l += 64) row[l] = first;
    if (lane == 0) oc[j] = filled;
  }
}

template <int NCH, bool RESIDENT>
int launch(const float* new_xyz, const float* xyz, int B, int n, int m, float radius2,
           int nsample, int* idx, int* counts, hipStream_t s) {
  // queries per wave: amortise the register fill of the cloud (n*12 B per wave)
  // while keeping >= ~2 workgroups per CU for B*m large enough.
  int qpw = 16;
  while (qpw > 1 && static_cast<long long>(B) * ((m + 4 * qpw - 1) / (4 * qpw)) < 1024) qpw >>= 1;
  dim3 grid((m + 4 * qpw - 1) / (4 * qpw), B);
  hipLaunchKernelGGL((ball_query_kernel<NCH, RESIDENT>), grid, dim3(256), 0, s, new_xyz, xyz, n,
                     m, radius2, nsample, qpw, idx, counts);
  return pdr::check_launch();
}

}  // namespace

extern "C" int pdr_ball_query(const float* new_xyz, const float* xyz, int B, int n, int m,
                              float radius, int nsample, int* idx, int* counts,
                              pdr_stream_t stream) {
  if (B < 0 || n <= 0 || m < 0 || nsample <= 0) return PDR_EINVAL;
  if (B == 0 || m == 0) return PDR_OK;
  if (!new_xyz || !xyz || !idx || !counts) return PDR_EINVAL;
  hipStream_t s = pdr::as_stream(stream);
  const float radius2 = radius * radius;  // f32, as ball_query_gpu.cu:24
#define PDR_BQ_CASE(NCH) \
  if (n <= 64 * (NCH))   \
  return launch<NCH, true>(new_xyz, xyz, B, n, m, radius2, nsample, idx, counts, s)
  PDR_BQ_CASE(1);
  PDR_BQ_CASE(2);
  PDR_BQ_CASE(4);
  PDR_BQ_CASE(8);
  PDR_BQ_CASE(16);
  PDR_BQ_CASE(32);
  PDR_BQ_CASE(48);
  PDR_BQ_CASE(64);
#undef PDR_BQ_CASE
  return launch<1, false>(new_xyz, xyz, B, n, m, radius2, nsample, idx, counts, s);
}
