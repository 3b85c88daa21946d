// neighbors.hip -- brute-force K-nearest search: three_nn (K=3) and knn_points.
//
// Replaces three_nn_kernel (reference interpolate_gpu.cu:9-59) and the
// un-vendored pytorch3d.ops.knn.knn_points (call sites pointnet2_utils.py:365,
// 496-497; chamfer_loss_new.py:149-150).
//
// One thread per query, 256 queries per workgroup; the searched cloud streams
// through LDS in 1024-point float4 tiles that every lane reads at the SAME
// address (hardware broadcast, conflict-free ds_read_b128).  Each thread keeps
// its K best (distance, index) pairs sorted in registers; insertion is a
// branch-free shift guarded by one wave-level `d < worst` test, so after the
// first few dozen points most iterations are 8 flops + 1 compare.  Strict `<`
// plus ascending scan order gives "equal distances: lower index first", which is
// both the reference three_nn cascade and the knn contract in include/pdr_hip.h.
#include <cstdlib>

#include "pdr_common.h"

namespace {

constexpr int kTile = 1024;

enum DistModel { kSum3 = 0, kAcc3 = 1 };

template <int K, int MODEL, typename IdxT, bool KNN_PAD>
__global__ __launch_bounds__(256) void nn_search_kernel(const float* __restrict__ queries,
                                                        const float* __restrict__ cloud, int nq,
                                                        int nc, int Kout,
                                                        float* __restrict__ dists,
                                                        IdxT* __restrict__ idx,
                                                        float* __restrict__ nn,
                                                        float* __restrict__ wgt = nullptr) {
  __shared__ float4 tile[kTile];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool active = j < nq;
  const float* q = queries + (static_cast<size_t>(b) * nq + (active ? j : 0)) * 3;
  const float* c = cloud + static_cast<size_t>(b) * nc * 3;
  const float qx = q[0], qy = q[1], qz = q[2];

  float bd[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bd[t] = __builtin_inff();
    bi[t] = KNN_PAD ? -1 : 0;
  }

  for (int k0 = 0; k0 < nc; k0 += kTile) {
    const int kn = (nc - k0) < kTile ? (nc - k0) : kTile;
    __syncthreads();
    for (int t = threadIdx.x; t < kn; t += 256) {
      const float* s = c + static_cast<size_t>(k0 + t) * 3;
      tile[t] = make_float4(s[0], s[1], s[2], 0.0f);
    }
    __syncthreads();
    if (active) {
      for (int t = 0; t < kn; ++t) {
        const float4 pt = tile[t];
        const float dx = qx - pt.x, dy = qy - pt.y, dz = qz - pt.z;
        const float d = MODEL == kSum3 ? PDR_SUM3(dx, dy, dz) : PDR_ACC3(dx, dy, dz);
        if (d < bd[K - 1]) {
          const int k = k0 + t;
#pragma unroll
          for (int s = K - 1; s >= 1; --s) {
            const bool shift = bd[s - 1] > d;  // sorted: implies bd[s] > d
            const bool here = bd[s] > d;
            bd[s] = shift ? bd[s - 1] : (here ? d : bd[s]);
            bi[s] = shift ? bi[s - 1] : (here ? k : bi[s]);
          }
          if (bd[0] > d) {
            bd[0] = d;
            bi[0] = k;
          }
        }
      }
    }
  }
  if (!active) return;
  float* od = dists + (static_cast<size_t>(b) * nq + j) * Kout;
  IdxT* oi = idx + (static_cast<size_t>(b) * nq + j) * Kout;
  if (wgt) {
    // group_knn's interpolation weights (pointnet2_utils.py:500-503): 1 / (d2 + 1e-8) normalised over the K
    // neighbours (SQUARED distances), summed in ascending-k order like pdr_knn_build
    float* ow = wgt + (static_cast<size_t>(b) * nq + j) * Kout;
    float norm = 0.0f;
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (t < Kout) norm += 1.0f / ((KNN_PAD && bi[t] < 0 ? 0.0f : bd[t]) + 1e-8f);
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (t < Kout) ow[t] = (1.0f / ((KNN_PAD && bi[t] < 0 ? 0.0f : bd[t]) + 1e-8f)) / norm;
  }
#pragma unroll
  for (int t = 0; t < K; ++t) {
    if (t < Kout) {
      const bool empty = KNN_PAD && bi[t] < 0;
      od[t] = empty ? 0.0f : bd[t];
      oi[t] = static_cast<IdxT>(bi[t]);
      if (nn) {
        float* on = nn + ((static_cast<size_t>(b) * nq + j) * Kout + t) * 3;
        const int a = empty ? 0 : bi[t];
        on[0] = empty ? 0.0f : c[a * 3 + 0];
        on[1] = empty ? 0.0f : c[a * 3 + 1];
        on[2] = empty ? 0.0f : c[a * 3 + 2];
      }
    }
  }
}

// ---- K <= 8 over a cloud of 64 .. 1024 points: one WAVE per query ------------------------------------------------
// The thread-per-query kernel above keeps a sorted top-K per LANE: its `d < worst` guard is a wave-level branch, and
// with 64 independent queries per wave some lane takes it in most iterations (P[insert] = K / t per lane at point t),
// so the 30-instruction shift sequence runs ~600 times per 1024 points; a (2048 x 1024, B = 32) call also launches
// only 256 workgroups = 1 wave per SIMD.  Measured 266 us = 2.0 TF (profiles/r2_roofline.json).
// Here the searched cloud lives in VGPRs (lane l holds points l, l + 64, ..., as ball_query.hip) and queries are
// streamed through as wave-uniform scalars; all 64 lanes work on the SAME query, so there is one threshold:
//   1. every lane evaluates its NCH distances (same ACC3 expression tree -> same bits as the kernel above);
//   2. T = max over the eight 8-lane groups of the group's minimum distance: at least 8 points have d <= T;
//   3. points with d <= T (typically ~20 of 1024) are compacted into LDS in index order (ballot + mbcnt);
//   4. every candidate's rank = number of candidates with a smaller (distance bits, index) key -- one readlane pair
//      + one 64-bit compare per candidate; ranks < K are the answer in ascending order, equal distances by lower
//      index, exactly the contract of the sorted-insertion kernel (bit-identical outputs, tested).
// More than 64 candidates (massive exact ties) take an exact K-round minimum extraction instead.
template <int NCH, typename IdxT>
__global__ __launch_bounds__(256) void knn_wave_kernel(const float* __restrict__ queries,
                                                       const float* __restrict__ cloud, int nq, int nc, int K,
                                                       int qpw, float* __restrict__ dists, IdxT* __restrict__ idx,
                                                       float* __restrict__ nn, float* __restrict__ wgt) {
  __shared__ unsigned long long cand[4][64];
  __shared__ float outd[4][8];
  __shared__ int outi[4][8];
  __shared__ float outr[4][8];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* c = cloud + static_cast<size_t>(b) * nc * 3;
  const float* q = queries + static_cast<size_t>(b) * nq * 3;
  float px[NCH], py[NCH], pz[NCH];
  // (unconditional loads from a clamped slot, all in flight together; a load under `k < nc` is a branch with its
  // own wait: 16 dependent round trips before the first query)
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int k = min(ch * 64 + lane, nc - 1);
    px[ch] = c[k * 3 + 0];
    py[ch] = c[k * 3 + 1];
    pz[ch] = c[k * 3 + 2];
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    // out-of-range slots: +inf -> d = +inf, behind every real point
    const bool ok = ch * 64 + lane < nc;
    px[ch] = ok ? px[ch] : __builtin_inff();
    py[ch] = ok ? py[ch] : __builtin_inff();
    pz[ch] = ok ? pz[ch] : __builtin_inff();
  }
  const int j0 = (blockIdx.x * 4 + wave) * qpw;
  // The cloud slots are complete BEFORE the query loop: without this the wait-count pass puts the vmcnt(0) of their
  // first use inside the loop, where (stores count in vmcnt on gfx9) it also drains the previous query's stores
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  // query coordinates one trip ahead (scalar loads: their latency was exposed at the top of every trip)
  const int jf = min(j0, nq - 1);
  float nqx = q[jf * 3 + 0], nqy = q[jf * 3 + 1], nqz = q[jf * 3 + 2];
  for (int jj = 0; jj < qpw; ++jj) {
    const int j = j0 + jj;   // wave-uniform
    if (j >= nq) break;
    const float qx = nqx, qy = nqy, qz = nqz;
    {
      const int jn = min(j + 1, nq - 1);
      nqx = q[jn * 3 + 0], nqy = q[jn * 3 + 1], nqz = q[jn * 3 + 2];
    }
    float d[NCH];
    float m = __builtin_inff();
    if constexpr (NCH % 2 == 0) {
      // two cloud slots per instruction: v_pk_add / v_pk_mul / v_pk_fma evaluate PDR_ACC3 on a pair of points with
      // the same roundings as the scalar form (a packed fma IS two fmas), one v_min3 folds the pair into the minimum
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
#pragma unroll
      for (int ch = 0; ch < NCH; ch += 2) {
        const f2 dx = qx2 - f2{px[ch], px[ch + 1]}, dy = qy2 - f2{py[ch], py[ch + 1]},
                 dz = qz2 - f2{pz[ch], pz[ch + 1]};
        const f2 dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        d[ch] = dd.x;
        d[ch + 1] = dd.y;
        m = fminf(m, fminf(dd.x, dd.y));
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const float dx = qx - px[ch], dy = qy - py[ch], dz = qz - pz[ch];
        d[ch] = PDR_ACC3(dx, dy, dz);
        m = fminf(m, d[ch]);
      }
    }
    // minimum of each aligned group of 8 lanes (xor 1, xor 2 inside a quad, then the other quad of the half row),
    // then the maximum over the wave as unsigned bits (distances are >= 0: bit order == value order)
    unsigned g = __float_as_uint(m);
    {
      unsigned o = __builtin_amdgcn_update_dpp(0u, g, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      g = o < g ? o : g;
      o = __builtin_amdgcn_update_dpp(0u, g, 0x4E, 0xf, 0xf, false);            // quad_perm [2,3,0,1]
      g = o < g ? o : g;
      o = __builtin_amdgcn_update_dpp(0u, g, 0x141, 0xf, 0xf, false);           // row_half_mirror
      g = o < g ? o : g;
    }
    const unsigned T = pdr::wave_max_u32(g);
    // candidates, compacted in index order
    int base = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const unsigned bits = __float_as_uint(d[ch]);
      const bool hit = bits <= T;
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull) {
        const int pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
        if (hit && pos < 64) cand[wave][pos] = pdr::u64_from(bits, static_cast<unsigned>(ch * 64 + lane));
        base += __builtin_popcountll(mask);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // base < K happens only when the distances are NaN (a non-finite query or point): no candidate passes
    // `bits <= T`.  Those queries take the exact extraction below, which orders NaN keys by index and never leaves
    // the output slots unwritten (the rank path then wrote nothing and stale LDS was stored).
    if (base >= K && base <= 64) {
      const unsigned long long key = lane < base ? cand[wave][lane] : ~0ull;
      const unsigned klo = static_cast<unsigned>(key), khi = static_cast<unsigned>(key >> 32);
      int rank = 0;
      for (int r = 0; r < base; ++r) {
        const unsigned long long kr =
            pdr::u64_from(__builtin_amdgcn_readlane(khi, r), __builtin_amdgcn_readlane(klo, r));
        rank += kr < key ? 1 : 0;
      }
      if (lane < base && rank < K) {
        outd[wave][rank] = __uint_as_float(khi);
        outi[wave][rank] = static_cast<int>(klo);
      }
    } else {
      // exact fallback: K rounds of wave-wide minimum extraction over (distance bits, index) keys
      unsigned long long last = 0ull;
      for (int r = 0; r < K; ++r) {
        unsigned long long best = ~0ull;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          // (slots beyond the cloud never win: with NaN distances their +inf would sort FIRST as unsigned bits)
          const unsigned long long key =
              ch * 64 + lane < nc ? pdr::u64_from(__float_as_uint(d[ch]), static_cast<unsigned>(ch * 64 + lane))
                                  : ~0ull;
          if ((r == 0 || key > last) && key < best) best = key;
        }
        best = ~pdr::wave_max_u64(~best);
        if (lane == 0) {
          outd[wave][r] = __uint_as_float(static_cast<unsigned>(best >> 32));
          outi[wave][r] = static_cast<int>(static_cast<unsigned>(best));
        }
        last = best;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < K) {
      const float dk = outd[wave][lane];
      const int ik = outi[wave][lane];
      const size_t o = (static_cast<size_t>(b) * nq + j) * K + lane;
      dists[o] = dk;
      idx[o] = static_cast<IdxT>(ik);
      if (nn) {
        nn[o * 3 + 0] = c[ik * 3 + 0];
        nn[o * 3 + 1] = c[ik * 3 + 1];
        nn[o * 3 + 2] = c[ik * 3 + 2];
      }
      if (wgt) {
        // group_knn's weights (pointnet2_utils.py:500-503), normalisation summed in ascending-k order as above
        const float rk = 1.0f / (dk + 1e-8f);
        outr[wave][lane] = rk;
        __builtin_amdgcn_wave_barrier();
        float norm = 0.0f;
        for (int t = 0; t < K; ++t) norm += outr[wave][t];
        wgt[o] = rk / norm;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// returns true when the wave-per-query kernel took the call
template <typename IdxT>
bool launch_knn_wave(const float* x, const float* y, int B, int n1, int n2, int K, float* dists, IdxT* idx, float* nn,
                     float* wgt, hipStream_t s) {
  const bool on = pdr::option(pdr::OPT_KNN_WAVE) != 0;
  if (!on || K > 8 || n2 < 64 || n2 > 1024) return false;
  // queries per wave: amortise the register fill of the cloud while the launch keeps >= 1024 workgroups (measured at
  // 2048 x 1024, B = 32: 4 / 8 / 16 queries per wave 60.6 / 60.1 / 60.3 us, 32: 88.6 us)
  int qpw = 16;
  while (qpw > 1 && static_cast<long long>(B) * ((n1 + 4 * qpw - 1) / (4 * qpw)) < 1024) qpw >>= 1;
  const dim3 grid((n1 + 4 * qpw - 1) / (4 * qpw), B);
#define PDR_KW(NCH)                                                                                          \
  hipLaunchKernelGGL((knn_wave_kernel<NCH, IdxT>), grid, dim3(256), 0, s, x, y, n1, n2, K, qpw, dists, idx, nn, wgt)
  if (n2 <= 64) PDR_KW(1);
  else if (n2 <= 128) PDR_KW(2);
  else if (n2 <= 256) PDR_KW(4);
  else if (n2 <= 512) PDR_KW(8);
  else PDR_KW(16);
#undef PDR_KW
  return true;
}

// ---- K = 1 (Chamfer) ------------------------------------------------------------------------
// chamfer_loss_new.py:149-167 asks knn_points(K=1) twice (x -> y and y -> x); 8.39 M pair evaluations per
// 2048^2 cloud pair, so the kernel is bound by VALU ISSUE SLOTS per pair, not by memory.  The generic kernel above
// spends ~9 slots per pair (6 distance + compare + 2 selects) plus one LDS read.  Here:
//   * a thread owns 2 QP queries held as QP float PAIRS: the distance expression runs on packed fp32
//     (v_pk_add / v_pk_mul / v_pk_fma: two queries per instruction, each lane IEEE-exact, same ACC3 expression
//     tree as the generic kernel -> bit-identical distances);
//   * every LDS point (one ds_read, broadcast) serves all 2 QP queries;
//   * no per-pair index bookkeeping: per chunk of CH points only the running minimum (v_min / v_min3) is kept,
//     and once per chunk `chunk_min < best` (STRICT) records the chunk; when the scan is over the ONE recorded
//     chunk of each query is rescanned in ascending order with strict `<`, which returns the first (lowest-index)
//     point attaining the minimum = "first minimum wins" (chamfer3D.cu:26-129, pdr_hip.h knn contract).
// ~4 issue slots per pair.  blockIdx.z selects the direction so both searches of a Chamfer evaluation are one launch.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int QP, typename IdxT>
__global__ __launch_bounds__(256) void nn1_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                  int na, int nb, float* __restrict__ da,
                                                  IdxT* __restrict__ ia, float* __restrict__ db,
                                                  IdxT* __restrict__ ib) {
  constexpr int CH = 8;
  constexpr int NQ = 2 * QP;                       // queries per thread
  __shared__ float4 tile[kTile];
  const int dir = blockIdx.z;
  const float* queries = dir ? xb : xa;
  const float* cloud = dir ? xa : xb;
  const int nq = dir ? nb : na, nc = dir ? na : nb;
  float* dists = dir ? db : da;
  IdxT* idx = dir ? ib : ia;
  if (static_cast<int>(blockIdx.x) * 256 * NQ >= nq) return;   // uniform: the grid covers max(na, nb)
  const int b = blockIdx.y;
  const float* c = cloud + static_cast<size_t>(b) * nc * 3;
  // thread t owns queries q0 + t + 256 i (coalesced loads / stores); out-of-range slots shadow the last query
  const int q0 = blockIdx.x * 256 * NQ + threadIdx.x;
  f32x2 qx[QP], qy[QP], qz[QP], best[QP];
  int bchunk[NQ];
#pragma unroll
  for (int p = 0; p < QP; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = min(q0 + 256 * (2 * p + h), nq - 1);
      const float* q = queries + (static_cast<size_t>(b) * nq + j) * 3;
      qx[p][h] = q[0];
      qy[p][h] = q[1];
      qz[p][h] = q[2];
      bchunk[2 * p + h] = 0;
    }
    best[p] = f32x2{__builtin_inff(), __builtin_inff()};
  }
  for (int k0 = 0; k0 < nc; k0 += kTile) {
    const int kn = (nc - k0) < kTile ? (nc - k0) : kTile;
    const int kpad = (kn + CH - 1) / CH * CH;
    __syncthreads();
    for (int t = threadIdx.x; t < kpad; t += 256) {
      const float* s = c + static_cast<size_t>(k0 + min(t, kn - 1)) * 3;
      // slots beyond the cloud: +inf coordinates -> distance +inf, never below any minimum
      tile[t] = t < kn ? make_float4(s[0], s[1], s[2], 0.0f)
                       : make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f);
    }
    __syncthreads();
    for (int t0 = 0; t0 < kpad; t0 += CH) {
      f32x2 cm[QP];
#pragma unroll
      for (int p = 0; p < QP; ++p) cm[p] = f32x2{__builtin_inff(), __builtin_inff()};
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const float4 pt = tile[t0 + u];
#pragma unroll
        for (int p = 0; p < QP; ++p) {
          const f32x2 dx = qx[p] - pt.x, dy = qy[p] - pt.y, dz = qz[p] - pt.z;
          f32x2 d = dx * dx;                                   // PDR_ACC3 on both lanes of the pair
          d = __builtin_elementwise_fma(dy, dy, d);
          d = __builtin_elementwise_fma(dz, dz, d);
          cm[p] = __builtin_elementwise_min(cm[p], d);
        }
      }
#pragma unroll
      for (int p = 0; p < QP; ++p) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool up = cm[p][h] < best[p][h];               // strict: the EARLIEST chunk holding the minimum
          best[p][h] = up ? cm[p][h] : best[p][h];
          bchunk[2 * p + h] = up ? k0 + t0 : bchunk[2 * p + h];
        }
      }
    }
  }
  // index recovery: rescan each query's recorded chunk (CH points from global memory, ascending, strict <)
#pragma unroll
  for (int p = 0; p < QP; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = q0 + 256 * (2 * p + h);
      if (j >= nq) continue;
      const float x0 = qx[p][h], y0 = qy[p][h], z0 = qz[p][h];
      float lb = __builtin_inff();
      int li = -1;
      const int cs = bchunk[2 * p + h];
      for (int u = 0; u < CH; ++u) {
        const int k = cs + u;
        if (k < nc) {
          const float dx = x0 - c[k * 3 + 0], dy = y0 - c[k * 3 + 1], dz = z0 - c[k * 3 + 2];
          const float d = PDR_ACC3(dx, dy, dz);
          if (d < lb) {
            lb = d;
            li = k;
          }
        }
      }
      const size_t o = static_cast<size_t>(b) * nq + j;
      dists[o] = li < 0 ? 0.0f : lb;                           // empty cloud: pytorch3d padding (0, -1)
      idx[o] = static_cast<IdxT>(li);
    }
  }
}

#ifndef PDR_NN1_QP
#define PDR_NN1_QP 2     // query PAIRS per thread of the K = 1 kernel (lab builds: -DPDR_NN1_QP=4)
#endif

template <int K>
int launch_knn(const float* x, const float* y, int B, int n1, int n2, int Kout, float* dists,
               int64_t* idx, float* nn, hipStream_t s) {
  dim3 grid((n1 + 255) / 256, B);
  hipLaunchKernelGGL((nn_search_kernel<K, kAcc3, int64_t, true>), grid, dim3(256), 0, s, x, y, n1,
                     n2, Kout, dists, idx, nn);
  return pdr::check_launch();
}

}  // namespace

extern "C" int pdr_three_nn(const float* unknown, const float* known, int B, int n, int m,
                            float* dist2, int* idx, pdr_stream_t stream) {
  if (B < 0 || n < 0 || m < 0) return PDR_EINVAL;
  if (B == 0 || n == 0) return PDR_OK;
  if (!unknown || !dist2 || !idx || (m > 0 && !known)) return PDR_EINVAL;
  dim3 grid((n + 255) / 256, B);
  hipLaunchKernelGGL((nn_search_kernel<3, kSum3, int, false>), grid, dim3(256), 0,
                     pdr::as_stream(stream), unknown, known, n, m, 3, dist2, idx,
                     static_cast<float*>(nullptr));
  return pdr::check_launch();
}

extern "C" int pdr_knn_points(const float* x, const float* y, int B, int n1, int n2, int K,
                              float* dists, int64_t* idx, float* nn, pdr_stream_t stream) {
  if (B < 0 || n1 < 0 || n2 < 0 || K <= 0) return PDR_EINVAL;
  if (K > 32) return PDR_EUNSUPPORTED;
  if (B == 0 || n1 == 0) return PDR_OK;
  if (!x || !dists || !idx || (n2 > 0 && !y)) return PDR_EINVAL;
  hipStream_t s = pdr::as_stream(stream);
  if (K == 1 && !nn && n2 > 0) {
    // dedicated packed-math kernel (one direction of pdr_chamfer_nn); bit-identical results
    constexpr int QPB = 256 * 2 * PDR_NN1_QP;   // queries per workgroup
    hipLaunchKernelGGL((nn1_kernel<PDR_NN1_QP, int64_t>), dim3((n1 + QPB - 1) / QPB, B, 1), dim3(256), 0, s, x, y, n1,
                       n2, dists, idx, static_cast<float*>(nullptr), static_cast<int64_t*>(nullptr));
    return pdr::check_launch();
  }
  if (K == 1) return launch_knn<1>(x, y, B, n1, n2, K, dists, idx, nn, s);
  if (K <= n2 && launch_knn_wave<int64_t>(x, y, B, n1, n2, K, dists, idx, nn, nullptr, s)) return pdr::check_launch();
  if (K <= 4) return launch_knn<4>(x, y, B, n1, n2, K, dists, idx, nn, s);
  if (K <= 8) return launch_knn<8>(x, y, B, n1, n2, K, dists, idx, nn, s);
  if (K <= 16) return launch_knn<16>(x, y, B, n1, n2, K, dists, idx, nn, s);
  return launch_knn<32>(x, y, B, n1, n2, K, dists, idx, nn, s);
}

// Both nearest-neighbour searches of a Chamfer evaluation (chamfer_loss_new.py:149-150, 166-167) in ONE launch:
//   dist_xy[b,i] = min_j |x_i - y_j|^2, idx_xy = argmin (first minimum wins);  dist_yx / idx_yx the reverse.
// Identical to two pdr_knn_points(K = 1) calls, bit for bit.
extern "C" int pdr_chamfer_nn(const float* x, const float* y, int B, int n1, int n2, float* dist_xy,
                              int64_t* idx_xy, float* dist_yx, int64_t* idx_yx, pdr_stream_t stream) {
  if (B < 0 || n1 <= 0 || n2 <= 0) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  if (!x || !y || !dist_xy || !idx_xy || !dist_yx || !idx_yx) return PDR_EINVAL;
  const int nmax = n1 > n2 ? n1 : n2;
  constexpr int QPB = 256 * 2 * PDR_NN1_QP;
  hipLaunchKernelGGL((nn1_kernel<PDR_NN1_QP, int64_t>), dim3((nmax + QPB - 1) / QPB, B, 2), dim3(256), 0,
                     pdr::as_stream(stream), x, y, n1, n2, dist_xy, idx_xy, dist_yx, idx_yx);
  return pdr::check_launch();
}

// knn_points for the fused network's feature propagation (group_knn, pointnet2_utils.py:487-514): the same search
// as pdr_knn_points with int32 indices (what pdr_gather_add reads) and the normalised inverse-squared-distance
// interpolation weights of :500-503 in the same pass.  Requires K <= n2 (no padding slots).
extern "C" int pdr_knn_group(const float* x, const float* y, int B, int n1, int n2, int K, float* dists,
                             int* idx, float* weights, pdr_stream_t stream) {
  if (B < 0 || n1 < 0 || n2 <= 0 || K <= 0 || K > n2) return PDR_EINVAL;
  if (K > 16) return PDR_EUNSUPPORTED;
  if (B == 0 || n1 == 0) return PDR_OK;
  if (!x || !y || !dists || !idx || !weights) return PDR_EINVAL;
  hipStream_t s = pdr::as_stream(stream);
  if (launch_knn_wave<int>(x, y, B, n1, n2, K, dists, idx, nullptr, weights, s)) return pdr::check_launch();
  const dim3 grid((n1 + 255) / 256, B);
#define PDR_KG(KK)                                                                                      \
  hipLaunchKernelGGL((nn_search_kernel<KK, kAcc3, int, false>), grid, dim3(256), 0, s, x, y, n1, n2, K,  \
                     dists, idx, static_cast<float*>(nullptr), weights)
  if (K <= 4) PDR_KG(4);
  else if (K <= 8) PDR_KG(8);
  else PDR_KG(16);
#undef PDR_KG
  return pdr::check_launch();
}

// ---- kNN backward (pytorch3d knn_points backward, norm 2; cf. chamfer3D.cu:155-195 for K = 1) ----
// One thread per query point: its K terms are summed in registers (k ascending, as the oracle) and
// written once; the scattered grad_y contributions use float atomics (pytorch3d does the same), so
// grad_y is reproducible only up to the summation order.
namespace {
__global__ __launch_bounds__(256) void knn_grad_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ y,
                                                       const int64_t* __restrict__ idx,
                                                       const float* __restrict__ gd, int n1, int n2, int K,
                                                       float* __restrict__ gx, float* __restrict__ gy) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n1) return;
  const float* q = x + (static_cast<size_t>(b) * n1 + j) * 3;
  const float* p = y + static_cast<size_t>(b) * n2 * 3;
  float* gp = gy + static_cast<size_t>(b) * n2 * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  float ax = 0.0f, ay = 0.0f, az = 0.0f;
  for (int t = 0; t < K; ++t) {
    const int64_t k = idx[(static_cast<size_t>(b) * n1 + j) * K + t];
    if (k < 0) continue;
    const float g = 2.0f * gd[(static_cast<size_t>(b) * n1 + j) * K + t];
    const float dx = g * (qx - p[k * 3 + 0]);
    const float dy = g * (qy - p[k * 3 + 1]);
    const float dz = g * (qz - p[k * 3 + 2]);
    ax += dx;
    ay += dy;
    az += dz;
    atomicAdd(gp + k * 3 + 0, -dx);
    atomicAdd(gp + k * 3 + 1, -dy);
    atomicAdd(gp + k * 3 + 2, -dz);
  }
  float* o = gx + (static_cast<size_t>(b) * n1 + j) * 3;
  o[0] = ax;
  o[1] = ay;
  o[2] = az;
}
}  // namespace

extern "C" int pdr_knn_points_grad(const float* x, const float* y, const int64_t* idx,
                                   const float* grad_dists, int B, int n1, int n2, int K, float* grad_x,
                                   float* grad_y, pdr_stream_t stream) {
  if (B < 0 || n1 < 0 || n2 < 0 || K <= 0) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  hipStream_t s = pdr::as_stream(stream);
  if (n2 > 0) {
    if (!grad_y) return PDR_EINVAL;
    if (hipMemsetAsync(grad_y, 0, sizeof(float) * static_cast<size_t>(B) * n2 * 3, s) != hipSuccess)
      return PDR_ELAUNCH;
  }
  if (n1 == 0) return PDR_OK;
  if (!x || !idx || !grad_dists || !grad_x || (n2 > 0 && !y)) return PDR_EINVAL;
  hipLaunchKernelGGL(knn_grad_kernel, dim3((n1 + 255) / 256, B), dim3(256), 0, s, x, y, idx, grad_dists, n1, n2,
                     K, grad_x, grad_y);
  return pdr::check_launch();
}
