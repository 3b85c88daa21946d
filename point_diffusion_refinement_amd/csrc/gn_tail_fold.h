// gn_tail_fold.h -- GroupNorm fold at the END of the kernel that produced the statistics.
//
// A GroupNorm between two 1x1 convs used to cost a launch of its own (pdr_gn_fold: per-tile partial moments ->
// per-(batch, channel) scale / shift).  On the dependency chains of the deep levels that launch is pure latency:
// rocprofv3 shows 23 us per fold inside the two-stream step against 5-7 us alone -- it waits for a workgroup slot
// next to the persistent layer kernels of the other stream -- and ~100 of them per step.  Here the producing kernel
// finishes the job itself, without any grid-wide wait:
//   * every workgroup, when it has written all its partial rows, adds the number of (row tile, column block) units it
//     produced for batch element b to ticket[b] (one lane per batch element: a single atomic instruction);
//   * the workgroup whose add completes batch element b (old + mine == units of b) folds b right there -- the other
//     workgroups have long exited or are still computing, nobody spins; batch elements complete at different times
//     and are folded by different workgroups in parallel;
//   * tickets are reset by the folding workgroup, so the same buffer serves every launch (and every graph replay).
// Visibility across XCDs WITHOUT a release / acquire fence: an agent-scope release fence on gfx950 is `buffer_wbl2`,
// a write-back of EVERY dirty line of the XCD's L2 -- the layer's whole output -- and one per workgroup made a
// reverse step 2.6x slower (measured: 9.15 -> 23.6 ms).  Instead the few bytes that cross workgroups do not live in
// the non-coherent L2 at all: partial rows are written with agent-scope (write-through, `sc1`) stores
// (pdr::store_partial), a workgroup waits for its own stores to be acknowledged (s_waitcnt vmcnt(0) + barrier)
// before its ticket add, and the folding workgroup reads the rows with agent-scope loads.  Tickets are agent-scope
// atomics throughout.
// Sums are taken in double in a fixed order (row slices ascending, then slices, then the group's channels ascending):
// deterministic, independent of which workgroup folds.
#pragma once
#include "pdr_common.h"

namespace pdr {

// One (sum, sum of squares) entry of a per-tile partial row, written through to the agent-coherent level so that
// another XCD's workgroup can read it without an L2 write-back (see above); plain consumers (pdr_gn_fold in a later
// launch) read it like any other memory.
__device__ __forceinline__ void store_partial(float* o, float s1, float s2) {
  __hip_atomic_store(o, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(o + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float2 load_partial(const float* q) {
  const unsigned long long v =
      __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float(static_cast<unsigned>(v)), __uint_as_float(static_cast<unsigned>(v >> 32)));
}

// host-side validation of a fold request against the launch that carries it
inline int check_fold(const pdr_fold_t& f, const float* partial, int Cout, int B) {
  if (!f.ticket || !f.scale || !f.shift || !partial || f.C0 <= 0 || f.col0 < 0 || f.col0 + f.C0 > Cout || f.G <= 0 ||
      f.n <= 0.0)
    return PDR_EINVAL;
  if (f.part1 ? (f.C1 <= 0 || f.tpb1 <= 0 || f.ldp1 < f.C1) : f.C1 != 0) return PDR_EINVAL;
  const int C = f.C0 + f.C1;
  if (f.Cn < 0 || f.Cn > C || (f.Cn > 0 && (f.Cn % f.G != 0 || !f.gamma || !f.beta))) return PDR_EINVAL;
  if (B > 64 || (f.Cn > 0 && f.Cn / f.G > 256)) return PDR_EUNSUPPORTED;   // one ticket lane per batch element
  return PDR_OK;
}

// Up to two statistics sources covering C = C0 + C1 channels in order (the attention score GroupNorm normalises
// [q.expand(K) | key]: its q half comes from the layer that carries the fold, the key half from an earlier launch).
// The fold of batch element b by NT threads; `lds` >= 16 * NT bytes of scratch.
template <int NT>
__device__ __forceinline__ void fold_batch_element(const pdr_fold_t& f, const float* __restrict__ partial, int ldp,
                                                   int tpb, int b, double* lds) {
  const int tid = threadIdx.x;
  const int C = f.C0 + f.C1;
  const int cpg = f.Cn > 0 ? f.Cn / f.G : 1;
  // channel blocks of CB <= NT channels made of whole groups; S row slices per channel
  const int CB = C <= NT ? C : (NT / cpg) * cpg;
  for (int c0 = 0; c0 < C; c0 += CB) {
    const int nc = min(CB, C - c0);
    const int S = max(1, NT / nc);
    const int cl = tid % nc, sl = tid / nc;
    double s1 = 0.0, s2 = 0.0;
    if (sl < S) {
      const int c = c0 + cl;
      const bool own = c < f.C0;
      const float* src = own ? partial : f.part1;
      const int ld = own ? ldp : f.ldp1, ntile = own ? tpb : f.tpb1, col = own ? f.col0 + c : c - f.C0;
      const float* q = src + (static_cast<long>(b) * ntile * ld + col) * 2;
      const long stride = static_cast<long>(ld) * 2;
      for (int t = sl; t < ntile; t += 4 * S) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tt = t + u * S;
          v[u] = tt < ntile ? load_partial(q + tt * stride) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s1 += v[u].x;
          s2 += v[u].y;
        }
      }
      const double mult = own ? f.mult0 : f.mult1;
      s1 *= mult;
      s2 *= mult;
    }
    __syncthreads();                       // (previous block's readers are done with lds)
    if (sl < S) {
      lds[(sl * nc + cl) * 2 + 0] = s1;
      lds[(sl * nc + cl) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < nc) {
      double a1 = 0.0, a2 = 0.0;
      for (int k = 0; k < S; ++k) {
        a1 += lds[(k * nc + tid) * 2 + 0];
        a2 += lds[(k * nc + tid) * 2 + 1];
      }
      // channel totals behind the slice sums (S * nc <= NT entries used, NT more available)
      lds[(NT + tid) * 2 + 0] = a1;
      lds[(NT + tid) * 2 + 1] = a2;
    }
    __syncthreads();
    if (tid < nc) {
      const int c = c0 + tid;
      float sc = 1.0f, sh = 0.0f;
      if (c < f.Cn) {
        const int g0 = (c / cpg) * cpg - c0;
        double g1 = 0.0, g2 = 0.0;
        for (int j = 0; j < cpg; ++j) {
          g1 += lds[(NT + g0 + j) * 2 + 0];
          g2 += lds[(NT + g0 + j) * 2 + 1];
        }
        const double cnt = static_cast<double>(f.n) * cpg;
        const double mean = g1 / cnt;
        double var = g2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(f.eps)));
        sc = rstd * f.gamma[c];
        sh = __builtin_fmaf(-sc, static_cast<float>(mean), f.beta[c]);
      }
      f.scale[static_cast<long>(b) * C + c] = sc;
      f.shift[static_cast<long>(b) * C + c] = sh;
    }
  }
}

// Called by ALL NT threads of a workgroup after its last partial row is written (no early return may skip it).
// units(b): (row tile, column block) units of batch element b this workgroup produced; units_per_b: all of them.
// nB <= 64.  `lds`: 32 * NT + 16 bytes of scratch, 8-byte aligned (the kernels hand over their dead staging buffers).
template <int NT, typename UnitsFn>
__device__ __forceinline__ void tail_fold(const pdr_fold_t& f, const float* __restrict__ partial, int ldp, int tpb,
                                          int nB, int units_per_b, UnitsFn units, double* lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through partial-row stores are acknowledged
  __syncthreads();
  unsigned long long* done = reinterpret_cast<unsigned long long*>(lds + 4 * NT);
  if (threadIdx.x < 64) {
    const int b = threadIdx.x;
    const int mine = b < nB ? units(b) : 0;
    bool last = false;
    if (mine > 0) {
      const int old = __hip_atomic_fetch_add(f.ticket + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = old + mine == units_per_b;
      // every unit of b has arrived: reset for the next launch
      if (last) __hip_atomic_store(f.ticket + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long m = __ballot(last);
    if (threadIdx.x == 0) *done = m;
  }
  __syncthreads();
  unsigned long long m = *done;
  if (m == 0ull) return;                   // uniform
  while (m != 0ull) {
    const int b = __builtin_ctzll(m);
    m &= m - 1ull;
    fold_batch_element<NT>(f, partial, ldp, tpb, b, lds);
  }
}

}  // namespace pdr
