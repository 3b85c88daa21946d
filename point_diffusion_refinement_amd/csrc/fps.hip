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

#include <cstdlib>

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

// One WAVE per cloud (N <= 64 * PPT <= 2048): no workgroup barrier and no LDS exchange in the round.
//   * lane l, slot i owns the point of tie rank r = 64 i + l (k = (r % Q) R + bitrev(r / Q)): inside a lane the
//     slots are already in the reference's tie order, so "first strictly greater wins" needs no sorting;
//   * the distance update runs on PACKED fp32 (two points per v_pk_* instruction, each half IEEE-exact, the same
//     SUM3 expression tree -> identical bits);
//   * the wave arg-max is two 32-bit DPP reductions (value, then rank|index among the lanes holding the value)
//     instead of one on a 64-bit key; the winner's coordinates come back from LDS by index (one ds_read_b128).
// A round is ~7 VALU slots per point + ~150 cycles: 2048 -> 1024 in ~0.26 ms (the 4-wave form needs 0.49 ms
// because every round ends in a barrier + LDS exchange); the deeper levels of the chain shrink likewise.
typedef float fps_f2 __attribute__((ext_vector_type(2)));

// (da, la) <- (db > da) ? (db, lb) : (da, la)
__device__ __forceinline__ void fps_combine(float& da, unsigned& la, float db, unsigned lb) {
  asm volatile(
      "v_cmp_gt_f32 vcc, %2, %0\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %0, %0, %2, vcc\n\t"
      "v_cndmask_b32 %1, %1, %3, vcc"
      : "+v"(da), "+v"(la)
      : "v"(db), "v"(lb)
      : "vcc");
}
// two independent combines, interleaved so that neither mask is read in the two issue slots after its write
__device__ __forceinline__ void fps_combine2(float& da, unsigned& la, float db, unsigned lb, float& dc,
                                             unsigned& lc, float dd, unsigned ld) {
  unsigned long long m;
  asm volatile(
      "v_cmp_gt_f32 vcc, %5, %0\n\t"
      "v_cmp_gt_f32 %4, %7, %2\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32 %0, %0, %5, vcc\n\t"
      "v_cndmask_b32 %1, %1, %6, vcc\n\t"
      "v_cndmask_b32 %2, %2, %7, %4\n\t"
      "v_cndmask_b32 %3, %3, %8, %4"
      : "+v"(da), "+v"(la), "+v"(dc), "+v"(lc), "=&s"(m)
      : "v"(db), "v"(lb), "v"(dd), "v"(ld)
      : "vcc");
}

// first tree level on two slot pairs (2H, 2H+1) and (2H+2, 2H+3): the slot numbers are immediates
template <int H>
__device__ __forceinline__ void fps_first2(const fps_f2& a, const fps_f2& b, float& bd0, unsigned& bs0, float& bd1,
                                           unsigned& bs1) {
  unsigned long long m;
  asm volatile(
      "v_cmp_gt_f32 vcc, %6, %5\n\t"
      "v_cmp_gt_f32 %4, %8, %7\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32 %0, %5, %6, vcc\n\t"
      "v_cndmask_b32 %1, %9, %10, vcc\n\t"
      "v_cndmask_b32 %2, %7, %8, %4\n\t"
      "v_cndmask_b32 %3, %11, %12, %4"
      : "=&v"(bd0), "=&v"(bs0), "=&v"(bd1), "=&v"(bs1), "=&s"(m)
      : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]), "n"(2 * H), "n"(2 * H + 1), "n"(2 * H + 2), "n"(2 * H + 3)
      : "vcc");
}
template <int H>
__device__ __forceinline__ void fps_first1(const fps_f2& a, float& bd0, unsigned& bs0) {
  asm volatile(
      "v_cmp_gt_f32 vcc, %3, %2\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %0, %2, %3, vcc\n\t"
      "v_cndmask_b32 %1, %4, %5, vcc"
      : "=&v"(bd0), "=&v"(bs0)
      : "v"(a[0]), "v"(a[1]), "n"(2 * H), "n"(2 * H + 1)
      : "vcc");
}
template <int H, int NH>
__device__ __forceinline__ void fps_level0(const fps_f2 (&tmp)[NH], float (&bd)[NH], unsigned (&bs)[NH]) {
  if constexpr (H + 1 < NH) {
    fps_first2<H>(tmp[H], tmp[H + 1], bd[H], bs[H], bd[H + 1], bs[H + 1]);
    fps_level0<H + 2, NH>(tmp, bd, bs);
  } else if constexpr (H < NH) {
    fps_first1<H>(tmp[H], bd[H], bs[H]);
  }
}

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_wave_kernel(const float* __restrict__ xyz, int N, int m, int R,
                                                     int Rbits, int Q, int* __restrict__ idxs) {
  static_assert(PPT % 2 == 0, "points are processed in pairs");
  constexpr int W = T / 64;
  extern __shared__ __attribute__((aligned(16))) float4 cloud4[];
  __shared__ unsigned long long slots[2][W > 1 ? W : 1];   // per-wave (value, rank) of a round, two parities
  const int b = blockIdx.x;
  const int lane = threadIdx.x;                        // thread index in the workgroup (slot stride = T)
  const float* p = xyz + static_cast<size_t>(b) * N * 3;
  int* out = idxs + static_cast<size_t>(b) * m;
  for (int i = lane; i < N; i += T) cloud4[i] = make_float4(p[i * 3 + 0], p[i * 3 + 1], p[i * 3 + 2], 0.0f);

  // slot i of thread l holds the point of tie rank r = T i + l; only coordinates and running distances stay in
  // registers -- the slot number is an immediate in the arg-max tree and the point index is recovered from the
  // winning rank with scalar arithmetic once per round
  fps_f2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], tmp[PPT / 2];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int r = i * T + lane;
    const int hi = r / Q, lo = r - hi * Q;
    const int k = lo * R + static_cast<int>(pdr::bitrev(static_cast<unsigned>(hi), Rbits));
    const bool valid = hi < R && k < N;
    const int kk = valid ? k : 0;
    const float x = p[kk * 3 + 0], y = p[kk * 3 + 1], z = p[kk * 3 + 2];
    const float mag = PDR_SUM3(x, y, z);
    // reference: `if (mag <= 1e-3) continue;` -- float promoted to double
    const float t0 = (!valid || static_cast<double>(mag) <= 1e-3) ? -1.0f : 1e10f;
    px[i >> 1][i & 1] = valid ? x : 0.0f;
    py[i >> 1][i & 1] = valid ? y : 0.0f;
    pz[i >> 1][i & 1] = valid ? z : 0.0f;
    tmp[i >> 1][i & 1] = t0;                           // padding: never selectable, never updated upward
  }
  if (lane == 0) out[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float4 c = cloud4[old];
    // ---- phase 1: all distance updates (the results ARE the new running minima).  Written stage by stage over
    // groups of G pairs so that consecutive instructions are independent: one wave per SIMD has nothing else to
    // hide the VALU latency of a dependent chain behind
    constexpr int G = (PPT / 2) < 8 ? (PPT / 2) : 8;
#pragma unroll
    for (int h0 = 0; h0 < PPT / 2; h0 += G) {
      fps_f2 dx[G], dy[G], dz[G], d[G];
#pragma unroll
      for (int g = 0; g < G; ++g) dy[g] = py[h0 + g] - c.y;
#pragma unroll
      for (int g = 0; g < G; ++g) dx[g] = px[h0 + g] - c.x;
#pragma unroll
      for (int g = 0; g < G; ++g) d[g] = dy[g] * dy[g];        // PDR_SUM3: fma(c, c, fma(a, a, b * b))
#pragma unroll
      for (int g = 0; g < G; ++g) dz[g] = pz[h0 + g] - c.z;
#pragma unroll
      for (int g = 0; g < G; ++g) d[g] = __builtin_elementwise_fma(dx[g], dx[g], d[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) d[g] = __builtin_elementwise_fma(dz[g], dz[g], d[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        // v_min_f32 directly (fminf / elementwise_min first canonicalise both operands: +1 VALU per point)
        fps_f2 r;
        asm("v_min_f32 %0, %1, %2" : "=v"(r[0]) : "v"(d[g][0]), "v"(tmp[h0 + g][0]));
        asm("v_min_f32 %0, %1, %2" : "=v"(r[1]) : "v"(d[g][1]), "v"(tmp[h0 + g][1]));
        tmp[h0 + g] = r;
      }
    }
    // ---- phase 2: this lane's arg-max as a TREE over its slots: combine(a, b) with a before b in slot order lets b
    // win only if STRICTLY greater = "first maximum in slot order", what the sequential strict-'>' scan returns.
    // The combines are spelled in assembly, two per block (one mask in VCC, one in an SGPR pair, each consumed >= 2
    // issue slots after it is written): the compiler's own rendering of the select pairs (v_max + v_cmp_eq to
    // re-derive the winner, an s_nop behind every v_cmp) needed ~900 instructions per round.
    float bd[PPT / 2];
    unsigned bs[PPT / 2];                              // winning slot of the sub-tree
    fps_level0<0, PPT / 2>(tmp, bd, bs);
#pragma unroll
    for (int st = 1; st < PPT / 2; st <<= 1) {
#pragma unroll
      for (int i = 0; i + st < PPT / 2; i += 4 * st) {
        if (i + 3 * st < PPT / 2) fps_combine2(bd[i], bs[i], bd[i + st], bs[i + st], bd[i + 2 * st], bs[i + 2 * st],
                                               bd[i + 3 * st], bs[i + 3 * st]);
        else fps_combine(bd[i], bs[i], bd[i + st], bs[i + st]);
      }
    }
    const float best = bd[0];
    // value first (non-negative floats order like their bit patterns; "no candidate" (-1) -> 0), then the tie rank
    // among the lanes that hold the winning value: smaller rank preferred -> reduce 0xFFFF - r with max
    const unsigned vb = best < 0.0f ? 0u : __float_as_uint(best);
    const unsigned vmax = pdr::wave_max_u32(vb);
    const unsigned rk = 0xFFFFu - (bs[0] * static_cast<unsigned>(T) + static_cast<unsigned>(lane));
    const unsigned cand = (vb == vmax && best >= 0.0f) ? rk : 0u;
    unsigned win = pdr::wave_max_u32(cand);            // uniform in the wave
    if constexpr (W > 1) {
      // one barrier per round: every wave publishes (value, rank); all read the W entries back
      unsigned long long* sl = slots[j & 1];
      if ((lane & 63) == 0) sl[lane >> 6] = pdr::u64_from(vmax, win);
      __syncthreads();
      unsigned long long key = sl[0];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        const unsigned long long o = sl[w];
        key = o > key ? o : key;
      }
      win = static_cast<unsigned>(key & 0xFFFFFFFFull);
    }
    // rank -> point index with scalar arithmetic (win == 0: no candidate at all -> index 0, as the reference)
    const int rw = 0xFFFF - static_cast<int>(win);
    const int hw = rw / Q, lw = rw - hw * Q;
    old = win == 0u ? 0 : lw * R + static_cast<int>(pdr::bitrev(static_cast<unsigned>(hw), Rbits));
    if (lane == 0) out[j] = old;
  }
}

template <int T, int PPT>
int launch_wave(const float* xyz, int B, int N, int m, int R, int Rbits, int Q, int* idx, hipStream_t s) {
  hipLaunchKernelGGL((fps_wave_kernel<T, PPT>), dim3(B), dim3(T), static_cast<size_t>(N) * sizeof(float4), s, xyz, N,
                     m, R, Rbits, Q, idx);
  return pdr::check_launch();
}

// ---- the resident kernel, instruction-lean (round 5) ------------------------------------------------------------
// With one wave per SIMD a round is bound by the NUMBER of VALU instructions it issues, not by their latency alone:
// the disassembly of fps_resident_kernel<256, 8> holds ~155 per round (72 for the eight distance updates incl. a
// canonicalising v_max in front of every v_min, 24 for the in-lane arg-max, 47 for the 64-bit DPP arg-max -- two
// v_mov_dpp, a 64-bit compare and two selects per step --, a dozen for the exchange) against 1,150 cycles measured.
// fps_wave_kernel above trimmed the updates but pays an integer division per round to turn its tie rank back into a
// point index, and ended up with as many instructions (532 vs 490 us).  This form keeps the resident kernel's
// (distance bits, tie key | index) key -- no division -- and removes the rest:
//   * the cloud lives in LDS as float4: one ds_read_b128 for the last pick instead of an address computation and
//     three ds_read_b32;
//   * distance updates on packed fp32 (two points per v_pk_add / v_pk_mul / v_pk_fma: each half IEEE-exact, the same
//     SUM3 tree -> the same bits), v_min_f32 spelled directly: 32 instructions for eight points instead of 72;
//   * the wave arg-max as two 32-bit reductions whose DPP shift is an operand modifier of the v_max_u32 itself (one
//     instruction per step): the value (non-negative floats order like their bit patterns), then the key among the
//     lanes that hold it: 12 + 6 instructions instead of 47.
// Same picks as fps_resident_kernel for every input (tests/test_ops_gpu.py::test_fps_index_exact runs both:
// option fps_lean = 0 selects the old kernel).
__device__ __forceinline__ unsigned fps_wave_max_u32(unsigned v) {
  // lanes without a DPP source are disabled for that step and keep their own value; lane 63 ends with the maximum.
  // (two wait states between a VALU write of a register and a DPP read of it: the assembler does not insert them)
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_lean_kernel(const float* __restrict__ xyz, int N, int m, int R, int Rbits,
                                                     int Q, int* __restrict__ idxs) {
  static_assert(PPT % 2 == 0, "points are updated in pairs");
  constexpr int W = T / 64;
  extern __shared__ __attribute__((aligned(16))) float4 lean_cloud[];
  __shared__ unsigned long long slots[2][W > 1 ? W : 1];   // per-wave key of a round, two parities
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* p = xyz + static_cast<size_t>(b) * N * 3;
  int* out = idxs + static_cast<size_t>(b) * m;
  for (int i = tid; i < N; i += T) lean_cloud[i] = make_float4(p[i * 3 + 0], p[i * 3 + 1], p[i * 3 + 2], 0.0f);

  float sx[PPT], sy[PPT], sz[PPT], st[PPT];
  unsigned low[PPT];  // (0xFFFF - tierank) << 16 | k ; larger = preferred
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * T;
    if (k < N) {
      sx[i] = p[k * 3 + 0];
      sy[i] = p[k * 3 + 1];
      sz[i] = p[k * 3 + 2];
      const float mag = PDR_SUM3(sx[i], sy[i], sz[i]);
      // reference: `if (mag <= 1e-3) continue;` -- float promoted to double
      st[i] = (static_cast<double>(mag) <= 1e-3) ? -1.0f : 1e10f;
      const unsigned rank = pdr::bitrev(static_cast<unsigned>(k % R), Rbits) * Q + k / R;
      low[i] = ((0xFFFFu - rank) << 16) | static_cast<unsigned>(k);
    } else {
      sx[i] = sy[i] = sz[i] = 0.0f;
      st[i] = -1.0f;  // padding: never selectable, never updated upward
      low[i] = 0u;
    }
  }
  // this thread's points by descending `low` (= ascending tie rank): "first strictly greater wins" inside the thread
  // is then the reference's order
#pragma unroll
  for (int a = 0; a < PPT - 1; ++a) {
#pragma unroll
    for (int c = 0; c < PPT - 1 - a; ++c) {
      if (low[c] < low[c + 1]) {
        float t;
        unsigned u;
        t = sx[c]; sx[c] = sx[c + 1]; sx[c + 1] = t;
        t = sy[c]; sy[c] = sy[c + 1]; sy[c + 1] = t;
        t = sz[c]; sz[c] = sz[c + 1]; sz[c + 1] = t;
        t = st[c]; st[c] = st[c + 1]; st[c + 1] = t;
        u = low[c]; low[c] = low[c + 1]; low[c + 1] = u;
      }
    }
  }
  fps_f2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], tmp[PPT / 2];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    px[i >> 1][i & 1] = sx[i];
    py[i >> 1][i & 1] = sy[i];
    pz[i >> 1][i & 1] = sz[i];
    tmp[i >> 1][i & 1] = st[i];
  }
  if (tid == 0) out[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float4 c = lean_cloud[old];
    constexpr int G = PPT / 2;
    fps_f2 dx[G], dy[G], dz[G], d[G];
#pragma unroll
    for (int g = 0; g < G; ++g) dy[g] = py[g] - c.y;
#pragma unroll
    for (int g = 0; g < G; ++g) dx[g] = px[g] - c.x;
#pragma unroll
    for (int g = 0; g < G; ++g) d[g] = dy[g] * dy[g];          // PDR_SUM3: fma(c, c, fma(a, a, b * b))
#pragma unroll
    for (int g = 0; g < G; ++g) dz[g] = pz[g] - c.z;
#pragma unroll
    for (int g = 0; g < G; ++g) d[g] = __builtin_elementwise_fma(dx[g], dx[g], d[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) d[g] = __builtin_elementwise_fma(dz[g], dz[g], d[g]);
    float best = -1.0f;
    unsigned bestlow = 0u;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      fps_f2 r;
      asm("v_min_f32 %0, %1, %2" : "=v"(r[0]) : "v"(d[g][0]), "v"(tmp[g][0]));
      asm("v_min_f32 %0, %1, %2" : "=v"(r[1]) : "v"(d[g][1]), "v"(tmp[g][1]));
      tmp[g] = r;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool gt = r[e] > best;
        best = gt ? r[e] : best;
        bestlow = gt ? low[2 * g + e] : bestlow;
      }
    }
    // best >= 0 -> monotone uint bits; "no candidate" (-1) -> 0
    const unsigned vb = best < 0.0f ? 0u : __float_as_uint(best);
    const unsigned vmax = fps_wave_max_u32(vb);                 // uniform
    // (a candidate at distance +0.0 has value bits 0 like "no candidate", but keeps its key)
    const unsigned cand = (vb == vmax && !(best < 0.0f)) ? bestlow : 0u;
    unsigned wlow = fps_wave_max_u32(cand);
    unsigned long long key = pdr::u64_from(vmax, wlow);
    if constexpr (W > 1) {
      unsigned long long* sl = slots[j & 1];
      if ((tid & 63) == 0) sl[tid >> 6] = key;
      __syncthreads();
      unsigned long long r = sl[0];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        const unsigned long long o = sl[w];
        r = o > r ? o : r;
      }
      key = r;
    }
    old = static_cast<int>(key & 0xFFFFull);
    if (tid == 0) out[j] = old;
  }
}

template <int T, int PPT>
int launch_lean(const float* xyz, int B, int N, int m, int R, int Rbits, int Q, int* idx, hipStream_t s) {
  hipLaunchKernelGGL((fps_lean_kernel<T, PPT>), dim3(B), dim3(T), static_cast<size_t>(N) * sizeof(float4), s, xyz, N, m,
                     R, Rbits, Q, idx);
  return pdr::check_launch();
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
  // Rank-ordered variant (fps_wave_kernel).  Measured on MI355X, B = 32 (tools/lab/fps_time.py), us per call:
  //                      2048->1024  1024->256  256->64  3072->1024
  //   4 waves, barrier        490        105       25        566      (fps_resident_kernel, default above 256 slots)
  //   1 wave, 32 slots/lane   696        121       20         --      (one wave issues <= 1 instruction / ~4-5 cycles)
  //   4 waves, rank-ordered   532        127       20        665      (fewer VALU slots, but the round is a LATENCY
  //                                                                    chain: LDS read -> update -> tree -> 2 DPP
  //                                                                    reductions -> exchange -> index recovery)
  // so it is used where it wins (<= 256 slots: no barrier at all).  Option fps_wave:
  // 0 = never, 2 = for every size up to 4096 slots (A/B and test runs).
  const int wave_mode = pdr::option(pdr::OPT_FPS_WAVE);
  const bool use_wave = wave_mode == 2 || (wave_mode == 1 && static_cast<long>(R) * Q <= 256);
  // the wave kernel keeps the cloud in N * 16 bytes of dynamic LDS next to its static exchange slots: stay inside
  // the 64 KiB a launch gets without raising hipFuncAttributeMaxDynamicSharedMemorySize, else the resident kernel
  const bool wave_fits = static_cast<size_t>(N) * sizeof(float4) + 256 <= 64 * 1024;
  if (use_wave && wave_fits && static_cast<long>(R) * Q <= 4096) {
    // rank-ordered slots + packed distances + tree arg-max (fps_wave_kernel): ONE wave while a lane holds <= 4
    // slots (no barrier at all), four waves (one per SIMD: a single wave issues at most one instruction every
    // ~4-5 cycles, measured 680 ns per round with 32 slots per lane) above that
    const long slots = static_cast<long>(R) * Q;
    if (slots <= 128) return launch_wave<64, 2>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (slots <= 256) return launch_wave<64, 4>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (slots <= 512) return launch_wave<256, 2>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (slots <= 1024) return launch_wave<256, 4>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (slots <= 2048) return launch_wave<256, 8>(xyz, B, N, m, R, Rbits, Q, idx, s);
    return launch_wave<256, 16>(xyz, B, N, m, R, Rbits, Q, idx, s);
  }
  // instruction-lean resident kernel (fps_lean_kernel): clouds whose float4 image fits the 64 KiB a launch gets without
  // raising the dynamic LDS limit, an even number of points per thread.  Option fps_lean = 0: the round-1 resident kernel.
  const bool lean = pdr::option(pdr::OPT_FPS_LEAN) != 0;
  if (lean && N > 128 && static_cast<size_t>(N) * sizeof(float4) + 256 <= 64 * 1024) {
    if (N <= 512) return launch_lean<256, 2>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (N <= 1024) return launch_lean<256, 4>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (N <= 2048) return launch_lean<256, 8>(xyz, B, N, m, R, Rbits, Q, idx, s);
    if (N <= 3072) return launch_lean<256, 12>(xyz, B, N, m, R, Rbits, Q, idx, s);
    return launch_lean<256, 16>(xyz, B, N, m, R, Rbits, Q, idx, s);
  }
  static_assert(kMaxResidentN >= 12288, "resident path must cover the LDS-resident range");
#define PDR_FPS_CASE(T, PPT) \
  if (N <= (T) * (PPT)) return launch_resident<T, PPT>(xyz, B, N, m, R, Rbits, Q, idx, s)
  // (512 / 1024 threads per cloud -- fewer points per thread against a wider exchange --, round 5, us per call at B = 32:
  // 2048->1024 492 / 525 / 885, 1024->256 105 / 124 / 215 for 256 / 512 / 1024 threads: not taken)
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
