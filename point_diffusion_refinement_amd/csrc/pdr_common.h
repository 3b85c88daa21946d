// pdr_common.h -- shared device/host helpers for libpdr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdr_hip.h"

// FP contraction model (see oracle/pdr_oracle.c header; build with
// -ffp-contract=off so these are the ONLY fusions):
//   SUM3: nvcc --fmad=true on  a*a + b*b + c*c   (pointnet2_ops, EMD kernels)
//   ACC3: nvcc --fmad=true on  dist += diff*diff (pytorch3d knn)
#define PDR_SUM3(a, b, c) __builtin_fmaf((c), (c), __builtin_fmaf((a), (a), (b) * (b)))
#define PDR_ACC3(a, b, c) __builtin_fmaf((c), (c), __builtin_fmaf((b), (b), (a) * (a)))

#define PDR_WAVE 64

namespace pdr {

void set_last_error(hipError_t e);

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(e);
    return PDR_ELAUNCH;
  }
  return PDR_OK;
}

inline hipStream_t as_stream(pdr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Process-wide kernel-selection options (pdr_set_option / pdr_get_option of include/pdr_hip.h; the table with names,
// defaults and ranges is in abi.hip).  Rounds 1-5 read ten of them from the environment inside the library; since ABI
// 0.2.0 the library never looks at the environment -- the caller sets them.
enum Opt {
  OPT_FUSED_WS, OPT_NARROW_KC32, OPT_FPS_WAVE, OPT_FPS_LEAN, OPT_KNN_WAVE, OPT_GN_FOLD_SMALL, OPT_WS_NARROW3,
  OPT_WS_XCD_ORDER, OPT_DEEP_CHUNKS, OPT_DEEP_KS, OPT_DEEP_JOBS32, OPT_DEEP_JOBS64, OPT_COUNT
};
int option(Opt o);

// POOL epilogue of the layer kernels: the GEMM output is the attention SCORE of every (query, neighbour) position;
// instead of being stored it is masked by the ball count, soft-maxed over the K neighbours of its query and used to
// weight the value rows (attention.py:83-96) -- the (P x D) score tensor never exists.
struct PoolArgs {
  const float* values;   // (P, ldv) value conv output (pre-GroupNorm)
  const float* vscale;   // (B, D) folded GroupNorm of the values, or NULL
  const float* vshift;
  const int* counts;     // (P / K) valid neighbours per query, or NULL = all
  float* out;            // (P / K, ldo)
  int ldv, ldo, K, v_relu;
};

// Twin blocks of pdr_gather_add_tiles_twin (fused_gather.hip): the per-query rows of a deduplicated block's first conv
// and their weighted moments, computed by extra workgroups of the launch that walks the tile subset.
struct GatherTwin {
  const int* idx0;     // (B, m) first neighbour of every query
  float* Y;            // (B m, ldy): the first conv of the per-query rows, every column
  const int* wrow0;    // (B): first query of cloud b whose row counts in the moments
  int ldy, n_main;     // n_main: workgroups of the main tiles (0: no twin blocks)
  float wmul;          // weight of the counted rows (K)
};

// Second problem of a PAIRED wave-specialised launch (fused_layer_ws.hip, round 6): the same layer (weights, bias, output
// width) over another set of rows -- the per-QUERY rows of a deduplicated block beside the tile subset of its
// per-neighbour rows -- computed by the workgroups gx .. gridDim.x - 1 of ONE launch instead of a launch of their own.
struct WsTwin {
  // [0] = the first problem (a copy of the launch's own arguments), [1] = the second: the kernel indexes these arrays
  // with a uniform 0 / 1 (a select between two kernel-argument STRUCTS did not survive instruction selection)
  pdr_layer_in_t in[2];
  float* Y[2];
  float* partial[2];
  int ldy[2], n_row_tiles[2];
  int gx;               // workgroups (grid.x) of the first problem; 0 = not a paired launch
};

// fused_layer_ws.hip: wave-specialised layer kernel; false = no instantiation for this tile variant
bool fused_layer_ws_supported(int variant, bool radd, bool gath, const pdr_layer_in_t& in, int Cin);
bool launch_fused_layer_ws(int variant, bool radd, bool gath, const pdr_layer_in_t& in, int Cin,
                           const float* Wt, int ldw, const float* bias, int Cout, float* Y, int ldy,
                           float* partial, int relu_col0, int n_row_tiles, int ncol, hipStream_t s,
                           bool split = false, const PoolArgs* pool = nullptr);
// one launch for (in, Y, partial) and (twin.in, twin.Y, twin.partial): plain or ball-gathered sources without a residual
// in the first problem, plain sources in the second; false = no paired instantiation for this tile variant
bool launch_fused_layer_ws_pair(int variant, bool gath, const pdr_layer_in_t& in, int Cin, const float* Wt, int ldw,
                                const float* bias, int Cout, float* Y, int ldy, float* partial, int relu_col0,
                                int n_row_tiles, int ncol, WsTwin twin, hipStream_t s);

// ---- DPP wave reductions (wave64, gfx9 row_shr / row_bcast) -------------------
// After wave_max_*: lane 63 holds the maximum; callers broadcast with readlane.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  // old = 0, bound_ctrl = false: lanes without a valid source read 0, which is
  // the identity for max over non-negative keys.
  return __builtin_amdgcn_update_dpp(0u, v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned long long u64_from(unsigned hi, unsigned lo) {
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_step_u64(unsigned long long v) {
  const unsigned lo = dpp_u32<CTRL>(static_cast<unsigned>(v));
  const unsigned hi = dpp_u32<CTRL>(static_cast<unsigned>(v >> 32));
  const unsigned long long o = u64_from(hi, lo);
  return o > v ? o : v;
}

// returns the wave-wide maximum, uniform across the wave
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_step_u64<0x111>(v);  // row_shr:1
  v = dpp_max_step_u64<0x112>(v);  // row_shr:2
  v = dpp_max_step_u64<0x114>(v);  // row_shr:4
  v = dpp_max_step_u64<0x118>(v);  // row_shr:8  -> lane 15 of each row = row max
  v = dpp_max_step_u64<0x142>(v);  // row_bcast:15 -> rows 1,3 absorb rows 0,2
  v = dpp_max_step_u64<0x143>(v);  // row_bcast:31 -> rows 2,3 absorb row 1
  const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(v), 63);
  const unsigned hi = __builtin_amdgcn_readlane(static_cast<unsigned>(v >> 32), 63);
  return u64_from(hi, lo);
}

// 32-bit form (keys >= 0; lanes without a source read 0 = the identity)
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_step_u32(unsigned v) {
  const unsigned o = dpp_u32<CTRL>(v);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = dpp_max_step_u32<0x111>(v);
  v = dpp_max_step_u32<0x112>(v);
  v = dpp_max_step_u32<0x114>(v);
  v = dpp_max_step_u32<0x118>(v);
  v = dpp_max_step_u32<0x142>(v);
  v = dpp_max_step_u32<0x143>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) {
  return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own 4 MB L2.
// This maps the hardware id to a logical id such that every XCD walks one CONTIGUOUS range of
// logical ids (so tiles that re-read the same per-cloud source rows share an L2).
__device__ __forceinline__ int xcd_contiguous(int bid, int nb) {
  constexpr int X = 8;
  const int x = bid % X, w = bid / X;
  const int q = nb / X, r = nb % X;
  return x * q + (x < r ? x : r) + w;
}

}  // namespace pdr
