// point_chain.hip -- a CHAIN of per-point layers (1x1 conv -> GroupNorm -> ReLU -> + embedding, ... , + residual) of one
// block as ONE launch: SURVEY 8(f)2, the fused per-level block kernel, for the per-point halves of the deep levels.
//
// Reference composition: Mlp_plus_t_emb (pointnet2_modules.py:69-174) as PointnetKnnFPModule applies it to the
// interpolated features of a level (mlp2, :829-839), the query conv + first score conv of AttentionModule
// (attention.py:70-82).  Layer by layer (fused_layer.hip) such a chain is conv launch -> GroupNorm fold launch -> conv
// launch -> fold launch -> activation launch: five dependent launches of 5-30 us for a few microseconds of MFMA work at
// the 16- / 64- / 256-point levels (profiles/r5_timeline.json: the chip holds one small kernel for 1.3 ms of a step).
//
// Why the folds could not move into their neighbours before and why they can here.  The layer kernels
// split the ROWS of a layer among workgroups, so a GroupNorm needs every workgroup's moments: a fold inside a running
// kernel is a dependent global round trip or two per workgroup, more than the launch it replaces.  A per-point layer of a
// deep level has so few rows per cloud (<= 256) that a workgroup can own ALL rows of its cloud for a block of output
// COLUMNS -- and a GroupNorm group is a set of adjacent channels over all rows of ONE cloud: with column blocks that are
// whole groups the statistics, the fold and the activation of a workgroup's outputs are LOCAL.  No partial rows, no
// fold, no cross-workgroup reduction at all.  What crosses workgroups is the activated output itself: the G workgroups
// of a cloud (a cluster) publish their column blocks write-through, meet at one counter per cloud and read each other's
// blocks as the next layer's input (the agent-scope hand-off of the guide: sc1 payload, drained, relaxed counter, one
// acquire).  One hand-off per layer boundary instead of two launches + a fold.
//
// Geometry: B clouds x G workgroups (G = 8 at B = 32: 256 workgroups, one per CU, a cluster on ONE XCD when B % 8 == 0 --
// placement is a speed matter only), 512 threads: four waves multiply, four stage the next chunk.  A workgroup computes n rows x w columns of every layer
// (w = Cout / G) with v_mfma_f32_16x16x4_f32 in the TRANSPOSED orientation (weights as the A operand, rows as the B
// operand): a lane then holds four consecutive channels of one row, i.e. the epilogue stores 16-byte row pieces straight
// from the accumulators and the moments of a column are an in-lane sum over row tiles + a 16-lane reduction.  Rows and
// weights are staged through LDS in 32-channel chunks, double-buffered, one barrier per chunk; row stride 36 floats and
// weight-row stride 144 floats make every operand read of the 16 x 4 / 4 x 16 fragments conflict-free.
// Exact fp32 (same MFMA family as the layer kernels); results equal the layer-by-layer chain up to fp32 summation order
// (the GroupNorm sums are taken in double from per-lane fp32 sums of at most 16 rows).
#include "pdr_common.h"

#include <mutex>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;        // input channels per staged chunk
constexpr int LDA = KC + 4;   // floats per staged row: (row * 36 + k) % 64 is distinct over 16 rows x 4 k
constexpr int LDW = 144;      // floats per staged weight row: (k * 144 + col) % 64 is distinct over 4 k x 16 columns
constexpr int MAXW = 128;     // staged weight columns per workgroup and layer (main + residual block)
constexpr int SPIN_LIMIT = 1 << 22;

#ifdef PDR_LAB_TRACE
// development probe (tools/lab/chain_trace.py): s_memtime stamps of thread 0 of workgroup 0, slot 0 = start, 8 per layer
__device__ unsigned long long pdr_lab_chain[64];
#define PDR_CT(slot)                                                                            \
  do {                                                                                          \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 64) pdr_lab_chain[slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PDR_CT(slot) do {} while (0)
#endif

__device__ __forceinline__ void store_sc1(float* p, f32x4 v) {
  // write-through (sc1) 16-byte store: the payload of an in-launch hand-off (no release fence needed)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int NMAX>
struct ChainMem {
  __attribute__((aligned(16))) float As[2][NMAX][LDA];
  __attribute__((aligned(16))) float Ws[2][KC][LDW];
  double red[4][MAXW][2];
  double csum[MAXW][2];
  float ss[MAXW][2];
};

// RT / CT: row tiles / column tiles of 16 per wave, COMPILE-TIME (the launcher picks the smallest instantiation that
// covers every layer): a tile this wave does not own in some layer is multiplied anyway -- on in-bounds LDS data that
// means nothing -- and dropped at the statistics / the store.  (Per-tile guards inside the multiply loop compiled to
// a branch and an lgkmcnt(0) around every MFMA: +100 us per launch.)
template <int NMAX, int RT, int CT>
__global__ __launch_bounds__(512, 1) void point_chain_kernel(pdr_point_chain_t P, int B, int n, int G, int place) {
  constexpr int APT = NMAX / 32;                 // float4 of A per thread and chunk
  constexpr int WPT = KC * (MAXW / 4) / 256;     // float4 of W per thread and chunk
  __shared__ ChainMem<NMAX> sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Waves 0-3 multiply (and run the epilogue: they own the accumulators), waves 4-7 stage: while chunk c is multiplied
  // out of one LDS stage the loaders commit chunk c + 1 into the other and issue the loads of chunk c + 2.  (Round-6
  // first version, every wave doing both in turn: 1,100 cycles of load issue + 960 of LDS writes per chunk next to
  // 4,500 of MFMAs at 256 rows per cloud -- the CU's vector-memory path moves 64 B per clock, a chunk is 40 KB.)
  const bool loader = wave >= 4;                 // uniform
  const int lt = tid & 255;                      // thread number within its role
  // cloud / column block of this workgroup.  place = 1: ids b, b + B, ... of a cloud's cluster (same XCD when B % 8 == 0)
  const int b = place ? static_cast<int>(blockIdx.x) % B : static_cast<int>(blockIdx.x) / G;
  const int g = place ? static_cast<int>(blockIdx.x) / B : static_cast<int>(blockIdx.x) % G;
  const int NRT = n >> 4;
  const int WR = NRT < 4 ? NRT : 4, WC = 4 / WR;
  const int wr = wave % WR, wc = wave / WR;
  // row tiles of this wave: wr, wr + WR, ... (< NRT); column tiles: wc, wc + WC, ... (< NCTm / NCTr)
  const int l16 = lane & 15, lq = lane >> 4;
  const long row_base = static_cast<long>(b) * n;

  f32x4 res[RT][CT];                             // residual tiles (raw columns of layer 0), added by the last layer
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) res[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  PDR_CT(0);
  long scratch_off = 0;                          // floats: activated output of the layer before
  const float* xin = nullptr;
  int xin_ld = 0;
  for (int l = 0; l < P.n_layers; ++l) {
    const pdr_chain_layer_t L = P.layer[l];
    const bool first = l == 0, last = l == P.n_layers - 1;
    const int res_cols = (first && P.residual) ? L.Cout - L.main_cols : 0;
    const int w_main = L.main_cols / G, w_res = res_cols / G;
    const int col_main0 = g * w_main, col_res0 = L.main_cols + g * w_res;
    const int wtot = w_main + w_res, wtot4 = wtot >> 2;
    const int NCTm = w_main >> 4, NCTr = w_res >> 4;
    // ---- chunk list: layer 0 walks its input segments, later layers the previous layer's published block
    int nch = 0;
    if (first) {
      for (int s = 0; s < P.n_seg; ++s) nch += (P.seg[s].C + KC - 1) / KC;
    } else {
      nch = (L.Cin + KC - 1) / KC;
    }
    // chunk in registers
    // (vector VALUES: float4 struct copies global -> private -> LDS pin the arrays in scratch)
    f32x4 ra[APT];
    f32x4 rw[WPT];
    int cvalid = 4;
    // this thread's weight quads of a chunk: (k row, local column quad) -> global column; fixed per layer
    int wk[WPT], wq[WPT], wg[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = lt + 256 * i;
      wk[i] = e / wtot4;
      wq[i] = e - wk[i] * wtot4;
      const int lc = 4 * wq[i];                                    // local column: [main block | residual block]
      wg[i] = lc < w_main ? col_main0 + lc : col_res0 + (lc - w_main);
    }
    // Chunk cursor (segment, channel offset inside it, global input channel of the segment's first channel): chunks are
    // fetched in order.  Per-thread byte offsets are fixed per (layer, segment): a fetch is a scalar base + 32-bit
    // offsets and the loads -- its instruction count is what a chunk period pays besides its MFMAs (first version, with
    // the addresses rebuilt per chunk: 1,100-1,400 cycles of issue per chunk next to 4,400 of MFMAs).
    int cs = 0, cks = 0, ccb = 0, off_seg = -1;
    unsigned aoff[APT], woff[WPT];
    bool partial = false;                                          // chunk in registers: a segment's last, short chunk
    const int c4 = lt & 7, r0 = lt >> 3;
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      woff[i] = static_cast<unsigned>(min(wk[i], KC - 1) * L.ldw + wg[i]) * 4u;
    auto fetch = [&](int) __attribute__((always_inline)) {
      const float* sp = first ? P.seg[cs].ptr : xin;
      const int sC = first ? P.seg[cs].C : L.Cin;
      const int sld = first ? P.seg[cs].ld : xin_ld;
      if (off_seg != cs) {                                         // uniform
        off_seg = cs;
#pragma unroll
        for (int i = 0; i < APT; ++i) aoff[i] = static_cast<unsigned>(min(r0 + 32 * i, n - 1) * sld + 4 * c4) * 4u;
      }
      const char* ab = reinterpret_cast<const char*>(sp + row_base * sld + cks);
      const char* wb = reinterpret_cast<const char*>(L.Wt + static_cast<long>(ccb + cks) * L.ldw);
      partial = cks + KC > sC;                                     // uniform
      // (nothing consumes a loaded value here: no wait is emitted and the loads fly through the multiply loop that
      // follows; the channel mask of a segment's last chunk is applied at the commit)
      if (!partial) {
#pragma unroll
        for (int i = 0; i < APT; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ab + aoff[i]);
#pragma unroll
        for (int i = 0; i < WPT; ++i) rw[i] = *reinterpret_cast<const f32x4*>(wb + woff[i]);
        cvalid = 4;
      } else {
        const int cl = 4 * c4;                                     // relative to cks
        const int clc = min(cl, ((sC + 3) & ~3) - 4 - cks);        // keep the 16-byte load inside the row
        cvalid = cl == clc ? max(0, min(4, sC - cks - clc)) : 0;   // valid channels of this thread's quad
        const int kmax = sC - cks;
#pragma unroll
        for (int i = 0; i < APT; ++i)
          ra[i] = *reinterpret_cast<const f32x4*>(ab + static_cast<unsigned>(min(r0 + 32 * i, n - 1) * sld + clc) * 4u);
#pragma unroll
        for (int i = 0; i < WPT; ++i)   // rows beyond the chunk meet zeroed A columns; clamped: finite weights
          rw[i] = *reinterpret_cast<const f32x4*>(
              wb + static_cast<unsigned>(min(min(wk[i], KC - 1), kmax - 1) * L.ldw + wg[i]) * 4u);
      }
      cks += KC;
      if (cks >= sC) {                                             // uniform: next segment
        ccb += sC;
        cks = 0;
        cs += 1;
        if (!first || cs >= P.n_seg) cs = first ? P.n_seg - 1 : 0; // (past the end: never fetched)
      }
    };
    auto commit = [&](int st) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < APT; ++i) {
        const int r = r0 + 32 * i;
        f32x4 v = ra[i];
        if (partial) {                                             // uniform
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = j < cvalid ? v[j] : 0.0f;
        }
        if (r < n) *reinterpret_cast<f32x4*>(&sm.As[st][r][4 * c4]) = v;
      }
#pragma unroll
      for (int i = 0; i < WPT; ++i)
        if (wk[i] < KC) *reinterpret_cast<f32x4*>(&sm.Ws[st][wk[i]][4 * wq[i]]) = rw[i];
    };

    // ---- accumulators start at the bias (lane: four consecutive columns of one row per tile)
    f32x4 acc[RT][CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ct = wc + WC * j;
      f32x4 bm = {0.0f, 0.0f, 0.0f, 0.0f}, br = {0.0f, 0.0f, 0.0f, 0.0f};
      if (L.bias && ct < NCTm) bm = *reinterpret_cast<const f32x4*>(L.bias + col_main0 + 16 * ct + 4 * lq);
      if (L.bias && first && ct < NCTr) br = *reinterpret_cast<const f32x4*>(L.bias + col_res0 + 16 * ct + 4 * lq);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        acc[i][j] = bm;
        if (first) res[i][j] = br;
      }
    }
    // LDS offsets of this lane's operand fragments (clamped into the staged arrays: a tile that does not exist reads
    // some other tile's data and is dropped later)
    int xrow[RT], wcol[CT], wcolr[CT];
#pragma unroll
    for (int i = 0; i < RT; ++i) xrow[i] = min((wr + WR * i) * 16 + l16, NMAX - 1);
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      wcol[j] = min(16 * (wc + WC * j) + l16, LDW - 1);
      wcolr[j] = min(w_main + 16 * (wc + WC * j) + l16, LDW - 1);
    }

    // (two loops, one per role, with the same number of barriers: the loaders' chunk registers are not live in the
    // multiply loop and the operand fragments are not live in the loaders')
    if (loader) {
      fetch(0);
      commit(0);
      if (nch > 1) fetch(1);
      __syncthreads();
      for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) {
          commit((c & 1) ^ 1);
          if (c + 2 < nch) fetch(c + 2);
        }
        __syncthreads();
      }
    } else {
      __syncthreads();
      PDR_CT(1 + 8 * l);
      for (int c = 0; c < nch; ++c) {
        const int st = c & 1;
#ifdef PDR_LAB_TRACE
        if (l == 0 && c < 8) PDR_CT(32 + 4 * c);
#endif
        if (first && res_cols > 0) {            // uniform: layer 0 also multiplies its residual column block
#pragma unroll
          for (int k4 = 0; k4 < KC / 4; ++k4) {
            const int k = 4 * k4 + lq;
            float xb[RT], wa[CT], war[CT];
#pragma unroll
            for (int i = 0; i < RT; ++i) xb[i] = sm.As[st][xrow[i]][k];
#pragma unroll
            for (int j = 0; j < CT; ++j) {
              wa[j] = sm.Ws[st][k][wcol[j]];
              war[j] = sm.Ws[st][k][wcolr[j]];
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int j = 0; j < CT; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb[i], acc[i][j], 0, 0, 0);
                res[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(war[j], xb[i], res[i][j], 0, 0, 0);
              }
          }
        } else {
#pragma unroll
          for (int k4 = 0; k4 < KC / 4; ++k4) {
            const int k = 4 * k4 + lq;
            float xb[RT], wa[CT];
#pragma unroll
            for (int i = 0; i < RT; ++i) xb[i] = sm.As[st][xrow[i]][k];
#pragma unroll
            for (int j = 0; j < CT; ++j) wa[j] = sm.Ws[st][k][wcol[j]];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int j = 0; j < CT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb[i], acc[i][j], 0, 0, 0);
          }
        }
#ifdef PDR_LAB_TRACE
        if (l == 0 && c < 8) PDR_CT(34 + 4 * c);
#endif
        __syncthreads();
#ifdef PDR_LAB_TRACE
        if (l == 0 && c < 8) PDR_CT(35 + 4 * c);
#endif
      }
    }

    PDR_CT(2 + 8 * l);
    // ---- epilogue: [ReLU] -> GroupNorm (local: this workgroup holds every row of its cloud for whole groups) ->
    // [ReLU] -> + embedding row (-> + residual) -> 16-byte row pieces
    const float lo_pre = L.relu_pre ? 0.0f : -__builtin_inff();
    const float lo_post = L.relu_post ? 0.0f : -__builtin_inff();
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], lo_pre);
    if (L.gamma) {
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int ct = wc + WC * j;
        float s1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const bool rok = wr + WR * i < NRT;          // (a row tile this wave does not own holds garbage)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float f = rok ? acc[i][j][r] : 0.0f;
            s1[r] += f;
            s2[r] = __builtin_fmaf(f, f, s2[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            s1[r] += __shfl_xor(s1[r], m, 64);
            s2[r] += __shfl_xor(s2[r], m, 64);
          }
        }
        if (!loader && l16 == 0 && ct < NCTm) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sm.red[wr][16 * ct + 4 * lq + r][0] = static_cast<double>(s1[r]);
            sm.red[wr][16 * ct + 4 * lq + r][1] = static_cast<double>(s2[r]);
          }
        }
      }
      __syncthreads();
      if (tid < w_main) {
        double a1 = 0.0, a2 = 0.0;
        for (int w = 0; w < WR; ++w) {
          a1 += sm.red[w][tid][0];
          a2 += sm.red[w][tid][1];
        }
        sm.csum[tid][0] = a1;
        sm.csum[tid][1] = a2;
      }
      __syncthreads();
      if (tid < w_main) {
        const int col = col_main0 + tid;
        float sc = 1.0f, sh = 0.0f;                    // channels behind the normalised range pass through
        if (col < L.Cn) {
          const int cpg = L.Cn / L.groups;
          const int g0 = (tid / cpg) * cpg;            // (column blocks are whole groups: host-checked)
          double g1 = 0.0, g2 = 0.0;
          for (int j = 0; j < cpg; ++j) {
            g1 += sm.csum[g0 + j][0];
            g2 += sm.csum[g0 + j][1];
          }
          // the arithmetic of gn_fold_kernel (fused_layer.hip)
          const double cnt = static_cast<double>(n) * cpg;
          const double mean = g1 / cnt;
          double var = g2 / cnt - mean * mean;
          if (var < 0.0) var = 0.0;
          const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(L.eps)));
          sc = rstd * L.gamma[col];
          sh = __builtin_fmaf(-sc, static_cast<float>(mean), L.beta[col]);
        }
        sm.ss[tid][0] = sc;
        sm.ss[tid][1] = sh;
      }
      __syncthreads();
    }
    PDR_CT(3 + 8 * l);
    float* dst = last ? P.out : P.scratch + scratch_off;
    const int dld = last ? P.ldo : L.main_cols;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int ct = wc + WC * j;
      if (!loader && ct < NCTm) {
        const int lc = 16 * ct + 4 * lq, col = col_main0 + lc;
        f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f}, ad = {0.0f, 0.0f, 0.0f, 0.0f};
        if (L.gamma) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sc[r] = sm.ss[lc + r][0];
            sh[r] = sm.ss[lc + r][1];
          }
        }
        if (L.add) ad = *reinterpret_cast<const f32x4*>(L.add + static_cast<long>(b) * L.add_ld + col);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          if (wr + WR * i < NRT) {
            const int row = (wr + WR * i) * 16 + l16;
            f32x4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = L.gamma ? __builtin_fmaf(acc[i][j][r], sc[r], sh[r]) : acc[i][j][r];
              v = fmaxf(v, lo_post) + ad[r];
              if (last && P.residual) v += res[i][j][r];
              z[r] = v;
            }
            float* q = dst + (row_base + row) * dld + col;
            if (last) *reinterpret_cast<f32x4*>(q) = z;
            else store_sc1(q, z);
          }
        }
      }
    }
    PDR_CT(4 + 8 * l);
    if (!last) {
      // ---- hand-off: every wave drains its write-through stores, one lane arrives at the cloud's counter and waits
      // for the other column blocks, one acquire drops this CU's stale lines; then plain loads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      PDR_CT(5 + 8 * l);
      if (tid == 0) {
        int* ctr = P.sync + b;
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (l + 1) * G;
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > SPIN_LIMIT) {                 // never hang the device: flag the launch and go on
            __hip_atomic_store(P.sync + 2 * B, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        PDR_CT(6 + 8 * l);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      PDR_CT(7 + 8 * l);
      xin = P.scratch + scratch_off;
      xin_ld = L.main_cols;
      scratch_off += static_cast<long>(B) * n * L.main_cols;
    }
  }
  // ---- leave the counters as they were found: the last workgroup of the cloud to finish zeroes them (every
  // workgroup of the cluster is past its last wait when it takes an exit ticket)
  if (tid == 0 && P.n_layers > 1) {
    const int old = __hip_atomic_fetch_add(P.sync + B + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == G - 1) {
      __hip_atomic_store(P.sync + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.sync + B + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct ChainPlan {
  int G;                // workgroups per cloud (0: unsupported)
  int ct;               // column tiles per wave the widest layer needs (main or residual block)
  long scratch_floats;  // inter-layer activations
};

bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

// PDR_OK + plan, PDR_EINVAL (argument error) or PDR_EUNSUPPORTED (outside the kernel's shapes: the caller runs the chain
// layer by layer)
int plan_chain(const pdr_point_chain_t& p, int B, int n, ChainPlan* out) {
  out->G = 0;
  out->ct = 0;
  out->scratch_floats = 0;
  if (B < 0 || n <= 0 || p.n_layers < 1 || p.n_layers > 4 || p.n_seg < 1 || p.n_seg > 3 || !p.out) return PDR_EINVAL;
  if (p.n_layers > 1 && (!p.scratch || !p.sync)) return PDR_EINVAL;
  int cin0 = 0;
  for (int s = 0; s < p.n_seg; ++s) {
    const pdr_chain_seg_t& g = p.seg[s];
    if (!g.ptr || g.C <= 0 || g.ld < ((g.C + 3) & ~3)) return PDR_EINVAL;
    if (g.ld % 4 != 0 || !aligned16(g.ptr)) return PDR_EUNSUPPORTED;
    cin0 += g.C;
  }
  for (int l = 0; l < p.n_layers; ++l) {
    const pdr_chain_layer_t& L = p.layer[l];
    if (!L.Wt || L.Cin <= 0 || L.Cout <= 0 || L.ldw < L.Cout || L.main_cols <= 0 || L.main_cols > L.Cout)
      return PDR_EINVAL;
    if (L.ldw % 4 != 0 || !aligned16(L.Wt) || (L.bias && !aligned16(L.bias))) return PDR_EUNSUPPORTED;
    if (L.Cin != (l == 0 ? cin0 : p.layer[l - 1].main_cols)) return PDR_EINVAL;
    const bool res0 = l == 0 && p.residual;
    if (!res0 && L.main_cols != L.Cout) return PDR_EINVAL;
    if (L.gamma) {
      if (!L.beta || L.groups <= 0 || L.Cn <= 0 || L.Cn > L.main_cols || L.Cn % L.groups != 0) return PDR_EINVAL;
    }
    if (L.add && (L.add_ld < L.main_cols || L.add_ld % 4 != 0 || !aligned16(L.add))) return PDR_EINVAL;
  }
  const pdr_chain_layer_t& last = p.layer[p.n_layers - 1];
  if (p.residual && p.layer[0].Cout - p.layer[0].main_cols != last.main_cols) return PDR_EINVAL;
  if (p.ldo < last.main_cols || p.ldo % 4 != 0 || !aligned16(p.out)) return PDR_EINVAL;
  if (!(n == 16 || n == 32 || n == 64 || n == 128 || n == 256)) return PDR_EUNSUPPORTED;
  const int NRT = n / 16, WR = NRT < 4 ? NRT : 4, WC = 4 / WR;
  for (int G = 8; G >= 1; G >>= 1) {
    if (B > 0 && static_cast<long>(B) * G > 256) continue;       // every workgroup resident: one per CU
    bool ok = true;
    for (int l = 0; l < p.n_layers && ok; ++l) {
      const pdr_chain_layer_t& L = p.layer[l];
      const int res_cols = (l == 0 && p.residual) ? L.Cout - L.main_cols : 0;
      if (L.main_cols % (16 * G) != 0 || res_cols % (16 * G) != 0) ok = false;
      const int wm = L.main_cols / G, wres = res_cols / G;
      if (ok && (wm + wres > MAXW || (wm / 16 + WC - 1) / WC > 4 || (wres / 16 + WC - 1) / WC > 4)) ok = false;
      if (ok && L.gamma) {
        const int cpg = L.Cn / L.groups;
        if (wm % cpg != 0) ok = false;                           // a column block = whole groups
      }
    }
    if (!ok) continue;
    out->G = G;
    out->ct = 1;
    for (int l = 0; l < p.n_layers; ++l) {
      const pdr_chain_layer_t& L = p.layer[l];
      const int res_cols = (l == 0 && p.residual) ? L.Cout - L.main_cols : 0;
      const int cm = (L.main_cols / G / 16 + WC - 1) / WC, cr = (res_cols / G / 16 + WC - 1) / WC;
      out->ct = cm > out->ct ? cm : out->ct;
      out->ct = cr > out->ct ? cr : out->ct;
    }
    for (int l = 0; l + 1 < p.n_layers; ++l)
      out->scratch_floats += static_cast<long>(B) * n * p.layer[l].main_cols;
    return PDR_OK;
  }
  return PDR_EUNSUPPORTED;
}

}  // namespace

#ifdef PDR_LAB_TRACE
extern "C" int pdr_lab_chain_read(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pdr_lab_chain), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int pdr_point_chain_plan(const pdr_point_chain_t* p, int B, int n, long* out) {
  if (!p || !out) return PDR_EINVAL;
  // (pointers are not dereferenced: a caller may plan before it allocates the scratch)
  pdr_point_chain_t q = *p;
  if (!q.scratch) q.scratch = reinterpret_cast<float*>(16);
  if (!q.sync) q.sync = reinterpret_cast<int*>(16);
  if (!q.out) {
    q.out = reinterpret_cast<float*>(16);
    if (q.n_layers >= 1 && q.n_layers <= 4) q.ldo = (q.layer[q.n_layers - 1].main_cols + 3) & ~3;
  }
  ChainPlan pl;
  const int rc = plan_chain(q, B, n, &pl);
  out[0] = pl.G;
  out[1] = pl.scratch_floats;
  out[2] = 2L * B + 1;                                             // ints of `sync`
  out[3] = rc == PDR_OK ? static_cast<long>(B) * pl.G : 0;         // workgroups of the launch
  return rc;
}

extern "C" int pdr_point_chain(const pdr_point_chain_t* p, int B, int n, pdr_stream_t stream) {
  if (!p) return PDR_EINVAL;
  ChainPlan pl;
  const int rc = plan_chain(*p, B, n, &pl);
  if (rc != PDR_OK) return rc;
  if (B == 0) return PDR_OK;
  const int place = (B % 8 == 0) ? 1 : 0;
  const dim3 grid(static_cast<unsigned>(B * pl.G));
  hipStream_t s = pdr::as_stream(stream);
  // instantiations: rows per cloud <= 64 (one row tile per wave) / <= 256 (four); 2 or 4 column tiles per wave
#define PDR_CHAIN(NM, R, C) hipLaunchKernelGGL((point_chain_kernel<NM, R, C>), grid, dim3(512), 0, s, *p, B, n, pl.G, place)
  if (n <= 64) {
    if (pl.ct <= 2) PDR_CHAIN(64, 1, 2);
    else PDR_CHAIN(64, 1, 4);
  } else {
    if (pl.ct <= 2) PDR_CHAIN(256, 4, 2);
    else PDR_CHAIN(256, 4, 4);
  }
#undef PDR_CHAIN
  return pdr::check_launch();
}
