// fused_layer_ws.hip -- wave-specialised form of the fused layer kernel (fused_layer.hip) for
// float4-staged sources (plain, neighbour-broadcast, or GATHERED first-conv tables), i.e. every
// steady-state layer of the reverse step.
//
// Why: in fused_layer_kernel every wave both stages the next K-chunk (address arithmetic, prologue
// math, LDS stores: ~450 issue slots per chunk) and runs the MFMAs of the current one, at 247 VGPRs
// = 2 waves per SIMD.  PMC on the 512x512 layers: MFMA pipe busy 64 %, waves parked 17 % and
// issue-stalled 59 % of their cycles.  Here the two jobs belong to different waves of a 512-thread
// workgroup:
//   waves 0-3  CONSUMERS  own the accumulators; per chunk: barrier, ds_read + v_mfma only.
//   waves 4-7  PRODUCERS  global -> registers -> prologue -> LDS for chunk g+1 while chunk g is being
//                         multiplied; the loads of chunk g+2 are issued before the barrier and fly
//                         through the next chunk period; they never touch an accumulator.
// LDS holds two stages; ONE barrier per chunk hands stage g to the consumers and stage g+1 back to
// the producers.  Workgroups are persistent over row tiles and the chunk sequence runs across tile
// boundaries, so the first chunk of the next tile is staged during the epilogue of the current one.
// Register budget 128 (consumers: 64 accumulators, producers: one chunk in flight) = 4 waves / SIMD.
// Epilogue: accumulators start at the bias, half tiles are transposed through LDS into dwordx4 row
// stores, the GroupNorm moments of the tile are folded across waves with an LDS ticket (no barrier
// that the producers would have to join).  The reasons behind each of these choices (VALU
// instructions wait behind MFMAs, FLAT loads break counted waits, ...) are in DESIGN.md section 4.1.
//
// Results equal fused_layer_kernel up to fp32 summation order (the bias is the accumulators' initial value here).
#include <cstdlib>

#include "pdr_common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // W quads: moved as values (float4 struct copies stay memcpy)

#ifdef PDR_LAB_TRACE
// development probe (tools/lab): per-chunk timestamps of one consumer and one producer wave of one workgroup
__device__ unsigned long long pdr_lab_trace[3][4096];
// start / end (s_memrealtime, 100 MHz, one clock for the whole device) and XCC id of every workgroup of the last launch
__device__ unsigned long long pdr_lab_wg[2048][3];
#define PDR_T(role, slot)                                                                   \
  do {                                                                                      \
    if (trace_on && (threadIdx.x & 63) == 0 && (slot) < 4096)                               \
      pdr_lab_trace[role][slot] = __builtin_amdgcn_s_memtime();                             \
  } while (0)
#else
#define PDR_T(role, slot) do {} while (0)
#endif

// identity prologue parameters for layers without scale / shift / add: read like the real per-channel
// arrays (same addressing), so the loads need no branch and no per-lane select
constexpr int kMaxCin = 4096;
// (not `const`: a constant-address-space object mixed with global pointers would turn the loads
// into FLAT loads, which also disable counted vmcnt waits)
__device__ float k_ones[kMaxCin + 8] = {[0 ... kMaxCin + 7] = 1.0f};
__device__ float k_zeros[kMaxCin + 8] = {[0 ... kMaxCin + 7] = 0.0f};

// IEEE max without the canonicalising pre-pass fmaxf() compiles to (identical for non-NaN inputs)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// LDS image of the two pipeline stages.
//   fp32 (exact):  A k-major [k][row] (+1 pad), W k-major [k][col]: one ds_read_b32 per MFMA operand element.
//   SPLIT (f16x3): every fp32 value x is held as hi = f16(x), lo = f16(x - hi); A and W are ROW-major
//   [row][32 k] / [col][32 k] halves = 64-byte rows so that a lane's MFMA operand (8 consecutive k of one row) is ONE
//   ds_read_b128.  Rows are unpadded; the 16-byte granule g of row r sits at position g ^ ((r >> 2) & 3): the 16
//   lanes of a ds_read_b128 service group (rows distinct mod 16, same g) then hit 16 distinct 16-byte slots of the
//   256-byte bank row, and a producer's 16-lane ds_write_b64 group covers two whole rows = 32 distinct banks.
template <bool SPLIT, int KC, int TM, int TN>
struct StageMem;
template <int KC, int TM, int TN>
struct StageMem<false, KC, TM, TN> {
  // A: channel PAIRS interleaved per row, [k / 2][row][k & 1]: a producer's four consecutive channels of a row
  // are two ds_write_b64 (k-major fp32 needed four ds_write_b32), a consumer's operand pair one ds_read_b64.
  // Row padding: 1 (KC = 32: 8 channel quads per 16-lane store group) or 2 (KC = 16: 4 quads) rows make the
  // 16 lanes of a ds_write_b64 group cover 32 distinct banks; ds_read_b64 of 32 consecutive rows is contiguous.
  static constexpr int APAD = KC == 32 ? 1 : 2;
  __attribute__((aligned(8))) float As[2][KC / 2][TM + APAD][2];
  __attribute__((aligned(16))) float Bs[2][KC][TN];
};
template <int KC, int TM, int TN>
struct StageMem<true, KC, TM, TN> {
  static_assert(KC == 32, "split layout: 32 halves = 64-byte rows");
  __attribute__((aligned(16))) unsigned char A[2][2][TM * 64];   // [stage][hi / lo][row][64 B]
  __attribute__((aligned(16))) unsigned char B[2][2][TN * 64];   // [stage][hi / lo][col][64 B]
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// byte offset of k-granule g (8 halves = 16 B) of row r inside a [rows][64 B] image
__device__ __forceinline__ int split_off(int r, int g) { return r * 64 + ((g ^ ((r >> 2) & 3)) << 4); }

// GATH: 0 = plain sources, 1 = gathered ball-query first conv (U[idx] + V, empty balls), 2 = gathered kNN first conv
// (U[idx] + V + d2 r1 + w r2: the two per-position terms of group_knn's distance / weight channels)
// POOL: attention pooling epilogue (pdr::PoolArgs) -- the scores stay in the accumulators
// Workgroups per CU: two (4 waves per SIMD, 128 registers) for every tile shape but the 128 x 32 narrow tiles without a
// residual source: those are bound by the latency of their single chunk in flight, hold 16 accumulators and fit 80
// registers and 51 KB of LDS, so THREE workgroups per CU (6 waves per SIMD) keep half as many chunks again in flight:
// 8.69 / 8.68 / 8.67 -> 8.54 / 8.56 / 8.59 ms per step (split step 7.25 -> 7.07).  The residual forms need 92-200
// bytes of scratch per lane under the 80-register cap and lose (8.6 -> 9.1 ms): they stay at two.
template <int RT, int CT, bool RADD, bool SPLIT>
constexpr int ws_waves_per_simd() { return (RT * CT == 1 && !RADD && !SPLIT) ? 6 : 4; }

// PAIR (round 6): the launch carries a SECOND problem (pdr::WsTwin: the same layer over the per-query rows of a
// deduplicated block; plain sources) for its workgroups tw.gx .. gridDim.x - 1 -- one launch instead of two in a chain
// of dependent launches.  A workgroup belongs to one problem for its whole life; the choice is made once, here.  The
// gathered instantiation runs the plain second problem through its runtime plain-segment path (f_g false) and the
// weighted per-row statistics (WSTAT) are compiled in.  PAIR = false: the kernel as it was (tw unused).
template <int RT, int CT, int WR, int WC, int KC, bool RADD, int GATH = 0, bool SPLIT = false, bool POOL = false,
          bool PAIR = false>
__global__ __launch_bounds__(512, (ws_waves_per_simd<RT, CT, RADD, SPLIT>())) void fused_layer_ws_kernel(
    pdr_layer_in_t in_a, int Cin, const float* __restrict__ Wt, int ldw,
    const float* __restrict__ bias, int Cout, float* __restrict__ Y_a, int ldy_a,
    float* __restrict__ partial_a, int relu_col0, int n_row_tiles_a, int tile_order, pdr::PoolArgs pool,
    pdr::WsTwin tw) {
  static_assert(!PAIR || (!RADD && !SPLIT && !POOL && GATH != 2), "paired launches: plain / ball-gathered, no residual");
  // (readfirstlane: provably uniform for the compiler too -- the selected values feed scalar operands)
  const int sel = PAIR ? __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) >= tw.gx ? 1 : 0) : 0;
  const bool second = PAIR && sel != 0;
  const pdr_layer_in_t& in = PAIR ? tw.in[sel] : in_a;
  float* const __restrict__ Y = PAIR ? tw.Y[sel] : Y_a;
  float* const __restrict__ partial = PAIR ? tw.partial[sel] : partial_a;
  const int ldy = PAIR ? tw.ldy[sel] : ldy_a;
  const int n_row_tiles = PAIR ? tw.n_row_tiles[sel] : n_row_tiles_a;
  // workgroup number / workgroups of this problem
  const int vbx = (PAIR && second) ? static_cast<int>(blockIdx.x) - tw.gx : static_cast<int>(blockIdx.x);
  const int vnwg = PAIR ? (second ? static_cast<int>(gridDim.x) - tw.gx : tw.gx) : static_cast<int>(gridDim.x);
  // SPLIT: `Wt` points at the packed f16 hi / lo weight image (pack_f16x3 of fused_network.py: per column block and
  // K-chunk one 16-KiB [hi | lo] x [128 cols][64 B] block in exactly the LDS layout), ldw = chunks per column block
  static_assert(WR * WC == 4, "4 consumer waves");
  constexpr int TM = WR * RT * 32, TN = WC * CT * 32;
  static_assert(!SPLIT || TN == 128 || TN == 64, "split image: 128- or 64-column blocks");
  constexpr int C4 = KC / 4;          // float4 columns of an A chunk
  constexpr int PT = 256;             // producer threads: all four producer waves stage every chunk
  constexpr int VSTEP = PT / C4;      // rows covered per step
  constexpr int APT4 = TM / VSTEP;    // float4 of A per producer thread
  constexpr int TN4 = TN / 4;
  constexpr int WPT4 = (KC * TN4 + PT - 1) / PT;
  static_assert(TM % VSTEP == 0, "tile rows");
  __shared__ StageMem<SPLIT, KC, TM, TN> sm;
  __shared__ float red[WR][TN][2];
  __shared__ int epi_ticket;          // consumer waves that finished the statistics of a tile (4 per tile)
  // per consumer wave: half of a 32 x 32 accumulator tile, row-major, for 16-byte coalesced stores
  __shared__ __attribute__((aligned(16))) float Tt[4][16][36];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rpb = in.rows_per_batch;
  const int tpb = (rpb + TM - 1) / TM;
  const int n0 = blockIdx.y * TN;
  // chunks per tile, tiles of this workgroup
  int nch = 0;
  for (int s = 0; s < in.n_seg; ++s) nch += (in.seg[s].C + KC - 1) / KC;
  // Tile order.  A workgroup walks LOCAL tile numbers l = first + k step (k = 0 .. my_tiles - 1); local -> row tile:
  //   plain      first = blockIdx.x, step = gridDim.x, row tile = l: at any moment the resident workgroups work on
  //              consecutive tiles of ONE cloud (and every XCD's L2 fetches that cloud's gathered table);
  //   XCD-local  workgroup ids are dealt to the 8 XCDs round-robin (id % 8 == XCC id, tools/lab/xcc_probe.hip), so
  //              XCD x owns the clouds x, x + 8, ...: first = blockIdx.x / 8, step = gridDim.x / 8, and local tile l
  //              is tile l % tpb of cloud x + 8 (l / tpb) -- a cloud's gathered table is fetched into ONE L2.
  // Same cost per chunk either way (one division by tpb); no change of which rows a tile holds.
  const int nwg = vnwg;
  const int nB = n_row_tiles / tpb;
  // (whole groups of 8 clouds only: otherwise some XCDs would own fewer clouds than others)
  //   listed     (in.tile_list, round 4): the launch computes only the row tiles tile_list[0 .. *n_tiles) -- the tiles of
  //              a grouped block whose 32-row neighbourhoods are not all copies of their first row;
  //              local tile l = list entry l, plain strided walk over the list; every other tile is neither read nor
  //              written (its rows of Y / partial keep whatever they held).
  const bool listed = in.tile_list != nullptr;                                            // uniform
  const bool xcd_order = !listed && tile_order != 0 && (nwg & 7) == 0 && nB >= 8 && (nB & 7) == 0;   // uniform
  const int xcd = vbx & 7;
  const int tile_first = xcd_order ? vbx >> 3 : vbx;
  const int tile_step = xcd_order ? nwg >> 3 : nwg;
  const int tile_limit = listed ? min(*in.n_tiles, n_row_tiles)
                                : (xcd_order ? ((nB - xcd + 7) >> 3) * tpb : n_row_tiles);   // local tiles of this walk
  // local tile number -> row tile (listed: one scalar load; the list is a few KB, read by every workgroup)
  //   reversed   (in.walk_reverse, round 6): the same local tiles from the last to the first.  A layer's output is larger
  //              than the 256-MB memory-side cache at the 524,288-row level; the layer that reads it in the order it
  //              was written finds none of it there, the one that starts where the writer ended finds the writer's tail.
  const bool rev = in.walk_reverse != 0 && !listed;                                       // uniform
  auto row_tile = [&](int l) __attribute__((always_inline)) -> int {
    return listed ? in.tile_list[l] : (rev ? tile_limit - 1 - l : l);
  };
  const int ptpb = in.partial_tpb > 0 ? in.partial_tpb : tpb;   // rows of `partial` per batch element
  const int cloud_mul = xcd_order ? 8 : 1, cloud_add = xcd_order ? xcd : 0;
  const int my_tiles = tile_limit > tile_first ? (tile_limit - tile_first + tile_step - 1) / tile_step : 0;
  // (XCD-local order with more workgroups per XCD than local tiles: nothing to do -- and the producers' first fetch
  // below must not run on a tile that does not exist)
  if constexpr (POOL) {
    // ---- pooled rows of the queries in SKIPPED tiles (in.patch_values; round 5: was the pdr_patch_rows launch behind
    // this one): a one-point neighbourhood's pooled row is its activated value row.  Inputs of this launch only (the
    // per-query value rows, the folded value GroupNorm), rows that no tile of this launch writes: done first, by all
    // eight waves, grid-strided over the queries -- also by the workgroups that own no tile.
    if (in.patch_values) {
      constexpr int TN4p = TN / 4;
      constexpr int QP = 512 / TN4p;                       // queries per pass of this workgroup
      const int Kp = pool.K;
      const long nq = static_cast<long>(n_row_tiles) * TM / Kp;
      const int qpb = rpb / Kp;
      const int c = n0 + 4 * (tid % TN4p);
      const float lo = pool.v_relu ? 0.0f : -__builtin_inff();
      if (tid < QP * TN4p && c < Cout) {
        for (long q = static_cast<long>(blockIdx.x) * QP + tid / TN4p; q < nq; q += static_cast<long>(gridDim.x) * QP) {
          if (in.patch_w[q] > 0.0f) {
            const long bq = q / qpb;
            const f32x4 v = *reinterpret_cast<const f32x4*>(in.patch_values + q * in.patch_ld + c);
            f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f};
            if (pool.vscale) sc = *reinterpret_cast<const f32x4*>(pool.vscale + bq * Cout + c);
            if (pool.vshift) sh = *reinterpret_cast<const f32x4*>(pool.vshift + bq * Cout + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = vmax(__builtin_fmaf(v[j], sc[j], sh[j]), lo);
            const long orow = in.out_rows ? static_cast<long>(in.out_rows[q]) : q;
            *reinterpret_cast<f32x4*>(pool.out + orow * pool.ldo + c) = o;
          }
        }
      }
    }
  }
  if (my_tiles == 0) return;
#ifdef PDR_LAB_TRACE
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 2048) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pdr_lab_wg[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
    pdr_lab_wg[blockIdx.x][2] = (xcc & 15) | (static_cast<unsigned long long>(my_tiles) << 8);
  }
#endif
  const int G = my_tiles * nch;
  const bool has_partial = partial != nullptr;
  if (tid == 0) epi_ticket = 0;       // ordered before its first use by the barrier B(0)
#ifdef PDR_LAB_TRACE
  const bool trace_on = blockIdx.x == 37 % gridDim.x && blockIdx.y == 0 && (wave == 0 || wave == 4 || wave == 6);
#endif

  // cursor over (tile, segment, channel offset)
  struct Cur { int tile, sg, ks, cbase, ci; };   // ci: chunk number within the tile (0 .. nch-1)
  // branch-free (scalar selects): a branch between a fetch and the following commit makes the
  // compiler's wait-count pass fall back to vmcnt(0), which would drain the prefetch
  auto advance = [&](Cur& c, bool really = true) {
    const int segC = in.seg[c.sg].C;
    const int ks1 = c.ks + KC;
    const bool seg_end = ks1 >= segC;
    const bool tile_end = seg_end && (c.sg + 1 >= in.n_seg);
    Cur n;
    n.ks = seg_end ? 0 : ks1;
    n.cbase = tile_end ? 0 : (seg_end ? c.cbase + segC : c.cbase);
    n.sg = tile_end ? 0 : (seg_end ? c.sg + 1 : c.sg);
    n.tile = tile_end ? c.tile + tile_step : c.tile;
    n.ci = tile_end ? 0 : c.ci + 1;
    c.ci = really ? n.ci : c.ci;
    c.ks = really ? n.ks : c.ks;
    c.cbase = really ? n.cbase : c.cbase;
    c.sg = really ? n.sg : c.sg;
    c.tile = really ? n.tile : c.tile;
  };
  auto last_of_tile = [&](const Cur& c) { return c.sg == in.n_seg - 1 && c.ks + KC >= in.seg[c.sg].C; };

  if (wave >= 4) {
    // =================================== PRODUCERS ===================================
    // Short bursts of VALU / LDS work between long waits: issue them ahead of the consumers' MFMAs
    // (an MFMA occupies the matrix pipe for 64 cycles; a delayed staging instruction delays a barrier).
#ifndef PDR_LAB_NO_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    if constexpr (SPLIT) {
      // MODE.FP16_OVFL (hwreg 1, bit 23) = 1 in the producer waves: an f32 -> f16 conversion that overflows returns
      // +-65504 instead of +-inf.  Activations here have passed a GroupNorm (O(1)), but the gathered first-conv sums,
      // source tables and the head's input are un-normalised: with the default mode |x| > 65504 gave hi = inf,
      // lo = f16(x - inf) = -inf and NaN products.  With the clamp hi = 65504 and lo = f16(x - 65504)
      // carry |x| <= 131008 like any other value; beyond that both halves saturate (finite, documented in
      // include/pdr_hip.h) -- no instruction is added to the staging path.
      __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    }
    const float lo_pre = in.pre_relu ? 0.0f : -__builtin_inff();
    const float lo_post = in.post_relu ? 0.0f : -__builtin_inff();
    const int pt = tid - 256;
    const int vc4 = pt % C4, vr0 = pt / C4;
    // Rows of this thread.  Plain variants: vr0 + VSTEP i (a wave's load instruction covers 64 / C4 consecutive rows).
    // Gathered variants (ROWQ): APT4 CONSECUTIVE rows APT4 vr0 + i -- they belong to ONE query (APT4 divides K), so
    // the query row V, the ball count and the empty-ball decision are loaded / taken once per thread and tile, and
    // the APT4 neighbour indices are one vector load: 20 -> 11 vector-memory instructions per thread and tile (the
    // gathered narrow layers are bound by the CU's vector-memory issue, not by bytes or latency).
    constexpr bool ROWQ = GATH != 0;
    const int gsh_ = (GATH && in.gK > 0) ? __builtin_ctz(in.gK) : 0;   // (gK = 0: the plain problem of a pair)
    auto prow = [&](int i) __attribute__((always_inline)) -> int { return ROWQ ? APT4 * vr0 + i : vr0 + VSTEP * i; };
    const int ss_ld = in.ss_ld > 0 ? in.ss_ld : Cin;
    const bool has_pre = in.pre_relu != 0, has_add = in.add != nullptr;
    // per-thread byte offsets that do not change from chunk to chunk (full tiles, full chunks): the
    // per-chunk part of every address is a scalar base, so a fetch costs (almost) no VALU work
    unsigned a_off[APT4], r_off[RADD ? APT4 : 1], w_off[WPT4];
    int off_sg = -1;                                   // segment a_off was built for
    {
      const int nmax = ldw - n0 - 4;                   // last in-row float4 offset
#pragma unroll
      for (int i = 0; i < WPT4; ++i) {
        const int e = pt + PT * i;
        const int k = e / TN4, n4 = e - k * TN4;
        w_off[i] = static_cast<unsigned>(min(k, KC - 1) * ldw + min(4 * n4, nmax)) * 4u;
      }
      if constexpr (RADD) {
#pragma unroll
        for (int i = 0; i < APT4; ++i)
          r_off[i] = static_cast<unsigned>(prow(i) * in.rseg.ld + 4 * vc4) * 4u;
      }
    }
    // GATHERED sources (first conv of a grouped block consumed without materialising it, see
    // pdr_gather_add): x[p] = U[b, idx[p]] + V[p / K]; an empty ball reads the table's zero row + V0.
    // Per tile: this thread's neighbour indices (-1 = empty ball); per (tile, segment): byte offsets.
    // (the kNN form has no ball counts and, for the sake of its register budget -- two more per-position values and
    // two more row quads live in the producer -- no next-tile index prefetch)
    int g_idx[GATH ? APT4 : 1], n_idx[GATH == 1 ? APT4 : 1], n_cnt = 1;
    unsigned v_off = 0;
    // neighbour indices of this thread's APT4 consecutive rows of the tile starting at r0 (nv valid rows) + the
    // ball count of their query: one vector load + one dword load for whole tiles
    auto load_idx = [&](long r0, int nv, int (&id)[GATH ? APT4 : 1], int& cnt) __attribute__((always_inline)) {
      if constexpr (GATH != 0) {
        const int rf = min(APT4 * vr0, nv - 1);
        cnt = in.gcnt ? in.gcnt[(r0 + rf) >> gsh_] : 1;
        if (nv == TM) {                                  // uniform
          __builtin_memcpy(&id[0], in.gidx + r0 + APT4 * vr0, sizeof(int) * APT4);
        } else {
#pragma unroll
          for (int i = 0; i < APT4; ++i) id[i] = in.gidx[r0 + min(APT4 * vr0 + i, nv - 1)];
        }
      }
    };
    constexpr bool KNN = GATH == 2;
    float gs1v[KNN ? APT4 : 1], gs2v[KNN ? APT4 : 1];   // per tile: d2 / weight of this thread's positions
    f32x4 Rq1, Rq2;                                      // per chunk: the conv rows of those two channels
    int g_tile = -1, n_tile = -1;
    const int gsh = (GATH && in.gK > 0) ? __builtin_ctz(in.gK) : 0;
    bool Rgath = false;                                // chunk in flight comes from a gathered segment
    // one chunk in registers (plain arrays: as members of a struct one W quad ended up in scratch)
    // RG: the RESIDUAL is a gathered first-conv window (U_res[idx] + V_res: the residual conv of a block whose
    // first conv is virtual) -- the main sources are plain then
    constexpr bool RG = RADD && GATH != 0;
    float4 Rrv[APT4], Rrv2[1];
    f32x4 Rrrv[RADD ? APT4 : 1], Rrrv2[1], Rqr1, Rqr2;   // Rqr: kNN-form residual, rows of the d2 / weight channels   // (vector values: conditional float4 struct copies go through scratch)
    f32x4 Rrw[WPT4];
    float Rps[4], Rph[4], Rpa[4];
    int Rkmax = KC, Rcvalid = 4;   // valid k rows of the chunk; valid channels of this thread's float4
    // fetch: chunk at cursor c -> registers (address arithmetic + loads only)
    auto fetch = [&](const Cur& c) __attribute__((always_inline)) {
      const int lt = row_tile(c.tile);
      const int bl = lt / tpb, tb = lt - bl * tpb;
      const int b = bl * cloud_mul + cloud_add;
#ifdef PDR_LAB_SAME_ROWS
      const long row0 = static_cast<long>(tb & 1) * TM;     // lab: every tile reads the first rows (cache hits)
#else
      const long row0 = static_cast<long>(b) * rpb + static_cast<long>(tb) * TM;
#endif
      const int nvalid = min(TM, rpb - tb * TM);
      const pdr_seg_t seg = in.seg[c.sg];
      const int shift = __builtin_ctz(seg.row_div);
      Rkmax = min(KC, seg.C - c.ks);
      // prologue parameters of channels cbase + ks + 4 vc4 + j
      const float* sc_b = (in.scale ? in.scale + static_cast<long>(b) * ss_ld : k_ones) + c.cbase + c.ks;
      const float* sh_b = (in.shift ? in.shift + static_cast<long>(b) * ss_ld : k_zeros) + c.cbase + c.ks;
      const float* ad_b = (in.add ? in.add + static_cast<long>(b) * in.add_ld : k_zeros) + c.cbase + c.ks;
      const bool f_g = GATH && seg.gV != nullptr;               // uniform
      if constexpr (GATH) {
        if (c.tile != g_tile && (!PAIR || in.gidx)) {            // uniform: first chunk of a tile (of a gathered problem)
          off_sg = -1;
          if (GATH == 1 && c.tile == n_tile) {
            // indices prefetched while the previous tile's last chunk was fetched: the dependent
            // index -> row load chain is off the per-tile critical path
#pragma unroll
            for (int i = 0; i < APT4; ++i) g_idx[i] = n_cnt <= 0 ? -1 : n_idx[i];
          } else {
            int cnt = 1;
            load_idx(row0, nvalid, g_idx, cnt);
#pragma unroll
            for (int i = 0; i < APT4; ++i) g_idx[i] = cnt <= 0 ? -1 : g_idx[i];
          }
          if constexpr (KNN) {
            // consumed at the commit of this tile's first chunk, one chunk period from now: no prefetch needed
            if (nvalid == TM) {                          // uniform: one vector load per array
              __builtin_memcpy(&gs1v[0], in.gs1 + row0 + APT4 * vr0, sizeof(float) * APT4);
              __builtin_memcpy(&gs2v[0], in.gs2 + row0 + APT4 * vr0, sizeof(float) * APT4);
            } else {
#pragma unroll
              for (int i = 0; i < APT4; ++i) {
                const long p = row0 + min(prow(i), nvalid - 1);
                gs1v[i] = in.gs1[p];
                gs2v[i] = in.gs2[p];
              }
            }
          }
          g_tile = c.tile;
        }
        const int nt = c.tile + tile_step;
        if (GATH == 1 && last_of_tile(c) && nt < tile_limit && (!PAIR || in.gidx)) {  // uniform: prefetch the next tile's indices
          const int nlt = row_tile(nt);
          const int nbl = nlt / tpb, ntb = nlt - nbl * tpb;
          const int nb = nbl * cloud_mul + cloud_add;
          const long nrow0 = static_cast<long>(nb) * rpb + static_cast<long>(ntb) * TM;
          const int nnv = min(TM, rpb - ntb * TM);
          if constexpr (GATH == 1) load_idx(nrow0, nnv, n_idx, n_cnt);
          n_tile = nt;
        }
      }
      // every load = uniform base (scalar registers) + 32-bit per-thread byte offset
      const char* ab = reinterpret_cast<const char*>(
          seg.ptr + (f_g ? static_cast<long>(b) * seg.g_nsrc : (row0 >> shift)) * seg.ld + c.ks);
      const int zrow = f_g ? seg.g_zrow - b * seg.g_nsrc : 0;    // the zero row, relative to this cloud
      const int v0d = (f_g && seg.gV0) ? static_cast<int>(seg.gV0 - seg.gV) : 0;
      // SPLIT: chunk image number (column block, chunk) of the packed weights; 16 KiB each, copied linearly
      const char* wb = SPLIT ? reinterpret_cast<const char*>(Wt) +
                                   (static_cast<long>(blockIdx.y) * ldw + c.ci) * (2L * TN * 64)
                             : reinterpret_cast<const char*>(Wt + static_cast<long>(c.cbase + c.ks) * ldw + n0);
      unsigned ao[APT4], ro[RADD ? APT4 : 1], wo[WPT4], po[4], vo = 0;
      if (Rkmax == KC && nvalid == TM) {
        // ---- fast path (uniform): the precomputed per-thread offsets
        if (off_sg != c.sg) {
          off_sg = c.sg;
#pragma unroll
          for (int i = 0; i < APT4; ++i) {
            int row = prow(i) >> shift;
            if constexpr (GATH) {
              if (f_g) {
                row = g_idx[i] < 0 ? zrow : g_idx[i];
                if (i == 0)
                  v_off = static_cast<unsigned>((prow(0) >> gsh) * seg.g_ldv + 4 * vc4 + (g_idx[0] < 0 ? v0d : 0)) * 4u;
              }
            }
            a_off[i] = static_cast<unsigned>(row * seg.ld + 4 * vc4) * 4u;
          }
        }
        if constexpr (GATH) vo = v_off;
        Rcvalid = 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) po[j] = 16u * vc4 + 4u * j;
#pragma unroll
        for (int i = 0; i < APT4; ++i) ao[i] = a_off[i];
        if constexpr (RADD) {
#pragma unroll
          for (int i = 0; i < APT4; ++i) ro[i] = r_off[i];
        }
#pragma unroll
        for (int i = 0; i < WPT4; ++i) wo[i] = SPLIT ? static_cast<unsigned>(pt + PT * i) * 16u : w_off[i];
      } else {
        // ---- general path: partial chunk (last of a segment) or partial row tile
        const int cl = 4 * vc4;                                   // relative to ks
        const int clc = min(cl, ((seg.C + 3) & ~3) - 4 - c.ks);   // keep the 16-B load inside the row
        Rcvalid = cl == clc ? min(4, seg.C - c.ks - clc) : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) po[j] = static_cast<unsigned>(min(clc + j, seg.C - 1 - c.ks)) * 4u;
#pragma unroll
        for (int i = 0; i < APT4; ++i) {
          // rows beyond nvalid re-read the tile's last row; masked in the epilogue
          const int r = min(prow(i), nvalid - 1);
          int row = r >> shift;
          if constexpr (GATH) {
            if (f_g) {
              row = g_idx[i] < 0 ? zrow : g_idx[i];
              if (i == 0) vo = static_cast<unsigned>((r >> gsh) * seg.g_ldv + clc + (g_idx[0] < 0 ? v0d : 0)) * 4u;
            }
          }
          ao[i] = static_cast<unsigned>(row * seg.ld + clc) * 4u;
          if constexpr (RADD) ro[i] = static_cast<unsigned>(r * in.rseg.ld + clc) * 4u;
        }
        const int nmax = ldw - n0 - 4;   // last in-row float4 offset
#pragma unroll
        for (int i = 0; i < WPT4; ++i) {
          const int e = pt + PT * i;
          const int k = e / TN4, n4 = e - k * TN4;
          wo[i] = SPLIT ? static_cast<unsigned>(pt + PT * i) * 16u
                        : static_cast<unsigned>(min(k, Rkmax - 1) * ldw + min(4 * n4, nmax)) * 4u;
        }
      }
      if (Rkmax == KC) {
        // full chunk (uniform): this thread's four channels are consecutive and all valid -> the prologue
        // parameters are three 16-byte loads (4-byte aligned in general: dwordx4 tolerates that) instead of twelve
        // dword loads -- 20 -> 11 VMEM instructions per chunk and thread
        f32x4 s4, h4, a4;
        __builtin_memcpy(&s4, reinterpret_cast<const char*>(sc_b) + 16u * vc4, 16);
        __builtin_memcpy(&h4, reinterpret_cast<const char*>(sh_b) + 16u * vc4, 16);
        __builtin_memcpy(&a4, reinterpret_cast<const char*>(ad_b) + 16u * vc4, 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Rps[j] = s4[j];
          Rph[j] = h4[j];
          Rpa[j] = a4[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Rps[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sc_b) + po[j]);
          Rph[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sh_b) + po[j]);
          Rpa[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ad_b) + po[j]);
        }
      }
#pragma unroll
      for (int i = 0; i < APT4; ++i) Rrv[i] = *reinterpret_cast<const float4*>(ab + ao[i]);
      if constexpr (GATH) {
        Rgath = f_g;
        if (f_g) {
          // ONE query row per thread: its APT4 consecutive rows belong to the same query
          const char* vb = reinterpret_cast<const char*>(seg.gV + (row0 >> gsh) * seg.g_ldv + c.ks);
          Rrv2[0] = *reinterpret_cast<const float4*>(vb + vo);
          if constexpr (KNN) {
            // this thread's four channels of the d2 / weight rows (4-byte aligned in general; the column
            // offset stays inside the segment's 4-padded width, as for the A loads)
            const unsigned qo = Rkmax == KC ? 16u * vc4
                                            : static_cast<unsigned>(min(4 * vc4, ((seg.C + 3) & ~3) - 4 - c.ks)) * 4u;
            __builtin_memcpy(&Rq1, reinterpret_cast<const char*>(seg.g_r1 + c.ks) + qo, 16);
            __builtin_memcpy(&Rq2, reinterpret_cast<const char*>(seg.g_r2 + c.ks) + qo, 16);
          }
        }
      }
      if constexpr (RG) {
        // (this instantiation is launched only for a gathered residual: no runtime branch -- conditional assignments
        // of the chunk registers would pin them in scratch)
        const int rs_ld = in.rseg.ld, rs_ldv = in.rseg.g_ldv, rs_nsrc = in.rseg.g_nsrc;
        const char* rb = reinterpret_cast<const char*>(in.rseg.ptr + static_cast<long>(b) * rs_nsrc * rs_ld +
                                                       c.cbase + c.ks);
        const char* rvb = reinterpret_cast<const char*>(in.rseg.gV + (row0 >> gsh) * rs_ldv + c.cbase + c.ks);
        const int rz = in.rseg.g_zrow - b * rs_nsrc;            // the zero row, relative to this cloud
        const int rv0d = in.rseg.gV0 ? static_cast<int>(in.rseg.gV0 - in.rseg.gV) : 0;
        const bool fast = Rkmax == KC && nvalid == TM;
        const int colq = fast ? 4 * vc4 : min(4 * vc4, ((Cin + 3) & ~3) - 4 - c.cbase - c.ks);
#pragma unroll
        for (int i = 0; i < APT4; ++i) {
          const int urow = g_idx[i] < 0 ? rz : g_idx[i];
          Rrrv[i] = *reinterpret_cast<const f32x4*>(rb + static_cast<unsigned>(urow * rs_ld + colq) * 4u);
        }
        Rrrv2[0] = *reinterpret_cast<const f32x4*>(
            rvb + static_cast<unsigned>((min(prow(0), nvalid - 1) >> gsh) * rs_ldv + colq +
                                        (g_idx[0] < 0 ? rv0d : 0)) * 4u);
        if constexpr (GATH == 2) {
          __builtin_memcpy(&Rqr1, in.rseg.g_r1 + c.cbase + c.ks + colq, 16);
          __builtin_memcpy(&Rqr2, in.rseg.g_r2 + c.cbase + c.ks + colq, 16);
        }
      } else if constexpr (RADD) {
        const char* rb = reinterpret_cast<const char*>(in.rseg.ptr + row0 * in.rseg.ld + c.cbase + c.ks);
#pragma unroll
        for (int i = 0; i < APT4; ++i) Rrrv[i] = *reinterpret_cast<const f32x4*>(rb + ro[i]);
      }
#pragma unroll
      for (int i = 0; i < WPT4; ++i) Rrw[i] = *reinterpret_cast<const f32x4*>(wb + wo[i]);
    };
    // commit: prologue math + LDS stores into stage st.  The staging instructions compete with the
    // consumers' MFMAs for issue slots (measured ~20 cycles per instruction next to two MFMA-bound
    // waves), so the instruction count is what matters here:
    //  - v_max_f32 directly (fmaxf adds a canonicalising v_max per operand);
    //  - channel masks only in a segment's last, partial chunk (uniform branch);
    //  - W is stored unmasked: rows k >= kmax meet zeroed A columns (and hold finite, clamped-row
    //    data), columns >= Cout are never stored or counted.
    auto commit = [&](int st) __attribute__((always_inline)) {
      auto stage_a = [&](auto masked, auto pre, auto add) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked)::value, PRE = decltype(pre)::value, ADD = decltype(add)::value;
#pragma unroll
        for (int i = 0; i < APT4; ++i) {
          float x[4] = {Rrv[i].x, Rrv[i].y, Rrv[i].z, Rrv[i].w};
          if constexpr (GATH) {
            if (Rgath) {   // uniform: neighbour row + query row
              x[0] += Rrv2[0].x; x[1] += Rrv2[0].y; x[2] += Rrv2[0].z; x[3] += Rrv2[0].w;
              if constexpr (KNN) {   // + d2 r1 + w r2, in pdr_gather_add's order (same bits as its statistics saw)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  x[j] = __builtin_fmaf(gs1v[i], Rq1[j], x[j]);
                  x[j] = __builtin_fmaf(gs2v[i], Rq2[j], x[j]);
                }
              }
            }
          }
          float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if constexpr (RADD) {
            q[0] = Rrrv[i][0]; q[1] = Rrrv[i][1]; q[2] = Rrrv[i][2]; q[3] = Rrrv[i][3];
            if constexpr (RG) {   // neighbour row + query row, as pdr_gather_add would have written them
              q[0] += Rrrv2[0][0]; q[1] += Rrrv2[0][1]; q[2] += Rrrv2[0][2]; q[3] += Rrrv2[0][3];
              if constexpr (GATH == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  q[j] = __builtin_fmaf(gs1v[i], Rqr1[j], q[j]);
                  q[j] = __builtin_fmaf(gs2v[i], Rqr2[j], q[j]);
                }
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = x[j];
            if constexpr (PRE) v = vmax(v, lo_pre);
            v = __builtin_fmaf(v, Rps[j], Rph[j]);
            v = vmax(v, lo_post);
            if constexpr (ADD) v = v + Rpa[j];
            if constexpr (RADD) v = v + q[j];
            if constexpr (MASKED) v = j < Rcvalid ? v : 0.0f;
            x[j] = v;
          }
          if constexpr (!SPLIT) {
            *reinterpret_cast<f32x2*>(&sm.As[st][2 * vc4][prow(i)][0]) = f32x2{x[0], x[1]};
            *reinterpret_cast<f32x2*>(&sm.As[st][2 * vc4 + 1][prow(i)][0]) = f32x2{x[2], x[3]};
          }
          if constexpr (SPLIT) {
            // x = hi + lo + O(max(2^-23 |x|, 2^-25)): hi = f16(x) (round to nearest even), lo = f16(x - hi) -- lo is
            // a SUBNORMAL half for |x| < 0.25, which v_mfma_*_f16 honours (measured: tools/lab/split_half.py); the four
            // channels of this thread are 4 consecutive k of row r: 8 bytes of the hi image, 8 of the lo image
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 v01 = {x[0], x[1]}, v23 = {x[2], x[3]};
            const half2v h01 = __builtin_convertvector(v01, half2v), h23 = __builtin_convertvector(v23, half2v);
            const f2 r01 = v01 - __builtin_convertvector(h01, f2), r23 = v23 - __builtin_convertvector(h23, f2);
            const half2v l01 = __builtin_convertvector(r01, half2v), l23 = __builtin_convertvector(r23, half2v);
            const int r = prow(i);
            const int off = split_off(r, vc4 >> 1) + ((vc4 & 1) << 3);
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u2*>(&sm.A[st][0][off]) =
                u2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
            *reinterpret_cast<u2*>(&sm.A[st][1][off]) =
                u2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
          }
        }
      };
      using T = std::true_type;
      using F = std::false_type;
      if (Rkmax < KC) stage_a(T(), T(), T());          // partial chunk: general form
      else if (has_pre) stage_a(F(), T(), T());
      else if (has_add) stage_a(F(), F(), T());
      else stage_a(F(), F(), F());
      if constexpr (SPLIT) {
        // the packed image is already in LDS layout ([hi | lo][col][64 B], granules swizzled): linear copy
        unsigned char* wdst = &sm.B[st][0][0];
#pragma unroll
        for (int i = 0; i < WPT4; ++i) *reinterpret_cast<f32x4*>(wdst + (pt + PT * i) * 16) = Rrw[i];
      } else {
#pragma unroll
        for (int i = 0; i < WPT4; ++i) {
          const int e = pt + PT * i;
          const int k = e / TN4, n4 = e - k * TN4;
          if (KC * TN4 % PT == 0 || e < KC * TN4) *reinterpret_cast<f32x4*>(&sm.Bs[st][k][4 * n4]) = Rrw[i];
        }
      }
    };

    // Per chunk: wait for its loads (issued one chunk period earlier), stage it, issue the loads of
    // the next chunk, barrier.  The staging math is spread over all four producer waves because its
    // LATENCY (not its issue cost) is what can delay the barrier: next to two MFMA-bound waves a
    // VALU instruction waits ~a whole MFMA issue slot.
    Cur co{tile_first, 0, 0, 0, 0};
    fetch(co);
    for (int g = 0; g < G; ++g) {
      PDR_T(1, 4 * g + 0);
#ifdef PDR_LAB_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PDR_T(1, 4 * g + 3);
#endif
      commit(g & 1);
      PDR_T(1, 4 * g + 1);
      advance(co, g + 1 < G);
      fetch(co);                                    // past the end: re-reads the last chunk
      PDR_T(1, 4 * g + 2);
      __syncthreads();                                 // B(g): stage g full, stage g+1 free
    }
    return;
  }

  // =================================== CONSUMERS ===================================
  const int lane = tid & 63;
  const int wr = wave % WR, wc = wave / WR;
  const int il = lane & 31, hi = lane >> 5;
  float bias_r[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + (wc * CT + j) * 32 + il;
    bias_r[j] = (bias && col < Cout) ? bias[col] : 0.0f;
  }
  // accumulators start at the bias: the epilogue has no add to do (every VALU instruction of an
  // epilogue waits behind the other workgroup's MFMAs)
  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bias_r[j];

  Cur cur{tile_first, 0, 0, 0, 0};
  for (int g = 0; g < G; ++g) {
    PDR_T(0, 4 * g + 0);
    __syncthreads();   // B(g)
    PDR_T(0, 4 * g + 1);
    const int st = g & 1;
    if constexpr (SPLIT) {
      // f16x3: x . w = xh wh + xh wl + xl wh (+ xl wl ~ 2^-22 relative, dropped), fp32 accumulation, on
      // v_mfma_f32_32x32x16_f16: lane (il, hi) supplies 8 consecutive k (k = 16 ks + 8 hi ...) of its row / column
      const int ksteps16 = (min(KC, in.seg[cur.sg].C - cur.ks) + 15) >> 4;
      for (int ks = 0; ks < ksteps16; ++ks) {
        half8 ah[RT], al[RT], wh[CT], wl[CT];
        const int gq = 2 * ks + hi;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int off = split_off((wr * RT + i) * 32 + il, gq);
          ah[i] = *reinterpret_cast<const half8*>(&sm.A[st][0][off]);
          al[i] = *reinterpret_cast<const half8*>(&sm.A[st][1][off]);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          const int off = split_off((wc * CT + j) * 32 + il, gq);
          wh[j] = *reinterpret_cast<const half8*>(&sm.B[st][0][off]);
          wl[j] = *reinterpret_cast<const half8*>(&sm.B[st][1][off]);
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh[j], acc[i][j], 0, 0, 0);
          }
      }
    } else {
      // four channels per iteration = two MFMAs: lane half `hi` holds channels 4 kq + 2 hi (first MFMA) and
      // 4 kq + 2 hi + 1 (second), i.e. the pair it read with ONE ds_read_b64; channels >= kmax are zero in A
      const int kquads = (min(KC, in.seg[cur.sg].C - cur.ks) + 3) >> 2;
      for (int kq = 0; kq < kquads; ++kq) {
        f32x2 a[RT];
        float w0[CT], w1[CT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
          a[i] = *reinterpret_cast<const f32x2*>(&sm.As[st][2 * kq + hi][(wr * RT + i) * 32 + il][0]);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          w0[j] = sm.Bs[st][4 * kq + 2 * hi][(wc * CT + j) * 32 + il];
          w1[j] = sm.Bs[st][4 * kq + 2 * hi + 1][(wc * CT + j) * 32 + il];
        }
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, w0[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, w1[j], acc[i][j], 0, 0, 0);
          }
      }
    }
    PDR_T(0, 4 * g + 2);
    if (last_of_tile(cur)) {
      // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8 (reg>>2) + 4 (lane>>5)
      const int lt = row_tile(cur.tile);
      const int bl = lt / tpb, tb = lt - bl * tpb;
      const int b = bl * cloud_mul + cloud_add;
      const int tile = b * ptpb + tb;          // index of the tile's partial row
#ifdef PDR_LAB_SAME_OUT
      const long row0 = static_cast<long>(blockIdx.x & 511) * TM;   // lab: a workgroup rewrites one tile's rows
#else
      const long row0 = static_cast<long>(b) * rpb + static_cast<long>(tb) * TM;
#endif
      const int nvalid = min(TM, rpb - tb * TM);
      // (weighted statistics -- in.wrow0, the per-query launches of a deduplicated block -- take the per-row path below:
      // a few small launches per step; the full-tile path of every other launch stays as it is)
      constexpr bool WSTAT = GATH == 0 || PAIR;   // (the per-query rows are materialised: the gathered forms stay as they were)
      // (a row map of the per-query term with fewer than 32 positions per query -- sixteen dependent index loads per
      // lane in the full-tile path cost the 80-register instantiations 24 bytes of scratch -- takes the per-row path too)
      const int osh = in.oadd ? __builtin_ctz(in.oadd_div) : 0;
      const bool rows_full = nvalid == TM && !(WSTAT && (in.wrow0 || (in.oadd_rows && osh < 5)));   // uniform
      // opaque copies: keep the per-row store offsets from being hoisted out of the chunk loop
      // (64 live 64-bit addresses would spill)
      int il_e = il, hi_e = hi, lane_e = lane;
      asm volatile("" : "+v"(il_e), "+v"(hi_e), "+v"(lane_e));
      long ldy_e = ldy;                       // same for the uniform row offsets (scalar registers)
      asm volatile("" : "+s"(ldy_e));
      if constexpr (POOL) {
        // ---- attention pooling (whole row tiles only: the launcher guarantees rows_per_batch % TM == 0): a
        // 32-row MFMA block holds 32 / K whole queries (K in {8, 16, 32}); for a fixed column a query's K rows sit
        // in the 8-row register groups {r >> 2} of both lane halves.  Value rows are read in accumulator layout
        // (a wave instruction = two 128-byte row pieces) through a running scalar row pointer + one lane offset.
        const int K = pool.K, ksh = __builtin_ctz(K);
        const float lo = pool.v_relu ? 0.0f : -__builtin_inff();
        const long vrow_bytes = static_cast<long>(pool.ldv) * 4;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const long rb = row0 + (wr * RT + i) * 32;             // first row of this 32-row block (uniform)
          int cn[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            int c = K;
            if (pool.counts) {
              c = pool.counts[(rb + 8 * g4) >> ksh];
              c = c < 1 ? 1 : c;                                  // attention.py:85 clamp(min=1)
            }
            cn[g4] = c;
          }
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const int col = n0 + (wc * CT + j) * 32 + il_e;
            const bool colok = col < Cout;
            const int cc = colok ? col : 0;
            float v[16];
            {
              unsigned voff = static_cast<unsigned>(4 * hi_e * pool.ldv + cc) * 4u;
              asm volatile("" : "+v"(voff));
              const char* q = reinterpret_cast<const char*>(pool.values + rb * pool.ldv);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                if (r > 0) q += ((r & 3) == 0 ? 5 : 1) * vrow_bytes;
                v[r] = *reinterpret_cast<const float*>(q + voff);
              }
            }
            const float vs = pool.vscale ? pool.vscale[static_cast<long>(b) * Cout + cc] : 1.0f;
            const float vh = pool.vshift ? pool.vshift[static_cast<long>(b) * Cout + cc] : 0.0f;
            // (in place: the scaled scores overwrite the accumulators, the activated values their raw loads)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kk = ((r & 3) + 8 * (r >> 2) + 4 * hi_e) & (K - 1);
              // masked slots are exactly -1e9 as in the reference; base-2 exponentials (as pdr_attention_pool)
              acc[i][j][r] = (kk < cn[r >> 2] ? acc[i][j][r] : -1e9f) * 1.44269504088896340736f;
              v[r] = vmax(__builtin_fmaf(v[r], vs, vh), lo);
            }
            auto reduce = [&](auto gsz_c) __attribute__((always_inline)) {
              constexpr int GSZ = decltype(gsz_c)::value;        // 8-row register groups per query
#pragma unroll
              for (int g = 0; g < 4; g += GSZ) {
                float m = -__builtin_inff();
#pragma unroll
                for (int r = 4 * g; r < 4 * (g + GSZ); ++r) m = fmaxf(m, acc[i][j][r]);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float l = 0.0f, a = 0.0f;
#pragma unroll
                for (int r = 4 * g; r < 4 * (g + GSZ); ++r) {
                  const float w = __builtin_amdgcn_exp2f(acc[i][j][r] - m);
                  l += w;
                  a = __builtin_fmaf(v[r], w, a);
                }
                l += __shfl_xor(l, 32, 64);
                a += __shfl_xor(a, 32, 64);
                if (hi_e == 0 && colok) {
                  const long q = (rb + 8 * g) >> ksh;                         // this query (uniform)
                  pool.out[(in.out_rows ? static_cast<long>(in.out_rows[q]) : q) * pool.ldo + col] = a / l;
                }
              }
            };
            if (K == 8) reduce(std::integral_constant<int, 1>());
            else if (K == 16) reduce(std::integral_constant<int, 2>());
            else reduce(std::integral_constant<int, 4>());
            float bj = bias_r[j];
            asm volatile("" : "+v"(bj));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bj;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int cl = (wc * CT + j) * 32 + il_e;
        const int col = n0 + cl;
        const bool colok = col < Cout;
        const int c0 = n0 + (wc * CT + j) * 32;                  // first column of this 32-wide tile
        float s1 = 0.0f, s2 = 0.0f;
        if (rows_full) {
          // ---- full row tile (the common case): stores = scalar row base + one per-lane offset,
          // statistics without selects when the ReLU boundary does not cut this column tile
          if (in.oadd) {
            const float* ob = in.oadd + (colok ? col : 0);
            if (osh >= 5) {
              // K >= 32 neighbours per query: a 32-row MFMA tile belongs to ONE query row of `oadd`
#pragma unroll
              for (int i = 0; i < RT; ++i) {
                long oq = (row0 + (wr * RT + i) * 32) >> osh;                // this block's query (uniform)
                // (row map: not in the kNN-form instantiations -- their blocks are never sorted, and the residual form
                // has no register to spare)
                if constexpr (GATH != 2) oq = in.oadd_rows ? static_cast<long>(in.oadd_rows[oq]) : oq;
                const float v = ob[oq * in.oadd_ld];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += v;
              }
            } else {
#pragma unroll
              for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const int rl = (wr * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi_e;
                  acc[i][j][r] += ob[((row0 + rl) >> osh) * in.oadd_ld];
                }
            }
          }
          const unsigned lane_off = static_cast<unsigned>(4 * hi_e * ldy + col) * 4u;   // bytes
          const long row_bytes = ldy_e * 4;
          const bool none_relu = c0 + 32 <= relu_col0, all_relu = c0 >= relu_col0;   // uniform
          // MODE 0: no column of this tile takes the ReLU in its statistics, 1: all do, 2: the
          // boundary cuts the tile (per-lane select)
          // Stores: the MFMA layout has a lane per COLUMN, i.e. 64 dword stores per wave and tile, and
          // the epilogue is store-issue bound (~110 cycles per wave store).  With a 16-byte aligned
          // output each half tile goes through LDS (8 ds_write_b32 + 2 ds_read_b128 per lane) and
          // leaves as 2 dwordx4 stores covering 8 full 128-byte rows each: 4x fewer store instructions.
          const bool wide_store = (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0;   // uniform
          const int rr = lane_e >> 3, c4 = (lane_e & 7) * 4;
          // only the (4-padded) output columns: Y may be a column window of a wider row
          const bool col4ok = c0 + c4 < ((Cout + 3) & ~3);
          unsigned toff = static_cast<unsigned>(rr * ldy + c0 + c4) * 4u;
          asm volatile("" : "+v"(toff));   // not hoistable: 64-bit store addresses per (tile, half) would spill
          auto store_tile = [&](auto mode, auto wide) {
            constexpr int MODE = decltype(mode)::value;
            constexpr bool WIDE = decltype(wide)::value;
            const float lo = (MODE == 2 && col < relu_col0) ? -__builtin_inff() : 0.0f;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
              // running row pointer (scalar adds only): rows 0,1,2,3, 8,9,10,11, 16.. of the MFMA tile
              char* q = reinterpret_cast<char*>(Y + (row0 + (wr * RT + i) * 32) * ldy_e);
#pragma unroll
              for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                  const float y = acc[i][j][8 * h + r8];
                  if (WIDE) {
                    Tt[wave][8 * (r8 >> 2) + (r8 & 3) + 4 * hi_e][il_e] = y;
                  } else {
                    if (h + r8 > 0) q += ((r8 & 3) == 0 ? 5 : 1) * row_bytes;
                    if (colok) *reinterpret_cast<float*>(q + lane_off) = y;
                  }
                }
                // statistics: plain scalar adds / fmas (v_pk_* forms measured slower next to MFMAs)
                if (colok) {
#pragma unroll
                  for (int r8 = 0; r8 < 8; ++r8) {
                    const float y = acc[i][j][8 * h + r8];
                    const float f = MODE == 0 ? y : vmax(y, lo);
                    s1 += f;
                    s2 = __builtin_fmaf(f, f, s2);
                  }
                }
                if (WIDE) {
                  const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Tt[wave][rr][c4]);
                  const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Tt[wave][rr + 8][c4]);
                  if (col4ok) {
                    char* qh = q + (16 * h) * row_bytes;
#if defined(PDR_LAB_NT_STORE)
                    __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(qh + toff));
                    __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(qh + 8 * row_bytes + toff));
#elif !defined(PDR_LAB_NO_STORE)
                    *reinterpret_cast<f32x4*>(qh + toff) = v0;
                    *reinterpret_cast<f32x4*>(qh + 8 * row_bytes + toff) = v1;
#endif
                  }
                }
              }
            }
          };
          // (all lanes take part: a lane's 16-byte column group is independent of its own column)
          using M0 = std::integral_constant<int, 0>;
          using M1 = std::integral_constant<int, 1>;
          using M2 = std::integral_constant<int, 2>;
#ifdef PDR_LAB_DIRECT_STORE
          // lab: the wide tiles' rows leave from the accumulator layout (64 dword stores per wave and tile, two 128-byte
          // row pieces each) instead of through the LDS transpose
          if (RT * CT == 4) {
            if (none_relu) store_tile(M0(), std::false_type());
            else if (all_relu) store_tile(M1(), std::false_type());
            else store_tile(M2(), std::false_type());
          } else
#endif
          if (wide_store) {
            if (none_relu) store_tile(M0(), std::true_type());
            else if (all_relu) store_tile(M1(), std::true_type());
            else store_tile(M2(), std::true_type());
          } else {
            store_tile(M2(), std::false_type());   // unaligned output: scalar stores, general statistics
          }
        } else {
          // ---- partial row tile (last tile of a batch element) or weighted statistics: per-row predicates.
          // Weighted: only the rows r >= wrow0[b] of the batch element count, times wmul (pdr_layer_in_t.wrow0).
          int wlo = 0;                                             // uniform
          if constexpr (WSTAT) wlo = in.wrow0 ? in.wrow0[b] - tb * TM : 0;
          const bool relu_stat = col >= relu_col0;
          float* ybase = Y + row0 * ldy + col;
          // (rare path: one row at a time keeps its register footprint small)
          const float* ob = in.oadd ? in.oadd + (colok ? col : 0) : nullptr;
          if (colok) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int rl = (wr * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi_e;
                if (rl < nvalid) {
                  float y = acc[i][j][r];
                  if (ob) {
                    long oq = (row0 + rl) >> osh;
                    // (the row map: plain-source instantiations only -- in the gathered ones its index load cost the
                    // 80-register forms 24 bytes of scratch; fused_layer_ws_supported sends such a call elsewhere)
                    if constexpr (WSTAT) oq = in.oadd_rows ? static_cast<long>(in.oadd_rows[oq]) : oq;
                    y += ob[oq * in.oadd_ld];
                  }
                  ybase[rl * ldy] = y;
                  if (rl >= wlo) {
                    const float f = relu_stat ? fmaxf(y, 0.0f) : y;
                    s1 += f;
                    s2 = __builtin_fmaf(f, f, s2);
                  }
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          if constexpr (WSTAT) {
            if (in.wrow0) {                  // uniform: every counted row stands for wmul copies
              s1 *= in.wmul;
              s2 *= in.wmul;
            }
          }
        }
        PDR_T(2, 8 * (g / nch) + j);
        if (has_partial) {
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (hi_e == 0) {
            red[wr][cl][0] = s1;
            red[wr][cl][1] = s2;
          }
        }
        // (opaque: otherwise the 16-wide splat of the bias is kept live -- and spilled -- across
        // the whole chunk loop instead of being rebuilt here with 16 moves)
        float bj = bias_r[j];
        asm volatile("" : "+v"(bj));
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = bj;
      }
      }   // !POOL
      PDR_T(2, 8 * (g / nch) + 6);
      if (has_partial) {
        // Cross-wave fold of the statistics WITHOUT a workgroup barrier (the producers would have to
        // join it, and they are busy staging the next chunk): every consumer wave takes a ticket
        // after its LDS writes; the wave drawing the last ticket of this tile sums the WR rows (fixed
        // order w = 0..WR-1, deterministic) and writes the tile's partial row.
        int ticket = 0;
        if (lane == 0)
          ticket = __hip_atomic_fetch_add(&epi_ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if ((ticket & 3) == 3) {
          for (int c = lane; c < TN; c += 64) {
            if (n0 + c < Cout) {
              float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
              for (int w = 0; w < WR; ++w) {
                s1 += red[w][c][0];
                s2 += red[w][c][1];
              }
              float* o = partial + (static_cast<long>(tile) * Cout + n0 + c) * 2;
              o[0] = s1;
              o[1] = s2;
            }
          }
        }
      }
    }
    PDR_T(0, 4 * g + 3);
    advance(cur);
  }
#ifdef PDR_LAB_TRACE
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 2048) pdr_lab_wg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

#ifdef PDR_LAB_TRACE
extern "C" int pdr_lab_wg_read(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pdr_lab_wg), sizeof(unsigned long long) * 2048 * 3) == hipSuccess ? 0 : -1;
}
extern "C" int pdr_lab_trace_read(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pdr_lab_trace), sizeof(unsigned long long) * 3 * 4096) == hipSuccess ? 0 : -1;
}
#endif

namespace pdr {

// Whether tile variant `id` (pick_tile() of fused_layer.hip) has a wave-specialised instantiation for this input.
bool fused_layer_ws_supported(int id, bool radd, bool gath, const pdr_layer_in_t& in, int Cin) {
  if (Cin > kMaxCin) return false;   // identity scale / shift / add arrays cover kMaxCin channels
  if (in.wrow0 && gath) return false;   // weighted statistics: the plain-source instantiations (and the uniform kernel)
  // a row map of the per-query term: gathered instantiations read it per 32-row block only (one query per block)
  if (in.oadd_rows && gath && (in.oadd_div < 32 || in.rows_per_batch % 128 != 0)) return false;
  if (id == 3 || id == 6 || id > 8) return false;   // 128 x 160 (80 accumulators) and 32-row tiles: uniform-wave kernel
  bool knn = false;
  for (int sg = 0; sg < in.n_seg; ++sg) knn = knn || in.seg[sg].g_r1 != nullptr;
  const bool knn_res = in.rseg.gV && in.rseg.g_r1;
  if (in.oadd_rows && (knn || knn_res)) return false;   // (no row map in the kNN-form instantiations)
  if (knn) {
    // kNN-form gathered sources: both per-position arrays, both rows on every gathered segment, no empty balls
    if (!gath || radd || !in.gs1 || !in.gs2 || in.gcnt) return false;
    for (int sg = 0; sg < in.n_seg; ++sg)
      if (in.seg[sg].gV && (!in.seg[sg].g_r1 || !in.seg[sg].g_r2)) return false;
  }
  if (knn_res && (!in.gs1 || !in.gs2 || in.gcnt || !in.rseg.g_r2)) return false;   // kNN-form gathered residual
  if (gath) {
    // gathered sources here: either the main sources (plain residual or none) or the residual alone (ball form);
    // empty balls through the table's zero row and a V0 that sits a small non-negative offset behind V (one
    // allocation)
    if (in.gK < 4) return false;   // a producer thread's 2 or 4 consecutive rows share one query
    bool main_g = false;
    for (int sg = 0; sg < in.n_seg; ++sg) main_g = main_g || in.seg[sg].gV != nullptr;
    if (radd && main_g) return false;
    if (in.rseg.gV && (main_g || knn || !in.gidx)) return false;
    for (int sg = 0; sg <= in.n_seg; ++sg) {
      const pdr_seg_t& g = sg < in.n_seg ? in.seg[sg] : in.rseg;
      if (!g.gV) continue;
      if (in.gcnt) {
        if (g.g_zrow < 0 || !g.gV0) return false;
        const long d = g.gV0 - g.gV;
        if (d < 0 || d >= (1L << 28)) return false;
      }
    }
  }
  return true;
}

// Launches the wave-specialised kernel for tile variant `id`.  Returns false when the variant has no
// wave-specialised instantiation.  split: f16x3 arithmetic (Wt = packed weight image, ldw = chunks per column
// block); instantiated for the 128-column tile variants 4 and 5 and the 64-column variant 8.
bool launch_fused_layer_ws(int id, bool radd, bool gath, const pdr_layer_in_t& in, int Cin, const float* Wt,
                           int ldw, const float* bias, int Cout, float* Y, int ldy, float* partial,
                           int relu_col0, int n_row_tiles, int ncol, hipStream_t s, bool split, const PoolArgs* pool) {
  if (!fused_layer_ws_supported(id, radd, gath, in, Cin)) return false;
  if (split && id != 4 && id != 5 && id != 8) return false;
  if (pool && (radd || gath || in.oadd)) return false;   // pooled epilogue: plain sources
  const PoolArgs pa = pool ? *pool : PoolArgs();
  // persistent: at most 2 workgroups per CU, all co-resident
  long gx = n_row_tiles;
  // persistent: every workgroup co-resident -- 2 per CU, 3 for the narrow tiles without a residual (see above);
  // option ws_narrow3 = 0: 2 for all (A/B)
  const bool narrow3 = pdr::option(pdr::OPT_WS_NARROW3) != 0;
  // (launching only half of the co-resident workgroups, so that the kernels of the two block-half streams share every
  // CU instead of taking turns, measured 9.51 / 9.58 vs 8.78 / 8.79 ms per step in round 3 -- removed)
  const long resident = (narrow3 && id == 7 && !radd && !split) ? 768 : 512;
  // Option ws_xcd_order: 1 (default) = XCD-local cloud-major tile order for the gathered kernels, 2 = for every layer,
  // 0 = plain.  Measured (same box, B = 32): HBM traffic of the kNN-gathered wide tiles 213.7 -> 170.5 MB per launch
  // (143 MB algorithmic), of the kNN-gathered narrow tiles 178 -> 143 MB, ball-gathered kernels unchanged; step time
  // 8.75 / 8.75 / 8.76 (plain) vs 8.79 / 8.69 / 8.78 (gathered kernels) vs 8.80 / 8.80 / 8.79 (all), and again at the
  // end of round 3 8.78 / 8.82 / 8.75 vs 8.72 / 8.82 / 8.76 (split-f16: 7.36 / 7.39 / 7.35 vs 7.31 / 7.40 / 7.38); the
  // dominant kernel alone on the chip 149.3 / 149.3 us (plain) vs 152.3 / 149.8 us: a fifth fewer bytes at the same
  // time (the kernels are MFMA-bound), which is why the gathered kernels take it by default and the others do not.
  // Results are bit-identical (tools/lab/order_check.py, tests).
  const int xcd_knob = pdr::option(pdr::OPT_WS_XCD_ORDER);
  const int tile_order = (xcd_knob >= 2 || (xcd_knob == 1 && gath)) ? 1 : 0;
  long cap = (resident + ncol - 1) / ncol;
  // (whole groups of 8 workgroups for the XCD-local walk; fewer than 8 resident column-block workgroups -- ncol > 64
  // -- keep the plain walk instead of rounding the grid down to nothing)
  int tile_order_eff = tile_order;
  if (tile_order && cap >= 8) cap = cap / 8 * 8;
  else tile_order_eff = 0;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(ncol));
#define PDR_WS_K(RT, CT, WR, WC, KC, RA, GA, SP)                                                          \
  hipLaunchKernelGGL((fused_layer_ws_kernel<RT, CT, WR, WC, KC, RA, GA, SP>), grid, dim3(512), 0, s, in, \
                     Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, n_row_tiles, tile_order_eff, pa, pdr::WsTwin())
#define PDR_WS_POOL(RT, CT, WR, WC, KC)                                                                       \
  hipLaunchKernelGGL((fused_layer_ws_kernel<RT, CT, WR, WC, KC, false, 0, false, true>), grid, dim3(512), 0, s, \
                     in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, n_row_tiles, tile_order_eff, pa, pdr::WsTwin())
  bool knn = false;
  for (int sg = 0; sg < in.n_seg; ++sg) knn = knn || in.seg[sg].g_r1 != nullptr;
  const bool knn_res = in.rseg.gV && in.rseg.g_r1;
#define PDR_WS(RT, CT, WR, WC, KC)                            \
  do {                                                        \
    if (gath && knn) PDR_WS_K(RT, CT, WR, WC, KC, false, 2, false); \
    else if (gath && radd && knn_res) PDR_WS_K(RT, CT, WR, WC, KC, true, 2, false);  \
    else if (gath && radd) PDR_WS_K(RT, CT, WR, WC, KC, true, 1, false);  \
    else if (gath) PDR_WS_K(RT, CT, WR, WC, KC, false, 1, false);  \
    else if (radd) PDR_WS_K(RT, CT, WR, WC, KC, true, 0, false); \
    else PDR_WS_K(RT, CT, WR, WC, KC, false, 0, false);       \
  } while (0)
#define PDR_WS_SPLIT(RT, CT, WR, WC, KC)                      \
  do {                                                        \
    if (gath && knn) PDR_WS_K(RT, CT, WR, WC, KC, false, 2, true); \
    else if (gath && radd && knn_res) PDR_WS_K(RT, CT, WR, WC, KC, true, 2, true);   \
    else if (gath && radd) PDR_WS_K(RT, CT, WR, WC, KC, true, 1, true);   \
    else if (gath) PDR_WS_K(RT, CT, WR, WC, KC, false, 1, true);   \
    else if (radd) PDR_WS_K(RT, CT, WR, WC, KC, true, 0, true); \
    else PDR_WS_K(RT, CT, WR, WC, KC, false, 0, true);        \
  } while (0)
  if (pool && split) {
#define PDR_WS_POOL_SPLIT(RT, CT, WR, WC, KC)                                                                  \
  hipLaunchKernelGGL((fused_layer_ws_kernel<RT, CT, WR, WC, KC, false, 0, true, true>), grid, dim3(512), 0, s, \
                     in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, n_row_tiles, tile_order_eff, pa, pdr::WsTwin())
    if (id == 4) PDR_WS_POOL_SPLIT(2, 2, 2, 2, 32);
    else if (id == 5) PDR_WS_POOL_SPLIT(1, 2, 2, 2, 32);
    else PDR_WS_POOL_SPLIT(1, 2, 4, 1, 32);
#undef PDR_WS_POOL_SPLIT
    return true;
  }
  if (pool) {
    switch (id) {
      case 0: PDR_WS_POOL(2, 1, 4, 1, 16); return true;
      case 1: PDR_WS_POOL(2, 2, 4, 1, 16); return true;
      case 2: PDR_WS_POOL(1, 3, 4, 1, 32); return true;
      case 4: PDR_WS_POOL(2, 2, 2, 2, 32); return true;
      case 5: PDR_WS_POOL(1, 2, 2, 2, 32); return true;
      case 7: PDR_WS_POOL(1, 1, 4, 1, 32); return true;
      case 8: PDR_WS_POOL(1, 2, 4, 1, 32); return true;
      default: return false;
    }
  }
  if (split) {
    if (id == 4) PDR_WS_SPLIT(2, 2, 2, 2, 32);
    else if (id == 5) PDR_WS_SPLIT(1, 2, 2, 2, 32);
    else PDR_WS_SPLIT(1, 2, 4, 1, 32);
    return true;
  }
  switch (id) {
    case 0: PDR_WS(2, 1, 4, 1, 16); return true;
    case 1: PDR_WS(2, 2, 4, 1, 16); return true;
    case 2: PDR_WS(1, 3, 4, 1, 32); return true;
    // case 3 (128 x 160: 80 accumulators) does not fit the 128-register budget: uniform-wave kernel
    case 4: PDR_WS(2, 2, 2, 2, 32); return true;
    case 5: PDR_WS(1, 2, 2, 2, 32); return true;
    case 7: PDR_WS(1, 1, 4, 1, 32); return true;
    case 8: PDR_WS(1, 2, 4, 1, 32); return true;
    default: return false;
  }
#undef PDR_WS
#undef PDR_WS_SPLIT
#undef PDR_WS_K
#undef PDR_WS_POOL
}

// One launch for two problems of the same layer (see PAIR above).  Grid: the first problem's persistent workgroups (as
// launch_fused_layer_ws would size them) followed by the second's.
bool launch_fused_layer_ws_pair(int id, bool gath, const pdr_layer_in_t& in, int Cin, const float* Wt, int ldw,
                                const float* bias, int Cout, float* Y, int ldy, float* partial, int relu_col0,
                                int n_row_tiles, int ncol, WsTwin twin, hipStream_t s) {
  if (!(id == 2 || id == 4 || id == 7 || id == 8)) return false;                 // 128-row tiles
  if (!fused_layer_ws_supported(id, false, gath, in, Cin)) return false;
  const pdr_layer_in_t& in2 = twin.in[1];
  if (!fused_layer_ws_supported(id, false, false, in2, Cin)) return false;
  bool knn = false;
  for (int sg = 0; sg < in.n_seg; ++sg) knn = knn || in.seg[sg].g_r1 != nullptr;
  if (knn || in.rseg.ptr || in2.rseg.ptr || in2.tile_list || in2.gidx) return false;
  for (int sg = 0; sg < in2.n_seg; ++sg)
    if (in2.seg[sg].gV) return false;
  const long resident = 512;
  long cap = (resident + ncol - 1) / ncol;
  long gx1 = n_row_tiles < cap ? n_row_tiles : cap;
  long gx2 = twin.n_row_tiles[1] < cap ? twin.n_row_tiles[1] : cap;
  if (gx1 < 1) gx1 = 1;
  if (gx2 < 1) gx2 = 1;
  twin.gx = static_cast<int>(gx1);
  twin.in[0] = in;
  twin.Y[0] = Y;
  twin.partial[0] = partial;
  twin.ldy[0] = ldy;
  twin.n_row_tiles[0] = n_row_tiles;
  const dim3 grid(static_cast<unsigned>(gx1 + gx2), static_cast<unsigned>(ncol));
  const PoolArgs pa = PoolArgs();
#define PDR_WS_PAIR(RT, CT, WR, WC, KC)                                                                              \
  do {                                                                                                               \
    if (gath)                                                                                                        \
      hipLaunchKernelGGL((fused_layer_ws_kernel<RT, CT, WR, WC, KC, false, 1, false, false, true>), grid, dim3(512), \
                         0, s, in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, n_row_tiles, 0, pa, twin);  \
    else                                                                                                             \
      hipLaunchKernelGGL((fused_layer_ws_kernel<RT, CT, WR, WC, KC, false, 0, false, false, true>), grid, dim3(512), \
                         0, s, in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, n_row_tiles, 0, pa, twin);  \
  } while (0)
  switch (id) {
    case 2: PDR_WS_PAIR(1, 3, 4, 1, 32); return true;
    case 4: PDR_WS_PAIR(2, 2, 2, 2, 32); return true;
    case 7: PDR_WS_PAIR(1, 1, 4, 1, 32); return true;
    case 8: PDR_WS_PAIR(1, 2, 4, 1, 32); return true;
    default: return false;
  }
#undef PDR_WS_PAIR
}

}  // namespace pdr
