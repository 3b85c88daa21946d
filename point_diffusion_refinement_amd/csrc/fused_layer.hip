// fused_layer.hip -- one pass per shared-MLP layer of PDR's grouped point MLPs.
//
// The reference evaluates every `Conv2d 1x1 -> GroupNorm -> ReLU (+ t / condition
// embedding) (+ residual)` stage of Mlp_plus_t_emb / AttentionModule
// (pointnet2_modules.py:69-174, attention.py:35-96) as 4-8 separate full passes
// over a materialised (B, C, npoint, K) tensor (conv, GN moments, GN apply, ReLU,
// broadcast adds, torch.cat of the inputs).  On MI355X those passes are pure HBM
// traffic: the first profile of a reverse step spent 70 % of its time in
// cat / elementwise / moments kernels and 14 TFLOP/s in the convolutions.
//
// Here a layer is ONE kernel over channel-LAST activations X (P positions x Cin):
//
//   A-loader : x' = post( pre(x) * scale[b,c] + shift[b,c] ) + add[b,c] + R[p,c]
//              - x is the concatenation of up to 4 channel segments, each with its own
//                pointer / leading dimension / neighbour-broadcast divisor, so the
//                torch.cat([q.expand(K), k]) and cat([feat, skip, xyz]) tensors of the
//                reference are never built;
//              - scale/shift = the PRODUCER layer's GroupNorm folded to per-(batch,
//                channel) affine form, pre/post = ReLU placement (conv->GN->ReLU of the
//                MLPs vs ReLU->GN->conv of the attention score net), add = fc(t_emb) /
//                fc_condition / fc_second_condition rows, R = the residual branch.
//   GEMM     : Y = x' . Wt (+ bias) with v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise
//              an fmaf chain), 4 waves x (32 rows x up to 128 columns), A and W staged
//              k-major through LDS so every ds_read_b32 is conflict-free.
//   epilogue : per-(tile, channel) partial sums of f(y), f(y)^2 (f = identity or ReLU)
//              for the NEXT GroupNorm -- reduced deterministically by
//              pdr_gn_reduce/pdr_gn_finalize, no atomics.
//
// Roofline: at the dominant shapes (P = 2.1 M positions, C = 32-96) the layer moves
// 4 (Cin + Cout) bytes per position for 2 Cin Cout flops: 5-20 flop/B, i.e. HBM-bound
// (machine balance ~25 flop/B at the 157 TF fp32-MFMA peak).
#include "pdr_common.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// resolved source of ONE input channel (a thread's column is fixed within a K-chunk)
struct ColSrc {
  const float* ptr;  // segment base + channel offset
  int ld;
  int shift;         // log2(row_div): neighbour-broadcast divisors are powers of two here
};

__device__ __forceinline__ ColSrc resolve_col(const pdr_layer_in_t& in, int c) {
  ColSrc r;
  r.ptr = in.seg[0].ptr;
  r.ld = in.seg[0].ld;
  r.shift = 0;
  int c0 = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < in.n_seg) {
      const int cs = in.seg[s].C;
      if (c >= c0 && c < c0 + cs) {
        r.ptr = in.seg[s].ptr + (c - c0);
        r.ld = in.seg[s].ld;
        r.shift = __builtin_ctz(in.seg[s].row_div);
      }
      c0 += cs;
    }
  }
  return r;
}

__device__ __forceinline__ float load_col(const ColSrc& s, long row) {
  return s.ptr[(row >> s.shift) * s.ld];
}

// ---------------------------------------------------------------------------------------------
// Block tile TM x TN = (WR*RT*32) x (WC*CT*32); wave (wr, wc) owns RT x CT MFMA tiles of 32x32.
// grid.x strides over the row tiles (cut per batch element), grid.y = column blocks.
// K is walked segment by segment in chunks of KC channels, so within a chunk the source
// pointer / leading dimension / broadcast shift are wave-uniform (scalar registers) and every A
// load is `scalar base + 32-bit lane offset`.
// Pipeline per chunk: global -> registers (prologue applied) is issued BEFORE the MFMAs of the
// previous chunk, registers -> LDS after them: the loads overlap the matrix pipe.
using pdr::PoolArgs;   // POOL epilogue (pdr_common.h)

// KS (round 5): the four waves split the K walk KS ways -- wave (wr, wc, kq) multiplies the k-pairs kq, kq + KS, ... of
// every chunk into its own accumulators, the partial products are summed through LDS before the epilogue (in the
// fixed order kq = 0, 1, ...).  For the tiny layers of plan_layer: their time is the length of a wave's MFMA chain.
template <int RT, int CT, int WR, int WC, int KC, bool RADD, bool VEC, bool GATH, bool POOL = false, int KS = 1>
__global__ __launch_bounds__(256, 2) void fused_layer_kernel(
    pdr_layer_in_t in, int Cin, const float* __restrict__ Wt, int ldw,
    const float* __restrict__ bias, int Cout, float* __restrict__ Y, int ldy,
    float* __restrict__ partial, int relu_col0, int n_row_tiles, PoolArgs pool = PoolArgs()) {
  static_assert(WR * WC * KS == 4, "4 waves per workgroup");
  static_assert(KS == 1 || !POOL, "pooled epilogue: no K split");
  constexpr int TM = WR * RT * 32, TN = WC * CT * 32;
  // scalar A path: thread -> (column ac, rows ar0 + RSTEP i)
  constexpr int APT = TM * KC / 256;
  constexpr int RSTEP = 256 / KC;
  // vector A path (VEC: every segment 16-B aligned, ld % 4 == 0): thread -> (float4 column vc4,
  // rows vr0 + VSTEP i): 4x fewer load instructions and address computations
  constexpr int C4 = KC / 4;
  constexpr int VSTEP = 256 / C4;
  constexpr int APT4 = TM / VSTEP;
  // W chunk (always float4; Wt is packed with ldw % 4 == 0): element e4 = tid + 256 i
  constexpr int TN4 = TN / 4;
  constexpr int WPT4 = (KC * TN4 + 255) / 256;
  __shared__ float As[KC][TM + 1];
  __shared__ __attribute__((aligned(16))) float Bs[KC][TN + 4];
  __shared__ float red[WR][TN][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave % WR, wc = KS == 1 ? wave / WR : (wave / WR) % WC, kq = KS == 1 ? 0 : wave / (WR * WC);
  const int il = lane & 31, hi = lane >> 5;
  const int rpb = in.rows_per_batch;
  const int tpb = (rpb + TM - 1) / TM;
  const int n0 = blockIdx.y * TN;
  const float lo_pre = in.pre_relu ? 0.0f : -__builtin_inff();
  const float lo_post = in.post_relu ? 0.0f : -__builtin_inff();
  const bool single_chunk = in.n_seg == 1 && Cin <= KC;   // W staged once per workgroup
  bool w_loaded = false;

  float bias_r[CT];   // this lane's bias per column tile (the column block is fixed per workgroup)
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + (wc * CT + j) * 32 + il;
    bias_r[j] = (bias && col < Cout) ? bias[col] : 0.0f;
  }
  // registers of the chunk in flight
  float ra[VEC ? 1 : APT], rr[(RADD && !VEC) ? APT : 1];
  float4 rv[VEC ? APT4 : 1], rrv[(RADD && VEC) ? APT4 : 1];
  // gathered sources (GATH): second operand (query row) of the chunk / residual in flight, and this
  // thread's per-tile neighbour indices / empty-ball flags (its rows are the same for every chunk)
  float4 rv2[GATH ? APT4 : 1], rrv2[(GATH && RADD) ? APT4 : 1];
  int tidx[GATH ? APT4 : 1], tem[GATH ? APT4 : 1];
  bool f_g = false;            // chunk in flight comes from a gathered segment
  const int gshift = GATH ? __builtin_ctz(in.gK) : 0;
  float4 rwv[WPT4];
  float ps[VEC ? 4 : 1], ph[VEC ? 4 : 1], pa[VEC ? 4 : 1];   // per-channel prologue parameters
  float pok[(VEC && RADD) ? 4 : 1];                           // 1 / 0: channel inside the segment
  float f_s = 1.0f, f_h = 0.0f, f_a = 0.0f;
  bool f_cok = false;
  int f_kmax = 0;

  for (int tile = blockIdx.x; tile < n_row_tiles; tile += gridDim.x) {
    const int b = tile / tpb, tb = tile - b * tpb;
    const long row0 = static_cast<long>(b) * rpb + static_cast<long>(tb) * TM;
    const int nvalid = min(TM, rpb - tb * TM);
    const int ss_ld = in.ss_ld > 0 ? in.ss_ld : Cin;
    const float* sc_b = in.scale ? in.scale + static_cast<long>(b) * ss_ld : nullptr;
    const float* sh_b = in.shift ? in.shift + static_cast<long>(b) * ss_ld : nullptr;
    const float* ad_b = in.add ? in.add + static_cast<long>(b) * in.add_ld : nullptr;
    const float* rd_b = in.rseg.ptr ? in.rseg.ptr + row0 * in.rseg.ld : nullptr;   // plain residual
    if constexpr (GATH) {
#pragma unroll
      for (int i = 0; i < APT4; ++i) {
        const int r = min(tid / C4 + VSTEP * i, nvalid - 1);
        tidx[i] = in.gidx[row0 + r];
        tem[i] = (in.gcnt && in.gcnt[(row0 + r) >> gshift] <= 0) ? 1 : 0;
      }
    }

    // fetch(): ONLY address arithmetic + loads -- nothing consumes a loaded value, so no
    // s_waitcnt is emitted and the loads stay in flight across the MFMA loop that follows.
    // commit(): prologue math on the arrived values + LDS stores.
    // (sg, ks, cbase): segment, channel offset inside it, global channel index of its first channel
    auto fetch = [&](int sg, int ks, int cbase) {
      // `opaque`: re-materialise the per-thread indices on every call.  Without it LLVM hoists the
      // loop-invariant row / address values out of the chunk loop and keeps them alive across the
      // MFMA loop (>100 VGPRs, occupancy 1).
      int t = tid;
      asm volatile("" : "+v"(t));
      const pdr_seg_t seg = in.seg[sg];
      const int shift = __builtin_ctz(seg.row_div);
      const float* abase = seg.ptr + (row0 >> shift) * seg.ld;      // uniform; row0 % row_div == 0
      f_kmax = min(KC, seg.C - ks);                                  // valid channels in this chunk
      if constexpr (VEC) {
        const int vc4 = t % C4, vr0 = t / C4;
        const int cl = ks + 4 * vc4;                                 // first channel of this float4
        const int clc = min(cl, ((seg.C + 3) & ~3) - 4);             // keep the 16-B load in the row
        f_cok = cl == clc;                                           // else: all 4 lanes are padding
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = f_cok && (clc + j) < seg.C;
          const int cg = cbase + min(clc + j, seg.C - 1);
          // channels beyond the segment are forced to 0 (scale = shift = add = 0)
          ps[j] = ok ? (sc_b ? sc_b[cg] : 1.0f) : 0.0f;
          ph[j] = (ok && sh_b) ? sh_b[cg] : 0.0f;
          pa[j] = (ok && ad_b) ? ad_b[cg] : 0.0f;
          if constexpr (RADD) pok[j] = ok ? 1.0f : 0.0f;
        }
        f_g = GATH && seg.gV != nullptr;                             // uniform
        if (GATH && f_g) {
          const float* ub = seg.ptr + static_cast<long>(b) * seg.g_nsrc * seg.ld + clc;
#pragma unroll
          for (int i = 0; i < APT4; ++i) {
            const int r = min(vr0 + VSTEP * i, nvalid - 1);
            const long q = (row0 + r) >> gshift;
            rv[i] = *reinterpret_cast<const float4*>(ub + static_cast<long>(tidx[i]) * seg.ld);
            rv2[i] = *reinterpret_cast<const float4*>((tem[i] ? seg.gV0 : seg.gV) + q * seg.g_ldv + clc);
          }
        } else {
#pragma unroll
          for (int i = 0; i < APT4; ++i) {
            // rows beyond nvalid re-read the tile's last row; masked in the epilogue
            const int r = min(vr0 + VSTEP * i, nvalid - 1);
            rv[i] = *reinterpret_cast<const float4*>(abase + (r >> shift) * seg.ld + clc);
          }
        }
        if constexpr (RADD) {
          if (GATH && in.rseg.gV != nullptr) {
            const float* ub = in.rseg.ptr + static_cast<long>(b) * in.rseg.g_nsrc * in.rseg.ld + cbase + clc;
#pragma unroll
            for (int i = 0; i < APT4; ++i) {
              const int r = min(vr0 + VSTEP * i, nvalid - 1);
              const long q = (row0 + r) >> gshift;
              rrv[i] = *reinterpret_cast<const float4*>(ub + static_cast<long>(tidx[i]) * in.rseg.ld);
              rrv2[i] = *reinterpret_cast<const float4*>((tem[i] ? in.rseg.gV0 : in.rseg.gV) +
                                                         q * in.rseg.g_ldv + cbase + clc);
            }
          } else {
#pragma unroll
            for (int i = 0; i < APT4; ++i) {
              const int r = min(vr0 + VSTEP * i, nvalid - 1);
              rrv[i] = *reinterpret_cast<const float4*>(rd_b + r * in.rseg.ld + cbase + clc);
            }
          }
        }
      } else {
        const int ac = t % KC, ar0 = t / KC;
        const int cl = ks + ac;                                      // channel inside the segment
        f_cok = cl < seg.C;
        const int cc = f_cok ? cl : seg.C - 1;
        const int cg = cbase + cc;                                   // global input channel
        f_s = sc_b ? sc_b[cg] : 1.0f;
        f_h = sh_b ? sh_b[cg] : 0.0f;
        f_a = ad_b ? ad_b[cg] : 0.0f;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
          const int r = min(ar0 + RSTEP * i, nvalid - 1);
          ra[i] = abase[(r >> shift) * seg.ld + cc];
        }
        if constexpr (RADD) {
#pragma unroll
          for (int i = 0; i < APT; ++i) {
            const int r = min(ar0 + RSTEP * i, nvalid - 1);
            rr[i] = rd_b[r * in.rseg.ld + cg];
          }
        }
      }
      if (!(single_chunk && w_loaded)) {
        const float* wbase = Wt + static_cast<long>(cbase + ks) * ldw + n0;
        const int nmax = ldw - n0 - 4;                               // last in-row float4 offset
#pragma unroll
        for (int i = 0; i < WPT4; ++i) {
          const int e = t + 256 * i;
          const int k = e / TN4, n4 = e - k * TN4;
          rwv[i] = *reinterpret_cast<const float4*>(wbase + min(k, f_kmax - 1) * ldw + min(4 * n4, nmax));
        }
      }
    };
    auto commit = [&]() {
      int t = tid;
      asm volatile("" : "+v"(t));
      if constexpr (VEC) {
        const int vc4 = t % C4, vr0 = t / C4;
#pragma unroll
        for (int i = 0; i < APT4; ++i) {
          float x[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
          if constexpr (GATH) {
            if (f_g) {   // neighbour row + query row; empty ball: the query row alone (V0)
              const float v2[4] = {rv2[i].x, rv2[i].y, rv2[i].z, rv2[i].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) x[j] = tem[i] ? v2[j] : x[j] + v2[j];
            }
          }
          float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if constexpr (RADD) {
            q[0] = rrv[i].x; q[1] = rrv[i].y; q[2] = rrv[i].z; q[3] = rrv[i].w;
            if constexpr (GATH) {
              if (in.rseg.gV != nullptr) {
                const float v2[4] = {rrv2[i].x, rrv2[i].y, rrv2[i].z, rrv2[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = tem[i] ? v2[j] : q[j] + v2[j];
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = fmaxf(x[j], lo_pre);
            v = __builtin_fmaf(v, ps[j], ph[j]);
            v = fmaxf(v, lo_post) + pa[j];
            if constexpr (RADD) v = __builtin_fmaf(q[j], pok[j], v);
            As[4 * vc4 + j][vr0 + VSTEP * i] = v;
          }
        }
      } else {
        const int ac = t % KC, ar0 = t / KC;
        const float s = f_cok ? f_s : 0.0f, h = f_cok ? f_h : 0.0f, a = f_cok ? f_a : 0.0f;
        const float rok = f_cok ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
          float v = fmaxf(ra[i], lo_pre);
          v = __builtin_fmaf(v, s, h);
          v = fmaxf(v, lo_post) + a;
          if constexpr (RADD) v = __builtin_fmaf(rr[i], rok, v);
          As[ac][ar0 + RSTEP * i] = v;
        }
      }
      if (!(single_chunk && w_loaded)) {
#pragma unroll
        for (int i = 0; i < WPT4; ++i) {
          const int e = t + 256 * i;
          const int k = e / TN4, n4 = e - k * TN4;
          const bool kok = k < f_kmax;
          float4 w = rwv[i];
          w.x = (kok && n0 + 4 * n4 + 0 < Cout) ? w.x : 0.0f;
          w.y = (kok && n0 + 4 * n4 + 1 < Cout) ? w.y : 0.0f;
          w.z = (kok && n0 + 4 * n4 + 2 < Cout) ? w.z : 0.0f;
          w.w = (kok && n0 + 4 * n4 + 3 < Cout) ? w.w : 0.0f;
          if (e < KC * TN4) *reinterpret_cast<float4*>(&Bs[k][4 * n4]) = w;
        }
        w_loaded = true;
      }
    };

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int sg = 0, ks = 0, cbase = 0;
    fetch(0, 0, 0);
    while (true) {
      __syncthreads();                 // previous chunk's MFMAs (and epilogue reads of `red`) done
      commit();
      __syncthreads();
      const int segC = in.seg[sg].C;
      const int ksteps = (min(KC, segC - ks) + 1) >> 1;
      ks += KC;
      if (ks >= segC) {
        cbase += segC;
        ks = 0;
        ++sg;
      }
      const bool more = sg < in.n_seg;
      if (more) fetch(sg, ks, cbase);  // in flight during the MFMAs below
      for (int kk = (KS == 1 ? 0 : kq); kk < ksteps; kk += KS) {
        float a[RT], w[CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) a[i] = As[2 * kk + hi][(wr * RT + i) * 32 + il];
#pragma unroll
        for (int j = 0; j < CT; ++j) w[j] = Bs[2 * kk + hi][(wc * CT + j) * 32 + il];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], w[j], acc[i][j], 0, 0, 0);
      }
      if (!more) break;
    }

    if constexpr (KS > 1) {
      // sum the KS partial products of every MFMA tile: waves kq > 0 park their accumulators in the (now idle) W stage,
      // waves kq = 0 add them in the order kq = 1, 2, ...; only those run the epilogue below
      static_assert(sizeof(float) * (KS - 1) * WR * WC * RT * CT * 16 * 64 <= sizeof(Bs), "K-split scratch");
      float* scr = &Bs[0][0];
      __syncthreads();                 // every wave has read its last operands
      if (kq > 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              scr[((((kq - 1) * WR * WC + wc * WR + wr) * RT * CT + i * CT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (kq == 0) {
#pragma unroll
        for (int q = 1; q < KS; ++q)
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                acc[i][j][r] += scr[((((q - 1) * WR * WC + wc * WR + wr) * RT * CT + i * CT + j) * 16 + r) * 64 + lane];
      }
    }
    [[maybe_unused]] const bool epi_wave = kq == 0;      // uniform per wave
    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8 (reg>>2) + 4 (lane>>5).
    // Full row tiles (the common case) store through ONE predicated region per 32-column tile:
    // per-element conditions would put every store in its own basic block, each opened by an
    // s_waitcnt vmcnt(0) that drains the previous store (stores count in vmcnt on gfx950).
    int il_e = il;
    asm volatile("" : "+v"(il_e));
    // weighted statistics (in.wrow0, see pdr_layer_in_t): rows below wrow0[b] do not count, the rest x wmul
    const int wlo = in.wrow0 ? min(max(in.wrow0[b] - tb * TM, 0), TM) : 0;   // uniform
    const bool rows_full = nvalid == TM && wlo == 0;   // uniform
    if constexpr (POOL) {
      // a 32-row MFMA tile holds 32 / K whole queries (K in {8, 16, 32}); for a fixed column a
      // query's K rows sit in registers {r : (r >> 2) / (K / 8) == g} of both lane halves
      const int K = pool.K;
      const int gsz = K >> 3;                 // 8-row register groups per query
      const int ksh = __builtin_ctz(K);
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int col = n0 + (wc * CT + j) * 32 + il_e;
        const bool colok = col < Cout;
        const int cc = colok ? col : 0;
        const float bv = bias_r[j];
        const float vs = pool.vscale ? pool.vscale[static_cast<long>(b) * Cout + cc] : 1.0f;
        const float vh = pool.vshift ? pool.vshift[static_cast<long>(b) * Cout + cc] : 0.0f;
        const float lo = pool.v_relu ? 0.0f : -__builtin_inff();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int rbase = (wr * RT + i) * 32;
          if (rbase >= nvalid) continue;        // uniform (tiles end on query boundaries)
          float val[16], sc[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = rbase + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const long p = row0 + rl;
            int cnt = K;
            if (pool.counts) {
              cnt = pool.counts[p >> ksh];
              cnt = cnt < 1 ? 1 : cnt;
            }
            const int kk = rl & (K - 1);
            sc[r] = kk < cnt ? acc[i][j][r] + bv : -1e9f;
            val[r] = fmaxf(__builtin_fmaf(pool.values[p * pool.ldv + cc], vs, vh), lo);
          }
          for (int g = 0; g < 4; g += gsz) {      // uniform trip count
            float m = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if ((r >> 2) >= g && (r >> 2) < g + gsz) m = fmaxf(m, sc[r]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.0f, a = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if ((r >> 2) >= g && (r >> 2) < g + gsz) {
                const float w = expf(sc[r] - m);
                l += w;
                a = __builtin_fmaf(val[r], w, a);
              }
            l += __shfl_xor(l, 32, 64);
            a += __shfl_xor(a, 32, 64);
            if (hi == 0 && colok) {
              const long q = (row0 + rbase + 8 * g) >> ksh;
              pool.out[q * pool.ldo + col] = a / l;
            }
          }
        }
      }
      continue;   // next row tile
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      if constexpr (KS > 1) {
        if (!epi_wave) break;           // (K split: the other waves hold partial products only)
      }
      const int cl = (wc * CT + j) * 32 + il_e;
      const int col = n0 + cl;
      const bool colok = col < Cout;
      const float bv = bias_r[j];
      const bool relu_stat = col >= relu_col0;
      float s1 = 0.0f, s2 = 0.0f;
      float* ybase = Y + row0 * ldy + col;
      if (colok && in.oadd) {
        // per-query term of a split conv (one row of `oadd` per oadd_div positions)
        const int osh = __builtin_ctz(in.oadd_div);
        const float* ob = in.oadd + col;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = min((wr * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, nvalid - 1);
            const long oq = (row0 + rl) >> osh;
            acc[i][j][r] += ob[(in.oadd_rows ? static_cast<long>(in.oadd_rows[oq]) : oq) * in.oadd_ld];
          }
        }
      }
      if (colok) {
        if (rows_full) {
#pragma unroll
          for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = (wr * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
              const float y = acc[i][j][r] + bv;
              ybase[rl * ldy] = y;
              const float f = relu_stat ? fmaxf(y, 0.0f) : y;
              s1 += f;
              s2 = __builtin_fmaf(f, f, s2);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = (wr * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
              const float y = acc[i][j][r] + bv;
              if (rl < nvalid) {
                ybase[rl * ldy] = y;
                if (rl >= wlo) {
                  const float f = relu_stat ? fmaxf(y, 0.0f) : y;
                  s1 += f;
                  s2 = __builtin_fmaf(f, f, s2);
                }
              }
            }
          }
        }
      }
      if (partial) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (in.wrow0) {
          s1 *= in.wmul;
          s2 *= in.wmul;
        }
        if (hi == 0) {
          red[wr][cl][0] = s1;
          red[wr][cl][1] = s2;
        }
      }
    }
    if (partial) {
      __syncthreads();
      if (tid < TN && n0 + tid < Cout) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < WR; ++w) {
          s1 += red[w][tid][0];
          s2 += red[w][tid][1];
        }
        // (row b partial_tpb + tb of `partial`, as the wave-specialised kernels: partial_tpb = 0 -> tiles per batch)
        const long prow = static_cast<long>(b) * (in.partial_tpb > 0 ? in.partial_tpb : tpb) + tb;
        float* o = partial + (prow * Cout + n0 + tid) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
  }
}

// chan_stats[b, coff + c] (double2) = mult * sum over the tiles of batch b of partial[tile, c].
// 1024 threads = 32 channels x 32 tile-slices: the per-(b,c) sum over up to 512 tiles is split
// over 32 lanes' worth of independent loads and folded through LDS in a fixed order.
__global__ __launch_bounds__(1024) void gn_reduce_kernel(const float* __restrict__ partial, int ldp,
                                                         int tiles_per_batch, int C, double mult,
                                                         double* __restrict__ chan_stats, int Ctot,
                                                         int coff) {
  __shared__ double red[32][32][2];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const float* p = partial + (static_cast<long>(b) * tiles_per_batch * ldp + c) * 2;
    for (int t = sl; t < tiles_per_batch; t += 32) {
      const float2 v = *reinterpret_cast<const float2*>(p + static_cast<long>(t) * ldp * 2);
      s1 += v.x;
      s2 += v.y;
    }
  }
  red[sl][cl][0] = s1;
  red[sl][cl][1] = s2;
  __syncthreads();
  if (sl == 0 && c < C) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      a1 += red[k][cl][0];
      a2 += red[k][cl][1];
    }
    double* o = chan_stats + (static_cast<long>(b) * Ctot + coff + c) * 2;
    o[0] = a1 * mult;
    o[1] = a2 * mult;
  }
}

// gn_reduce + gn_finalize for up to two partial sources in ONE launch: block b reduces the tile
// partials of batch element b into LDS (double), then folds GroupNorm to scale / shift.
struct FoldPart {
  const float* partial;   // first of `C` columns inside rows of `ldp` columns
  int ldp, tiles_per_batch, C;
  double mult;
  // a tile SUBSET produced these rows (round 5): of the first tpb_main rows of batch element b only the first
  // nvalid[b] were written (sorted queries: a cloud's valid tiles are its first ones) -- the others are skipped, not
  // read as zeros (nobody zeroes them any more); rows >= tpb_main (the per-query rows' moments) always count
  const int* nvalid;
  int tpb_main;
};

__global__ __launch_bounds__(1024) void gn_fold_wide_kernel(FoldPart p0, FoldPart p1, int C, int Cn, int G,
                                                       double n, float eps,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       float* __restrict__ scale,
                                                       float* __restrict__ shift) {
  extern __shared__ __attribute__((aligned(16))) double cs[];   // [C][2]
  __shared__ double red[32][32][2];
  const int b = blockIdx.x;
  double* redf = &red[0][0][0];   // [1024][2]
  int coff = 0;
  for (int part = 0; part < 2; ++part) {
    const FoldPart p = part == 0 ? p0 : p1;
    if (!p.partial) continue;
    // W channels x (1024 / W) tile slices per pass: ONE pass (two barriers) for C <= 1024 instead of
    // one per 32 channels -- the kernel is barrier / latency bound, not bandwidth bound
    int W = 32;
    while (W < p.C && W < 1024) W <<= 1;
    const int nsl = 1024 / W;
    const int cl = threadIdx.x & (W - 1), sl = threadIdx.x / W;
    for (int c0 = 0; c0 < p.C; c0 += W) {
      const int c = c0 + cl;
      double s1 = 0.0, s2 = 0.0;
      if (c < p.C) {
        const float* q = p.partial + (static_cast<long>(b) * p.tiles_per_batch * p.ldp + c) * 2;
        const int nv = p.nvalid ? p.nvalid[b] : p.tpb_main;
        for (int t = sl; t < p.tiles_per_batch; t += nsl * 8) {
          float2 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int tt = t + nsl * u;
            v[u] = (tt < p.tiles_per_batch && !(tt >= nv && tt < p.tpb_main))
                       ? *reinterpret_cast<const float2*>(q + static_cast<long>(tt) * p.ldp * 2)
                       : make_float2(0.0f, 0.0f);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            s1 += v[u].x;
            s2 += v[u].y;
          }
        }
      }
      redf[threadIdx.x * 2 + 0] = s1;
      redf[threadIdx.x * 2 + 1] = s2;
      __syncthreads();
      if (sl == 0 && c < p.C) {
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < nsl; ++k) {
          a1 += redf[(k * W + cl) * 2 + 0];
          a2 += redf[(k * W + cl) * 2 + 1];
        }
        cs[(coff + c) * 2 + 0] = a1 * p.mult;
        cs[(coff + c) * 2 + 1] = a2 * p.mult;
      }
      __syncthreads();
    }
    coff += p.C;
  }
  for (int c = threadIdx.x; c < C; c += 1024) {
    float sc = 1.0f, sh = 0.0f;
    if (c < Cn) {
      const int cpg = Cn / G;
      const int g0 = (c / cpg) * cpg;
      double s1 = 0.0, s2 = 0.0;
      for (int j = 0; j < cpg; ++j) {
        s1 += cs[(g0 + j) * 2 + 0];
        s2 += cs[(g0 + j) * 2 + 1];
      }
      const double cnt = n * cpg;
      const double mean = s1 / cnt;
      double var = s2 / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
      sc = rstd * gamma[c];
      sh = __builtin_fmaf(-sc, static_cast<float>(mean), beta[c]);
    }
    scale[static_cast<long>(b) * C + c] = sc;
    shift[static_cast<long>(b) * C + c] = sh;
  }
}

// The fold that runs in the step (cpg = Cn / G <= 32): one 256-thread workgroup per (batch element, window of
// whole groups covering <= 32 channels).  A launch is B x ceil(Cn / window) small workgroups of four waves with
// 2.5 KB of LDS and < 40 VGPRs, so that it is admitted beside resident layer workgroups of the other stream
// instead of waiting for a CU to drain (the 1024-thread / 16 KB + 16 C form above could not co-reside with two
// persistent 512-thread layer workgroups; rocprofv3: 23.5 us per fold inside the two-stream step vs 7 us alone).
// Lanes 0-31 / 32-63 of a wave read the same 32 channels (256 contiguous bytes of a tile's partial row) of two
// different tile slices; 8 slices per workgroup, four loads in flight per thread; double sums, fixed order.
// U = partial rows in flight per thread and trip: 4 for up to 128 tiles per batch element (one or two trips), 16 above --
// the level-0 layers have 256 / 512 tiles per batch element, i.e. 32 / 64 rows per thread, and every trip is a dependent
// round trip to rows another kernel has just written (~1 us): 16 trips -> 4 (round 4; tools/lab/gn_fold_bench.py).
template <int U>
__global__ __launch_bounds__(256) void gn_fold_kernel(FoldPart p0, FoldPart p1, int C, int Cn, int G, int CW,
                                                      double n, float eps,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ scale,
                                                      float* __restrict__ shift) {
  __shared__ double red[4][32][2];
  __shared__ double cs[32][2];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CW;
  if (c0 >= Cn) {   // the trailing workgroup: channels outside the normalised range pass through
    for (int c = Cn + threadIdx.x; c < C; c += 256) {
      scale[static_cast<long>(b) * C + c] = 1.0f;
      shift[static_cast<long>(b) * C + c] = 0.0f;
    }
    return;
  }
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, wave = threadIdx.x >> 6;
  const int c = c0 + cl;
  const bool valid = cl < CW && c < Cn;
  double s1 = 0.0, s2 = 0.0, mult = 1.0;
  if (valid) {
    const bool first = c < p0.C;
    const FoldPart p = first ? p0 : p1;
    const int col = first ? c : c - p0.C;
    mult = p.mult;
    const long stride = static_cast<long>(p.ldp) * 2;
    const float* q = p.partial + (static_cast<long>(b) * p.tiles_per_batch * p.ldp + col) * 2;
    // (loaded beside the partial rows, consumed behind them: no extra dependent round trip)
    const int nv = p.nvalid ? p.nvalid[b] : p.tpb_main;
    const int tpb_main = p.tpb_main;
    for (int t = sl; t < p.tiles_per_batch; t += 8 * U) {
      // U loads in flight: unconditional, from a clamped tile (t itself is valid), zeroed afterwards -- a load
      // under `tt < tiles` compiles to a branch with its own vmcnt(0), i.e. U dependent round trips per trip
      float2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tt = t + 8 * u;
        v[u] = *reinterpret_cast<const float2*>(q + (tt < p.tiles_per_batch ? tt : t) * stride);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tt = t + 8 * u;
        const bool ok = tt < p.tiles_per_batch && !(tt >= nv && tt < tpb_main);   // (skipped tiles hold garbage)
        s1 += ok ? v[u].x : 0.0f;
        s2 += ok ? v[u].y : 0.0f;
      }
    }
  }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if ((threadIdx.x & 63) < 32) {
    red[wave][cl][0] = s1;
    red[wave][cl][1] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    cs[cl][0] = (red[0][cl][0] + red[1][cl][0] + red[2][cl][0] + red[3][cl][0]) * mult;
    cs[cl][1] = (red[0][cl][1] + red[1][cl][1] + red[2][cl][1] + red[3][cl][1]) * mult;
  }
  __syncthreads();
  if (threadIdx.x < 32 && valid) {
    const int cpg = Cn / G;
    const int g0 = (cl / cpg) * cpg;
    double g1 = 0.0, g2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
      g1 += cs[g0 + j][0];
      g2 += cs[g0 + j][1];
    }
    const double cnt = n * cpg;
    const double mean = g1 / cnt;
    double var = g2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    const float sc = rstd * gamma[c];
    scale[static_cast<long>(b) * C + c] = sc;
    shift[static_cast<long>(b) * C + c] = __builtin_fmaf(-sc, static_cast<float>(mean), beta[c]);
  }
}

// out (P, C; ld ldo) = prologue(X): materialise an activation (needed where the next consumer
// gathers whole feature rows, e.g. group_build / gather_rows)
__global__ __launch_bounds__(256) void apply_act_kernel(pdr_layer_in_t in, long P, int C,
                                                        float* __restrict__ out, int ldo) {
  const long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= P * C) return;
  const long row = e / C;
  const int c = static_cast<int>(e - row * C);
  const int b = static_cast<int>(row / in.rows_per_batch);
  float v = load_col(resolve_col(in, c), row);
  if (in.pre_relu) v = fmaxf(v, 0.0f);
  const int ss_ld = in.ss_ld > 0 ? in.ss_ld : C;
  const float s = in.scale ? in.scale[static_cast<long>(b) * ss_ld + c] : 1.0f;
  const float h = in.shift ? in.shift[static_cast<long>(b) * ss_ld + c] : 0.0f;
  v = __builtin_fmaf(v, s, h);
  if (in.post_relu) v = fmaxf(v, 0.0f);
  if (in.add) v += in.add[static_cast<long>(b) * in.add_ld + c];
  if (in.rseg.ptr) v += in.rseg.ptr[row * in.rseg.ld + c];
  out[row * ldo + c] = v;
}

// out (B, C) = max over the rows of every batch element of prologue(X): the global max-pooling of Pnet2Stage
// (pnet.py:27-40 of the reference: F.max_pool2d over all points) applied to a layer's lazily-activated output, without
// materialising the activation.  256 threads = 64 channels x 4 row slices; a wave reads 256-byte row pieces.
__global__ __launch_bounds__(256) void act_colmax_kernel(pdr_layer_in_t in, int C, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int rpb = in.rows_per_batch;
  float m = -__builtin_inff();
  if (c < C) {
    const ColSrc src = resolve_col(in, c);
    const int ss_ld = in.ss_ld > 0 ? in.ss_ld : C;
    const float s = in.scale ? in.scale[static_cast<long>(b) * ss_ld + c] : 1.0f;
    const float h = in.shift ? in.shift[static_cast<long>(b) * ss_ld + c] : 0.0f;
    const float a = in.add ? in.add[static_cast<long>(b) * in.add_ld + c] : 0.0f;
    const long row0 = static_cast<long>(b) * rpb;
    for (int r = sl; r < rpb; r += 16) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 4 * u;
        v[u] = load_col(src, row0 + (rr < rpb ? rr : r));       // (clamped: unconditional loads, all in flight)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float x = v[u];
        if (in.pre_relu) x = fmaxf(x, 0.0f);
        x = __builtin_fmaf(x, s, h);
        if (in.post_relu) x = fmaxf(x, 0.0f);
        m = fmaxf(m, x + a);
      }
    }
  }
  red[sl][cl] = m;
  __syncthreads();
  if (sl == 0 && c < C)
    out[static_cast<long>(b) * C + c] = fmaxf(fmaxf(red[0][cl], red[1][cl]), fmaxf(red[2][cl], red[3][cl]));
}

// GroupNorm(G groups over the first Cn of C channels, eps) folded to y = x*scale + shift;
// channels >= Cn pass through (MyGroupNorm).  n = elements per channel per batch.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ chan_stats, int C,
                                                          int Cn, int G, double n, float eps,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float sc = 1.0f, sh = 0.0f;
  if (c < Cn) {
    const int cpg = Cn / G;
    const int g0 = (c / cpg) * cpg;
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
      s1 += chan_stats[(static_cast<long>(b) * C + g0 + j) * 2 + 0];
      s2 += chan_stats[(static_cast<long>(b) * C + g0 + j) * 2 + 1];
    }
    const double cnt = n * cpg;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    sc = rstd * gamma[c];
    sh = __builtin_fmaf(-sc, static_cast<float>(mean), beta[c]);
  }
  scale[static_cast<long>(b) * C + c] = sc;
  shift[static_cast<long>(b) * C + c] = sh;
}

}  // namespace

// Tile shape selection.  Narrow outputs (Cout <= 64) use tall 256-row tiles with all four
// waves stacked along the rows; wide outputs 128 x 128 tiles (2 x 2 waves); small batch
// elements (few rows per GroupNorm instance) shrink the row tile.
namespace {
// Layers with <= 4 input channels and no prologue -- the per-query / per-source coordinate tables of the split
// first convs, xyz . W (+ bias): three fmas per output and a pure write stream, not a job for 32 x 32 MFMA tiles
// (30 launches per step).  A thread owns 4 consecutive output columns of one row; accumulation order = the tile
// kernels' (bias first, then the channels in MFMA order), so the results are the same bits.
__global__ __launch_bounds__(256) void fused_layer_thin_kernel(const float* __restrict__ X, int ldx, int shift,
                                                               int Cin, const float* __restrict__ Wt, int ldw,
                                                               const float* __restrict__ bias, int Cout,
                                                               float* __restrict__ Y, int ldy, long P, int qshift) {
  // 256 threads = (256 >> qshift) rows x (1 << qshift) column quads; a thread keeps its quad's weights in registers
  // and walks kThinIters rows; a wave stores whole contiguous row pieces
  constexpr int kThinIters = 8;
  const int ql = 1 << qshift;
  const int cq = threadIdx.x & (ql - 1), rl = threadIdx.x >> qshift;
  const int c0 = 4 * (static_cast<int>(blockIdx.y) * ql + cq);
  if (c0 >= Cout) return;                             // (columns up to the 4-padded width are written)
  const int rpp = 256 >> qshift;
  float4 w[4];
  float b4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    w[k] = k < Cin ? *reinterpret_cast<const float4*>(Wt + static_cast<long>(k) * ldw + c0) : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = (bias && c0 + j < Cout) ? bias[c0 + j] : 0.0f;
  const long r0 = static_cast<long>(blockIdx.x) * (rpp * kThinIters) + rl;
  // all eight row loads in flight before the first store (a load inside `if (row < P)` waits for itself AND -- stores
  // count in vmcnt on gfx9 -- for the previous trip's store: eight dependent round trips per workgroup)
  float4 xr[kThinIters];
#pragma unroll
  for (int it = 0; it < kThinIters; ++it) {
    const long row = r0 + static_cast<long>(it) * rpp;
    xr[it] = *reinterpret_cast<const float4*>(X + ((row < P ? row : P - 1) >> shift) * ldx);
  }
#pragma unroll
  for (int it = 0; it < kThinIters; ++it) {
    const long row = r0 + static_cast<long>(it) * rpp;
    if (row < P) {
      const float4 x = xr[it];
      const float xs[4] = {x.x, x.y, x.z, x.w};
      float acc[4] = {b4[0], b4[1], b4[2], b4[3]};
      // channel order 0, 2, 1, 3: the tile kernels' first MFMA of a channel quad multiplies the pair (0, 2), the
      // second (1, 3)
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = ((o & 1) << 1) | (o >> 1);
        if (k < Cin) {
          acc[0] = __builtin_fmaf(xs[k], w[k].x, acc[0]);
          acc[1] = __builtin_fmaf(xs[k], w[k].y, acc[1]);
          acc[2] = __builtin_fmaf(xs[k], w[k].z, acc[2]);
          acc[3] = __builtin_fmaf(xs[k], w[k].w, acc[3]);
        }
      }
      *reinterpret_cast<float4*>(Y + row * ldy + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}
}  // namespace

namespace {
// Tile shapes.  "Tall" shapes stack all four waves along the rows and give every wave the full
// output width (one column block => the input is read exactly once): used whenever Cout <= 160.
// Wide outputs use 128 x 128 tiles (2 x 2 waves) over a 2-D grid.
struct TileCfg { int tm, tn, id; };
// narrow outputs: 128-row tiles staged in 32-channel chunks (ids 7, 8) instead of 256-row tiles in 16-channel
// chunks (ids 0, 1): a row of <= 32 channels is then fetched as ONE whole 128-byte line (the half-line fetches of
// the 16-channel chunks were re-read from HBM).  Option narrow_kc32 = 0: the 256-row tiles.
inline bool narrow_kc32() { return pdr::option(pdr::OPT_NARROW_KC32) != 0; }
inline TileCfg pick_tile(int rows_per_batch, int Cout) {
  if (narrow_kc32() && rows_per_batch >= 128 && Cout <= 32) return {128, 32, 7};
  if (narrow_kc32() && rows_per_batch >= 128 && Cout <= 64) return {128, 64, 8};
  if (rows_per_batch >= 256 && Cout <= 32) return {256, 32, 0};
  if (rows_per_batch >= 256 && Cout <= 64) return {256, 64, 1};
  if (rows_per_batch >= 128 && Cout <= 96) return {128, 96, 2};
  if (rows_per_batch >= 128 && Cout > 128 && Cout <= 160) return {128, 160, 3};
  // (64-row tiles for the deep levels, whose 128 x 128 tiling has fewer jobs than the 512 resident workgroups -- B = 32:
  // 16 k rows x 128 columns = 128 jobs -- measured 8.72-8.77 vs 8.75-8.77 ms per step in round 4: no gain, not taken)
  if (rows_per_batch >= 128) return {128, 128, 4};
  if (rows_per_batch >= 64) return {64, 128, 5};
  return {32, 128, 6};
}
}  // namespace

extern "C" int pdr_fused_layer_tile_rows(int rows_per_batch, int Cout) {
  if (rows_per_batch <= 0 || Cout <= 0) return 0;
  return pick_tile(rows_per_batch, Cout).tm;
}

// which kernel instantiation pdr_fused_layer() launches (0..6, see pick_tile); lets profilers and
// bench.py attribute a launch to its kernel symbol
extern "C" int pdr_fused_layer_variant(int rows_per_batch, int Cout) {
  if (rows_per_batch <= 0 || Cout <= 0) return -1;
  return pick_tile(rows_per_batch, Cout).id;
}

// The dispatch decision of pdr_fused_layer, shared with pdr_fused_layer_plan (profilers / bench.py attribute a
// call to the kernel symbol it launches without duplicating these rules).
namespace {
struct LayerPlan {
  TileCfg t;
  bool vec, gath, radd, ws, knn, thin;
  long ntiles;
  int ncol;
  // right-sized launch of a tiny layer (see pdr_fused_layer): 0 = none, 5 / 4 = the 64 x 64 / 128 x 64 tiles of the
  // uniform-wave kernel with 128-channel chunks in place of the wave-specialised 64 x 128 / 128 x 128 ones, 6 = 32-row
  // tiles with 128-channel chunks
  int deep;
};

bool deep_chunks() { return pdr::option(pdr::OPT_DEEP_CHUNKS) != 0; }

bool use_ws_kernels() { return pdr::option(pdr::OPT_FUSED_WS) != 0; }

int plan_layer(const pdr_layer_in_t* in, long P, int Cin, const float* Wt, int ldw, int Cout, const float* Y,
               int ldy, LayerPlan* pl) {
  if (!in || !Wt || P < 0 || Cin <= 0 || Cout <= 0 || in->n_seg < 1 || in->n_seg > 4 || ldw < Cout ||
      (Y && ldy < Cout))
    return PDR_EINVAL;
  // the weight matrix is staged with 16-B loads: packed with a 4-float-aligned leading dimension
  if (ldw % 4 != 0 || reinterpret_cast<uintptr_t>(Wt) % 16 != 0) return PDR_EINVAL;
  int ctot = 0;
  for (int s = 0; s < in->n_seg; ++s) {
    if (!in->seg[s].ptr || in->seg[s].C <= 0 || in->seg[s].row_div < 1) return PDR_EINVAL;
    if (in->seg[s].row_div & (in->seg[s].row_div - 1)) return PDR_EUNSUPPORTED;  // power of two only
    ctot += in->seg[s].C;
  }
  if (ctot != Cin || in->rows_per_batch <= 0 || P % in->rows_per_batch != 0) return PDR_EINVAL;
  if (in->oadd && (in->oadd_div < 1 || (in->oadd_div & (in->oadd_div - 1)) || in->oadd_ld < Cout))
    return PDR_EINVAL;
  if (in->ss_ld != 0 && in->ss_ld < Cin) return PDR_EINVAL;
  const TileCfg t = pick_tile(in->rows_per_batch, Cout);
  for (int sg = 0; sg < in->n_seg; ++sg) {
    // every tile must start on a multiple of the broadcast divisor
    const int d = in->seg[sg].row_div;
    if (in->rows_per_batch % d != 0 || (in->rows_per_batch > t.tm && t.tm % d != 0)) return PDR_EUNSUPPORTED;
  }
  const long nb = P / in->rows_per_batch;
  // rows of `partial` per batch element: at least its tiles (a smaller stride would fold tiles of two batch elements
  // into one row); weighted statistics come with a weight
  if (in->partial_tpb > 0 && in->partial_tpb < (in->rows_per_batch + t.tm - 1) / t.tm) return PDR_EINVAL;
  pl->t = t;
  pl->ntiles = nb * ((in->rows_per_batch + t.tm - 1) / t.tm);
  pl->ncol = (Cout + t.tn - 1) / t.tn;
  // vector (float4) A staging needs every source 16-B aligned with a leading dimension that is a
  // multiple of 4 floats and rows padded to a multiple of 4 channels
  auto aligned = [](const float* p, int ld, int C) {
    return reinterpret_cast<uintptr_t>(p) % 16 == 0 && ld % 4 == 0 && ld >= ((C + 3) & ~3);
  };
  bool vec = true, gath = false;
  for (int sg = 0; sg < in->n_seg; ++sg) {
    const pdr_seg_t& g = in->seg[sg];
    vec = vec && aligned(g.ptr, g.ld, g.C);
    if (g.gV) {
      gath = true;
      vec = vec && aligned(g.gV, g.g_ldv, g.C) && (!g.gV0 || aligned(g.gV0, g.g_ldv, g.C));
      if (g.row_div != 1 || g.g_nsrc <= 0) return PDR_EINVAL;
    }
  }
  const bool radd = in->rseg.ptr != nullptr;
  if (radd) {
    vec = vec && in->n_seg == 1 && aligned(in->rseg.ptr, in->rseg.ld, Cin);
    if (in->rseg.gV) {
      gath = true;
      vec = vec && aligned(in->rseg.gV, in->rseg.g_ldv, Cin) &&
            (!in->rseg.gV0 || aligned(in->rseg.gV0, in->rseg.g_ldv, Cin));
    }
  }
  if (gath) {
    // gathered sources exist only in the vector path; one shared index array, K a power of two that
    // divides the row tile so that a tile starts on a query boundary
    if (!vec || !in->gidx || in->gK <= 0 || (in->gK & (in->gK - 1)) || in->rows_per_batch % in->gK != 0 ||
        t.tm % in->gK != 0)
      return PDR_EUNSUPPORTED;
    for (int sg = 0; sg < in->n_seg; ++sg)
      if (in->seg[sg].gV && in->gcnt && !in->seg[sg].gV0) return PDR_EINVAL;
  }
  bool knn = false;
  for (int sg = 0; sg < in->n_seg; ++sg) knn = knn || in->seg[sg].g_r1 || in->seg[sg].g_r2;
  knn = knn || (in->rseg.gV && (in->rseg.g_r1 || in->rseg.g_r2));
  pl->knn = knn;
  pl->vec = vec;
  pl->gath = gath;
  pl->radd = radd;
  // steady-state layers (float4-staged sources): wave-specialised kernel where an instantiation exists
  pl->ws = use_ws_kernels() && vec && pdr::fused_layer_ws_supported(t.id, radd, gath, *in, Cin);
  // TINY layers (round 5; option deep_chunks = 0: never).  The per-point layers of the deep levels are launches of a few
  // dozen workgroups, each a serial walk over the input channels: their time is (chunks) x (load latency) + (MFMAs per
  // wave) x 64 cycles on a chip that is half to seven eighths empty.  Where every workgroup of the launch is resident
  // at once they run on the uniform-wave kernel with 128-channel chunks (a quarter of the round trips) and, for the
  // 64- / 128-row tiles, half-width tiles (twice the workgroups, half the MFMAs per wave):
  //   32-row tiles (batch elements of < 64 rows: 16 points per cloud), <= 256 jobs          -> 32 x 128, 128-channel chunks
  //   64-row tiles (64 .. 127 rows: 64 points), plain sources, <= 512 jobs of 64 x 64       -> 64 x 64,  128-channel chunks
  //   128-row tiles (128 .. 511 rows at B = 32: 256 points), plain sources, <= 128 jobs     -> 128 x 64, 128-channel chunks
  // and, for plain sources, the first two narrower still with the K walk split among the waves (pdr_fused_layer):
  // 32 x 32 tiles / 4 ways, 64 x 32 tiles / 2 ways.
  // (128 for the last: beyond that every CU already holds a workgroup and the matrix pipes are what the launch waits
  // for -- bound at 256 / 512: step 5.85 / 6.06 ms against 5.79 at 128.)  Measured alone on the
  // chip, B = 32: 16 rows 512 -> 512 35.8 -> 27.6 us, 64 rows 256 -> 256 16.6 -> 14.0, 256 rows 128 -> 128 16.2 -> 12.5,
  // 256 -> 256 26.7 -> 22.0, 512 rows 128 -> 128 17.2 -> 13.6; step 6.03 -> 5.87 ms (profiles/r5_tiny_layers_ab.txt).
  pl->deep = 0;
  if (deep_chunks() && Cin > 64 && !in->tile_list) {
    if (t.id == 6 && pl->ntiles * pl->ncol <= pdr::option(pdr::OPT_DEEP_JOBS32)) pl->deep = 6;
    else if (t.id == 5 && vec && !gath && pl->ntiles * ((Cout + 63) / 64) <= pdr::option(pdr::OPT_DEEP_JOBS64)) pl->deep = 5;
    else if (t.id == 4 && vec && !gath && pl->ntiles * pl->ncol <= 128) pl->deep = 4;
  }
  // <= 4 input channels, nothing to apply on the way in, 16-byte rows on both sides: the thin kernel (statistics
  // are decided by the caller: pdr_fused_layer uses it only without `partial`)
  pl->thin = vec && Cin <= 4 && in->n_seg == 1 && !gath && !radd && !in->scale && !in->shift && !in->add &&
             !in->pre_relu && !in->post_relu && !in->oadd && Y && ldy % 4 == 0 &&
             reinterpret_cast<uintptr_t>(Y) % 16 == 0 && ldw >= ((Cout + 3) & ~3) && ldy >= ((Cout + 3) & ~3) &&
             P < (1L << 31);
  return PDR_OK;
}
}  // namespace

// out[0..6] = {wave-specialised kernel?, tile variant id, residual source?, gathered source (0 no / 1 ball / 2 kNN),
// float4 staging?, split-f16 arithmetic?, thin kernel when called without `partial`?} of the launch
// pdr_fused_layer would make for these arguments.
extern "C" int pdr_fused_layer_plan(const pdr_layer_in_t* in, long P, int Cin, const float* Wt, int ldw, int Cout,
                                    const float* Y, int ldy, int* out) {
  if (!out) return PDR_EINVAL;
  LayerPlan pl;
  const int rc = plan_layer(in, P, Cin, Wt, ldw, Cout, Y, ldy, &pl);
  if (rc != PDR_OK) return rc;
  out[0] = pl.ws && pl.deep != 4 && pl.deep != 5;
  out[1] = pl.t.id;
  out[2] = pl.radd;
  out[3] = pl.gath ? (pl.knn ? 2 : 1) : 0;
  out[4] = pl.vec;
  out[5] = 0;
  out[6] = pl.thin;
  out[7] = pl.deep ? 128 : 0;   // channels per chunk of a right-sized tiny launch (plan_layer), else 0
  return PDR_OK;
}

// Y (P, Cout; leading dim ldy) = prologue(X) . Wt + bias.  partial: NULL or
// (B * tiles_per_batch, Cout, 2) floats receiving per-tile sum / sum of squares of y
// (columns >= relu_col0: of relu(y)).
extern "C" int pdr_fused_layer(const pdr_layer_in_t* in, long P, int Cin, const float* Wt, int ldw,
                               const float* bias, int Cout, float* Y, int ldy, float* partial,
                               int relu_col0, pdr_stream_t stream) {
  if (!Y) return PDR_EINVAL;
  LayerPlan pl;
  const int prc = plan_layer(in, P, Cin, Wt, ldw, Cout, Y, ldy, &pl);
  if (prc != PDR_OK) return prc;
  if (P == 0) return PDR_OK;
  const TileCfg t = pl.t;
  const bool vec = pl.vec, gath = pl.gath, radd = pl.radd;
  hipStream_t s = pdr::as_stream(stream);
  const long ntiles = pl.ntiles;
  const int ncol = pl.ncol;
  // enough workgroups to fill 256 CUs a few times over; the rest is covered by the grid stride
  long gx = ntiles;
  const long cap = (256L * 6 + ncol - 1) / ncol;
  if (gx > cap) gx = cap;
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(ncol));
  const int nt = static_cast<int>(ntiles);
  // a tile subset (pdr_layer_in_t.tile_list) is walked by the wave-specialised kernels with 128-row tiles only
  if (in->tile_list && (!in->n_tiles || !pl.ws || t.tm != 128)) return PDR_EUNSUPPORTED;
  if (pl.thin && !partial && !in->tile_list) {
    const int c4n = (Cout + 3) / 4;
    int qshift = 3;                                    // 8 .. 64 column quads per row of threads
    while (qshift < 6 && (1 << qshift) < c4n) ++qshift;
    const int ql = 1 << qshift, rows_per_block = (256 >> qshift) * 8;
    const dim3 tgrid(static_cast<unsigned>((P + rows_per_block - 1) / rows_per_block),
                     static_cast<unsigned>((c4n + ql - 1) / ql));
    hipLaunchKernelGGL(fused_layer_thin_kernel, tgrid, dim3(256), 0, s, in->seg[0].ptr, in->seg[0].ld,
                       __builtin_ctz(in->seg[0].row_div), Cin, Wt, ldw, bias, Cout, Y, ldy, P, qshift);
    return pdr::check_launch();
  }
  // ... and their waves split the K walk (fused_layer_kernel's KS): 4 ways on 32 x 32 tiles for the 32-row family (16
  // rows per cloud: a wave's chain of Cin / 2 MFMAs was most of the launch), 2 ways on 64 x 32 tiles for the 64-row
  // family; not for the 128-row family (128 x 32 tiles, measured: 256 rows 256 -> 256 22.0 -> 54.3 us -- four times the
  // workgroups re-reading the same 128 input rows).  Plain sources.  Option deep_ks = 0: no split (A/B).
  const bool deep_ks = pdr::option(pdr::OPT_DEEP_KS) != 0;
  if ((pl.deep == 6 || pl.deep == 5) && deep_ks && vec && !gath) {
#define PDR_DEEP_KS_LAUNCH(RT, WR, WC, KSV, TNV)                                                                    \
  do {                                                                                                                \
    const dim3 gk(static_cast<unsigned>(ntiles), static_cast<unsigned>((Cout + TNV - 1) / TNV));                      \
    if (radd)                                                                                                         \
      hipLaunchKernelGGL((fused_layer_kernel<RT, 1, WR, WC, 128, true, true, false, false, KSV>), gk, dim3(256), 0, s, \
                         *in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, nt);                              \
    else                                                                                                              \
      hipLaunchKernelGGL((fused_layer_kernel<RT, 1, WR, WC, 128, false, true, false, false, KSV>), gk, dim3(256), 0, s, \
                         *in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, nt);                              \
  } while (0)
    if (pl.deep == 6) PDR_DEEP_KS_LAUNCH(1, 1, 1, 4, 32);
    else PDR_DEEP_KS_LAUNCH(1, 2, 1, 2, 32);
#undef PDR_DEEP_KS_LAUNCH
    return pdr::check_launch();
  }
  if (pl.deep == 5 || pl.deep == 4) {   // right-sized tiny layer (plan_layer)
    const dim3 g64(static_cast<unsigned>(ntiles), static_cast<unsigned>((Cout + 63) / 64));
#define PDR_DEEP(RT, WR)                                                                                          \
  do {                                                                                                            \
    if (radd)                                                                                                     \
      hipLaunchKernelGGL((fused_layer_kernel<RT, 1, WR, 2, 128, true, true, false>), g64, dim3(256), 0, s, *in, Cin, \
                         Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, nt);                                    \
    else                                                                                                          \
      hipLaunchKernelGGL((fused_layer_kernel<RT, 1, WR, 2, 128, false, true, false>), g64, dim3(256), 0, s, *in, Cin, \
                         Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, nt);                                    \
  } while (0)
    if (pl.deep == 5) PDR_DEEP(1, 2);
    else PDR_DEEP(2, 2);
#undef PDR_DEEP
    return pdr::check_launch();
  }
  if (pl.ws &&
      pdr::launch_fused_layer_ws(t.id, radd, gath, *in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, nt,
                                 ncol, s))
    return pdr::check_launch();
  // kNN-form gathered sources exist in the wave-specialised kernel only: the caller materialises instead
  if (pl.knn || in->tile_list) return PDR_EUNSUPPORTED;
#define PDR_LAUNCH_V(RT, CT, WR, WC, KC, RADD, VEC, GATH)                                            \
  hipLaunchKernelGGL((fused_layer_kernel<RT, CT, WR, WC, KC, RADD, VEC, GATH>), grid, dim3(256), 0, s, \
                     *in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0, nt)
#define PDR_LAUNCH(RT, CT, WR, WC, KC)                                \
  do {                                                                \
    if (radd) {                                                       \
      if (gath) PDR_LAUNCH_V(RT, CT, WR, WC, KC, true, true, true);   \
      else if (vec) PDR_LAUNCH_V(RT, CT, WR, WC, KC, true, true, false); \
      else PDR_LAUNCH_V(RT, CT, WR, WC, KC, true, false, false);      \
    } else {                                                          \
      if (gath) PDR_LAUNCH_V(RT, CT, WR, WC, KC, false, true, true);  \
      else if (vec) PDR_LAUNCH_V(RT, CT, WR, WC, KC, false, true, false); \
      else PDR_LAUNCH_V(RT, CT, WR, WC, KC, false, false, false);     \
    }                                                                 \
  } while (0)
  switch (t.id) {
    case 0: PDR_LAUNCH(2, 1, 4, 1, 16); break;
    case 1: PDR_LAUNCH(2, 2, 4, 1, 16); break;
    case 2: PDR_LAUNCH(1, 3, 4, 1, 32); break;
    case 3: PDR_LAUNCH(1, 5, 4, 1, 32); break;
    case 4: PDR_LAUNCH(2, 2, 2, 2, 32); break;
    case 5: PDR_LAUNCH(1, 2, 2, 2, 32); break;
    case 7: PDR_LAUNCH(1, 1, 4, 1, 32); break;
    case 8: PDR_LAUNCH(1, 2, 4, 1, 32); break;
    default:
      if (pl.deep == 6) PDR_LAUNCH(1, 1, 1, 4, 128);   // right-sized tiny layer (plan_layer)
      else PDR_LAUNCH(1, 1, 1, 4, 32);
      break;
  }
#undef PDR_LAUNCH
#undef PDR_LAUNCH_V
  return pdr::check_launch();
}

// The same layer over TWO row sets in ONE launch (round 6): `in` / Y / partial = the tile subset of a deduplicated block's
// per-neighbour rows (pdr_layer_in_t.tile_list; plain or ball-gathered sources, no residual), `in2` / Y2 / partial2 = its
// per-QUERY rows (plain sources, weighted statistics).  Replaces two dependent launches of a block's launch chain by
// one; the results are those of the two pdr_fused_layer calls, bit for bit (same tiles, same kernels' arithmetic).
// PDR_EUNSUPPORTED when the pair has no wave-specialised 128-row instantiation: the caller launches them one by one.
extern "C" int pdr_fused_layer_pair(const pdr_layer_in_t* in, long P, const pdr_layer_in_t* in2, long P2, int Cin,
                                    const float* Wt, int ldw, const float* bias, int Cout, float* Y, int ldy,
                                    float* Y2, int ldy2, float* partial, float* partial2, int relu_col0,
                                    pdr_stream_t stream) {
  if (!Y || !Y2 || !in || !in2) return PDR_EINVAL;
  LayerPlan pl, pl2;
  int rc = plan_layer(in, P, Cin, Wt, ldw, Cout, Y, ldy, &pl);
  if (rc != PDR_OK) return rc;
  rc = plan_layer(in2, P2, Cin, Wt, ldw, Cout, Y2, ldy2, &pl2);
  if (rc != PDR_OK) return rc;
  if (P == 0 || P2 == 0) return PDR_EUNSUPPORTED;
  if ((partial == nullptr) != (partial2 == nullptr)) return PDR_EINVAL;
  // the first problem decides the tile shape: a listed launch of 128-row tiles; the second runs on the same tiles
  // (its batch elements may be shorter than a tile: partial tiles, one row of `partial2` per 128 rows)
  if (!in->tile_list || !in->n_tiles || pl.t.tm != 128 || !pl.ws || !pl.vec || !pl2.vec || pl.radd || pl2.radd ||
      pl2.gath || pl.knn || in2->tile_list)
    return PDR_EUNSUPPORTED;
  const int tpb2 = (in2->rows_per_batch + 127) / 128;
  if (in2->partial_tpb > 0 && in2->partial_tpb < tpb2) return PDR_EINVAL;
  for (int sg = 0; sg < in2->n_seg; ++sg)
    if (in2->seg[sg].row_div != 1) return PDR_EUNSUPPORTED;
  pdr::WsTwin tw;
  tw.in[1] = *in2;
  tw.Y[1] = Y2;
  tw.partial[1] = partial2;
  tw.ldy[1] = ldy2;
  tw.n_row_tiles[1] = static_cast<int>((P2 / in2->rows_per_batch) * tpb2);
  tw.gx = 0;
  if (!pdr::launch_fused_layer_ws_pair(pl.t.id, pl.gath, *in, Cin, Wt, ldw, bias, Cout, Y, ldy, partial, relu_col0,
                                       static_cast<int>(pl.ntiles), pl.ncol, tw, pdr::as_stream(stream)))
    return PDR_EUNSUPPORTED;
  return pdr::check_launch();
}

// pdr_fused_layer with SPLIT-f16 arithmetic (opt-in): both GEMM operands are split into f16 hi + lo parts and the
// product is accumulated as xh wh + xh wl + xl wh on v_mfma_f32_32x32x16_f16 with fp32 accumulation (~22 mantissa
// bits kept).  Wp = weight image of pdr-side packing (see include/pdr_hip.h), nchunks = K-chunks per column block.
// Only the wave-specialised tile variants 4, 5 and 8 carry this mode: PDR_EUNSUPPORTED otherwise (callers fall back to
// the exact fp32 entry point).
extern "C" int pdr_fused_layer_f16x3(const pdr_layer_in_t* in, long P, int Cin, const void* Wp, int nchunks,
                                     const float* bias, int Cout, float* Y, int ldy, float* partial,
                                     int relu_col0, pdr_stream_t stream) {
  if (!Y || !Wp || nchunks <= 0 || reinterpret_cast<uintptr_t>(Wp) % 16 != 0) return PDR_EINVAL;
  LayerPlan pl;
  // (the fp32 weight arguments of the plan are placeholders: alignment-clean dummies)
  const int prc = plan_layer(in, P, Cin, reinterpret_cast<const float*>(Wp), (Cout + 3) & ~3, Cout, Y, ldy, &pl);
  if (prc != PDR_OK) return prc;
  if (P == 0) return PDR_OK;
  if (!pl.ws || (pl.t.id != 4 && pl.t.id != 5 && pl.t.id != 8)) return PDR_EUNSUPPORTED;
  // a tile subset is a list of 128-row tile numbers with its length on the device (as pdr_fused_layer)
  if (in->tile_list && (!in->n_tiles || pl.t.tm != 128)) return PDR_EUNSUPPORTED;
  int nch = 0;
  for (int s = 0; s < in->n_seg; ++s) nch += (in->seg[s].C + 31) / 32;
  if (nch != nchunks) return PDR_EINVAL;   // the image was packed for another segment structure
  if (!pdr::launch_fused_layer_ws(pl.t.id, pl.radd, pl.gath, *in, Cin, reinterpret_cast<const float*>(Wp), nchunks,
                                  bias, Cout, Y, ldy, partial, relu_col0, static_cast<int>(pl.ntiles), pl.ncol,
                                  pdr::as_stream(stream), true))
    return PDR_EUNSUPPORTED;
  return pdr::check_launch();
}

// scores = prologue(X) . Wt + bias are consumed by the POOL epilogue:
//   out[q, :] = sum_k softmax_k(mask(scores))[k, :] * act(values[q K + k, :] * vscale + vshift)
// K in {8, 16, 32}; Cout = D (channels of scores, values and out).
namespace {
// pooled launch that also writes the pooled rows of the skipped tiles' queries (pdr_layer_in_t.patch_values): the
// kernel reads patch_values / writes `out` in 16-byte pieces of 4 channels and reads patch_w[q] for every query
bool pool_patch_args_ok(const pdr_layer_in_t& in, int D, const float* out, int ldo) {
  if (!in.patch_values) return true;
  return in.patch_w != nullptr && in.patch_ld >= D && D % 4 == 0 && in.patch_ld % 4 == 0 && ldo % 4 == 0 &&
         reinterpret_cast<uintptr_t>(in.patch_values) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
}
}  // namespace

// pdr_fused_layer_pool with the score conv on split-f16 arithmetic (packed weight image as pdr_fused_layer_f16x3);
// the 128-column wave-specialised tiles only: PDR_EUNSUPPORTED otherwise (the caller uses the exact entry point).
extern "C" int pdr_fused_layer_pool_f16x3(const pdr_layer_in_t* in, long P, int Cin, const void* Wp, int nchunks,
                                          const float* bias, int D, const float* values, int ldv,
                                          const float* vscale, const float* vshift, int v_relu,
                                          const int* counts, int K, float* out, int ldo, pdr_stream_t stream) {
  if (!in || !Wp || nchunks <= 0 || reinterpret_cast<uintptr_t>(Wp) % 16 != 0 || !values || !out || P <= 0 ||
      Cin <= 0 || D <= 0 || in->n_seg < 1 || in->n_seg > 4 || ldv < D || ldo < D)
    return PDR_EINVAL;
  if (!pool_patch_args_ok(*in, D, out, ldo)) return PDR_EINVAL;
  if (!(K == 8 || K == 16 || K == 32)) return PDR_EUNSUPPORTED;
  if (in->rseg.ptr || in->oadd) return PDR_EUNSUPPORTED;
  int ctot = 0, nch = 0;
  bool vec = true;
  for (int s = 0; s < in->n_seg; ++s) {
    const pdr_seg_t& g = in->seg[s];
    if (!g.ptr || g.C <= 0 || g.row_div != 1 || g.gV) return PDR_EUNSUPPORTED;
    vec = vec && reinterpret_cast<uintptr_t>(g.ptr) % 16 == 0 && g.ld % 4 == 0 && g.ld >= ((g.C + 3) & ~3);
    ctot += g.C;
    nch += (g.C + 31) / 32;
  }
  if (ctot != Cin || in->rows_per_batch <= 0 || P % in->rows_per_batch != 0 || in->rows_per_batch % 32 != 0)
    return PDR_EINVAL;
  if (nch != nchunks) return PDR_EINVAL;   // the image was packed for another segment structure
  if (!vec || !use_ws_kernels()) return PDR_EUNSUPPORTED;
  const TileCfg t = pick_tile(in->rows_per_batch, D);
  if ((t.id != 4 && t.id != 5 && t.id != 8) || in->rows_per_batch % t.tm != 0) return PDR_EUNSUPPORTED;
  if (in->tile_list && (!in->n_tiles || t.tm != 128)) return PDR_EUNSUPPORTED;
  const long nb = P / in->rows_per_batch;
  const long ntiles = nb * ((in->rows_per_batch + t.tm - 1) / t.tm);
  const int ncol = (D + t.tn - 1) / t.tn;
  PoolArgs pa{values, vscale, vshift, counts, out, ldv, ldo, K, v_relu};
  if (!pdr::launch_fused_layer_ws(t.id, false, false, *in, Cin, reinterpret_cast<const float*>(Wp), nchunks, bias, D,
                                  nullptr, 0, nullptr, D, static_cast<int>(ntiles), ncol, pdr::as_stream(stream), true,
                                  &pa))
    return PDR_EUNSUPPORTED;
  return pdr::check_launch();
}

extern "C" int pdr_fused_layer_pool(const pdr_layer_in_t* in, long P, int Cin, const float* Wt, int ldw,
                                    const float* bias, int D, const float* values, int ldv,
                                    const float* vscale, const float* vshift, int v_relu,
                                    const int* counts, int K, float* out, int ldo,
                                    pdr_stream_t stream) {
  if (!in || !Wt || !values || !out || P <= 0 || Cin <= 0 || D <= 0 || in->n_seg < 1 || in->n_seg > 4 ||
      ldw < D || ldw % 4 != 0 || reinterpret_cast<uintptr_t>(Wt) % 16 != 0 || ldv < D || ldo < D)
    return PDR_EINVAL;
  if (!pool_patch_args_ok(*in, D, out, ldo)) return PDR_EINVAL;
  if (!(K == 8 || K == 16 || K == 32)) return PDR_EUNSUPPORTED;
  if (in->rseg.ptr || in->oadd) return PDR_EUNSUPPORTED;
  int ctot = 0;
  bool vec = true;
  for (int s = 0; s < in->n_seg; ++s) {
    const pdr_seg_t& g = in->seg[s];
    if (!g.ptr || g.C <= 0 || g.row_div != 1 || g.gV) return PDR_EUNSUPPORTED;
    vec = vec && reinterpret_cast<uintptr_t>(g.ptr) % 16 == 0 && g.ld % 4 == 0 && g.ld >= ((g.C + 3) & ~3);
    ctot += g.C;
  }
  if (ctot != Cin || in->rows_per_batch <= 0 || P % in->rows_per_batch != 0 || in->rows_per_batch % 32 != 0)
    return PDR_EINVAL;
  if (!vec) return PDR_EUNSUPPORTED;
  const TileCfg t = pick_tile(in->rows_per_batch, D);
  if (in->tile_list && (!in->n_tiles || t.tm != 128)) return PDR_EUNSUPPORTED;
  hipStream_t s = pdr::as_stream(stream);
  const long nb = P / in->rows_per_batch;
  const long ntiles = nb * ((in->rows_per_batch + t.tm - 1) / t.tm);
  const int ncol = (D + t.tn - 1) / t.tn;
  long gx = ntiles;
  const long cap = (256L * 6 + ncol - 1) / ncol;
  if (gx > cap) gx = cap;
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(ncol));
  const int nt = static_cast<int>(ntiles);
  PoolArgs pa{values, vscale, vshift, counts, out, ldv, ldo, K, v_relu};
  // wave-specialised kernels carry the pooled epilogue for the tile shapes whose rows tile whole queries
  if (use_ws_kernels() && in->rows_per_batch % t.tm == 0 &&
      pdr::launch_fused_layer_ws(t.id, false, false, *in, Cin, Wt, ldw, bias, D, nullptr, 0, nullptr, D, nt, ncol, s,
                                 false, &pa))
    return pdr::check_launch();
  // tile subsets / row maps / patched rows: wave-specialised kernels only
  if (in->tile_list || in->out_rows || in->patch_values) return PDR_EUNSUPPORTED;
#define PDR_LAUNCH_P(RT, CT, WR, WC, KC)                                                              \
  hipLaunchKernelGGL((fused_layer_kernel<RT, CT, WR, WC, KC, false, true, false, true>), grid, dim3(256), \
                     0, s, *in, Cin, Wt, ldw, bias, D, static_cast<float*>(nullptr), 0,                   \
                     static_cast<float*>(nullptr), D, nt, pa)
  switch (t.id) {
    case 0: PDR_LAUNCH_P(2, 1, 4, 1, 16); break;
    case 1: PDR_LAUNCH_P(2, 2, 4, 1, 16); break;
    case 2: PDR_LAUNCH_P(1, 3, 4, 1, 32); break;
    case 3: PDR_LAUNCH_P(1, 5, 4, 1, 32); break;
    case 4: PDR_LAUNCH_P(2, 2, 2, 2, 32); break;
    case 5: PDR_LAUNCH_P(1, 2, 2, 2, 32); break;
    case 7: PDR_LAUNCH_P(1, 1, 4, 1, 32); break;
    case 8: PDR_LAUNCH_P(1, 2, 4, 1, 32); break;
    default: PDR_LAUNCH_P(1, 1, 1, 4, 32); break;
  }
#undef PDR_LAUNCH_P
  return pdr::check_launch();
}

extern "C" int pdr_gn_reduce(const float* partial, int ldp, int B, int tiles_per_batch, int C,
                             double mult, double* chan_stats, int Ctot, int coff,
                             pdr_stream_t stream) {
  if (!partial || !chan_stats || B <= 0 || tiles_per_batch <= 0 || C <= 0 || coff < 0 ||
      coff + C > Ctot || ldp < C)
    return PDR_EINVAL;
  hipLaunchKernelGGL(gn_reduce_kernel, dim3((C + 31) / 32, B), dim3(1024), 0, pdr::as_stream(stream),
                     partial, ldp, tiles_per_batch, C, mult, chan_stats, Ctot, coff);
  return pdr::check_launch();
}

// One-launch GroupNorm fold: up to two partial sources (second may be NULL) covering C = C0 + C1
// channels in order; see pdr_gn_reduce / pdr_gn_finalize for the semantics.
extern "C" int pdr_gn_fold(const float* part0, int ldp0, int tpb0, int C0, double mult0,
                           const float* part1, int ldp1, int tpb1, int C1, double mult1, int B, int Cn,
                           int G, double n, float eps, const float* gamma, const float* beta,
                           float* scale, float* shift, const int* nvalid0, int tpb_main0, const int* nvalid1,
                           int tpb_main1, pdr_stream_t stream) {
  if (!part0 || C0 <= 0 || tpb0 <= 0 || ldp0 < C0 || B <= 0 || G <= 0 || !scale || !shift)
    return PDR_EINVAL;
  if (part1 && (C1 <= 0 || tpb1 <= 0 || ldp1 < C1)) return PDR_EINVAL;
  if ((nvalid0 && (tpb_main0 <= 0 || tpb_main0 > tpb0)) || (nvalid1 && (!part1 || tpb_main1 <= 0 || tpb_main1 > tpb1)))
    return PDR_EINVAL;
  const int C = C0 + (part1 ? C1 : 0);
  if (Cn < 0 || Cn > C || (Cn > 0 && (Cn % G != 0 || !gamma || !beta))) return PDR_EINVAL;
  // (without a subset: nv = tpb_main = 0 -- the range [nv, tpb_main) of skipped rows is empty)
  FoldPart p0{part0, ldp0, tpb0, C0, mult0, nvalid0, nvalid0 ? tpb_main0 : 0};
  FoldPart p1{part1, ldp1, tpb1, part1 ? C1 : 0, mult1, nvalid1, nvalid1 ? tpb_main1 : 0};
  const bool small_form = pdr::option(pdr::OPT_GN_FOLD_SMALL) != 0;
  const int cpg = Cn > 0 ? Cn / G : 1;
  if (small_form && cpg <= 32) {
    // windows of whole groups covering <= 32 channels; one more workgroup row for pass-through channels
    const int CW = (32 / cpg) * cpg;
    const int nw = (Cn + CW - 1) / CW + (C > Cn ? 1 : 0);
    // (same sums in the same order for either U: slices of 8 tiles, rows ascending within a slice)
    if (tpb0 > 128 || (part1 && tpb1 > 128))
      hipLaunchKernelGGL(gn_fold_kernel<16>, dim3(nw, B), dim3(256), 0, pdr::as_stream(stream), p0, p1, C, Cn, G,
                         CW, n, eps, gamma, beta, scale, shift);
    else
      hipLaunchKernelGGL(gn_fold_kernel<4>, dim3(nw, B), dim3(256), 0, pdr::as_stream(stream), p0, p1, C, Cn, G,
                         CW, n, eps, gamma, beta, scale, shift);
    return pdr::check_launch();
  }
  if (static_cast<size_t>(C) * 16 > 48 * 1024) return PDR_EUNSUPPORTED;
  hipLaunchKernelGGL(gn_fold_wide_kernel, dim3(B), dim3(1024), static_cast<size_t>(C) * 16,
                     pdr::as_stream(stream), p0, p1, C, Cn, G, n, eps, gamma, beta, scale, shift);
  return pdr::check_launch();
}

extern "C" int pdr_apply_act(const pdr_layer_in_t* in, long P, int C, float* out, int ldo,
                             pdr_stream_t stream) {
  if (!in || !out || P < 0 || C <= 0 || in->n_seg < 1 || in->n_seg > 4 || in->rows_per_batch <= 0)
    return PDR_EINVAL;
  if (in->rseg.gV) return PDR_EUNSUPPORTED;   // gathered sources: pdr_fused_layer only
  if (P == 0) return PDR_OK;
  int ctot = 0;
  for (int s = 0; s < in->n_seg; ++s) {
    if (in->seg[s].row_div < 1 || (in->seg[s].row_div & (in->seg[s].row_div - 1))) return PDR_EUNSUPPORTED;
    if (in->seg[s].gV) return PDR_EUNSUPPORTED;
    ctot += in->seg[s].C;
  }
  if (ctot != C) return PDR_EINVAL;
  hipLaunchKernelGGL(apply_act_kernel, dim3(static_cast<unsigned>((P * C + 255) / 256)), dim3(256), 0,
                     pdr::as_stream(stream), *in, P, C, out, ldo);
  return pdr::check_launch();
}

extern "C" int pdr_act_colmax(const pdr_layer_in_t* in, long P, int C, float* out, pdr_stream_t stream) {
  if (!in || !out || P < 0 || C <= 0 || in->n_seg < 1 || in->n_seg > 4 || in->rows_per_batch <= 0 ||
      P % in->rows_per_batch != 0)
    return PDR_EINVAL;
  if (in->rseg.ptr || in->oadd) return PDR_EUNSUPPORTED;   // plain prologue only
  if (P == 0) return PDR_OK;
  int ctot = 0;
  for (int s = 0; s < in->n_seg; ++s) {
    if (!in->seg[s].ptr || in->seg[s].row_div < 1 || (in->seg[s].row_div & (in->seg[s].row_div - 1)))
      return PDR_EUNSUPPORTED;
    if (in->seg[s].gV) return PDR_EUNSUPPORTED;
    ctot += in->seg[s].C;
  }
  if (ctot != C) return PDR_EINVAL;
  const dim3 grid(static_cast<unsigned>((C + 63) / 64), static_cast<unsigned>(P / in->rows_per_batch));
  hipLaunchKernelGGL(act_colmax_kernel, grid, dim3(256), 0, pdr::as_stream(stream), *in, C, out);
  return pdr::check_launch();
}

extern "C" int pdr_gn_finalize(const double* chan_stats, int B, int C, int Cn, int G, double n,
                               float eps, const float* gamma, const float* beta, float* scale,
                               float* shift, pdr_stream_t stream) {
  if (!chan_stats || !scale || !shift || B <= 0 || C <= 0 || Cn < 0 || Cn > C || G <= 0 ||
      (Cn > 0 && (Cn % G != 0 || !gamma || !beta)))
    return PDR_EINVAL;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((C + 255) / 256, B), dim3(256), 0,
                     pdr::as_stream(stream), chan_stats, C, Cn, G, n, eps, gamma, beta, scale, shift);
  return pdr::check_launch();
}
