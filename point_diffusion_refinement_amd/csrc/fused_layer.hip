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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;    // K-chunk (input channels per LDS stage)
constexpr int TN = 128;   // output columns per pass
constexpr int PAD = 1;

__device__ __forceinline__ float load_a(const pdr_layer_in_t& in, long row, int c, int b) {
  // segment lookup (n_seg <= 4, wave-uniform per c only when KC-aligned; kept branchy but tiny)
  int c0 = 0;
  float v = 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < in.n_seg) {
      const int cs = in.seg[s].C;
      if (c >= c0 && c < c0 + cs) {
        const long srow = in.seg[s].row_div == 1 ? row : row / in.seg[s].row_div;
        v = in.seg[s].ptr[srow * in.seg[s].ld + (c - c0)];
      }
      c0 += cs;
    }
  }
  return v;
}

// TM in {128, 64, 32}: rows per workgroup.  Wave w owns row block (w % RW) and column
// group (w / RW) where RW = TM/32; each wave computes 32 rows x (TN / CW) columns.
template <int TM>
__global__ __launch_bounds__(256) void fused_layer_kernel(
    pdr_layer_in_t in, long P, int Cin, const float* __restrict__ Wt, const float* __restrict__ bias,
    int Cout, float* __restrict__ Y, int ldy, float* __restrict__ partial, int relu_col0) {
  constexpr int RW = TM / 32;        // row-waves
  constexpr int CW = 4 / RW;         // column-waves
  constexpr int NT = TN / 32 / CW;   // 32-col MFMA tiles per wave
  __shared__ float As[KC][TM + PAD];
  __shared__ float Bs[KC][TN + PAD];
  __shared__ float red[4][TN][2];    // cross-row-wave stats reduction

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int rw = wave % RW, cw = wave / RW;
  // tiles are cut per batch element (a tile never straddles two GroupNorm instances); the last
  // tile of a batch element may be partial
  const int tpb = (in.rows_per_batch + TM - 1) / TM;
  const int b = blockIdx.x / tpb;
  const int tb = blockIdx.x - b * tpb;
  const long row0 = static_cast<long>(b) * in.rows_per_batch + static_cast<long>(tb) * TM;
  const int nvalid = min(TM, in.rows_per_batch - tb * TM);
  const float* sc = in.scale ? in.scale + static_cast<long>(b) * Cin : nullptr;
  const float* sh = in.shift ? in.shift + static_cast<long>(b) * Cin : nullptr;
  const float* ad = in.add ? in.add + static_cast<long>(b) * in.add_ld : nullptr;

  for (int n0 = 0; n0 < Cout; n0 += TN) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    for (int k0 = 0; k0 < Cin; k0 += KC) {
      __syncthreads();
      // ---- stage A (with the fused prologue): thread -> (row = tid/32 + 8 i, c = tid%32)
      {
        const int c = k0 + (tid & 31);
        const bool cok = c < Cin;
        const float s = (cok && sc) ? sc[c] : 1.0f;
        const float h = (cok && sh) ? sh[c] : 0.0f;
        const float a = (cok && ad) ? ad[c] : 0.0f;
#pragma unroll 4
        for (int i = 0; i < TM / 8; ++i) {
          const int r = (tid >> 5) + 8 * i;
          const long row = row0 + r;
          float v = 0.0f;
          if (cok && r < nvalid) {
            v = load_a(in, row, c, b);
            if (in.pre_relu) v = fmaxf(v, 0.0f);
            v = __builtin_fmaf(v, s, h);
            if (in.post_relu) v = fmaxf(v, 0.0f);
            v += a;
            if (in.radd) v += in.radd[row * in.radd_ld + c];
          }
          As[tid & 31][r] = v;
        }
      }
      // ---- stage W chunk: Wt is (Cin, Cout) row-major -> Bs[k][n], coalesced along n
      {
#pragma unroll 4
        for (int i = 0; i < KC * TN / 256; ++i) {
          const int e = tid + 256 * i;
          const int k = e / TN, n = e % TN;
          const bool ok = (k0 + k) < Cin && (n0 + n) < Cout;
          Bs[k][n] = ok ? Wt[static_cast<long>(k0 + k) * Cout + n0 + n] : 0.0f;
        }
      }
      __syncthreads();
      // ---- MFMA: A operand lane l -> A[row = l&31][k = 2 kk + (l>>5)], B -> W[k][col = l&31]
      const int kl = lane >> 5, il = lane & 31;
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        const float a = As[2 * kk + kl][rw * 32 + il];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float w = Bs[2 * kk + kl][(cw * NT + t) * 32 + il];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc[t], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8 (reg>>2) + 4 (lane>>5)
    const int il = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + (cw * NT + t) * 32 + il;
      const bool colok = col < Cout;
      const float bv = (colok && bias) ? bias[col] : 0.0f;
      const bool relu_stat = col >= relu_col0;
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const long row = row0 + rw * 32 + rr;
        const float y = acc[t][r] + bv;
        if (colok && rw * 32 + rr < nvalid) {
          Y[row * ldy + col] = y;
          const float f = relu_stat ? fmaxf(y, 0.0f) : y;
          s1 += f;
          s2 = __builtin_fmaf(f, f, s2);
        }
      }
      if (partial) {
        // lanes l and l+32 hold the same column: fold, then reduce across the RW row-waves
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (hi == 0) {
          red[rw][(cw * NT + t) * 32 + il][0] = s1;
          red[rw][(cw * NT + t) * 32 + il][1] = s2;
        }
      }
    }
    if (partial) {
      __syncthreads();
      if (tid < TN && n0 + tid < Cout) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int w = 0; w < RW; ++w) {
          s1 += red[w][tid][0];
          s2 += red[w][tid][1];
        }
        float* o = partial + (static_cast<long>(blockIdx.x) * Cout + n0 + tid) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
  }
}

// chan_stats[b, coff + c] (double2) = mult * sum over the tiles of batch b of partial[tile, c]
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float* __restrict__ partial, int ldp,
                                                        int tiles_per_batch, int C, double mult,
                                                        double* __restrict__ chan_stats, int Ctot,
                                                        int coff) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  const float* p = partial + (static_cast<long>(b) * tiles_per_batch * ldp + c) * 2;
  for (int t = 0; t < tiles_per_batch; ++t) {
    s1 += p[static_cast<long>(t) * ldp * 2 + 0];
    s2 += p[static_cast<long>(t) * ldp * 2 + 1];
  }
  double* o = chan_stats + (static_cast<long>(b) * Ctot + coff + c) * 2;
  o[0] = s1 * mult;
  o[1] = s2 * mult;
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
  float v = load_a(in, row, c, b);
  if (in.pre_relu) v = fmaxf(v, 0.0f);
  const float s = in.scale ? in.scale[static_cast<long>(b) * C + c] : 1.0f;
  const float h = in.shift ? in.shift[static_cast<long>(b) * C + c] : 0.0f;
  v = __builtin_fmaf(v, s, h);
  if (in.post_relu) v = fmaxf(v, 0.0f);
  if (in.add) v += in.add[static_cast<long>(b) * in.add_ld + c];
  if (in.radd) v += in.radd[row * in.radd_ld + c];
  out[row * ldo + c] = v;
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

extern "C" int pdr_fused_layer_tile_rows(int rows_per_batch) {
  if (rows_per_batch <= 0) return 0;
  if (rows_per_batch >= 128) return 128;
  if (rows_per_batch >= 64) return 64;
  return 32;
}

// Y (P, Cout; leading dim ldy) = prologue(X) . Wt + bias.  partial: NULL or
// (P / tile_rows, Cout, 2) floats receiving per-tile sum / sum of squares of y
// (columns >= relu_col0: of relu(y)).
extern "C" int pdr_fused_layer(const pdr_layer_in_t* in, long P, int Cin, const float* Wt,
                               const float* bias, int Cout, float* Y, int ldy, float* partial,
                               int relu_col0, pdr_stream_t stream) {
  if (!in || !Wt || !Y || P < 0 || Cin <= 0 || Cout <= 0 || in->n_seg < 1 || in->n_seg > 4)
    return PDR_EINVAL;
  if (P == 0) return PDR_OK;
  int ctot = 0;
  for (int s = 0; s < in->n_seg; ++s) {
    if (!in->seg[s].ptr || in->seg[s].C <= 0 || in->seg[s].row_div < 1) return PDR_EINVAL;
    ctot += in->seg[s].C;
  }
  if (ctot != Cin) return PDR_EINVAL;
  const int tm = pdr_fused_layer_tile_rows(in->rows_per_batch);
  if (tm == 0 || P % in->rows_per_batch != 0) return PDR_EINVAL;
  hipStream_t s = pdr::as_stream(stream);
  const long nb = P / in->rows_per_batch;
  const dim3 grid(static_cast<unsigned>(nb * ((in->rows_per_batch + tm - 1) / tm)));
  if (tm == 128)
    hipLaunchKernelGGL(fused_layer_kernel<128>, grid, dim3(256), 0, s, *in, P, Cin, Wt, bias, Cout, Y,
                       ldy, partial, relu_col0);
  else if (tm == 64)
    hipLaunchKernelGGL(fused_layer_kernel<64>, grid, dim3(256), 0, s, *in, P, Cin, Wt, bias, Cout, Y,
                       ldy, partial, relu_col0);
  else
    hipLaunchKernelGGL(fused_layer_kernel<32>, grid, dim3(256), 0, s, *in, P, Cin, Wt, bias, Cout, Y,
                       ldy, partial, relu_col0);
  return pdr::check_launch();
}

extern "C" int pdr_gn_reduce(const float* partial, int ldp, int B, int tiles_per_batch, int C,
                             double mult, double* chan_stats, int Ctot, int coff,
                             pdr_stream_t stream) {
  if (!partial || !chan_stats || B <= 0 || tiles_per_batch <= 0 || C <= 0 || coff < 0 ||
      coff + C > Ctot || ldp < C)
    return PDR_EINVAL;
  hipLaunchKernelGGL(gn_reduce_kernel, dim3((C + 255) / 256, B), dim3(256), 0, pdr::as_stream(stream),
                     partial, ldp, tiles_per_batch, C, mult, chan_stats, Ctot, coff);
  return pdr::check_launch();
}

extern "C" int pdr_apply_act(const pdr_layer_in_t* in, long P, int C, float* out, int ldo,
                             pdr_stream_t stream) {
  if (!in || !out || P < 0 || C <= 0 || in->n_seg < 1 || in->n_seg > 4 || in->rows_per_batch <= 0)
    return PDR_EINVAL;
  if (P == 0) return PDR_OK;
  int ctot = 0;
  for (int s = 0; s < in->n_seg; ++s) ctot += in->seg[s].C;
  if (ctot != C) return PDR_EINVAL;
  hipLaunchKernelGGL(apply_act_kernel, dim3(static_cast<unsigned>((P * C + 255) / 256)), dim3(256), 0,
                     pdr::as_stream(stream), *in, P, C, out, ldo);
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
