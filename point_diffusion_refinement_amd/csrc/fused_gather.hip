// fused_gather.hip -- channel-last neighbourhood assembly and attention pooling.
//
// group_build : QueryAndGroup.forward (reference pointnet2_utils.py:332-438) in one pass:
//               G[b,j,k,:] = [ feats[b, idx[b,j,k], :] | rel xyz | abs xyz | centre xyz ]
//               with the subset=False patch (no neighbour -> the query itself, zero feature).
//               The reference runs group_points twice + 3 elementwise passes + 2 torch.cat over
//               the K-times expanded tensor.
// knn_build   : group_knn (pointnet2_utils.py:487-514):
//               G[b,i,k,:] = [ feats_y[b, idx, :] | d2 | w | nn_abs | nn_rel | x ],
//               w = (1/(d2+1e-8)) / sum_k (1/(d2_k+1e-8))   (SQUARED distances, as the reference).
// attention_pool : tail of AttentionModule.forward (attention.py:83-96): count mask (-1e9),
//               softmax over the K neighbours, value = relu(GN(conv(h))) folded as scale/shift,
//               weighted sum -> (B*npoint, D).
// All tensors channel-LAST: a position's channels are contiguous, so neighbour feature rows are
// read as whole contiguous segments and every store is coalesced.
#include "pdr_common.h"

namespace {

// LPP lanes per position p = (b, j, k) (16 / 32 / 64 by row width); the lanes of a group stride over
// the output channels, so the neighbour's feature row is read as one contiguous segment and the
// output row is written contiguously; index / count are loaded once per group.
template <int LPP>
__global__ __launch_bounds__(256) void group_build_kernel(
    const float* __restrict__ feats, int Cs, int n, const float* __restrict__ xyz,
    const float* __restrict__ new_xyz, const int* __restrict__ idx, const int* __restrict__ counts,
    int m, int K, int patch_empty, int with_abs, int with_centre, long npos, int Cout, int ldo,
    float* __restrict__ out) {
  constexpr int GPB = 256 / LPP;           // position groups per workgroup
  const int lane = threadIdx.x % LPP;
  const long g0 = static_cast<long>(blockIdx.x) * GPB + threadIdx.x / LPP;
  const long ngroups = static_cast<long>(gridDim.x) * GPB;
  for (long p = g0; p < npos; p += ngroups) {
    const long bj = p / K;                 // b * m + j
    const int b = static_cast<int>(bj / m);
    const int a = idx[p];
    const bool empty = patch_empty && counts[bj] <= 0;
    const float* frow = feats + (static_cast<long>(b) * n + a) * Cs;
    float* orow = out + p * ldo;
    for (int c = lane; c < Cs; c += LPP) orow[c] = empty ? 0.0f : frow[c];
    if (lane < ldo - Cout) orow[Cout + lane] = 0.0f;   // padding columns
    if (lane < Cout - Cs) {
      const int g = lane;                  // 0..2 rel, 3..5 abs|centre, 6..8 centre
      const int d = g % 3;
      const float ctr = new_xyz[bj * 3 + d];
      const float ab = empty ? ctr : xyz[(static_cast<long>(b) * n + a) * 3 + d];
      const int kind = g / 3;
      float v;
      if (kind == 0) v = ab - ctr;
      else if (kind == 1) v = with_abs ? ab : ctr;
      else v = ctr;
      orow[Cs + g] = v;
    }
  }
}

template <int LPP>
__global__ __launch_bounds__(256) void knn_build_kernel(
    const float* __restrict__ feats_y, int C, int n2, const float* __restrict__ x,
    const float* __restrict__ y, const long long* __restrict__ idx, const float* __restrict__ d2,
    int n1, int K, long npos, int Cout, int ldo, float* __restrict__ out) {
  constexpr int GPB = 256 / LPP;
  const int lane = threadIdx.x % LPP;
  const long g0 = static_cast<long>(blockIdx.x) * GPB + threadIdx.x / LPP;
  const long ngroups = static_cast<long>(gridDim.x) * GPB;
  for (long p = g0; p < npos; p += ngroups) {
    const long bi = p / K;                 // (b, i)
    const int b = static_cast<int>(bi / n1);
    const long a = idx[p];
    const float* frow = feats_y + (static_cast<long>(b) * n2 + a) * C;
    float* orow = out + p * ldo;
    for (int c = lane; c < C; c += LPP) orow[c] = frow[c];
    if (lane >= 11 && lane < 11 + ldo - Cout) orow[Cout + lane - 11] = 0.0f;   // padding columns
    if (lane < 11) {
      float v;
      if (lane == 0) {
        v = d2[p];
      } else if (lane == 1) {
        float norm = 0.0f;
        for (int k = 0; k < K; ++k) norm += 1.0f / (d2[bi * K + k] + 1e-8f);
        v = (1.0f / (d2[p] + 1e-8f)) / norm;
      } else {
        const int g = lane - 2;            // 0..2 nn_abs, 3..5 nn_rel, 6..8 x
        const int d = g % 3;
        const float xq = x[bi * 3 + d];
        const float ab = y[(static_cast<long>(b) * n2 + a) * 3 + d];
        v = g < 3 ? ab : (g < 6 ? ab - xq : xq);
      }
      orow[C + lane] = v;
    }
  }
}

// One lane per (query row, 4 channels): scores and values are streamed ONCE as float4 with an
// online softmax (running max m, normaliser l, accumulator rescaled by exp(m_old - m_new)); the
// result equals softmax-then-sum up to fp32 rounding.
__global__ __launch_bounds__(256) void attention_pool_kernel(
    const float* __restrict__ scores, int lds, const float* __restrict__ values, int ldv,
    const float* __restrict__ vscale, const float* __restrict__ vshift, int v_relu,
    const int* __restrict__ counts, int K, int D, int npoint, long rows, float* __restrict__ out) {
  const int D4 = D >> 2;
  const long t = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= rows * D4) return;
  const long row = t / D4;
  const int d = static_cast<int>(t - row * D4) * 4;
  const int b = static_cast<int>(row / npoint);
  int cnt = K;
  if (counts) {
    cnt = counts[row];
    cnt = cnt < 1 ? 1 : cnt;             // attention.py:85 clamp(min=1)
  }
  float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
  if (vscale) sc = *reinterpret_cast<const float4*>(vscale + static_cast<long>(b) * D + d);
  if (vshift) sh = *reinterpret_cast<const float4*>(vshift + static_cast<long>(b) * D + d);
  const float lo = v_relu ? 0.0f : -__builtin_inff();
  const float* s = scores + row * K * lds + d;
  const float* v = values + row * K * ldv + d;
  float m[4], l[4], acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { m[j] = -__builtin_inff(); l[j] = 0.0f; acc[j] = 0.0f; }
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    float4 sk = *reinterpret_cast<const float4*>(s + static_cast<long>(k) * lds);
    const float4 vk = *reinterpret_cast<const float4*>(v + static_cast<long>(k) * ldv);
    if (k >= cnt) sk = make_float4(-1e9f, -1e9f, -1e9f, -1e9f);   // masked slots: exactly -1e9
    const float se[4] = {sk.x, sk.y, sk.z, sk.w};
    const float ve[4] = {vk.x, vk.y, vk.z, vk.w};
    const float sce[4] = {sc.x, sc.y, sc.z, sc.w}, she[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // softmax in the base-2 domain: exp(s - m) = exp2((s - m) log2 e) with the hardware v_exp_f32
      // (two instructions per exponential instead of the ~10 of expf: the kernel was VALU-, not
      // HBM-limited).  Arguments are <= 0, results in (0, 1]; the rounding of s * log2(e) perturbs
      // the weights by <= |s - m| * 2^-23 relative.
      const float s2 = se[j] * 1.44269504088896340736f;
      const float mn = fmaxf(m[j], s2);
      const float corr = __builtin_amdgcn_exp2f(m[j] - mn);   // exp2(-inf) = 0 on the first slot
      const float w = __builtin_amdgcn_exp2f(s2 - mn);
      const float val = fmaxf(__builtin_fmaf(ve[j], sce[j], she[j]), lo);
      l[j] = __builtin_fmaf(l[j], corr, w);
      acc[j] = __builtin_fmaf(acc[j], corr, val * w);
      m[j] = mn;
    }
  }
  *reinterpret_cast<float4*>(out + row * D + d) =
      make_float4(acc[0] / l[0], acc[1] / l[1], acc[2] / l[2], acc[3] / l[3]);
}

// Wave-per-query form for K * D / 4 <= 64 * NL float4 per query with D / 4 a power of two <= 64: a
// query's K x D block of scores (and of values) is CONTIGUOUS, so the wave reads it with NL fully
// coalesced 1-KiB loads each (lane -> (k = e / D4, c4 = e % D4), e = i * 64 + lane) instead of 8
// separate 128-byte pieces per instruction.  All scores sit in registers: exact two-pass softmax
// (max, then exp2 / sums), folded across the lanes that share a channel group with xor shuffles.
template <int NL>
__global__ __launch_bounds__(256) void attention_pool_wave_kernel(
    const float* __restrict__ scores, const float* __restrict__ values, const float* __restrict__ vscale,
    const float* __restrict__ vshift, int v_relu, const int* __restrict__ counts, int K, int D, int npoint,
    long rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int D4 = D >> 2;
  const int sh = __builtin_ctz(D4);
  const int c4 = lane & (D4 - 1);
  const int k0 = lane >> sh;                 // first neighbour of this lane
  const int kstep = 64 >> sh;                // neighbours covered per load
  const float lo = v_relu ? 0.0f : -__builtin_inff();
  const long nwaves = static_cast<long>(gridDim.x) * 4;
  for (long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const int b = static_cast<int>(row / npoint);
    int cnt = K;
    if (counts) {
      cnt = counts[row];
      cnt = cnt < 1 ? 1 : cnt;               // attention.py:85 clamp(min=1)
    }
    const float4* s4 = reinterpret_cast<const float4*>(scores + row * K * D) + lane;
    const float4* v4 = reinterpret_cast<const float4*>(values + row * K * D) + lane;
    float4 sv[NL], vv[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const bool ok = k0 + i * kstep < K;
      sv[i] = ok ? s4[i * 64] : make_float4(0, 0, 0, 0);
      vv[i] = ok ? v4[i * 64] : make_float4(0, 0, 0, 0);
    }
    float4 sc = make_float4(1, 1, 1, 1), shv = make_float4(0, 0, 0, 0);
    if (vscale) sc = *reinterpret_cast<const float4*>(vscale + static_cast<long>(b) * D + 4 * c4);
    if (vshift) shv = *reinterpret_cast<const float4*>(vshift + static_cast<long>(b) * D + 4 * c4);
    float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float se[NL][4];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int k = k0 + i * kstep;
      const float e[4] = {sv[i].x, sv[i].y, sv[i].z, sv[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // slots beyond K do not exist (-inf: weight 0); masked slots are exactly -1e9 as in the reference
        se[i][j] = k >= K ? -__builtin_inff() : (k >= cnt ? -1e9f : e[j]) * 1.44269504088896340736f;
        m[j] = fmaxf(m[j], se[i][j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      for (int off = D4; off < 64; off <<= 1) m[j] = fmaxf(m[j], __shfl_xor(m[j], off, 64));
    float l[4] = {0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
    const float sce[4] = {sc.x, sc.y, sc.z, sc.w}, she[4] = {shv.x, shv.y, shv.z, shv.w};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float ve[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float w = __builtin_amdgcn_exp2f(se[i][j] - m[j]);
        const float val = fmaxf(__builtin_fmaf(ve[j], sce[j], she[j]), lo);
        l[j] += w;
        acc[j] = __builtin_fmaf(val, w, acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      for (int off = D4; off < 64; off <<= 1) {
        l[j] += __shfl_xor(l[j], off, 64);
        acc[j] += __shfl_xor(acc[j], off, 64);
      }
    if (lane < D4)
      *reinterpret_cast<float4*>(out + row * D + 4 * c4) =
          make_float4(acc[0] / l[0], acc[1] / l[1], acc[2] / l[2], acc[3] / l[3]);
  }
}

// rows of a channel-last matrix: out[b, j, :] = src[b, idx[b,j], :]
__global__ __launch_bounds__(256) void gather_rows_cl_kernel(const float* __restrict__ src, int n,
                                                             int C, const int* __restrict__ idx,
                                                             int m, long total,
                                                             float* __restrict__ out) {
  const long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const long bj = e / C;
  const int c = static_cast<int>(e - bj * C);
  const int b = static_cast<int>(bj / m);
  out[e] = src[(static_cast<long>(b) * n + idx[bj]) * C + c];
}

// out[b, j, :] = [src0[b, idx[b,j], :] | src1[b, idx[b,j], :]]: gather of the rows of a concatenation that is never
// materialised (torch.cat + gather_rows as one launch)
__global__ __launch_bounds__(256) void gather_rows2_cl_kernel(const float* __restrict__ src0, int C0,
                                                              const float* __restrict__ src1, int C1, int n,
                                                              const int* __restrict__ idx, int m, long total,
                                                              float* __restrict__ out) {
  const long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int C = C0 + C1;
  const long bj = e / C;
  const int c = static_cast<int>(e - bj * C);
  const int b = static_cast<int>(bj / m);
  const long row = static_cast<long>(b) * n + idx[bj];
  out[e] = c < C0 ? src0[row * C0 + c] : src1[row * C1 + (c - C0)];
}

// rows of C floats -> rows of ldo >= C floats, zero-filled behind column C (F.pad as ONE launch: torch pads with a
// fill plus a strided copy)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int C, int ldo, long total,
                                                       float* __restrict__ out) {
  const long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const long r = e / ldo;
  const int c = static_cast<int>(e - r * ldo);
  out[e] = c < C ? src[r * C + c] : 0.0f;
}

inline unsigned blocks_for(long total) { return static_cast<unsigned>((total + 255) / 256); }

}  // namespace

extern "C" int pdr_group_build(const float* feats, int Cs, const float* xyz, const float* new_xyz,
                               const int* idx, const int* counts, int B, int n, int m, int K,
                               int patch_empty, int with_abs, int with_centre, float* out, int ldo,
                               pdr_stream_t stream) {
  if (B < 0 || n <= 0 || m < 0 || K <= 0 || Cs < 0) return PDR_EINVAL;
  if (B == 0 || m == 0) return PDR_OK;
  if (!xyz || !new_xyz || !idx || !out || (Cs > 0 && !feats) || (patch_empty && !counts))
    return PDR_EINVAL;
  const int Cout = Cs + 3 + (with_abs ? 3 : 0) + (with_centre ? 3 : 0);
  if (ldo < Cout || ldo - Cout > 8) return PDR_EINVAL;
  const long npos = static_cast<long>(B) * m * K;
  // channel order after the features: rel | (abs) | (centre); the kernel's `kind` 1 slot is abs
  // when with_abs else centre
  hipStream_t s = pdr::as_stream(stream);
#define PDR_GB(LPP)                                                                                 \
  do {                                                                                              \
    const long nblk = (npos + 256 / LPP - 1) / (256 / LPP);                                         \
    hipLaunchKernelGGL(group_build_kernel<LPP>, dim3(static_cast<unsigned>(nblk < 32768 ? nblk : 32768)), \
                       dim3(256), 0, s, feats, Cs, n, xyz, new_xyz, idx, counts, m, K, patch_empty, \
                       with_abs, with_centre, npos, Cout, ldo, out);                                \
  } while (0)
  if (Cout <= 16) PDR_GB(16);
  else if (Cout <= 48) PDR_GB(32);
  else PDR_GB(64);
#undef PDR_GB
  return pdr::check_launch();
}

extern "C" int pdr_knn_build(const float* feats_y, int C, const float* x, const float* y,
                             const long long* idx, const float* d2, int B, int n1, int n2, int K,
                             float* out, int ldo, pdr_stream_t stream) {
  if (B < 0 || n1 < 0 || n2 <= 0 || K <= 0 || C < 0) return PDR_EINVAL;
  if (B == 0 || n1 == 0) return PDR_OK;
  if (!x || !y || !idx || !d2 || !out || (C > 0 && !feats_y)) return PDR_EINVAL;
  const int Cout = C + 11;
  if (ldo < Cout || ldo - Cout > 8) return PDR_EINVAL;
  const long npos = static_cast<long>(B) * n1 * K;
  hipStream_t s = pdr::as_stream(stream);
#define PDR_KB(LPP)                                                                                 \
  do {                                                                                              \
    const long nblk = (npos + 256 / LPP - 1) / (256 / LPP);                                         \
    hipLaunchKernelGGL(knn_build_kernel<LPP>, dim3(static_cast<unsigned>(nblk < 32768 ? nblk : 32768)), \
                       dim3(256), 0, s, feats_y, C, n2, x, y, idx, d2, n1, K, npos, Cout, ldo, out); \
  } while (0)
  if (Cout <= 16) PDR_KB(16);
  else if (Cout <= 48) PDR_KB(32);
  else PDR_KB(64);
#undef PDR_KB
  return pdr::check_launch();
}

extern "C" int pdr_attention_pool(const float* scores, int lds, const float* values, int ldv,
                                  const float* vscale, const float* vshift, int v_relu,
                                  const int* counts, int B, int npoint, int K, int D, float* out,
                                  pdr_stream_t stream) {
  if (B < 0 || npoint < 0 || K <= 0 || D <= 0) return PDR_EINVAL;
  if (B == 0 || npoint == 0) return PDR_OK;
  if (!scores || !values || !out) return PDR_EINVAL;
  const long rows = static_cast<long>(B) * npoint;
  if (D % 4 || lds % 4 || ldv % 4) return PDR_EUNSUPPORTED;
  {
    // wave-per-query form: dense rows (ld == D), D / 4 a power of two <= 64, <= 8 loads per lane
    const int D4 = D / 4;
    const long per_query = static_cast<long>(K) * D4;
    const int kstep = D4 <= 64 ? 64 / (D4 ? D4 : 1) : 0;
    if (lds == D && ldv == D && D4 <= 64 && (D4 & (D4 - 1)) == 0 && per_query <= 512 && kstep > 0) {
      const int nl = static_cast<int>((K + kstep - 1) / kstep);
      long blocks = (rows + 3) / 4;
      if (blocks > 256L * 16) blocks = 256L * 16;
      const dim3 grid(static_cast<unsigned>(blocks));
      hipStream_t st = pdr::as_stream(stream);
#define PDR_AP(NL)                                                                                      \
  hipLaunchKernelGGL(attention_pool_wave_kernel<NL>, grid, dim3(256), 0, st, scores, values, vscale,    \
                     vshift, v_relu, counts, K, D, npoint, rows, out)
      if (nl <= 1) PDR_AP(1);
      else if (nl <= 2) PDR_AP(2);
      else if (nl <= 4) PDR_AP(4);
      else PDR_AP(8);
#undef PDR_AP
      return pdr::check_launch();
    }
  }
  hipLaunchKernelGGL(attention_pool_kernel, dim3(blocks_for(rows * (D / 4))), dim3(256), 0,
                     pdr::as_stream(stream), scores, lds, values, ldv, vscale, vshift, v_relu, counts, K,
                     D, npoint, rows, out);
  return pdr::check_launch();
}

extern "C" int pdr_gather_rows2(const float* src0, int C0, const float* src1, int C1, const int* idx, int B, int n,
                                int m, float* out, pdr_stream_t stream) {
  if (B < 0 || n <= 0 || C0 <= 0 || C1 <= 0 || m < 0) return PDR_EINVAL;
  if (B == 0 || m == 0) return PDR_OK;
  if (!src0 || !src1 || !idx || !out) return PDR_EINVAL;
  const long total = static_cast<long>(B) * m * (C0 + C1);
  hipLaunchKernelGGL(gather_rows2_cl_kernel, dim3(blocks_for(total)), dim3(256), 0, pdr::as_stream(stream), src0,
                     C0, src1, C1, n, idx, m, total, out);
  return pdr::check_launch();
}

extern "C" int pdr_pad_rows(const float* src, long rows, int C, float* out, int ldo, pdr_stream_t stream) {
  if (rows < 0 || C <= 0 || ldo < C) return PDR_EINVAL;
  if (rows == 0) return PDR_OK;
  if (!src || !out) return PDR_EINVAL;
  const long total = rows * ldo;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks_for(total)), dim3(256), 0, pdr::as_stream(stream), src, C, ldo,
                     total, out);
  return pdr::check_launch();
}

extern "C" int pdr_gather_rows(const float* src, const int* idx, int B, int n, int C, int m,
                               float* out, pdr_stream_t stream) {
  if (B < 0 || n <= 0 || C <= 0 || m < 0) return PDR_EINVAL;
  if (B == 0 || m == 0) return PDR_OK;
  if (!src || !idx || !out) return PDR_EINVAL;
  const long total = static_cast<long>(B) * m * C;
  hipLaunchKernelGGL(gather_rows_cl_kernel, dim3(blocks_for(total)), dim3(256), 0,
                     pdr::as_stream(stream), src, n, C, idx, m, total, out);
  return pdr::check_launch();
}

// ---------------------------------------------------------------------------------------------
// gather_add: first 1x1 conv of a grouped block WITHOUT the grouped tensor.
//
// The grouped input of QueryAndGroup / group_knn is a gather of per-point rows plus per-query
// terms, and the conv is linear, so for position p = (b, j, k) with neighbour a = idx[p]:
//     conv([feat[a] | xyz[a]-c_j | xyz[a] | c_j]) = U[b,a,:] + V[b,j,:]
//         U = [feat | xyz] . [W_f ; W_rel + W_abs]   (one row per SOURCE point,  n  rows)
//         V = c . (W_ctr - W_rel) + bias             (one row per QUERY,         m  rows)
//     kNN: conv([feat[a] | d2 | w | y[a] | y[a]-x_i | x_i]) = U[b,a,:] + V[b,i,:] + d2 r1 + w r2
// U and V are small GEMMs over n and m rows (fused_layer); this kernel produces the (m K)-row
// output with one gather + add per element and the GroupNorm moments of the result.  It removes
// 2 P Cin Cout flops (37 % of a reverse step's GEMM work) and the P x Cin grouped tensor.
// Empty balls (subset=False): the reference substitutes the query itself with a zero feature:
// Y = V0[b,j,:] = c . (W_abs + W_ctr) + bias.
//
// One workgroup = 128 positions; each wave owns 32 CONSECUTIVE positions whose neighbour indices /
// empty flags / per-position scalars are fetched with one coalesced load and then broadcast from
// registers.  A row is covered by LPR lanes x float4 (LPR = 16 / 32 / 64 by output width), so a wave
// instruction moves 64 / LPR rows and narrow outputs keep every lane busy; two row groups are in
// flight per iteration.  Every U / V / Y access is a contiguous row segment.
template <int LPR, bool KPOW2, bool HAS_S>
__global__ __launch_bounds__(256) void gather_add_kernel(
    const float* __restrict__ U, int ldu, int n_src, const float* __restrict__ V,
    const float* __restrict__ V0, int ldv, const int* __restrict__ idx, const int* __restrict__ counts,
    const float* __restrict__ s1, const float* __restrict__ r1, const float* __restrict__ s2,
    const float* __restrict__ r2, int rows_per_batch_, int K_, int Cout, float* __restrict__ Y_, int ldy_,
    float* __restrict__ partial, int relu_col0, int ycol0_, int ycol1_,
    const unsigned char* __restrict__ tile_valid, int partial_tpb, pdr::GatherTwin tw) {
  constexpr int TM = 128;
  // TWIN blocks (blockIdx.x >= tw.n_main; round 5): the same sum over the block's per-QUERY rows -- neighbour = the
  // query's first one (tw.idx0), K = 1 -- written whole to tw.Y, with the GroupNorm moments of the rows q >= wrow0[b]
  // (the queries of the cloud's skipped tiles) times tw.wmul in partial row b partial_tpb + tiles_per_batch + tile:
  // what a separate K = 1 launch + pdr_weighted_moments produced, in the launch that walks the tile subset.
  const bool twin = static_cast<int>(blockIdx.x) >= tw.n_main && tw.n_main > 0;   // uniform
  const int rows_per_batch = twin ? rows_per_batch_ / K_ : rows_per_batch_;
  const int K = twin ? 1 : K_;
  const int* __restrict__ idx_e = twin ? tw.idx0 : idx;
  float* __restrict__ Y = twin ? tw.Y : Y_;
  const int ldy = twin ? tw.ldy : ldy_;
  const int ycol0 = twin ? 0 : ycol0_, ycol1 = twin ? ((Cout + 3) & ~3) : ycol1_;
  constexpr int RPI = 64 / LPR;            // rows per wave instruction
  // row groups in flight per iteration: the kernel is bound by the latency of its L2 gathers, so every wave keeps
  // DEPTH x 2 independent 16-byte loads outstanding (4 measured against 2: see DESIGN.md)
  constexpr int DEPTH = 32 / RPI < 4 ? 32 / RPI : 4;
  constexpr int CW = 4 * LPR;              // columns covered per pass
  __shared__ float red[4][CW][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, cl = lane % LPR;
  const int tpb = (rows_per_batch + TM - 1) / TM;
  const int n_main = tw.n_main > 0 ? tw.n_main : static_cast<int>(gridDim.x);
  const int bid = twin ? static_cast<int>(blockIdx.x) - n_main : pdr::xcd_contiguous(blockIdx.x, n_main);
  const int b = bid / tpb, tb = bid - b * tpb;
  // a tile subset (pdr_dedup_plan): the other tiles are neither read nor written
  if (!twin && tile_valid && !tile_valid[bid]) return;   // uniform
  const int tpb_main = (rows_per_batch_ + TM - 1) / TM;
  // the tile's partial row (twin tiles behind the cloud's main tiles)
  const long prow = static_cast<long>(b) * (partial_tpb > 0 ? partial_tpb : tpb) + (twin ? tpb_main : 0) + tb;
  const long row0 = static_cast<long>(b) * rows_per_batch + static_cast<long>(tb) * TM;
  const int nvalid = min(TM, rows_per_batch - tb * TM);
  // statistics: rows >= wlo of the tile count (twin: the queries behind the cloud's valid tiles), times wmul
  const int wlo = twin ? min(max(tw.wrow0[b] - tb * TM, 0), TM) : 0;   // uniform
  const float wmul = twin ? tw.wmul : 1.0f;
  const float* Ub = U + static_cast<long>(b) * n_src * ldu;
  const int wr0 = wave * 32;
  const int myr = min(wr0 + (lane & 31), nvalid - 1);
  const long myp = row0 + myr;
  const int my_idx = idx_e[myp];
  const int my_empty = (counts && counts[myp / K] <= 0) ? 1 : 0;
  const float my_s1 = s1 ? s1[myp] : 0.0f;
  const float my_s2 = s2 ? s2[myp] : 0.0f;
  const int nrows = max(0, min(32, nvalid - wr0));   // uniform

  // The kernel is VALU-issue bound (PMC: 35 VALU instructions per 16-byte gather before this rewrite, waves
  // issue-stalled 47 % of their cycles), so the per-row work is kept minimal: the query row is an add + shift when
  // K is a power of two that divides the wave's 32 rows (every shipped config; the general form is a 64-bit
  // division per row group), the kNN terms and the empty-ball select exist only where their inputs do (uniform
  // branches), the ReLU of the statistics is one v_max against a per-lane bound, addresses are 32-bit offsets.
  constexpr bool has_s = HAS_S;                            // kNN terms d2 r1 + w r2 present
  const bool has_em = counts != nullptr;                  // uniform
  const int ksh = KPOW2 ? __builtin_ctz(K) : -1;          // KPOW2: K a power of two <= 32
  const long qbase = (row0 + wr0) / K;                     // exact when ksh >= 0 (row0 + wr0 is a multiple of K)
  const float* Vq = V + qbase * ldv;
  const long v0d = has_em ? V0 - V : 0;                    // elements from V to V0 (same address space)
  // (grid.y > 1, round 6: the column passes of a tile dealt to grid.y workgroups -- a launch of a few hundred tiles with
  // a wide output (the first conv of the 16- / 64-point levels: 128-512 tiles x 1,100 columns) was a serial walk of five
  // passes per wave on a half-empty chip)
  for (int c0 = static_cast<int>(blockIdx.y) * CW; c0 < Cout; c0 += CW * static_cast<int>(gridDim.y)) {
    const int c = c0 + 4 * cl;
    const bool cok = c < Cout;   // row widths are padded to a multiple of 4 in ldu / ldv / ldy
    const int cc = cok ? c : 0;
    float4 q1 = make_float4(0, 0, 0, 0), q2 = make_float4(0, 0, 0, 0);
    if (cok && r1) q1 = *reinterpret_cast<const float4*>(r1 + c);
    if (cok && r2) q2 = *reinterpret_cast<const float4*>(r2 + c);
    float lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] = (c + j >= relu_col0) ? 0.0f : -__builtin_inff();
    const bool ywin = Y && c >= ycol0 && c < ycol1;
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    for (int r = 0; r < nrows; r += DEPTH * RPI) {
      float4 u[DEPTH], v[DEPTH];
      float t1[DEPTH], t2[DEPTH];
      int em[DEPTH], rr[DEPTH];
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        rr[k] = r + k * RPI + sub;                               // this lane's row (per sub-group)
        const int rc = min(rr[k], nrows - 1);
        const int a = __shfl(my_idx, rc, 64);
        em[k] = has_em ? __shfl(my_empty, rc, 64) : 0;
        if (has_s) {
          t1[k] = __shfl(my_s1, rc, 64);
          t2[k] = __shfl(my_s2, rc, 64);
        }
        u[k] = *reinterpret_cast<const float4*>(Ub + static_cast<unsigned>(a * ldu + cc));
        const float* vp;
        if constexpr (KPOW2) vp = Vq + static_cast<unsigned>((rc >> ksh) * ldv + cc);
        else vp = V + ((row0 + wr0 + rc) / K) * ldv + cc;
        v[k] = *reinterpret_cast<const float4*>(vp + (em[k] ? v0d : 0));
      }
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        if (rr[k] < nrows && cok) {
          float4 y;
          if (em[k]) {
            y = v[k];
          } else {
            y = make_float4(u[k].x + v[k].x, u[k].y + v[k].y, u[k].z + v[k].z, u[k].w + v[k].w);
            if (has_s) {
              y.x = __builtin_fmaf(t1[k], q1.x, y.x); y.y = __builtin_fmaf(t1[k], q1.y, y.y);
              y.z = __builtin_fmaf(t1[k], q1.z, y.z); y.w = __builtin_fmaf(t1[k], q1.w, y.w);
              y.x = __builtin_fmaf(t2[k], q2.x, y.x); y.y = __builtin_fmaf(t2[k], q2.y, y.y);
              y.z = __builtin_fmaf(t2[k], q2.z, y.z); y.w = __builtin_fmaf(t2[k], q2.w, y.w);
            }
          }
          if (ywin) *reinterpret_cast<float4*>(Y + (row0 + wr0 + rr[k]) * ldy + (c - ycol0)) = y;
          const float e[4] = {y.x, y.y, y.z, y.w};
          if (wr0 + rr[k] >= wlo) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float f;
              asm("v_max_f32 %0, %1, %2" : "=v"(f) : "v"(e[j]), "v"(lo[j]));   // max(y, 0) or y (bound -inf)
              a1[j] += f;
              a2[j] = __builtin_fmaf(f, f, a2[j]);
            }
          }
        }
      }
    }
    if (partial) {
      // fold the RPI row sub-groups (lanes with equal `cl`), then the 4 waves through LDS
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
          a1[j] += __shfl_xor(a1[j], off, 64);
          a2[j] += __shfl_xor(a2[j], off, 64);
        }
      }
      if (sub == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[wave][4 * cl + j][0] = a1[j];
          red[wave][4 * cl + j][1] = a2[j];
        }
      }
      __syncthreads();
      const int cc2 = c0 + threadIdx.x;
      if (threadIdx.x < CW && cc2 < Cout) {
        const float t1 = (red[0][threadIdx.x][0] + red[1][threadIdx.x][0]) +
                         (red[2][threadIdx.x][0] + red[3][threadIdx.x][0]);
        const float t2 = (red[0][threadIdx.x][1] + red[1][threadIdx.x][1]) +
                         (red[2][threadIdx.x][1] + red[3][threadIdx.x][1]);
        float* o = partial + (prow * Cout + cc2) * 2;
        o[0] = t1 * wmul;      // (x 1.0f outside the twin tiles: exact)
        o[1] = t2 * wmul;
      }
      __syncthreads();
    }
  }
}

// Y (B*rows_per_batch, Cout; ld ldy) = U[b, idx[p]] + V[p / K] (+ s1[p] r1 + s2[p] r2), empty balls -> V0.
// U (B, n_src, ldu), V / V0 (B*rows_per_batch/K, ldv); all leading dimensions multiples of 4, 16-B
// aligned.  partial: NULL or (B * ceil(rows_per_batch / 128), Cout, 2) moments as in pdr_fused_layer.
static int gather_add_impl(const float* U, int ldu, int n_src, const float* V, const float* V0,
                           int ldv, const int* idx, const int* counts, const float* s1,
                           const float* r1, const float* s2, const float* r2, int B,
                           int rows_per_batch, int K, int Cout, float* Y, int ldy, float* partial,
                           int relu_col0, int ycol0, int ycols, const unsigned char* tile_valid, int partial_tpb,
                           pdr_stream_t stream, const int* idx0 = nullptr, float* Yd = nullptr, int ldyd = 0,
                           const int* wrow0 = nullptr, float wmul = 1.0f) {
  if (!U || !V || !idx || (!Y && !partial) || B < 0 || rows_per_batch <= 0 || K <= 0 || Cout <= 0 ||
      n_src <= 0)
    return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  if (rows_per_batch % K != 0 || (counts && !V0) || (s1 && !r1) || (s2 && !r2)) return PDR_EINVAL;
  const int c4 = (Cout + 3) & ~3;
  if (ycols < 0) ycols = Cout - ycol0;                     // -1: every column from ycol0 on
  // the written window [ycol0, ycol0 + ycols) starts on a float4 boundary; Y holds it from column 0
  if (Y && (ycol0 < 0 || ycol0 % 4 || ycols <= 0 || ycol0 + ycols > Cout)) return PDR_EINVAL;
  const int y4 = (ycols + 3) & ~3;
  if (ldu % 4 || ldv % 4 || ldu < c4 || ldv < c4 || (Y && (ldy % 4 || ldy < y4))) return PDR_EINVAL;
  // rows are moved as float4: every base pointer (possibly offset to a column sub-range) 16-B aligned
  auto al = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  if (!al(U) || !al(V) || (V0 && !al(V0)) || (Y && !al(Y)) || (r1 && !al(r1)) || (r2 && !al(r2)))
    return PDR_EINVAL;
  const int tpb = (rows_per_batch + 127) / 128;
  hipStream_t st = pdr::as_stream(stream);
  const bool kpow2 = (K & (K - 1)) == 0 && K <= 32;
  const bool has_s = s1 != nullptr || s2 != nullptr;
  pdr::GatherTwin tw{idx0, Yd, wrow0, ldyd, 0, wmul};
  long nblocks = static_cast<long>(B) * tpb;
  if (idx0) {
    // twin blocks: the per-query rows (rows_per_batch / K per cloud) behind the main tiles
    const int mq = rows_per_batch / K;
    if (!Yd || !wrow0 || !partial || has_s || ldyd % 4 || ldyd < c4 || !al(Yd) ||
        partial_tpb < tpb + (mq + 127) / 128)
      return PDR_EINVAL;
    tw.n_main = static_cast<int>(nblocks);
    nblocks += static_cast<long>(B) * ((mq + 127) / 128);
  }
  if (nblocks >= (1L << 31)) return PDR_EINVAL;
  // column passes per workgroup: all of them, unless the launch has fewer tiles than the chip holds workgroups
  const int cw = Cout <= 64 ? 64 : (Cout <= 128 ? 128 : 256);
  const int passes = (Cout + cw - 1) / cw;
  const int ncb = (nblocks <= 1024 && passes > 1) ? passes : 1;
  const dim3 grid(static_cast<unsigned>(nblocks), static_cast<unsigned>(ncb));
#define PDR_GA_K(LPR, KP, HS)                                                                          \
  hipLaunchKernelGGL((gather_add_kernel<LPR, KP, HS>), grid, dim3(256), 0, st, U, ldu, n_src, V, V0, ldv, idx,  \
                     counts, s1, r1, s2, r2, rows_per_batch, K, Cout, Y, ldy, partial, relu_col0, ycol0, \
                     ycol0 + y4, tile_valid, partial_tpb, tw)
#define PDR_GA(LPR)                                       \
  do {                                                    \
    if (kpow2 && has_s) PDR_GA_K(LPR, true, true);        \
    else if (kpow2) PDR_GA_K(LPR, true, false);           \
    else if (has_s) PDR_GA_K(LPR, false, true);           \
    else PDR_GA_K(LPR, false, false);                     \
  } while (0)
  if (Cout <= 64) PDR_GA(16);
  else if (Cout <= 128) PDR_GA(32);
  else PDR_GA(64);
#undef PDR_GA
#undef PDR_GA_K
  return pdr::check_launch();
}

extern "C" int pdr_gather_add(const float* U, int ldu, int n_src, const float* V, const float* V0,
                              int ldv, const int* idx, const int* counts, const float* s1,
                              const float* r1, const float* s2, const float* r2, int B,
                              int rows_per_batch, int K, int Cout, float* Y, int ldy, float* partial,
                              int relu_col0, int ycol0, int ycols, pdr_stream_t stream) {
  return gather_add_impl(U, ldu, n_src, V, V0, ldv, idx, counts, s1, r1, s2, r2, B, rows_per_batch, K, Cout, Y, ldy,
                         partial, relu_col0, ycol0, ycols, nullptr, 0, stream);
}

// pdr_gather_add over a SUBSET of its 128-row tiles: tile_valid (B * ceil(rows_per_batch / 128)) bytes from
// pdr_dedup_plan, tiles with a zero byte are neither read nor written; tile t of batch element b writes row
// b * partial_tpb + t of `partial` (partial_tpb >= tiles per batch element).
extern "C" int pdr_gather_add_tiles(const float* U, int ldu, int n_src, const float* V, const float* V0,
                                    int ldv, const int* idx, const int* counts, const float* s1,
                                    const float* r1, const float* s2, const float* r2, int B,
                                    int rows_per_batch, int K, int Cout, float* Y, int ldy, float* partial,
                                    int relu_col0, int ycol0, int ycols, const unsigned char* tile_valid,
                                    int partial_tpb, pdr_stream_t stream) {
  if (!tile_valid || partial_tpb < (rows_per_batch + 127) / 128) return PDR_EINVAL;
  return gather_add_impl(U, ldu, n_src, V, V0, ldv, idx, counts, s1, r1, s2, r2, B, rows_per_batch, K, Cout, Y, ldy,
                         partial, relu_col0, ycol0, ycols, tile_valid, partial_tpb, stream);
}

// pdr_gather_add_tiles + the block's per-QUERY rows in the same launch (round 5: was a second, K = 1 pdr_gather_add on
// the first neighbours followed by pdr_weighted_moments): Yd (B m, ldyd) <- U[b, idx0[q]] + V[q] (empty ball: V0[q]),
// every column; partial row b partial_tpb + tpb + j <- wmul x the moments of the rows q >= wrow0[b] of the cloud's
// j-th group of 128 queries (m = rows_per_batch / K queries per cloud).  Ball form only (no s1 / s2).
extern "C" int pdr_gather_add_tiles_twin(const float* U, int ldu, int n_src, const float* V, const float* V0,
                                         int ldv, const int* idx, const int* counts, int B, int rows_per_batch, int K,
                                         int Cout, float* Y, int ldy, float* partial, int relu_col0, int ycol0,
                                         int ycols, const unsigned char* tile_valid, int partial_tpb, const int* idx0,
                                         float* Yd, int ldyd, const int* wrow0, float wmul, pdr_stream_t stream) {
  if (!tile_valid || !idx0 || rows_per_batch <= 0 || K <= 0 || rows_per_batch % K != 0) return PDR_EINVAL;
  return gather_add_impl(U, ldu, n_src, V, V0, ldv, idx, counts, nullptr, nullptr, nullptr, nullptr, B, rows_per_batch,
                         K, Cout, Y, ldy, partial, relu_col0, ycol0, ycols, tile_valid, partial_tpb, stream, idx0, Yd,
                         ldyd, wrow0, wmul);
}

// ---------------------------------------------------------------------------------------------
// Neighbourhoods that are 32 copies of one row.
//
// ball_query pads a neighbourhood with its first hit; a query with at most ONE neighbour in its ball (count <= 1; an
// empty ball of a feature-transfer block is replaced by the query itself) therefore contributes K identical rows to
// every per-neighbour tensor of its block: the convs repeat one row K times, the attention pooling returns that row's
// value (one unmasked slot), the GroupNorm moments count it K times.  On x_t of a reverse process (noise for most of
// the trajectory) that is the rule, not the exception.  pdr_dedup_plan marks the 128-row tiles (128 / K queries)
// ALL of whose queries are such copies; the block's per-neighbour launches skip them (pdr_layer_in_t.tile_list,
// pdr_gather_add_tiles) and a K times smaller per-QUERY chain of the same layers supplies their moments
// (pdr_weighted_moments) and their pooled rows (pdr_patch_rows).  Same values as the full evaluation up to fp32
// summation order of the moments.
// (1) per tile, in parallel: the valid flag, its queries' weights and first neighbours
__global__ __launch_bounds__(256) void dedup_flags_kernel(const int* __restrict__ idx, const int* __restrict__ counts,
                                                          int K, long nq, int qpt, int* __restrict__ idx0,
                                                          float* __restrict__ row_w,
                                                          unsigned char* __restrict__ tile_valid) {
  const long q = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;   // 256 / qpt whole tiles per workgroup
  if (q >= nq) return;
  const int cnt = counts[q];
  idx0[q] = idx[q * K];
  // a tile's qpt (2 .. 16, a power of two) queries sit in consecutive lanes of one wave
  int v = cnt > 1 ? 1 : 0;
  for (int off = 1; off < qpt; off <<= 1) v |= __shfl_xor(v, off, 64);   // (every lane takes part in every exchange)
  const bool valid = v != 0;
  row_w[q] = valid ? 0.0f : static_cast<float>(K);
  if ((q & (qpt - 1)) == 0) tile_valid[q / qpt] = valid ? 1 : 0;
}

// (2) ONE workgroup: ordered compaction of the valid tile numbers (ballot prefix per wave, wave totals through LDS,
// chunks of 1024 tiles in ascending order)
__global__ __launch_bounds__(1024) void dedup_compact_kernel(const unsigned char* __restrict__ tile_valid, int ntiles,
                                                            int* __restrict__ tile_list, int* __restrict__ n_tiles) {
  __shared__ int wtot[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < ntiles; i0 += 1024) {
    const int i = i0 + tid;
    const bool valid = i < ntiles && tile_valid[i] != 0;
    const unsigned long long bal = __ballot(valid);
    const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __builtin_popcountll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      woff += w < wave ? wtot[w] : 0;
      tot += wtot[w];
    }
    const int base = base_s;
    if (valid) tile_list[base + woff + before] = i;
    __syncthreads();
    if (tid == 0) base_s = base + tot;
    __syncthreads();
  }
  if (tid == 0) *n_tiles = base_s;
}

// Stable partition of a cloud's queries: those with more than one neighbour first (in their original order), the
// one-point ones behind them.  perm[b][j] = original index of the query at sorted position j, inv = its inverse.
// One workgroup per cloud; chunks of 1024 queries, two passes (real neighbourhoods, then the rest).
__global__ __launch_bounds__(1024) void dedup_sort_kernel(const int* __restrict__ counts, int m,
                                                         int* __restrict__ perm, int* __restrict__ inv,
                                                         int* __restrict__ perm_rows) {
  __shared__ int wtot[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* cb = counts + static_cast<long>(blockIdx.x) * m;
  int* pb = perm + static_cast<long>(blockIdx.x) * m;
  int* ib = inv + static_cast<long>(blockIdx.x) * m;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    for (int i0 = 0; i0 < m; i0 += 1024) {
      const int i = i0 + tid;
      const bool take = i < m && ((cb[i] > 1) == (pass == 0));
      const unsigned long long bal = __ballot(take);
      const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wtot[wave] = __builtin_popcountll(bal);
      __syncthreads();
      int woff = 0, tot = 0;
      for (int w = 0; w < 16; ++w) {
        woff += w < wave ? wtot[w] : 0;
        tot += wtot[w];
      }
      const int base = base_s;
      if (take) {
        const int j = base + woff + before;
        pb[j] = i;
        ib[i] = j;
        if (perm_rows) perm_rows[static_cast<long>(blockIdx.x) * m + j] = static_cast<int>(blockIdx.x) * m + i;
      }
      __syncthreads();
      if (tid == 0) base_s = base + tot;
      __syncthreads();
    }
  }
}

// counts (B, m) -> perm, inv (B, m) int32: see dedup_sort_kernel.  A block evaluated on its queries in `perm` order
// (pdr_gather_rows of its per-query inputs) has its one-point neighbourhoods in whole tiles; pdr_gather_rows with
// `inv` puts its output back.
extern "C" int pdr_dedup_sort(const int* counts, int B, int m, int* perm, int* inv, int* perm_rows,
                              pdr_stream_t stream) {
  if (!counts || !perm || !inv || B < 0 || m <= 0 || static_cast<long>(B) * m >= (1L << 31)) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  hipLaunchKernelGGL(dedup_sort_kernel, dim3(B), dim3(1024), 0, pdr::as_stream(stream), counts, m, perm, inv, perm_rows);
  return pdr::check_launch();
}

// idx (B, m, K) int32 / counts (B, m) of a ball query ->
//   idx0 (B, m): the first neighbour of every query;  row_w (B, m) float: K for the queries of skipped tiles, else 0;
//   tile_valid (B * m K / 128) bytes, tile_list (same length, the valid tile numbers in ascending order), n_tiles (1).
// A tile = 128 rows = 128 / K queries is VALID (computed by the per-neighbour launches) when any of its queries has
// more than one neighbour.  K in {8, 16, 32}, m K a multiple of 128.
extern "C" int pdr_dedup_plan(const int* idx, const int* counts, int B, int m, int K, int* idx0, float* row_w,
                              unsigned char* tile_valid, int* tile_list, int* n_tiles, pdr_stream_t stream) {
  if (!idx || !counts || !idx0 || !row_w || !tile_valid || !tile_list || !n_tiles || B < 0 || m <= 0) return PDR_EINVAL;
  if (!(K == 8 || K == 16 || K == 32) || (static_cast<long>(m) * K) % 128 != 0) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  const int qpt = 128 / K;
  const long ntiles = static_cast<long>(B) * m / qpt;
  if (ntiles >= (1L << 30)) return PDR_EINVAL;
  const long nq = static_cast<long>(B) * m;
  hipLaunchKernelGGL(dedup_flags_kernel, dim3(blocks_for(nq)), dim3(256), 0, pdr::as_stream(stream), idx, counts, K, nq,
                     qpt, idx0, row_w, tile_valid);
  hipLaunchKernelGGL(dedup_compact_kernel, dim3(1), dim3(1024), 0, pdr::as_stream(stream), tile_valid,
                     static_cast<int>(ntiles), tile_list, n_tiles);
  return pdr::check_launch();
}

// ---- sort + gathers + plan in ONE launch (round 5) ------------------------------------------------------------------
// pdr_dedup_sort, the three pdr_gather_rows of the sorted index rows / counts / query coordinates and the two kernels
// of pdr_dedup_plan were six dependent launches behind every ball query of the x_t branch -- at the head of a step they
// sit between the first ball query and the first block (0.13 -> 0.38 ms in profiles/r4_timeline_markers.json).  With
// the queries SORTED a cloud's valid tiles are simply its first nv = ceil(real / (128 / K)) tiles, so the whole plan is
// a count + a stable partition per cloud: one 1024-thread workgroup per cloud does all of it.
//   (1) real neighbourhoods (count > 1) of clouds 0 .. b -> this cloud's nv and the offset of its tiles in the list
//       (every workgroup recounts its predecessors: B m int loads, L2 hits -- no inter-workgroup communication);
//   (2) the stable partition of pdr_dedup_sort (perm / inv / perm_rows);
//   (3) rows gathered into that order: index rows (K ints = 16-byte pieces), counts, coordinates, first neighbours,
//       weights (K behind the cloud's valid tiles, else 0);
//   (4) tile flags, the ascending tile list, per-cloud [nv | first weighted query], the probe counters.
constexpr int kMaxPrepareClouds = 1024;
constexpr int kMaxPrepareQueries = 4096;   // a cloud's permutation lives in LDS (16 KB)

__global__ __launch_bounds__(1024) void dedup_prepare_kernel(
    const int* __restrict__ idx, const int* __restrict__ counts, const float* __restrict__ xyz, int m, int K, int nB,
    int* __restrict__ perm, int* __restrict__ inv, int* __restrict__ perm_rows, int* __restrict__ idx_s,
    int* __restrict__ counts_s, float* __restrict__ xyz_s, int* __restrict__ idx0, float* __restrict__ row_w,
    unsigned char* __restrict__ tile_valid, int* __restrict__ tile_list, int* __restrict__ n_tiles,
    int* __restrict__ nvalid, int* __restrict__ probe_acc) {
  __shared__ int nreal_s[kMaxPrepareClouds];
  __shared__ int perm_s[kMaxPrepareQueries];
  __shared__ int wtot[16];
  __shared__ int base_s, prefix_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int qpt = 128 / K, tpb = m / qpt;
  if (tid == 0) base_s = 0;
  // (1) one WAVE per cloud 0 .. b (waves take clouds wave, wave + 16, ...): every lane's loads of a cloud are
  // independent and issued together -- a 16-byte load per 4 counts where the rows allow it
  const bool vec = (m & 3) == 0 && (reinterpret_cast<uintptr_t>(counts) & 15) == 0;   // uniform
  for (int bb = wave; bb <= b; bb += 16) {
    const int* cb = counts + static_cast<long>(bb) * m;
    int c = 0;
    if (vec) {
      const int4* c4 = reinterpret_cast<const int4*>(cb);
      const int n4 = m >> 2;
      for (int i0 = 0; i0 < n4; i0 += 64 * 8) {
        int4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 64 * u + lane;
          v[u] = c4[i < n4 ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = i0 + 64 * u + lane < n4;
          c += ok ? (v[u].x > 1) + (v[u].y > 1) + (v[u].z > 1) + (v[u].w > 1) : 0;
        }
      }
    } else {
      for (int i = lane; i < m; i += 64) c += cb[i] > 1 ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) nreal_s[bb] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int p = 0;
    for (int bb = 0; bb < b; ++bb) p += (nreal_s[bb] + qpt - 1) / qpt;
    prefix_s = p;
  }
  const int nv = (nreal_s[b] + qpt - 1) / qpt;       // valid tiles of this cloud: its first nv
  const int q0 = nv * qpt;                            // first query of the skipped tiles
  // (2) stable partition: real neighbourhoods first
  const int* cb = counts + static_cast<long>(b) * m;
  int* pb = perm + static_cast<long>(b) * m;
  int* ib = inv + static_cast<long>(b) * m;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i0 = 0; i0 < m; i0 += 1024) {
      const int i = i0 + tid;
      const bool take = i < m && ((cb[i] > 1) == (pass == 0));
      const unsigned long long bal = __ballot(take);
      const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wtot[wave] = __builtin_popcountll(bal);
      __syncthreads();
      int woff = 0, tot = 0;
      for (int w = 0; w < 16; ++w) {
        woff += w < wave ? wtot[w] : 0;
        tot += wtot[w];
      }
      const int base = base_s;
      if (take) {
        const int j = base + woff + before;
        perm_s[j] = i;
        pb[j] = i;
        ib[i] = j;
        if (perm_rows) perm_rows[static_cast<long>(b) * m + j] = b * m + i;
      }
      __syncthreads();
      if (tid == 0) base_s = base + tot;
      __syncthreads();
    }
  }
  // (3) per-query rows in sorted order (the permutation comes from LDS: no global round trip in front of every row)
  for (int j = tid; j < m; j += 1024) {
    const int src = perm_s[j];
    const long qs = static_cast<long>(b) * m + src, qd = static_cast<long>(b) * m + j;
    counts_s[qd] = cb[src];
    idx0[qd] = idx[qs * K];
    row_w[qd] = j >= q0 ? static_cast<float>(K) : 0.0f;
    if (xyz) {
      const float x = xyz[qs * 3 + 0], y = xyz[qs * 3 + 1], z = xyz[qs * 3 + 2];
      xyz_s[qd * 3 + 0] = x;
      xyz_s[qd * 3 + 1] = y;
      xyz_s[qd * 3 + 2] = z;
    }
  }
  // index rows as 16-byte pieces, four in flight per thread
  const int k4 = K / 4, ksh4 = __builtin_ctz(k4);     // (K in {8, 16, 32}: 2, 4 or 8 pieces per row)
  const int npiece = m * k4;
  const int4* src4 = reinterpret_cast<const int4*>(idx + static_cast<long>(b) * m * K);
  int4* dst4 = reinterpret_cast<int4*>(idx_s + static_cast<long>(b) * m * K);
  for (int e0 = 0; e0 < npiece; e0 += 4096) {
    int4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + 1024 * u + tid, npiece - 1);
      v[u] = src4[(perm_s[e >> ksh4] << ksh4) + (e & (k4 - 1))];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 1024 * u + tid;
      if (e < npiece) dst4[e] = v[u];
    }
  }
  // (4) tiles
  const int prefix = prefix_s;
  for (int t = tid; t < tpb; t += 1024) {
    tile_valid[static_cast<long>(b) * tpb + t] = t < nv ? 1 : 0;
    if (t < nv) tile_list[prefix + t] = b * tpb + t;
  }
  if (tid == 0) {
    nvalid[b] = nv;
    nvalid[nB + b] = q0;
    if (b == nB - 1) *n_tiles = prefix + nv;
    if (probe_acc) {
      atomicAdd(&probe_acc[0], nv);
      atomicAdd(&probe_acc[1], tpb);
    }
  }
}

// The same count without the plan (the step with every neighbourhood evaluated carries it so that the sampler can tell
// when the deduplicated step would be the faster one again): probe_acc[0] += tiles a plan would walk, [1] += tiles.
__global__ __launch_bounds__(256) void dedup_probe_kernel(const int* __restrict__ counts, int m, int K,
                                                          int* __restrict__ probe_acc) {
  __shared__ int tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  const int* cb = counts + static_cast<long>(blockIdx.x) * m;
  int c = 0;
  for (int i = threadIdx.x; i < m; i += 256) c += cb[i] > 1 ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&tot, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int qpt = 128 / K;
    atomicAdd(&probe_acc[0], (tot + qpt - 1) / qpt);
    atomicAdd(&probe_acc[1], m / qpt);
  }
}

// idx (B, m, K) int32 / counts (B, m) of a ball query, xyz (B, m, 3) query coordinates (may be NULL) -> everything a
// grouped block needs to evaluate its one-point neighbourhoods once, in ONE launch (= pdr_dedup_sort + pdr_gather_rows
// of idx / counts / xyz + pdr_dedup_plan on the sorted arrays, same values):
//   perm, inv, perm_rows (B, m): the stable partition (real neighbourhoods first) as in pdr_dedup_sort;
//   idx_s (B, m, K), counts_s (B, m), xyz_s (B, m, 3): the inputs in that order;
//   idx0, row_w (B, m), tile_valid (B m K / 128), tile_list, n_tiles: as pdr_dedup_plan on the sorted arrays;
//   nvalid (2 B ints): [b] = valid tiles of cloud b (its FIRST nv tiles), [B + b] = nv * (128 / K) = the first query of
//   its skipped tiles (pdr_layer_in_t.wrow0 of the per-query launches);
//   probe_acc (NULL or 2 ints): [0] += sum_b nv, [1] += B m K / 128.
extern "C" int pdr_dedup_prepare(const int* idx, const int* counts, const float* xyz, int B, int m, int K, int* perm,
                                 int* inv, int* perm_rows, int* idx_s, int* counts_s, float* xyz_s, int* idx0,
                                 float* row_w, unsigned char* tile_valid, int* tile_list, int* n_tiles, int* nvalid,
                                 int* probe_acc, pdr_stream_t stream) {
  if (!idx || !counts || !perm || !inv || !idx_s || !counts_s || (xyz && !xyz_s) || !idx0 || !row_w || !tile_valid ||
      !tile_list || !n_tiles || !nvalid || B < 0 || m <= 0)
    return PDR_EINVAL;
  if (!(K == 8 || K == 16 || K == 32) || (static_cast<long>(m) * K) % 128 != 0) return PDR_EINVAL;
  if (static_cast<long>(B) * m * K >= (1L << 31)) return PDR_EINVAL;
  if (B > kMaxPrepareClouds || m > kMaxPrepareQueries) return PDR_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(idx_s)) % 16 != 0) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  hipLaunchKernelGGL(dedup_prepare_kernel, dim3(B), dim3(1024), 0, pdr::as_stream(stream), idx, counts, xyz, m, K, B,
                     perm, inv, perm_rows, idx_s, counts_s, xyz_s, idx0, row_w, tile_valid, tile_list, n_tiles, nvalid,
                     probe_acc);
  return pdr::check_launch();
}

extern "C" int pdr_dedup_probe(const int* counts, int B, int m, int K, int* probe_acc, pdr_stream_t stream) {
  if (!counts || !probe_acc || B < 0 || m <= 0) return PDR_EINVAL;
  if (!(K == 8 || K == 16 || K == 32) || (static_cast<long>(m) * K) % 128 != 0) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  hipLaunchKernelGGL(dedup_probe_kernel, dim3(B), dim3(256), 0, pdr::as_stream(stream), counts, m, K, probe_acc);
  return pdr::check_launch();
}

// Per-tile GroupNorm moments of a materialised (B rpb, C) tensor with one WEIGHT per row, written behind the
// moments of a tile subset: block (j, cy) handles rows [128 j, 128 j + 128) of batch element b = j / tpbd and 64
// columns; the trailing blocks zero the partial rows of the tiles the subset skipped.
__global__ __launch_bounds__(256) void weighted_moments_kernel(const float* __restrict__ Y, int ldy, int C, int rpb,
                                                               int tpbd, int nB, int relu_col0,
                                                               const float* __restrict__ row_w,
                                                               float* __restrict__ partial, int ptpb, int tpb_full,
                                                               const unsigned char* __restrict__ tile_valid) {
  const int nmom = nB * tpbd;
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  if (static_cast<int>(blockIdx.x) >= nmom) {
    // zero the rows of skipped tiles: 16 tiles per block, this block's 64 columns
    const int t0 = (static_cast<int>(blockIdx.x) - nmom) * 16;
    for (int k = sl; k < 16; k += 4) {
      const int t = t0 + k;
      if (t < nB * tpb_full && !tile_valid[t] && c < C) {
        const int b = t / tpb_full, tb = t - b * tpb_full;
        float* o = partial + ((static_cast<long>(b) * ptpb + tb) * C + c) * 2;
        o[0] = 0.0f;
        o[1] = 0.0f;
      }
    }
    return;
  }
  __shared__ float red[4][64][2];
  const int b = blockIdx.x / tpbd, j = blockIdx.x - b * tpbd;
  const int r0 = j * 128 + sl * 32, r1 = min(r0 + 32, rpb);
  float s1 = 0.0f, s2 = 0.0f;
  if (c < C) {
    const float lo = c >= relu_col0 ? 0.0f : -__builtin_inff();
    const float* yp = Y + (static_cast<long>(b) * rpb) * ldy + c;
    const float* wp = row_w + static_cast<long>(b) * rpb;
    // a launch of a few microseconds: every load of a thread in flight at once (a row loop of dependent-looking loads
    // took 12 us), and no row of Y is read for a slice whose weights are all zero (sorted queries: the rule)
    float w[32];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      w[k] = r0 + k < r1 ? wp[min(r0 + k, rpb - 1)] : 0.0f;
      any = any || w[k] > 0.0f;
    }
    if (any) {
#pragma unroll
      for (int k0 = 0; k0 < 32; k0 += 16) {
        float y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = yp[static_cast<long>(min(r0 + k0 + k, rpb - 1)) * ldy];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float f = fmaxf(y[k], lo);
          s1 = __builtin_fmaf(w[k0 + k], f, s1);
          s2 = __builtin_fmaf(w[k0 + k] * f, f, s2);
        }
      }
    }
  }
  red[sl][cl][0] = s1;
  red[sl][cl][1] = s2;
  __syncthreads();
  if (sl == 0 && c < C) {
    float* o = partial + ((static_cast<long>(b) * ptpb + tpb_full + j) * C + c) * 2;
    o[0] = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
    o[1] = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
  }
}

// partial (B * ptpb, C, 2): rows [b ptpb + tpb_full + j] (j < ceil(rpb / 128)) <- sum_r w[r] f, sum_r w[r] f^2 over the
// rows of tile j of Y (B rpb, C; ld ldy), f = y (columns >= relu_col0: max(y, 0)); rows [b ptpb + t] of the tiles
// t < tpb_full with tile_valid[b tpb_full + t] == 0 <- 0.  ptpb >= tpb_full + ceil(rpb / 128).
extern "C" int pdr_weighted_moments(const float* Y, int ldy, int B, int rpb, int C, int relu_col0, const float* row_w,
                                    float* partial, int ptpb, int tpb_full, const unsigned char* tile_valid,
                                    pdr_stream_t stream) {
  if (!Y || !row_w || !partial || !tile_valid || B < 0 || rpb <= 0 || C <= 0 || ldy < C || tpb_full <= 0) return PDR_EINVAL;
  const int tpbd = (rpb + 127) / 128;
  if (ptpb < tpb_full + tpbd) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  const long nz = (static_cast<long>(B) * tpb_full + 15) / 16;
  const dim3 grid(static_cast<unsigned>(static_cast<long>(B) * tpbd + nz), static_cast<unsigned>((C + 63) / 64));
  hipLaunchKernelGGL(weighted_moments_kernel, grid, dim3(256), 0, pdr::as_stream(stream), Y, ldy, C, rpb, tpbd, B,
                     relu_col0, row_w, partial, ptpb, tpb_full, tile_valid);
  return pdr::check_launch();
}

// out[q, :D] = act(V[q, :D] * vscale[b] + vshift[b]) for the rows q with row_w[q] > 0 (the pooled output of a query
// whose neighbourhood is K copies of one row is that row's activated value); other rows untouched.
__global__ __launch_bounds__(256) void patch_rows_kernel(const float* __restrict__ V, int ldv,
                                                         const float* __restrict__ vscale,
                                                         const float* __restrict__ vshift, int v_relu,
                                                         const float* __restrict__ row_w, int rpb, int D, long total,
                                                         float* __restrict__ out, int ldo,
                                                         const int* __restrict__ out_rows) {
  const long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const long q = e / D;
  const int d = static_cast<int>(e - q * D);
  if (!(row_w[q] > 0.0f)) return;
  const long b = q / rpb;
  float v = V[q * ldv + d];
  const float s = vscale ? vscale[b * D + d] : 1.0f;
  const float h = vshift ? vshift[b * D + d] : 0.0f;
  v = __builtin_fmaf(v, s, h);
  if (v_relu) v = fmaxf(v, 0.0f);
  out[(out_rows ? static_cast<long>(out_rows[q]) : q) * ldo + d] = v;
}

extern "C" int pdr_patch_rows(const float* V, int ldv, const float* vscale, const float* vshift, int v_relu,
                              const float* row_w, int B, int rpb, int D, float* out, int ldo, const int* out_rows,
                              pdr_stream_t stream) {
  if (!V || !row_w || !out || B < 0 || rpb <= 0 || D <= 0 || ldv < D || ldo < D) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  const long total = static_cast<long>(B) * rpb * D;
  hipLaunchKernelGGL(patch_rows_kernel, dim3(blocks_for(total)), dim3(256), 0, pdr::as_stream(stream), V, ldv, vscale,
                     vshift, v_relu, row_w, rpb, D, total, out, ldo, out_rows);
  return pdr::check_launch();
}
