// embed.hip -- the step-embedding chain of one reverse step as three small launches.
//
// Reference (pointnet2_ssg_sem.py:14-31 `calc_t_emb`, pointnet2_with_pcld_condition.py:183-184): sinusoidal embedding
// of the B step values -> Linear(t_dim, 4 t_dim) -> swish -> Linear(4 t_dim, 4 t_dim) -> swish, then every block's
// fc(t_emb) Linear (pointnet2_modules.py:113-120).  In PyTorch that is sin, cos, cat, mul, three hipBLASLt GEMMs
// with M = B = 32, two sigmoids and two muls: ~12 dependent launches at the very start of a step -- measured
// (tools/lab/step_markers.py, untraced graph replay) 190-300 us before the first layer kernel of the step can start,
// twice what the first ball query beside them takes.
//
// One kernel, three launches: out[b, n] = act(bias[n] + sum_k in[b, k] W[n, k]) with W the nn.Linear weight (N, K).
// A workgroup owns a 32 (b) x 32 (n) output tile; its four waves split K, each accumulating a 32 x 32 partial with
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation); lane (row / column = lane & 31, half = lane >> 5)
// loads 4 consecutive k of its row of `in` and of its row of `W` as one 16-byte load each and feeds 4 MFMAs (MFMA e
// contracts k0 + 4 half + e: any pairing is valid as long as both operands use the same one).  The partials are
// summed through LDS in wave order (deterministic).  PROLOGUE: in[b, k] = sin / cos(ts[b] freq[k mod half]) computed
// on the fly from the step values, with the frequency table the reference computes on the CPU (same f32 values).
#include "pdr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float swish_f32(float x) { return x * (1.0f / (1.0f + expf(-x))); }

template <bool SINCOS>
__global__ __launch_bounds__(256) void embed_linear_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ ts, int ts_stride,
                                                           const float* __restrict__ freq, int half,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias, int B, int K, int N,
                                                           int act, float* __restrict__ out, int ldo) {
  __shared__ float part[4][32][33];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int il = lane & 31, hi = lane >> 5;
  const int n0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int b = b0 + il, n = n0 + il;
  const bool bok = b < B, nok = n < N;
  const float tsb = (SINCOS && bok) ? ts[static_cast<long>(b) * ts_stride] : 0.0f;
  const float* xr = x + static_cast<long>(bok ? b : 0) * ldx;
  const float* wr = W + static_cast<long>(nok ? n : 0) * K;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // k blocks of 8: wave w takes blocks w, w + 4, ...; four blocks per trip with all eight 16-byte loads issued
  // before the first MFMA (a trip per block was a chain of dependent ~2 us global loads: 16 trips for K = 512)
  for (int kb0 = wave; kb0 * 8 < K; kb0 += 16) {
    float a[4][4];
    float4 w4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (kb0 + 4 * u) * 8 + 4 * hi;
      const bool kok = (kb0 + 4 * u) * 8 < K;          // uniform
      if (SINCOS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = kok ? k + e : 0;
          // (torch evaluates ts * freq in f32, then sin / cos of the f32 product)
          const float arg = tsb * freq[kk < half ? kk : kk - half];
          a[u][e] = kok ? (kk < half ? sinf(arg) : cosf(arg)) : 0.0f;
        }
      } else {
        const float4 v = kok ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        a[u][0] = v.x, a[u][1] = v.y, a[u][2] = v.z, a[u][3] = v.w;
      }
      w4[u] = kok ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float av = bok ? a[u][e] : 0.0f, wv = nok ? w[e] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc, 0, 0, 0);
      }
    }
  }
  // C / D layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * hi][il] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int row = e >> 5, col = e & 31;
    if (b0 + row < B && n0 + col < N) {
      float v = part[0][row][col];
      v += part[1][row][col];
      v += part[2][row][col];
      v += part[3][row][col];
      v += bias ? bias[n0 + col] : 0.0f;
      if (act == 1) v = swish_f32(v);
      out[static_cast<long>(b0 + row) * ldo + n0 + col] = v;
    }
  }
}

// out[b, :W] = table[clamp(*t, 0, T - 1), :W] for every b < B (16-byte pieces)
__global__ __launch_bounds__(256) void embed_select_kernel(const float* __restrict__ table, int ldt, int T,
                                                           const long long* __restrict__ t_dev, int B, int W4,
                                                           float* __restrict__ out, int ldo) {
  long long t = *t_dev;
  t = t < 0 ? 0 : (t >= T ? T - 1 : t);
  const float4* row = reinterpret_cast<const float4*>(table + t * ldt);
  const int total = B * W4;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int b = e / W4, c = e - b * W4;
    *reinterpret_cast<float4*>(out + static_cast<long>(b) * ldo + 4 * c) = row[c];
  }
}

}  // namespace

extern "C" int pdr_embed_select(const float* table, int ldt, int T, const long long* t_dev, int B, int W, float* out,
                                int ldo, pdr_stream_t stream) {
  if (!table || !t_dev || !out || T <= 0 || B < 0 || W <= 0 || ldt < W || ldo < W) return PDR_EINVAL;
  if (W % 4 || ldt % 4 || ldo % 4 || (reinterpret_cast<uintptr_t>(table) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return PDR_EUNSUPPORTED;
  if (B == 0) return PDR_OK;
  const long total = static_cast<long>(B) * (W / 4);
  if (total >= (1L << 31)) return PDR_EINVAL;
  const unsigned blocks = static_cast<unsigned>((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(embed_select_kernel, dim3(blocks), dim3(256), 0, pdr::as_stream(stream), table, ldt, T, t_dev, B,
                     W / 4, out, ldo);
  return pdr::check_launch();
}

extern "C" int pdr_embed_linear(const float* x, int ldx, const float* ts, int ts_stride, const float* freq,
                                int half, const float* W, const float* bias, int B, int K, int N, int act,
                                float* out, int ldo, pdr_stream_t stream) {
  if (B < 0 || K <= 0 || N <= 0 || !W || !out || ldo < N || (act != 0 && act != 1)) return PDR_EINVAL;
  if (ts ? (!freq || half <= 0 || K != 2 * half || ts_stride < 0) : (!x || ldx < K)) return PDR_EINVAL;
  if (B == 0) return PDR_OK;
  // 16-byte operand loads: K a multiple of 8, rows 16-byte aligned
  if (K % 8 != 0 || (reinterpret_cast<uintptr_t>(W) & 15) || (!ts && ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15))))
    return PDR_EUNSUPPORTED;
  const dim3 grid((N + 31) / 32, (B + 31) / 32);
  hipStream_t s = pdr::as_stream(stream);
  if (ts)
    hipLaunchKernelGGL(embed_linear_kernel<true>, grid, dim3(256), 0, s, x, ldx, ts, ts_stride, freq, half, W, bias,
                       B, K, N, act, out, ldo);
  else
    hipLaunchKernelGGL(embed_linear_kernel<false>, grid, dim3(256), 0, s, x, ldx, ts, ts_stride, freq, half, W, bias,
                       B, K, N, act, out, ldo);
  return pdr::check_launch();
}
