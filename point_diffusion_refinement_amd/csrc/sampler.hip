// sampler.hip -- the elementwise tail of one reverse-diffusion step as ONE launch.
//
// DDPM (reference util.py:246-250):      x <- (x - c_eps[t] * eps) / sqrt_alpha[t] + sigma[t] * z
// FastDPM (util_fastdpmv2.py:186-204 `_ddim_update`, VAR / STEP loops :307-452):
//                                         x <- x * scale[t] + (c[t] * eps + sigma[t] * z)
// with the step index t read from DEVICE memory (a captured hipGraph replays the same launch for every step) and
// the per-step constants gathered from device tables.  PyTorch evaluates these expressions as 5 separate
// elementwise kernels after 3 index_selects; every intermediate is rounded to fp32.  This kernel performs the
// same operations in the same order (the library is built with -ffp-contract=off, IEEE division), so the result
// is bit-identical to the eager formula -- only the launches are gone.  eps may be a column window of a wider
// row (the network's last layer writes 4-float rows): `ld_eps` floats between consecutive points.
#include "pdr_common.h"

namespace {

template <int MODE>
__global__ __launch_bounds__(256) void reverse_update_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                             int ld_eps, const float* __restrict__ z,
                                                             const float* __restrict__ tab_a,
                                                             const float* __restrict__ tab_b,
                                                             const float* __restrict__ tab_c,
                                                             const long long* __restrict__ t_ptr, long npoints) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= npoints * 3) return;
  const long long t = *t_ptr;
  if (t < 0) return;   // a replay past the last step: leave x alone rather than index the tables at -1
  const float a = tab_a[t], b = tab_b[t], c = tab_c[t];
  const long p = i / 3;
  const int d = static_cast<int>(i - p * 3);
  const float e = eps[p * ld_eps + d];
  const float zz = z ? z[i] : 0.0f;
  const float xv = x[i];
  float r;
  if (MODE == 0) {
    // a = (1 - alpha_t) / sqrt(1 - alpha_bar_t), b = sqrt(alpha_t), c = sigma_t
    r = (xv - a * e) / b;
    r = r + c * zz;
  } else {
    // a = sqrt(alpha'/alpha), b = coefficient of eps, c = sigma
    r = xv * a;
    r = r + (b * e + c * zz);
  }
  x[i] = r;
}

// ---- the same update with the noise drawn in the kernel and the step bookkeeping folded in ------------------------
// A captured reverse step used to end with four launches: torch's Philox normal kernel (z, 786 KB written and read
// back), the update above, `t -= 1`, and at the start of the next step `ts = float(t)` (FastDPM: an index_select
// into the tau table).  Here ONE launch draws z (Philox4x32-10 keyed by a per-batch seed, counter = (element quad,
// draw number); Box-Muller on two uniform pairs -> four normals per thread), applies the reference expression in
// the reference's operation order, and the LAST workgroup to finish (device ticket) decrements the step counter,
// publishes the next step's network time input and advances the draw number.  z never touches memory.
// The stream of normals is this kernel's own (as `noise='device'` always was: same distribution, not torch's
// sequence); seed-parity with the reference is the `noise='cpu'` mode, which passes z explicitly.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = static_cast<unsigned>(p1 >> 32) ^ c1 ^ k0;
    const unsigned n2 = static_cast<unsigned>(p0 >> 32) ^ c3 ^ k1;
    c1 = static_cast<unsigned>(p1);
    c3 = static_cast<unsigned>(p0);
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

// (0, 1]: 24 bits + half an ulp (the top value rounds to 1.0f, whose log is 0 -- never 0.0f, whose log is -inf)
__device__ __forceinline__ float u01(unsigned v) {
  return static_cast<float>(v >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
}

template <int MODE>
__global__ __launch_bounds__(256) void reverse_step_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                           int ld_eps, const float* __restrict__ z,
                                                           const float* __restrict__ tab_a,
                                                           const float* __restrict__ tab_b,
                                                           const float* __restrict__ tab_c,
                                                           long long* __restrict__ t_ptr,
                                                           const float* __restrict__ ts_table,
                                                           float* __restrict__ ts_out,
                                                           unsigned long long* __restrict__ rng, int* __restrict__ ticket,
                                                           long total, int* __restrict__ probe_acc,
                                                           int* __restrict__ probe_out) {
  const long quad = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long long t = *t_ptr;
  // t < 0 is a replay past the last step: x is left alone (the tables are not indexed at -1), the ticket still runs
  const long long tt = t < 0 ? 0 : t;
  const float a = tab_a[tt], b = tab_b[tt], c = tab_c[tt];
  float zz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (t >= 0 && quad * 4 < total) {
    if (rng) {
      const unsigned long long seed = rng[0], draw = rng[1];
      unsigned r[4];
      philox4x32_10(static_cast<unsigned>(quad), static_cast<unsigned>(quad >> 32), static_cast<unsigned>(draw),
                    static_cast<unsigned>(draw >> 32), static_cast<unsigned>(seed), static_cast<unsigned>(seed >> 32),
                    r);
      const float r0 = sqrtf(-2.0f * logf(u01(r[0]))), r1 = sqrtf(-2.0f * logf(u01(r[2])));
      float s0, c0, s1, c1;
      sincosf(6.283185307179586f * u01(r[1]), &s0, &c0);
      sincosf(6.283185307179586f * u01(r[3]), &s1, &c1);
      zz[0] = r0 * c0, zz[1] = r0 * s0, zz[2] = r1 * c1, zz[3] = r1 * s1;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long i = quad * 4 + e;
      if (i < total) {
        const long p = i / 3;
        const int d = static_cast<int>(i - p * 3);
        const float ev = eps[p * ld_eps + d];
        const float zv = rng ? zz[e] : (z ? z[i] : 0.0f);
        const float xv = x[i];
        float r;
        if (MODE == 0) {
          r = (xv - a * ev) / b;
          r = r + c * zv;
        } else {
          r = xv * a;
          r = r + (b * ev + c * zv);
        }
        x[i] = r;
      }
    }
  }
  // step bookkeeping by the last workgroup: every workgroup has read *t_ptr / rng before it takes its ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int old = atomicAdd(ticket, 1);
    if (old == static_cast<int>(gridDim.x) - 1) {
      *ticket = 0;
      const long long tn = t - 1;
      *t_ptr = tn;
      if (ts_out && tn >= 0) *ts_out = ts_table ? ts_table[tn] : static_cast<float>(tn);
      if (rng) rng[1] += 1ull;
      if (probe_acc) {
        // this step's neighbourhood probe (accumulated by the geometry launches long before this kernel) -> slot
        // (t & 3) of the 4-slot ring the host reads, tagged with the step counter: the host asks for the probe of ONE
        // given step and can tell a slot that a later step has already overwritten (a run's switch decisions then do
        // not depend on how far the device has run ahead of the host); reset for the next step
        int* slot = probe_out + 4 * static_cast<int>(tt & 3);
        slot[0] = probe_acc[0];
        slot[1] = probe_acc[1];
        slot[2] = static_cast<int>(tt);
        slot[3] = 1;                                   // written (a zeroed ring holds no valid slot)
        probe_acc[0] = 0;
        probe_acc[1] = 0;
      }
    }
  }
}

// One thread stores the constant-rate (100 MHz) wall clock: a time stamp INSIDE a captured step, readable after an
// untraced replay (tools/lab/step_markers.py) -- rocprofv3's timeline of a two-stream graph is distorted by the tracer.
__global__ void mark_time_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

}  // namespace

extern "C" int pdr_mark_time(unsigned long long* slot, pdr_stream_t stream) {
  if (!slot) return PDR_EINVAL;
  hipLaunchKernelGGL(mark_time_kernel, dim3(1), dim3(1), 0, pdr::as_stream(stream), slot);
  return pdr::check_launch();
}

extern "C" int pdr_reverse_update(float* x, const float* eps, int ld_eps, const float* z, const float* tab_a,
                                  const float* tab_b, const float* tab_c, const long long* t_dev, long npoints,
                                  int mode, pdr_stream_t stream) {
  if (!x || !eps || !tab_a || !tab_b || !tab_c || !t_dev || npoints < 0 || ld_eps < 3 || (mode != 0 && mode != 1))
    return PDR_EINVAL;
  if (npoints == 0) return PDR_OK;
  const dim3 grid(static_cast<unsigned>((npoints * 3 + 255) / 256));
  hipStream_t s = pdr::as_stream(stream);
  if (mode == 0)
    hipLaunchKernelGGL(reverse_update_kernel<0>, grid, dim3(256), 0, s, x, eps, ld_eps, z, tab_a, tab_b, tab_c, t_dev,
                       npoints);
  else
    hipLaunchKernelGGL(reverse_update_kernel<1>, grid, dim3(256), 0, s, x, eps, ld_eps, z, tab_a, tab_b, tab_c, t_dev,
                       npoints);
  return pdr::check_launch();
}

extern "C" int pdr_reverse_step(float* x, const float* eps, int ld_eps, const float* z, const float* tab_a,
                                const float* tab_b, const float* tab_c, long long* t_dev, const float* ts_table,
                                float* ts_out, unsigned long long* rng_state, int* ticket, long npoints, int mode,
                                int* probe_acc, int* probe_out, pdr_stream_t stream) {
  if (!x || !eps || !tab_a || !tab_b || !tab_c || !t_dev || !ticket || npoints < 0 || ld_eps < 3 ||
      (mode != 0 && mode != 1) || ((probe_acc == nullptr) != (probe_out == nullptr)))
    return PDR_EINVAL;
  if (npoints == 0) return PDR_OK;
  const long total = npoints * 3;
  const dim3 grid(static_cast<unsigned>(((total + 3) / 4 + 255) / 256));
  hipStream_t s = pdr::as_stream(stream);
  if (mode == 0)
    hipLaunchKernelGGL(reverse_step_kernel<0>, grid, dim3(256), 0, s, x, eps, ld_eps, z, tab_a, tab_b, tab_c, t_dev,
                       ts_table, ts_out, rng_state, ticket, total, probe_acc, probe_out);
  else
    hipLaunchKernelGGL(reverse_step_kernel<1>, grid, dim3(256), 0, s, x, eps, ld_eps, z, tab_a, tab_b, tab_c, t_dev,
                       ts_table, ts_out, rng_state, ticket, total, probe_acc, probe_out);
  return pdr::check_launch();
}
