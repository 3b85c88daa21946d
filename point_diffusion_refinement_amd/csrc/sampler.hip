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
