// Stand-alone noise kernels (test support and LUGS normals).
#include "gss_internal.h"
#include "philox.h"

namespace gss {

// element e of a realisation lives at out[(e / n1) * ld + e % n1]: ld > n1 gives the padded rows an
// in-place real-to-complex transform wants
// (blockIdx.y: consecutive realisation numbers, outputs `bstride` doubles apart)
__global__ __launch_bounds__(256) void philox_uniform_kernel(uint64_t seed, uint32_t real, int64_t n,
                                                             double* __restrict__ out, int64_t ld, int64_t n1,
                                                             int64_t bstride) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;  // block -> elements 2b, 2b+1
  const int64_t e0 = 2 * b;
  if (e0 >= n) return;
  real += blockIdx.y;
  out += (int64_t)blockIdx.y * bstride;
  double ua, ub;
  philox_pair(seed, real, STREAM_UNIFORM, (uint64_t)b, ua, ub);
  out[(e0 / n1) * ld + e0 % n1] = ua;
  const int64_t e1 = e0 + 1;
  if (e1 < n) out[(e1 / n1) * ld + e1 % n1] = ub;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(uint64_t seed, uint32_t real, int64_t n,
                                                            double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) out[e] = philox_normal(seed, real, (uint64_t)e);
}

int32_t philox_uniform_dev(uint64_t seed, int64_t real, int64_t n, double* out, int64_t ld, int64_t n1,
                           hipStream_t s, int nreals, int64_t bstride) {
  if (n <= 0 || nreals <= 0) return GSS_OK;
  const int64_t nb = (n + 1) / 2;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)((nb + 255) / 256), (unsigned)nreals), dim3(256), 0, s, seed,
                     (uint32_t)real, n, out, ld, n1, bstride);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

int32_t philox_normal_dev(uint64_t seed, int64_t real, int64_t n, double* out, hipStream_t s) {
  if (n <= 0) return GSS_OK;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed,
                     (uint32_t)real, n, out);
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" {

int32_t gss_philox_uniform(uint64_t seed, int64_t real, int64_t n, double* out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(out != nullptr && n >= 0, "gss_philox_uniform: bad arguments");
  hipStream_t s = to_stream(stream);
  Staged so;
  GSS_TRY(so.out(out, sizeof(double) * n, mem));
  GSS_TRY(philox_uniform_dev(seed, real, n, so.as<double>(), n, n, s));
  return so.back(out, sizeof(double) * n, mem, s);
}

int32_t gss_philox_normal(uint64_t seed, int64_t real, int64_t n, double* out, int32_t mem, void* stream) {
  GSS_ENTRY();
  GSS_REQUIRE(out != nullptr && n >= 0, "gss_philox_normal: bad arguments");
  hipStream_t s = to_stream(stream);
  Staged so;
  GSS_TRY(so.out(out, sizeof(double) * n, mem));
  GSS_TRY(philox_normal_dev(seed, real, n, so.as<double>(), s));
  return so.back(out, sizeof(double) * n, mem, s);
}

}  // extern "C"
