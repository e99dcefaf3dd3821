// K1: pairwise covariance assembly  out(i, j) = sill - gamma(|a_i - b_j|).
// Replaces `sill(g) .- Variography.pairwise(g, A, B)` (/root/reference/src/simulation/fft.jl:98,
// lu.jl:124,131,132) and the covariance blocks GeoStatsModels.fit builds (krig.jl:176,223).
//
// Layout: one lane per column point b_j (its coordinates live in registers), rows a_i are wave
// uniform, so their coordinates arrive through the scalar cache; a wave writes 64 consecutive
// doubles (512 B) of one output row per iteration -> fully coalesced HBM stores.
#include "gss_internal.h"

namespace gss {

constexpr int ROWS_PER_BLOCK = 64;

template <int DIM>
__global__ __launch_bounds__(256) void cov_pairwise_kernel(VgDev vg, const double* __restrict__ a, int64_t na,
                                                           const double* __restrict__ b, int64_t nb,
                                                           double* __restrict__ out, int64_t ldo) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * ROWS_PER_BLOCK;
  const int64_t i1 = i0 + ROWS_PER_BLOCK < na ? i0 + ROWS_PER_BLOCK : na;
  if (j >= nb) return;
  double c[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) c[k] = b[j * DIM + k];
  for (int64_t i = i0; i < i1; ++i) {
    double x[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) x[k] = a[i * DIM + k];
    out[i * ldo + j] = cov_pair<DIM>(vg, x, c);
  }
}

int32_t cov_pairwise_dev(const VgDev& vg, const double* a, int64_t na, const double* b, int64_t nb, double* out,
                         int64_t ldo, hipStream_t s) {
  if (na <= 0 || nb <= 0) return GSS_OK;
  dim3 grid((unsigned)((nb + 255) / 256), (unsigned)((na + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  switch (vg.dim) {
    case 1: hipLaunchKernelGGL((cov_pairwise_kernel<1>), grid, dim3(256), 0, s, vg, a, na, b, nb, out, ldo); break;
    case 2: hipLaunchKernelGGL((cov_pairwise_kernel<2>), grid, dim3(256), 0, s, vg, a, na, b, nb, out, ldo); break;
    default: hipLaunchKernelGGL((cov_pairwise_kernel<3>), grid, dim3(256), 0, s, vg, a, na, b, nb, out, ldo); break;
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" int32_t gss_cov_pairwise(const gss_variogram_t* vg, const double* a, int64_t na, const double* b,
                                    int64_t nb, double* out, int64_t ldo, int32_t mem, void* stream) {
  GSS_ENTRY();
  VgDev v;
  GSS_TRY(make_vgdev(vg, &v));
  GSS_REQUIRE(a != nullptr && out != nullptr && na >= 0, "gss_cov_pairwise: bad arguments");
  if (b == nullptr) {
    b = a;
    nb = na;
  }
  GSS_REQUIRE(ldo >= nb, "gss_cov_pairwise: ldo %lld < nb %lld", (long long)ldo, (long long)nb);
  hipStream_t s = to_stream(stream);
  Staged sa, sb;
  GSS_TRY(sa.in(a, sizeof(double) * na * v.dim, mem, s));
  if (b == a) sb.p = sa.p;
  else GSS_TRY(sb.in(b, sizeof(double) * nb * v.dim, mem, s));
  if (mem == GSS_MEM_DEVICE) {
    GSS_TRY(cov_pairwise_dev(v, sa.as<double>(), na, sb.as<double>(), nb, out, ldo, s));
  } else {
    DevBuf tmp;  // compact device image, copied back row by row into the caller's pitch
    GSS_TRY(tmp.alloc(sizeof(double) * (size_t)(na * nb)));
    GSS_TRY(cov_pairwise_dev(v, sa.as<double>(), na, sb.as<double>(), nb, tmp.as<double>(), nb, s));
    GSS_HIP(hipMemcpy2DAsync(out, sizeof(double) * ldo, tmp.p, sizeof(double) * nb, sizeof(double) * nb, na,
                             hipMemcpyDeviceToHost, s));
    GSS_HIP(hipStreamSynchronize(s));
  }
  return GSS_OK;
}
