// K4: exact k-nearest-neighbour search by brute force, deterministic ordering.
// Replaces `search!(neighbors, center, searcher)` (/root/reference/src/estimation/krig.jl:210) with
// the KNearestSearch / KBallSearch built at /root/reference/src/ui.jl:27,30 ([DEP] Meshes ->
// NearestNeighbors KD-tree), and the 1-NN lookups of fft.jl:129-132 and initbuff (lu.jl:86).
//
// Contract: neighbours are ranked by ascending (d2, index) where d2 is the FP64
// squared distance accumulated in dimension order with one rounding per operation (no FMA), so
// the oracle reproduces the indices bit for bit.  With a ball only d2 <= r^2 qualifies.
//
// Mapping: one wave owns Q = 4 query points and keeps each query's current best-k list sorted
// across its lanes (lane l holds the l-th nearest so far; k <= 64).  Data points are staged in LDS
// as structure-of-arrays tiles; every 64-candidate batch is tested against the four thresholds with
// one ballot each, and only qualifying candidates pay for an insertion (ballot + lane shift).
#include "gss_internal.h"

#include <climits>

namespace gss {

constexpr int KNN_Q = 4;
constexpr int KNN_TILE = 2048;

__device__ __forceinline__ bool key_less(double ad, int ai, double bd, int bi) {
  return ad < bd || (ad == bd && ai < bi);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <int DIM>
__global__ __launch_bounds__(256) void knn_kernel(const double* __restrict__ xdata, int n,
                                                  const double* __restrict__ centers, int64_t m, int k, double r2,
                                                  int use_ball, int aniso, double ir0, double ir1, double ir2,
                                                  int* __restrict__ idx_out, int* __restrict__ count_out) {
  __shared__ double tile[3][KNN_TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t qbase = ((int64_t)blockIdx.x * 4 + wave) * KNN_Q;
  const double ir[3] = {ir0, ir1, ir2};

  double qc[KNN_Q][DIM];
  bool qvalid[KNN_Q];
#pragma unroll
  for (int q = 0; q < KNN_Q; ++q) {
    const int64_t p = qbase + q;
    qvalid[q] = p < m;
    const int64_t pc = qvalid[q] ? p : m - 1;
#pragma unroll
    for (int a = 0; a < DIM; ++a) qc[q][a] = centers[pc * DIM + a];
  }
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  double ld[KNN_Q], tau_d[KNN_Q];
  int li[KNN_Q], tau_i[KNN_Q];
#pragma unroll
  for (int q = 0; q < KNN_Q; ++q) {
    ld[q] = INF;
    li[q] = INT_MAX;
    tau_d[q] = INF;
    tau_i[q] = INT_MAX;
  }

  for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
    const int tn = (n - t0) < KNN_TILE ? (n - t0) : KNN_TILE;
    __syncthreads();
    for (int e = tid; e < tn * DIM; e += 256) tile[e % DIM][e / DIM] = xdata[(int64_t)t0 * DIM + e];
    __syncthreads();
    for (int b = 0; b < tn; b += 64) {
      const int j = b + lane;
      const bool valid = j < tn;
      double c[DIM];
#pragma unroll
      for (int a = 0; a < DIM; ++a) c[a] = valid ? tile[a][j] : 0.0;
      const int gidx = t0 + j;
#pragma unroll
      for (int q = 0; q < KNN_Q; ++q) {
        const double d2 = sqdist_nofma<DIM>(c, qc[q], ir, aniso != 0);
        const bool qual = valid && (!use_ball || d2 <= r2) && key_less(d2, gidx, tau_d[q], tau_i[q]);
        unsigned long long mask = __ballot(qual);
        while (mask) {
          const int src = __builtin_ctzll(mask);
          mask &= mask - 1;
          const double cd = readlane_f64(d2, src);
          const int ci = __builtin_amdgcn_readlane(gidx, src);
          if (!key_less(cd, ci, tau_d[q], tau_i[q])) continue;  // an earlier insertion tightened tau
          const int pos = __popcll(__ballot(key_less(ld[q], li[q], cd, ci)));
          const double up_d = __shfl_up(ld[q], 1);
          const int up_i = __shfl_up(li[q], 1);
          if (lane > pos) {
            ld[q] = up_d;
            li[q] = up_i;
          } else if (lane == pos) {
            ld[q] = cd;
            li[q] = ci;
          }
          tau_d[q] = readlane_f64(ld[q], k - 1);
          tau_i[q] = __builtin_amdgcn_readlane(li[q], k - 1);
        }
      }
    }
  }

#pragma unroll
  for (int q = 0; q < KNN_Q; ++q) {
    const int64_t p = qbase + q;
    const bool has = lane < k && li[q] != INT_MAX;
    const int cnt = __popcll(__ballot(has));
    if (p < m) {
      if (lane < k) idx_out[p * k + lane] = has ? li[q] : -1;
      if (lane == 0 && count_out) count_out[p] = cnt;
    }
  }
}

int32_t knn_search_dev(const double* xdata, int64_t n, int dim, const double* centers, int64_t m, int k,
                       double radius, const double* inv_radii_host, int* idx, int* count, hipStream_t s) {
  GSS_REQUIRE(k >= 1 && k <= 64, "maxneighbors = %d: the moving-neighbourhood kernels hold at most 64 neighbours "
                                 "(use the global neighbourhood beyond that)", k);
  GSS_REQUIRE(n >= 1 && n < INT_MAX && dim >= 1 && dim <= 3, "knn: bad sizes");
  if (m <= 0) return GSS_OK;
  const int use_ball = (radius >= 0.0 || inv_radii_host != nullptr) ? 1 : 0;
  const int aniso = inv_radii_host != nullptr ? 1 : 0;
  const double r2 = aniso ? 1.0 : radius * radius;
  double ir[3] = {1.0, 1.0, 1.0};
  if (aniso)
    for (int a = 0; a < dim; ++a) ir[a] = inv_radii_host[a];
  dim3 grid((unsigned)((m + 4 * KNN_Q - 1) / (4 * KNN_Q)));
  switch (dim) {
    case 1:
      hipLaunchKernelGGL((knn_kernel<1>), grid, dim3(256), 0, s, xdata, (int)n, centers, m, k, r2, use_ball, aniso,
                         ir[0], ir[1], ir[2], idx, count);
      break;
    case 2:
      hipLaunchKernelGGL((knn_kernel<2>), grid, dim3(256), 0, s, xdata, (int)n, centers, m, k, r2, use_ball, aniso,
                         ir[0], ir[1], ir[2], idx, count);
      break;
    default:
      hipLaunchKernelGGL((knn_kernel<3>), grid, dim3(256), 0, s, xdata, (int)n, centers, m, k, r2, use_ball, aniso,
                         ir[0], ir[1], ir[2], idx, count);
      break;
  }
  GSS_HIP(hipGetLastError());
  return GSS_OK;
}

}  // namespace gss

using namespace gss;

extern "C" int32_t gss_knn_search(const double* xdata, int64_t n, int32_t dim, const double* centers, int64_t m,
                                  int32_t k, double radius, const double* inv_radii, int32_t* idx, int32_t* count,
                                  int32_t mem, void* stream) {
  GSS_REQUIRE(xdata && centers && idx, "gss_knn_search: NULL array");
  GSS_REQUIRE(k >= 1 && k <= n, "gss_knn_search: k = %d outside 1..n = %lld", k, (long long)n);
  hipStream_t s = to_stream(stream);
  Staged sx, sc, si, sn;
  GSS_TRY(sx.in(xdata, sizeof(double) * (size_t)(n * dim), mem, s));
  GSS_TRY(sc.in(centers, sizeof(double) * (size_t)(m * dim), mem, s));
  GSS_TRY(si.out(idx, sizeof(int32_t) * (size_t)(m * k), mem));
  GSS_TRY(sn.out(count, sizeof(int32_t) * (size_t)m, mem));
  GSS_TRY(knn_search_dev(sx.as<double>(), n, dim, sc.as<double>(), m, k, radius, inv_radii, si.as<int>(),
                         sn.as<int>(), s));
  GSS_TRY(si.back(idx, sizeof(int32_t) * (size_t)(m * k), mem, s));
  GSS_TRY(sn.back(count, sizeof(int32_t) * (size_t)m, mem, s));
  return GSS_OK;
}
