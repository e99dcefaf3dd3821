// Shared by the moving-neighbourhood kernels (krig_local.hip: up to 64 neighbours and beyond 256; krig_tiles.hip:
// 65 .. 256): the specification of the per-point system and the finishing step on the Gram matrix.
#pragma once

#include "gss_internal.h"
#include "tile16.h"

namespace gss {

constexpr int LMAX_K = 64;
constexpr int LMAX_NC = 10;
constexpr int LMAX_RHS = LMAX_NC + 2;

struct LocalSpec {
  int variant;
  int nc;
  int dim;
  signed char e[LMAX_NC][3];
  double inv_scale;  // monomial scaling (1 / data extent), conditioning only
  double sk_mean;
  // block support (gss_krig_set_block_support): c0 is the mean covariance to the bsub^dim sub-cell centres of a cell of
  // size bcell about the estimation point; c00 = mean covariance between two of them (the sill for point support)
  int bsub;
  double bcell[3];
  double c00;
};

// c0 entry of one neighbour.  xs: the neighbour (already divided by the radii of the model's ball when KIND >= 0 and the
// model is anisotropic, and -- UNIT -- multiplied by the model's own scale, kpos_scale), c: the estimation point as given,
// cs: the same scaled like xs, sca: the per-axis factor xs and cs carry (UNIT only).  Point support: one covariance;
// block support: the mean over the sub-cell centres, first axis slowest as in the oracle.
template <int DIM, int KIND, bool UNIT = false>
__device__ __forceinline__ double c0_entry(const VgDev& vg, const LocalSpec& sp, const double* xs, const double* c,
                                           const double* cs, const double* sca = nullptr) {
  if (sp.bsub <= 0) return cov_pair_k<DIM, KIND, UNIT>(vg, xs, cs);
  const int nsub = sp.bsub;
  const int ns = DIM == 1 ? nsub : (DIM == 2 ? nsub * nsub : nsub * nsub * nsub);
  double acc = 0.0;
  for (int s = 0; s < ns; ++s) {
    int q = s;
    double pt[DIM];
#pragma unroll
    for (int a = DIM - 1; a >= 0; --a) {
      const int ia = q % nsub;
      q /= nsub;
      const double v = c[a] + (((double)ia + 0.5) / (double)nsub - 0.5) * sp.bcell[a];
      pt[a] = UNIT ? mul_rounded(v, sca[a]) : ((KIND >= 0 && vg.aniso) ? mul_rounded(v, vg.ir[a]) : v);
    }
    acc += cov_pair_k<DIM, KIND, UNIT>(vg, xs, pt);
  }
  return acc / (double)ns;
}


// Block elimination on the (2 + nc) x (2 + nc) Gram matrix G = Y'Y of the forward-substituted right-hand sides
// [c0 | z | F] (one wave; lane i = drift term i, everything in registers; lanes 16..63 shadow lanes 0..15):
// S = Y_F'Y_F = L L', u = L^-1 (Y_F'y_c - f0), v = L^-1 Y_F'y_z, r'S^-1 r = |u|^2, t'S^-1 r = u.v, then
// sigma^2 = max(0, sill - q + r'S^-1 r), mu = a - t'S^-1 r (module header).  Writes the point's outputs when `live`.
__device__ __forceinline__ void gram_finish(const double (*G)[17], const signed char (*se)[4], const VgDev& vg,
                                            const LocalSpec& sp, const double* __restrict__ drift_dom, int64_t p,
                                            int lane, bool live, double* __restrict__ mean_out,
                                            double* __restrict__ var_out, uint8_t* __restrict__ status_out) {
  const int nc = sp.nc;
  const double NaN = __longlong_as_double(0x7ff8000000000000LL);
  const double qf = G[0][0], af = G[1][0];
  double rsr = 0.0, tsr = 0.0;
  bool okS = true;
  if (nc > 0) {
    const int i = (lane & 15) < LMAX_NC ? (lane & 15) : LMAX_NC - 1;  // lanes 16..63 shadow lanes 0..15 (bc16)
    double srow[LMAX_NC];
#pragma unroll
    for (int cc = 0; cc < LMAX_NC; ++cc) srow[cc] = G[2 + i][2 + cc];
    double f0 = 1.0;
    if (sp.variant == GSS_KRIG_UNIVERSAL) f0 = (se[i][0] + se[i][1] + se[i][2]) == 0 ? 1.0 : 0.0;
    else if (sp.variant == GSS_KRIG_EXTDRIFT) f0 = drift_dom[p * nc + (i < nc ? i : 0)];
    double u = G[2 + i][0] - f0, v = G[2 + i][1];
    static_for<0, LMAX_NC>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if (j < nc) {
        double d = bc16<j>(srow[j]);
        if (!(d > 0.0)) {
          okS = false;
          d = 1.0;
        }
        double y = __builtin_amdgcn_rsq(d);
        const double h = 0.5 * d;
        y = fma(y, fma(-h * y, y, 0.5), y);
        y = fma(y, fma(-h * y, y, 0.5), y);
        double lij = srow[j] * y;  // L[i][j] for i > j; lane cc holds L[cc][j]
        if (j + 1 < LMAX_NC) dpp_fence(lij);
        static_for<j + 1, LMAX_NC>([&](auto CC) {
          constexpr int cc = decltype(CC)::value;
          fmac_bc16<cc, true>(srow[cc], lij, lij);  // srow[cc] -= L[cc][j] * L[i][j]
        });
        const double uj = bc16<j>(u) * y, vj = bc16<j>(v) * y;
        rsr = fma(uj, uj, rsr);
        tsr = fma(uj, vj, tsr);
        u = fma(-lij, uj, u);
        v = fma(-lij, vj, v);
      }
    });
  }
  if (live && lane == 0) {
    if (!okS) {
      mean_out[p] = NaN;
      var_out[p] = NaN;
      status_out[p] = GSS_PT_SINGULAR;
    } else {
      const double mu = (sp.variant == GSS_KRIG_SIMPLE ? sp.sk_mean : 0.0) + af - tsr;
      const double vv = sp.c00 - qf + rsr;
      mean_out[p] = mu;
      var_out[p] = vv > 0.0 ? vv : 0.0;
      status_out[p] = GSS_PT_OK;
    }
  }
}



// 65 .. 256 neighbours (krig_tiles.hip): per-point systems on the tile core, one workgroup per point.  Neighbour lists
// idx (m x k) / count (m) as the search wrote them; asynchronous on s.
int32_t krig_local_tiles_launch(const VgDev& vg, const LocalSpec& sp, int dim, const double* xdata, const double* z,
                                const double* drift_data, const double* x0, const double* drift_dom, int64_t m, int k,
                                int minneighbors, const int* idx, const int* count, double* mean, double* var,
                                uint8_t* status, hipStream_t s);

// 257 .. 768 neighbours (krig_slab.hip): the same block algorithm with the tile triangle in a per-workgroup slab of global
// memory (round 4).  Synchronises s before it returns (the slabs are scratch of the call).
int32_t krig_local_slab_launch(const VgDev& vg, const LocalSpec& sp, int dim, const double* xdata, const double* z,
                               const double* drift_data, const double* x0, const double* drift_dom, int64_t m, int k,
                               int minneighbors, const int* idx, const int* count, double* mean, double* var,
                               uint8_t* status, hipStream_t s);

}  // namespace gss
