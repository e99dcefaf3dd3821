// Moving-neighbourhood kriging with 65 .. 256 neighbours (krig.jl:201-210 and ui.jl:16-23 accept any count): the
// per-point systems of approxsolve's loop (/root/reference/src/estimation/krig.jl:205-228) on the MFMA tile core.
// This unit holds the general instantiations (any model, nested models, 1-D) and the dispatch; the common
// single-structure models have units of their own (krig_tiles_exponential / _spherical / _matern32.hip).
#include "krig_tiles_kernel.h"

namespace gss {

int32_t krig_local_tiles_launch(const VgDev& vg, const LocalSpec& sp, int dim, const double* xdata, const double* z,
                                const double* drift_data, const double* x0, const double* drift_dom, int64_t m, int k,
                                int minneighbors, const int* idx, const int* count, double* mean, double* var,
                                uint8_t* status, hipStream_t s) {
  GSS_REQUIRE(k > LMAX_K && k <= 256, "krig_local_tiles_launch: %d neighbours outside 65..256", k);
  TilesArgs a;
  a.ntmax = k <= 96 ? 6 : (k <= 128 ? 8 : 16);   // 3 waves (four workgroups per CU) / 4 waves (three) / 8 waves (one)
  const int per_cu = a.ntmax == 6 ? 4 : (a.ntmax == 8 ? 3 : 1);
  a.blocks = (int64_t)256 * per_cu * 2;        // two rounds of resident workgroups, points handed out by stride
  if (a.blocks > m) a.blocks = m;
  a.s = s; a.vg = &vg; a.sp = &sp;
  a.xdata = xdata; a.z = z; a.drift_data = drift_data; a.x0 = x0; a.drift_dom = drift_dom;
  a.m = m; a.k = k; a.minneighbors = minneighbors; a.idx = idx; a.count = count;
  a.mean = mean; a.var = var; a.status = status;
  const int kind = vg.nextra == 0 ? vg.kind : -1;
  if (dim == 2 || dim == 3) {
    switch (kind) {
      case GSS_VG_EXPONENTIAL: return krig_tiles_exponential(dim, a);
      case GSS_VG_SPHERICAL: return krig_tiles_spherical(dim, a);
      case VG_MATERN32: return krig_tiles_matern32(dim, a);
      default: break;
    }
  }
  if (dim == 3) return tiles_launch<3, -1>(a);
  if (dim == 2) return tiles_launch<2, -1>(a);
  return tiles_launch<1, -1>(a);
}

}  // namespace gss
