// Entry points declared in gss.h whose kernels are not built yet.  They fail loudly.
#include "gss_internal.h"
using namespace gss;
#define GSS_STUB(name) do { set_error(name ": not implemented in this build"); return GSS_ERR_UNSUPPORTED; } while (0)
extern "C" {
int32_t gss_krig_predict_global_batch(gss_krig_t*, const double*, int64_t, const double*, int64_t, double*, int32_t,
                                      void*) { GSS_STUB("gss_krig_predict_global_batch"); }
int32_t gss_lugs_create(gss_lugs_t**, const gss_variogram_t*, const double*, int64_t, const int64_t*, const double*,
                        int64_t, double, int32_t, void*) { GSS_STUB("gss_lugs_create"); }
int32_t gss_lugs_destroy(gss_lugs_t*) { return GSS_OK; }
int32_t gss_lugs_info(const gss_lugs_t*, int64_t*, int64_t*) { GSS_STUB("gss_lugs_info"); }
int32_t gss_lugs_factor(gss_lugs_t*, double*, double*, int32_t, void*) { GSS_STUB("gss_lugs_factor"); }
int32_t gss_lugs_state_buffer(gss_lugs_t*, void**, int64_t*) { GSS_STUB("gss_lugs_state_buffer"); }
int32_t gss_lugs_adopt_state(gss_lugs_t*) { GSS_STUB("gss_lugs_adopt_state"); }
int32_t gss_lugs_realize(gss_lugs_t*, uint64_t, int64_t, int64_t, const double*, double, const double*, double*,
                         double*, int32_t, void*) { GSS_STUB("gss_lugs_realize"); }
}
