// 65 .. 256 neighbours, spherical model: 2-D and 3-D instantiations of krig_local_tiles_kernel (krig_tiles_kernel.h).
#include "krig_tiles_kernel.h"

namespace gss {

int32_t krig_tiles_spherical(int dim, const TilesArgs& a) {
  return dim == 3 ? tiles_launch<3, GSS_VG_SPHERICAL>(a) : tiles_launch<2, GSS_VG_SPHERICAL>(a);
}

}  // namespace gss
