// 65 .. 256 neighbours, matern32 model: 2-D and 3-D instantiations of krig_local_tiles_kernel (krig_tiles_kernel.h).
#include "krig_tiles_kernel.h"

namespace gss {

int32_t krig_tiles_matern32(int dim, const TilesArgs& a) {
  return dim == 3 ? tiles_launch<3, VG_MATERN32>(a) : tiles_launch<2, VG_MATERN32>(a);
}

}  // namespace gss
