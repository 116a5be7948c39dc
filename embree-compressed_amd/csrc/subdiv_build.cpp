#include "subdiv_build.h"

namespace rtamd {

void build_subdiv_accel(Scene* s)
{
  s->subdivAccel.clear();
  for (Geometry* g : s->geometries) {
    if (!g || !g->enabled || g->type != RTC_GEOMETRY_TYPE_SUBDIVISION) continue;
    RT_THROW(RTC_ERROR_INVALID_OPERATION, "subdivision geometry: device accel not built yet in this revision");
  }
}

} // namespace rtamd
