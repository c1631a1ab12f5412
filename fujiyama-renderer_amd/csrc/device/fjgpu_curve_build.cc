// fjgpu_curve_build.cc -- BLAS over cubic Bezier curve primitives (config 5).
#include "fjgpu_build.h"
#include "fjgpu.h"

namespace fjgpu {

int BuildCurveSet(const fj_curve_desc &, HostPrimSet *, std::string *err)
{
  *err = "Curve primitives are not on the device path yet";
  return FJGPU_EUNSUPPORTED;
}

}  // namespace fjgpu
