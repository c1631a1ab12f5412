// oracle/restate/fjo_curve.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Placeholder: Bezier curve primitives (src/fj_curve.cc) are restated later in
// round 1; until then a scene with curves is rejected by the caller.
#include "fjo_scene.h"
namespace fjo {
bool CurveRayIntersect(const PrimSet &, int, const Ray &, double, Isect *) { return false; }
void CurvePrimBounds(const fj_curve_desc &, int, Box *b) { *b = Box(); }
bool CurveBoxIntersect(const fj_curve_desc &, int, const Box &) { return false; }
void CurveCacheSplitDepth(PrimSet *) {}
}
