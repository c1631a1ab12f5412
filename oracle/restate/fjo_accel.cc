// oracle/restate/fjo_accel.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Restatement of the reference's accelerators and two-level scene structure:
//   uniform grid per mesh/curve     src/fj_grid_accelerator.cc
//   BVH over a group's instances    src/fj_bvh_accelerator.cc
//   Accelerator / PrimitiveSet      src/fj_accelerator.cc, src/fj_primitive_set.cc
//   ObjectInstance / ObjectGroup    src/fj_object_instance.cc, src/fj_object_group.cc
#include "fjo_scene.h"

#include <algorithm>
#include <cmath>

namespace fjo {

static const double ACC_PADDING = .0001;   // src/fj_accelerator.cc:13
static const int GRID_MAXCELLS = 512;      // src/fj_grid_accelerator.cc:14

// ---------------------------------------------------------------- primset glue
static void prim_bounds(const PrimSet &ps, int prim, Box *b)
{
  if (ps.type == FJ_PRIMSET_MESH) MeshPrimBounds(*ps.mesh, prim, b);
  else CurvePrimBounds(*ps.curve, prim, b);
}
static bool prim_box_intersect(const PrimSet &ps, int prim, const Box &box)
{
  if (ps.type == FJ_PRIMSET_MESH) return MeshBoxIntersect(*ps.mesh, prim, box);
  return CurveBoxIntersect(*ps.curve, prim, box);
}
static int prim_count(const PrimSet &ps)
{
  return ps.type == FJ_PRIMSET_MESH ? ps.mesh->n_faces : ps.curve->n_curves;
}

// PrimitiveSet::RayIntersect, src/fj_primitive_set.cc:10-26
static bool primset_ray_intersect(const PrimSet &ps, int prim, const Ray &ray, double time, Isect *isect)
{
  const bool hit = (ps.type == FJ_PRIMSET_MESH)
      ? MeshRayIntersect(*ps.mesh, prim, ray, time, isect)
      : CurveRayIntersect(ps, prim, ray, time, isect);
  if (!hit) { isect->t_hit = REAL_MAX; return false; }
  if (!(ray.tmin <= isect->t_hit && isect->t_hit <= ray.tmax)) { isect->t_hit = REAL_MAX; return false; }
  return true;
}

// ------------------------------------------------------------------ grid build
static Box grid_cell(const Box &gb, const V3 &cs, int x, int y, int z)   // :334-343
{
  Box c;
  c.min = gb.min + V3(x, y, z) * cs;
  c.max = c.min + cs;
  return c;
}

static void build_grid(PrimSet *ps)                                       // :69-160
{
  Grid &g = ps->grid;
  Box b = ps->bounds;
  b.Expand(ACC_PADDING);
  const int NPRIMS = prim_count(*ps);

  // compute_grid_cellsizes, :318-332
  const V3 size = b.Diagonal();
  const double max_width = Max(Max(size.x, size.y), size.z);
  const double cube_root = 3 * std::pow(NPRIMS, 1. / 3);
  const double per_unit = cube_root / max_width;
  int n[3];
  for (int i = 0; i < 3; i++) {
    const int c = static_cast<int>(std::floor(size[i] * per_unit + .5));
    n[i] = static_cast<int>(Clamp(c, 1, GRID_MAXCELLS));
  }
  const V3 cs = (b.max - b.min) / V3(n[0], n[1], n[2]);

  g.bounds = b;
  g.cellsize = cs;
  for (int i = 0; i < 3; i++) g.ncells[i] = n[i];
  const size_t ncell = static_cast<size_t>(n[0]) * n[1] * n[2];

  // two passes over the same candidate enumeration: count, then fill.  Prims are
  // visited in DESCENDING id in the fill pass so each cell's slice lists them
  // in the order the reference's push-front linked list yields (:127-135).
  std::vector<uint32_t> count(ncell + 1, 0);
  const double HALF_PADDING = .5 * ACC_PADDING;
  auto for_each_cell = [&](int prim, auto &&fn) {
    Box pb;
    prim_bounds(*ps, prim, &pb);
    pb.Expand(HALF_PADDING);
    int lo[3], hi[3];
    for (int i = 0; i < 3; i++) {
      int a0 = static_cast<int>(std::floor((pb.min[i] - b.min[i]) / cs[i]));
      int a1 = static_cast<int>(std::floor((pb.max[i] - b.min[i]) / cs[i]) + 1);
      lo[i] = static_cast<int>(Clamp(a0, 0, n[i]));
      hi[i] = static_cast<int>(Clamp(a1, 0, n[i]));
    }
    for (int z = lo[2]; z < hi[2]; z++)
      for (int y = lo[1]; y < hi[1]; y++)
        for (int x = lo[0]; x < hi[0]; x++) {
          const Box cb = grid_cell(b, cs, x, y, z);
          if (!prim_box_intersect(*ps, prim, cb)) continue;
          fn(static_cast<size_t>(z) * n[1] * n[0] + static_cast<size_t>(y) * n[0] + x);
        }
  };
  for (int p = 0; p < NPRIMS; p++) for_each_cell(p, [&](size_t c) { count[c + 1]++; });
  for (size_t c = 0; c < ncell; c++) count[c + 1] += count[c];
  g.cell_start = count;
  g.cell_prims.assign(count[ncell], 0);
  std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
  for (int p = NPRIMS - 1; p >= 0; p--) for_each_cell(p, [&](size_t c) { g.cell_prims[cursor[c]++] = p; });
}

// ------------------------------------------------------------- grid traversal
static bool grid_intersect(const PrimSet &ps, const Ray &ray, double time, Isect *isect)  // :162-306
{
  const Grid &g = ps.grid;
  double boxhit_tmin = REAL_MAX, boxhit_tmax = REAL_MAX;
  if (!BoxRayIntersect(g.bounds, ray.orig, ray.dir, ray.tmin, ray.tmax, &boxhit_tmin, &boxhit_tmax))
    return false;

  V3 start;
  double t_start = REAL_MAX, t_end = REAL_MAX;
  if (g.bounds.ContainsPoint(ray.orig)) {
    start = ray.orig;
    t_start = 0;
  } else {
    t_start = boxhit_tmin;
    t_end = boxhit_tmax;
    start = ray.orig + t_start * ray.dir;
  }
  t_end = Min(t_end, ray.tmax);

  const int *N = g.ncells;
  int cell_id[3], cell_step[3], cell_end[3];
  double t_next[3], t_delta[3];
  for (int i = 0; i < 3; i++) {
    cell_id[i] = static_cast<int>(std::floor((start[i] - g.bounds.min[i]) / g.cellsize[i]));
    cell_id[i] = static_cast<int>(Clamp(cell_id[i], 0, N[i] - 1));
    const double d = ray.dir[i];
    if (d > 0) {
      t_next[i] = t_start + (((cell_id[i] + 1) * g.cellsize[i] + g.bounds.min[i]) - start[i]) / d;
      t_delta[i] = g.cellsize[i] / d;
      cell_step[i] = +1;
      cell_end[i] = N[i];
    } else if (d < 0) {
      t_next[i] = t_start + ((cell_id[i] * g.cellsize[i] + g.bounds.min[i]) - start[i]) / d;
      t_delta[i] = -1 * g.cellsize[i] / d;
      cell_step[i] = -1;
      cell_end[i] = -1;
    } else {
      t_next[i] = REAL_MAX;
      t_delta[i] = 0;
      cell_step[i] = 0;
      cell_end[i] = -1;
    }
  }

  bool hit = false;
  for (;;) {
    Isect cand[2];
    Isect *imin = &cand[0], *itmp = &cand[1];
    imin->t_hit = REAL_MAX;
    const size_t id = static_cast<size_t>(N[0]) * N[1] * cell_id[2] + static_cast<size_t>(N[0]) * cell_id[1] + cell_id[0];
    for (uint32_t k = g.cell_start[id]; k < g.cell_start[id + 1]; k++) {
      if (!primset_ray_intersect(ps, g.cell_prims[k], ray, time, itmp)) continue;
      const Box cb = grid_cell(g.bounds, g.cellsize, cell_id[0], cell_id[1], cell_id[2]);
      const V3 P_hit = ray.orig + itmp->t_hit * ray.dir;
      if (!cb.ContainsPoint(P_hit)) continue;
      if (itmp->t_hit < imin->t_hit) { std::swap(imin, itmp); hit = true; }
    }
    if (hit) { *isect = *imin; break; }

    int ax;
    if ((t_next[0] < t_next[1]) && (t_next[0] < t_next[2])) ax = 0;
    else if (t_next[2] < t_next[1]) ax = 2;
    else ax = 1;
    if (t_end < t_next[ax]) break;
    cell_id[ax] += cell_step[ax];
    if (cell_id[ax] == cell_end[ax]) break;
    t_next[ax] += t_delta[ax];
  }
  return hit;
}

// Accelerator::Intersect, src/fj_accelerator.cc:94-113
static bool primset_accel_intersect(const PrimSet &ps, const Ray &ray, double time, Isect *isect)
{
  double a = 0, b = 0;
  if (!BoxRayIntersect(ps.acc_bounds, ray.orig, ray.dir, ray.tmin, ray.tmax, &a, &b)) return false;
  return grid_intersect(ps, ray, time, isect);
}

// ------------------------------------------------------------------ transforms
static void lerp_channel(const fj_xform_sample *s, int n, double time, double out[3])  // src/fj_property.cc:317-345
{
  if (s[0].time >= time || n == 1) { for (int i = 0; i < 3; i++) out[i] = s[0].v[i]; return; }
  if (s[n - 1].time <= time) { for (int i = 0; i < 3; i++) out[i] = s[n - 1].v[i]; return; }
  for (int k = 0; k < n; k++) {
    if (s[k].time == time) { for (int i = 0; i < 3; i++) out[i] = s[k].v[i]; return; }
    if (s[k].time > time) {
      const double t = Fit(time, s[k - 1].time, s[k].time, 0, 1);
      for (int i = 0; i < 3; i++) out[i] = (1 - t) * s[k - 1].v[i] + t * s[k].v[i];
      return;
    }
  }
}

void LerpXfm(const fj_xform_desc &x, double time, Xfm *out)   // src/fj_transform.cc:306-322
{
  double T[3], R[3], S[3];
  lerp_channel(x.translate, x.n_translate, time, T);
  lerp_channel(x.rotate, x.n_rotate, time, R);
  lerp_channel(x.scale, x.n_scale, time, S);
  XfmSetTransform(out, x.transform_order, x.rotate_order, T[0], T[1], T[2], R[0], R[1], R[2], S[0], S[1], S[2]);
}

// ObjectInstance::merge_sampled_bounds, src/fj_object_instance.cc:313-356
static void instance_bounds(Instance *inst)
{
  const fj_xform_desc &x = inst->d->xform;
  Box original = inst->primset->acc_bounds;
  if (x.n_rotate > 1) {
    const double half_diagonal = .5 * Length(original.Diagonal());
    const V3 c = original.Centroid();
    original = Box(c, c);
    original.Expand(half_diagonal);
  }
  double S[3] = {0, 0, 0};
  for (int i = 0; i < x.n_scale; i++)
    for (int k = 0; k < 3; k++) S[k] = Max(S[k], std::abs(x.scale[i].v[k]));
  Box merged;
  merged.ReverseInfinite();
  for (int i = 0; i < x.n_translate; i++) {
    // the reference indexes rotate.samples[i] with the translate index; slots
    // past n_rotate hold default (zero) samples
    const double *T = x.translate[i].v;
    double R[3] = {0, 0, 0};
    if (i < x.n_rotate) for (int k = 0; k < 3; k++) R[k] = x.rotate[i].v[k];
    Xfm t;
    XfmSetTransform(&t, x.transform_order, x.rotate_order, T[0], T[1], T[2], R[0], R[1], R[2], S[0], S[1], S[2]);
    Box sb = original;
    MatTransformBounds(t.matrix, &sb);
    merged.AddBox(sb);
  }
  inst->bounds = merged;
}

// ObjectInstance::RayIntersect, src/fj_object_instance.cc:213-243
static bool instance_ray_intersect(const Scene &sc, int idx, const Ray &ray, double time, Isect *isect)
{
  const Instance &inst = sc.instances[idx];
  Xfm lerped;
  const Xfm *x = &inst.xfm_static;
  if (!inst.is_static) { LerpXfm(inst.d->xform, time, &lerped); x = &lerped; }

  Ray ro = ray;
  ro.orig = MatTransformPoint(x->inverse, ray.orig);
  ro.dir = MatTransformVector(x->inverse, ray.dir);     // not renormalised: t is preserved

  if (!primset_accel_intersect(*inst.primset, ro, time, isect)) return false;

  isect->P = MatTransformPoint(x->matrix, isect->P);
  isect->N = Normalize(MatTransformVector(x->matrix, isect->N));
  isect->dPdu = MatTransformVector(x->matrix, isect->dPdu);
  isect->dPdv = MatTransformVector(x->matrix, isect->dPdv);
  isect->object = idx;
  return true;
}

// ------------------------------------------------------------------- group BVH
struct BvhPrim { Box bounds; V3 centroid; int index; };

static int find_median(BvhPrim **p, int begin, int end, int axis)   // src/fj_bvh_accelerator.cc:313-334
{
  int low = begin, high = end - 1, mid = -1;
  const double key = (p[low]->centroid[axis] + p[high]->centroid[axis]) / 2;
  while (low != mid) {
    mid = (low + high) / 2;
    if (key < p[mid]->centroid[axis]) high = mid;
    else if (p[mid]->centroid[axis] < key) low = mid;
    else break;
  }
  return mid + 1;
}

static int build_bvh(Group *g, BvhPrim **p, int begin, int end, int axis)   // :253-296
{
  const int id = static_cast<int>(g->nodes.size());
  g->nodes.push_back(BvhNode{-1, -1, Box(), -1});
  if (end - begin == 1) {
    g->nodes[id].prim_id = p[begin]->index;
    g->nodes[id].bounds = p[begin]->bounds;
    return id;
  }
  // std::sort on the pointer range with a strict centroid[axis] comparator, as
  // the reference does (same libstdc++ introsort => same permutation on ties)
  std::sort(p + begin, p + end, [axis](BvhPrim *a, BvhPrim *b) { return a->centroid[axis] < b->centroid[axis]; });
  const int median = find_median(p, begin, end, axis);
  const int new_axis = (axis + 1) % 3;
  const int l = build_bvh(g, p, begin, median, new_axis);
  const int r = build_bvh(g, p, median, end, new_axis);
  g->nodes[id].left = l;
  g->nodes[id].right = r;
  g->nodes[id].bounds = g->nodes[l].bounds;
  g->nodes[id].bounds.AddBox(g->nodes[r].bounds);
  return id;
}

static void build_group(const Scene &sc, Group *g)
{
  const int n = static_cast<int>(g->instances.size());
  g->root = -1;
  Box set_bounds;
  set_bounds.ReverseInfinite();     // ObjectSet ctor, src/fj_object_set.cc:11-14
  for (int i = 0; i < n; i++) set_bounds.AddBox(sc.instances[g->instances[i]].bounds);
  if (n == 0) set_bounds = Box();   // NullPrimitiveSet bounds
  g->acc_bounds = set_bounds;
  g->acc_bounds.Expand(ACC_PADDING);
  if (n == 0) return;
  std::vector<BvhPrim> prims(n);
  std::vector<BvhPrim *> ptrs(n);
  for (int i = 0; i < n; i++) {
    prims[i].bounds = sc.instances[g->instances[i]].bounds;
    prims[i].centroid = prims[i].bounds.Centroid();
    prims[i].index = i;
    ptrs[i] = &prims[i];
  }
  g->root = build_bvh(g, ptrs.data(), 0, n, 0);
}

// Accelerator::Intersect + intersect_bvh_loop, src/fj_bvh_accelerator.cc:164-241
bool GroupIntersect(const Scene &sc, int group, const Ray &ray, double time, Isect *isect)
{
  const Group &g = sc.groups[group];
  double a = 0, b = 0;
  if (!BoxRayIntersect(g.acc_bounds, ray.orig, ray.dir, ray.tmin, ray.tmax, &a, &b)) return false;
  if (g.root < 0) return false;

  bool hit = false;
  int node = g.root;
  int stack[128];
  int sp = 0;
  Isect cand[2];
  Isect *imin = &cand[0], *itmp = &cand[1];
  for (;;) {
    const BvhNode &nd = g.nodes[node];
    if (nd.left < 0) {
      // ObjectSet::ray_intersect through PrimitiveSet::RayIntersect (range re-check)
      bool h = instance_ray_intersect(sc, g.instances[nd.prim_id], ray, time, itmp);
      if (!h) itmp->t_hit = REAL_MAX;
      else if (!(ray.tmin <= itmp->t_hit && itmp->t_hit <= ray.tmax)) { itmp->t_hit = REAL_MAX; h = false; }
      if (h && itmp->t_hit < imin->t_hit) { std::swap(imin, itmp); hit = true; }
      if (sp == 0) break;
      node = stack[--sp];
      continue;
    }
    double t0, t1;
    const bool hl = BoxRayIntersect(g.nodes[nd.left].bounds, ray.orig, ray.dir, ray.tmin, ray.tmax, &t0, &t1);
    const bool hr = BoxRayIntersect(g.nodes[nd.right].bounds, ray.orig, ray.dir, ray.tmin, ray.tmax, &t0, &t1);
    if (hl && hr) { stack[sp++] = nd.right; node = nd.left; }
    else if (hl) node = nd.left;
    else if (hr) node = nd.right;
    else { if (sp == 0) break; node = stack[--sp]; }
  }
  if (hit) *isect = *imin;
  return hit;
}

// ----------------------------------------------------------------- scene build
void BuildScene(const fj_scene_desc *d, Scene *sc)
{
  sc->d = d;
  sc->meshes.resize(d->n_meshes);
  for (int i = 0; i < d->n_meshes; i++) {
    PrimSet &ps = sc->meshes[i];
    ps.type = FJ_PRIMSET_MESH;
    ps.mesh = &d->meshes[i];
    ps.curve = nullptr;
    const double *b = d->meshes[i].bounds;
    ps.bounds = Box(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5]));
    ps.acc_bounds = ps.bounds;
    ps.acc_bounds.Expand(ACC_PADDING);      // Accelerator::ComputeBounds, src/fj_accelerator.cc:62-66
    build_grid(&ps);
  }
  sc->curves.resize(d->n_curves);
  for (int i = 0; i < d->n_curves; i++) {
    PrimSet &ps = sc->curves[i];
    ps.type = FJ_PRIMSET_CURVE;
    ps.mesh = nullptr;
    ps.curve = &d->curves[i];
    const double *b = d->curves[i].bounds;
    ps.bounds = Box(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5]));
    ps.acc_bounds = ps.bounds;
    ps.acc_bounds.Expand(ACC_PADDING);
    CurveCacheSplitDepth(&ps);
    build_grid(&ps);
  }
  sc->instances.resize(d->n_instances);
  for (int i = 0; i < d->n_instances; i++) {
    Instance &in = sc->instances[i];
    in.d = &d->instances[i];
    in.primset = (in.d->primset_type == FJ_PRIMSET_MESH) ? &sc->meshes[in.d->primset] : &sc->curves[in.d->primset];
    const fj_xform_desc &x = in.d->xform;
    in.is_static = (x.n_translate == 1 && x.n_rotate == 1 && x.n_scale == 1);
    LerpXfm(x, 0, &in.xfm_static);
    instance_bounds(&in);
  }
  sc->groups.resize(d->n_groups);
  for (int i = 0; i < d->n_groups; i++) {
    Group &g = sc->groups[i];
    g.instances.assign(d->groups[i].instances, d->groups[i].instances + d->groups[i].n_instances);
    build_group(*sc, &g);
  }
}

}  // namespace fjo
