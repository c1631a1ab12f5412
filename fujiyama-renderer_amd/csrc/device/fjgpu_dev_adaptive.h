// fjgpu_dev_adaptive.h -- AdaptiveGridSampler (src/fj_adaptive_grid_sampler.cc) as a
// level-synchronous wavefront.
//
// The reference walks a LIFO stack of lattice rectangles per tile: trace the corners nobody
// has written yet, then either split the rectangle in four (corner values differ by more
// than the threshold) or fill it by bilinear interpolation -- overwriting the samples on its
// border, which its neighbours share.  What a rectangle sees at its corners therefore
// depends on which neighbours were processed before it.  That order is fixed by position
// (level-0 rectangles in reverse raster order, children in reverse Z order), and a sample
// can only be overwritten by a leaf of a COARSER level than the rectangles that use it as a
// corner, so the whole walk unrolls by level without changing a single value:
//
//   level k:  1. k_adaptive_points: for every lattice point that first appears at level k
//                decide from the level k-1 decisions whether it is traced now, or already
//                holds the interpolation of an earlier, coarser neighbouring leaf (then that
//                value is what the rectangles of this and deeper levels see), or is untouched;
//                traced points get a camera ray
//             2. the wavefront traces those rays (all recursion levels, shadow rays)
//             3. k_adaptive_decide: every level-k rectangle compares its corners -> split / leaf
//   last:     k_adaptive_fill: every sample takes the interpolation of the LAST leaf (in the
//             reference's processing order) that holds it on its border or inside, or keeps
//             its traced value; k_resolve filters those f64 samples
//
// (tests/adaptive_model.py proves the two formulations equal on random lattices; the parity
// tests compare frames with the sequential restatement in oracle/.)
#ifndef FJGPU_DEV_ADAPTIVE_H
#define FJGPU_DEV_ADAPTIVE_H

struct ATile {                 // lattice geometry of one tile
  int W0, H0;                  // level-0 cells = pixels incl. the filter margin
  int nx;                      // samples per lattice row: div * W0 + 1
  uint32_t cell_offset, sample_offset;
};

__device__ __forceinline__ ATile a_tile(const TileDesc &T, const AdaptiveParams &ap)
{
  ATile t;
  t.W0 = T.xmax - T.xmin + 2 * ap.margin_x;
  t.H0 = T.ymax - T.ymin + 2 * ap.margin_y;
  t.nx = T.nx;
  t.cell_offset = T.cell_offset;
  t.sample_offset = T.sample_offset;
  return t;
}

// split / leaf decisions: one byte per cell, the levels of the whole batch back to back
#define A_NONE 0
#define A_LEAF 1
#define A_SPLIT 2
__device__ __forceinline__ size_t a_cell_index(const AdaptiveParams &ap, const ATile &t, int k, int cx, int cy)
{
  return (size_t) ap.cells0 * ((((size_t) 1 << (2 * k)) - 1) / 3) + ((size_t) t.cell_offset << (2 * k)) +
      (size_t) cy * ((size_t) t.W0 << k) + (size_t) cx;
}

struct ACell { int k, cx, cy; };

// position of a cell in the reference's processing order; LARGER = EARLIER
// (generate_samples pushes the level-0 rectangles in raster order, subdivide_rect pushes
// top-left, top-right, bottom-left, bottom-right; a stack pops in reverse)
__device__ __forceinline__ unsigned long long a_order(const AdaptiveParams &ap, const ATile &t, const ACell &c)
{
  const int X0 = c.cx >> c.k, Y0 = c.cy >> c.k;
  const int lx = (c.cx << (ap.D - c.k)) & (ap.div - 1), ly = (c.cy << (ap.D - c.k)) & (ap.div - 1);
  unsigned long long z = 0;
  for (int b = 0; b < ap.D; b++)
    z |= ((unsigned long long) ((lx >> b) & 1) << (2 * b)) | ((unsigned long long) ((ly >> b) & 1) << (2 * b + 1));
  return ((unsigned long long) (Y0 * t.W0 + X0) << (2 * ap.D)) | z;
}

// the value a sample holds for the rectangles that use it as a corner: its traced colour
// (Color4 -> Vector4, src/fj_renderer.cc:1078-1084) or an earlier leaf's interpolation
__device__ __forceinline__ void a_seen(const uint8_t *pstate, const double *seen, const float *accum, size_t slot, double out[4])
{
  if (pstate[slot] == 2) {
    const double4 s = reinterpret_cast<const double4 *>(seen)[slot];
    out[0] = s.x; out[1] = s.y; out[2] = s.z; out[3] = s.w;
  } else {
    const float4 a = reinterpret_cast<const float4 *>(accum)[slot];
    out[0] = (double) a.x; out[1] = (double) a.y; out[2] = (double) a.z; out[3] = (double) a.w;
  }
}

// interpolate_rect (src/fj_adaptive_grid_sampler.cc:304-330) at one lattice point
__device__ __noinline__ void a_interpolate(const AdaptiveParams &ap, const ATile &t, const ACell &leaf, int fx, int fy,
    const uint8_t *pstate, const double *seen, const float *accum, double out[4])
{
  const int s = ap.div >> leaf.k;
  const int x0 = leaf.cx * s, y0 = leaf.cy * s, x1 = x0 + s, y1 = y0 + s;
  double c00[4], c10[4], c01[4], c11[4];
  a_seen(pstate, seen, accum, (size_t) t.sample_offset + (size_t) y0 * t.nx + x0, c00);
  a_seen(pstate, seen, accum, (size_t) t.sample_offset + (size_t) y0 * t.nx + x1, c10);
  a_seen(pstate, seen, accum, (size_t) t.sample_offset + (size_t) y1 * t.nx + x0, c01);
  a_seen(pstate, seen, accum, (size_t) t.sample_offset + (size_t) y1 * t.nx + x1, c11);
  const double ty = 1. * (fy - y0) / (y1 - y0);
  const double tx = 1. * (fx - x0) / (x1 - x0);
  for (int c = 0; c < 4; c++) {
    const double left = __dadd_rn(__dmul_rn(1 - ty, c00[c]), __dmul_rn(ty, c01[c]));    // Lerp(Vector4): (1 - t) * a + t * b, no fma
    const double right = __dadd_rn(__dmul_rn(1 - ty, c10[c]), __dmul_rn(ty, c11[c]));
    out[c] = __dadd_rn(__dmul_rn(1 - tx, left), __dmul_rn(tx, right));
  }
}

// what lies on one side of a lattice edge at level k1: nothing (tile border), a split cell
// (its children use the edge's midpoint as a corner) or a leaf of level <= k1
__device__ __forceinline__ int a_side(const AdaptiveParams &ap, const ATile &t, const uint8_t *cells, int k1, int cx, int cy, ACell *out)
{
  if (cx < 0 || cy < 0 || cx >= (t.W0 << k1) || cy >= (t.H0 << k1)) return A_NONE;
  for (int j = k1; j >= 0; j--, cx >>= 1, cy >>= 1) {
    const uint8_t s = cells[a_cell_index(ap, t, j, cx, cy)];
    if (s != A_NONE) { out->k = j; out->cx = cx; out->cy = cy; return s; }
  }
  return A_NONE;      // (not reached: every level-0 cell is decided)
}

// ----------------------------------------------------------- k_adaptive_uv
// generate_samples (src/fj_adaptive_grid_sampler.cc:35-97): every lattice sample gets its
// jittered position whether it will be traced or not (the filter weights use it)
__global__ void __launch_bounds__(BLOCK) k_adaptive_uv(AdaptiveParams ap, const TileDesc *tiles, const double *jitter_tab, double *s_uv)
{
  const TileDesc T = tiles[blockIdx.y];
  const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= (uint32_t) T.nx * (uint32_t) T.ny) return;
  const int x = (int) (k % (uint32_t) T.nx), y = (int) (k / (uint32_t) T.nx);
  const int xoffset = (T.xmin - ap.margin_x) * ap.div;
  const int yoffset = (T.ymin - ap.margin_y) * ap.div;
  double u = (x + xoffset) * ap.udelta;
  double v = 1 - (y + yoffset) * ap.vdelta;
  if (ap.jittered) {
    const double u_jitter = jitter_tab[2 * (size_t) k] * ap.jitter;
    const double v_jitter = jitter_tab[2 * (size_t) k + 1] * ap.jitter;
    u += ap.udelta * (u_jitter - .5);
    v += ap.vdelta * (v_jitter - .5);
  }
  const size_t slot = (size_t) T.sample_offset + k;
  s_uv[2 * slot] = u;
  s_uv[2 * slot + 1] = v;
}

// ------------------------------------------------------- k_adaptive_points
// step 1 of a level (see the head of this file).  pstate: 0 untouched, 1 traced, 2 holds an
// earlier leaf's interpolation.
template <bool kMovingCamera>
__global__ void __launch_bounds__(BLOCK) k_adaptive_points(DScene S, AdaptiveParams ap, const TileDesc *tiles, const uint8_t *cells,
    const double *s_uv, const float *s_accum, uint8_t *pstate, double *seen, DRay *rays, DPath *paths, DCounters *cnt)
{
  const TileDesc T = tiles[blockIdx.y];
  const ATile t = a_tile(T, ap);
  const int k = ap.level;
  const int npx = (t.W0 << k) + 1, npy = (t.H0 << k) + 1;
  const uint32_t q = blockIdx.x * BLOCK + threadIdx.x;
  bool trace = false;
  int fx = 0, fy = 0;
  if (q < (uint32_t) npx * (uint32_t) npy) {
    const int px = (int) (q % (uint32_t) npx), py = (int) (q / (uint32_t) npx);
    const int s = ap.div >> k;
    fx = px * s; fy = py * s;
    const int ox = px & 1, oy = py & 1;
    if (k == 0) {
      trace = true;                                    // the corners of the level-0 rectangles
    } else if (ox && oy) {
      trace = cells[a_cell_index(ap, t, k - 1, px >> 1, py >> 1)] == A_SPLIT;      // centre of a split cell
    } else if (ox || oy) {
      // midpoint of an edge of the level k-1 lattice
      ACell a, b;
      int ka, kb;
      if (ox) { ka = a_side(ap, t, cells, k - 1, px >> 1, py / 2 - 1, &a); kb = a_side(ap, t, cells, k - 1, px >> 1, py / 2, &b); }
      else { ka = a_side(ap, t, cells, k - 1, px / 2 - 1, py >> 1, &a); kb = a_side(ap, t, cells, k - 1, px / 2, py >> 1, &b); }
      // (a split cell reported by a_side is always of level k-1: a cell below a split one exists)
      if (ka == A_SPLIT && kb == A_SPLIT) trace = true;
      else if (ka == A_SPLIT || kb == A_SPLIT) {
        const ACell &split = ka == A_SPLIT ? a : b;
        const ACell &other = ka == A_SPLIT ? b : a;
        const int ko = ka == A_SPLIT ? kb : ka;
        if (ko == A_NONE) trace = true;
        else if (a_order(ap, t, other) > a_order(ap, t, split)) {
          // the leaf was interpolated before the split cell's children look at this sample
          const size_t slot = (size_t) t.sample_offset + (size_t) fy * t.nx + fx;
          double val[4];
          a_interpolate(ap, t, other, fx, fy, pstate, seen, s_accum, val);
          reinterpret_cast<double4 *>(seen)[slot] = make_double4(val[0], val[1], val[2], val[3]);
          pstate[slot] = 2;
        } else trace = true;       // traced first; the leaf overwrites it afterwards (k_adaptive_fill)
      }
    }
  }
  // queue the camera rays of the traced samples (one atomic per wave)
  const unsigned long long m = __ballot(trace);
  if (m == 0) return;
  const unsigned lane = __lane_id();
  const int leader = __ffsll((long long) m) - 1;
  uint32_t base = 0;
  if ((int) lane == leader) base = atomicAdd(&cnt->cam_count, (uint32_t) __popcll(m));
  base = __shfl(base, leader);
  if (!trace) return;
  const uint32_t at = base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
  if (at >= ap.ray_capacity) { cnt->overflow = 1; return; }
  const uint32_t kk = (uint32_t) fy * (uint32_t) t.nx + (uint32_t) fx;
  const uint32_t slot = t.sample_offset + kk;
  pstate[slot] = 1;
  camera_ray<kMovingCamera>(S, s_uv[2 * (size_t) slot], s_uv[2 * (size_t) slot + 1], T.id, kk, slot, rays + at, paths + at);
}

// ------------------------------------------------------- k_adaptive_decide
// subdivide_or_interpolate / compare_corners (src/fj_adaptive_grid_sampler.cc:224-262,332-343)
__global__ void __launch_bounds__(BLOCK) k_adaptive_decide(AdaptiveParams ap, const TileDesc *tiles, uint8_t *cells,
    const float *s_accum, const uint8_t *pstate, const double *seen)
{
  const TileDesc T = tiles[blockIdx.y];
  const ATile t = a_tile(T, ap);
  const int k = ap.level;
  const int ncx = t.W0 << k, ncy = t.H0 << k;
  const uint32_t q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= (uint32_t) ncx * (uint32_t) ncy) return;
  const int cx = (int) (q % (uint32_t) ncx), cy = (int) (q / (uint32_t) ncx);
  if (k > 0 && cells[a_cell_index(ap, t, k - 1, cx >> 1, cy >> 1)] != A_SPLIT) return;     // (stays A_NONE)
  uint8_t verdict = A_LEAF;
  if (k < ap.D) {                                       // size >= 2: not at the subdivision limit
    const int s = ap.div >> k;
    const int x0 = cx * s, y0 = cy * s, x1 = x0 + s, y1 = y0 + s;
    double c[4][4];
    a_seen(pstate, seen, s_accum, (size_t) t.sample_offset + (size_t) y0 * t.nx + x0, c[0]);
    a_seen(pstate, seen, s_accum, (size_t) t.sample_offset + (size_t) y0 * t.nx + x1, c[1]);
    a_seen(pstate, seen, s_accum, (size_t) t.sample_offset + (size_t) y1 * t.nx + x0, c[2]);
    a_seen(pstate, seen, s_accum, (size_t) t.sample_offset + (size_t) y1 * t.nx + x1, c[3]);
    for (int ch = 0; ch < 4; ch++) {
      double lo = c[0][ch], hi = c[0][ch];
      for (int i = 1; i < 4; i++) { lo = c[i][ch] < lo ? c[i][ch] : lo; hi = c[i][ch] > hi ? c[i][ch] : hi; }
      if (hi - lo > ap.threshold) verdict = A_SPLIT;
    }
  }
  cells[a_cell_index(ap, t, k, cx, cy)] = verdict;
}

// --------------------------------------------------------- k_adaptive_fill
// the sample values the pixel filter reads: interpolate_rect of the last leaf that covers
// the sample without having it as a corner, else the traced colour
__global__ void __launch_bounds__(BLOCK) k_adaptive_fill(AdaptiveParams ap, const TileDesc *tiles, const uint8_t *cells,
    const float *s_accum, const uint8_t *pstate, const double *seen, double *final_data)
{
  const TileDesc T = tiles[blockIdx.y];
  const ATile t = a_tile(T, ap);
  const uint32_t q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= (uint32_t) T.nx * (uint32_t) T.ny) return;
  const int fx = (int) (q % (uint32_t) T.nx), fy = (int) (q / (uint32_t) T.nx);
  bool have = false;
  ACell best;
  unsigned long long best_order = 0;
  for (int dy = -1; dy <= 0; dy++)
    for (int dx = -1; dx <= 0; dx++) {
      const int fcx = fx + dx, fcy = fy + dy;          // finest cell next to the sample
      if (fcx < 0 || fcy < 0 || fcx >= ap.div * t.W0 || fcy >= ap.div * t.H0) continue;
      ACell leaf;
      leaf.k = -1;
      for (int k = 0; k <= ap.D; k++) {
        const int cx = fcx >> (ap.D - k), cy = fcy >> (ap.D - k);
        if (cells[a_cell_index(ap, t, k, cx, cy)] == A_LEAF) { leaf.k = k; leaf.cx = cx; leaf.cy = cy; break; }
      }
      if (leaf.k < 0) continue;                        // (not reached)
      const int s = ap.div >> leaf.k;
      const int x0 = leaf.cx * s, y0 = leaf.cy * s;
      if ((fx == x0 || fx == x0 + s) && (fy == y0 || fy == y0 + s)) continue;     // a corner keeps its value
      const unsigned long long o = a_order(ap, t, leaf);
      if (!have || o < best_order) { have = true; best = leaf; best_order = o; }
    }
  const size_t slot = (size_t) t.sample_offset + q;
  double val[4];
  if (have) a_interpolate(ap, t, best, fx, fy, pstate, seen, s_accum, val);
  else a_seen(pstate, seen, s_accum, slot, val);
  reinterpret_cast<double4 *>(final_data)[slot] = make_double4(val[0], val[1], val[2], val[3]);
}

#endif
