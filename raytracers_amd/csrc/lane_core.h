// lane_core.h -- per-lane arithmetic of the render hot path, shared by every kernel
// family (render_kernels.hip) and by the host-side wave simulator (tools/wavesim.cpp,
// a design tool that executes the same per-lane code with 64 emulated lanes).
//
// Parity contract: IEEE binary32, no FMA contraction (build with
// -ffp-contract=off), correctly rounded / and sqrt, fmaxf/fminf NaN semantics, and the
// reference's operation order.  Each function cites the Futhark lines it follows.
//
// What is NOT taken from the reference is the traversal ORDER.  bvh_fold
// (futhark/bvh.fut:61-84) walks parent pointers and meets leaves in increasing index
// order; here a leaf is tested iff every ancestor box passes aabb_hit with the FIXED
// interval (0, 1e9) (ray.fut:77 closes over the outer t_max), and the winner is the
// smallest accepted root, ties to the lowest leaf index -- the same (j, t) the fold
// returns, for any visiting order.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtk {

constexpr float kTMax = 1000000000.0f;   // ray.fut:130
constexpr float kEps = 0.1f;             // scene_epsilon, ray.fut:3
constexpr float kNoHit = __builtin_inff();

struct Cam {   // camera (ray.fut:88-91) as 12 floats
  float ox, oy, oz, lx, ly, lz, hx, hy, hz, vx, vy, vz;
};

struct Ray {
  float ox, oy, oz;
  float dx, dy, dz;
  float ix, iy, iz;   // 1/d per axis: aabb_hit's invD (ray.fut:55) depends on the ray only
  float a;            // dot d d: sphere_hit's `a` (ray.fut:34) and norm's argument (prim.fut:26)
};

RT_HD float dot3(float ax, float ay, float az, float bx, float by, float bz) {   // prim.fut:22-24
  float px = ax * bx, py = ay * by, pz = az * bz;
  return (px + py) + pz;
}

RT_HD void ray_derive(Ray &r) {
  r.ix = 1.0f / r.dx;
  r.iy = 1.0f / r.dy;
  r.iz = 1.0f / r.dz;
  r.a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
}

// trace_ray + get_ray (ray.fut:150-154, :109-114).  `col` = i, `row` = image row from
// the top; the reference evaluates pixel (j = row) at v = (height - j) / height.
RT_HD float pixel_u(int col, int width) { return (float)col / (float)width; }
RT_HD float pixel_v(int row, int height) { return (float)(height - row) / (float)height; }

// get_ray for precomputed (u, v): the pooled kernel reads u and v from per-column / per-row
// tables the host fills with pixel_u / pixel_v (two correctly rounded divisions per pixel
// become two loads; the values are the same bits).
RT_HD void primary_dir_uv(const Cam &c, float u, float v, Ray &r) {   // origin + direction only
  r.ox = c.ox; r.oy = c.oy; r.oz = c.oz;
  r.dx = ((c.lx + u * c.hx) + v * c.vx) - c.ox;
  r.dy = ((c.ly + u * c.hy) + v * c.vy) - c.oy;
  r.dz = ((c.lz + u * c.hz) + v * c.vz) - c.oz;
}

RT_HD Ray primary_ray(const Cam &c, int col, int row, int width, int height) {
  const float u = pixel_u(col, width);
  const float v = pixel_v(row, height);
  Ray r;
  r.ox = c.ox; r.oy = c.oy; r.oz = c.oz;
  r.dx = ((c.lx + u * c.hx) + v * c.vx) - c.ox;
  r.dy = ((c.ly + u * c.hy) + v * c.vy) - c.oy;
  r.dz = ((c.lz + u * c.hz) + v * c.vz) - c.oz;
  ray_derive(r);
  return r;
}

// aabb_hit (ray.fut:53-70) with tmin0 = 0, tmax0 = 1e9.
//
// The reference tests `tmax_k <= tmin_k -> false` after each of the x, y, z slabs.  Only
// the last test is evaluated here, which is the same predicate: fmax/fmin return the
// non-NaN operand, so starting from the finite (0, 1e9) no tmin_k/tmax_k is ever NaN,
// tmin_1 <= tmin_2 <= tmin_3 and tmax_1 >= tmax_2 >= tmax_3; hence tmax_3 > tmin_3 implies
// tmax_k > tmin_k for k = 1, 2, and a failed earlier test implies tmax_3 <= tmin_3.
// (The per-axis swap on `invD < 0` must stay a select: with a zero direction component,
// 0 * inf = NaN bounds arise that min/max-based swapping would treat differently.)
typedef float rt_f2 __attribute__((vector_size(8)));
// `tclamp`: the interval's upper end.  kTMax gives aabb_hit itself (box_hit below); the CULL instantiations of the pooled kernel pass
// min(kTMax, a proven lower bound on every sphere root the box's subtree could still contribute) -- see cull_limit.
RT_HD bool box_hit_clamped(const Ray &r, float lox, float loy, float loz, float hix, float hiy, float hiz, float tclamp) {
  // x and y as a pair: two packed subtracts + two packed multiplies (v_pk_add/mul_f32 round
  // each lane exactly like the scalar instructions)
  const rt_f2 o2 = {r.ox, r.oy}, i2 = {r.ix, r.iy};
  const rt_f2 lo2 = {lox, loy}, hi2 = {hix, hiy};
  const rt_f2 t0 = (lo2 - o2) * i2, t1 = (hi2 - o2) * i2;
  const float t0z = (loz - r.oz) * r.iz, t1z = (hiz - r.oz) * r.iz;
  const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
  float tmin = fmaxf(nx ? t1[0] : t0[0], 0.0f);
  float tmax = fminf(nx ? t0[0] : t1[0], tclamp);
  tmin = fmaxf(ny ? t1[1] : t0[1], tmin);
  tmax = fminf(ny ? t0[1] : t1[1], tmax);
  tmin = fmaxf(nz ? t1z : t0z, tmin);
  tmax = fminf(nz ? t0z : t1z, tmax);
  return !(tmax <= tmin);
}
RT_HD bool box_hit(const Ray &r, float lox, float loy, float loz, float hix, float hiy, float hiz) {
  return box_hit_clamped(r, lox, loy, loz, hix, hiy, hiz, kTMax);
}

// ---- culling by the best hit so far (the CULL instantiations of the pooled kernel) -------------------------
// The reference's fold tests every leaf whose ancestors' boxes pass with the FIXED interval (0, 1e9) (ray.fut:77) -- it never
// narrows the interval.  Only its RESULT is the contract: the smallest accepted root, ties to the lowest leaf.  A subtree may
// therefore be skipped when every root any of its spheres could produce is proven LARGER than a root already found.
//
// The bound (proof and constants: DESIGN.md 3.4; tools/cull_bound_check.cpp hammers the two inequalities it rests on).  For a
// ray (o, d) and a sphere (p, r) let g be ANY root sphere_root computes in binary32, P = o + g d the exact point at that
// parameter, D = |o - p|, a = d.d.  Then
//     | |P - p|^2 - r^2 |  <=  2^-18 (D^2 + r^2)                                                     (E1)
// i.e. P lies within rho = 2^-18 (D^2 / r + r) of the sphere, hence inside any box that contains the sphere's binary32 box
// (pos -+ r, rounded: off by at most 2^-24 max|coordinate|) widened by rho' = rho + 2^-24 C_max on every side, hence
//     g  >=  tmin(box) (1 - 3.01 * 2^-24)  -  rho' max_k |1 / d_k|                                   (E2)
// with tmin the entry parameter box_hit computes.  With g <= best (only such a root could matter), D <= 2 (a best^2 + r^2)^(1/2)
// under the scene guard of cull_scene_constants, and the right-hand side exceeds `best` whenever
//     tmin  >=  best + W2 (best^2 + kappa),    W2 = max_k |1 / d_k| * a * c2
// c2 and kappa being per-scene constants.  A box is tested against min(kTMax, that limit) instead of kTMax: one instruction
// stream, and a box that fails only against the limit is a subtree whose every root is > best -- it cannot win or tie.
// Requires every box to contain the boxes of the spheres below it (tree height <= the reference's number of sweeps).
constexpr float kCullALo = 1.0f / 64.0f, kCullAHi = 1048576.0f;   // a = d.d outside [2^-6, 2^20]: the ray is not culled
RT_HD float cull_weight(const Ray &r, float c2) {                 // W2 of a ray (inf: never cull)
  const float m = fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz));
  return (r.a >= kCullALo && r.a <= kCullAHi) ? m * r.a * c2 : kNoHit;
}
RT_HD float cull_limit(float best, float w2, float kappa) {       // the interval's upper end for a slot whose best root is `best`
  return fminf(__builtin_fmaf(w2, __builtin_fmaf(best, best, kappa), best), kTMax);
}

// The root closest_hit would accept for this sphere if its running t_max were large
// (ray.fut:32-51 called with t_min = 0.1, :79): root1 if root1 > 0.1, else root2 if
// root2 > 0.1, else none.  (root2 >= root1, so "root1 >= t_max, try root2" can never
// succeed; the caller's `g < best` test is then exactly the reference's `temp < t_max`.)
RT_HD float sphere_root(const Ray &r, float px, float py, float pz, float rad) {
  const float ocx = r.ox - px, ocy = r.oy - py, ocz = r.oz - pz;
  const float b = dot3(ocx, ocy, ocz, r.dx, r.dy, r.dz);
  const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - rad * rad;
  const float disc = b * b - r.a * c;
  if (disc <= 0.0f) return kNoHit;
  const float sq = sqrtf(disc);
  float t = (-b - sq) / r.a;
  if (!(t > kEps)) {
    t = (-b + sq) / r.a;
    if (!(t > kEps)) return kNoHit;
  }
  return t;
}

// closest_hit's accumulator update (ray.fut:78-81) made order-independent.
RT_HD void closest_update(float g, int idx, float &best, int &bestj) {
  if (g < best || (g == best && idx < bestj)) {
    best = g;
    bestj = idx;
  }
}

RT_HD int32_t pack_pixel(float r, float g, float b) {   // colour_to_pixel, ray.fut:158-162
  const int32_t ir = (int32_t)(255.99f * r);
  const int32_t ig = (int32_t)(255.99f * g);
  const int32_t ib = (int32_t)(255.99f * b);
  return (ir << 16) | (ig << 8) | ib;
}

// sphere_root plus what the later re-intersection needs to know: *near_root is set when the
// fold took root2 (root1 <= 0.1) although root1 > 0 -- exactly the case in which
// `sphere_hit s r 0.0 (t+1)` (ray.fut:83-85) returns root1 instead of the fold's t.
RT_HD float sphere_root_flag(const Ray &r, float px, float py, float pz, float rad, bool *near_root) {
  const float ocx = r.ox - px, ocy = r.oy - py, ocz = r.oz - pz;
  const float b = dot3(ocx, ocy, ocz, r.dx, r.dy, r.dz);
  const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - rad * rad;
  const float disc = b * b - r.a * c;
  *near_root = false;
  if (disc <= 0.0f) return kNoHit;
  const float sq = sqrtf(disc);
  float t = (-b - sq) / r.a;
  if (!(t > kEps)) {
    *near_root = t > 0.0f;
    t = (-b + sq) / r.a;
    if (!(t > kEps)) return kNoHit;
  }
  return t;
}

// The literal re-intersection of the winning sphere, `sphere_hit s r 0.0 (best+1)`
// (ray.fut:83-85, :32-51): may pick the OTHER root than the fold did, or none.
RT_HD bool rehit_full(const Ray &r, float best, float spx, float spy, float spz, float srad, float *t_out) {
  const float ocx = r.ox - spx, ocy = r.oy - spy, ocz = r.oz - spz;
  const float b = dot3(ocx, ocy, ocz, r.dx, r.dy, r.dz);
  const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - srad * srad;
  const float disc = b * b - r.a * c;
  bool have = false;
  float t = 0.0f;
  if (!(disc <= 0.0f)) {
    const float sq = sqrtf(disc);
    const float lim = best + 1.0f;
    t = (-b - sq) / r.a;
    have = (t < lim) && (t > 0.0f);
    if (!have) {
      t = (-b + sq) / r.a;
      have = (t < lim) && (t > 0.0f);
    }
  }
  *t_out = t;
  return have;
}

// Shortcut for the same call when the fold's accepted root `best` is known not to be
// displaced: with near_root clear, root1 (if it was the fold's root) or root2 passes
// `0 < t < best + 1` iff best + 1 > best, and the re-intersection returns t = best.
// Returns false when the caller must run rehit_full instead.
RT_HD bool rehit_is_best(float best, bool near_root) { return !near_root && (best + 1.0f > best); }

// The rest of one ray_colour iteration (ray.fut:126-148, :119-124): scatter or terminate.
// `have`/`t`: result of the re-intersection.  Returns true when the pixel continues with the
// scattered ray: r.o/r.d, light and depth are updated (and, if DERIVE, r's derived fields;
// otherwise the caller runs ray_derive); false when the pixel is finished and *pixel holds
// its packed colour.  sph = {pos.xyz}, col = colour, inv_rad = 1.0f / radius as an IEEE
// division (tabulated by the host).
template <bool DERIVE>
RT_HD bool shade_ray(Ray &r, bool have, float t, float spx, float spy, float spz, float scr, float scg, float scb,
                     float inv_rad, float &lr, float &lg, float &lb, int &depth, int max_depth, int32_t *pixel) {
  const float inv_norm = 1.0f / sqrtf(r.a);   // normalise r.dir = scale (1/norm d) d
  if (have) {
    // hit record (ray.fut:40-46)
    const float hpx = r.ox + t * r.dx, hpy = r.oy + t * r.dy, hpz = r.oz + t * r.dz;
    const float nx = inv_rad * (hpx - spx), ny = inv_rad * (hpy - spy), nz = inv_rad * (hpz - spz);
    // scatter (ray.fut:119-124), reflect (ray.fut:116-117)
    const float ux = inv_norm * r.dx, uy = inv_norm * r.dy, uz = inv_norm * r.dz;
    const float k = 2.0f * dot3(ux, uy, uz, nx, ny, nz);
    const float rx = ux - k * nx, ry = uy - k * ny, rz = uz - k * nz;
    if (dot3(rx, ry, rz, nx, ny, nz) > 0.0f && depth + 1 < max_depth) {
      r.ox = hpx; r.oy = hpy; r.oz = hpz;
      r.dx = rx; r.dy = ry; r.dz = rz;
      if (DERIVE) ray_derive(r);
      lr = lr * scr; lg = lg * scg; lb = lb * scb;
      depth = depth + 1;
      return true;
    }
    // absorbed, or the bounce budget is spent: colour = light * (0,0,0)
    *pixel = pack_pixel(lr * 0.0f, lg * 0.0f, lb * 0.0f);
    return false;
  }
  // miss: sky gradient (ray.fut:140-148)
  const float uy = inv_norm * r.dy;
  const float tt = 0.5f * (uy + 1.0f);
  const float w = 1.0f - tt;
  const float sr = w * 1.0f + tt * 0.5f, sg = w * 1.0f + tt * 0.7f, sb = w * 1.0f + tt * 1.0f;
  *pixel = pack_pixel(lr * sr, lg * sg, lb * sb);
  return false;
}

// One whole iteration of ray_colour's loop body AFTER the fold: re-intersect the winning
// sphere with (0.0, best+1), then scatter or terminate.
RT_HD bool finish_ray(Ray &r, float best, int bestj, float spx, float spy, float spz, float srad,
                      float scr, float scg, float scb, float inv_rad, float &lr, float &lg, float &lb,
                      int &depth, int max_depth, int32_t *pixel) {
  float t = 0.0f;
  const bool have = bestj >= 0 && rehit_full(r, best, spx, spy, spz, srad, &t);
  return shade_ray<true>(r, have, t, spx, spy, spz, scr, scg, scb, inv_rad, lr, lg, lb, depth, max_depth, pixel);
}

}  // namespace rtk
