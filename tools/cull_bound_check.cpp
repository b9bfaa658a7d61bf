// cull_bound_check.cpp -- CPU check of the inequalities the CULL instantiations rest on (lane_core.h: cull_limit;
// DESIGN.md 3.4).  Not a product path, not the oracle: a hammer for a proof.  In the CPU test suite (a short run).
//
// For random and adversarial (ray, sphere) pairs inside the guards of rt::cull_scene_constants it checks, with the product's own
// binary32 code for the roots, the box entry parameter and the limit, and __float128 for the truth:
//   (E1)  every root g sphere_root could return (root1 and root2, whatever their sign):  | |o + g d - p|^2 - r^2 | <= 2^-18 (D^2 + r^2)
//   (S)   the safety property itself: the sphere's OWN binary32 box (pos -+ r, the smallest box any ancestor can have around it)
//         is never culled by a limit computed from that sphere's own root:  NOT ( tmin(box) >= cull_limit(g, W2, kappa) )
//         -- cull_limit grows with `best` and an ancestor's tmin is no larger, so this is the worst case of "a subtree is dropped
//         although one of its spheres has a root <= best".
//   build/cull_bound_check [millions of samples = 20] [seed = 1] [weaken = 1]
// weaken < 1 scales the limit's margin down: the checker must then FIND violations of (S) (the test suite runs it once that way).
#include <omp.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "lane_core.h"
#include "rt_host.hpp"

using namespace rtk;
typedef __float128 q128;

static inline float box_tnear(const Ray &r, const float lo[3], const float hi[3]) {   // box_hit_clamped's arithmetic: tmin, or +inf on a miss
  const float t0x = (lo[0] - r.ox) * r.ix, t1x = (hi[0] - r.ox) * r.ix;
  const float t0y = (lo[1] - r.oy) * r.iy, t1y = (hi[1] - r.oy) * r.iy;
  const float t0z = (lo[2] - r.oz) * r.iz, t1z = (hi[2] - r.oz) * r.iz;
  const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
  float tmin = fmaxf(nx ? t1x : t0x, 0.0f), tmax = fminf(nx ? t0x : t1x, kTMax);
  tmin = fmaxf(ny ? t1y : t0y, tmin); tmax = fminf(ny ? t0y : t1y, tmax);
  tmin = fmaxf(nz ? t1z : t0z, tmin); tmax = fminf(nz ? t0z : t1z, tmax);
  return !(tmax <= tmin) ? tmin : INFINITY;
}

// both roots as sphere_root computes them (ray.fut:32-51), no acceptance test
static inline bool roots(const Ray &r, float px, float py, float pz, float rad, float g[2]) {
  const float ocx = r.ox - px, ocy = r.oy - py, ocz = r.oz - pz;
  const float b = dot3(ocx, ocy, ocz, r.dx, r.dy, r.dz);
  const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - rad * rad;
  const float disc = b * b - r.a * c;
  if (disc <= 0.0f) return false;
  const float sq = sqrtf(disc);
  g[0] = (-b - sq) / r.a;
  g[1] = (-b + sq) / r.a;
  return true;
}

int main(int argc, char **argv) {
  const long long total = (long long)((argc > 1 ? atof(argv[1]) : 20.0) * 1e6);
  const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
  const float weaken = argc > 3 ? (float)atof(argv[3]) : 1.0f;
  unsigned long long n_roots = 0, n_e1 = 0, n_s = 0, n_hits = 0, n_cullable = 0;
  double worst_e1 = 0.0;
#pragma omp parallel reduction(+ : n_roots, n_e1, n_s, n_hits, n_cullable) reduction(max : worst_e1)
  {
    std::mt19937_64 rng(seed * 7919u + 104729u * (unsigned)omp_get_thread_num());
    std::uniform_real_distribution<double> U(0.0, 1.0);
    auto sym = [&](double s) { return (2.0 * U(rng) - 1.0) * s; };
#pragma omp for schedule(dynamic, 1024)
    for (long long it = 0; it < total / 64; ++it) {
      // a "scene": radii in [r_min, r_max], coordinates up to ext; one sphere of it and a batch of rays
      const double scale = std::exp2(std::floor(sym(12.0)));                 // 2^-12 .. 2^12
      const double r_min = scale * (0.05 + U(rng)), r_max = r_min * (U(rng) < 0.5 ? 1.0 : 1.0 + 30.0 * U(rng));
      const double reach_max = 0x1p14 * r_min;                               // 2 reach <= 2^15 r_min
      const double ext = std::min(reach_max * 0.5, r_max * std::exp2(10.0 * U(rng)));
      std::vector<rt::Sphere> two(2);
      const float rad = (float)(U(rng) < 0.5 ? r_min : r_min + (r_max - r_min) * U(rng));
      two[0] = rt::Sphere{(float)sym(ext), (float)sym(ext), (float)sym(ext), 1, 1, 1, rad};
      two[1] = rt::Sphere{(float)sym(ext), (float)sym(ext), (float)sym(ext), 1, 1, 1, (float)r_min};
      if (U(rng) < 0.3) two[1].radius = (float)r_max;
      const rt::CullConst cc = rt::cull_scene_constants(two, 0);
      if (!cc.ok) continue;
      const float px = two[0].px, py = two[0].py, pz = two[0].pz;
      const float lo[3] = {px - rad, py - rad, pz - rad}, hi[3] = {px + rad, py + rad, pz + rad};   // sphere_aabb, ray.fut:28-30
      for (int k = 0; k < 64; ++k) {
        // origin: anywhere the guard admits (often far: the error grows with D^2), or near / on / inside the sphere
        const int mode = (int)(U(rng) * 6.0);
        double od = mode == 0 ? rad * (1.0 + 4.0 * U(rng)) : mode == 1 ? rad * std::exp2(14.0 * U(rng)) : mode == 2 ? rad * U(rng) : ext * (0.5 + 2.0 * U(rng));
        double dir[3] = {sym(1), sym(1), sym(1)};
        if (U(rng) < 0.25) dir[(int)(U(rng) * 3.0) % 3] *= std::exp2(-40.0 * U(rng));   // nearly axis-parallel
        const double dn = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]) + 1e-300;
        const float org[3] = {(float)(px - od * dir[0] / dn), (float)(py - od * dir[1] / dn), (float)(pz - od * dir[2] / dn)};
        if (!rt::cull_origin_ok(cc, org)) continue;
        // direction: at the sphere, with an offset from the centre of 0 .. a bit more than r -- grazing hits on purpose
        const double off = rad * (U(rng) < 0.5 ? 1.0 + sym(1e-3) * U(rng) : 1.2 * U(rng));
        double perp[3] = {sym(1), sym(1), sym(1)};
        const double pd = (perp[0] * dir[0] + perp[1] * dir[1] + perp[2] * dir[2]) / (dn * dn);
        for (int a = 0; a < 3; ++a) perp[a] -= pd * dir[a];
        const double pn = std::sqrt(perp[0] * perp[0] + perp[1] * perp[1] + perp[2] * perp[2]) + 1e-300;
        const double len = U(rng) < 0.6 ? 1.0 : std::exp2(sym(3.0));           // |d|: reflected rays ~1, primary rays some other length
        Ray r;
        r.ox = org[0]; r.oy = org[1]; r.oz = org[2];
        const double tgt[3] = {px + off * perp[0] / pn, py + off * perp[1] / pn, pz + off * perp[2] / pn};
        double dd[3] = {tgt[0] - org[0], tgt[1] - org[1], tgt[2] - org[2]};
        const double ddn = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-300;
        r.dx = (float)(dd[0] / ddn * len); r.dy = (float)(dd[1] / ddn * len); r.dz = (float)(dd[2] / ddn * len);
        ray_derive(r);
        float g[2];
        if (!roots(r, px, py, pz, rad, g)) continue;
        n_hits++;
        const float w2 = cull_weight(r, cc.c2 * weaken);
        const float tn = box_tnear(r, lo, hi);
        const q128 ocx = (q128)r.ox - px, ocy = (q128)r.oy - py, ocz = (q128)r.oz - pz;
        const q128 D2 = ocx * ocx + ocy * ocy + ocz * ocz, r2 = (q128)rad * rad;
        for (int j = 0; j < 2; ++j) {
          if (!std::isfinite(g[j])) continue;
          n_roots++;
          const q128 qx = ocx + (q128)g[j] * r.dx, qy = ocy + (q128)g[j] * r.dy, qz = ocz + (q128)g[j] * r.dz;
          q128 F = qx * qx + qy * qy + qz * qz - r2;
          if (F < 0) F = -F;
          const q128 bound = (D2 + r2) * (q128)0x1p-18;
          const double ratio = (double)(F / bound);
          worst_e1 = std::max(worst_e1, ratio);
          if (F > bound) n_e1++;
          // (S): only roots the fold could accept matter (g > 0.1), and only a box the ray enters can be culled at all
          if (g[j] > kEps && std::isfinite(tn)) {
            const float lim = cull_limit(g[j], w2, cc.kappa);
            if (std::isfinite(w2)) n_cullable++;
            if (tn >= lim) n_s++;
          }
        }
      }
    }
  }
  printf("cull_bound_check: %llu ray/sphere pairs with roots, %llu roots checked (%llu with a finite weight): E1 violations %llu (worst |F| / bound %.4f), safety violations %llu\n",
         n_hits, n_roots, n_cullable, n_e1, worst_e1, n_s);
  return (n_e1 || n_s) ? 1 : 0;
}
