// cull_probe.cpp -- host-side DESIGN EXPERIMENT (not a product path, not the oracle).
// How many box tests would culling by the current best hit save under the pooled kernel's
// level-synchronous order, and does it keep the pixels?  (SURVEY.md 8a "order freedom": culling
// is NOT automatically bit-identical.)
//   g++ -O2 -fopenmp -ffp-contract=off -std=c++17 -Iraytracers_amd/csrc -Iinclude -o build/cull_probe \
//       tools/cull_probe.cpp raytracers_amd/csrc/host_build.cpp
//   build/cull_probe <rgbbox|irreg> <h> <w> [margin_abs margin_rel]
#include <omp.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "lane_core.h"
#include "rt_host.hpp"
using namespace rtk;
struct F4 { float x, y, z, w; };

// box entry parameter with the kernel's arithmetic (box_hit), or +inf on a miss
static inline float box_tnear(const Ray &r, const float lo[3], const float hi[3]) {
  const float t0x = (lo[0] - r.ox) * r.ix, t1x = (hi[0] - r.ox) * r.ix;
  const float t0y = (lo[1] - r.oy) * r.iy, t1y = (hi[1] - r.oy) * r.iy;
  const float t0z = (lo[2] - r.oz) * r.iz, t1z = (hi[2] - r.oz) * r.iz;
  const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
  float tmin = fmaxf(nx ? t1x : t0x, 0.0f), tmax = fminf(nx ? t0x : t1x, kTMax);
  tmin = fmaxf(ny ? t1y : t0y, tmin); tmax = fminf(ny ? t0y : t1y, tmax);
  tmin = fmaxf(nz ? t1z : t0z, tmin); tmax = fminf(nz ? t0z : t1z, tmax);
  return !(tmax <= tmin) ? tmin : INFINITY;
}

int main(int argc, char **argv) {
  const std::string name = argc > 1 ? argv[1] : "rgbbox";
  const int h = argc > 2 ? atoi(argv[2]) : 200, w = argc > 3 ? atoi(argv[3]) : 200;
  const float m_abs = argc > 4 ? atof(argv[4]) : 0.0f, m_rel = argc > 5 ? atof(argv[5]) : 0.0f;
  rt::SceneDesc sc = name == "rgbbox" ? rt::make_rgbbox() : rt::make_floor(100, 600.0f);
  const rt::Lbvh bvh = rt::build_lbvh(sc.spheres);
  const rt::TravLayout tl = rt::make_trav_layout(bvh);
  const rt::Camera camh = rt::scene_camera(sc, h, w);
  Cam cam; std::memcpy(&cam, &camh, sizeof cam);
  const std::vector<rt::TravNode> &nodes = tl.nodes;
  const F4 *sph = (const F4 *)tl.sph.data(), *col = (const F4 *)tl.col.data();
  for (int mode = 0; mode < 3; ++mode) {   // 0: no culling  1: level-synchronous + cull  2: near-first DFS + cull
    unsigned long long rays = 0, box = 0, sphs = 0, diff = 0;
    std::vector<int32_t> out((size_t)h * w);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays, box, sphs)
    for (int row = 0; row < h; ++row)
      for (int c0 = 0; c0 < w; ++c0) {
        Ray r = primary_ray(cam, c0, row, w, h);
        float lr = 1, lg = 1, lb = 1; int depth = 0; int32_t pixel = 0;
        for (;;) {
          float best = kTMax; int bestj = -1;
          rays++;
          std::vector<int> cur, nxt, leaves;
          box++;
          if (box_tnear(r, nodes[0].lo, nodes[0].hi) < INFINITY) cur.push_back(0);
          if (mode < 2) {
            while (!cur.empty() || !leaves.empty()) {
              for (int lf : leaves) { sphs++; closest_update(sphere_root(r, sph[lf].x, sph[lf].y, sph[lf].z, sph[lf].w), lf, best, bestj); }
              leaves.clear(); nxt.clear();
              const float lim = best + m_abs + m_rel * best;
              for (int n : cur) {
                const int kids[2] = {nodes[n].left, nodes[n].right};
                for (int k : kids) {
                  if (k < 0) { leaves.push_back(~k); continue; }
                  box++;
                  const float tn = box_tnear(r, nodes[k].lo, nodes[k].hi);
                  if (tn == INFINITY) continue;
                  if (mode == 1 && tn > lim) continue;
                  nxt.push_back(k);
                }
              }
              cur.swap(nxt);
            }
          } else {
            struct E { int n; float t; };
            std::vector<E> st; if (!cur.empty()) st.push_back({0, 0.f});
            while (!st.empty()) {
              E e = st.back(); st.pop_back();
              if (e.t > best + m_abs + m_rel * best) continue;
              const int kids[2] = {nodes[e.n].left, nodes[e.n].right};
              E pass[2]; int np = 0;
              for (int k : kids) {
                if (k < 0) { sphs++; closest_update(sphere_root(r, sph[~k].x, sph[~k].y, sph[~k].z, sph[~k].w), ~k, best, bestj); continue; }
                box++;
                const float tn = box_tnear(r, nodes[k].lo, nodes[k].hi);
                if (tn == INFINITY) continue;
                pass[np++] = {k, tn};
              }
              if (np == 2 && pass[0].t < pass[1].t) std::swap(pass[0], pass[1]);   // nearer on top
              for (int i = 0; i < np; ++i) st.push_back(pass[i]);
            }
          }
          F4 s{0, 0, 0, 1}, c{0, 0, 0, 0};
          if (bestj >= 0) { s = sph[bestj]; c = col[bestj]; }
          if (!finish_ray(r, best, bestj, s.x, s.y, s.z, s.w, c.x, c.y, c.z, c.w, lr, lg, lb, depth, 50, &pixel)) break;
        }
        out[(size_t)row * w + c0] = pixel;
      }
    static std::vector<int32_t> ref;
    if (mode == 0) ref = out;
    for (size_t i = 0; i < out.size(); ++i) diff += out[i] != ref[i];
    printf("mode %d: rays %llu box %llu (%.1f/ray) sphere %llu (%.2f/ray) differing pixels %llu\n", mode, rays, box,
           (double)box / rays, sphs, (double)sphs / rays, diff);
  }
  return 0;
}
