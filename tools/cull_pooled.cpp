// cull_pooled.cpp -- host-side DESIGN EXPERIMENT (not a product path, not the oracle).
//
// would culling (slot, node) items by the ray's best hit so far pay in the POOLED kernel's order --
// a wave-shared LIFO of items of 64 rays, children tested from the parent's record, leaf tests deferred to batches of 64?
// tools/cull_probe.cpp answers for a per-ray depth-first walk; this plays render_kernels.hip's pooled loop (BOX / LEAF /
// SHADE choice, append order, refill in place) with 64 emulated lanes per wave and counts box tests B', sphere tests T' and
// wave operations, with the pixels compared against the un-culled run.
//
//   build/cull_pooled <rgbbox|irreg|floor:n:k> <h> <w> [nwaves abs rel near_first thr_shade look_max]
//
// Culling rule played: in BOX, a passing inner child whose entry parameter tmin > best * (1 + rel) + abs is dropped (best = the
// slot's hit key at the time of the operation); near_first = 1 appends the farther child first so the LIFO pops the nearer.
#include <algorithm>
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

static inline float box_tnear(const Ray &r, const float lo[3], const float hi[3]) {   // box_hit's arithmetic; +inf on a miss
  const float t0x = (lo[0] - r.ox) * r.ix, t1x = (hi[0] - r.ox) * r.ix;
  const float t0y = (lo[1] - r.oy) * r.iy, t1y = (hi[1] - r.oy) * r.iy;
  const float t0z = (lo[2] - r.oz) * r.iz, t1z = (hi[2] - r.oz) * r.iz;
  const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
  float tmin = fmaxf(nx ? t1x : t0x, 0.0f), tmax = fminf(nx ? t0x : t1x, kTMax);
  tmin = fmaxf(ny ? t1y : t0y, tmin); tmax = fminf(ny ? t0y : t1y, tmax);
  tmin = fmaxf(nz ? t1z : t0z, tmin); tmax = fminf(nz ? t0z : t1z, tmax);
  return !(tmax <= tmin) ? tmin : INFINITY;
}

struct Scene {
  std::vector<rt::TravNode> nodes;
  std::vector<uint8_t> tight;   // node's box contains every sphere box of its subtree (a converged box)
  std::vector<F4> sph, col;
  Cam cam;
  int w, h, tiles_x, ntiles;
};

struct Slot {
  Ray r{};
  float lr = 1, lg = 1, lb = 1;
  int depth = 0, pix = -1;
  float best = kTMax;
  int bestj = -1;
  int cnt = 0;
};
struct Item { int sl, ref; };
struct Wave {
  Slot slot[64];
  std::vector<Item> box, leaf;
  unsigned q_next = 0, q_end = 0;
  bool exhausted = false, done = false;
};
struct Counters {
  unsigned long long rays = 0, box = 0, sph = 0, ops[3] = {0, 0, 0}, items[2] = {0, 0}, culled = 0, skipped_untight = 0;
  size_t max_box = 0;
};
struct Knobs { float m_abs = 0, m_rel = 0; int mode = 0, near_first = 0, thr_shade = 40, look_max = 32; float c2 = 0, kappa = 0, a_lo = 1.0f / 64; int rule = 0; };
// the product's limit: best + W2 (best^2 + kappa), W2 = max|1/d_k| a c2 per ray (inf when a < a_lo)
static inline float lim_rule(const Knobs &K, const Ray &r, float best) {
  const float M = fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz));
  const float W2 = r.a >= K.a_lo ? M * r.a * K.c2 : INFINITY;
  return fminf(__builtin_fmaf(W2, __builtin_fmaf(best, best, K.kappa), best), kTMax);
}

static bool step(Wave &W, const Scene &S, const Knobs &K, unsigned &ticket, std::vector<int32_t> &out, Counters &C) {
  const int nbox = (int)W.box.size(), nleaf = (int)W.leaf.size();
  bool leaf_op = nleaf >= 64;
  if (nbox < 64 && nleaf < 64) {
    bool drain = false;
    int live = 0, vacant = 0;
    for (auto &s : W.slot) { live += s.pix >= 0; vacant += s.pix < 0 && !W.exhausted; }
    if (nbox == 0 || (nbox < K.look_max && live + vacant >= K.thr_shade)) {
      int ns = 0;
      for (auto &s : W.slot) ns += (s.pix >= 0 && s.cnt == 0) || (s.pix < 0 && !W.exhausted);
      if (ns >= K.thr_shade || nbox == 0) {
        if (nleaf > 0) drain = true;
        else {
          if (ns == 0) { W.done = true; return false; }
          C.ops[2]++;
          for (int l = 0; l < 64; ++l) {
            Slot &s = W.slot[l];
            bool root = false;
            if (s.pix >= 0 && s.cnt == 0) {
              F4 sp{0, 0, 0, 1}, c{0, 0, 0, 0};
              if (s.bestj >= 0) { sp = S.sph[s.bestj]; c = S.col[s.bestj]; }
              int32_t pixel;
              if (finish_ray(s.r, s.best, s.bestj, sp.x, sp.y, sp.z, sp.w, c.x, c.y, c.z, c.w, s.lr, s.lg, s.lb, s.depth, 50, &pixel)) root = true;
              else { out[s.pix] = pixel; s.pix = -1; }
            }
            while (s.pix < 0 && !W.exhausted) {
              if (W.q_next == W.q_end) {
                const unsigned t = ticket++;
                if (t >= (unsigned)S.ntiles) { W.exhausted = true; break; }
                W.q_next = t * 64u; W.q_end = W.q_next + 64u;
              }
              const unsigned sidx = W.q_next++;
              const int tile = (int)(sidx >> 6), within = (int)(sidx & 63u);
              const int col = (tile % S.tiles_x) * 8 + (within & 7), row = (tile / S.tiles_x) * 8 + (within >> 3);
              if (col < S.w && row < S.h) {
                s.r = primary_ray(S.cam, col, row, S.w, S.h);
                s.lr = s.lg = s.lb = 1.0f; s.depth = 0; s.pix = row * S.w + col;
                root = true;
              }
            }
            if (root) {
              s.best = kTMax; s.bestj = -1;
              C.rays++; C.box++;
              const bool hit = box_tnear(s.r, S.nodes[0].lo, S.nodes[0].hi) < INFINITY;
              s.cnt = hit ? 1 : 0;
              if (hit) W.box.push_back({l, 0});
            }
          }
          return true;
        }
      }
    }
    leaf_op = drain || nbox == 0;
  }
  if (leaf_op) {
    const int n = std::min(nleaf, 64);
    C.ops[1]++; C.items[1] += n;
    for (int k = 0; k < n; ++k) {
      const Item it = W.leaf.back(); W.leaf.pop_back();
      Slot &s = W.slot[it.sl];
      const F4 &sp = S.sph[it.ref];
      C.sph++;
      closest_update(sphere_root(s.r, sp.x, sp.y, sp.z, sp.w), it.ref, s.best, s.bestj);
    }
    return true;
  }
  // BOX: the 64 newest items; left children are appended before right children (near_first: farther before nearer)
  const int n = std::min(nbox, 64);
  C.ops[0]++; C.items[0] += n;
  std::vector<Item> items(W.box.end() - n, W.box.end());
  W.box.resize(nbox - n);
  std::vector<Item> first, second;
  for (int k = 0; k < n; ++k) {
    const Item it = items[n - 1 - k];   // lane k takes the k-th newest
    Slot &s = W.slot[it.sl];
    const rt::TravNode &nd = S.nodes[it.ref];
    const int kids[2] = {nd.left, nd.right};
    float tn[2] = {INFINITY, INFINITY};
    bool push[2] = {false, false};
    const float lim = K.rule ? lim_rule(K, s.r, s.best) : s.best * (1.0f + K.m_rel) + K.m_abs;
    for (int c = 0; c < 2; ++c) {
      if (kids[c] < 0) { W.leaf.push_back({it.sl, ~kids[c]}); continue; }
      C.box++;
      tn[c] = box_tnear(s.r, S.nodes[kids[c]].lo, S.nodes[kids[c]].hi);
      if (tn[c] == INFINITY) continue;
      if (K.mode == 1 && (K.rule ? tn[c] >= lim : tn[c] > lim)) {
        if (S.tight[kids[c]]) { C.culled++; continue; }
        C.skipped_untight++;
      }
      push[c] = true;
    }
    const bool swap = K.near_first && push[0] && push[1] && tn[0] < tn[1];   // left nearer: left goes second (on top)
    if (push[0]) (swap ? second : first).push_back({it.sl, kids[0]});
    if (push[1]) (swap ? first : second).push_back({it.sl, kids[1]});
    s.cnt += (int)push[0] + (int)push[1] - 1;
  }
  // (the kernel appends lane by lane in rank order: lane 0's child lowest)
  for (auto &i : first) W.box.push_back(i);
  for (auto &i : second) W.box.push_back(i);
  C.max_box = std::max(C.max_box, W.box.size());
  return true;
}

int main(int argc, char **argv) {
  const std::string name = argc > 1 ? argv[1] : "rgbbox";
  const int h = argc > 2 ? atoi(argv[2]) : 200, w = argc > 3 ? atoi(argv[3]) : 200;
  const int nwaves = argc > 4 ? atoi(argv[4]) : 4096;
  Knobs K;
  K.m_abs = argc > 5 ? atof(argv[5]) : 0.05f;
  K.m_rel = argc > 6 ? atof(argv[6]) : 1e-3f;
  K.near_first = argc > 7 ? atoi(argv[7]) : 0;
  K.thr_shade = argc > 8 ? atoi(argv[8]) : 40;
  K.look_max = argc > 9 ? atoi(argv[9]) : 32;
  K.rule = argc > 10 ? atoi(argv[10]) : 0;
  rt::SceneDesc sc;
  if (name == "rgbbox") sc = rt::make_rgbbox();
  else if (name == "irreg") sc = rt::make_floor(100, 600.0f);
  else { int n = 0; float k = 0; if (sscanf(name.c_str(), "floor:%d:%f", &n, &k) != 2) return 2; sc = rt::make_floor(n, k); }
  const rt::Lbvh bvh = rt::build_lbvh(sc.spheres);
  const rt::TravLayout tl = rt::make_trav_layout(bvh, 2);
  const rt::Camera camh = rt::scene_camera(sc, h, w);
  Scene S;
  S.nodes = tl.nodes;
  S.sph.resize(bvh.n); S.col.resize(bvh.n);
  std::memcpy(S.sph.data(), tl.sph.data(), sizeof(F4) * bvh.n);
  std::memcpy(S.col.data(), tl.col.data(), sizeof(F4) * bvh.n);
  std::memcpy(&S.cam, &camh, sizeof S.cam);
  S.w = w; S.h = h; S.tiles_x = (w + 7) / 8; S.ntiles = S.tiles_x * ((h + 7) / 8);
  // tight[node]: its box contains the box of every sphere below it (false for the reference's unconverged upper boxes)
  S.tight.assign(S.nodes.size(), 0);
  {
    // children have larger traversal indices only within/after the treelet order -- do it by recursion instead
    std::vector<int> order, st{0};
    while (!st.empty()) { int n = st.back(); st.pop_back(); order.push_back(n); for (int c : {S.nodes[n].left, S.nodes[n].right}) if (c >= 0) st.push_back(c); }
    std::vector<float> lo(3 * S.nodes.size()), hi(3 * S.nodes.size());   // true union of the subtree's sphere boxes
    for (size_t i = order.size(); i-- > 0;) {
      const int n = order[i];
      float l[3] = {INFINITY, INFINITY, INFINITY}, u[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int c : {S.nodes[n].left, S.nodes[n].right}) {
        for (int a = 0; a < 3; ++a) {
          float cl, cu;
          if (c < 0) { const F4 &s = S.sph[~c]; const float p = (&s.x)[a]; cl = p - s.w; cu = p + s.w; }
          else { cl = lo[3 * c + a]; cu = hi[3 * c + a]; }
          l[a] = fminf(l[a], cl); u[a] = fmaxf(u[a], cu);
        }
      }
      bool t = true;
      for (int a = 0; a < 3; ++a) { lo[3 * n + a] = l[a]; hi[3 * n + a] = u[a]; t &= S.nodes[n].lo[a] <= l[a] && S.nodes[n].hi[a] >= u[a]; }
      S.tight[n] = t;
    }
    size_t nt = 0; for (auto t : S.tight) nt += t;
    printf("scene %s %dx%d: %zu inner nodes, %zu with a box that contains its subtree (height %d, sweeps %d)\n", name.c_str(), h, w, S.nodes.size(), nt, tl.height, bvh.sweeps);
  }
  {
    float rmin = INFINITY, rmax = 0, cmax = 0;
    for (auto &sp : S.sph) { rmin = fminf(rmin, sp.w); rmax = fmaxf(rmax, sp.w); for (int a = 0; a < 3; ++a) cmax = fmaxf(cmax, fabsf((&sp.x)[a]) + sp.w); }
    const float c2 = 1.01f * (ldexpf(1.0f, -16) / rmin + ldexpf(1.0f, -22));
    const float c0 = 1.01f * (ldexpf(1.0f, -16) * rmax * rmax / rmin + ldexpf(1.0f, -18) * rmax + ldexpf(1.0f, -24) * cmax + ldexpf(1.0f, -22));
    K.c2 = c2; K.kappa = c0 / (c2 * K.a_lo);
    printf("rule constants: r_min %g r_max %g c_max %g  c2 %g c0 %g kappa %g\n", rmin, rmax, cmax, c2, c0, K.kappa);
  }
  std::vector<int32_t> ref;
  for (int mode = 0; mode < 2; ++mode) {
    K.mode = mode;
    std::vector<int32_t> out((size_t)h * w, -1);
    std::vector<Wave> waves(nwaves);
    Counters C;
    unsigned ticket = 0;
    bool any = true;
    unsigned long long rounds = 0;
    while (any) {
      any = false;
      for (auto &W : waves) if (!W.done) any |= step(W, S, K, ticket, out, C);
      rounds++;
    }
    if (mode == 0) ref = out;
    size_t diff = 0;
    for (size_t i = 0; i < out.size(); ++i) diff += out[i] != ref[i];
    uint32_t cs = 0; for (int32_t p : out) cs = cs * 31u + (uint32_t)p;
    printf("mode %d (abs %g rel %g near_first %d): checksum %08x diff %zu rays %llu box %llu (%.2f/ray) sphere %llu (%.2f/ray) culled %llu untight-kept %llu\n",
           mode, K.m_abs, K.m_rel, K.near_first, cs, diff, C.rays, C.box, (double)C.box / C.rays, C.sph, (double)C.sph / C.rays, C.culled, C.skipped_untight);
    printf("   ops BOX %llu (%.1f items/op) LEAF %llu (%.1f) SHADE %llu   rounds %llu max box stack %zu\n", C.ops[0], (double)C.items[0] / C.ops[0],
           C.ops[1], (double)C.items[1] / C.ops[1], C.ops[2], rounds, C.max_box);
  }
  return 0;
}
