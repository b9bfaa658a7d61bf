// treelet_probe.cpp -- host-side DESIGN TOOL (not a product path, not the oracle).
//
// How many DEPENDENT wave operations does one ray's fold (bvh_fold, futhark/bvh.fut:61-84) need when a wave serves
// that ray alone?  Compares the pooled kernel's BOX2 operation (<= 32 items, two tree levels per operation) with
// treelet operations: the tree is cut at depths 0, D, 2D, ...; an operation gives G = 2^D lanes to each popped item (a
// treelet root whose box passed), every lane tests both children's boxes of one node of the treelet speculatively, and a
// node counts as reached iff the boxes on its path inside the treelet passed.  Uses the product's host BVH builder and
// lane_core.h, on the rays of the long bounce chains of a frame (they bound a frame's time).
//
// It is also the CPU CHECK of the numbering and the masks (tests/test_host_logic.py): every fold is redone from the 64-byte
// records of the layouts cut at D = 2 .. 5 with the kernel's own decoding (treelet.h: tl_reached / tl_frontier) on emulated
// lanes, and must meet exactly the leaves the plain depth-first walk meets.
//
//   build/treelet_probe <rgbbox|irreg|n (a floor of n x n spheres)> <size> [min_chain]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lane_core.h"
#include "rt_host.hpp"

using namespace rtk;
struct F4 { float x, y, z, w; };

struct Tree {
  std::vector<rt::TravNode> nodes;
  std::vector<int> depth;
  std::vector<F4> sph, col;
};

static bool child_box_hit(const Tree &T, const Ray &r, int c) {
  const rt::TravNode &nd = T.nodes[c];
  return box_hit(r, nd.lo[0], nd.lo[1], nd.lo[2], nd.hi[0], nd.hi[1], nd.hi[2]);
}

// BOX2: items = nodes whose box passed; op pops <= 32, each expands two levels
static int fold_ops_box2(const Tree &T, const Ray &r, std::vector<int> &leaves) {
  std::vector<int> st{0};
  int ops = 0;
  while (!st.empty()) {
    ++ops;
    const int take = std::min<int>(32, st.size());
    std::vector<int> items(st.end() - take, st.end());
    st.resize(st.size() - take);
    for (int it : items) {
      const int kids[2] = {T.nodes[it].left, T.nodes[it].right};
      for (int c : kids) {
        if (c < 0) { leaves.push_back(~c); continue; }
        if (!child_box_hit(T, r, c)) continue;
        const int gk[2] = {T.nodes[c].left, T.nodes[c].right};
        for (int g : gk) {
          if (g < 0) { leaves.push_back(~g); continue; }
          if (child_box_hit(T, r, g)) st.push_back(g);
        }
      }
    }
  }
  return ops;
}

// treelets of depth D (G = 2^D lanes per item, 64 / G items per op)
static int fold_ops_treelet(const Tree &T, const Ray &r, int D, std::vector<int> &leaves, int *max_items) {
  const int per_op = 64 >> D;
  std::vector<int> st{0};
  int ops = 0;
  while (!st.empty()) {
    ++ops;
    *max_items = std::max<int>(*max_items, st.size());
    const int take = std::min<int>(per_op, st.size());
    std::vector<int> items(st.end() - take, st.end());
    st.resize(st.size() - take);
    for (int root : items) {
      // walk the treelet: nodes at relative depth < D are inside; children at relative depth D are exits
      std::vector<std::pair<int, int>> work{{root, 0}};
      while (!work.empty()) {
        auto [nd, rel] = work.back();
        work.pop_back();
        const int kids[2] = {T.nodes[nd].left, T.nodes[nd].right};
        for (int c : kids) {
          if (c < 0) { leaves.push_back(~c); continue; }
          if (!child_box_hit(T, r, c)) continue;
          if (rel + 1 == D) st.push_back(c);
          else work.push_back({c, rel + 1});
        }
      }
    }
  }
  return ops;
}

// The treelet operation as the pooled kernel runs it, on emulated lanes: from the 64-byte records of a layout built
// with treelet depth D (numbering + masks of treelet.h), decoded with the kernel's own helpers.
static int fold_ops_records(const rt::TravLayout &t, const Ray &r, std::vector<int> &leaves) {
  const int D = t.treelet_depth, G = 1 << D, per_op = 64 >> D;
  const int nn = (int)t.nodes.size();
  std::vector<int> st{0};
  int ops = 0;
  while (!st.empty()) {
    ++ops;
    const int take = std::min<int>(per_op, st.size());
    std::vector<int> items(st.end() - take, st.end());
    st.resize(st.size() - take);
    for (int root : items) {
      uint32_t hl = 0, hr = 0, ml[32], mr[32];
      int cl[32], cr[32];
      for (int p = 0; p < G; ++p) {   // every lane of the group reads record root + p, whatever it is
        int idx = root + p;
        if (idx >= nn) idx = 0;
        const float *q = &t.nodes64[16 * (size_t)idx];
        int32_t l8, r8;
        memcpy(&l8, &q[3], 4); memcpy(&r8, &q[7], 4);
        memcpy(&ml[p], &q[11], 4); memcpy(&mr[p], &q[15], 4);
        cl[p] = l8 >> 8; cr[p] = r8 >> 8;
        if (box_hit(r, q[0], q[1], q[2], q[4], q[5], q[6])) hl |= 1u << p;
        if (box_hit(r, q[8], q[9], q[10], q[12], q[13], q[14])) hr |= 1u << p;
      }
      for (int p = 0; p < G; ++p) {
        if (!tl_reached(ml[p], mr[p], p, hl, hr)) continue;
        const bool fr = tl_frontier(ml[p]);
        if (cl[p] < 0) leaves.push_back(~cl[p]);
        else if (fr && (hl >> p & 1u)) st.push_back(cl[p]);
        if (cr[p] < 0) leaves.push_back(~cr[p]);
        else if (fr && (hr >> p & 1u)) st.push_back(cr[p]);
      }
    }
  }
  return ops;
}

int main(int argc, char **argv) {
  const std::string scene = argc > 1 ? argv[1] : "rgbbox";
  const int size = argc > 2 ? atoi(argv[2]) : 1000;
  const int min_chain = argc > 3 ? atoi(argv[3]) : 12;
  rt::SceneDesc sc = scene == "irreg" ? rt::make_floor(100, 600.0f) : scene == "rgbbox" ? rt::make_rgbbox() : rt::make_floor(atoi(scene.c_str()), 6.0f * atoi(scene.c_str()));
  const rt::Lbvh b = rt::build_lbvh(sc.spheres);
  const rt::TravLayout t = rt::make_trav_layout(b);
  rt::TravLayout tl[rtk::kTreeletMaxDepth + 1];
  for (int D = 2; D <= rtk::kTreeletMaxDepth; ++D) tl[D] = rt::make_trav_layout(b, D);
  Tree T;
  T.nodes = t.nodes;
  T.depth.assign(T.nodes.size(), 0);
  for (size_t i = 0; i < T.nodes.size(); ++i)
    for (int c : {T.nodes[i].left, T.nodes[i].right})
      if (c >= 0) T.depth[c] = T.depth[i] + 1;   // breadth-first numbering: parents come first
  for (size_t i = 0; i < t.sph.size() / 4; ++i) {
    T.sph.push_back({t.sph[4 * i], t.sph[4 * i + 1], t.sph[4 * i + 2], t.sph[4 * i + 3]});
    T.col.push_back({t.col[4 * i], t.col[4 * i + 1], t.col[4 * i + 2], t.col[4 * i + 3]});
  }
  const rt::Camera cam = rt::scene_camera(sc, size, size);
  Cam c;
  memcpy(&c, &cam, sizeof(c));
  printf("%s %dx%d: %zu inner nodes, height %d; rays of pixels with >= %d scatters\n", scene.c_str(), size, size, T.nodes.size(), t.height, min_chain);
  unsigned long long folds = 0, ops_b2 = 0, ops_t[7] = {0}, ops_r[7] = {0}, nleaves = 0, chains = 0;
  int max_items[7] = {0};
  // treelet size histogram for D = 4
  for (int row = 0; row < size; ++row)
    for (int col = 0; col < size; ++col) {
      // first pass: chain length
      std::vector<Ray> rays;
      Ray r = primary_ray(c, col, row, size, size);
      float lr = 1, lg = 1, lb = 1;
      int depth = 0;
      int32_t pixel = 0;
      for (;;) {
        rays.push_back(r);
        float best = kTMax;
        int bestj = -1;
        int stack[64], sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
          const rt::TravNode &nd = T.nodes[stack[--sp]];
          if (!box_hit(r, nd.lo[0], nd.lo[1], nd.lo[2], nd.hi[0], nd.hi[1], nd.hi[2])) continue;
          for (int k : {nd.left, nd.right}) {
            if (k < 0) closest_update(sphere_root(r, T.sph[~k].x, T.sph[~k].y, T.sph[~k].z, T.sph[~k].w), ~k, best, bestj);
            else stack[sp++] = k;
          }
        }
        F4 s{0, 0, 0, 1}, cc{0, 0, 0, 0};
        if (bestj >= 0) { s = T.sph[bestj]; cc = T.col[bestj]; }
        if (!finish_ray(r, best, bestj, s.x, s.y, s.z, s.w, cc.x, cc.y, cc.z, cc.w, lr, lg, lb, depth, 50, &pixel)) break;
      }
      if (depth < min_chain) continue;
      ++chains;
      for (const Ray &q : rays) {
        if (!box_hit(q, t.root_lo[0], t.root_lo[1], t.root_lo[2], t.root_hi[0], t.root_hi[1], t.root_hi[2])) continue;
        ++folds;
        std::vector<int> lv;
        ops_b2 += fold_ops_box2(T, q, lv);
        nleaves += lv.size();
        for (int D = 2; D <= 6; ++D) {
          std::vector<int> lv2;
          ops_t[D] += fold_ops_treelet(T, q, D, lv2, &max_items[D]);
          if (lv2.size() != lv.size()) { printf("leaf set mismatch\n"); return 1; }
          if (D <= rtk::kTreeletMaxDepth) {   // the same fold from the records of the layout cut at D
            std::vector<int> lv3, want = lv;
            ops_r[D] += fold_ops_records(tl[D], q, lv3);
            std::sort(lv3.begin(), lv3.end());
            std::sort(want.begin(), want.end());
            if (lv3 != want) { printf("FAIL: treelet records D=%d: leaf set differs (%zu vs %zu leaves)\n", D, lv3.size(), want.size()); return 1; }
          }
        }
      }
    }
  printf("  %llu chains, %llu folds (root box passed), %.1f leaf items per fold\n", chains, folds, (double)nleaves / folds);
  printf("  BOX2 (<= 32 items, 2 levels / op):   %.2f ops per fold\n", (double)ops_b2 / folds);
  for (int D = 2; D <= 6; ++D)
    printf("  treelets D=%d (%2d lanes, %d items / op): %.2f ops per fold (most items on the stack: %d)%s\n", D, 1 << D, 64 >> D, (double)ops_t[D] / folds, max_items[D],
           D <= rtk::kTreeletMaxDepth ? "; same leaf sets from the records" : "");
  printf("treelet check OK\n");
  return 0;
}
