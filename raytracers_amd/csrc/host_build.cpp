// host_build.cpp -- scene generators, camera, LBVH construction and the derived
// traversal layout.  Host-only C++ (compiled with -ffp-contract=off: results must be
// bit-identical to the reference's fp32 arithmetic).
//
// Follows (does not copy) the reference's Futhark program:
//   scenes   futhark/ray.fut:176-237       camera  futhark/ray.fut:93-107, :243-244
//   bvh_mk   futhark/bvh.fut:30-59         morton  futhark/bvh.fut:8-22
//   radix tree  futhark/radixtree.fut:11-72 (Karras 2012, index tie-break on equal keys)
#include "rt_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace rt {
namespace {

struct V3 {
  float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
// prim.fut:22-24: three rounded products, summed left to right.
inline float dot3(V3 a, V3 b) {
  float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
  return (px + py) + pz;
}
inline V3 unit(V3 v) { return (1.0f / std::sqrt(dot3(v, v))) * v; }   // prim.fut:26-28
inline V3 cross3(V3 a, V3 b) {                                         // prim.fut:30-33
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// One grid wall of the rgbbox scene.  `place(a, b)` maps the two grid coordinates to a
// position; coordinates are -k/2 + (k/n)*idx (ray.fut:180-215).
template <class Place>
void add_wall(std::vector<Sphere> &out, int n, float k, float r, float g, float b, Place place) {
  const float fn = static_cast<float>(n);
  const float radius = k / (fn * 2.0f);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      const float ci = -k / 2.0f + (k / fn) * static_cast<float>(i);
      const float cj = -k / 2.0f + (k / fn) * static_cast<float>(j);
      V3 p = place(ci, cj);
      out.push_back(Sphere{p.x, p.y, p.z, r, g, b, radius});
    }
  }
}

}  // namespace

SceneDesc make_rgbbox() {
  SceneDesc sc;
  const int n = 10;
  const float k = 60.0f;
  sc.spheres.reserve(4 * n * n);
  // order matters (it fixes the stable-sort order of coincident spheres): left, mid, right, bottom
  add_wall(sc.spheres, n, k, 1.0f, 0.0f, 0.0f, [&](float y, float z) { return V3{-k / 2.0f, y, z}; });
  add_wall(sc.spheres, n, k, 1.0f, 1.0f, 0.0f, [&](float x, float y) { return V3{x, y, -k / 2.0f}; });
  add_wall(sc.spheres, n, k, 0.0f, 0.0f, 1.0f, [&](float y, float z) { return V3{k / 2.0f, y, z}; });
  add_wall(sc.spheres, n, k, 1.0f, 1.0f, 1.0f, [&](float x, float z) { return V3{x, -k / 2.0f, z}; });
  const float from[3] = {0.0f, 30.0f, 30.0f}, at[3] = {0.0f, -1.0f, -1.0f};
  std::copy(from, from + 3, sc.look_from);
  std::copy(at, at + 3, sc.look_at);
  sc.fov = 75.0f;
  return sc;
}

SceneDesc make_floor(int n, float k) {
  SceneDesc sc;
  sc.spheres.reserve(static_cast<size_t>(n) * n);
  add_wall(sc.spheres, n, k, 1.0f, 1.0f, 1.0f, [](float x, float z) { return V3{x, 0.0f, z}; });
  const float from[3] = {0.0f, 12.0f, 30.0f}, at[3] = {0.0f, 10.0f, -1.0f};
  std::copy(from, from + 3, sc.look_from);
  std::copy(at, at + 3, sc.look_at);
  sc.fov = 75.0f;
  return sc;
}

Camera make_camera(const float look_from[3], const float look_at[3], const float vup[3], float vfov, float aspect) {
  const float pi32 = 3.14159265358979323846f;
  const float theta = vfov * pi32 / 180.0f;
  const float half_h = std::tan(theta / 2.0f);   // float overload == tanf
  const float half_w = aspect * half_h;
  const V3 from{look_from[0], look_from[1], look_from[2]};
  const V3 at{look_at[0], look_at[1], look_at[2]};
  const V3 up{vup[0], vup[1], vup[2]};
  const V3 w = unit(from - at);
  const V3 u = unit(cross3(up, w));
  const V3 v = cross3(w, u);
  const V3 llc = ((from - half_w * u) - half_h * v) - w;
  const V3 hor = (2.0f * half_w) * u;
  const V3 ver = (2.0f * half_h) * v;
  Camera c;
  c.origin[0] = from.x; c.origin[1] = from.y; c.origin[2] = from.z;
  c.llc[0] = llc.x; c.llc[1] = llc.y; c.llc[2] = llc.z;
  c.horizontal[0] = hor.x; c.horizontal[1] = hor.y; c.horizontal[2] = hor.z;
  c.vertical[0] = ver.x; c.vertical[1] = ver.y; c.vertical[2] = ver.z;
  return c;
}

Camera scene_camera(const SceneDesc &sc, int64_t h, int64_t w) {
  const float up[3] = {0.0f, 1.0f, 0.0f};
  return make_camera(sc.look_from, sc.look_at, up, sc.fov, static_cast<float>(w) / static_cast<float>(h));
}

namespace {

// bvh.fut:8-13: spread the low 10 bits of v so that two zero bits follow each bit.
inline uint32_t spread10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

inline uint32_t quantise10(float q) {   // bvh.fut:16-18 + u32.f32
  return static_cast<uint32_t>(std::fmin(std::fmax(q * 1024.0f, 0.0f), 1023.0f));
}

// delta of radixtree.fut:13-21: common-prefix length of keys i and j, with the index as
// tie-break for equal keys, and -1 for j out of range.
struct PrefixLen {
  const uint32_t *key;
  int32_t n;
  int32_t operator()(int32_t i, int32_t j) const {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = key[i], b = key[j];
    if (a == b) {
      const uint32_t x = static_cast<uint32_t>(i) ^ static_cast<uint32_t>(j);
      return 32 + (x ? __builtin_clz(x) : 32);
    }
    return __builtin_clz(a ^ b);
  }
};

}  // namespace

Lbvh build_lbvh(const std::vector<Sphere> &ts) {
  Lbvh out;
  const int64_t n = static_cast<int64_t>(ts.size());
  out.n = n;
  if (n < 2) return out;
  const size_t ni = static_cast<size_t>(n - 1);

  // centre of sphere_aabb: min + 0.5*(max - min) per axis (ray.fut:28-30, prim.fut:47-50)
  std::vector<float> cx(n), cy(n), cz(n);
  const float inf = std::numeric_limits<float>::infinity();
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  for (int64_t i = 0; i < n; ++i) {
    const Sphere &s = ts[i];
    const float mn[3] = {s.px - s.radius, s.py - s.radius, s.pz - s.radius};
    const float mx[3] = {s.px + s.radius, s.py + s.radius, s.pz + s.radius};
    const float c[3] = {mn[0] + 0.5f * (mx[0] - mn[0]), mn[1] + 0.5f * (mx[1] - mn[1]), mn[2] + 0.5f * (mx[2] - mn[2])};
    cx[i] = c[0]; cy[i] = c[1]; cz[i] = c[2];
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::fmin(lo[a], c[a]);
      hi[a] = std::fmax(hi[a], c[a]);
    }
  }
  // Morton keys of the normalised centres (bvh.fut:38-41, :15-22).  A degenerate axis
  // gives 0/0 = NaN, which fmax(NaN*1024, 0) turns into 0 -- as in the reference.
  std::vector<uint32_t> key(n);
  for (int64_t i = 0; i < n; ++i) {
    const float qx = (cx[i] - lo[0]) / (hi[0] - lo[0]);
    const float qy = (cy[i] - lo[1]) / (hi[1] - lo[1]);
    const float qz = (cz[i] - lo[2]) / (hi[2] - lo[2]);
    key[i] = spread10(quantise10(qx)) * 4u + spread10(quantise10(qy)) * 2u + spread10(quantise10(qz));
  }
  // stable sort by key (bvh.fut:43)
  std::vector<int32_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  out.L.resize(n);
  out.morton.resize(n);
  for (int64_t i = 0; i < n; ++i) {
    out.L[i] = ts[order[i]];
    out.morton[i] = key[order[i]];
  }

  // radix tree (radixtree.fut:23-72)
  out.left.assign(ni, 0);
  out.right.assign(ni, 0);
  out.parent.assign(ni, -1);
  const PrefixLen plen{out.morton.data(), static_cast<int32_t>(n)};
  for (int32_t i = 0; i < static_cast<int32_t>(ni); ++i) {
    const int32_t diff = plen(i, i + 1) - plen(i, i - 1);
    const int32_t dir = (diff > 0) - (diff < 0);
    const int32_t floor_len = plen(i, i - dir);
    int32_t span = 2;
    while (plen(i, i + span * dir) > floor_len) span *= 2;
    int32_t len = 0;
    for (int32_t step = span / 2; step > 0; step /= 2)
      if (plen(i, i + (len + step) * dir) > floor_len) len += step;
    const int32_t other = i + len * dir;
    const int32_t node_len = plen(i, other);
    int32_t split = 0;
    for (int32_t q = 1; q <= len; q *= 2) {
      const int32_t step = (len + 2 * q - 1) / (2 * q);
      if (plen(i, i + (split + step) * dir) > node_len) split += step;
    }
    const int32_t gamma = i + split * dir + std::min(dir, 0);
    if (std::min(i, other) == gamma) {
      out.left[i] = ptr_leaf(gamma);
    } else {
      out.left[i] = ptr_inner(gamma);
      out.parent[gamma] = i;
    }
    if (std::max(i, other) == gamma + 1) {
      out.right[i] = ptr_leaf(gamma + 1);
    } else {
      out.right[i] = ptr_inner(gamma + 1);
      out.parent[gamma + 1] = i;
    }
  }

  // AABB propagation: boxes start at {0,0,0}/{0,0,0}; exactly floor(log2 n)+2 Jacobi
  // sweeps, each reading the previous sweep's array (bvh.fut:44-58).  NOT run to a fixed
  // point: with fewer sweeps than the tree height the top boxes stay partial, exactly as
  // in the reference.
  out.sweeps = static_cast<int>(std::log2(static_cast<float>(n))) + 2;
  std::vector<float> cur_min(3 * ni, 0.0f), cur_max(3 * ni, 0.0f), nxt_min(3 * ni), nxt_max(3 * ni);
  auto child_box = [&](int32_t p, float mn[3], float mx[3]) {
    if (ptr_is_leaf(p)) {
      const Sphere &s = out.L[ptr_leaf_index(p)];
      mn[0] = s.px - s.radius; mn[1] = s.py - s.radius; mn[2] = s.pz - s.radius;
      mx[0] = s.px + s.radius; mx[1] = s.py + s.radius; mx[2] = s.pz + s.radius;
    } else {
      for (int a = 0; a < 3; ++a) {
        mn[a] = cur_min[3 * static_cast<size_t>(p) + a];
        mx[a] = cur_max[3 * static_cast<size_t>(p) + a];
      }
    }
  };
  for (int s = 0; s < out.sweeps; ++s) {
    for (size_t i = 0; i < ni; ++i) {
      float amn[3], amx[3], bmn[3], bmx[3];
      child_box(out.left[i], amn, amx);
      child_box(out.right[i], bmn, bmx);
      for (int a = 0; a < 3; ++a) {
        nxt_min[3 * i + a] = std::fmin(amn[a], bmn[a]);   // enclosing, prim.fut:38-45
        nxt_max[3 * i + a] = std::fmax(amx[a], bmx[a]);
      }
    }
    cur_min.swap(nxt_min);
    cur_max.swap(nxt_max);
  }
  out.bmin = std::move(cur_min);
  out.bmax = std::move(cur_max);
  return out;
}

TravLayout make_trav_layout(const Lbvh &b, int treelet_depth) {
  TravLayout t;
  const int64_t n = b.n;
  if (n < 2) return t;
  const size_t ni = static_cast<size_t>(n - 1);
  const int D = std::min(std::max(treelet_depth, 1), rtk::kTreeletMaxDepth);
  t.treelet_depth = D;
  // depth of every inner node (canonical node 0 is the root), and on which side of its parent it hangs
  std::vector<int32_t> depth(ni, 0);
  std::vector<uint8_t> is_right(ni, 0);
  int height = 1;
  {
    std::vector<int32_t> todo{0};
    for (size_t head = 0; head < todo.size(); ++head) {
      const int32_t c = todo[head];
      height = std::max(height, depth[c] + 1);
      const int32_t kids[2] = {b.left[c], b.right[c]};
      for (int k = 0; k < 2; ++k) {
        if (ptr_is_leaf(kids[k])) continue;
        depth[kids[k]] = depth[c] + 1;
        is_right[kids[k]] = static_cast<uint8_t>(k);
        todo.push_back(kids[k]);
      }
    }
  }
  // Treelet-major numbering (treelet.h): treelets in (depth of the root, canonical index of the root) order, the nodes
  // of a treelet by heap index.  D == 1: every node is its own treelet = numbering by (depth, canonical index).
  std::vector<int32_t> root_of(ni), heap_of(ni);
  std::vector<uint32_t> occ(ni, 0u);
  for (size_t c = 0; c < ni; ++c) {
    int32_t a = static_cast<int32_t>(c), path = 0;
    const int rel = depth[c] % D;
    for (int s = 0; s < rel; ++s) {   // path bits, the step nearest the root ends up most significant
      path |= static_cast<int32_t>(is_right[a]) << s;
      a = b.parent[a];
    }
    root_of[c] = a;
    heap_of[c] = (1 << rel) - 1 + path;
    occ[static_cast<size_t>(a)] |= 1u << heap_of[c];
  }
  std::vector<int32_t> by_depth(ni);
  for (size_t c = 0; c < ni; ++c) by_depth[c] = static_cast<int32_t>(c);
  std::stable_sort(by_depth.begin(), by_depth.end(), [&](int32_t x, int32_t y) { return depth[x] < depth[y]; });
  std::vector<int32_t> base(ni, 0);
  {
    int32_t run = 0;
    for (int32_t c : by_depth)
      if (depth[c] % D == 0) {
        base[c] = run;
        run += rtk::tl_popc(occ[c]);
      }
  }
  std::vector<int32_t> bfs(ni);      // traversal index -> canonical index
  t.bfs_of_canon.assign(ni, -1);
  for (size_t c = 0; c < ni; ++c) {
    const int32_t ti = base[root_of[c]] + rtk::tl_pos(occ[root_of[c]], heap_of[c]);
    t.bfs_of_canon[c] = ti;
    bfs[static_cast<size_t>(ti)] = static_cast<int32_t>(c);
  }
  t.height = height;
  t.nodes.resize(ni);
  auto ref = [&](int32_t p) -> int32_t { return ptr_is_leaf(p) ? ~ptr_leaf_index(p) : t.bfs_of_canon[p]; };
  for (size_t ti = 0; ti < ni; ++ti) {
    const int32_t c = bfs[ti];
    TravNode &nd = t.nodes[ti];
    for (int a = 0; a < 3; ++a) {
      nd.lo[a] = b.bmin[3 * static_cast<size_t>(c) + a];
      nd.hi[a] = b.bmax[3 * static_cast<size_t>(c) + a];
    }
    nd.left = ref(b.left[c]);
    nd.right = ref(b.right[c]);
  }
  t.nodes64.assign(16 * ni, 0.0f);
  for (size_t ti = 0; ti < ni; ++ti) {
    const TravNode &nd = t.nodes[ti];
    float *q = &t.nodes64[16 * ti];
    const int32_t kid[2] = {nd.left, nd.right};
    for (int k = 0; k < 2; ++k) {
      if (kid[k] < 0) continue;   // leaf child: no box
      const TravNode &ch = t.nodes[static_cast<size_t>(kid[k])];
      for (int a = 0; a < 3; ++a) {
        q[8 * k + a] = ch.lo[a];
        q[8 * k + 4 + a] = ch.hi[a];
      }
    }
    // child references pre-shifted: a pooled work item is (reference << 8) | (slot * 4)
    const int32_t l8 = static_cast<int32_t>(static_cast<uint32_t>(nd.left) << 8), r8 = static_cast<int32_t>(static_cast<uint32_t>(nd.right) << 8);
    std::memcpy(&q[3], &l8, 4);
    std::memcpy(&q[7], &r8, 4);
    const int32_t c = bfs[ti];
    const rtk::TlMasks m = rtk::tl_masks(occ[root_of[c]], heap_of[c], D);
    std::memcpy(&q[11], &m.l, 4);
    std::memcpy(&q[15], &m.r, 4);
  }
  for (int a = 0; a < 3; ++a) {
    t.root_lo[a] = t.nodes[0].lo[a];
    t.root_hi[a] = t.nodes[0].hi[a];
  }
  t.sph.resize(4 * static_cast<size_t>(n));
  t.col.resize(4 * static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) {
    const Sphere &s = b.L[i];
    float *p = &t.sph[4 * static_cast<size_t>(i)], *c = &t.col[4 * static_cast<size_t>(i)];
    p[0] = s.px; p[1] = s.py; p[2] = s.pz; p[3] = s.radius;
    c[0] = s.cr; c[1] = s.cg; c[2] = s.cb; c[3] = 1.0f / s.radius;
  }
  return t;
}

int64_t part_rows(int64_t h, int32_t rows_per_tile, int32_t part, int32_t nparts) {
  if (h <= 0 || rows_per_tile <= 0 || nparts <= 0 || part < 0 || part >= nparts) return 0;
  const int64_t ntiles = (h + rows_per_tile - 1) / rows_per_tile;
  int64_t rows = 0;
  for (int64_t t = part; t < ntiles; t += nparts) rows += std::min<int64_t>(rows_per_tile, h - t * rows_per_tile);
  return rows;
}

// ---- culling constants (rt_host.hpp; the derivation is DESIGN.md 3.4) ----
CullConst cull_scene_constants(const std::vector<Sphere> &ts, int height) {
  CullConst c;
  const size_t n = ts.size();
  if (n < 2) return c;
  const int sweeps = static_cast<int>(log2f(static_cast<float>(n))) + 2;   // bvh.fut:47
  if (height > sweeps) return c;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  double r_min = INFINITY, r_max = 0.0, c_max = 0.0;
  for (const Sphere &s : ts) {
    const double p[3] = {s.px, s.py, s.pz}, r = s.radius;
    if (!(r >= 0x1p-20) || !std::isfinite(r)) return c;
    r_min = std::min(r_min, r);
    r_max = std::max(r_max, r);
    for (int a = 0; a < 3; ++a) {
      if (!std::isfinite(p[a])) return c;
      lo[a] = std::min(lo[a], p[a]);
      hi[a] = std::max(hi[a], p[a]);
      c_max = std::max(c_max, std::fabs(p[a]) + r);
    }
  }
  if (c_max > 0x1p40) return c;
  double diag2 = 0.0;
  for (int a = 0; a < 3; ++a) {
    c.centre[a] = 0.5 * (lo[a] + hi[a]);
    diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
  }
  c.reach = 0.5 * std::sqrt(diag2) * (1.0 + 0x1p-20) + r_max;
  c.r_min = r_min;
  if (2.0 * c.reach > 0x1p15 * r_min) return c;
  // rho' <= c2' a best^2 + c0' with c2' = 2^-16 / r_min, c0' = 2^-16 r_max^2 / r_min + 2^-18 r_max + 2^-24 C_max; both carry 2^-21 for the
  // rounding of tmin and of the limit itself, and 1 % for the roundings of W2 and of the two fused multiply-adds
  const double c2 = 1.01 * (0x1p-16 / r_min + 0x1p-21);
  const double c0 = 1.01 * (0x1p-16 * r_max * r_max / r_min + 0x1p-18 * r_max + 0x1p-24 * c_max + 0x1p-21);
  c.c2 = std::nextafter(static_cast<float>(c2), INFINITY);
  c.kappa = std::nextafter(static_cast<float>(c0 / (c2 * 0.015625)), INFINITY);   // / kCullALo: W2 kappa >= max|1 / d_k| c0 for every a >= 2^-6
  c.ok = std::isfinite(c.c2) && std::isfinite(c.kappa);
  return c;
}

bool cull_origin_ok(const CullConst &c, const float origin[3]) {
  if (!c.ok) return false;
  double d2 = 0.0;
  for (int a = 0; a < 3; ++a) {
    if (!std::isfinite(origin[a])) return false;
    d2 += (origin[a] - c.centre[a]) * (origin[a] - c.centre[a]);
  }
  return std::sqrt(d2) * (1.0 + 0x1p-20) + c.reach <= 0x1p15 * c.r_min;
}

}  // namespace rt
