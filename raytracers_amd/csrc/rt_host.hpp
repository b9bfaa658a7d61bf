// rt_host.hpp -- host-side data model of the render path (no HIP in this header).
//
// Canonical data keeps the reference's layout: `bvh = {L: [n]sphere, I: [n-1]inner}`
// (futhark/bvh.fut:24-28) stored struct-of-arrays, sum-type `ptr` as one tagged int32.
// The traversal copy (TravLayout) is DERIVED from it for the kernels; the canonical
// arrays are what parity checks dump (rt_prepared_get_bvh).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "treelet.h"

namespace rt {

// sphere = {pos, colour, radius} (ray.fut:22-24); 7 packed floats.
struct Sphere {
  float px, py, pz;
  float cr, cg, cb;
  float radius;
};
static_assert(sizeof(Sphere) == 28, "Sphere must be 7 packed floats");

// scene = {look_from, look_at, fov, spheres} (ray.fut:171-174)
struct SceneDesc {
  std::vector<Sphere> spheres;
  float look_from[3];
  float look_at[3];
  float fov;
};

// camera = {origin, llc, horizontal, vertical} (ray.fut:88-91); 12 packed floats.
struct Camera {
  float origin[3], llc[3], horizontal[3], vertical[3];
};
static_assert(sizeof(Camera) == 48, "Camera must be 12 packed floats");

// ptr encoding shared with rt_prepared_get_bvh: inner i -> i (>= 0), leaf i -> -2 - i.
inline int32_t ptr_inner(int32_t i) { return i; }
inline int32_t ptr_leaf(int32_t i) { return -2 - i; }
inline bool ptr_is_leaf(int32_t p) { return p <= -2; }
inline int32_t ptr_leaf_index(int32_t p) { return -2 - p; }

struct Lbvh {
  int64_t n = 0;                 // leaves
  std::vector<Sphere> L;         // [n]   sorted by Morton key (stable)
  std::vector<uint32_t> morton;  // [n]   sorted keys
  std::vector<float> bmin, bmax; // [n-1][3]
  std::vector<int32_t> left, right, parent;  // [n-1]
  int sweeps = 0;                // AABB propagation sweeps run (bvh.fut:47)
};

// ---- scenes (ray.fut:176-237) ----
SceneDesc make_rgbbox();
SceneDesc make_floor(int n, float k);   // irreg == (100, 600)

// ---- prepare_scene (ray.fut:241-244) ----
Camera make_camera(const float look_from[3], const float look_at[3], const float vup[3], float vfov, float aspect);
Camera scene_camera(const SceneDesc &sc, int64_t h, int64_t w);
Lbvh build_lbvh(const std::vector<Sphere> &ts);   // bvh_mk sphere_aabb (bvh.fut:30-59)

// ---- traversal copy consumed by the kernels ----
// One 32-byte record per inner node, renumbered treelet by treelet (treelet.h; treelets in order of their
// roots' depth, so the nodes nearest the root form a prefix: the part staged in LDS).  Child references:
// >= 0 -> inner node (traversal numbering); < 0 -> leaf, sphere index = ~ref
// (leaf indices are NOT renumbered: the lowest index wins ties, bvh.fut:61-84).
struct TravNode {
  float lo[3];
  int32_t left;
  float hi[3];
  int32_t right;
};
static_assert(sizeof(TravNode) == 32, "TravNode must be 2 x float4");

struct TravLayout {
  std::vector<TravNode> nodes;       // [n-1] breadth-first
  std::vector<float> sph;            // [n][4] pos.xyz, radius
  std::vector<float> col;            // [n][4] colour.rgb, 1/radius (hit normal's scale, ray.fut:44)
  // 64-byte records for the pooled kernel: a work item is an inner node whose own box already
  // passed, so the record carries what its CHILDREN need: {L.lo.xyz, left << 8} {L.hi.xyz, right << 8}
  // {R.lo.xyz, mask_l} {R.hi.xyz, mask_r}, where L/R are the child's box when the child is an inner node
  // (unused for a leaf child: the reference keeps no leaf boxes, bvh.fut:84).  The references are stored
  // pre-shifted: a work item of the pooled kernel is (reference << 8) | (slot * 4).
  std::vector<float> nodes64;        // [n-1][16]; mask_l / mask_r: the node's treelet masks (treelet.h)
  int treelet_depth = 1;             // levels per treelet (1: every node is its own treelet = numbering by depth)
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};   // box of the root (tested when a ray starts)
  std::vector<int32_t> bfs_of_canon; // canonical inner index -> traversal index
  int height = 0;                    // edges on the longest root -> leaf path
};
TravLayout make_trav_layout(const Lbvh &b, int treelet_depth = 1);

// ---- culling by the best hit so far (lane_core.h: cull_limit) ----
// The scene's side of the proof: whether its rays may be culled at all, and the two constants of the limit
// best + W2 (best^2 + kappa), W2 = max|1 / d_k| * (d.d) * c2.  `ok` needs
//   * every box to contain the boxes of the spheres below it: tree height <= the reference's floor(log2 n) + 2 sweeps
//     (bvh.fut:47 -- a taller tree keeps unconverged upper boxes, which are the reference's answer and may not contain their subtree);
//   * finite spheres with 2^-20 <= radius, |coordinate| + radius <= 2^40;
//   * the scene guard: 2 (R + r_max) <= 2^15 r_min, R the half diagonal of the centres' box -- it keeps a root's error (E1) small
//     against the sphere that produced it, which is what bounds a ray's distance to a sphere by its best root.
// A launch additionally needs its camera inside the same guard (cull_origin_ok).
struct CullConst {
  bool ok = false;
  float c2 = 0.0f, kappa = 0.0f;
  double centre[3] = {0, 0, 0}, reach = 0.0, r_min = 0.0;   // centre of the centres' box; R + r_max
};
CullConst cull_scene_constants(const std::vector<Sphere> &ts, int height);
bool cull_origin_ok(const CullConst &c, const float origin[3]);   // |origin - centre| + R + r_max <= 2^15 r_min

// rows owned by part p of nparts under the cyclic row-tile partition
int64_t part_rows(int64_t h, int32_t rows_per_tile, int32_t part, int32_t nparts);

}  // namespace rt
