// rt_device.hpp -- kernel parameter block and launcher declarations (host <-> .hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lane_core.h"

namespace rtk {

constexpr int kStackPixel = 64;   // pixel_kernel: LDS stack entries per lane (>= any LBVH height)
// pooled_kernel: a wave's LDS region in dwords: hit keys 64 x u64, counters 64, dump 4, then the ray
// table (ray_planes x 64 float4), the box stack and the leaf list
constexpr int kPooledWaveFixedDw = 128 + 64 + 4;
constexpr int pooled_wave_dw(int ray_planes, int capb, int capl) { return kPooledWaveFixedDw + 256 * ray_planes + capb + capl; }

struct KParams {
  // scene (traversal copy; see rt::TravLayout)
  const float4 *nodes;   // [2*(n-1)]  {lo.xyz, left}, {hi.xyz, right}; child >= 0 inner, < 0 ~leaf
  const float4 *nodes64; // [4*(n-1)]  pooled family: {L.lo,left<<8} {L.hi,right<<8} {R.lo,0} {R.hi,0} (children's boxes)
  const float4 *sph;     // [n] {pos.xyz, radius}
  const float4 *col;     // [n] {colour.rgb, 1/radius}
  float root_lo[3], root_hi[3];   // the root's own box
  int n_nodes, n_sph;    // n-1, n
  Cam cam;
  // image + partition
  int w, h;              // full image
  int rows_local;        // rows of this part (packed)
  int rows_per_tile, part, nparts;
  int rpt_log2;          // log2(rows_per_tile) when it is a power of two, else -1
  int tiles_x;           // ceil(w / 8)
  int max_depth;
  int32_t *out;          // [rows_local * w]  (batch launch: frame f at out + f * frame_stride)
  int nframes;           // pooled family: frames rendered by this one launch (>= 1)
  int tpt_log2;          // pooled family: a ticket of the tile queue covers 1 << tpt_log2 consecutive tiles
  int frame_stride;      // int32 elements between consecutive frames' buffers
  const Cam *cams;       // [nframes] per-frame cameras (nullptr: `cam` for every frame)
  unsigned long long *stats;   // [3] rays, box tests, sphere tests (instrumented launches only)
  unsigned long long *trace;   // [waves][8] per-wave timeline (instrumented pooled launch only)
  // persistent family
  unsigned *queue;       // [0] monotonic ticket counter (never reset; see Context::queue_base)  [1] waves that have left (deep-tile pieces)
  unsigned queue_base;   // counter value at which this launch's ticket 0 sits
  int nchunks;           // 8x8 tiles in this part
  int lds_nodes;         // breadth-first node prefix staged in LDS
  int lds_sph;           // sphere prefix staged in LDS
  int smax, lmax;        // per-lane LDS stack / deferred-leaf capacities
  int thr_shade, thr_leaf;   // phase-vote thresholds (lanes)
  // pooled family
  int capb, capl;        // per-wave box-stack / leaf-list capacities (dwords)
  int ray_planes;        // ray table: 3 = {o, a} {1/d} {d} per slot; 2 = without {d} (LEAF then pulls d with ds_bpermute: 1 KB per wave less)
  int prio_depth;        // bounce depth at which a wave raises its issue priority (0: never)
  const int *order;      // [nchunks + 16] ticket -> tile (nullptr: identity), then the first ticket of each cost class
  int box2;              // pooled family: a wave with <= 32 box items runs the two-level BOX2 operation (0: off)
  int deep_class;        // tickets below order[nchunks + deep_class] are "deep" tiles (0: feature off)
  int deep_split;        // log2 of the pieces a deep tile is handed out in (2: four tickets of two rows each; 0: whole)
  int *cost;             // [nchunks] longest bounce chain seen per tile (nullptr: not recorded)
  const float *u_tab;    // [w]  pixel_u(col, w)
  const float *v_tab;    // [h]  pixel_v(row, h), indexed by the FULL image row
};

hipError_t launch_pixel(const KParams &p, bool stats, hipStream_t stream);
// block = 64 * waves_per_wg threads (4, 8 or 16 waves); grid = persistent workgroups
hipError_t launch_persistent(const KParams &p, bool stats, int grid, int waves_per_wg, hipStream_t stream);
size_t persistent_lds_bytes(int lds_nodes, int lds_sph, int smax, int lmax, int waves_per_wg);
hipError_t launch_pooled(const KParams &p, bool stats, int grid, int waves_per_wg, hipStream_t stream);
size_t pooled_lds_bytes(int lds_nodes, int lds_sph, int capb, int capl, int ray_planes, int waves_per_wg);
// prepare_scene on the GPU (bvh_build.hip).  Canonical {L, I} arrays + the traversal copy.
struct GpuBvhOut {
  float *L7;                 // [n][7]
  float *bmin, *bmax;        // [n-1][3]
  int *left, *right, *parent;   // [n-1]  ptr: inner i -> i, leaf i -> -2 - i
  float4 *nodes32;           // [2*(n-1)]
  float4 *nodes64;           // [4*(n-1)]
  float4 *sph, *col;         // [n]
};
void warm_render_kernels();
void warm_build_kernels();
size_t gpu_build_scratch_bytes(int n);   // device scratch one build of n spheres needs
hipError_t gpu_copy_from_pinned(void *dst_dev, const void *src_pinned, size_t bytes, hipStream_t stream);
size_t gpu_build_pinned_bytes();         // host-pinned block the build kernels report through
// `sph7_dev`: the n spheres in device memory; `scratch`, `pinned`: blocks of the sizes above.
// Synchronises the stream; returns the tree height and the root's box.
hipError_t gpu_build_bvh(const float *sph7_dev, int n, const GpuBvhOut &out, char *scratch, char *pinned, hipStream_t stream,
                         int *height_out, float root_lo[3], float root_hi[3]);

hipError_t launch_tile_order(int *cost, int *order, int ntiles, hipStream_t stream);
hipError_t launch_place_part(const int32_t *part, int32_t *image, int w, int rows_local, int rows_per_tile, int part_id,
                             int nparts, hipStream_t stream);

hipError_t launch_place_all(const int32_t *stacked, int32_t *image, int w, int h, int rows_per_tile, int nparts,
                            size_t part_stride, hipStream_t stream);

}  // namespace rtk
