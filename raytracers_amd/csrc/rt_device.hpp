// rt_device.hpp -- kernel parameter block and launcher declarations (host <-> .hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lane_core.h"

namespace rtk {

constexpr int kTraceWords = 16;   // u64 per wave of the instrumented pooled launch's timeline (rt_render_trace)
constexpr int kStackPixel = 64;   // pixel_kernel: LDS stack entries per lane (>= any LBVH height)
// pooled_kernel: a wave's LDS region in dwords: hit keys 64 x u64, counters 64, dump 4, then the ray
// table (ray_planes x 64 float4), the box stack and the leaf list
constexpr int kPooledWaveFixedDw = 128 + 64 + 4;
constexpr int pooled_wave_dw(int ray_planes, int capb, int capl) { return kPooledWaveFixedDw + 256 * ray_planes + capb + capl; }
// the shape of twenty waves per CU: the LDS box stack a wave can have there (2 000 - 196 - 512 - 192 dwords, whole batches), and the tiny one that
// option stack_cap selects to make the SPILL kernels spill all the time (testing)
constexpr int kSpillCapb = 1088, kSpillCapbTest = 192;

// ---- the tile queue of the persistent families --------------------------------------------------
// Tickets are drawn with returning device-scope atomics; one word saturates at ~88 draws per microsecond
// (MI355X_MICROARCH.md, "dequeue"), which a 4000x4000 frame's 250 000 tiles reach, and 4096 waves drawing their first
// ticket keep it busy for ~46 us of a 0.4 ms frame.  So (a) a wave's FIRST ticket is its own number -- no atomic --
// and (b) the queue has nshards in {1, 8} counters, each on its own 128-byte line; counter s is the home of the
// workgroups with blockIdx % 8 == s (the workgroups one XCD runs).  Two layouts:
//   * the counters TAKE TURNS over one queue (interleave, the default): counter s hands out tickets s, s + 8, ... of
//     the one adaptive order over all tiles -- the deepest tiles stay first, spread over all XCDs;
//   * STRIPS: counter s owns the s-th vertical strip of tile columns with its own segment of the order table, so an
//     XCD's L2 (4 MiB, not shared with the others) serves one strip's part of the scene.
// A wave whose home counter has run dry takes tickets from the next ones (every wave fails at most once on every
// counter before it leaves); the last wave to leave zeroes the counters for the next launch on the stream.
constexpr int kQueueStride = 32;                                // dwords between the shards' counters
constexpr int kMaxShards = 8;
constexpr int kQueueExit = kQueueStride * kMaxShards;           // dword index of the counter of waves that have left
constexpr int kQueueDwords = kQueueStride * (kMaxShards + 1);
// Behind the ticket -> tile table of a view (order[0 .. ntiles)) sits one table per shard: the first position
// (relative to the shard's segment) of each of the 8 cost classes, then the shard's tile count.
constexpr int kOrderTableDw = 16;
constexpr int order_table_ints(int ntiles) { return ntiles + kOrderTableDw * kMaxShards; }

struct Shard {
  int x0, sw;      // tile columns [x0, x0 + sw)
  int seg;         // first position of the shard's segment in the order table
  int ntiles;
};
// (the number of shards is 1 << ns_log2: 1 or 8)
__host__ __device__ inline Shard shard_of(int s, int ns_log2, int tiles_x, int tiles_y) {
  const int x0 = (s * tiles_x) >> ns_log2, x1 = ((s + 1) * tiles_x) >> ns_log2;
  return Shard{x0, x1 - x0, x0 * tiles_y, (x1 - x0) * tiles_y};
}
// k-th tile of a strip, row-major within the strip -> index in the image's tile grid (one shard: the identity)
__host__ __device__ inline int shard_tile(const Shard &sh, int k, int tiles_x) {
  const int r = k / sh.sw;
  return r * tiles_x + sh.x0 + (k - r * sh.sw);
}
// Tickets of a shard with `npos` positions (tiles x frames), whose first n_deep positions are deep tiles (waves that
// draw one do not refill until it is finished): the first n_split <= n_deep of them (the deepest tiles of the order) are
// handed out in 2^ds pieces each, the other deep tiles one per ticket (several to one wave would be traced one after
// the other), the rest 2^tpt positions per ticket.
__host__ __device__ inline unsigned shard_tickets(unsigned npos, unsigned n_split, unsigned n_deep, int ds, int tpt) {
  return (n_split << ds) + (n_deep - n_split) + ((npos - n_deep + (1u << tpt) - 1u) >> tpt);
}
// ticket t < shard_tickets(...) -> the pixels it covers, as [q_next, q_end) in units of (position * 64 + pixel in tile)
struct TicketSpan { unsigned q_next, q_end; unsigned cls; };   // cls: pixel tickets only (kPxClasses - 1 for tile tickets)

// ---- pixel tickets (round 5): an ORDERED single frame's queue runs over a list of the view's PIXELS sorted by the length of
// their bounce chains (recorded by the view's first frame, px_order kernels in render_kernels.hip), longest first, cut into
// kPxClasses segments by chain length.  A ticket of class k covers 1 << kPxLog2[k] consecutive pixels of the list: the longest
// chains go out one pixel per ticket (traced by the solo loop), the next ones 8 / 16 / 32 per ticket to waves that do not refill
// while they trace them (a wave's bounce cadence grows with the rays it carries: ~8 us per bounce with 16 rays, ~17 with 40), the
// bulk 64 per ticket.  The table's header (kPxHdrInts ints, device memory, written by px_header_kernel):
//   [0 .. 5] first list position of class k (k = 5: the number of pixels)   [8 .. 13] first ticket of class k (k = 5: all tickets)
//   [6] the cuts that were used (diagnostics)   [7] != 0: the bulk's tickets are zipped
constexpr int kPxClasses = 5;
constexpr int kPxHdrInts = 16;
__host__ __device__ inline int px_log2(int cls) { return cls == 0 ? 0 : cls + 2; }   // 1, 8, 16, 32, 64 pixels
__host__ __device__ inline TicketSpan px_ticket_span(unsigned t, const int *hdr) {
  int k = 0;
  while (k + 1 < kPxClasses && t >= (unsigned)hdr[8 + k + 1]) ++k;
  TicketSpan sp;
  sp.cls = (unsigned)k;
  unsigned j = t - (unsigned)hdr[8 + k];
  if (k == kPxClasses - 1 && hdr[7] != 0) {
    // the bulk ZIPPED (hdr[7], the default): its tickets alternately from the long end and from the short end of the segment.  Sorted
    // straight through, the list ends in one- and two-ray pixels -- nothing but SHADE operations, 60 of whose 250 instructions are
    // division and square-root expansions, on every wave of the chip at once; zipped, that work runs next to the box tests of the
    // longer bulk chains all the way, and the queue ends in the middle of the bulk (chains of ~3 rays: a short drain).  irreg 1000 x 1000
    // 0.274 -> 0.249 ms, 700 x 700 -7.5 %, rgbbox 500 x 500 -7 %, 1000 x 1000 +-0 (profiles/r05/exp/e13)
    const unsigned n = (unsigned)hdr[8 + k + 1] - (unsigned)hdr[8 + k];
    j = (j & 1u) ? n - 1u - (j >> 1) : (j >> 1);
  }
  sp.q_next = (unsigned)hdr[k] + (j << px_log2(k));
  const unsigned end = sp.q_next + (1u << px_log2(k)), seg_end = (unsigned)hdr[k + 1];
  sp.q_end = end < seg_end ? end : seg_end;
  return sp;
}
// the header for a list whose classes start at the positions pos[0 .. kPxClasses] (pos[0] = 0, non-decreasing)
__host__ __device__ inline void px_make_header(const int *pos, int *hdr) {
  int tick = 0;
  for (int k = 0; k < kPxClasses; ++k) {
    hdr[k] = pos[k];
    hdr[8 + k] = tick;
    tick += (pos[k + 1] - pos[k] + (1 << px_log2(k)) - 1) >> px_log2(k);
  }
  hdr[kPxClasses] = pos[kPxClasses];
  hdr[8 + kPxClasses] = tick;
  hdr[6] = hdr[7] = hdr[14] = hdr[15] = 0;
}
__host__ __device__ inline TicketSpan ticket_span(unsigned t, unsigned seg, unsigned npos, unsigned n_split, unsigned n_deep, int ds, int tpt) {
  TicketSpan sp;
  sp.cls = kPxClasses - 1;
  if (t < (n_split << ds)) {
    const unsigned piece = 64u >> ds;
    sp.q_next = (seg + (t >> ds)) * 64u + (t & ((1u << ds) - 1u)) * piece;
    sp.q_end = sp.q_next + piece;
    sp.cls = kPxClasses - 1;
    return sp;
  }
  t -= n_split << ds;
  if (t < n_deep - n_split) {
    sp.q_next = (seg + n_split + t) * 64u;
    sp.q_end = sp.q_next + 64u;
    return sp;
  }
  const unsigned k0 = n_deep + ((t - (n_deep - n_split)) << tpt);
  const unsigned k1 = k0 + (1u << tpt) < npos ? k0 + (1u << tpt) : npos;
  sp.q_next = (seg + k0) * 64u;
  sp.q_end = (seg + k1) * 64u;
  return sp;
}

// Drawing a ticket (wave-uniform; shared by the kernel and tools/queue_check.cpp, which plays the protocol on the
// CPU).  `fetch_add(shard)` returns the shard's counter before its increment.
struct QueueConst {
  int ns_log2, tiles_x, tiles_y, nframes;
  int interleave;            // 1: the shards are not strips but every (1 << ns_log2)-th ticket of ONE queue over all tiles (geometry and tables of a single shard)
  int ds, tpt;               // deep_split, tpt_log2
  int cap_log2;              // the pieces of split tiles may occupy one in 2^cap_log2 of the waves
  int ntiles;                // all shards' tiles: the class tables sit at order[ntiles + kOrderTableDw * shard]
  const int *order;          // nullptr: no tables
  int deep_class;            // 0: no deep tiles
  const int *px;             // pixel tickets: the list's header (px_ticket_span); nullptr: tile tickets.  (One queue: interleave, or a single counter.)
  unsigned home_waves;       // waves whose home is one shard (total waves >> ns_log2)
  unsigned q_static;         // tickets [0, q_static) of every shard are the home waves' first tickets: never drawn from the counter
};
// A wave's queue state is one word (it lives in an SGPR across the whole render loop):
// bits 0-2 the shard being drawn from, bit 7 "the first draw is still to come", bits 8-15 the shards seen dry.
constexpr unsigned kQueueFirst = 0x80u;
__host__ __device__ inline unsigned queue_state_init(int home_shard, bool static_first) {
  return (unsigned)home_shard | (static_first ? kQueueFirst : 0u);
}
__host__ __device__ inline int queue_shard(unsigned state) { return (int)(state & 7u); }
// the shard whose geometry (strip, segment, class table) the ticket in hand belongs to
__host__ __device__ inline int queue_geo_shard(const QueueConst &c, unsigned state) { return c.interleave ? 0 : queue_shard(state); }
__host__ __device__ inline int queue_geo_log2(const QueueConst &c) { return c.interleave ? 0 : c.ns_log2; }
// deep tiles at the head of a shard's segment (0: none / feature off)
__host__ __device__ inline int queue_ndeep(const QueueConst &c, int shard) {
  return (c.order != nullptr && c.deep_class > 0) ? c.order[c.ntiles + kOrderTableDw * shard + c.deep_class] : 0;
}
// deep tiles handed out in pieces: the deepest ones, as many as one in 2^cap_log2 of the waves the queue (strips: the
// shard) serves can take
__host__ __device__ inline unsigned queue_nsplit(const QueueConst &c, int ndeep) {
  const int cap = (int)((c.interleave ? c.home_waves << c.ns_log2 : c.home_waves) >> (c.cap_log2 + c.ds));
  return c.ds > 0 ? (unsigned)(ndeep < cap ? ndeep : cap) : 0u;
}
template <class FetchAdd>
__host__ __device__ inline bool queue_draw(unsigned &state, const QueueConst &c, unsigned wave_rank, FetchAdd &&fetch_add, TicketSpan *sp) {
  const unsigned all = ((1u << (1 << c.ns_log2)) - 1u) << 8;
  for (;;) {   // until a shard yields a ticket or all have run dry
    const int sh = queue_shard(state);
    const int geo = c.interleave ? 0 : sh;
    const Shard s = shard_of(geo, queue_geo_log2(c), c.tiles_x, c.tiles_y);
    const unsigned npos = (unsigned)s.ntiles * (unsigned)c.nframes;
    const int ndeep = queue_ndeep(c, geo);               // (<= the shard's tiles: a position of its class table)
    const unsigned n_split = queue_nsplit(c, ndeep);
    unsigned tickets = c.px != nullptr ? (unsigned)c.px[8 + kPxClasses] : shard_tickets(npos, n_split, (unsigned)ndeep, c.ds, c.tpt);
    // interleaved: this counter's tickets are the queue's tickets sh, sh + n, sh + 2 n, ... (n counters)
    if (c.interleave) tickets = (tickets + (1u << c.ns_log2) - 1u - (unsigned)sh) >> c.ns_log2;
    unsigned t = 0;
    bool got = false;
    if (state & kQueueFirst) {
      state &= ~kQueueFirst;
      t = wave_rank;
      got = t < tickets;           // (not: the shard has no tickets beyond the static ones either)
    } else if (tickets > c.q_static) {
      t = fetch_add(sh) + c.q_static;
      got = t < tickets;
    }
    if (got) {
      if (c.interleave) t = (t << c.ns_log2) + (unsigned)sh;
      *sp = c.px != nullptr ? px_ticket_span(t, c.px) : ticket_span(t, (unsigned)s.seg, npos, n_split, (unsigned)ndeep, c.ds, c.tpt);
      return true;
    }
    state |= 0x100u << sh;
    if ((state & all) == all) return false;
    unsigned nx = (unsigned)sh;
    do nx = (nx + 1u) & ((1u << c.ns_log2) - 1u); while ((state >> (8 + nx)) & 1u);
    state = (state & ~7u) | nx;
  }
}

struct KParams {
  // scene (traversal copy; see rt::TravLayout)
  const float4 *nodes;   // [2*(n-1)]  {lo.xyz, left}, {hi.xyz, right}; child >= 0 inner, < 0 ~leaf
  const float4 *nodes64; // [4*(n-1)]  pooled family: {L.lo,left<<8} {L.hi,right<<8} {R.lo,mask_l} {R.hi,mask_r} (children's boxes; treelet masks)
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
  int tiles_y;           // ceil(rows_local / 8) (nchunks = tiles_x * tiles_y; on the host: no integer division in the kernel)
  int max_depth;
  int32_t *out;          // [rows_local * w]  (batch launch: frame f at out + f * frame_stride)
  int out_skip;          // 0: the part's rows are PACKED at out.  In place (rt_render_part_inplace): out = the full image's row
                         // part * rows_per_tile, and the part's row tile k lies k * out_skip = k * (nparts - 1) * rows_per_tile * w
                         // elements further than in the packed layout -- every pixel is stored where the assembled image has it
                         // (that image may be another device's memory: the stores then are the framebuffer exchange)
  int nframes;           // pooled family: frames rendered by this one launch (>= 1)
  int tpt_log2;          // pooled family: a ticket of the tile queue covers 1 << tpt_log2 consecutive tiles
  int frame_stride;      // int32 elements between consecutive frames' buffers
  const Cam *cams;       // [nframes] per-frame cameras (nullptr: `cam` for every frame)
  unsigned long long *stats;   // [3] rays, box tests, sphere tests (instrumented launches only)
  unsigned long long *trace;   // [waves][kTraceWords] per-wave timeline (instrumented pooled launch only)
  // persistent family
  unsigned *queue;       // [kQueueDwords] ticket counter of shard s at [kQueueStride * s], waves that have left at [kQueueExit]; all zero between launches
  int nshards;           // 1, or 8 (pooled family, one frame per launch): one ticket counter per XCD ...
  int interleave;        // ... 0: and one strip of tile columns per counter; 1: the counters take turns over ONE queue (order table of a single shard)
  int static_first;      // pooled family: a wave's first ticket is its own number among its home counter's waves (no atomic: no ramp at launch)
  int nchunks;           // 8x8 tiles in this part
  int lds_nodes;         // breadth-first node prefix staged in LDS
  int lds_sph;           // sphere prefix staged in LDS
  int smax, lmax;        // per-lane LDS stack / deferred-leaf capacities
  int thr_shade, thr_leaf;   // phase-vote thresholds (lanes)
  // pooled family
  int capb, capl;        // per-wave box-stack / leaf-list capacities (dwords)
  int ray_planes;        // ray table: 3 = {o, a} {1/d} {d} per slot; 2 = without {d} (LEAF then pulls d with ds_bpermute: 1 KB per wave less)
  int prio_depth;        // bounce depth at which a wave raises its issue priority (0: never)
  const int *order;      // [order_table_ints(nchunks)] position -> tile (nullptr: the strips in row-major order), then the shards' class tables
  int donate;            // pooled family: > 0 = the DONATE instantiation: a wave that cannot refill and holds at most `donate` rays gives them to waiting sibling waves of its workgroup
  int cold;              // pooled family: > 0 = the COLD instantiation: a wave that cannot refill hands its last `cold` rays to the solo loop from inside the pooled loop
  int look_max;          // pooled family: a wave with this many box items or more does not look at finished folds / vacant slots (64: it looks whenever it has less than a full batch)
  int box2;              // pooled family: a wave with <= 32 box items runs the two-level BOX2 operation (0: off)
  int solo;              // pooled family: single-pixel tickets (deep_split == 6) are traced by solo_trace in a PROLOGUE ahead of the pooled loop (the SOLO instantiation; 0: off).  The hand-over of a wave's last rays from INSIDE the loop is `cold`, below
  int tl_log2;           // levels per treelet of the traversal copy (treelet.h; the masks in nodes64)
  int deep_class;        // a shard's positions below its class table's entry [deep_class] are "deep" tiles (0: feature off)
  int deep_split;        // log2 of the pieces a deep tile is handed out in (2: four tickets of two rows each; 0: whole)
  int deep_cap_log2;     // ... to at most one in 2^this of the launch's waves (5)
  int *cost;             // [nchunks] longest bounce chain seen per tile (nullptr: not recorded)
  unsigned char *cost_px;   // [h * w] with `cost`, single frames: rays traced per PIXEL (depth + 1, saturating), indexed like `out` (nullptr: not recorded)
  const int *px_hdr;     // pixel tickets (the ORD instantiation; nullptr: tile tickets): header of the view's pixel list (kPxHdrInts) ...
  const unsigned *px_list;   // ... and the list itself: (local row << 16) | column, longest bounce chains first
  int px_hold;           // bit k: a wave that draws a ticket of class k does not refill until it is finished
  int px_prio;           // ... and runs at this issue priority meanwhile (0 .. 3)
  int cull;              // pooled family, workgroups of 16 waves: != 0 = the CULL instantiations (boxes are tested against the slot's best root so far: lane_core.h, cull_limit) ...
  float cull_c2;         // ... the scene's constant in a ray's weight W2 = max|1 / d_k| * (d.d) * c2 ...
  float cull_kappa;      // ... and in the limit best + W2 (best^2 + kappa)   (api.cpp: the prepared scene's CullConst)
  const float *u_tab;    // [w]  pixel_u(col, w)
  const float *v_tab;    // [h]  pixel_v(row, h), indexed by the FULL image row
  // pooled family, workgroups of four waves (the shape of twenty waves per CU): a box stack whose LDS capacity `capb` is BELOW its bound (64 H + 63: trees taller
  // than 15 levels) keeps its oldest items in device memory when it would overflow -- [waves][spill_stride] dwords, a wave's own region (nullptr: capb holds the bound)
  unsigned *spill;
  int spill_stride;
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
hipError_t warm_scratch(hipStream_t stream, int *sink_dev);   // the queue's scratch allocated now, not inside the first frame
void warm_build_kernels();
size_t gpu_build_scratch_bytes(int n);   // device scratch one build of n spheres needs
hipError_t gpu_copy_from_pinned(void *dst_dev, const void *src_pinned, size_t bytes, hipStream_t stream);
size_t gpu_build_pinned_bytes();         // host-pinned block the build kernels report through
// `sph7_dev`: the n spheres in device memory; `scratch`, `pinned`: blocks of the sizes above.
// Synchronises the stream; returns the tree height and the root's box.
hipError_t gpu_build_bvh(const float *sph7_dev, int n, const GpuBvhOut &out, char *scratch, char *pinned, hipStream_t stream,
                         int *height_out, float root_lo[3], float root_hi[3]);

constexpr int kOrderBlocksMax = 64;                                   // workgroups per shard of the tile-order sort
constexpr int kOrderScratchInts = kMaxShards * 64 * kOrderBlocksMax;   // its scratch: [shard][bin][workgroup] counts
hipError_t launch_tile_order(int *cost, int *order, int ntiles, int tiles_x, int nshards, int *scratch, hipStream_t stream);
// The view's pixel list from the per-pixel record of its first frame (px_count / px_scan / px_place kernels): pixels by
// descending chain length, header per px_make_header with the classes cut at chains of >= thr[k] rays (thr[0] >= thr[1] >= ...;
// the one-pixel class also capped at `solo_cap` pixels).  `scratch`: px_scratch_ints(ntiles) ints.
struct PxGeom {
  int w, rows_local, rpt_log2, out_skip, tiles_x, tiles_y;
};
// How the classes are cut.  thr[0] > 0: by hand -- class k (1, 8, 16, 32 pixels per ticket) holds the chains of >= thr[k] rays that no
// earlier class holds.  thr[0] == 0: from a model of the launch, evaluated on the device from the list's histogram: a wave that carries
// n rays of equal chain length advances them one bounce per g[k] (units of 0.1 us; k = 0 the solo loop, then 8, 16, 32, 64 rays; they
// differ by where the scene lives: LDS or L2), so a class may hold chains of up to T / g[k] rays if the frame is to end by T, and T is
// the larger of what the longest chain takes in the solo loop and what the waves' time adds up to -- rays of class k cost g[k] / width
// of a wave's time each, those of the 64-pixel class ray_ns -- iterated to a fixed point.
struct PxPolicy {
  int thr[kPxClasses - 1];
  int g[kPxClasses];
  int ray_ns;        // a wave's time per ray in the 64-pixel class (rays of mixed phases share the wave: cheaper than g[4] / 64, the lockstep figure)
  int nwaves;
  int solo_cap;      // the one-pixel class: at most this many pixels (0: no solo loop on this launch)
  int zip;           // the bulk's tickets alternately from both ends of its segment (px_ticket_span)
};
constexpr int kPxBlocksMax = 2048;
constexpr size_t px_scratch_ints() { return (size_t)64 * kPxBlocksMax + 256; }
hipError_t launch_px_order(const unsigned char *cost_px, const PxGeom &g, const PxPolicy &pol, unsigned *list, int *hdr, int *scratch, hipStream_t stream);
hipError_t launch_place_part(const int32_t *part, int32_t *image, int w, int rows_local, int rows_per_tile, int part_id,
                             int nparts, hipStream_t stream);

// (a batch: nframes frames, frame f's parts frame_stride_in elements behind frame f - 1's, its image frame_stride_out behind)
hipError_t launch_place_all(const int32_t *stacked, int32_t *image, int w, int h, int rows_per_tile, int nparts,
                            size_t part_stride, hipStream_t stream, int nframes = 1, size_t frame_stride_in = 0,
                            size_t frame_stride_out = 0);

}  // namespace rtk
