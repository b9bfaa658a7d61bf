// rt_internal.hpp -- the library's internal data model, shared by api.cpp (single-device surface) and
// multi_gpu.cpp (one process, several devices).  Not installed; the public surfaces are include/*.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device.hpp"
#include "rt_host.hpp"

struct rt_group;   // multi_gpu.cpp: the devices behind a multi-device context

// ------------------------------------------------------------------------------------
// Concurrency (SURVEY.md 8b "Async / threading": a Futhark context serialises concurrent calls with an internal lock): every entry
// of the C ABI that takes a context holds the context's lock for the duration of the call -- host threads may share a context, a
// prepared scene and a scene freely; their calls are serialised, their frames are enqueued on the context's one stream in the
// order the calls were admitted.  Lock order: futhark_context -> rt_context (a multi-device parent before its children) ->
// rt_prepared -> rt_scene.  The locks are recursive: entries call one another.
struct rt_context {
  std::recursive_mutex mu;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  bool synced_since_render = true;   // the caller has synchronised the context (rt_context_sync, rt_copy_to_host) since its last render entry: it syncs between frames
  std::string last_launch;  // what the last render entry enqueued (rt_context_last_launch): family, tickets, instantiation, launch shape
  int num_cu = 0;
  int lds_bytes = 0;
  std::string name;
  // configuration
  int variant = RT_VARIANT_AUTO;
  int waves_per_wg = 0;     // persistent families: waves per workgroup (4, 8, 12, 16); 0 = chosen per scene (make_plan)
  int wgs_per_cu = 1;       // persistent workgroups per CU (used with a configured waves_per_wg)
  int thr_shade = 40;       // lanes with a finished fold / a vacant slot that trigger the shade phase
  int thr_leaf = 24;        // phase vote: lanes holding deferred leaves that trigger the sphere phase
  int lmax = 8;             // deferred-leaf capacity per lane
  int lds_scene_bytes = -1; // < 0: as much as fits
  int lds_sph_first = 0;    // stage spheres before nodes when LDS is short
  int handover = 1;         // pooled family, single frames: 1 = an unordered frame (a view's first) is rendered by the DONATE instantiation (a wave that cannot refill gives its rays to waiting sibling waves, LDS mailboxes), an ordered one of ~360 x 360 .. 800 x 800 pixels by the COLD instantiation (a wave's last rays go to the solo loop from inside the pooled loop); 0 = the ordinary kernels; 2 (testing) = DONATE for every single frame
  int donate_max = 64;      // handover == 2: a wave offers its rays when it holds at most this many
  int look_max = 0;         // pooled family: box-stack size from which a wave skips the look at finished folds (0 = auto: 32 for batches and launches of more than 16 384 tiles, 64 otherwise; 1 .. 64)
  int box2 = 1;             // pooled family: two tree levels per operation for a wave with a nearly empty box stack
  int solo = 1;             // pooled family: a wave left with one ray it cannot add to traces the rest of that pixel in the solo loop
  int treelet = rtk::kTreeletDepth;   // the HOST builder cuts the traversal copy into treelets of this many levels (treelet.h; the GPU builder: always kTreeletDepth; another value switches the solo loop off -- a test aid for the numbering)
  int trace_solo = 0;       // rt_render_trace: 1 = a list's one-pixel tickets go through solo_trace as in production (its cycle counters: words 13 .. 15 of a wave's record); 0 = through the pooled loop (the record's item counts are then the frame's complete work)
  int trace_part = 0, trace_nparts = 1;   // rt_render_trace: which part of the row-tile partition the instrumented launch renders
  int ray_planes = 0;       // pooled family: planes of the LDS ray table (0 = chosen with the workgroup shape, 2, 3)
  int gpu_build = 1;        // prepare_scene builds the BVH on the GPU (0: host build + upload)
  int prio_depth = 4;       // pooled family: s_setprio steps at 1x/2x/4x this bounce depth (0: off)
  int grid_div = 0;         // persistent families: launch (CUs * wgs_per_cu) / grid_div workgroups; 0 = by frame size
  int adaptive_order = 1;   // pooled family: order tiles by the previous frame's bounce-chain record
  int deep_class = -1;      // pooled family: tiles of cost classes below this (3: chains of >= 32 bounces) get a wave that does not refill (0: off; -1: chosen per view together with deep_split / deep_cap_log2, api.cpp: deep_policy)
  int deep_split = 2;       // ... and is handed out in 2^this pieces to as many waves (a wave with 16 rays walks a chain faster than one with 64)
  int deep_cap_log2 = 5;    // ... while the pieces occupy at most one in 2^this of the launch's waves
  int xcd_queues = -1;      // pooled family: the tile queue's ticket counters (rt_device.hpp). -1 (auto) = 2: eight counters, one per XCD, taking turns over ONE queue, for every frame and batch; 1: a strip of tile columns per counter (single frames only); 0: one counter
  int tpt_log2 = -1;        // pooled family: log2 of the tiles a ticket covers. -1 (auto): 0 for a single frame; a batch 2, 1 or 0 by the tiles a wave gets (>= 48, >= 24, fewer)
  int static_first = 1;     // pooled family: a wave's first ticket is its own number (no atomic)
  int first_order = 1;      // pooled family: a view's first single frame (no record yet) visits the tile rows, and blocks of 8 tiles inside a row, in bit-reversed order instead of top to bottom, left to right (0: raster)
  int pixel_order = 1;      // pooled family: an ordered single frame draws its tickets from the view's PIXEL list (rt_device.hpp: pixel tickets; the ORD instantiation). 0 = tile tickets only; 1 = where measured faster (api.cpp); 2 = whenever the view has a list (testing)
  int px_thr[4] = {0, 24, 14, 9};    // ... the list's classes: [0] == 0 (default): cut by the model of rt_device.hpp (PxPolicy) from the view's histogram; else chains of >= px_thr[0] rays go out one pixel per ticket (solo loop), >= [1] 8 per ticket, >= [2] 16, >= [3] 32, the rest 64
  int px_g[5] = {0, 0, 0, 0, 0};     // ... the model's bounce cadences for 1 / 8 / 16 / 32 / 64 rays per wave, 0.1 us (0: the built-in figures, by where the scene lives)
  int px_max_tiles = 40000; // ... pixel_order = 1: launches of more tiles than this keep the tile tickets (a work-bound frame gains nothing from the list)
  int px_ray_ns = 0;        // ... and a wave's time per ray of the 64-pixel class, ns (0: 250)
  int px_hold = 0xf;        // ... bit k: a wave holding a ticket of class k does not refill (classes 0 .. 3: 1, 8, 16, 32 pixels)
  int px_zip = 1;           // ... the bulk's tickets alternately from the long and from the short end of its segment (0: sorted straight through)
  int px_prio = 3;          // ... at this issue priority (s_setprio 0 .. 3)
  int px_solo_div = 4;      // ... at most (waves / this) one-pixel tickets
  int stack_cap = 0;        // testing: caps the LDS box stack of the twenty-wave shape (192 dwords; 0: as large as fits) -- a small cap makes the spill path run all the time
  int wide_waves = 1;       // pooled family: five workgroups of four waves per CU (five waves per SIMD) for scenes read from L2 -- 1 (default): launches of 100 000 tiles or more (batches; frames beyond the pixel list's range; from 40 000 tiles for scenes larger than the L2s); 2: every launch (testing); 0: never
  int cull = -1;            // pooled family, workgroups of 16 waves: the CULL instantiations -- boxes tested against the slot's best root so far (lane_core.h: cull_limit; same pixels, fewer tests).  -1 (auto): where the scene and the camera pass the proof's guards (rt_host.hpp: CullConst) and the scene is not wholly LDS resident (rgbbox-sized scenes: the walk is short, the tests saved do not pay for the limit's three instructions per item); 1: wherever the guards pass; 0: never
  int eager_sort = 1;       // pooled family: 1 = the sorts that turn a view's record into its tile order and pixel list are launched right behind the recording frame on the context's SECOND stream (they run while the caller synchronises / sets up its next call; the view's next frame -- or a new view that borrows the order -- waits for their event, usually long past); 0 = lazily on the main stream, ahead of the view's next frame (round 5)
  int borrow = 1;           // pooled family: a NEW view (same image size, partition and bounce limit as a view of this prepared scene already rendered, another camera) renders its first frame through that view's order / pixel list while it records its own -- the order of independent pixels never changes them; the mispredicted long chains are what the DONATE tail is for.  0: a new view's first frame is unordered; 1 (auto): scenes read from L2 only, through the other view's pixel list (a scene that lives in LDS renders its new views unordered: measured as fast); 2: the tile order, 3: the pixel list, for either kind of scene; 4: the pixel list without its holds (testing)
  int sync_policy = 0;      // 1: a render entry WAITS for the view's class table (deep_policy) instead of polling for it -- which instantiation renders frame k is then the same in every run (measurements, PMC passes); 0: no render entry ever waits for the device
  // ticket counters of the persistent families (rtk::kQueueDwords): all zero between launches -- the last
  // wave of a launch to leave the queue zeroes them (rt_context_sync re-zeroes them after a failed launch)
  hipStream_t sort_stream = nullptr, sort_stream_px = nullptr;   // the sort streams (eager_sort): a recorded view's tile order; its pixel list
  hipEvent_t rec_event = nullptr;      // main stream -> sort stream: "the recording frame has been enqueued up to here"
  unsigned *queue_dev = nullptr;
  unsigned *spill_dev = nullptr;   // pooled family, twenty waves per CU, trees taller than 15 levels: the waves' box-stack overflow regions (rt_device.hpp: KParams::spill); allocated by rt_prepare_scene of such a tree (api.cpp: ensure_spill)
  size_t spill_bytes = 0;
  int *order_scratch = nullptr;   // the tile-order sort's chunk counts (rtk::kOrderScratchInts), allocated with the first record
  int *px_scratch = nullptr;      // the pixel-list sort's counts (rtk::px_scratch_ints()), likewise
  unsigned long long *stats_dev = nullptr;
  // per-(w, h) tables of the primary-ray parameters u = i / w and v = (h - row) / h
  struct UvTable {
    int64_t w, h;
    float *u, *v;
    size_t bytes = 0;   // of the one block behind both (u first)
  };
  std::vector<UvTable> uv;
  // per-(tile grid) visiting order of a view's FIRST frame (first_order = 1): tile rows, and blocks of 8 tiles inside a row, in bit-reversed order
  struct FirstOrder {
    int tiles_x, tiles_y;
    int *order;   // [rtk::order_table_ints(tiles_x * tiles_y)] a permutation of the tiles, then zeroed class tables
    size_t bytes;
  };
  std::vector<FirstOrder> first_orders;
  // Freed device blocks kept for the next prepare_scene (the reference's harness prepares the same
  // scene `runs` times: hipMalloc / hipFree of a few MB cost more than the build itself).
  struct Block {
    char *p;
    size_t bytes;
  };
  std::vector<Block> pool;
  // One arena allocated with the context serves the blocks of small scenes (64 KiB granules, first
  // fit): the first prepare_scene then pays no hipMalloc either.
  char *arena = nullptr;
  std::vector<unsigned char> arena_used;   // one flag per granule
  float *cams_dev = nullptr;   // rt_render_batch: the batch's cameras on the device ...
  float *cams_host = nullptr;  // ... and the pinned block they are uploaded from (the caller's array is copied into it)
  size_t cams_bytes = 0;
  hipEvent_t cams_event = nullptr;   // the last upload from cams_host
  bool cams_event_valid = false;
  rt_group *group = nullptr;   // multi-device context: the devices behind it (multi_gpu.cpp); this context is the first device's
  // host-pinned landing area of the views' class tables, kClassChunks chunks of kClassSlots x kClassSlotInts ints, one per prepared scene
  // that has an ordered view (allocated with the context: a hipHostMalloc inside a view's first frames would be timed with them)
  int *class_slab = nullptr;
  unsigned long long class_chunks_used = 0;
  char *pinned = nullptr;  // host-pinned block the build kernels report through
  char *stage = nullptr;   // host-pinned staging (kStageBytes) for uploads of small scenes
};

struct rt_scene {
  rt::SceneDesc desc;
  // Device copies of the spheres, one per device, made by the first prepare_scene on that device (the
  // reference's scene is a device-resident value too: futhark_entry_rgbbox/irreg build it there).
  struct DevCopy {
    int device;
    float *p;
  };
  mutable std::mutex mu;                 // a multi-device prepare_scene uploads from several host threads
  mutable std::vector<DevCopy> copies;
  mutable bool cull_done = false;        // the spheres' side of the culling guards (rt_host.hpp: CullConst), computed by the first prepare_scene
  mutable rt::CullConst cull;
};

// Tile-order state of one (image size, partition, depth, camera) view of a prepared scene.
struct TileOrder {
  int64_t h, w;
  int32_t rows_per_tile, part, nparts, max_depth;
  float cam[12];
  int ntiles = 0;
  int nshards = 1;        // shards of the tile queue the table is laid out for (segments + class tables)
  char *block = nullptr;  // one device block (a context's arena / block pool) behind cost, order, cost_px and px_list ...
  size_t block_bytes = 0;
  rt_context *block_owner = nullptr;   // ... and the context whose arena it came from: the prepared scene's home context, or nullptr = a plain hipMalloc (a frame rendered through another context -- a multi-device context's first child renders the parent's prepared scene)
  int *cost = nullptr;    // [ntiles] record written by the render kernel
  int *order = nullptr;   // [rtk::order_table_ints(ntiles)] position -> tile table for the next frames, then the shards' class tables
  bool valid = false;     // order[] has been computed from a previous frame
  // pixel tickets (single frames of the view; rt_device.hpp): the per-pixel record of the first frame, the list sorted from it
  unsigned char *cost_px = nullptr;   // [cost_px_bytes] rays traced per pixel, indexed like the framebuffer (h * w: a part rendered in place stores at its rows' places)
  unsigned *px_list = nullptr;        // [px_elems] the part's pixels, longest chains first; then the header (rtk::kPxHdrInts)
  size_t cost_px_bytes = 0, px_elems = 0;
  bool px_valid = false;
  bool sort_pending = false, sort_px = false;   // the view's last frame recorded its chains; the sorts (tile order; pixel list) are still to be launched
  hipEvent_t sort_event = nullptr, sort_event_px = nullptr;    // eager_sort: the view's sorts (tile order; pixel list) on the context's sort streams ...
  bool sort_inflight = false, sort_inflight_px = false;        // ... have been launched and the main stream has not waited for them yet
  int rec_out_skip = 0;               // ... KParams::out_skip of the recording frame (how cost_px is indexed)
  bool px_solo = false;               // the list was cut with a one-pixel class (the SOLO flavour of the ORD instantiation renders it)
  uint64_t stamp = 0;     // last use (rt_prepared::order_clock)
  bool have_classes = false;   // classes[] is the host's copy of the (single) class table behind order[]
  int classes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // ... which arrives ASYNCHRONOUSLY: the frame that computes the table enqueues a copy of it into a pinned slot and records
  // `classes_event`; a later render takes it over once the event has completed (no render entry ever waits for the device)
  int classes_slot = -1;             // slot of rt_prepared::classes_pinned
  hipEvent_t classes_event = nullptr;
  bool classes_pending = false;
};
constexpr int kClassSlotInts = 16, kClassSlots = 8;   // per prepared scene: one pinned slot per kept view
constexpr int kClassChunks = 64;                       // per context: chunks of the pinned slab (prepared scenes with ordered views alive at a time; more: own allocations)

struct rt_prepared {
  rt_context *home = nullptr;        // the context that prepared it (its arena serves the views' blocks when it also renders them)
  mutable std::recursive_mutex mu;   // the views' orders are state of the prepared scene that render entries update: held while a frame is set up
  mutable std::vector<TileOrder> orders;
  mutable uint64_t order_clock = 0;
  int64_t n = 0;
  int64_t h = 0, w = 0;
  rt::Camera cam{};
  int height = 0;   // tree height
  int tl_depth = 1; // levels per treelet of the traversal copy (1: no treelets)
  rt::CullConst cull;   // may this scene's rays be culled by their best hit, and with which constants (ok == false: never)
  // canonical {L, I} on the device (SoA, as bvh.fut:28 lays them out)
  float *L7 = nullptr, *bmin = nullptr, *bmax = nullptr;
  int32_t *left = nullptr, *right = nullptr, *parent = nullptr;
  // traversal copy
  float4 *nodes = nullptr, *nodes64 = nullptr, *sph = nullptr, *col = nullptr;
  char *block = nullptr;   // one device allocation behind all of the arrays above
  size_t block_bytes = 0;
  float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
  int *classes_pinned = nullptr;     // [kClassSlots][kClassSlotInts] host-pinned landing area of the views' class tables: a chunk of the context's slab (rt_context::class_slab), or -- slab exhausted -- its own allocation
  int classes_chunk = -1;            // ... which chunk of the slab (-1: own allocation / none) ...
  rt_context *classes_owner = nullptr;   // ... of which context
  std::vector<rt_prepared *> replicas;   // multi-device context: [i] = the scene prepared on device i (i >= 1; [0] unused)
  std::vector<std::future<int>> replica_jobs;   // ... while they are being built (rt_prepare_scene joins them)
};


namespace rtk {
// bvh_build.hip: the bit-reversed visiting order of a view's first frame, built on the device (api.cpp: get_first_order)
hipError_t launch_first_order(int *order, int *rank, int tiles_x, int tiles_y, hipStream_t stream);
// ... and the u / v tables of an image size (api.cpp: get_uv)
hipError_t launch_uv_tables(float *u, float *v, int w, int h, hipStream_t stream);
}

namespace rti {
int fail(rt_context *ctx, const std::string &msg);
int hip_fail(rt_context *ctx, hipError_t e, const char *what);
// Enqueue one part of a frame on ctx's stream (single device).  cam12 == nullptr: the prepared camera.
// inplace: out_dev is the FULL image (frame f at out_dev + f * frame_stride, frame_stride >= h * w) and the part's rows are
// stored at their places in it instead of packed.
int enqueue_render(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                   int32_t part, int32_t nparts, int32_t *out_dev, bool stats, const float *cam12 = nullptr, int32_t nframes = 1,
                   int64_t frame_stride = 0, const float *cams_dev = nullptr, bool inplace = false);
// multi_gpu.cpp
// (nframes > 1: a batch -- frame f to out_dev + f * frame_stride, through cams12 + 12 f when cams12 is given)
int group_render(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t *out_dev,
                 const float *cam12, int32_t nframes = 1, int64_t frame_stride = 0, const float *cams12 = nullptr);
int stage_cams(rt_context *ctx, const float *cams12, int32_t nframes, const float **cams_dev);
void group_prepare_begin(rt_context *ctx, rt_prepared *ps, int64_t h, int64_t w, const rt_scene *scene);   // starts the replicas' builds
int group_prepare_end(rt_context *ctx, rt_prepared *ps);                                                    // joins them
void group_prepared_free(rt_context *ctx, rt_prepared *ps);
int group_sync(rt_context *ctx);
void group_mark_synced(rt_context *ctx);   // the children of a multi-device context: the caller has synchronised
int group_set_variant(rt_context *ctx, int variant);
int group_set_option(rt_context *ctx, const char *name, int64_t value);
void group_destroy(rt_context *ctx);
}  // namespace rti

// first statement of every C-ABI entry that takes a context (a null context is refused by the entry's own checks)
// (-DRT_NO_CONTEXT_LOCK: the locks compiled out -- only for build/tsan_nolock, the run that shows what the thread sanitizer reports without them)
#ifndef RT_NO_CONTEXT_LOCK
#define RT_LOCK(ctx) std::unique_lock<std::recursive_mutex> rt_lock_ = (ctx) ? std::unique_lock<std::recursive_mutex>((ctx)->mu) : std::unique_lock<std::recursive_mutex>()
#define RT_LOCK_PS(ps) std::lock_guard<std::recursive_mutex> ps_lock_((ps)->mu)
#else
#define RT_LOCK(ctx) (void)(ctx)
#define RT_LOCK_PS(ps) (void)(ps)
#endif

#define RT_HIP(ctx, call)                                             \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) return rti::hip_fail((ctx), e_, #call);     \
  } while (0)
