// api.cpp -- the C ABI of libray_mi355x.so: the rt_* surface (include/rt_mi355x.h) and,
// on top of it, the Futhark-shaped drop-in boundary (include/ray.h) that the reference's
// futhark/main.c is written against.
//
// There is no CPU fallback anywhere in this library: every render entry launches HIP
// kernels, and context creation fails loudly when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ray.h"
#include "../../include/rt_mi355x.h"
#include "rt_device.hpp"
#include "rt_host.hpp"
#include "rt_internal.hpp"

namespace rti {
int fail(rt_context *ctx, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return 1;
}
int hip_fail(rt_context *ctx, hipError_t e, const char *what) {
  return fail(ctx, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace rti
using rti::fail;
using rti::hip_fail;

namespace {

// ---- device-block pool -------------------------------------------------------------------------
constexpr size_t kPoolMaxBlocks = 8, kPoolMaxBlockBytes = size_t(256) << 20;
// RT_TICKS=1 (a measurement aid): host microseconds between the stages of a render call, on stderr
struct Ticks {
  bool on = std::getenv("RT_TICKS") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void operator()(const char *what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  [tick] %-28s +%.1f us\n", what, std::chrono::duration<double, std::micro>(n - t).count());
    t = n;
  }
};
constexpr size_t kGranule = size_t(64) << 10, kArenaGranules = 2048;   // 128 MiB arena per context (views' blocks, tables, the Futhark ABI's images)
constexpr size_t kStageBytes = size_t(1) << 20;

hipError_t pool_alloc(rt_context *ctx, char **out, size_t *bytes_io) {
  const size_t want = (*bytes_io + kGranule - 1) & ~(kGranule - 1);   // granules make repeats hit
  // 1. the arena (small scenes)
  const size_t g = want / kGranule;
  if (ctx->arena && g <= kArenaGranules / 2) {
    size_t run = 0;
    for (size_t i = 0; i < kArenaGranules; ++i) {
      run = ctx->arena_used[i] ? 0 : run + 1;
      if (run == g) {
        const size_t first = i + 1 - g;
        std::fill(ctx->arena_used.begin() + static_cast<long>(first), ctx->arena_used.begin() + static_cast<long>(i + 1), 1);
        *out = ctx->arena + first * kGranule;
        *bytes_io = want;
        return hipSuccess;
      }
    }
  }
  // 2. a cached block of a fitting size
  size_t best = ctx->pool.size();
  for (size_t i = 0; i < ctx->pool.size(); ++i)
    if (ctx->pool[i].bytes >= want && ctx->pool[i].bytes <= 2 * want &&
        (best == ctx->pool.size() || ctx->pool[i].bytes < ctx->pool[best].bytes))
      best = i;
  if (best != ctx->pool.size()) {
    *out = ctx->pool[best].p;
    *bytes_io = ctx->pool[best].bytes;
    ctx->pool.erase(ctx->pool.begin() + static_cast<long>(best));
    return hipSuccess;
  }
  *bytes_io = want;
  return hipMalloc(reinterpret_cast<void **>(out), want);
}
// (the caller has drained every stream that used the block)
void pool_free(rt_context *ctx, char *p, size_t bytes) {
  if (!p) return;
  if (ctx && ctx->arena && p >= ctx->arena && p < ctx->arena + kArenaGranules * kGranule) {
    const size_t first = static_cast<size_t>(p - ctx->arena) / kGranule, g = bytes / kGranule;
    std::fill(ctx->arena_used.begin() + static_cast<long>(first), ctx->arena_used.begin() + static_cast<long>(first + g), 0);
    return;
  }
  if (!ctx || bytes > kPoolMaxBlockBytes) {
    (void)hipFree(p);
    return;
  }
  if (ctx->pool.size() >= kPoolMaxBlocks) {
    (void)hipFree(ctx->pool.front().p);
    ctx->pool.erase(ctx->pool.begin());
  }
  ctx->pool.push_back({p, bytes});
}

template <class T>
int upload(rt_context *ctx, T **dev, const void *host, size_t bytes) {
  RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(dev), std::max<size_t>(bytes, 16)));
  RT_HIP(ctx, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// u/v tables for one image size (host computes the same correctly rounded divisions the
// per-pixel code would: rtk::pixel_u / pixel_v), cached on the context.
int get_uv(rt_context *ctx, int64_t w, int64_t h, const float **u, const float **v) {
  for (const auto &t : ctx->uv)
    if (t.w == w && t.h == h) { *u = t.u; *v = t.v; return 0; }
  // (computed on the device, stream-ordered ahead of the frame, in a block of the context's arena: no hipMalloc, no blocking copy)
  rt_context::UvTable t{w, h, nullptr, nullptr};
  const size_t ub = (static_cast<size_t>(w) * 4 + 255) & ~size_t(255);
  char *blk = nullptr;
  size_t bytes = ub + static_cast<size_t>(h) * 4;
  RT_HIP(ctx, pool_alloc(ctx, &blk, &bytes));
  t.u = reinterpret_cast<float *>(blk);
  t.v = reinterpret_cast<float *>(blk + ub);
  t.bytes = bytes;
  RT_HIP(ctx, rtk::launch_uv_tables(t.u, t.v, static_cast<int>(w), static_cast<int>(h), ctx->stream));
  if (ctx->uv.size() >= 16) {   // small LRU-less cap: drop the oldest
    (void)hipStreamSynchronize(ctx->stream);
    pool_free(ctx, reinterpret_cast<char *>(ctx->uv.front().u), ctx->uv.front().bytes);
    ctx->uv.erase(ctx->uv.begin());
  }
  ctx->uv.push_back(t);
  *u = t.u; *v = t.v;
  return 0;
}

// The visiting order of a frame nothing is known about (first_order = 1, the default): the tile ROWS in bit-reversed order and, inside a
// row, blocks of 8 tiles in bit-reversed order of the blocks -- whatever band or corner of the image holds the expensive pixels (irreg:
// the lower half, which a top-to-bottom raster reaches last), a share of it starts early, and the sky's SHADE-only tiles are spread over
// the whole queue.  Measured against the raster, first frames (profiles/r05/exp/e12): rgbbox 500 / 1000 / 2000 -7 / -2 / -8 %, irreg 1000 /
// 2000 -5 / -5.5 %, irreg 500 +-1 %; rows alone about the same; a golden-ratio stride over all tiles: irreg 1000 +8 %.
int get_first_order(rt_context *ctx, int tiles_x, int tiles_y, const int **out) {
  for (const auto &t : ctx->first_orders)
    if (t.tiles_x == tiles_x && t.tiles_y == tiles_y) { *out = t.order; return 0; }
  const int ntiles = tiles_x * tiles_y, nb = (tiles_x + 7) / 8;
  // (built on the device, stream-ordered ahead of the frame: no host copy, nothing waits)
  rt_context::FirstOrder t{tiles_x, tiles_y, nullptr, 0};
  {
    char *blk = nullptr;
    t.bytes = sizeof(int) * static_cast<size_t>(rtk::order_table_ints(ntiles) + tiles_y + nb);
    RT_HIP(ctx, pool_alloc(ctx, &blk, &t.bytes));
    t.order = reinterpret_cast<int *>(blk);
  }
  RT_HIP(ctx, rtk::launch_first_order(t.order, t.order + rtk::order_table_ints(ntiles), tiles_x, tiles_y, ctx->stream));
  if (ctx->first_orders.size() >= 16) {
    (void)hipStreamSynchronize(ctx->stream);
    pool_free(ctx, reinterpret_cast<char *>(ctx->first_orders.front().order), ctx->first_orders.front().bytes);
    ctx->first_orders.erase(ctx->first_orders.begin());
  }
  ctx->first_orders.push_back(t);
  *out = t.order;
  return 0;
}

struct Plan {
  int variant;
  int lds_nodes, lds_sph, smax, lmax, waves, grid;
  int grid_full;   // every persistent workgroup (grid == grid_full unless grid_div says otherwise)
  int capb, capl, ray_planes;
  int spill_stride;   // > 0: capb is below the box stack's bound -- the launch needs a spill region of this many dwords per wave (the twenty-wave shape, tall trees)
  size_t lds_bytes;
};

// Decide the launch shape of the persistent family for one prepared scene.
// `ntiles`: 8x8 tiles of the launch (grid_div == 0 picks the launch size from it: up to ~1000x1000 a
// half-size launch keeps the waves better filled -- 5-18 % per frame --, larger frames want every wave).
// `force_waves` != 0: only workgroups of that many waves (the instrumented instantiation has 8).
// `wide`: the launch may take the shape of FIVE workgroups of four waves per CU (a batch, or a frame too large for a pixel list): five waves
// per SIMD for scenes that are read from L2 (the plain and CULL kernels only; a wave's region must fit a twentieth of the LDS: trees taller than 15 levels get a box stack that may spill to device memory, Plan::spill_stride).
int make_plan(rt_context *ctx, const rt_prepared *ps, Plan *pl, int64_t ntiles, int force_waves = 0, bool wide = false) {
  const bool auto_variant = ctx->variant == RT_VARIANT_AUTO;
  pl->variant = ctx->variant;
  if (auto_variant) pl->variant = ps->n < (int64_t(1) << 22) ? RT_VARIANT_POOLED : RT_VARIANT_PIXEL;
  if (pl->variant == RT_VARIANT_PIXEL) return 0;
  const bool pooled = pl->variant == RT_VARIANT_POOLED;
  const int ni = static_cast<int>(ps->n - 1), n = static_cast<int>(ps->n);
  // depth-first with one node held in a register: at most one pending sibling per level
  const int need = ps->height + 1;
  int smax = 64;
  for (int cand : {12, 16, 20, 24, 32, 48, 64})
    if (cand >= need) { smax = cand; break; }
  if (need > 64) {
    if (auto_variant) { pl->variant = RT_VARIANT_PIXEL; return 0; }
    return fail(ctx, "BVH deeper than 64 levels: unsupported by the persistent kernels");
  }
  pl->smax = smax;
  pl->lmax = ctx->lmax;
  // pooled family: box stack <= 64*H + 128 items (see the kernel's header), leaf list <= 63 + 128
  // (the shapes of twenty waves per CU: 64 * (H + 2) -- what is provable: a SHADE finds at most 63 items and adds at most 64 roots; the blocks
  // pushed behind it have strictly increasing minimum depths 1 .. H - 1, every block but the top one holds at most 64 items, the top one at
  // most 128: 63 + 64 (H - 2) + 128 = 64 H + 63 items)
  struct Shape { int wgs, waves, planes; };
  // (... capped at what a twentieth of the LDS leaves a wave -- 2 000 - 196 - 512 - 192 dwords -- for taller trees: the kernels of that shape keep the oldest items
  // of a stack that would overflow in device memory, render_kernels.hip: SPILL)
  const int cap20 = ctx->stack_cap ? rtk::kSpillCapbTest : rtk::kSpillCapb;
  auto capb_of = [&](const Shape &sh) { return sh.wgs * sh.waves == 20 ? std::min(64 * (ps->height + 2), cap20) : 64 * (ps->height + 3); };
  pl->capb = 64 * (ps->height + 3);
  pl->capl = 192;
  // Workgroup shape {workgroups per CU, waves per workgroup}.  The per-wave scratch grows with the
  // tree height (equal Morton keys make tall trees) and every workgroup stages its own copy of the
  // scene prefix, so: a configured shape is tried first; otherwise the shape with the most waves per
  // CU in which the WHOLE scene still fits in LDS, and for scenes that cannot fit the shape list in
  // the order measured best for them.  When nothing fits AUTO renders with the pixel kernel (no LDS).
  // (pooled family: a shape also says whether the wave's ray table keeps the {d} plane -- 1 KB per wave that
  // lets LEAF read the direction with one ds_read_b128 instead of three ds_bpermute; dropping it is what fits
  // 16 waves next to the whole rgbbox scene)
  std::vector<Shape> shapes;
  const int pl_cfg = ctx->ray_planes == 2 || ctx->ray_planes == 3 ? ctx->ray_planes : 0;
  auto add = [&](int wgs, int waves) {
    if (!pooled) { shapes.push_back({wgs, waves, 0}); return; }
    if (pl_cfg != 2) shapes.push_back({wgs, waves, 3});
    if (pl_cfg != 3) shapes.push_back({wgs, waves, 2});
  };
  if (force_waves) {
    add(std::max(1, ctx->wgs_per_cu), force_waves);
    add(1, force_waves);
  } else {
    if (ctx->waves_per_wg > 0) add(std::max(1, ctx->wgs_per_cu), ctx->waves_per_wg);
    if (pooled) for (int waves : {16, 12, 8, 4}) add(1, waves);
    else for (const auto &sh : {std::pair<int, int>{2, 8}, {1, 8}, {1, 4}}) add(sh.first, sh.second);
  }
  const int node_bytes = pooled ? 64 : 32;
  const int64_t scene_bytes = static_cast<int64_t>(ni) * node_bytes + static_cast<int64_t>(n) * 16;
  auto budget_of = [&](const Shape &sh) {
    // (five workgroups per CU: the LDS is handed out in blocks of 1280 bytes -- 25 of the 128 each, measured: a workgroup of 32 256 bytes
    // is resident four times, one of 31 808 five times, profiles/r06/exp/e11 -- and the shape stages no scene prefix at all)
    if (sh.wgs == 5) return pooled && sh.waves * rtk::pooled_wave_dw(sh.planes, capb_of(sh), pl->capl) * 4 <= 25 * 1280 && ctx->lds_bytes >= 160 * 1024 ? 0 : -1;
    const int total = std::min(ctx->lds_bytes, 160 * 1024) / sh.wgs;
    const int scratch = pooled ? sh.waves * rtk::pooled_wave_dw(sh.planes, capb_of(sh), pl->capl) * 4
                               : sh.waves * (pl->smax + 1 + pl->lmax) * 64 * 4;
    return total - scratch - 512;
  };
  int pick = -1;
  const bool configured = !force_waves && ctx->waves_per_wg > 0;
  if (configured && budget_of(shapes[0]) >= 0) pick = 0;
  if (pick < 0)
    for (size_t i = 0; i < shapes.size() && pick < 0; ++i)
      if (budget_of(shapes[i]) >= scene_bytes) pick = static_cast<int>(i);
  if (pick < 0 && wide && pooled && !force_waves && !configured) {   // the scene does not fit in LDS: five waves per SIMD if they fit
    const Shape sh{5, 4, 2};
    if (budget_of(sh) >= 0) {
      shapes.push_back(sh);
      pick = static_cast<int>(shapes.size()) - 1;
    }
  }
  if (pick < 0)
    for (size_t i = 0; i < shapes.size() && pick < 0; ++i)
      if (budget_of(shapes[i]) >= 0) pick = static_cast<int>(i);
  if (pick < 0) {
    if (auto_variant && !force_waves) { pl->variant = RT_VARIANT_PIXEL; return 0; }
    return fail(ctx, "LDS budget too small for the per-wave traversal scratch");
  }
  const int wgs = shapes[static_cast<size_t>(pick)].wgs;
  pl->waves = shapes[static_cast<size_t>(pick)].waves;
  pl->ray_planes = shapes[static_cast<size_t>(pick)].planes;
  pl->capb = capb_of(shapes[static_cast<size_t>(pick)]);
  pl->spill_stride = (wgs * pl->waves == 20 && pl->capb < 64 * (ps->height + 2)) ? 64 * (ps->height + 2) : 0;
  int budget = budget_of(shapes[static_cast<size_t>(pick)]);
  if (ctx->lds_scene_bytes >= 0) budget = std::min(budget, ctx->lds_scene_bytes);
  int ln, ls;
  if (ctx->lds_sph_first) {
    ls = std::min(n, budget / 16);
    ln = std::min(ni, (budget - ls * 16) / node_bytes);
  } else {
    ln = std::min(ni, budget / node_bytes);
    ls = std::min(n, (budget - ln * node_bytes) / 16);
  }
  pl->lds_nodes = ln;
  pl->lds_sph = ls;
  pl->lds_bytes = pooled ? rtk::pooled_lds_bytes(ln, ls, pl->capb, pl->capl, pl->ray_planes, pl->waves)
                         : rtk::persistent_lds_bytes(ln, ls, pl->smax, pl->lmax, pl->waves);
  // launch size: every persistent workgroup, except that a scene which is only partly LDS resident
  // renders frames up to ~1000x1000 faster with half of them (fuller waves, less L2 traffic in flight)
  const bool whole_scene = ln == ni && ls == n;
  const int div = ctx->grid_div > 0 ? ctx->grid_div : (pooled && !whole_scene && ntiles <= 32768 ? 2 : 1);
  pl->grid = std::max(1, ctx->num_cu * wgs / div);
  // workgroups go round-robin to the 8 XCDs: keep their number a multiple of 8 so that no XCD carries
  // one more persistent workgroup than the others (grid_div=12 -> 42 workgroups measured +15 %)
  if (pl->grid >= 8) pl->grid -= pl->grid % 8;
  pl->grid_full = std::max(1, ctx->num_cu * wgs);
  if (pl->grid_full >= 8) pl->grid_full -= pl->grid_full % 8;
  return 0;
}

// Which tiles of an ordered view are "deep" (their waves do not refill while they trace them) and in how many pieces the
// deepest ones are handed out.  deep_class >= 0: as configured.  Auto (-1), from the view's class table (tiles with a bounce
// chain of >= 32 / 16 / 8 scatters; it reaches the host asynchronously, see deep_policy):
//   * a frame of at most 32768 tiles whose tiles with chains of >= 8 (or >= 16) bounces would park only a few per cent of
//     the launch's wave time if each of them kept a wave to itself (irreg 1000x1000: ~700 of 15 625 tiles; weight = sum
//     of chain-length classes <= 2.5 per wave): all of them are deep, and the deepest waves / 64 tiles go out PIXEL BY
//     PIXEL, one pixel per wave at launch, each traced by the solo loop (render_kernels.hip: solo_trace) -- the frame's
//     longest chains get a whole wave each from t = 0; every persistent workgroup is launched (the chains no longer bound
//     the frame, the work does);
//   * otherwise round 2's setting: chains of >= 32 bounces are deep, the deepest waves / 128 tiles go out in quarters.
//     rgbbox 1000x1000 (141 tiles of >= 32 bounces, ~3 500 of >= 16): pixel-by-pixel tickets for 64 of them and 77 whole
//     held tiles measured 5 % slower.  Larger frames are bound by their work, not by their chains: irreg 2000x2000 with
//     its 56 deepest tiles pixel by pixel +6 %, with chains of >= 16 deep +3.6 %; a rank's quarter of irreg 4000x4000
//     741 us against 699 (profiles/r03/exp/e24_ab.txt).
struct DeepPolicy {
  int deep_class, deep_split, cap_log2;
  bool sparse;
};
// The class table reaches the host without any render entry waiting for it: the frame that sorts a view's tiles enqueues a
// copy of the table into a pinned slot (request_classes, below), and a later render of the view adopts it once that copy
// has completed -- until then the view runs on the default setting (same pixels).  With eager_sort the sorts and the copy are launched
// behind the recording frame, so a caller that syncs after every frame, as the reference's harness does, has the policy from the
// view's second or third frame on; with eager_sort = 0 the copy is enqueued inside the view's second frame and the policy arrives
// with its third; a caller that enqueues frames back to back gets it a few frames later.  Only the tile-ticket path reads it (views
// outside the pixel list's gates); option sync_policy = 1 waits for it instead (measurements).  `may_wait`: the diagnostic entry (rt_render_trace), which synchronises anyway.
int deep_policy(rt_context *ctx, const rt_prepared *ps, TileOrder *to, int waves_full, DeepPolicy *dp, bool may_wait = false) {
  *dp = DeepPolicy{ctx->deep_class < 0 ? 3 : ctx->deep_class, ctx->deep_split, ctx->deep_cap_log2, false};
  if (ctx->deep_class >= 0 || !to || !to->valid || to->nshards != 1) return 0;
  if (!to->have_classes) {
    if (to->classes_pending && ps->classes_pinned && to->classes_slot >= 0) {
      hipError_t q = (may_wait || ctx->sync_policy) ? hipEventSynchronize(to->classes_event) : hipEventQuery(to->classes_event);
      if (q == hipSuccess) {
        std::memcpy(to->classes, ps->classes_pinned + kClassSlotInts * to->classes_slot, sizeof to->classes);
        to->have_classes = true;
        to->classes_pending = false;
      } else {
        (void)hipGetLastError();   // (hipErrorNotReady is not an error)
      }
    } else if (may_wait) {
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      RT_HIP(ctx, hipMemcpy(to->classes, to->order + to->ntiles, sizeof to->classes, hipMemcpyDeviceToHost));
      to->have_classes = true;
    }
    if (!to->have_classes) return 0;
  }
  if (!ctx->solo || ps->tl_depth != rtk::kTreeletDepth) return 0;
  const int64_t t3 = to->classes[3], t4 = to->classes[4], t5 = to->classes[5];
  const int64_t w4 = 32 * t3 + 16 * (t4 - t3), w5 = w4 + 8 * (t5 - t4), budget = int64_t(5) * waves_full / 2;
  if (to->ntiles > 32768) return 0;
  if (w5 <= budget) *dp = DeepPolicy{5, 6, 0, true};
  else if (w4 <= budget) *dp = DeepPolicy{4, 6, 0, true};
  return 0;
}

// Behind the tile-order sort of a view: its class table on its way to the host (stream-ordered copy into the view's pinned
// slot + an event).  Nothing waits here.
int request_classes(rt_context *ctx, const rt_prepared *ps_c, TileOrder *to, hipStream_t stream) {
  rt_prepared *ps = const_cast<rt_prepared *>(ps_c);   // (the orders are `mutable` state of a prepared scene; so is their landing area)
  to->classes_pending = false;
  if (ctx->deep_class >= 0 || to->nshards != 1) return 0;   // no policy reads the table
  if (!ps->classes_pinned) {
    for (int c = 0; c < kClassChunks && !ps->classes_pinned && ctx->class_slab; ++c)
      if (!(ctx->class_chunks_used >> c & 1ull)) {
        ctx->class_chunks_used |= 1ull << c;
        ps->classes_chunk = c;
        ps->classes_owner = ctx;
        ps->classes_pinned = ctx->class_slab + static_cast<size_t>(c) * kClassSlotInts * kClassSlots;
      }
    if (!ps->classes_pinned)
      RT_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ps->classes_pinned), sizeof(int) * kClassSlotInts * kClassSlots, hipHostMallocDefault));
  }
  if (to->classes_slot < 0) {
    unsigned used = 0;
    for (const auto &o : ps->orders)
      if (o.classes_slot >= 0) used |= 1u << o.classes_slot;
    for (int sl = 0; sl < kClassSlots && to->classes_slot < 0; ++sl)
      if (!(used >> sl & 1u)) to->classes_slot = sl;
    if (to->classes_slot < 0) return 0;   // (cannot happen: at most kClassSlots views are kept)
  }
  if (!to->classes_event) RT_HIP(ctx, hipEventCreateWithFlags(&to->classes_event, hipEventDisableTiming));
  RT_HIP(ctx, hipMemcpyAsync(ps->classes_pinned + kClassSlotInts * to->classes_slot, to->order + to->ntiles, sizeof to->classes,
                             hipMemcpyDeviceToHost, stream));
  RT_HIP(ctx, hipEventRecord(to->classes_event, stream));
  to->classes_pending = true;
  return 0;
}

// The sorts that turn a view's record (its last recording frame's cost / cost_px) into its tile order and pixel list, + the class table
// on its way to the host.  eager_sort (the default): on the context's SECOND stream, behind an event of the main stream -- they run
// while the caller synchronises and sets up its next call; whoever uses the order next makes the main stream wait for `sort_event`
// (await_view).  Otherwise on the main stream, in line.  `p` / `pl`: the geometry and launch plan of a frame of this view (any frame of the
// view has the same).
int sort_view(rt_context *ctx, const rt_prepared *ps, TileOrder *v, const rtk::KParams &p, const Plan &pl) {
  if (!v->sort_pending) return 0;
  v->sort_pending = false;
  hipStream_t st = ctx->stream, st_px = ctx->stream;
  if (ctx->eager_sort) {
    // (two sort streams: the tile order's three launches and the pixel list's four are independent and run side by side)
    if (!ctx->sort_stream) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->sort_stream, hipStreamNonBlocking));
    if (!ctx->sort_stream_px) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->sort_stream_px, hipStreamNonBlocking));
    if (!ctx->rec_event) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->rec_event, hipEventDisableTiming));
    RT_HIP(ctx, hipEventRecord(ctx->rec_event, ctx->stream));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->sort_stream, ctx->rec_event, 0));
    st = ctx->sort_stream;
    if (v->sort_px) {
      RT_HIP(ctx, hipStreamWaitEvent(ctx->sort_stream_px, ctx->rec_event, 0));
      st_px = ctx->sort_stream_px;
    }
  }
  if (v->sort_px) {
    // the view's pixel list from the per-pixel record (first: a frame that borrows the list waits for this one)
    if (!ctx->px_scratch) RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->px_scratch), sizeof(int) * rtk::px_scratch_ints()));
    const rtk::PxGeom g{p.w, p.rows_local, p.rpt_log2, v->rec_out_skip, p.tiles_x, p.tiles_y};
    rtk::PxPolicy pol{};
    for (int k = 0; k < 4; ++k) pol.thr[k] = ctx->px_thr[k];
    // the model's bounce cadences (0.1 us; measured, profiles/r05/README.md): a scene that lives in LDS, one that is read from L2
    const bool whole_scene = pl.lds_nodes == static_cast<int>(ps->n - 1) && pl.lds_sph == static_cast<int>(ps->n);
    // (the solo loop's cadence with treelets of 4 levels, round 6: 2.0 / 3.8 us per bounce -- 25 / 45 with 2 levels; irreg 700 x 700 -7 %, rgbbox 500 x 500 -7 %,
    // 700 x 700 -4 %, the other sizes +-1 %: profiles/r06/exp/e7_solo_cadence_ab.txt.  LDS: 64 rays 24 us sorted straight through, 18 with the bulk zipped: r05 e14)
    static const int g_lds[5] = {20, 45, 65, 100, 180}, g_l2[5] = {38, 120, 170, 230, 330};
    for (int k = 0; k < 5; ++k) pol.g[k] = ctx->px_g[k] > 0 ? ctx->px_g[k] : (whole_scene ? g_lds[k] : g_l2[k]);
    pol.ray_ns = ctx->px_ray_ns > 0 ? ctx->px_ray_ns : 250;
    pol.nwaves = pl.grid_full * pl.waves;
    // (no one-pixel class for a launch of more than 32 768 tiles: its work bounds it, not its longest chains -- and the kernel
    // without the solo call is 1-5 % faster)
    pol.solo_cap = (ctx->solo && ps->tl_depth == rtk::kTreeletDepth && p.nchunks <= 32768) ? pl.grid_full * pl.waves / std::max(1, ctx->px_solo_div) : 0;
    v->px_solo = pol.solo_cap > 0;
    pol.zip = ctx->px_zip;
    RT_HIP(ctx, rtk::launch_px_order(v->cost_px, g, pol, v->px_list, reinterpret_cast<int *>(v->px_list + v->px_elems), ctx->px_scratch, st_px));
    v->px_valid = true;
    if (st_px != ctx->stream) {
      if (!v->sort_event_px) RT_HIP(ctx, hipEventCreateWithFlags(&v->sort_event_px, hipEventDisableTiming));
      RT_HIP(ctx, hipEventRecord(v->sort_event_px, st_px));
      v->sort_inflight_px = true;
    }
  }
  if (!ctx->order_scratch) RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->order_scratch), sizeof(int) * rtk::kOrderScratchInts));
  RT_HIP(ctx, rtk::launch_tile_order(v->cost, v->order, v->ntiles, p.tiles_x, v->nshards, ctx->order_scratch, st));
  v->valid = true;
  v->have_classes = false;
  if (int rc = request_classes(ctx, ps, v, st)) return rc;
  if (st != ctx->stream) {
    if (!v->sort_event) RT_HIP(ctx, hipEventCreateWithFlags(&v->sort_event, hipEventDisableTiming));
    RT_HIP(ctx, hipEventRecord(v->sort_event, st));
    v->sort_inflight = true;
  }
  return 0;
}
// the main stream waits for a view's sorts (once: everything behind the wait is ordered after them).  `tiles` / `px`: which of the two chains the caller depends on --
// a frame that renders through the view's pixel list and records nothing does not wait for the tile order (three launches and the class table's copy to the host: 0.1 ms
// on the side stream against 0.03 for the list, profiles/r06/exp/e14)
int await_view(rt_context *ctx, TileOrder *v, bool tiles = true, bool px = true) {
  if (tiles && v->sort_inflight) {
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, v->sort_event, 0));
    v->sort_inflight = false;
  }
  if (px && v->sort_inflight_px) {
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, v->sort_event_px, 0));
    v->sort_inflight_px = false;
  }
  return 0;
}
void drain_streams(rt_context *ctx);
// a view's block back to where it came from: its home context's arena / pool (under that context's lock when it is not the caller's), or hipFree.
// The caller has drained its own streams.
void free_view_block(rt_context *ctx, TileOrder &v) {
  if (!v.block) return;
  if (!v.block_owner) {
    (void)hipFree(v.block);
  } else if (v.block_owner == ctx) {
    pool_free(ctx, v.block, v.block_bytes);
  } else {
    RT_LOCK(v.block_owner);
    drain_streams(v.block_owner);
    pool_free(v.block_owner, v.block, v.block_bytes);
  }
  v.block = nullptr;
}
// have the view's sorts finished on the device?  (never blocks; clears the in-flight flags when they have: nothing needs to wait any more)
bool sorts_complete(TileOrder *v) {
  if (v->sort_inflight) {
    if (hipEventQuery(v->sort_event) != hipSuccess) { (void)hipGetLastError(); return false; }
    v->sort_inflight = false;
  }
  if (v->sort_inflight_px) {
    if (hipEventQuery(v->sort_event_px) != hipSuccess) { (void)hipGetLastError(); return false; }
    v->sort_inflight_px = false;
  }
  return true;
}
// both streams drained (before buffers the sorts may touch are freed)
void drain_streams(rt_context *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->sort_stream) (void)hipStreamSynchronize(ctx->sort_stream);
  if (ctx->sort_stream_px) (void)hipStreamSynchronize(ctx->sort_stream_px);
}

// The overflow regions of the waves' box stacks (KParams::spill): `stride` dwords for each wave of the twenty-wave shape.  Allocated by rt_prepare_scene for a tree
// taller than 15 levels (a synchronisation point anyway) -- or, failing that (options changed since), by the first launch that needs them.
int ensure_spill(rt_context *ctx, int stride) {
  const size_t need = sizeof(unsigned) * static_cast<size_t>(stride) * static_cast<size_t>(std::max(1, ctx->num_cu)) * 20;
  if (ctx->spill_bytes >= need) return 0;
  drain_streams(ctx);
  if (ctx->spill_dev) (void)hipFree(ctx->spill_dev);
  ctx->spill_dev = nullptr; ctx->spill_bytes = 0;
  RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->spill_dev), need));
  ctx->spill_bytes = need;
  return 0;
}

// Is the traversal copy of the scene (64 bytes per inner node, 16 per sphere) larger than the eight L2s together?
bool rt_scene_exceeds_l2(const rt_prepared *ps) { return static_cast<int64_t>(ps->n) * 80 > (int64_t(32) << 20); }

// May this launch cull (lane_core.h: cull_limit)?  The scene's guards (rt_prepared::cull), the launch shape the CULL instantiations
// exist for, and every camera origin of the launch inside the scene guard -- a batch's cameras are read from the context's pinned
// copy of them (stage_cams); cameras that live only on the device switch culling off.
bool cull_allowed(const rt_context *ctx, const rt_prepared *ps, const Plan &pl, const rtk::KParams &p, const float *cams_dev, int nframes) {
  if (ctx->cull == 0 || !ps->cull.ok || pl.variant != RT_VARIANT_POOLED || (pl.waves != 16 && pl.waves != 4)) return false;
  if (ctx->cull < 0 && pl.lds_nodes == p.n_nodes && pl.lds_sph == p.n_sph) return false;
  if (cams_dev == nullptr) return rt::cull_origin_ok(ps->cull, &p.cam.ox);
  if (cams_dev != ctx->cams_dev || ctx->cams_host == nullptr) return false;
  for (int f = 0; f < nframes; ++f)
    if (!rt::cull_origin_ok(ps->cull, ctx->cams_host + 12 * f)) return false;
  return true;
}

}  // namespace

int rti::enqueue_render(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                        int32_t part, int32_t nparts, int32_t *out_dev, bool stats, const float *cam12, int32_t nframes,
                        int64_t frame_stride, const float *cams_dev, bool inplace) {
  if (!ctx || !ps) return fail(ctx, "null context or prepared scene");
  RT_LOCK(ctx);                                                  // (multi_gpu.cpp calls this on the children of a parent it holds)
  RT_LOCK_PS(ps);                                                // the view's record / order / pixel list are updated below
  if (!out_dev) return fail(ctx, "null output pointer");
  if (h <= 0 || w <= 0 || h > (1 << 20) || w > (1 << 20) || h * w > (int64_t(1) << 30))
    return fail(ctx, "image size out of range");
  // (h, w) need not be the size given to prepare_scene: like the reference's `render h w {objs, cam}`
  // (ray.fut:246) the frame is then traced through the camera prepare_scene derived (its aspect ratio)
  if (rows_per_tile <= 0 || nparts <= 0 || part < 0 || part >= nparts) return fail(ctx, "bad row-tile partition");
  if (max_depth < 0) return fail(ctx, "negative max_depth");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  (void)hipGetLastError();   // (the launchers below read the thread's last error right behind their launches: an earlier call's leftover is not theirs)
  // Does this caller synchronise between its frames?  It called rt_context_sync (futhark_context_sync: the completion point of the ABI) since
  // its last render entry, or the stream is idle right now.  Then a recording frame's sorts are launched behind the frame and run while
  // the caller synchronises; a caller that enqueues frames back to back gets them lazily, ahead of the view's next frame, as in round 5 --
  // sorts squeezed in between its frames cost it 0.03-0.07 ms per new view, and it may never render the view again.
  bool stream_was_idle = ctx->synced_since_render;
  ctx->synced_since_render = false;
  if (!stream_was_idle && ctx->eager_sort && nframes == 1) {
    stream_was_idle = hipStreamQuery(ctx->stream) == hipSuccess;
    (void)hipGetLastError();
  }
  rtk::KParams p{};
  p.nodes = ps->nodes; p.nodes64 = ps->nodes64; p.sph = ps->sph; p.col = ps->col;
  std::copy(ps->root_lo, ps->root_lo + 3, p.root_lo);
  std::copy(ps->root_hi, ps->root_hi + 3, p.root_hi);
  p.n_nodes = static_cast<int>(ps->n - 1); p.n_sph = static_cast<int>(ps->n);
  std::memcpy(&p.cam, cam12 ? static_cast<const void *>(cam12) : static_cast<const void *>(&ps->cam), sizeof(p.cam));
  p.w = static_cast<int>(w); p.h = static_cast<int>(h);
  p.rows_local = static_cast<int>(rt::part_rows(h, rows_per_tile, part, nparts));
  p.rows_per_tile = rows_per_tile; p.part = part; p.nparts = nparts;
  p.rpt_log2 = (rows_per_tile & (rows_per_tile - 1)) == 0 ? __builtin_ctz(rows_per_tile) : -1;
  p.tiles_x = (p.w + 7) / 8;
  p.tiles_y = (p.rows_local + 7) / 8;
  p.max_depth = max_depth;
  p.out = out_dev;
  p.stats = ctx->stats_dev;
  p.nframes = 1;
  if (p.rows_local == 0) {
    ctx->last_launch = "family=none (no rows)";
    return 0;
  }
  const int64_t frame_elems = inplace ? h * w : static_cast<int64_t>(p.rows_local) * p.w;
  if (nframes < 1 || (nframes > 1 && (frame_stride < frame_elems || frame_stride * nframes >= (int64_t(1) << 31))))
    return fail(ctx, "bad batch: nframes >= 1, frame_stride >= rows * w (in place: h * w), nframes * frame_stride < 2^31");
  if (inplace) {
    // the part's row tile k (rows_per_tile rows) starts at image row (k * nparts + part) * rows_per_tile: k * rows_per_tile
    // of that is the packed position the kernels compute anyway, `part * rows_per_tile` goes into the base pointer, the rest
    // -- k * (nparts - 1) * rows_per_tile rows -- is k * out_skip
    p.out = out_dev + static_cast<int64_t>(part) * rows_per_tile * w;
    p.out_skip = static_cast<int>(static_cast<int64_t>(nparts - 1) * rows_per_tile * w);
  }
  if (max_depth == 0) {
    // `while depth < 0`: no ray is traced, every pixel is the initial colour (0,0,0)
    for (int f = 0; f < nframes; ++f) {
      if (!inplace) {
        RT_HIP(ctx, hipMemsetAsync(out_dev + f * frame_stride, 0, sizeof(int32_t) * static_cast<size_t>(p.rows_local) * p.w, ctx->stream));
        continue;
      }
      for (int64_t k = 0; k * rows_per_tile < p.rows_local; ++k) {   // the part's row tiles, one by one, at their places
        const int64_t rows = std::min<int64_t>(rows_per_tile, p.rows_local - k * rows_per_tile);
        RT_HIP(ctx, hipMemsetAsync(out_dev + f * frame_stride + (k * nparts + part) * rows_per_tile * w, 0,
                                   sizeof(int32_t) * static_cast<size_t>(rows * w), ctx->stream));
      }
    }
    ctx->last_launch = "family=none (memset)";
    return 0;
  }
  Ticks tick;
  Plan pl{};
  if (stats) pl.variant = RT_VARIANT_PIXEL;
  else {
    const int64_t ntiles1 = static_cast<int64_t>((w + 7) / 8) * ((p.rows_local + 7) / 8);
    // (twenty waves per CU: batches and frames of 100 000 tiles or more -- launches bound by their work.  Measured, profiles/r06/exp/e11: irreg's batch of 20
    // frames of 1000 x 1000 0.110 -> 0.104 ms per frame, one frame of 4000 x 4000 1.72 -> 1.63 ms, 2000 x 2000 -- 62 500 tiles -- the same; single frames
    // within the pixel list's range keep the 16-wave kernels, whose ORD / SOLO / DONATE instantiations they are rendered by)
    // (a scene larger than the chip's L2s -- 32 MB; the 10^6-sphere scene is 80 -- takes the shape from every frame beyond the pixel list's range, together with a
    // ticket counter per XCD over its own STRIP of the image, below: 2000 x 2000 1.01 -> 0.925 ms, 4000 x 4000 2.47 -> 2.30; e13.  Trees taller than 15 levels run
    // the SPILL kernels in that shape)
    const bool huge = rt_scene_exceeds_l2(ps);
    const bool wide = ctx->wide_waves == 2 || (ctx->wide_waves == 1 && ntiles1 * nframes >= (huge ? 40000 : 100000) && (nframes > 1 || ntiles1 > ctx->px_max_tiles));
    if (int rc = make_plan(ctx, ps, &pl, ntiles1 * nframes, 0, wide)) return rc;
  }
  tick("plan");
  if (nframes > 1 && pl.variant != RT_VARIANT_POOLED) {
    // only the pooled family renders a batch in one launch: the others take the frames one by one
    std::vector<float> hc;
    if (cams_dev) {
      hc.resize(static_cast<size_t>(nframes) * 12);
      RT_HIP(ctx, hipMemcpyAsync(hc.data(), cams_dev, hc.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (int f = 0; f < nframes; ++f)
      if (int rc = enqueue_render(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, out_dev + f * frame_stride, stats,
                                  cams_dev ? hc.data() + 12 * f : cam12, 1, 0, nullptr, inplace))
        return rc;
    return 0;
  }
  if (pl.variant == RT_VARIANT_PIXEL) {
    RT_HIP(ctx, rtk::launch_pixel(p, stats, ctx->stream));
    ctx->last_launch = stats ? "family=pixel (instrumented)" : "family=pixel";
    return 0;
  }
  p.nframes = nframes;
  p.frame_stride = static_cast<int>(frame_stride);
  p.cams = reinterpret_cast<const rtk::Cam *>(cams_dev);
  p.queue = ctx->queue_dev;
  p.nchunks = p.tiles_x * ((p.rows_local + 7) / 8);
  // Tiles per ticket.  A frame: one (with eight counters nothing saturates, and several tiles per ticket cost the 10^6-sphere
  // frame 37 %: a wave then walks adjacent expensive tiles one after the other).
  // (a batch: four tiles per ticket while every wave still gets a few dozen tiles -- the bench's 20 frames of 1000x1000 are 76
  // tiles per wave -- fewer when a launch is small: a rank's eighth of those frames is 10-20 tiles per wave, and with four per
  // ticket the waves' loads differ by whole tickets)
  const int64_t tiles_per_wave = static_cast<int64_t>(p.nchunks) * nframes / std::max(1, pl.grid * pl.waves);
  // (twenty waves per CU: two tiles per ticket at most, and the look from 16 items down -- a wave there has a quarter of a SIMD's issue slots
  // less and waits longer for each of its loads; irreg's batch 0.104 -> 0.101-0.103 ms per frame, a floor of 6 400 spheres 0.101 -> 0.096, profiles/r06/exp/e11)
  const bool twenty = pl.waves * (pl.grid_full / std::max(1, ctx->num_cu)) == 20;
  p.tpt_log2 = ctx->tpt_log2 >= 0 ? ctx->tpt_log2 : (nframes > 1 ? (tiles_per_wave >= 48 && !twenty ? 2 : tiles_per_wave >= 24 ? 1 : 0) : 0);
  // Eight ticket counters (one per XCD: workgroup b runs on XCD b % 8).  Default: they take turns over ONE queue (counter
  // s hands out tickets s, s + 8, ...): the adaptive order stays global and one word no longer carries every draw --
  // measured against one counter: irreg 1000x1000 -8 %, 4000x4000 -38 %, the 10^6-sphere frame -18 %, rgbbox +-1 %.
  // xcd_queues=1 gives every counter its own strip of tile columns instead (an XCD's L2 then serves one strip of the scene:
  // the 10^6-sphere frame's L2 hit rate 65.9 -> 66.9 %; slower than taking turns because the deepest tiles -- the ones
  // handed out in pieces -- are not spread evenly over the strips); single frames only: a batch's class-major ticket
  // order is defined over one queue.
  // (... except for one frame of a scene larger than the L2s in the twenty-wave shape: there the strips win -- an XCD's L2 then serves the part of the scene its
  // strip of the image looks at; 10^6 spheres at 2000 x 2000: 0.99-1.05 ms taking turns, 0.925 in strips; with 16 waves per CU 1.01 / 1.00)
  const int xq = ctx->xcd_queues < 0 ? ((nframes == 1 && pl.variant == RT_VARIANT_POOLED && pl.waves * (pl.grid_full / std::max(1, ctx->num_cu)) == 20 && rt_scene_exceeds_l2(ps)) ? 1 : 2) : ctx->xcd_queues;
  p.nshards = (xq && (nframes == 1 || xq == 2) && pl.variant == RT_VARIANT_POOLED && pl.grid % rtk::kMaxShards == 0) ? rtk::kMaxShards : 1;
  p.interleave = p.nshards > 1 && xq == 2;
  const int order_shards = p.interleave ? 1 : p.nshards;   // layout of the view's order table
  p.static_first = ctx->static_first;
  p.lds_nodes = pl.lds_nodes; p.lds_sph = pl.lds_sph;
  p.smax = pl.smax; p.lmax = pl.lmax;
  p.thr_shade = ctx->thr_shade; p.thr_leaf = ctx->thr_leaf;
  p.capb = pl.capb; p.capl = pl.capl; p.ray_planes = pl.ray_planes;
  if (pl.spill_stride > 0) {
    // (the overflow regions of the waves' box stacks: there since rt_prepare_scene for a tall tree; a no-op then)
    if (int rc = ensure_spill(ctx, pl.spill_stride)) return rc;
    p.spill = ctx->spill_dev;
    p.spill_stride = pl.spill_stride;
  }
  p.prio_depth = ctx->prio_depth;
  p.box2 = ctx->box2;
  // (batches, and launches of more than 16 384 tiles -- frames beyond 1000 x 1000 and a rank's share of a 4000 x 4000 one: -1.8 .. -3.6 %
  // with 32, profiles/r04/exp/e10, e11; a 1000 x 1000 frame is the same within +-1 % either way and keeps 64)
  p.look_max = ctx->look_max > 0 ? ctx->look_max : (twenty ? 16 : (nframes > 1 || p.nchunks > 16384) ? 32 : 64);
  p.tl_log2 = ps->tl_depth;
  p.solo = ctx->solo;
  if (pl.variant == RT_VARIANT_POOLED) {
    if (ps->n >= (int64_t(1) << 22)) return fail(ctx, "pooled kernel: at most 2^22 spheres (work items and hit keys carry the leaf index in 22 bits)");
    if (p.rpt_log2 < 0) return fail(ctx, "pooled kernel: rows_per_tile must be a power of two");
    if (int rc = get_uv(ctx, w, h, &p.u_tab, &p.v_tab)) return rc;
    tick("uv tables");
    // May a view of this shape ever render through a pixel list (the ORD launch condition's static part)?  Only then does it get the
    // per-pixel buffers (5 bytes per pixel) and does its first frame store the per-pixel record.
    const bool px_static_ok = ctx->pixel_order == 2 ||
                              (ctx->pixel_order == 1 && ctx->deep_class < 0 && max_depth > 4 && pl.waves == 16 && ctx->adaptive_order == 1 && p.nchunks <= ctx->px_max_tiles &&
                               (p.nchunks >= 1024 || (pl.lds_nodes == p.n_nodes && pl.lds_sph == p.n_sph)));
    TileOrder *to = nullptr;      // the view this frame belongs to (its record, its order, its pixel list)
    TileOrder *use = nullptr;     // the view whose order / pixel list this frame is rendered through: `to`, or the view it borrows from
    bool borrowed = false;
    if (ctx->adaptive_order && !p.cams) {   // (a batch with its own cameras has no single view to order tiles by)
      auto same_shape = [&](const TileOrder &o) {
        return o.h == h && o.w == w && o.rows_per_tile == rows_per_tile && o.part == part && o.nparts == nparts && o.max_depth == max_depth &&
               o.ntiles == p.nchunks && o.nshards == order_shards;
      };
      for (auto &o : ps->orders)
        if (same_shape(o) && std::memcmp(o.cam, &p.cam, sizeof o.cam) == 0) to = &o;
      if (!to) {
        // A view not seen before: at most 8 are kept; the least recently used one gives up its buffers, which are reused as
        // they are when the sizes match (a camera path rendered frame by frame: stream-ordered, no synchronisation and no
        // hipFree / hipMalloc per new view).
        TileOrder o{};
        // (pixel tickets: the per-pixel record and the pixel list of the view, if this context may use them)
        const bool px_ok = px_static_ok && w < 65536 && p.rows_local < 65536;
        const size_t px_bytes = px_ok ? static_cast<size_t>(h) * static_cast<size_t>(w) : 0;
        const size_t px_elems = px_ok ? static_cast<size_t>(p.rows_local) * static_cast<size_t>(p.w) : 0;
        auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
        const size_t off_order = up(sizeof(int) * static_cast<size_t>(p.nchunks)), off_cost_px = off_order + up(sizeof(int) * static_cast<size_t>(rtk::order_table_ints(p.nchunks))),
                     off_px_list = off_cost_px + up(px_bytes), need_bytes = off_px_list + up(sizeof(unsigned) * (px_elems + (px_ok ? rtk::kPxHdrInts : 0)));
        if (ps->orders.size() >= 8) {
          size_t lru = 0;
          for (size_t i = 1; i < ps->orders.size(); ++i)
            if (ps->orders[i].stamp < ps->orders[lru].stamp) lru = i;
          TileOrder &v = ps->orders[lru];
          o.classes_event = v.classes_event;   // (a copy still in flight lands in the slot before any later one: same stream)
          o.classes_slot = v.classes_slot;
          o.sort_event = v.sort_event;
          o.sort_event_px = v.sort_event_px;
          if (int rc = await_view(ctx, &v)) return rc;   // (sorts of the evicted view still running on the sort stream touch these buffers: the main stream goes behind them)
          if (v.block_bytes >= need_bytes) {       // the evicted view's block as it is
            o.block = v.block;
            o.block_bytes = v.block_bytes;
            o.block_owner = v.block_owner;
          } else {
            drain_streams(ctx);
            free_view_block(ctx, v);
          }
          ps->orders.erase(ps->orders.begin() + static_cast<std::ptrdiff_t>(lru));
        }
        o.h = h; o.w = w; o.rows_per_tile = rows_per_tile; o.part = part; o.nparts = nparts; o.max_depth = max_depth;
        std::memcpy(o.cam, &p.cam, sizeof o.cam);
        o.ntiles = p.nchunks;
        o.nshards = order_shards;
        // one block of the context's arena / block pool behind the view's four arrays (a hipMalloc each inside a view's first render call
        // cost the reference's harness ~0.3 ms of its first frame)
        if (!o.block) {
          o.block_bytes = need_bytes;
          if (ctx == ps->home) {
            RT_HIP(ctx, pool_alloc(ctx, &o.block, &o.block_bytes));
            o.block_owner = ctx;
          } else {      // (rendered through another context than the one that prepared the scene: no arena of ours to take it from)
            RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&o.block), o.block_bytes));
          }
        }
        o.cost = reinterpret_cast<int *>(o.block);
        o.order = reinterpret_cast<int *>(o.block + off_order);
        if (px_ok) {
          o.cost_px = reinterpret_cast<unsigned char *>(o.block + off_cost_px);
          o.px_list = reinterpret_cast<unsigned *>(o.block + off_px_list);
          o.cost_px_bytes = px_bytes;
          o.px_elems = px_elems;
        }
        RT_HIP(ctx, hipMemsetAsync(o.cost, 0, sizeof(int) * static_cast<size_t>(o.ntiles), ctx->stream));
        ps->orders.push_back(o);
        to = &ps->orders.back();
      }
      to->stamp = ++ps->order_clock;
      // the view's last frame left a record that nothing has sorted yet (eager_sort = 0, or a record made before the option was set)
      if (int rc = sort_view(ctx, ps, to, p, pl)) return rc;
      use = to->valid ? to : nullptr;
      // A NEW view (no order of its own yet) borrows the order / pixel list of the most recently rendered view of the same shape
      // (`borrow`): the reference's render is stateless (ray.fut:246) and a caller that moves the camera renders nothing but first
      // frames -- which were unordered (tiles in bit-reversed order, DONATE tail).  Neighbouring views agree on WHERE the long chains
      // are (they cluster at the walls' edges / at grazing angles, profiles/r04/README.md) even though single pixels do not; the
      // chains the borrowed list places wrongly are what the DONATE tail catches.  Only the order of independent pixels changes.
      // WHICH view: one of the two views rendered before this one whose sorts are THROUGH on the device (an event query: a borrowed frame never
      // waits for anything) -- a caller that synchronises after every frame finds the previous view there, or the one before it.  A caller
      // that enqueues new views back to back is far ahead of the device, finds none and renders unordered as before: waiting for sorts that
      // cannot start before their frame ends serialises frame, sorts, frame, and measured 0.45-0.54 ms per irreg 1000 x 1000 view against
      // 0.43 unordered, from run to run; an OLDER sorted view predicts worse than no order at all (0.54).  (sync_policy = 1: the most recent
      // view whatever its state, behind a stream wait -- the same launches in every run, for tests and measurements; eager_sort = 0: the
      // pending sorts run in line, which waits for nothing either.)
      // (borrow = 1, auto: scenes read from L2 only.  A scene that lives in LDS has nothing pixel-stable to borrow -- rgbbox's long chains are chaotic
      // pixel by pixel -- and its unordered frame, whose tail the DONATE waves walk in 4-level treelets, is now as fast as one through a neighbour's
      // TILE order: rgbbox 1000 x 1000 0.462 ms unordered against 0.483 borrowed, 500 x 500 0.272 / 0.276; irreg 0.431 / 0.355, 0.273 / 0.229 --
      // profiles/r06/cold_probe_*.txt.  borrow = 2 / 3 force the tile order / the list for either kind.)
      const bool scene_in_lds = pl.lds_nodes == p.n_nodes && pl.lds_sph == p.n_sph;
      if (!use && ctx->borrow && !(ctx->borrow == 1 && scene_in_lds) && nframes == 1 && ctx->adaptive_order == 1) {
        TileOrder *from = nullptr;
        for (auto &o : ps->orders) {
          if (&o == to || !same_shape(o) || !(o.valid || o.sort_pending)) continue;
          const bool ok = ctx->sync_policy ? true : (o.stamp + 2 >= to->stamp && (o.sort_pending ? !ctx->eager_sort : sorts_complete(&o)));
          if (ok && (!from || o.stamp > from->stamp)) from = &o;
        }
        if (from) {
          if (int rc = sort_view(ctx, ps, from, p, pl)) return rc;
          use = from;
        }
      }
      borrowed = use != nullptr && use != to;
      // The record of a view is a deterministic function of the view, so the table is computed
      // once (after the view's first frame) and kept; adaptive_order == 2 re-records and
      // recomputes every frame (testing aid).
      // (a view whose tiles were first recorded by a batch has no per-pixel record yet: its first single frame records again)
      const bool px_can = px_static_ok && nframes == 1 && to->px_list != nullptr && to->px_elems >= static_cast<size_t>(p.rows_local) * p.w &&
                          to->cost_px_bytes >= static_cast<size_t>(h) * w && w < 65536 && p.rows_local < 65536 && (p.nshards == 1 || p.interleave);
      const bool rerecord = !to->valid || ctx->adaptive_order == 2 || (px_can && !to->px_valid);
      p.cost = rerecord ? to->cost : nullptr;
      p.cost_px = rerecord && px_can ? to->cost_px : nullptr;
      p.order = use ? use->order : nullptr;
      DeepPolicy dp;
      if (int rc = deep_policy(ctx, ps, (nframes == 1 && !borrowed) ? to : nullptr, pl.grid_full * pl.waves, &dp)) return rc;
      if (borrowed) dp = DeepPolicy{0, 0, ctx->deep_cap_log2, true};   // (a borrowed order: no tile holds its wave -- which tiles are deep is the other view's truth)
      // A view's FIRST frame (no order yet): every workgroup -- the half-size launch that serves a partly LDS-resident scene's
      // ordered frames best lets an unordered one wait for its late chains with half the chip (irreg, first frame: 700 x 700
      // 0.605 -> 0.545 ms, 1000 x 1000 0.714 -> 0.625, 1400 x 1400 0.909 -> 0.738; profiles/r04/exp/e7).
      if (!to->valid && nframes == 1 && ctx->adaptive_order == 1 && ctx->deep_class < 0 && ctx->grid_div == 0 && p.nchunks <= 32768) {
        dp.sparse = true;
      }
      // Small ORDERED single frames (2 048 .. 10 000 tiles) are what their last bounce chains take: the COLD instantiation hands a
      // wave's last three rays to the solo loop from inside the pooled loop -- rgbbox 500 x 500 0.280 -> 0.225 ms; neutral to +2 %
      // from 1000 x 1000 on, where it is not used (profiles/r04/exp/e7, e8).
      if (ctx->handover && nframes == 1 && to->valid && p.nchunks >= 2048 && p.nchunks <= 10000 && max_depth > 4 && pl.waves == 16 && ctx->solo &&
          ps->tl_depth == rtk::kTreeletDepth && ctx->deep_class < 0 && ctx->adaptive_order == 1)
        p.cold = 3;
      p.deep_class = dp.deep_class;
      p.deep_split = dp.deep_split;
      p.deep_cap_log2 = dp.cap_log2;
      if (dp.sparse && ctx->grid_div == 0 && pl.grid != pl.grid_full) {
        pl.grid = pl.grid_full;
        // the queue layout follows the grid that is launched -- as long as the view's order table (laid out for
        // `order_shards` shards) still fits it
        const int ns = (xq && (nframes == 1 || xq == 2) && pl.grid % rtk::kMaxShards == 0) ? rtk::kMaxShards : 1;
        const int il = ns > 1 && xq == 2;
        if ((il ? 1 : ns) == order_shards) { p.nshards = ns; p.interleave = il; }
      }
    }
    // Pixel tickets (the ORD instantiation): an ordered single frame of a view that has its pixel list draws from it -- every
    // workgroup is launched (the longest chains ride in waves of their own from t = 0: the work bounds the frame, not they).
    // (a BORROWED order goes through the other view's pixel list, holds and all; borrow = 2: through its TILE order.  Camera path view by view, mean of
    // the views behind the first, none / tiles / list / list without holds, with 2-level treelets, profiles/r06/exp/e4_borrow_modes_*.txt: irreg 500 x 500
    // 0.315 / 0.308 / 0.267 / 0.271 ms, 1000 x 1000 0.506 / 0.453 / 0.376 / 0.402, 1400 x 1400 0.620 / 0.556 / 0.498 / 0.516; rgbbox 0.335 / 0.313 / 0.343 / 0.347,
    // 0.561 / 0.508 / 0.551 / 0.553, 0.785 / 0.734 / 0.791 / 0.803.)
    const bool borrow_tiles_only = borrowed && ctx->borrow == 2;
    if (use && use->valid && use->px_valid && nframes == 1 && pl.waves == 16 && ctx->adaptive_order == 1 && !borrow_tiles_only &&
        (p.nshards == 1 || p.interleave) && use->px_elems >= static_cast<size_t>(p.rows_local) * p.w && px_static_ok) {
      p.px_list = use->px_list;
      p.px_hdr = reinterpret_cast<const int *>(use->px_list + use->px_elems);
      p.px_hold = (borrowed && ctx->borrow == 4) ? 0 : ctx->px_hold;   // (borrow = 4, testing: a borrowed list without its holds)
      p.px_prio = ctx->px_prio;
      p.cold = 0;
      p.solo = (use->px_solo && ctx->solo && ps->tl_depth == rtk::kTreeletDepth) ? 1 : 0;
      if (ctx->grid_div == 0 && pl.grid != pl.grid_full) {
        const int ns = (xq && pl.grid_full % rtk::kMaxShards == 0) ? rtk::kMaxShards : 1;
        const int il = ns > 1 && xq == 2;
        if (ns == 1 || il) { pl.grid = pl.grid_full; p.nshards = ns; p.interleave = il; }
      }
    }
    // An UNORDERED single frame (a view's first; every frame when adaptive_order is 0) ends long after its first waves have run
    // dry -- its long chains start whenever the raster reaches them.  The DONATE instantiation: a wave that cannot refill gives
    // the rays it is left with, at a bounce boundary, to sibling waves of its workgroup that have left the loop and wait, one ray
    // each, walked in the solo loop (LDS mailboxes, workgroup-scope atomics only).  First frames, profiles/r04/exp/e13, e14: irreg
    // 500 x 500 0.519 -> 0.373 ms, 1000 x 1000 0.588 -> 0.486, a rank's eighth of 4000 x 4000 0.83 / 0.92 -> 0.67 / 0.72, the
    // 10^6-sphere frame 1.74 -> 1.50, rgbbox 1000 x 1000 0.617 -> 0.549; ordered frames do not gain (within 1 % at every size)
    // and keep their kernels.  handover=2 (testing): every single frame, a wave offers its rays when it holds <= donate_max.
    // (... and a frame rendered through a BORROWED order / pixel list: the long chains that list places wrongly start late too)
    if (nframes == 1 && max_depth > 4 && pl.waves == 16 && ctx->solo && ps->tl_depth == rtk::kTreeletDepth &&
        (ctx->handover == 2 || (ctx->handover == 1 && (p.order == nullptr || borrowed)))) {
      p.cold = 0;
      p.donate = ctx->handover == 2 ? ctx->donate_max : 64;
    }
    // The sorts this launch depends on (they may still be running on the side streams): the list's when it draws pixel tickets, the tile order's when it draws
    // tiles; both when it records (the sorts read -- and the tile order's clears -- the record this frame writes) or borrows (its own view is new: nothing of it is in flight).
    if (use) {
      const bool records = p.cost != nullptr && use == to;
      if (int rc = await_view(ctx, use, p.px_hdr == nullptr || records, p.px_hdr != nullptr || records)) return rc;
    }
    bool first_order = false;
    if (ctx->first_order && nframes == 1 && p.order == nullptr && !p.px_hdr && (p.nshards == 1 || p.interleave) && p.tiles_y > 1 &&
        p.tiles_y <= 4096 && p.tiles_x <= 32768) {
      // a frame nothing is known about: not top to bottom (the kernel reads the table like a view's order, with no deep tiles)
      tick("view, sorts, borrow");
      if (int rc = get_first_order(ctx, p.tiles_x, p.tiles_y, &p.order)) return rc;
      tick("first order");
      p.deep_class = 0;
      first_order = true;
    }
    // Culling by the best hit so far (the CULL instantiations; lane_core.h: cull_limit): where the scene's and the
    // camera's guards pass.  Auto leaves wholly LDS-resident scenes alone: their walks are short and LDS-fast, and the limit's three
    // instructions per item cost more than the tests it saves (rgbbox 1000 x 1000: -3 % box tests; tools/cull_pooled.cpp).
    if (cull_allowed(ctx, ps, pl, p, cams_dev, nframes)) {
      p.cull = 1;
      p.cull_c2 = ps->cull.c2;
      p.cull_kappa = ps->cull.kappa;
    }
    tick("before launch");
    RT_HIP(ctx, rtk::launch_pooled(p, false, pl.grid, pl.waves, ctx->stream));
    tick("launch_pooled");
    {
      // (which instantiation launch_pooled picks, in its own order of precedence)
      const bool single_px = p.solo && p.nframes == 1 && p.order != nullptr && p.deep_class > 0 && p.deep_split == 6 && p.tl_log2 == rtk::kTreeletDepth;
      const char *inst = p.px_hdr ? (p.donate ? (p.solo ? "ORD+SOLO+DONATE" : "ORD+DONATE") : (p.solo ? "ORD+SOLO" : "ORD")) : (p.cold && pl.waves == 16) ? (single_px ? "COLD+SOLO" : "COLD")
                         : (p.donate && pl.waves == 16) ? (single_px ? "DONATE+SOLO" : "DONATE") : (single_px ? "SOLO" : "plain");
      char buf[256];
      std::snprintf(buf, sizeof buf, "family=pooled tickets=%s%s instantiation=%s%s%s frames=%d tiles=%d grid=%d waves=%d counters=%d%s deep_class=%d deep_split=%d recording=%d",
                    p.px_hdr ? "pixel-list" : first_order ? "tiles-bit-reversed" : (p.order ? "tiles-ordered" : "tiles-raster"), borrowed ? "(borrowed)" : "", inst, p.cull ? "+CULL" : "", p.spill ? "+SPILL" : "", p.nframes, p.nchunks, pl.grid, pl.waves, p.nshards,
                    p.interleave ? "(turns)" : "", p.px_hdr ? 0 : p.deep_class, p.px_hdr ? 0 : p.deep_split, p.cost ? (p.cost_px ? 2 : 1) : 0);
      ctx->last_launch = buf;
    }
    if (to && p.cost) {
      // This frame recorded the view's bounce chains.  The sorts that turn the record into the view's tile order and pixel list are
      // NOT launched here: a caller that never renders the view again (the reference's `render` keeps nothing between calls,
      // ray.fut:246; a camera path rendered view by view) should not pay for them -- ~0.07 ms behind a 1000 x 1000 frame.  They run
      // ahead of the view's next frame (sort_pending_record, above).
      to->sort_pending = true;
      to->sort_px = p.cost_px != nullptr;
      to->rec_out_skip = p.out_skip;
      // (eager_sort, the default: ... but they are LAUNCHED here, on the context's second stream, behind this frame: they run while the caller
      // synchronises and sets up its next call, off every frame's critical path -- the view's second frame no longer carries them, and a
      // new view can borrow this one's order at once)
      if (ctx->eager_sort && (stream_was_idle || ctx->sync_policy))
        if (int rc = sort_view(ctx, ps, to, p, pl)) return rc;
      tick("record: eager sorts");
    }
  }
  else {
    RT_HIP(ctx, rtk::launch_persistent(p, false, pl.grid, pl.waves, ctx->stream));
    ctx->last_launch = "family=persistent";
  }
  return 0;
}
using rti::enqueue_render;

// ------------------------------------------------------------------------------------ context
extern "C" void rt_context_destroy(rt_context *ctx);

extern "C" int rt_context_create(rt_context **out, int device, void *hip_stream, int use_caller_stream) {
  if (!out) return 1;
  *out = nullptr;
  auto ctx = std::make_unique<rt_context>();
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    std::fprintf(stderr, "libray_mi355x: no usable HIP device (%s); this library has no CPU path\n",
                 e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return 2;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= count) return 3;
  if (hipSetDevice(device) != hipSuccess) return 4;
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 5;
  // from here on a failure releases whatever was created so far
  auto bail = [&](int code) {
    rt_context_destroy(ctx.release());
    return code;
  };
  ctx->num_cu = prop.multiProcessorCount;
  ctx->lds_bytes = static_cast<int>(prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
                                                                           : prop.sharedMemPerBlock);
  ctx->name = prop.gcnArchName;
  if (use_caller_stream) {
    ctx->stream = static_cast<hipStream_t>(hip_stream);   // NULL is a valid handle: the default stream
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(6);
    ctx->own_stream = true;
  }
  if (hipMalloc(reinterpret_cast<void **>(&ctx->queue_dev), sizeof(unsigned) * rtk::kQueueDwords) != hipSuccess) return bail(7);
  if (hipMemset(ctx->queue_dev, 0, sizeof(unsigned) * rtk::kQueueDwords) != hipSuccess) return bail(7);
  if (hipMalloc(reinterpret_cast<void **>(&ctx->stats_dev), 256) != hipSuccess) return bail(7);
  if (hipMemset(ctx->stats_dev, 0, 256) != hipSuccess) return bail(7);
  // the memsets run on the null stream, the context's own stream does not wait for it: the counters must BE zero
  // before the first launch draws a ticket
  if (hipStreamSynchronize(nullptr) != hipSuccess) return bail(7);
  if (hipMalloc(reinterpret_cast<void **>(&ctx->arena), kArenaGranules * kGranule) != hipSuccess) return bail(7);
  ctx->arena_used.assign(kArenaGranules, 0);
  if (hipHostMalloc(reinterpret_cast<void **>(&ctx->pinned), rtk::gpu_build_pinned_bytes(), hipHostMallocDefault) != hipSuccess)
    return bail(7);
  if (hipHostMalloc(reinterpret_cast<void **>(&ctx->stage), kStageBytes, hipHostMallocDefault) != hipSuccess) return bail(7);
  // The sort streams, their event and the sorts' scratch: created here, not at a view's first recording frame -- a HIP stream is a
  // hardware queue (milliseconds to create), and the reference's harness times its first render call with the rest (main.c:107-124).
  if (hipStreamCreateWithFlags(&ctx->sort_stream, hipStreamNonBlocking) != hipSuccess) return bail(6);
  if (hipStreamCreateWithFlags(&ctx->sort_stream_px, hipStreamNonBlocking) != hipSuccess) return bail(6);
  if (hipEventCreateWithFlags(&ctx->rec_event, hipEventDisableTiming) != hipSuccess) return bail(6);
  if (hipMalloc(reinterpret_cast<void **>(&ctx->order_scratch), sizeof(int) * rtk::kOrderScratchInts) != hipSuccess) return bail(7);
  if (hipMalloc(reinterpret_cast<void **>(&ctx->px_scratch), sizeof(int) * rtk::px_scratch_ints()) != hipSuccess) return bail(7);
  if (hipHostMalloc(reinterpret_cast<void **>(&ctx->class_slab), sizeof(int) * kClassSlotInts * kClassSlots * kClassChunks, hipHostMallocDefault) != hipSuccess) return bail(7);
  rtk::warm_render_kernels();
  rtk::warm_build_kernels();
  if (rtk::warm_scratch(ctx->stream, ctx->order_scratch) != hipSuccess) return bail(6);
  if (const char *v = std::getenv("RT_VARIANT")) ctx->variant = std::atoi(v);
  // (test aids: the whole suite under the other queue layouts)
  if (const char *v = std::getenv("RT_XCD_QUEUES")) ctx->xcd_queues = std::max(-1, std::min(2, std::atoi(v)));
  if (const char *v = std::getenv("RT_TPT_LOG2")) ctx->tpt_log2 = std::max(-1, std::min(4, std::atoi(v)));
  if (const char *v = std::getenv("RT_STATIC_FIRST")) ctx->static_first = std::atoi(v) != 0;
  *out = ctx.release();
  return 0;
}

extern "C" void rt_context_destroy(rt_context *ctx) {
  if (!ctx) return;
  if (ctx->group) rti::group_destroy(ctx);
  (void)hipSetDevice(ctx->device);
  drain_streams(ctx);
  if (ctx->sort_stream) (void)hipStreamDestroy(ctx->sort_stream);
  if (ctx->sort_stream_px) (void)hipStreamDestroy(ctx->sort_stream_px);
  if (ctx->rec_event) (void)hipEventDestroy(ctx->rec_event);
  for (auto &t : ctx->uv) pool_free(ctx, reinterpret_cast<char *>(t.u), t.bytes);
  for (auto &t : ctx->first_orders) pool_free(ctx, reinterpret_cast<char *>(t.order), t.bytes);
  if (ctx->class_slab) (void)hipHostFree(ctx->class_slab);
  if (ctx->cams_dev) (void)hipFree(ctx->cams_dev);
  if (ctx->cams_host) (void)hipHostFree(ctx->cams_host);
  if (ctx->cams_event) (void)hipEventDestroy(ctx->cams_event);
  if (ctx->queue_dev) (void)hipFree(ctx->queue_dev);
  if (ctx->spill_dev) (void)hipFree(ctx->spill_dev);
  if (ctx->order_scratch) (void)hipFree(ctx->order_scratch);
  if (ctx->px_scratch) (void)hipFree(ctx->px_scratch);
  if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
  for (auto &b : ctx->pool) (void)hipFree(b.p);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int rt_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" const char *rt_last_error(const rt_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" const char *rt_context_last_launch(const rt_context *ctx) { return ctx ? ctx->last_launch.c_str() : ""; }

extern "C" int rt_context_sync(rt_context *ctx) {
  RT_LOCK(ctx);
  if (!ctx) return 1;
  if (ctx->group) {   // every frame ends on the parent's stream, synchronised last
    const int grc = rti::group_sync(ctx);
    if (!grc) rti::group_mark_synced(ctx);
    return grc;
  }
  // Frames take well under a millisecond: poll briefly (a frame's worth) before falling back to the
  // blocking wait, whose wake-up latency alone is a sizeable fraction of a frame.  The poll is short on
  // purpose: a process with many contexts must not burn a host core per context.
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t q;
  for (;;) {
    q = hipStreamQuery(ctx->stream);
    if (q != hipErrorNotReady) break;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(800)) {
      q = hipStreamSynchronize(ctx->stream);
      break;
    }
  }
  if (q == hipSuccess) {
    ctx->synced_since_render = true;
    return 0;
  }
  // A launch that died leaves the ticket counters non-zero (its last wave never zeroed them): every later
  // frame would draw out-of-range tickets and silently render nothing.  Re-zero them.
  (void)hipGetLastError();
  (void)hipMemset(ctx->queue_dev, 0, sizeof(unsigned) * rtk::kQueueDwords);
  return hip_fail(ctx, q, "stream synchronisation (the ticket counters were reset)");
}

extern "C" int rt_context_set_variant(rt_context *ctx, int variant) {
  RT_LOCK(ctx);
  if (!ctx) return 1;
  if (variant < RT_VARIANT_AUTO || variant > RT_VARIANT_POOLED) return fail(ctx, "unknown variant");
  ctx->variant = variant;
  if (ctx->group) return rti::group_set_variant(ctx, variant);
  return 0;
}

extern "C" int rt_context_set_option(rt_context *ctx, const char *name, int64_t value) {
  RT_LOCK(ctx);
  if (!ctx || !name) return 1;
  if (ctx->group) {
    const int grc = rti::group_set_option(ctx, name, value);
    if (grc >= 0) return grc;   // a group-only option, or a child refused the value
  }
  const std::string k(name);
  const int v = static_cast<int>(value);
  if (k == "waves_per_wg") {
    if (v != 0 && v != 4 && v != 8 && v != 12 && v != 16) return fail(ctx, "waves_per_wg must be 0 (auto), 4, 8, 12 or 16");
    ctx->waves_per_wg = v;
  } else if (k == "wgs_per_cu") {
    if (v < 1 || v > 8) return fail(ctx, "wgs_per_cu must be 1..8");
    ctx->wgs_per_cu = v;
  } else if (k == "thr_shade") {
    ctx->thr_shade = std::max(1, std::min(64, v));
  } else if (k == "thr_leaf") {
    ctx->thr_leaf = std::max(1, std::min(64, v));
  } else if (k == "lmax") {
    if (v < 2 || v > 32) return fail(ctx, "lmax must be 2..32");
    ctx->lmax = v;
  } else if (k == "lds_scene_bytes") {
    ctx->lds_scene_bytes = v;
  } else if (k == "lds_sph_first") {
    ctx->lds_sph_first = v != 0;
  } else if (k == "grid_div") {
    ctx->grid_div = std::max(0, std::min(64, v));
  } else if (k == "prio_depth") {
    ctx->prio_depth = std::max(0, v);
  } else if (k == "gpu_build") {
    ctx->gpu_build = v != 0;
  } else if (k == "adaptive_order") {
    ctx->adaptive_order = v;
  } else if (k == "box2") {
    ctx->box2 = v != 0;
  } else if (k == "look_max") {
    ctx->look_max = std::max(0, std::min(64, v));
  } else if (k == "handover") {
    ctx->handover = std::max(0, std::min(2, v));
  } else if (k == "donate_max") {
    ctx->donate_max = std::max(1, std::min(64, v));
  } else if (k == "solo") {
    ctx->solo = v != 0;
  } else if (k == "treelet") {
    if (v < 1 || v > rtk::kTreeletMaxDepth) return fail(ctx, "treelet (host builder: levels per treelet) must be 1 .. 5");
    ctx->treelet = v;
  } else if (k == "trace_part") {
    ctx->trace_part = v;
  } else if (k == "trace_nparts") {
    ctx->trace_nparts = v;
  } else if (k == "trace_solo") {
    ctx->trace_solo = v != 0;
  } else if (k == "ray_planes") {
    if (v != 0 && v != 2 && v != 3) return fail(ctx, "ray_planes must be 0 (auto), 2 or 3");
    ctx->ray_planes = v;
  } else if (k == "deep_class") {
    ctx->deep_class = std::min(8, std::max(-1, v));   // -1: chosen per view (deep_policy)
  } else if (k == "deep_cap_log2") {
    ctx->deep_cap_log2 = std::min(8, std::max(0, v));
  } else if (k == "deep_split") {
    ctx->deep_split = std::min(6, std::max(0, v));
  } else if (k == "xcd_queues") {
    if (v < -1 || v > 2) return fail(ctx, "xcd_queues must be -1 (auto), 0 (one counter), 1 (a strip of tile columns per counter) or 2 (counters take turns)");
    ctx->xcd_queues = v;
  } else if (k == "tpt_log2") {
    if (v < -1 || v > 4) return fail(ctx, "tpt_log2 must be -1 (auto) or 0..4");
    ctx->tpt_log2 = v;
  } else if (k == "static_first") {
    ctx->static_first = v != 0;
  } else if (k == "first_order") {
    ctx->first_order = v != 0;
  } else if (k == "pixel_order") {
    ctx->pixel_order = std::max(0, std::min(2, v));
  } else if (k == "px_solo" || k == "px_w8" || k == "px_w16" || k == "px_w32") {
    ctx->px_thr[k == "px_solo" ? 0 : k == "px_w8" ? 1 : k == "px_w16" ? 2 : 3] = std::max(0, std::min(255, v));   // px_solo = 0: the model cuts the classes
  } else if (k == "px_g1" || k == "px_g8" || k == "px_g16" || k == "px_g32" || k == "px_g64") {
    ctx->px_g[k == "px_g1" ? 0 : k == "px_g8" ? 1 : k == "px_g16" ? 2 : k == "px_g32" ? 3 : 4] = std::max(0, std::min(100000, v));   // 0.1 us per bounce; 0 = the built-in figure
  } else if (k == "px_max_tiles") {
    ctx->px_max_tiles = std::max(0, v);
  } else if (k == "px_ray_ns") {
    ctx->px_ray_ns = std::max(0, std::min(100000, v));
  } else if (k == "px_zip") {
    ctx->px_zip = v != 0;
  } else if (k == "px_prio") {
    ctx->px_prio = std::max(0, std::min(3, v));
  } else if (k == "px_hold") {
    ctx->px_hold = v & 0x1f;
  } else if (k == "px_solo_div") {
    ctx->px_solo_div = std::max(1, std::min(4096, v));
  } else if (k == "eager_sort") {
    drain_streams(ctx);            // (sorts in flight on either stream finish under the setting they were launched with)
    ctx->eager_sort = v != 0;
  } else if (k == "borrow") {
    ctx->borrow = std::max(0, std::min(4, v));
  } else if (k == "stack_cap") {
    if (v != 0 && v != rtk::kSpillCapbTest) return fail(ctx, "stack_cap must be 0 or 192");
    ctx->stack_cap = v;
  } else if (k == "wide_waves") {
    ctx->wide_waves = std::max(0, std::min(2, v));
  } else if (k == "cull") {
    ctx->cull = std::max(-1, std::min(1, v));
  } else if (k == "sync_policy") {
    ctx->sync_policy = v != 0;
  } else {
    return fail(ctx, "unknown option: " + k);
  }
  return 0;
}

extern "C" int rt_context_device_info(const rt_context *ctx, int *device, int *num_cu, int *lds_bytes, char *name,
                                      int name_len) {
  RT_LOCK(const_cast<rt_context *>(ctx));
  if (!ctx) return 1;
  if (device) *device = ctx->device;
  if (num_cu) *num_cu = ctx->num_cu;
  if (lds_bytes) *lds_bytes = ctx->lds_bytes;
  if (name && name_len > 0) {
    std::strncpy(name, ctx->name.c_str(), static_cast<size_t>(name_len) - 1);
    name[name_len - 1] = 0;
  }
  return 0;
}

// ------------------------------------------------------------------------------------ scenes
static int new_scene(rt_context *ctx, rt_scene **out, rt::SceneDesc &&d) {
  RT_LOCK(ctx);
  if (!ctx || !out) return fail(ctx, "null argument");
  auto *s = new rt_scene;
  s->desc = std::move(d);
  *out = s;
  return 0;
}
extern "C" int rt_scene_rgbbox(rt_context *ctx, rt_scene **out) { return new_scene(ctx, out, rt::make_rgbbox()); }
extern "C" int rt_scene_irreg(rt_context *ctx, rt_scene **out) { return new_scene(ctx, out, rt::make_floor(100, 600.0f)); }
extern "C" int rt_scene_floor(rt_context *ctx, rt_scene **out, int n, float k) {
  RT_LOCK(ctx);
  if (n < 2 || n > 4096) return fail(ctx, "floor scene: n out of range (2..4096)");
  return new_scene(ctx, out, rt::make_floor(n, k));
}
extern "C" int rt_scene_from_spheres(rt_context *ctx, rt_scene **out, const float *spheres7, int64_t n,
                                     const float look_from[3], const float look_at[3], float fov) {
  RT_LOCK(ctx);
  if (!spheres7 || !look_from || !look_at) return fail(ctx, "null argument");
  if (n < 2 || n > (int64_t(1) << rtk::kMaxSpheresLog2)) return fail(ctx, "scene needs 2 .. 2^26 spheres");   // (treelet.h: the builders' index fields)
  rt::SceneDesc d;
  d.spheres.resize(static_cast<size_t>(n));
  std::memcpy(d.spheres.data(), spheres7, sizeof(rt::Sphere) * static_cast<size_t>(n));
  std::copy(look_from, look_from + 3, d.look_from);
  std::copy(look_at, look_at + 3, d.look_at);
  d.fov = fov;
  return new_scene(ctx, out, std::move(d));
}
extern "C" int64_t rt_scene_num_spheres(const rt_scene *scene) {
  return scene ? static_cast<int64_t>(scene->desc.spheres.size()) : 0;
}
extern "C" int rt_scene_free(rt_context *ctx, rt_scene *scene) {
  RT_LOCK(ctx);
  if (scene)
    for (auto &c : scene->copies) {
      (void)hipSetDevice(c.device);
      (void)hipFree(c.p);
    }
  if (ctx) (void)hipSetDevice(ctx->device);
  delete scene;
  return 0;
}

// ------------------------------------------------------------------------------------ prepare_scene
extern "C" int rt_prepare_scene(rt_context *ctx, rt_prepared **out, int64_t h, int64_t w, const rt_scene *scene) {
  RT_LOCK(ctx);
  if (!ctx || !out || !scene) return fail(ctx, "null argument");
  if (h <= 0 || w <= 0) return fail(ctx, "image size must be positive");
  if (scene->desc.spheres.size() < 2) return fail(ctx, "scene needs at least 2 spheres");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  (void)hipGetLastError();   // (as in enqueue_render)
  auto ps = std::make_unique<rt_prepared>();
  const size_t n = scene->desc.spheres.size(), ni = n - 1;
  ps->n = static_cast<int64_t>(n);
  ps->home = ctx;
  ps->h = h; ps->w = w;
  ps->cam = rt::scene_camera(scene->desc, h, w);
  int rc = 0;
  hipError_t e = hipSuccess;
  // one device allocation for every array of the prepared scene (256-byte aligned pieces)
  {
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~size_t(255); return at; };
    const size_t o_L7 = carve(n * 28), o_bmin = carve(ni * 12), o_bmax = carve(ni * 12), o_left = carve(ni * 4),
                 o_right = carve(ni * 4), o_parent = carve(ni * 4), o_nodes = carve(ni * 32), o_nodes64 = carve(ni * 64),
                 o_sph = carve(n * 16), o_col = carve(n * 16);
    ps->block_bytes = off;
    RT_HIP(ctx, pool_alloc(ctx, &ps->block, &ps->block_bytes));
    char *b = ps->block;
    ps->L7 = reinterpret_cast<float *>(b + o_L7); ps->bmin = reinterpret_cast<float *>(b + o_bmin);
    ps->bmax = reinterpret_cast<float *>(b + o_bmax); ps->left = reinterpret_cast<int32_t *>(b + o_left);
    ps->right = reinterpret_cast<int32_t *>(b + o_right); ps->parent = reinterpret_cast<int32_t *>(b + o_parent);
    ps->nodes = reinterpret_cast<float4 *>(b + o_nodes); ps->nodes64 = reinterpret_cast<float4 *>(b + o_nodes64);
    ps->sph = reinterpret_cast<float4 *>(b + o_sph); ps->col = reinterpret_cast<float4 *>(b + o_col);
  }
  // multi-device context: the other devices build their replicas while this one builds its own (no early return from
  // here to group_prepare_end)
  if (ctx->group) rti::group_prepare_begin(ctx, ps.get(), h, w, scene);
  auto put = [&](void *dst, const void *src, size_t bytes) {
    if (!rc && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, "hipMemcpyAsync failed");
  };
  if (ctx->gpu_build) {
    // ---- BVH construction on the GPU (bvh_build.hip): upload the spheres, build in place ----
    // (a multi-device prepare_scene uploads from one host thread per device: the lock covers the look-up and the
    // insertion only, not the upload and its stream synchronisation)
    float *scene_dev = nullptr;
    {
      std::lock_guard<std::mutex> lock(scene->mu);
      for (const auto &c : scene->copies)
        if (c.device == ctx->device) scene_dev = c.p;
    }
    if (!scene_dev) {
      const size_t sbytes = n * sizeof(rt::Sphere);
      float *fresh = nullptr;
      e = hipMalloc(reinterpret_cast<void **>(&fresh), sbytes + 16);
      if (e == hipSuccess) {
        if (sbytes + 16 <= kStageBytes) {
          // small scene: through the pinned staging block and a copy kernel (a pageable hipMemcpy of
          // a few hundred KB costs milliseconds)
          std::memcpy(ctx->stage, scene->desc.spheres.data(), sbytes);
          e = rtk::gpu_copy_from_pinned(fresh, ctx->stage, sbytes, ctx->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        } else {
          e = hipMemcpy(fresh, scene->desc.spheres.data(), sbytes, hipMemcpyHostToDevice);
        }
      }
      if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(scene->mu);
        for (const auto &c : scene->copies)
          if (c.device == ctx->device) scene_dev = c.p;     // (another context of this device was faster)
        if (!scene_dev) {
          scene->copies.push_back({ctx->device, fresh});
          scene_dev = fresh;
          fresh = nullptr;
        }
      }
      if (fresh) (void)hipFree(fresh);
    }
    size_t tmp_bytes = rtk::gpu_build_scratch_bytes(static_cast<int>(n));
    char *tmp = nullptr;
    if (e == hipSuccess) e = pool_alloc(ctx, &tmp, &tmp_bytes);
    if (e == hipSuccess) {
      rtk::GpuBvhOut o{ps->L7, ps->bmin, ps->bmax, ps->left, ps->right, ps->parent, ps->nodes, ps->nodes64, ps->sph, ps->col};
      e = rtk::gpu_build_bvh(scene_dev, static_cast<int>(n), o, tmp, ctx->pinned, ctx->stream, &ps->height, ps->root_lo,
                             ps->root_hi);
      ps->tl_depth = rtk::kTreeletDepth;
    }
    if (tmp) {
      (void)hipStreamSynchronize(ctx->stream);
      pool_free(ctx, tmp, tmp_bytes);
    }
  } else {
    const rt::Lbvh bvh = rt::build_lbvh(scene->desc.spheres);
    const rt::TravLayout tl = rt::make_trav_layout(bvh, ctx->treelet);
    ps->height = tl.height;
    ps->tl_depth = tl.treelet_depth;
    put(ps->L7, bvh.L.data(), n * sizeof(rt::Sphere));
    put(ps->bmin, bvh.bmin.data(), ni * 3 * sizeof(float));
    put(ps->bmax, bvh.bmax.data(), ni * 3 * sizeof(float));
    put(ps->left, bvh.left.data(), ni * sizeof(int32_t));
    put(ps->right, bvh.right.data(), ni * sizeof(int32_t));
    put(ps->parent, bvh.parent.data(), ni * sizeof(int32_t));
    put(ps->nodes, tl.nodes.data(), ni * sizeof(rt::TravNode));
    put(ps->nodes64, tl.nodes64.data(), ni * 64);
    std::copy(tl.root_lo, tl.root_lo + 3, ps->root_lo);
    std::copy(tl.root_hi, tl.root_hi + 3, ps->root_hi);
    put(ps->sph, tl.sph.data(), n * 16);
    put(ps->col, tl.col.data(), n * 16);
    // the host staging vectors die at scope exit: drain the copies first
    e = hipStreamSynchronize(ctx->stream);
  }
  if (!rc && e == hipSuccess) {
    // culling by the best hit (lane_core.h: cull_limit): the scene's guards and constants; the spheres' side once per scene
    std::lock_guard<std::mutex> lock(scene->mu);
    if (!scene->cull_done) {
      scene->cull = rt::cull_scene_constants(scene->desc.spheres, 0);
      scene->cull_done = true;
    }
    ps->cull = scene->cull;
    ps->cull.ok = ps->cull.ok && ps->height <= static_cast<int>(log2f(static_cast<float>(n))) + 2;   // every box contains its subtree (bvh.fut:47)
  }
  // (a tree taller than 15 levels: the twenty-wave shape's box stacks may spill -- their regions are allocated here, not inside a render call)
  if (!rc && e == hipSuccess && ctx->wide_waves != 0 && ps->height > 15 && n < (int64_t(1) << 22)) rc = ensure_spill(ctx, 64 * (ps->height + 2));
  const int grc = ctx->group ? rti::group_prepare_end(ctx, ps.get()) : 0;   // (joins the replica builds whatever happened here)
  if (rc || e != hipSuccess) {
    rt_prepared_free(ctx, ps.release());
    return rc ? rc : hip_fail(ctx, e, "rt_prepare_scene");
  }
  if (grc) {
    rt_prepared_free(ctx, ps.release());
    return grc;
  }
  *out = ps.release();
  return 0;
}

extern "C" int rt_prepared_free(rt_context *ctx, rt_prepared *ps) {
  RT_LOCK(ctx);
  if (!ps) return 0;
  if (!ps->replicas.empty()) rti::group_prepared_free(ctx, ps);
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    drain_streams(ctx);
  }
  pool_free(ctx, ps->block, ps->block_bytes);
  for (auto &o : ps->orders) {
    free_view_block(ctx, o);
    if (o.classes_event) (void)hipEventDestroy(o.classes_event);
    if (o.sort_event) (void)hipEventDestroy(o.sort_event);
    if (o.sort_event_px) (void)hipEventDestroy(o.sort_event_px);
  }
  if (ps->classes_pinned && ps->classes_chunk < 0) (void)hipHostFree(ps->classes_pinned);
  if (ps->classes_chunk >= 0 && ps->classes_owner) ps->classes_owner->class_chunks_used &= ~(1ull << ps->classes_chunk);
  delete ps;
  return 0;
}

extern "C" int64_t rt_prepared_num_spheres(const rt_prepared *ps) { return ps ? ps->n : 0; }
extern "C" int32_t rt_prepared_height(const rt_prepared *ps) { return ps ? ps->height : 0; }

extern "C" int rt_prepared_get_bvh(rt_context *ctx, const rt_prepared *ps, float *L7, float *bmin, float *bmax,
                                   int32_t *left, int32_t *right, int32_t *parent) {
  RT_LOCK(ctx);
  if (!ctx || !ps) return fail(ctx, "null argument");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t n = static_cast<size_t>(ps->n), ni = n - 1;
  if (L7) RT_HIP(ctx, hipMemcpy(L7, ps->L7, n * 28, hipMemcpyDeviceToHost));
  if (bmin) RT_HIP(ctx, hipMemcpy(bmin, ps->bmin, ni * 12, hipMemcpyDeviceToHost));
  if (bmax) RT_HIP(ctx, hipMemcpy(bmax, ps->bmax, ni * 12, hipMemcpyDeviceToHost));
  if (left) RT_HIP(ctx, hipMemcpy(left, ps->left, ni * 4, hipMemcpyDeviceToHost));
  if (right) RT_HIP(ctx, hipMemcpy(right, ps->right, ni * 4, hipMemcpyDeviceToHost));
  if (parent) RT_HIP(ctx, hipMemcpy(parent, ps->parent, ni * 4, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int rt_prepared_get_camera(rt_context *ctx, const rt_prepared *ps, float cam12[12]) {
  RT_LOCK(ctx);
  if (!ctx || !ps || !cam12) return fail(ctx, "null argument");
  std::memcpy(cam12, &ps->cam, sizeof(rt::Camera));
  return 0;
}

// ------------------------------------------------------------------------------------ render
// One frame (or one part of it) on the context: a multi-device context fans a WHOLE frame out over its
// devices (multi_gpu.cpp); it does not nest a caller's partition inside its own.
static int render_entry(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                        int32_t part, int32_t nparts, int32_t *out_dev, const float *cam12) {
  RT_LOCK(ctx);
  if (ctx && ctx->group) {
    if (!ps) return fail(ctx, "null prepared scene");
    if (part != 0 || nparts != 1) return fail(ctx, "a multi-device context renders whole frames: it partitions them itself");
    if (max_depth < 0) return fail(ctx, "negative max_depth");
    return rti::group_render(ctx, ps, h, w, max_depth, out_dev, cam12);
  }
  return enqueue_render(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, out_dev, false, cam12);
}

extern "C" int rt_render(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t *out_dev) {
  return render_entry(ctx, ps, h, w, 50, 8, 0, 1, out_dev, nullptr);
}

extern "C" int rt_render_part(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                              int32_t rows_per_tile, int32_t part, int32_t nparts, int32_t *out_dev) {
  return render_entry(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, out_dev, nullptr);
}

extern "C" int rt_render_image(rt_context *ctx, const rt_prepared *objs, int64_t width, int64_t height, const float cam12[12],
                               int32_t max_depth, int32_t rows_per_tile, int32_t part, int32_t nparts, int32_t *out_dev) {
  return render_entry(ctx, objs, height, width, max_depth, rows_per_tile, part, nparts, out_dev, cam12);
}

// N frames of one prepared scene in ONE launch of the pooled kernel (its ticket queue simply runs over the tiles of
// all frames, so the waves stay full across frame boundaries and the launch's fill and drain are paid once).
// The batch's cameras on the context's device.  The caller's array is copied into the context's own pinned block first:
// the asynchronous upload then never reads memory the caller may already have changed or freed (pinned caller memory
// would otherwise be read when the copy EXECUTES).
int rti::stage_cams(rt_context *ctx, const float *cams12, int32_t nframes, const float **cams_dev) {
  RT_LOCK(ctx);
  *cams_dev = nullptr;
  if (!cams12) return 0;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(float) * 12 * static_cast<size_t>(nframes);
  if (bytes > ctx->cams_bytes) {
    // grow: the stream is drained first, so nothing reads the old blocks any more; the event object is kept (it is
    // re-recorded below) -- only its "an upload is pending" meaning lapses with the old pinned block
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cams_dev) (void)hipFree(ctx->cams_dev);
    if (ctx->cams_host) (void)hipHostFree(ctx->cams_host);
    ctx->cams_dev = nullptr;
    ctx->cams_host = nullptr;
    ctx->cams_bytes = 0;
    ctx->cams_event_valid = false;
    RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->cams_dev), bytes));
    RT_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->cams_host), bytes, hipHostMallocDefault));
    ctx->cams_bytes = bytes;
  } else if (ctx->cams_event_valid) {
    RT_HIP(ctx, hipEventSynchronize(ctx->cams_event));   // the previous batch's upload has read the pinned block
  }
  std::memcpy(ctx->cams_host, cams12, bytes);
  RT_HIP(ctx, hipMemcpyAsync(ctx->cams_dev, ctx->cams_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  if (!ctx->cams_event) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->cams_event, hipEventDisableTiming));
  RT_HIP(ctx, hipEventRecord(ctx->cams_event, ctx->stream));
  ctx->cams_event_valid = true;
  *cams_dev = ctx->cams_dev;
  return 0;
}

extern "C" int rt_render_batch(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                               int32_t part, int32_t nparts, int32_t nframes, const float *cams12, int64_t frame_stride,
                               int32_t *out_dev) {
  RT_LOCK(ctx);
  if (!ctx || !ps) return fail(ctx, "null context or prepared scene");
  if (nframes < 1 || nframes > 4096) return fail(ctx, "rt_render_batch: 1 .. 4096 frames");
  if (ctx->group) {   // every device renders its rows of ALL the frames in one launch; one gather, one assembly launch
    if (part != 0 || nparts != 1) return fail(ctx, "a multi-device context renders whole frames: it partitions them itself");
    if (max_depth < 0) return fail(ctx, "negative max_depth");
    return rti::group_render(ctx, ps, h, w, max_depth, out_dev, nullptr, nframes, frame_stride, cams12);
  }
  const float *cams_dev = nullptr;
  if (int rc = rti::stage_cams(ctx, cams12, nframes, &cams_dev)) return rc;
  return enqueue_render(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, out_dev, false, nullptr, nframes, frame_stride, cams_dev);
}

// A part's rows of `nframes` frames stored straight into the FULL images (no packed part buffer, no gather, no assembly):
// image_dev may be memory of another device of the node -- a peer allocation, or another process's buffer mapped with
// rt_ipc_import -- and the pixel stores then are the framebuffer exchange, travelling over xGMI while the frame is traced.
extern "C" int rt_render_part_inplace(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                                      int32_t part, int32_t nparts, int32_t nframes, const float *cams12, int64_t frame_stride,
                                      int32_t *image_dev) {
  RT_LOCK(ctx);
  if (!ctx || !ps) return fail(ctx, "null context or prepared scene");
  if (ctx->group) return fail(ctx, "a multi-device context renders whole frames: it partitions them itself (option gather=3 stores in place)");
  if (nframes < 1 || nframes > 4096) return fail(ctx, "rt_render_part_inplace: 1 .. 4096 frames");
  if (nframes == 1)   // (one frame: its camera travels in the kernel arguments)
    return enqueue_render(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, image_dev, false, cams12, 1, h * w, nullptr, true);
  const float *cams_dev = nullptr;
  if (int rc = rti::stage_cams(ctx, cams12, nframes, &cams_dev)) return rc;
  return enqueue_render(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, image_dev, false, nullptr, nframes, frame_stride, cams_dev, true);
}

// Sharing a device buffer between the processes of one node (one process per GPU): the owner exports the allocation
// (rt_device_alloc's pointer, i.e. the base of a hipMalloc block) as 64 opaque bytes, the other ranks import them and get a
// device pointer they can hand to rt_render_part_inplace.
extern "C" int rt_ipc_export(rt_context *ctx, void *dev, unsigned char handle64[64]) {
  RT_LOCK(ctx);
  if (!ctx || !dev || !handle64) return fail(ctx, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI passes an IPC handle as 64 bytes");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipIpcMemHandle_t hnd;
  RT_HIP(ctx, hipIpcGetMemHandle(&hnd, dev));
  std::memcpy(handle64, &hnd, 64);
  return 0;
}
extern "C" int rt_ipc_import(rt_context *ctx, const unsigned char handle64[64], void **out_dev) {
  RT_LOCK(ctx);
  if (!ctx || !handle64 || !out_dev) return fail(ctx, "null argument");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  hipIpcMemHandle_t hnd;
  std::memcpy(&hnd, handle64, 64);
  RT_HIP(ctx, hipIpcOpenMemHandle(out_dev, hnd, hipIpcMemLazyEnablePeerAccess));
  return 0;
}
extern "C" int rt_ipc_close(rt_context *ctx, void *imported_dev) {
  RT_LOCK(ctx);
  if (!ctx) return 1;
  if (!imported_dev) return 0;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipIpcCloseMemHandle(imported_dev));
  return 0;
}

extern "C" int64_t rt_part_rows(int64_t h, int32_t rows_per_tile, int32_t part, int32_t nparts) {
  return rt::part_rows(h, rows_per_tile, part, nparts);
}

extern "C" int rt_place_part(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t part, int32_t nparts,
                             const int32_t *part_dev, int32_t *image_dev) {
  RT_LOCK(ctx);
  if (!ctx || !part_dev || !image_dev) return fail(ctx, "null argument");
  if (rows_per_tile <= 0 || nparts <= 0 || part < 0 || part >= nparts) return fail(ctx, "bad row-tile partition");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  const int rows_local = static_cast<int>(rt::part_rows(h, rows_per_tile, part, nparts));
  RT_HIP(ctx, rtk::launch_place_part(part_dev, image_dev, static_cast<int>(w), rows_local, rows_per_tile, part, nparts,
                                     ctx->stream));
  return 0;
}

extern "C" int rt_place_parts_strided(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t nparts,
                                      int64_t part_stride, const int32_t *stacked_dev, int32_t *image_dev) {
  RT_LOCK(ctx);
  if (!ctx || !stacked_dev || !image_dev) return fail(ctx, "null argument");
  if (rows_per_tile <= 0 || nparts <= 0 || h <= 0 || w <= 0) return fail(ctx, "bad row-tile partition");
  int64_t need = 0;
  for (int p = 0; p < nparts; ++p) need = std::max<int64_t>(need, rt::part_rows(h, rows_per_tile, p, nparts));
  if (part_stride < need * w) return fail(ctx, "part stride smaller than the largest part");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, rtk::launch_place_all(stacked_dev, image_dev, static_cast<int>(w), static_cast<int>(h), rows_per_tile, nparts,
                                    static_cast<size_t>(part_stride), ctx->stream));
  return 0;
}

extern "C" int rt_place_parts_batch(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t nparts, int64_t part_stride,
                                    int32_t nframes, int64_t frame_stride_in, int64_t frame_stride_out, const int32_t *stacked_dev,
                                    int32_t *images_dev) {
  RT_LOCK(ctx);
  if (!ctx || !stacked_dev || !images_dev) return fail(ctx, "null argument");
  if (rows_per_tile <= 0 || nparts <= 0 || h <= 0 || w <= 0 || nframes < 1) return fail(ctx, "bad row-tile partition");
  int64_t need = 0;
  for (int p = 0; p < nparts; ++p) need = std::max<int64_t>(need, rt::part_rows(h, rows_per_tile, p, nparts));
  if (frame_stride_in < need * w || part_stride < (nframes - 1) * frame_stride_in + need * w || frame_stride_out < h * w)
    return fail(ctx, "strides smaller than the parts / frames they separate");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, rtk::launch_place_all(stacked_dev, images_dev, static_cast<int>(w), static_cast<int>(h), rows_per_tile, nparts,
                                    static_cast<size_t>(part_stride), ctx->stream, nframes, static_cast<size_t>(frame_stride_in),
                                    static_cast<size_t>(frame_stride_out)));
  return 0;
}

extern "C" int rt_place_parts(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t nparts, int64_t pad_rows,
                              const int32_t *stacked_dev, int32_t *image_dev) {
  if (pad_rows < 0 || w <= 0) return fail(ctx, "bad row-tile partition");
  return rt_place_parts_strided(ctx, h, w, rows_per_tile, nparts, pad_rows * w, stacked_dev, image_dev);
}

extern "C" int rt_render_stats(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                               uint64_t stats3[3]) {
  RT_LOCK(ctx);
  if (!ctx || !ps || !stats3) return fail(ctx, "null argument");
  if (h <= 0 || w <= 0 || h > (1 << 20) || w > (1 << 20) || h * w > (int64_t(1) << 30)) return fail(ctx, "image size out of range");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  int32_t *tmp = nullptr;
  RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&tmp), sizeof(int32_t) * static_cast<size_t>(h) * w));
  hipError_t e = hipMemsetAsync(ctx->stats_dev, 0, 3 * sizeof(unsigned long long), ctx->stream);
  int rc = e == hipSuccess ? enqueue_render(ctx, ps, h, w, max_depth, 8, 0, 1, tmp, true) : 0;
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  unsigned long long host[3] = {0, 0, 0};
  if (!rc && e == hipSuccess) e = hipMemcpy(host, ctx->stats_dev, sizeof host, hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(ctx, e, "rt_render_stats");
  for (int i = 0; i < 3; ++i) stats3[i] = host[i];
  return 0;
}

// Diagnostic: one instrumented pooled launch that records, per wave, clock64 at start / at
// queue exhaustion / at exit, the number of BOX / LEAF / SHADE operations, the items they
// processed ((box << 32) | leaf) and the deepest bounce chain finished.  records: waves x 16 u64.
// The instrumented instantiation has neither the solo prologue nor the in-loop hand-over: for a view whose policy hands out
// single-pixel tickets (irreg 1000x1000) those tickets go through the pooled loop here, so the timeline overstates that
// view's longest chains by the difference between the two (~7 against ~4.5 us per bounce).
extern "C" int rt_render_trace(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                               uint64_t *records, int32_t max_waves, int32_t *num_waves) {
  RT_LOCK(ctx);
  if (!ctx || !ps || !records || !num_waves) return fail(ctx, "null argument");
  RT_LOCK_PS(ps);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  Plan pl{};
  const int saved = ctx->variant;
  ctx->variant = RT_VARIANT_POOLED;
  // the instrumented kernel exists for workgroups of 16 and of 8 waves: the production plan's shape if it is one of those
  int rc = make_plan(ctx, ps, &pl, ((w + 7) / 8) * ((h + 7) / 8));
  if (rc || pl.variant != RT_VARIANT_POOLED || (pl.waves != 16 && pl.waves != 8)) rc = make_plan(ctx, ps, &pl, ((w + 7) / 8) * ((h + 7) / 8), 8);
  ctx->variant = saved;
  if (rc) return rc;
  int nw = std::max(pl.grid, pl.grid_full) * pl.waves;   // (the buffers: the view's policy may ask for every workgroup, below)
  if (nw > max_waves) return fail(ctx, "trace buffer too small");
  if (h <= 0 || w <= 0 || h > (1 << 20) || w > (1 << 20) || h * w > (int64_t(1) << 30)) return fail(ctx, "image size out of range");
  int32_t *tmp = nullptr;
  unsigned long long *trace = nullptr;
  RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&tmp), sizeof(int32_t) * static_cast<size_t>(h) * w));
  if (hipError_t em = hipMalloc(reinterpret_cast<void **>(&trace), sizeof(unsigned long long) * rtk::kTraceWords * static_cast<size_t>(nw)); em != hipSuccess) {
    (void)hipFree(tmp);
    return hip_fail(ctx, em, "hipMalloc(trace)");
  }
  if (hipError_t em = hipMemsetAsync(trace, 0, sizeof(unsigned long long) * rtk::kTraceWords * static_cast<size_t>(nw), ctx->stream); em != hipSuccess) {
    (void)hipFree(tmp);
    (void)hipFree(trace);
    return hip_fail(ctx, em, "hipMemsetAsync(trace)");
  }
  rtk::KParams p{};
  p.nodes = ps->nodes; p.nodes64 = ps->nodes64; p.sph = ps->sph; p.col = ps->col;
  std::copy(ps->root_lo, ps->root_lo + 3, p.root_lo);
  std::copy(ps->root_hi, ps->root_hi + 3, p.root_hi);
  p.n_nodes = static_cast<int>(ps->n - 1); p.n_sph = static_cast<int>(ps->n);
  std::memcpy(&p.cam, &ps->cam, sizeof(p.cam));
  p.w = static_cast<int>(w); p.h = static_cast<int>(h);
  // (diagnostic knobs trace_part / trace_nparts: the timeline of one part of the cyclic row-tile partition -- a band alone on the chip)
  p.rows_per_tile = 8; p.part = ctx->trace_part; p.nparts = std::max(1, ctx->trace_nparts); p.rpt_log2 = 3;
  if (p.part < 0 || p.part >= p.nparts) p.part = 0;
  p.rows_local = static_cast<int>(rt::part_rows(h, 8, p.part, p.nparts));
  p.tiles_x = (p.w + 7) / 8;
  p.tiles_y = (p.rows_local + 7) / 8;
  p.max_depth = max_depth;
  p.out = tmp;
  p.nframes = 1;
  p.stats = ctx->stats_dev;
  p.trace = trace;
  p.queue = ctx->queue_dev;
  p.nchunks = p.tiles_x * ((p.rows_local + 7) / 8);
  p.tpt_log2 = ctx->tpt_log2 >= 0 ? ctx->tpt_log2 : 0;
  const int xq = ctx->xcd_queues < 0 ? 2 : ctx->xcd_queues;
  p.nshards = (xq && pl.grid % rtk::kMaxShards == 0) ? rtk::kMaxShards : 1;
  p.interleave = p.nshards > 1 && xq == 2;
  p.static_first = ctx->static_first;
  p.lds_nodes = pl.lds_nodes; p.lds_sph = pl.lds_sph;
  p.smax = pl.smax; p.lmax = pl.lmax;
  p.thr_shade = ctx->thr_shade; p.thr_leaf = ctx->thr_leaf;
  p.capb = pl.capb; p.capl = pl.capl; p.ray_planes = pl.ray_planes;
  p.prio_depth = ctx->prio_depth;
  p.box2 = ctx->box2;
  p.look_max = ctx->look_max > 0 ? ctx->look_max : (p.nchunks > 16384 ? 32 : 64);
  p.tl_log2 = ps->tl_depth;
  hipError_t e = hipSuccess;
  if (get_uv(ctx, w, h, &p.u_tab, &p.v_tab)) rc = 1;
  if (!rc) {
    // use the adaptive order of the matching view if one exists (a record nothing has sorted yet is sorted first: production would, ahead
    // of the view's next frame)
    for (auto &o : ps->orders)
      if (o.h == h && o.w == w && o.part == p.part && o.nparts == p.nparts && o.max_depth == max_depth && o.rows_per_tile == 8 &&
          std::memcmp(o.cam, &p.cam, sizeof o.cam) == 0 && ctx->adaptive_order && o.ntiles == p.nchunks && o.nshards == (p.interleave ? 1 : p.nshards)) {
        if (sort_view(ctx, ps, &o, p, pl) || await_view(ctx, &o)) rc = 1;
        if (!o.valid) continue;
        p.order = o.order;
        DeepPolicy dp;
        if (deep_policy(ctx, ps, &o, pl.grid_full * pl.waves, &dp, true)) rc = 1;
        p.deep_class = dp.deep_class;
        p.deep_split = dp.deep_split;
        p.deep_cap_log2 = dp.cap_log2;
        if (dp.sparse && ctx->grid_div == 0) pl.grid = pl.grid_full;   // as enqueue_render launches this view
        // ... through its pixel list when enqueue_render would (same conditions)
        if (o.px_valid && pl.waves == 16 && ctx->adaptive_order == 1 && (p.nshards == 1 || p.interleave) &&
            o.px_elems >= static_cast<size_t>(p.rows_local) * p.w && o.rows_per_tile == 8 &&
            (ctx->pixel_order == 2 || (ctx->pixel_order == 1 && ctx->deep_class < 0 && max_depth > 4 && p.nchunks <= ctx->px_max_tiles &&
                                       (p.nchunks >= 1024 || (pl.lds_nodes == p.n_nodes && pl.lds_sph == p.n_sph))))) {
          p.px_list = o.px_list;
          p.px_hdr = reinterpret_cast<const int *>(o.px_list + o.px_elems);
          p.px_hold = ctx->px_hold;
          p.px_prio = ctx->px_prio;
          // (the instrumented launch walks a list's one-pixel tickets in the pooled loop, as tickets of one held pixel -- the trace's item
          // counts are then the frame's complete work --, unless trace_solo asks for the production path: solo_trace with its own cycle
          // counters, words 13 .. 15 of a wave's record)
          p.solo = (ctx->trace_solo && o.px_solo && ctx->solo && ps->tl_depth == rtk::kTreeletDepth) ? 1 : 0;
          if (ctx->grid_div == 0 && pl.grid != pl.grid_full) {
            const int ns = (xq && pl.grid_full % rtk::kMaxShards == 0) ? rtk::kMaxShards : 1;
            const int il = ns > 1 && xq == 2;
            if (ns == 1 || il) { pl.grid = pl.grid_full; p.nshards = ns; p.interleave = il; }
          }
        }
      }
    nw = pl.grid * pl.waves;
    if (cull_allowed(ctx, ps, pl, p, nullptr, 1)) {   // as enqueue_render launches this view
      p.cull = 1;
      p.cull_c2 = ps->cull.c2;
      p.cull_kappa = ps->cull.kappa;
    }
    e = rtk::launch_pooled(p, true, pl.grid, pl.waves, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(records, trace, sizeof(unsigned long long) * rtk::kTraceWords * static_cast<size_t>(nw), hipMemcpyDeviceToHost);
  }
  (void)hipFree(tmp);
  (void)hipFree(trace);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(ctx, e, "rt_render_trace");
  *num_waves = nw;
  return 0;
}

extern "C" int rt_render_timed(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                               int32_t rows_per_tile, int32_t part, int32_t nparts, int32_t *out_dev, int32_t warmup,
                               int32_t iters, float *ms_out) {
  RT_LOCK(ctx);
  if (!ctx || !ms_out || iters <= 0 || warmup < 0) return fail(ctx, "bad argument");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < warmup; ++i)
    if (int rc = render_entry(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, out_dev, nullptr)) return rc;
  std::vector<hipEvent_t> ev(static_cast<size_t>(iters) + 1);
  for (auto &e : ev) RT_HIP(ctx, hipEventCreate(&e));
  RT_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  int rc = 0;
  for (int i = 0; i < iters && !rc; ++i) {
    rc = render_entry(ctx, ps, h, w, max_depth, rows_per_tile, part, nparts, out_dev, nullptr);
    if (!rc && hipEventRecord(ev[i + 1], ctx->stream) != hipSuccess) rc = fail(ctx, "hipEventRecord failed");
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (!rc && e == hipSuccess)
    for (int i = 0; i < iters; ++i) (void)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
  for (auto &x : ev) (void)hipEventDestroy(x);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(ctx, e, "rt_render_timed");
  return 0;
}

// ------------------------------------------------------------------------------------ buffers
extern "C" int rt_device_alloc(rt_context *ctx, void **out_dev, int64_t bytes) {
  RT_LOCK(ctx);
  if (!ctx || !out_dev || bytes < 0) return fail(ctx, "bad argument");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMalloc(out_dev, std::max<int64_t>(bytes, 16)));
  return 0;
}
extern "C" int rt_device_free(rt_context *ctx, void *dev) {
  RT_LOCK(ctx);
  if (!ctx) return 1;
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipFree(dev));
  return 0;
}
extern "C" int rt_copy_to_host(rt_context *ctx, void *dst_host, const void *src_dev, int64_t bytes) {
  RT_LOCK(ctx);
  if (!ctx || !dst_host || !src_dev || bytes < 0) return fail(ctx, "bad argument");
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, static_cast<size_t>(bytes), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->synced_since_render = true;
  return 0;
}

// ====================================================================================
// The Futhark-shaped boundary (include/ray.h) -- thin wrappers over the rt_* surface.
// ====================================================================================
struct futhark_context_config {
  int device = -1;
  std::vector<int> devices;   // more than one entry: a multi-device context (set_device("0-7") / "0,2,4"; env RT_DEVICES)
  int debugging = 0, logging = 0, profiling = 0;
};
struct futhark_context {
  std::recursive_mutex mu;   // Futhark's context lock: every futhark_* entry holds it (the image pool, the pending error, the counters)
  rt_context *rt = nullptr;
  // freed images are kept for reuse: main.c frees and re-renders every run, and a
  // hipFree/hipMalloc pair per frame costs more than the frame itself
  struct PooledImage { int64_t elems; int32_t *dev; size_t block_bytes; };
  std::vector<PooledImage> image_pool;
  std::string pending;   // error not yet collected by futhark_context_get_error
  int logging = 0;
  uint64_t renders = 0, prepares = 0;
};
struct futhark_opaque_scene {
  rt_scene *s = nullptr;
};
struct futhark_opaque_prepared_scene {
  rt_prepared *p = nullptr;
};
struct futhark_i32_2d {
  int32_t *dev = nullptr;
  int64_t shape[2] = {0, 0};
  size_t block_bytes = 0;   // != 0: a block of the context's arena / block pool (image_alloc); 0: rt_device_alloc's
};
namespace {
// An entry point's result array from the context's arena (pool_alloc): a hipMalloc of 4 MB inside the harness's first render call was ~0.1 ms
// of its first frame.  Freed (image_release) behind a drained stream, like rt_device_free.
int image_alloc(rt_context *ctx, void **dev, size_t *block_bytes, int64_t bytes) {
  RT_LOCK(ctx);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  char *blk = nullptr;
  *block_bytes = static_cast<size_t>(std::max<int64_t>(bytes, 16));
  RT_HIP(ctx, pool_alloc(ctx, &blk, block_bytes));
  *dev = blk;
  return 0;
}
int image_release(rt_context *ctx, void *dev, size_t block_bytes) {
  if (!block_bytes) return rt_device_free(ctx, dev);
  RT_LOCK(ctx);
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pool_free(ctx, static_cast<char *>(dev), block_bytes);
  return 0;
}
}  // namespace

#ifndef RT_NO_CONTEXT_LOCK
#define FUT_LOCK(ctx) std::unique_lock<std::recursive_mutex> fut_lock_ = (ctx) ? std::unique_lock<std::recursive_mutex>((ctx)->mu) : std::unique_lock<std::recursive_mutex>()
#else
#define FUT_LOCK(ctx) (void)(ctx)
#endif

namespace {
int fut_fail(futhark_context *ctx, int rc) {
  if (rc && ctx) ctx->pending = rt_last_error(ctx->rt);
  return rc;
}
}  // namespace

extern "C" struct futhark_context_config *futhark_context_config_new(void) { return new futhark_context_config(); }
extern "C" void futhark_context_config_free(struct futhark_context_config *cfg) { delete cfg; }
extern "C" void futhark_context_config_set_debugging(struct futhark_context_config *cfg, int flag) { if (cfg) cfg->debugging = flag; }
extern "C" void futhark_context_config_set_logging(struct futhark_context_config *cfg, int flag) { if (cfg) cfg->logging = flag; }
extern "C" void futhark_context_config_set_profiling(struct futhark_context_config *cfg, int flag) { if (cfg) cfg->profiling = flag; }
// "k" / "#k": HIP device k (the Futhark convention); "a-b" or "a,b,c": several devices = one multi-device
// context (the frame is partitioned over them, include/rt_mi355x.h: rt_context_create_multi)
static std::vector<int> parse_device_list(const char *s) {
  std::vector<int> out;
  while (*s) {
    while (*s == ' ' || *s == ',' || *s == '#') ++s;
    if (!*s) break;
    char *end = nullptr;
    const long a = std::strtol(s, &end, 10);
    if (end == s || a < 0) return {};
    s = end;
    long b = a;
    if (*s == '-') {
      b = std::strtol(s + 1, &end, 10);
      if (end == s + 1 || b < a) return {};
      s = end;
    }
    for (long d = a; d <= b && out.size() < 64; ++d) out.push_back(static_cast<int>(d));
  }
  return out;
}
extern "C" void futhark_context_config_set_device(struct futhark_context_config *cfg, const char *s) {
  if (!cfg || !s) return;
  cfg->devices = parse_device_list(s);
  cfg->device = cfg->devices.empty() ? std::atoi(*s == '#' ? s + 1 : s) : cfg->devices[0];
}

extern "C" void futhark_context_config_set_num_threads(struct futhark_context_config *, int) {}
extern "C" void futhark_context_config_set_cache_file(struct futhark_context_config *, const char *) {}

extern "C" struct futhark_context *futhark_context_new(struct futhark_context_config *cfg) {
  auto ctx = std::make_unique<futhark_context>();
  ctx->logging = cfg ? (cfg->logging | cfg->debugging) : 0;
  // RT_DEVICES lets a harness that never calls set_device (futhark/main.c does not) use several GPUs
  // (an explicit futhark_context_config_set_device wins over the environment)
  std::vector<int> devs = cfg ? cfg->devices : std::vector<int>();
  if (const char *env = std::getenv("RT_DEVICES"); env && devs.empty() && !(cfg && cfg->device >= 0)) devs = parse_device_list(env);
  const int rc = devs.size() > 1 ? rt_context_create_multi(&ctx->rt, devs.data(), static_cast<int>(devs.size()))
                                 : rt_context_create(&ctx->rt, !devs.empty() ? devs[0] : (cfg ? cfg->device : -1), nullptr, 0);
  if (rc) {
    // Futhark returns a context whose error is set; main.c asserts it is NULL.
    ctx->pending = "libray_mi355x: cannot create a HIP context (code " + std::to_string(rc) + "); no CPU path exists";
  }
  return ctx.release();
}
extern "C" void futhark_context_free(struct futhark_context *ctx) {
  if (!ctx) return;
  for (auto &e : ctx->image_pool) (void)image_release(ctx->rt, e.dev, e.block_bytes);
  rt_context_destroy(ctx->rt);
  delete ctx;
}
extern "C" char *futhark_context_get_error(struct futhark_context *ctx) {
  FUT_LOCK(ctx);
  if (!ctx || ctx->pending.empty()) return nullptr;
  char *s = static_cast<char *>(std::malloc(ctx->pending.size() + 1));
  if (s) std::memcpy(s, ctx->pending.c_str(), ctx->pending.size() + 1);
  ctx->pending.clear();
  return s;
}
extern "C" int futhark_context_sync(struct futhark_context *ctx) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt) return 1;
  return fut_fail(ctx, rt_context_sync(ctx->rt));
}
extern "C" char *futhark_context_report(struct futhark_context *ctx) {
  FUT_LOCK(ctx);
  char buf[512];
  int dev = -1, cus = 0, lds = 0;
  char name[64] = "";
  if (ctx && ctx->rt) rt_context_device_info(ctx->rt, &dev, &cus, &lds, name, sizeof name);
  std::snprintf(buf, sizeof buf, "libray_mi355x: device %d (%s), %d CUs, %d B LDS/CU; %d device(s), framebuffer gather: %s; %llu prepare_scene, %llu render calls\n",
                dev, name, cus, lds, ctx && ctx->rt ? rt_context_num_devices(ctx->rt) : 0,
                ctx && ctx->rt ? rt_context_gather_mode(ctx->rt) : "none", ctx ? (unsigned long long)ctx->prepares : 0ull,
                ctx ? (unsigned long long)ctx->renders : 0ull);
  char *s = static_cast<char *>(std::malloc(std::strlen(buf) + 1));
  if (s) std::strcpy(s, buf);
  return s;
}

extern "C" int futhark_entry_rgbbox(struct futhark_context *ctx, struct futhark_opaque_scene **out0) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt || !out0) return 1;
  auto o = std::make_unique<futhark_opaque_scene>();
  if (int rc = rt_scene_rgbbox(ctx->rt, &o->s)) return fut_fail(ctx, rc);
  *out0 = o.release();
  return 0;
}
extern "C" int futhark_entry_irreg(struct futhark_context *ctx, struct futhark_opaque_scene **out0) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt || !out0) return 1;
  auto o = std::make_unique<futhark_opaque_scene>();
  if (int rc = rt_scene_irreg(ctx->rt, &o->s)) return fut_fail(ctx, rc);
  *out0 = o.release();
  return 0;
}
extern "C" int futhark_entry_prepare_scene(struct futhark_context *ctx, struct futhark_opaque_prepared_scene **out0,
                                           const int64_t in0, const int64_t in1, const struct futhark_opaque_scene *in2) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt || !out0 || !in2) return 1;
  auto o = std::make_unique<futhark_opaque_prepared_scene>();
  if (int rc = rt_prepare_scene(ctx->rt, &o->p, in0, in1, in2->s)) return fut_fail(ctx, rc);
  ctx->prepares++;
  *out0 = o.release();
  return 0;
}
extern "C" int futhark_entry_render(struct futhark_context *ctx, struct futhark_i32_2d **out0, const int64_t in0,
                                    const int64_t in1, const struct futhark_opaque_prepared_scene *in2) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt || !out0 || !in2) return 1;
  auto img = std::make_unique<futhark_i32_2d>();
  void *dev = nullptr;
  for (size_t i = 0; i < ctx->image_pool.size(); ++i)
    if (ctx->image_pool[i].elems == in0 * in1) {
      dev = ctx->image_pool[i].dev;   // same stream => the new frame orders after any pending use
      img->block_bytes = ctx->image_pool[i].block_bytes;
      ctx->image_pool.erase(ctx->image_pool.begin() + static_cast<long>(i));
      break;
    }
  if (!dev)
    if (int rc = image_alloc(ctx->rt, &dev, &img->block_bytes, static_cast<int64_t>(sizeof(int32_t)) * in0 * in1)) return fut_fail(ctx, rc);
  img->dev = static_cast<int32_t *>(dev);
  img->shape[0] = in0;
  img->shape[1] = in1;
  if (int rc = rt_render(ctx->rt, in2->p, in0, in1, img->dev)) {
    (void)image_release(ctx->rt, dev, img->block_bytes);
    return fut_fail(ctx, rc);
  }
  ctx->renders++;
  *out0 = img.release();
  return 0;
}
extern "C" int futhark_values_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr, int32_t *data) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt || !arr || !data) return 1;
  return fut_fail(ctx, rt_copy_to_host(ctx->rt, data, arr->dev, static_cast<int64_t>(sizeof(int32_t)) * arr->shape[0] * arr->shape[1]));
}
extern "C" int futhark_free_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr) {
  FUT_LOCK(ctx);
  if (!arr) return 0;
  int rc = 0;
  if (ctx && ctx->rt && arr->dev) {
    if (ctx->image_pool.size() < 4) ctx->image_pool.push_back({arr->shape[0] * arr->shape[1], arr->dev, arr->block_bytes});
    else rc = image_release(ctx->rt, arr->dev, arr->block_bytes);
  }
  delete arr;
  return fut_fail(ctx, rc);
}
extern "C" struct futhark_i32_2d *futhark_new_i32_2d(struct futhark_context *ctx, const int32_t *data, int64_t dim0, int64_t dim1) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt || !data || dim0 < 0 || dim1 < 0) return nullptr;
  auto arr = std::make_unique<futhark_i32_2d>();
  arr->shape[0] = dim0; arr->shape[1] = dim1;
  const int64_t bytes = static_cast<int64_t>(sizeof(int32_t)) * dim0 * dim1;
  if (rt_device_alloc(ctx->rt, reinterpret_cast<void **>(&arr->dev), std::max<int64_t>(bytes, 4)) != 0) { fut_fail(ctx, 1); return nullptr; }
  if (bytes > 0 && hipMemcpy(arr->dev, data, static_cast<size_t>(bytes), hipMemcpyHostToDevice) != hipSuccess) {
    (void)rt_device_free(ctx->rt, arr->dev);
    ctx->pending = "futhark_new_i32_2d: host to device copy failed";
    return nullptr;
  }
  return arr.release();
}
extern "C" int32_t *futhark_values_raw_i32_2d(struct futhark_context *, struct futhark_i32_2d *arr) { return arr ? arr->dev : nullptr; }
extern "C" int futhark_context_clear_caches(struct futhark_context *ctx) {
  FUT_LOCK(ctx);
  if (!ctx || !ctx->rt) return 1;
  if (rt_context_sync(ctx->rt)) return fut_fail(ctx, 1);
  for (auto &im : ctx->image_pool) (void)image_release(ctx->rt, im.dev, im.block_bytes);
  ctx->image_pool.clear();
  for (auto &b : ctx->rt->pool) (void)hipFree(b.p);
  ctx->rt->pool.clear();
  return 0;
}
extern "C" void futhark_context_pause_profiling(struct futhark_context *) {}
extern "C" void futhark_context_unpause_profiling(struct futhark_context *) {}

extern "C" const int64_t *futhark_shape_i32_2d(struct futhark_context *, struct futhark_i32_2d *arr) {
  return arr ? arr->shape : nullptr;
}
extern "C" int futhark_free_opaque_prepared_scene(struct futhark_context *ctx, struct futhark_opaque_prepared_scene *obj) {
  FUT_LOCK(ctx);
  if (!obj) return 0;
  if (ctx && ctx->rt) rt_prepared_free(ctx->rt, obj->p);
  delete obj;
  return 0;
}
extern "C" int futhark_free_opaque_scene(struct futhark_context *ctx, struct futhark_opaque_scene *obj) {
  FUT_LOCK(ctx);
  if (!obj) return 0;
  rt_scene_free(ctx ? ctx->rt : nullptr, obj->s);
  delete obj;
  return 0;
}
