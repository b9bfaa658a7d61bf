// multi_gpu.cpp -- one process, several MI355X: the row-tile partition of SURVEY.md 8(e) behind the
// C ABI (rt_context_create_multi / futhark_context_config_set_device("0-7")), so that the reference's
// own harness (futhark/main.c:59-64 context, :107-124 render + sync loop, :126-135 values) drives N GPUs
// without a line changed.
//
// A multi-device context is a PARENT context on the first device plus one child context (own
// stream) per device entry.  prepare_scene replicates the scene: every device builds its own BVH from
// the same spheres (bit-identical by construction; <= 66 MB even for the 10^6-sphere scene).  render
// fans out rt_render_part over the children -- part i of N = the tiles t of 8 rows with t % N == i,
// cyclic because contiguous bands would give irreg's devices 0.1 % .. 25 % of the work each -- and
// gathers the packed parts on the first device, where one kernel (place_all_kernel) scatters them
// into the [h][w]i32 image:
//   * RCCL (default when the devices are distinct and librccl loads): one ncclCommInitAll communicator
//     per device, grouped ncclSend (child stream, right behind the render) / ncclRecv (parent stream)
//     straight into the part's slice of the stacked buffer -- point-to-point over xGMI, 7 links into
//     device 0 in parallel; nothing is reduced, so no ring collective is involved;
//   * peer copies (hipMemcpyPeerAsync on the child's stream; also the fallback, and the only mode
//     for a device list with repeats, which exists to test the fan-out on a one-GPU box).
// librccl is loaded on demand (dlopen): the single-device library has no RCCL dependency.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <future>
#include <memory>

#include "rt_internal.hpp"

struct rt_group {
  std::vector<rt_context *> kids;   // one per device entry, own stream each
  std::vector<int> devices;
  bool distinct = true;
  int gather = 0;                   // 0 auto, 1 peer copies, 2 RCCL (then part 0 goes through RCCL too)
  // gather buffers, grown on demand
  int64_t buf_elems = 0;            // capacity of one part in int32
  std::vector<int32_t *> part;      // part[i] on device i (kid 0 renders into the stacked buffer unless gather == 2)
  int32_t *stacked = nullptr;       // first device: kids x buf_elems
  std::vector<hipEvent_t> ev_part;  // kid i's render (+ peer copy) of the current frame
  hipEvent_t ev_placed = nullptr;   // the parent's assembly of the current frame
  bool placed_valid = false;
  // RCCL, loaded on demand
  void *rccl = nullptr;
  bool rccl_tried = false;
  std::vector<ncclComm_t> comms;
  decltype(&ncclCommInitAll) p_init = nullptr;
  decltype(&ncclCommDestroy) p_destroy = nullptr;
  decltype(&ncclGroupStart) p_gstart = nullptr;
  decltype(&ncclGroupEnd) p_gend = nullptr;
  decltype(&ncclSend) p_send = nullptr;
  decltype(&ncclRecv) p_recv = nullptr;
  decltype(&ncclGetErrorString) p_errstr = nullptr;
  std::string rccl_note;            // why RCCL is not in use (rt_context_report)
};

namespace {

using rti::fail;

bool load_rccl(rt_group *g) {
  if (g->rccl_tried) return !g->comms.empty();
  g->rccl_tried = true;
  if (!g->distinct) { g->rccl_note = "device list has repeats (test mode): peer copies"; return false; }
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    g->rccl = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (g->rccl) break;
  }
  if (!g->rccl) { g->rccl_note = std::string("librccl not loadable: ") + dlerror(); return false; }
#define RT_SYM(field, sym)                                                         \
  g->field = reinterpret_cast<decltype(g->field)>(dlsym(g->rccl, #sym));          \
  if (!g->field) { g->rccl_note = "librccl lacks " #sym; return false; }
  RT_SYM(p_init, ncclCommInitAll) RT_SYM(p_destroy, ncclCommDestroy) RT_SYM(p_gstart, ncclGroupStart)
  RT_SYM(p_gend, ncclGroupEnd) RT_SYM(p_send, ncclSend) RT_SYM(p_recv, ncclRecv) RT_SYM(p_errstr, ncclGetErrorString)
#undef RT_SYM
  g->comms.assign(g->devices.size(), nullptr);
  const ncclResult_t r = g->p_init(g->comms.data(), static_cast<int>(g->devices.size()), g->devices.data());
  if (r != ncclSuccess) {
    g->rccl_note = std::string("ncclCommInitAll failed: ") + g->p_errstr(r);
    g->comms.clear();
    return false;
  }
  return true;
}

int ensure_buffers(rt_context *ctx, int64_t elems) {
  rt_group *g = ctx->group;
  if (elems <= g->buf_elems) return 0;
  if (int rc = rti::group_sync(ctx)) return rc;
  const size_t n = g->kids.size();
  for (size_t i = 0; i < n; ++i)
    if (g->part[i]) {
      (void)hipSetDevice(g->devices[i]);
      (void)hipFree(g->part[i]);
      g->part[i] = nullptr;
    }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (g->stacked) (void)hipFree(g->stacked);
  g->stacked = nullptr;
  g->buf_elems = 0;
  RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&g->stacked), sizeof(int32_t) * static_cast<size_t>(elems) * n));
  for (size_t i = 0; i < n; ++i) {
    RT_HIP(ctx, hipSetDevice(g->devices[i]));
    RT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&g->part[i]), sizeof(int32_t) * static_cast<size_t>(elems)));
  }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  g->buf_elems = elems;
  g->placed_valid = false;
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int rt_context_create_multi(rt_context **out, const int *devices, int ndev) {
  if (!out || !devices || ndev < 1 || ndev > 64) return 1;
  *out = nullptr;
  rt_context *parent = nullptr;
  if (int rc = rt_context_create(&parent, devices[0], nullptr, 0)) return rc;
  rt_group *g = new rt_group;
  parent->group = g;   // owned by the parent from here on: rt_context_destroy releases whatever exists so far
  g->devices.assign(devices, devices + ndev);
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (devices[i] == devices[j]) g->distinct = false;
  auto bail = [&](int code, const std::string &why) {
    std::fprintf(stderr, "libray_mi355x: rt_context_create_multi: %s\n", why.c_str());
    rt_context_destroy(parent);
    return code;
  };
  g->part.assign(static_cast<size_t>(ndev), nullptr);
  g->ev_part.assign(static_cast<size_t>(ndev), nullptr);
  for (int i = 0; i < ndev; ++i) {
    rt_context *kid = nullptr;
    if (int rc = rt_context_create(&kid, devices[i], nullptr, 0)) return bail(rc, "cannot create a context on device " + std::to_string(devices[i]));
    g->kids.push_back(kid);
    if (hipSetDevice(devices[i]) != hipSuccess || hipEventCreateWithFlags(&g->ev_part[static_cast<size_t>(i)], hipEventDisableTiming) != hipSuccess)
      return bail(8, "hipEventCreate failed");
    if (devices[i] != devices[0]) {
      // direct xGMI access both ways (an already enabled pair reports an error that is not one)
      (void)hipDeviceEnablePeerAccess(devices[0], 0);
      (void)hipSetDevice(devices[0]);
      (void)hipDeviceEnablePeerAccess(devices[i], 0);
      (void)hipGetLastError();
    }
  }
  if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&g->ev_placed, hipEventDisableTiming) != hipSuccess)
    return bail(8, "hipEventCreate failed");
  *out = parent;
  return 0;
}

extern "C" int rt_context_num_devices(const rt_context *ctx) {
  if (!ctx) return 0;
  return ctx->group ? static_cast<int>(ctx->group->kids.size()) : 1;
}

extern "C" const char *rt_context_gather_mode(rt_context *ctx) {
  if (!ctx || !ctx->group || ctx->group->kids.size() < 2) return ctx && ctx->group && ctx->group->gather == 2 ? "rccl" : "none";
  rt_group *g = ctx->group;
  if (g->gather == 1) return "peer-copy";
  return load_rccl(g) ? "rccl" : "peer-copy";
}

void rti::group_destroy(rt_context *ctx) {
  rt_group *g = ctx ? ctx->group : nullptr;
  if (!g) return;
  (void)rti::group_sync(ctx);
  for (ncclComm_t c : g->comms)
    if (c && g->p_destroy) (void)g->p_destroy(c);
  for (size_t i = 0; i < g->devices.size(); ++i) {
    (void)hipSetDevice(g->devices[i]);
    if (i < g->part.size() && g->part[i]) (void)hipFree(g->part[i]);
    if (i < g->ev_part.size() && g->ev_part[i]) (void)hipEventDestroy(g->ev_part[i]);
    if (i < g->kids.size()) rt_context_destroy(g->kids[i]);
  }
  (void)hipSetDevice(ctx->device);
  if (g->stacked) (void)hipFree(g->stacked);
  if (g->ev_placed) (void)hipEventDestroy(g->ev_placed);
  // librccl stays loaded: unloading a library that owns device state at exit time is asking for trouble
  delete g;
  ctx->group = nullptr;
}

int rti::group_sync(rt_context *ctx) {
  rt_group *g = ctx->group;
  for (rt_context *kid : g->kids) {
    RT_HIP(ctx, hipSetDevice(kid->device));
    RT_HIP(ctx, hipStreamSynchronize(kid->stream));
  }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int rti::group_set_variant(rt_context *ctx, int variant) {
  for (rt_context *kid : ctx->group->kids)
    if (rt_context_set_variant(kid, variant)) return fail(ctx, rt_last_error(kid));
  return 0;
}

int rti::group_set_option(rt_context *ctx, const char *name, int64_t value) {
  if (std::strcmp(name, "gather") == 0) {
    if (value < 0 || value > 2) return fail(ctx, "gather must be 0 (auto), 1 (peer copies) or 2 (RCCL)");
    if (int rc = rti::group_sync(ctx)) return rc;
    ctx->group->gather = static_cast<int>(value);
    return 0;
  }
  for (rt_context *kid : ctx->group->kids)
    if (rt_context_set_option(kid, name, value)) return fail(ctx, rt_last_error(kid));
  return -1;   // not a group-only option: the caller applies it to the parent as well
}

// prepare_scene on every device: the parent's own prepared scene (first device) serves child 0,
// children 1.. build replicas concurrently (one host thread per device: the build ends in a
// stream synchronise, the reference's harness times this call).
int rti::group_prepare(rt_context *ctx, rt_prepared *ps, int64_t h, int64_t w, const rt_scene *scene) {
  rt_group *g = ctx->group;
  const size_t n = g->kids.size();
  ps->replicas.assign(n, nullptr);
  std::vector<std::future<int>> jobs;
  for (size_t i = 1; i < n; ++i)
    jobs.push_back(std::async(std::launch::async, [=] { return rt_prepare_scene(g->kids[i], &ps->replicas[i], h, w, scene); }));
  int rc = 0;
  for (size_t i = 1; i < n; ++i)
    if (jobs[i - 1].get() != 0 && !rc) rc = fail(ctx, std::string("device ") + std::to_string(g->devices[i]) + ": " + rt_last_error(g->kids[i]));
  (void)hipSetDevice(ctx->device);
  return rc;
}

void rti::group_prepared_free(rt_context *ctx, rt_prepared *ps) {
  rt_group *g = ctx ? ctx->group : nullptr;
  if (g) (void)rti::group_sync(ctx);
  for (size_t i = 0; i < ps->replicas.size(); ++i)
    if (ps->replicas[i]) rt_prepared_free(g && i < g->kids.size() ? g->kids[i] : nullptr, ps->replicas[i]);
  ps->replicas.clear();
  if (ctx) (void)hipSetDevice(ctx->device);
}

int rti::group_render(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t *out_dev,
                      const float *cam12) {
  rt_group *g = ctx->group;
  const int n = static_cast<int>(g->kids.size());
  if (ps->replicas.size() != static_cast<size_t>(n)) return fail(ctx, "prepared scene does not belong to this multi-device context");
  if (!out_dev) return fail(ctx, "null output pointer");
  if (h <= 0 || w <= 0 || h * w > (int64_t(1) << 30)) return fail(ctx, "image size out of range");
  constexpr int32_t kRows = 8;
  const bool want_rccl = g->gather == 2 || (g->gather == 0 && n > 1);
  const bool rccl = want_rccl && load_rccl(g);
  if (g->gather == 2 && !rccl) return fail(ctx, "gather=2 (RCCL) requested but unavailable: " + g->rccl_note);
  if (n == 1 && !rccl) {
    if (rti::enqueue_render(g->kids[0], ps, h, w, max_depth, kRows, 0, 1, out_dev, false, cam12)) return fail(ctx, rt_last_error(g->kids[0]));
    // the frame lives on child 0's stream: order the parent's stream (values, frees) behind it
    RT_HIP(ctx, hipEventRecord(g->ev_part[0], g->kids[0]->stream));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, g->ev_part[0], 0));
    return 0;
  }
  int64_t pad_rows = 0;
  for (int p = 0; p < n; ++p) pad_rows = std::max<int64_t>(pad_rows, rt::part_rows(h, kRows, p, n));
  const int64_t stride = pad_rows * w;
  if (int rc = ensure_buffers(ctx, stride)) return rc;
  const bool all_rccl = rccl && g->gather == 2;   // forced: part 0 travels through RCCL as well (self send/recv)
  for (int i = 0; i < n; ++i) {
    rt_context *kid = g->kids[static_cast<size_t>(i)];
    RT_HIP(ctx, hipSetDevice(kid->device));
    // the previous frame's assembly must have read the stacked buffer before this frame overwrites it
    if (g->placed_valid) RT_HIP(ctx, hipStreamWaitEvent(kid->stream, g->ev_placed, 0));
    const bool direct = i == 0 && !all_rccl;
    int32_t *dst = direct ? g->stacked : g->part[static_cast<size_t>(i)];
    const rt_prepared *kps = i == 0 ? ps : ps->replicas[static_cast<size_t>(i)];
    if (rti::enqueue_render(kid, kps, h, w, max_depth, kRows, i, n, dst, false, cam12)) return fail(ctx, rt_last_error(kid));
    const int64_t rows = rt::part_rows(h, kRows, i, n);
    if (!direct && !rccl && rows > 0)
      RT_HIP(ctx, hipMemcpyPeerAsync(g->stacked + static_cast<int64_t>(i) * stride, g->devices[0], dst, kid->device,
                                     sizeof(int32_t) * static_cast<size_t>(rows * w), kid->stream));
    RT_HIP(ctx, hipEventRecord(g->ev_part[static_cast<size_t>(i)], kid->stream));
  }
  RT_HIP(ctx, hipSetDevice(ctx->device));
  if (rccl) {
    // part 0 (or, forced, its send) must be complete on the parent's stream's view
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, g->ev_part[0], 0));
    ncclResult_t r = g->p_gstart();
    for (int i = all_rccl ? 0 : 1; i < n && r == ncclSuccess; ++i) {
      const size_t count = static_cast<size_t>(rt::part_rows(h, kRows, i, n) * w);
      if (count == 0) continue;
      // comm[0]'s operations all sit on the parent's stream; a child's send follows its render on its own stream
      hipStream_t sstream = i == 0 ? ctx->stream : g->kids[static_cast<size_t>(i)]->stream;
      r = g->p_send(g->part[static_cast<size_t>(i)], count, ncclInt32, 0, g->comms[static_cast<size_t>(i)], sstream);
      if (r == ncclSuccess) r = g->p_recv(g->stacked + static_cast<int64_t>(i) * stride, count, ncclInt32, i, g->comms[0], ctx->stream);
    }
    const ncclResult_t e = g->p_gend();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return fail(ctx, std::string("RCCL gather failed: ") + g->p_errstr(r));
  } else {
    for (int i = 0; i < n; ++i) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, g->ev_part[static_cast<size_t>(i)], 0));
  }
  RT_HIP(ctx, rtk::launch_place_all(g->stacked, out_dev, static_cast<int>(w), static_cast<int>(h), kRows, n,
                                    static_cast<size_t>(stride), ctx->stream));
  RT_HIP(ctx, hipEventRecord(g->ev_placed, ctx->stream));
  g->placed_valid = true;
  return 0;
}
