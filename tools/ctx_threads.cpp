// ctx_threads.cpp -- several host threads sharing ONE context and ONE prepared scene (SURVEY.md 8b "Async / threading": a Futhark
// context serialises concurrent calls with an internal lock; so does this library -- rt_internal.hpp).  Not a product path.
//
// Mode `rt`: thread k renders view k of the one prepared scene (the prepared camera moved along x by k * 0.37: each thread its own
// view, all views' tile orders / pixel lists living in the ONE rt_prepared) `frames` times through rt_render_image + rt_context_sync +
// rt_copy_to_host and checks every frame against its own first one; mode `futhark`: the same through futhark_entry_render /
// futhark_context_sync / futhark_values_i32_2d / futhark_free_i32_2d on one futhark_context with one prepared scene per thread
// (the reference's API has no camera argument).  Prints one checksum per thread (c = c * 31 + pixel); the test compares them with the
// CPU checker's.  Built twice: plainly (build/ctx_threads) and with -fsanitize=thread over the library's host code (build/tsan/).
//   ctx_threads <rt|futhark> <rgbbox|irreg> <size> <frames> <threads>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ray.h"
#include "rt_mi355x.h"

static uint32_t checksum(const std::vector<int32_t> &px) {
  uint32_t c = 0;
  for (int32_t p : px) c = c * 31u + (uint32_t)p;
  return c;
}

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "rt", scene = argc > 2 ? argv[2] : "irreg";
  const int n = argc > 3 ? atoi(argv[3]) : 256, frames = argc > 4 ? atoi(argv[4]) : 200, nthreads = argc > 5 ? atoi(argv[5]) : 2;
  std::atomic<int> bad{0};
  std::vector<uint32_t> sums((size_t)nthreads, 0u);
  if (mode == "rt") {
    rt_context *ctx = nullptr;
    if (rt_context_create(&ctx, -1, nullptr, 0)) { fprintf(stderr, "no context\n"); return 2; }
    rt_scene *sc = nullptr;
    if (scene == "rgbbox" ? rt_scene_rgbbox(ctx, &sc) : rt_scene_irreg(ctx, &sc)) return 2;
    rt_prepared *ps = nullptr;
    if (rt_prepare_scene(ctx, &ps, n, n, sc)) { fprintf(stderr, "%s\n", rt_last_error(ctx)); return 2; }
    float cam0[12];
    rt_prepared_get_camera(ctx, ps, cam0);
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; ++k)
      th.emplace_back([&, k] {
        float cam[12];
        memcpy(cam, cam0, sizeof cam);
        cam[0] += 0.37f * (float)k;
        void *dev = nullptr;
        if (rt_device_alloc(ctx, &dev, (int64_t)n * n * 4)) { bad++; return; }
        std::vector<int32_t> first, px((size_t)n * n);
        for (int f = 0; f < frames; ++f) {
          if (rt_render_image(ctx, ps, n, n, cam, 50, 8, 0, 1, (int32_t *)dev) || rt_context_sync(ctx) ||
              rt_copy_to_host(ctx, px.data(), dev, (int64_t)n * n * 4)) { bad++; break; }
          if (f == 0) first = px;
          else if (px != first) { bad++; fprintf(stderr, "thread %d frame %d differs from its first frame\n", k, f); break; }
          if (f % 16 == 7) rt_context_set_option(ctx, "thr_shade", 40);   // (an option write in between: the lock covers it too)
        }
        sums[(size_t)k] = checksum(px);
        rt_device_free(ctx, dev);
      });
    for (auto &t : th) t.join();
    rt_prepared_free(ctx, ps);
    rt_scene_free(ctx, sc);
    rt_context_destroy(ctx);
  } else {
    futhark_context_config *cfg = futhark_context_config_new();
    futhark_context *ctx = futhark_context_new(cfg);
    if (char *e = futhark_context_get_error(ctx)) { fprintf(stderr, "%s\n", e); return 2; }
    futhark_opaque_scene *sc = nullptr;
    if (scene == "rgbbox" ? futhark_entry_rgbbox(ctx, &sc) : futhark_entry_irreg(ctx, &sc)) return 2;
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; ++k)
      th.emplace_back([&, k] {
        const int64_t h = n + 8 * k, w = n;                 // thread k's own prepared scene and image size
        futhark_opaque_prepared_scene *ps = nullptr;
        if (futhark_entry_prepare_scene(ctx, &ps, h, w, sc) || futhark_context_sync(ctx)) { bad++; return; }
        std::vector<int32_t> first, px((size_t)(h * w));
        futhark_i32_2d *img = nullptr;
        for (int f = 0; f < frames; ++f) {
          if (img) futhark_free_i32_2d(ctx, img);
          img = nullptr;
          if (futhark_entry_render(ctx, &img, h, w, ps) || futhark_context_sync(ctx) || futhark_values_i32_2d(ctx, img, px.data())) { bad++; break; }
          if (f == 0) first = px;
          else if (px != first) { bad++; fprintf(stderr, "thread %d frame %d differs from its first frame\n", k, f); break; }
        }
        sums[(size_t)k] = checksum(px);
        if (img) futhark_free_i32_2d(ctx, img);
        futhark_free_opaque_prepared_scene(ctx, ps);
      });
    for (auto &t : th) t.join();
    futhark_free_opaque_scene(ctx, sc);
    futhark_context_free(ctx);
    futhark_context_config_free(cfg);
  }
  for (int k = 0; k < nthreads; ++k) printf("thread %d checksum %08x\n", k, sums[(size_t)k]);
  printf("%s: %d threads x %d frames on one context: %s\n", mode.c_str(), nthreads, frames, bad ? "FAILED" : "every frame equals its view's first");
  return bad ? 1 : 0;
}
