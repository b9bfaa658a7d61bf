// first_call_probe.c -- what the reference's harness pays per call, call by call (futhark/main.c times the MEAN of its runs, first call included):
// wall clock of each prepare_scene and of each render + sync through the Futhark-shaped ABI.   first_call_probe <rgbbox|irreg> <size> [calls = 6]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "ray.h"
static double now(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; }
int main(int argc, char **argv) {
  const char *scene_name = argc > 1 ? argv[1] : "rgbbox";
  const int n = argc > 2 ? atoi(argv[2]) : 1000, calls = argc > 3 ? atoi(argv[3]) : 6;
  double t0 = now();
  struct futhark_context_config *cfg = futhark_context_config_new();
  struct futhark_context *ctx = futhark_context_new(cfg);
  printf("context: %.3f ms\n", now() - t0);
  struct futhark_opaque_scene *scene = NULL;
  if (strcmp(scene_name, "irreg") == 0) futhark_entry_irreg(ctx, &scene); else futhark_entry_rgbbox(ctx, &scene);
  struct futhark_opaque_prepared_scene *ps = NULL;
  for (int i = 0; i < 3; ++i) {
    if (ps) futhark_free_opaque_prepared_scene(ctx, ps);
    t0 = now();
    futhark_entry_prepare_scene(ctx, &ps, n, n, scene);
    futhark_context_sync(ctx);
    printf("prepare_scene %d: %.3f ms\n", i, now() - t0);
  }
  struct futhark_i32_2d *img = NULL;
  for (int i = 0; i < calls; ++i) {
    if (img) futhark_free_i32_2d(ctx, img);
    t0 = now();
    futhark_entry_render(ctx, &img, n, n, ps);
    const double t1 = now();
    futhark_context_sync(ctx);
    printf("render %d: call %.3f ms + sync %.3f ms\n", i, t1 - t0, now() - t1);
  }
  futhark_free_i32_2d(ctx, img);
  futhark_free_opaque_prepared_scene(ctx, ps);
  futhark_free_opaque_scene(ctx, scene);
  futhark_context_free(ctx);
  futhark_context_config_free(cfg);
  return 0;
}
