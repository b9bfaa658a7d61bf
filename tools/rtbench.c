/*
 * rtbench.c -- native bench front-end for libray_mi355x.so (plain C, no torch).
 *
 * Same protocol as the reference's harness (futhark/main.c:88-124): BVH construction and
 * rendering are each timed as the mean over `-r` runs with the completion point being the
 * context sync.  On top of that it reports Mray/s, the algorithmic-bytes roofline figure
 * (32 B per box test + 16 B per sphere test + 4 B per pixel) and per-launch
 * HIP-event times, and lets every kernel knob be set from the command line:
 *
 *   rtbench -s rgbbox|irreg|big|floor:N:K -n H -m W -r RUNS -d MAX_DEPTH -v VARIANT
 *           -o name=value (repeatable) -f out.ppm -g PARTS -L LANES
 *
 * -g N: ONE multi-device context over devices 0..N-1 (rt_context_create_multi: cyclic row tiles, RCCL /
 * peer-copy gather on device 0); with fewer than N devices present the N parts all run on device 0
 * (test mode).  -L LANES > 1 adds a throughput figure: LANES contexts (own stream, own prepared scene, own
 * framebuffer each) keep one frame in flight each, all enqueued before any is awaited.
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "rt_mi355x.h"

static double now_s(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

#define CHECK(ctx, call)                                                              \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != 0) {                                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, (ctx) ? rt_last_error(ctx) : "?"); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

static void write_ppm(const char *path, const int32_t *px, int h, int w) {
  FILE *f = fopen(path, "w");
  if (!f) { perror(path); exit(1); }
  fprintf(f, "P3\n%d %d\n255\n", w, h);
  for (long i = 0; i < (long)h * w; i++)
    fprintf(f, "%d %d %d\n", (px[i] >> 16) & 0xFF, (px[i] >> 8) & 0xFF, px[i] & 0xFF);
  fclose(f);
}

int main(int argc, char **argv) {
  int h = 200, w = 200, runs = 10, depth = 50, variant = 0, parts = 1, lanes = 1, batch = 0;
  const char *scene_name = "rgbbox", *ppm = NULL;
  const char *opts[32];
  int nopts = 0, c;
  while ((c = getopt(argc, argv, "s:n:m:r:d:v:o:f:g:L:B:")) != -1) {
    switch (c) {
    case 's': scene_name = optarg; break;
    case 'n': h = atoi(optarg); break;
    case 'm': w = atoi(optarg); break;
    case 'r': runs = atoi(optarg); break;
    case 'd': depth = atoi(optarg); break;
    case 'v': variant = atoi(optarg); break;
    case 'o': if (nopts < 32) opts[nopts++] = optarg; break;
    case 'f': ppm = optarg; break;
    case 'g': parts = atoi(optarg); break;
    case 'L': lanes = atoi(optarg); break;
    case 'B': batch = atoi(optarg); break;
    default:
      fprintf(stderr, "usage: %s [-s scene] [-n height] [-m width] [-r runs] [-d max_depth] [-v variant] [-o k=v] [-f out.ppm] [-g parts]\n", argv[0]);
      return 2;
    }
  }
  rt_context *ctx = NULL;
  if (parts > 1) {
    int devs[64];
    const int have = rt_device_count();
    if (parts > 64) parts = 64;
    for (int i = 0; i < parts; i++) devs[i] = have >= parts ? i : 0;
    if (rt_context_create_multi(&ctx, devs, parts) != 0) { fprintf(stderr, "cannot create the multi-device context\n"); return 1; }
    printf("Multi-device context: %d parts on %s; framebuffer gather: %s\n", parts,
           have >= parts ? "devices 0..N-1" : "device 0 only (fewer devices present: test mode)", rt_context_gather_mode(ctx));
  } else if (rt_context_create(&ctx, -1, NULL, 0) != 0) { fprintf(stderr, "no HIP device\n"); return 1; }
  CHECK(ctx, rt_context_set_variant(ctx, variant));
  for (int i = 0; i < nopts; i++) {
    char key[64];
    const char *eq = strchr(opts[i], '=');
    if (!eq || (size_t)(eq - opts[i]) >= sizeof key) { fprintf(stderr, "bad -o %s\n", opts[i]); return 2; }
    memcpy(key, opts[i], (size_t)(eq - opts[i]));
    key[eq - opts[i]] = 0;
    CHECK(ctx, rt_context_set_option(ctx, key, atoll(eq + 1)));
  }
  int dev, cus, lds;
  char arch[64];
  rt_context_device_info(ctx, &dev, &cus, &lds, arch, sizeof arch);
  printf("Device %d: %s, %d CUs, %d B LDS/CU\n", dev, arch, cus, lds);

  rt_scene *scene = NULL;
  int fn; float fk;
  if (strcmp(scene_name, "rgbbox") == 0) CHECK(ctx, rt_scene_rgbbox(ctx, &scene));
  else if (strcmp(scene_name, "irreg") == 0) CHECK(ctx, rt_scene_irreg(ctx, &scene));
  else if (strcmp(scene_name, "big") == 0) CHECK(ctx, rt_scene_floor(ctx, &scene, 1000, 6000.0f));
  else if (sscanf(scene_name, "floor:%d:%f", &fn, &fk) == 2) CHECK(ctx, rt_scene_floor(ctx, &scene, fn, fk));
  else { fprintf(stderr, "Unknown scene: %s (rgbbox, irreg, big, floor:N:K)\n", scene_name); return 1; }
  printf("Using scene %s (%lld spheres), %dx%d, max_depth %d, variant %d.\n", scene_name,
         (long long)rt_scene_num_spheres(scene), w, h, depth, variant);
  printf("Timing over average of %d runs.\n", runs);

  rt_prepared *ps = NULL;
  double t0 = now_s();
  uint64_t st[3];
  int32_t *img = NULL;
  float *ms = NULL;
  if (runs == 0 && batch > 1) {
    /* batch-only mode (what the PMC passes profile): no single-frame launches of the pooled kernel */
    CHECK(ctx, rt_prepare_scene(ctx, &ps, h, w, scene));
    CHECK(ctx, rt_render_stats(ctx, ps, h, w, depth, st));
  } else {
  for (int i = 0; i < runs; i++) {
    if (ps) rt_prepared_free(ctx, ps);
    CHECK(ctx, rt_prepare_scene(ctx, &ps, h, w, scene));
    CHECK(ctx, rt_context_sync(ctx));
  }
  printf("Scene BVH construction in %fs.\n", (now_s() - t0) / runs);

  CHECK(ctx, rt_device_alloc(ctx, (void **)&img, (int64_t)sizeof(int32_t) * h * w));
  /* warm-up launch (module load, clocks), then the timed loop */
  CHECK(ctx, rt_render_part(ctx, ps, h, w, depth, 8, 0, 1, img));
  CHECK(ctx, rt_context_sync(ctx));
  t0 = now_s();
  for (int i = 0; i < runs; i++) {
    CHECK(ctx, rt_render_part(ctx, ps, h, w, depth, 8, 0, 1, img));   /* a multi-device context fans the frame out itself */
    CHECK(ctx, rt_context_sync(ctx));
  }
  double t_render = (now_s() - t0) / runs;
  printf("Rendering in %fs.\n", t_render);

  ms = (float *)malloc(sizeof(float) * (size_t)runs);
  CHECK(ctx, rt_render_timed(ctx, ps, h, w, depth, 8, 0, 1, img, 2, runs, ms));
  double sum = 0, mn = 1e30;
  for (int i = 0; i < runs; i++) { sum += ms[i]; if (ms[i] < mn) mn = ms[i]; }
  double t_kernel = sum / runs * 1e-3;

  CHECK(ctx, rt_render_stats(ctx, ps, h, w, depth, st));
  double bytes_alg = 32.0 * (double)st[1] + 16.0 * (double)st[2] + 4.0 * (double)w * h;
  printf("Frame work: %llu rays, %llu box tests, %llu sphere tests, %.0f algorithmic bytes (%.1f B/ray)\n",
         (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], bytes_alg,
         bytes_alg / (double)st[0]);
  printf("HIP-event time per launch: mean %.4f ms, min %.4f ms\n", t_kernel * 1e3, mn);
  printf("Throughput: %.1f Mray/s (host-timed), %.1f Mray/s (kernel)\n", (double)st[0] / t_render * 1e-6,
         (double)st[0] / t_kernel * 1e-6);
  printf("Algorithmic bandwidth: %.1f GB/s = %.3f of the 8000 GB/s HBM3E roofline\n", bytes_alg / t_kernel * 1e-9,
         bytes_alg / t_kernel * 1e-9 / 8000.0);
  {
    /* the last timed frame's pixels: c = c * 31 + pixel (SURVEY.md 8c's convenience checksums) */
    int32_t *host = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
    CHECK(ctx, rt_copy_to_host(ctx, host, img, (int64_t)sizeof(int32_t) * h * w));
    uint32_t cs = 0;
    for (long i = 0; i < (long)h * w; i++) cs = cs * 31u + (uint32_t)host[i];
    printf("Checksum: %08x\n", cs);
    free(host);
  }

  }
  if (batch > 1) {   /* (on a multi-device context: every device its rows of all the frames in one launch, one gather) */
    int32_t *bimg = NULL;
    CHECK(ctx, rt_device_alloc(ctx, (void **)&bimg, (int64_t)sizeof(int32_t) * h * w * batch));
    for (int k = 0; k < 2; k++) CHECK(ctx, rt_render_batch(ctx, ps, h, w, depth, 8, 0, 1, batch, NULL, (int64_t)h * w, bimg));
    CHECK(ctx, rt_context_sync(ctx));
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
      double tb = now_s();
      CHECK(ctx, rt_render_batch(ctx, ps, h, w, depth, 8, 0, 1, batch, NULL, (int64_t)h * w, bimg));
      CHECK(ctx, rt_context_sync(ctx));
      const double t = (now_s() - tb) / batch;
      if (t < best) best = t;
    }
    {
      /* checksum of the LAST frame of the batch (c = c * 31 + pixel), as for a single frame */
      int32_t *hb = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
      uint32_t c = 0;
      if (hb && rt_copy_to_host(ctx, hb, bimg + (size_t)(batch - 1) * h * w, (int64_t)sizeof(int32_t) * h * w) == 0)
        for (size_t i = 0; i < (size_t)h * w; i++) c = c * 31u + (uint32_t)hb[i];
      free(hb);
      printf("Batch: %d frames in one launch%s: %.4f ms per frame, %.1f Mray/s; last frame's checksum %08x\n", batch,
             parts > 1 ? " per device" : "", best * 1e3, (double)st[0] / best * 1e-6, c);
    }
    rt_device_free(ctx, bimg);
  }

  if (lanes > 1) {
    /* throughput: `lanes` independent frames in flight (GPU_MAX_HW_QUEUES must allow that many queues) */
    if (lanes > 64) lanes = 64;
    rt_context *lc[64]; rt_scene *lsn[64]; rt_prepared *lp[64]; int32_t *li[64];
    for (int l = 0; l < lanes; l++) {
      if (rt_context_create(&lc[l], -1, NULL, 0) != 0) { fprintf(stderr, "lane context failed\n"); return 1; }
      CHECK(lc[l], rt_context_set_variant(lc[l], variant));
      for (int i = 0; i < nopts; i++) {
        char key[64];
        const char *eq = strchr(opts[i], '=');
        memcpy(key, opts[i], (size_t)(eq - opts[i]));
        key[eq - opts[i]] = 0;
        CHECK(lc[l], rt_context_set_option(lc[l], key, atoll(eq + 1)));
      }
      if (strcmp(scene_name, "rgbbox") == 0) CHECK(lc[l], rt_scene_rgbbox(lc[l], &lsn[l]));
      else if (strcmp(scene_name, "irreg") == 0) CHECK(lc[l], rt_scene_irreg(lc[l], &lsn[l]));
      else if (strcmp(scene_name, "big") == 0) CHECK(lc[l], rt_scene_floor(lc[l], &lsn[l], 1000, 6000.0f));
      else CHECK(lc[l], rt_scene_floor(lc[l], &lsn[l], fn, fk));
      CHECK(lc[l], rt_prepare_scene(lc[l], &lp[l], h, w, lsn[l]));
      CHECK(lc[l], rt_device_alloc(lc[l], (void **)&li[l], (int64_t)sizeof(int32_t) * h * w));
      for (int k = 0; k < 3; k++) CHECK(lc[l], rt_render_part(lc[l], lp[l], h, w, depth, 8, 0, 1, li[l]));   /* per-view caches */
      CHECK(lc[l], rt_context_sync(lc[l]));
    }
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      t0 = now_s();
      for (int i = 0; i < runs; i++)
        for (int l = 0; l < lanes; l++) CHECK(lc[l], rt_render_part(lc[l], lp[l], h, w, depth, 8, 0, 1, li[l]));
      for (int l = 0; l < lanes; l++) CHECK(lc[l], rt_context_sync(lc[l]));
      const double t = (now_s() - t0) / ((double)runs * lanes);
      if (t < best) best = t;
    }
    printf("Overlapped: %d lanes x %d frames: %.4f ms per frame, %.1f Mray/s\n", lanes, runs, best * 1e3,
           (double)st[0] / best * 1e-6);
    for (int l = 0; l < lanes; l++) {
      rt_device_free(lc[l], li[l]);
      rt_prepared_free(lc[l], lp[l]);
      rt_scene_free(lc[l], lsn[l]);
      rt_context_destroy(lc[l]);
    }
  }

  if (ppm && img) {
    int32_t *host = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
    CHECK(ctx, rt_render_part(ctx, ps, h, w, depth, 8, 0, 1, img));
    CHECK(ctx, rt_copy_to_host(ctx, host, img, (int64_t)sizeof(int32_t) * h * w));
    printf("Writing image to %s.\n", ppm);
    write_ppm(ppm, host, h, w);
    free(host);
  }
  free(ms);
  if (img) rt_device_free(ctx, img);
  rt_prepared_free(ctx, ps);
  rt_scene_free(ctx, scene);
  rt_context_destroy(ctx);
  return 0;
}
