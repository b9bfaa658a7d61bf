/*
 * rt_mi355x.h -- C ABI of the MI355X-native render hot path (libray_mi355x.so).
 *
 * This is the "internal" surface the Futhark-shaped drop-in boundary (include/ray.h)
 * is built on.  It mirrors the reference's language-level call surface
 *
 *     render(objs, width, height, cam) -> [pixel]
 *       futhark/ray.fut:166-169 (render_image), :241-247 (prepare_scene / render)
 *       rust/src/lib.rs:430-444, haskell/Raytracing.hs:187-190
 *
 * with plain pointers and sizes only (no torch / HIP types in any signature; a HIP
 * stream is passed as an opaque void*).  Every entry returns 0 on success and a
 * non-zero code on failure; rt_last_error() gives the message (the reference's
 * harness convention: `assert(ret == 0)`, futhark/main.c:74,97,116,131).
 *
 * All rendering entries ENQUEUE work on the context's stream and return; the
 * completion point is rt_context_sync() (futhark_context_sync, main.c:98,117).
 * There is no CPU fallback: without a usable HIP device context creation fails.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;    /* one device + one stream + work-queue state */
typedef struct rt_scene rt_scene;        /* `scene` (ray.fut:171-174): spheres + look_from/look_at/fov, host */
typedef struct rt_prepared rt_prepared;  /* `prepared_scene` (ray.fut:239): device-resident BVH + camera */

/* Kernel families (rt_context_set_variant).  All produce bit-identical pixels. */
enum {
  RT_VARIANT_AUTO = 0,        /* the pooled family; the pixel family for scenes beyond its limits: 2^22 spheres or more
                                 (a pooled work item is one dword, (node or leaf << 8) | ray slot) or a tree deeper than
                                 64 levels (the per-wave box stack no longer fits LDS) -- ~2 Gray/s instead of 6-16 */
  RT_VARIANT_PIXEL = 1,       /* one thread per pixel, BVH in HBM/L2, no LDS staging (BASELINE configs[1]) */
  RT_VARIANT_PERSISTENT = 2,  /* CROSS-CHECK family, selected by nothing: one ray per lane, work queue + in-place lane
                                 refill + phase voting.  A structurally different implementation of the same fold that
                                 the parity suite renders every case through; 5-15x slower than the pooled family */
  RT_VARIANT_POOLED = 3       /* persistent waves whose lanes share LDS work lists of (ray slot, node) items: any lane
                                 tests any ray's node; ballot/mbcnt compaction; work queue of tiles, or of the view's
                                 pixels sorted by bounce-chain length (BASELINE configs[2..4]) */
};

/* ---- context ------------------------------------------------------------------ */
/* device < 0: the current HIP device.  use_caller_stream == 0: the context creates and
 * owns a non-blocking stream (hip_stream is ignored).  use_caller_stream != 0: all work is
 * enqueued on the caller's hipStream_t `hip_stream` -- NULL then means the default stream
 * (which is what torch.cuda.current_stream().cuda_stream is unless the caller switched). */
int rt_context_create(rt_context **out, int device, void *hip_stream, int use_caller_stream);
int rt_device_count(void);   /* usable HIP devices (0: none -- there is no CPU path) */
/* One process, several devices: `devices[0..ndev)` are HIP device ordinals.  The context
 * behaves like a single-device one on devices[0] -- scenes, prepare_scene, render, sync, values -- but
 * prepare_scene replicates the scene on every device and rt_render / rt_render_image cut the frame into
 * cyclic tiles of 8 rows (part i of ndev on devices[i]) and gather the parts on devices[0]: RCCL
 * point-to-point over xGMI (librccl is loaded on demand), peer copies, or -- option "gather" = 3 -- no gather at
 * all: every device stores its pixels straight into the image on devices[0] over xGMI while it renders (peer
 * access; rt_render_part_inplace).  Option "gather": 0 auto (direct stores when every device can reach devices[0],
 * else RCCL, else peer copies), 1 peer copies, 2 RCCL, 3 direct stores.  A device may be listed more than once
 * (no RCCL then): that is how the fan-out is tested on a one-GPU box.  rt_render_part with nparts > 1 is refused
 * on such a context. */
int rt_context_create_multi(rt_context **out, const int *devices, int ndev);
int rt_context_num_devices(const rt_context *ctx);          /* 1 for an ordinary context */
const char *rt_context_gather_mode(rt_context *ctx);        /* "none", "direct-store", "rccl" or "peer-copy" (static strings) */
int rt_context_rccl_ranks(rt_context *ctx);                 /* ranks of the RCCL communicator behind the gather (0: RCCL is not what carries it) */
void rt_context_destroy(rt_context *ctx);
const char *rt_last_error(const rt_context *ctx);       /* "" when no error; owned by ctx */
/* What the last render entry of this context enqueued, for benches and profiles that want to assert which kernel ran (pixels never
 * depend on it).  Owned by ctx; "" before the first render.  (A multi-device context: the first device's part.)  Values:
 *   "family=none (memset)"        max_depth == 0: every pixel is the initial colour, no kernel
 *   "family=none (no rows)"       the part owns no row of the image
 *   "family=pixel" | "family=pixel (instrumented)" | "family=persistent"
 *   "family=pooled tickets=T instantiation=I[+CULL] frames=.. tiles=.. grid=.. waves=.. counters=..[(turns)] deep_class=.. deep_split=.. recording=0|1|2"
 *     T = pixel-list | tiles-ordered | tiles-bit-reversed (a view's first frame with nothing to borrow, first_order = 1) | tiles-raster,
 *         followed by "(borrowed)" when the order / list is another view's (a new view of a prepared scene that has rendered a view of the same shape)
 *     I = plain | SOLO | COLD | COLD+SOLO | DONATE | DONATE+SOLO | ORD | ORD+SOLO | ORD+DONATE | ORD+SOLO+DONATE;  +CULL: boxes tested against the best hit so far; +SPILL: a box stack that may overflow into device memory (twenty waves per CU, trees taller than 15 levels)
 *     recording: 0 nothing, 1 the tiles' longest chains, 2 also every pixel's chain length */
const char *rt_context_last_launch(const rt_context *ctx);
int rt_context_sync(rt_context *ctx);
/* Threads: every entry that takes a context holds that context's lock for the duration of the call (as a Futhark context does:
 * SURVEY.md 8b) -- host threads may share a context, a prepared scene and a scene; their calls are serialised and their frames
 * run on the context's one stream in the order the calls were admitted.  rt_last_error / rt_context_last_launch return
 * pointers into the context: read them before another thread's call replaces the string (or use a context per thread). */
int rt_context_set_variant(rt_context *ctx, int variant);
/* Tuning knobs by name (see DESIGN.md "knobs"); unknown name -> error. */
int rt_context_set_option(rt_context *ctx, const char *name, int64_t value);
int rt_context_device_info(const rt_context *ctx, int *device, int *num_cu, int *lds_bytes, char *name, int name_len);

/* ---- scenes (ray.fut:176-237) -------------------------------------------------- */
int rt_scene_rgbbox(rt_context *ctx, rt_scene **out);
int rt_scene_irreg(rt_context *ctx, rt_scene **out);
/* The irreg generator with its constants exposed: n x n spheres, extent k, y = 0.
 * irreg == (100, 600); SURVEY 8(d) "big" == (1000, 6000). */
int rt_scene_floor(rt_context *ctx, rt_scene **out, int n, float k);
/* Arbitrary scene: spheres7 = n x {pos.xyz, colour.xyz, radius}. */
int rt_scene_from_spheres(rt_context *ctx, rt_scene **out, const float *spheres7, int64_t n,
                          const float look_from[3], const float look_at[3], float fov);
int64_t rt_scene_num_spheres(const rt_scene *scene);
int rt_scene_free(rt_context *ctx, rt_scene *scene);

/* ---- prepare_scene (ray.fut:241-244): BVH build + camera, on the device ------------------
 * Values are tied to their context, as in the Futhark C API: the first prepare_scene of a scene
 * uploads its spheres to the context's device (the scene is device resident from then on), the
 * prepared scene's arrays live in the context's memory pool -- free it with the SAME context,
 * before that context is destroyed. */
int rt_prepare_scene(rt_context *ctx, rt_prepared **out, int64_t h, int64_t w, const rt_scene *scene);
int rt_prepared_free(rt_context *ctx, rt_prepared *ps);
int64_t rt_prepared_num_spheres(const rt_prepared *ps);
int32_t rt_prepared_height(const rt_prepared *ps);   /* levels of inner nodes on the longest root-to-leaf path */
/* Canonical `bvh = {L, I}` (bvh.fut:28) copied back from the device for parity checks.
 * L7: n x 7 floats; bmin/bmax: (n-1) x 3; left/right: (n-1) encoded ptr (inner i -> i,
 * leaf i -> -2 - i); parent: (n-1).  Any pointer may be NULL. */
int rt_prepared_get_bvh(rt_context *ctx, const rt_prepared *ps, float *L7, float *bmin, float *bmax,
                        int32_t *left, int32_t *right, int32_t *parent);
int rt_prepared_get_camera(rt_context *ctx, const rt_prepared *ps, float cam12[12]);

/* ---- render (ray.fut:246-247 -> render_image :166-169) -------------------------- */
/* Whole image: out_dev = device pointer to h*w int32, row-major from the top row. */
int rt_render(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t *out_dev);
/* Row-tile partition for multi-GPU: the image is cut into tiles of rows_per_tile rows;
 * part p of nparts renders tiles t with t % nparts == p, packed in tile order into
 * out_dev (rt_part_rows(h, rows_per_tile, p, nparts) * w int32).  max_depth: the
 * reference's bounce limit is 50 (ray.fut:154). */
int rt_render_part(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                   int32_t rows_per_tile, int32_t part, int32_t nparts, int32_t *out_dev);
/* The language-level surface itself: render_image objs width height cam (ray.fut:166) with an
 * explicit camera (12 floats: origin, llc, horizontal, vertical; ray.fut:88-91) instead of the
 * one prepare_scene derived.  cam12 == NULL: the prepared camera, at any h x w -- as `render h w
 * prepared` does in the reference (ray.fut:246): the aspect ratio stays the one prepare_scene was given. */
int rt_render_image(rt_context *ctx, const rt_prepared *objs, int64_t width, int64_t height, const float cam12[12],
                    int32_t max_depth, int32_t rows_per_tile, int32_t part, int32_t nparts, int32_t *out_dev);
/* Throughput entry: `nframes` frames of one prepared scene in ONE launch (a camera path, or the same view again and
 * again as the reference's harness does, main.c:107-124).  Frame f is traced through cams12 + 12 f (host memory; NULL:
 * the prepared camera for every frame) into out_dev + f * frame_stride (int32 elements, >= rows * w).  The persistent
 * waves run straight across frame boundaries, so a launch's fill and drain are paid once per batch instead of once
 * per frame.  Partition arguments as rt_render_part.  The array at cams12 is copied before the call returns.
 * On a multi-device context (part 0 of 1 only): every device renders its row tiles of ALL the frames in one launch, the
 * framebuffer gather moves nframes x part per device, one assembly launch writes the nframes images. */
int rt_render_batch(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                    int32_t part, int32_t nparts, int32_t nframes, const float *cams12, int64_t frame_stride, int32_t *out_dev);
/* The same part of the same frames, but stored IN PLACE: image_dev is the FULL image (frame f at image_dev + f *
 * frame_stride, frame_stride >= h * w; 0 = h * w for one frame) and every pixel of the part goes to its place in it
 * -- no packed part buffer, no gather, no assembly launch.  image_dev may be memory of ANOTHER device of the node
 * (a peer allocation, or a buffer of another process mapped with rt_ipc_import): the 4-byte pixel stores then are
 * the framebuffer exchange of SURVEY.md 8(e) -- they cross xGMI while the frame is being traced instead of in a
 * gather behind it (futhark/main.c:107-135 has nothing to correspond: the reference is a single device).  The
 * caller orders the consumers of the image behind every part's stream (event, collective, barrier). */
int rt_render_part_inplace(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth, int32_t rows_per_tile,
                           int32_t part, int32_t nparts, int32_t nframes, const float *cams12, int64_t frame_stride,
                           int32_t *image_dev);
/* One process per GPU: the owner of an image exports the allocation (a pointer rt_device_alloc returned) as 64
 * opaque bytes (hipIpcMemHandle_t), the other processes import them and get a device pointer usable as image_dev
 * above; rt_ipc_close unmaps it (before the owner frees the buffer). */
int rt_ipc_export(rt_context *ctx, void *dev, unsigned char handle64[64]);
int rt_ipc_import(rt_context *ctx, const unsigned char handle64[64], void **out_dev);
int rt_ipc_close(rt_context *ctx, void *imported_dev);
int64_t rt_part_rows(int64_t h, int32_t rows_per_tile, int32_t part, int32_t nparts);
/* Scatter one part's packed rows into a full h*w image on the device (rank-0 side of
 * the framebuffer gather). */
int rt_place_part(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t part, int32_t nparts,
                  const int32_t *part_dev, int32_t *image_dev);

/* The same for ALL parts at once: stacked_dev = nparts x pad_rows x w int32 (what a gather of the
 * ranks' padded send buffers delivers on rank 0), one kernel. */
int rt_place_parts(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t nparts, int64_t pad_rows,
                   const int32_t *stacked_dev, int32_t *image_dev);
/* General form: part p's packed rows start at stacked_dev + p * part_stride (int32 elements), so one
 * gathered buffer may carry the parts of several frames (one gather per step, see dist.py). */
int rt_place_parts_strided(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t nparts,
                           int64_t part_stride, const int32_t *stacked_dev, int32_t *image_dev);

/* ... and of a batch, in one launch: frame f's parts lie frame_stride_in elements behind frame f - 1's inside every
 * part (part_stride >= (nframes - 1) * frame_stride_in + the largest part), its image frame_stride_out (>= h * w)
 * elements behind the previous one at images_dev. */
int rt_place_parts_batch(rt_context *ctx, int64_t h, int64_t w, int32_t rows_per_tile, int32_t nparts, int64_t part_stride,
                         int32_t nframes, int64_t frame_stride_in, int64_t frame_stride_out, const int32_t *stacked_dev,
                         int32_t *images_dev);

/* Work counters of one frame, computed on the device by an instrumented launch of the
 * same traversal: rays (objs_hit calls), box tests, sphere tests. */
int rt_render_stats(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                    uint64_t stats3[3]);

/* Diagnostic: one instrumented launch of the pooled kernel; per wave 16 x u64 = {wall clock (100 MHz ticks,
 * chip-wide) at start, at queue exhaustion, at exit; #BOX | #LEAF << 21 | #SHADE << 42 operations; shader
 * cycles lived; #BOXT | #BOX2 << 32 (both are counted in #BOX as well); (box items << 32 | leaf items); deepest bounce
 * chain finished | max box stack << 16 | max leaf list << 32; shader cycles spent inside BOX, BOX2, BOXT, LEAF, SHADE
 * operations (words 8..12); 13..15 reserved}. */
int rt_render_trace(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                    uint64_t *records, int32_t max_waves, int32_t *num_waves);

/* Times `iters` back-to-back launches of rt_render_part with HIP events recorded on the
 * context's stream (after `warmup` untimed launches); ms_out[i] = duration of launch i. */
int rt_render_timed(rt_context *ctx, const rt_prepared *ps, int64_t h, int64_t w, int32_t max_depth,
                    int32_t rows_per_tile, int32_t part, int32_t nparts, int32_t *out_dev,
                    int32_t warmup, int32_t iters, float *ms_out);

/* ---- device buffers for hosts that have no allocator of their own (the C harness) -- */
int rt_device_alloc(rt_context *ctx, void **out_dev, int64_t bytes);
int rt_device_free(rt_context *ctx, void *dev);
int rt_copy_to_host(rt_context *ctx, void *dst_host, const void *src_dev, int64_t bytes);  /* synchronous */

#ifdef __cplusplus
}
#endif
#endif
