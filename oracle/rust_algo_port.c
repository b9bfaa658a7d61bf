/*
 * rust_algo_port.c -- SECOND CPU BASELINE, TIMING ONLY.  Test infrastructure (see ray_oracle.c).
 *
 * A C restatement of the reference's Rust implementation of the render path
 * (/root/reference/rust/src/lib.rs), the other CPU comparator BASELINE.json's north_star
 * names.  It traces a DIFFERENT image than the Futhark program: top-down
 * median-split BVH instead of the LBVH, hit epsilon 0.001, box interval narrowed by the
 * current best hit, bounce check after the hit -- so it is NOT a parity oracle and nothing
 * is compared against it.  "parity unpinned": the reference ships no known answers for the
 * Rust variant, and no Rust toolchain exists here to produce any.  It only answers "how fast
 * does the reference's Rust algorithm run on this box's cores".
 *
 *   Bvh::new      lib.rs:293-338   sort by centre on axis d%3 (z compares a with a: no-op),
 *                                  split at n/2, recurse
 *   Objs::hit     lib.rs:342-361   recursive, right subtree searched with t_max = left hit's t
 *   Aabb::hit     lib.rs:100-123   Sphere::hit lib.rs:237-267   Ray::colour lib.rs:198-219
 *   render        lib.rs:430-444   pixel l -> (i = l % w, j = h - l / w)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 v = {x, y, z}; return v; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 normalise(v3 a) { return scl(a, 1.0f / sqrtf(dot(a, a))); }
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 reflect(v3 v, v3 n) { return sub(v, scl(n, 2.0f * dot(v, n))); }

typedef struct { v3 pos, colour; float radius; } sphere;   /* same layout as orc_sphere */
typedef struct { v3 mn, mx; } aabb;
typedef struct { v3 origin, dir; } ray;
typedef struct { float t; v3 p, normal, colour; } hit;

/* Bvh<T> as a flat array of nodes: leaf when left < 0 (then `obj` is the sphere index) */
typedef struct { aabb box; int32_t left, right, obj; } node;
typedef struct {
  int64_t n;
  sphere *spheres;   /* reordered copy */
  node *nodes;       /* 2n - 1 */
  int32_t used;
} rust_bvh;

static inline aabb to_aabb(const sphere *s) {
  aabb b;
  b.mn = sub(s->pos, V(s->radius, s->radius, s->radius));
  b.mx = add(s->pos, V(s->radius, s->radius, s->radius));
  return b;
}
static inline v3 centre(aabb b) {
  return V(b.mn.x + 0.5f * (b.mx.x - b.mn.x), b.mn.y + 0.5f * (b.mx.y - b.mn.y), b.mn.z + 0.5f * (b.mx.z - b.mn.z));
}
static inline aabb enclosing(aabb a, aabb b) {
  aabb r;
  r.mn = V(fminf(a.mn.x, b.mn.x), fminf(a.mn.y, b.mn.y), fminf(a.mn.z, b.mn.z));
  r.mx = V(fmaxf(a.mx.x, b.mx.x), fmaxf(a.mx.y, b.mx.y), fmaxf(a.mx.z, b.mx.z));
  return r;
}

/* stable merge sort of xs[0..n) by key (par_sort_by is a stable merge sort) */
static void stable_sort(sphere *xs, float *key, int64_t n, sphere *tmp, float *ktmp) {
  if (n < 2) return;
  int64_t h = n / 2;
  stable_sort(xs, key, h, tmp, ktmp);
  stable_sort(xs + h, key + h, n - h, tmp, ktmp);
  int64_t i = 0, j = h, k = 0;
  while (i < h && j < n) {
    if (key[j] < key[i]) { tmp[k] = xs[j]; ktmp[k++] = key[j++]; }
    else { tmp[k] = xs[i]; ktmp[k++] = key[i++]; }
  }
  while (i < h) { tmp[k] = xs[i]; ktmp[k++] = key[i++]; }
  while (j < n) { tmp[k] = xs[j]; ktmp[k++] = key[j++]; }
  memcpy(xs, tmp, sizeof(sphere) * (size_t)n);
  memcpy(key, ktmp, sizeof(float) * (size_t)n);
}

static int32_t build(rust_bvh *b, int d, sphere *xs, int64_t n, sphere *tmp, float *key, float *ktmp) {
  int32_t me = b->used++;
  if (n == 1) {
    b->nodes[me].box = to_aabb(&xs[0]);
    b->nodes[me].left = b->nodes[me].right = -1;
    b->nodes[me].obj = (int32_t)(xs - b->spheres);
    return me;
  }
  if (d % 3 != 2) {   /* the z comparator compares a with a (lib.rs:311): Equal, a stable no-op */
    for (int64_t i = 0; i < n; i++) {
      v3 c = centre(to_aabb(&xs[i]));
      key[i] = d % 3 == 0 ? c.x : c.y;
    }
    stable_sort(xs, key, n, tmp, ktmp);
  }
  int32_t l = build(b, d + 1, xs, n / 2, tmp, key, ktmp);
  int32_t r = build(b, d + 1, xs + n / 2, n - n / 2, tmp, key, ktmp);
  b->nodes[me].left = l; b->nodes[me].right = r; b->nodes[me].obj = -1;
  b->nodes[me].box = enclosing(b->nodes[l].box, b->nodes[r].box);
  return me;
}

int rust_bvh_build(const sphere *spheres, int64_t n, rust_bvh *out) {
  memset(out, 0, sizeof *out);
  out->n = n;
  out->spheres = (sphere *)malloc(sizeof(sphere) * (size_t)n);
  out->nodes = (node *)malloc(sizeof(node) * (size_t)(2 * n));
  sphere *tmp = (sphere *)malloc(sizeof(sphere) * (size_t)n);
  float *key = (float *)malloc(sizeof(float) * (size_t)n), *ktmp = (float *)malloc(sizeof(float) * (size_t)n);
  if (!out->spheres || !out->nodes || !tmp || !key || !ktmp) return 1;
  memcpy(out->spheres, spheres, sizeof(sphere) * (size_t)n);
  build(out, 0, out->spheres, n, tmp, key, ktmp);
  free(tmp); free(key); free(ktmp);
  return 0;
}
void rust_bvh_free(rust_bvh *b) { free(b->spheres); free(b->nodes); memset(b, 0, sizeof *b); }

static inline int aabb_hit(const aabb *b, const ray *r, float tmin, float tmax) {
#define SLAB(mn, mx, o, d)                          \
  do {                                               \
    float inv = 1.0f / (d);                          \
    float t0 = ((mn) - (o)) * inv, t1 = ((mx) - (o)) * inv; \
    if (inv < 0.0f) { float s = t0; t0 = t1; t1 = s; }     \
    tmin = fmaxf(t0, tmin); tmax = fminf(t1, tmax);  \
  } while (0)
  SLAB(b->mn.x, b->mx.x, r->origin.x, r->dir.x);
  if (tmax <= tmin) return 0;
  SLAB(b->mn.y, b->mx.y, r->origin.y, r->dir.y);
  if (tmax <= tmin) return 0;
  SLAB(b->mn.z, b->mx.z, r->origin.z, r->dir.z);
  return tmax > tmin;
#undef SLAB
}

static inline int sphere_hit(const sphere *s, const ray *r, float t_min, float t_max, hit *h) {
  v3 oc = sub(r->origin, s->pos);
  float a = dot(r->dir, r->dir), b = dot(oc, r->dir), c = dot(oc, oc) - s->radius * s->radius;
  float disc = b * b - a * c;
  if (disc <= 0.0f) return 0;
  float t = (-b - sqrtf(b * b - a * c)) / a;
  if (!(t < t_max && t > t_min)) {
    t = (-b + sqrtf(b * b - a * c)) / a;
    if (!(t < t_max && t > t_min)) return 0;
  }
  h->t = t;
  h->p = add(r->origin, scl(r->dir, t));
  h->normal = scl(sub(h->p, s->pos), 1.0f / s->radius);
  h->colour = s->colour;
  return 1;
}

static int objs_hit(const rust_bvh *b, int32_t ni, const ray *r, float t_min, float t_max, hit *out) {
  const node *nd = &b->nodes[ni];
  if (nd->left < 0) return sphere_hit(&b->spheres[nd->obj], r, t_min, t_max, out);
  if (!aabb_hit(&nd->box, r, t_min, t_max)) return 0;
  hit h;
  if (objs_hit(b, nd->left, r, t_min, t_max, &h)) {
    hit h2;
    *out = objs_hit(b, nd->right, r, t_min, h.t, &h2) ? h2 : h;
    return 1;
  }
  return objs_hit(b, nd->right, r, t_min, t_max, out);
}

static v3 colour(const rust_bvh *b, ray r, int depth, uint64_t *rays) {
  hit h;
  (*rays)++;
  if (objs_hit(b, 0, &r, 0.001f, 1000000000.0f, &h)) {
    ray sc;
    sc.origin = h.p;
    sc.dir = reflect(normalise(r.dir), h.normal);
    if (dot(sc.dir, h.normal) > 0.0f) {
      if (depth < 50) return mul(h.colour, colour(b, sc, depth + 1, rays));
      return V(0, 0, 0);
    }
    return V(0, 0, 0);
  }
  v3 u = normalise(r.dir);
  float t = 0.5f * (u.y + 1.0f);
  return add(scl(V(1, 1, 1), 1.0f - t), scl(V(0.5f, 0.7f, 1.0f), t));
}

/* camera: 12 floats origin, llc, horizontal, vertical (the same values the oracle computes) */
int rust_render(const rust_bvh *b, const float cam12[12], int64_t width, int64_t height, int threads, int32_t *out,
                uint64_t *rays_out) {
  const v3 origin = V(cam12[0], cam12[1], cam12[2]), llc = V(cam12[3], cam12[4], cam12[5]);
  const v3 hor = V(cam12[6], cam12[7], cam12[8]), ver = V(cam12[9], cam12[10], cam12[11]);
  uint64_t total = 0;
#ifdef _OPENMP
  omp_set_num_threads(threads > 0 ? threads : omp_get_num_procs());
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 128) reduction(+ : total)
  for (int64_t l = 0; l < width * height; l++) {
    const int64_t i = l % width, j = height - l / width;
    const float u = (float)i / (float)width, v = (float)j / (float)height;
    ray r;
    r.origin = origin;
    r.dir = sub(add(add(llc, scl(hor, u)), scl(ver, v)), origin);
    uint64_t rays = 0;
    v3 c = colour(b, r, 0, &rays);
    out[l] = ((int32_t)(c.x * 255.99f) << 16) | ((int32_t)(c.y * 255.99f) << 8) | (int32_t)(c.z * 255.99f);
    total += rays;
  }
  if (rays_out) *rays_out = total;
  return 0;
}
