/*
 * ray_oracle.c -- CPU ORACLE for the render hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under raytracers_amd/ links,
 * imports or calls it.
 *
 * It is a plain-C restatement of the reference's Futhark program, written to be
 * bit-identical to it in IEEE-754 binary32 (build with -ffp-contract=off, no
 * -ffast-math; x86-64 SSE has no excess precision).  Every function cites the
 * reference lines it follows (paths relative to /root/reference):
 *
 *   futhark/prim.fut       vec3 / aabb arithmetic (operation order defines parity)
 *   futhark/radixtree.fut  Karras binary radix tree with index tie-break
 *   futhark/bvh.fut        Morton codes, stable sort, AABB sweeps, stackless fold
 *   futhark/ray.fut        sphere/aabb hit, scatter loop, camera, scenes, pixels
 *
 * Parity pin: tests/test_oracle_golden.py checks this oracle against the reference's
 * only known-answer fixtures for the path, /root/reference/rgbbox.png and irreg.png
 * (500x500), committed in decoded form under tests/golden/ (0 differing pixels).
 *
 * The reference itself (Futhark) cannot be compiled in this environment (no futhark
 * compiler; the generated ray.c is git-ignored upstream), so there is no oracle/_ref.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ prim.fut */

typedef struct { float x, y, z; } vec3;                 /* prim.fut:1 */
typedef struct { vec3 min, max; } aabb;                 /* prim.fut:36 */

static inline vec3 vec(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 vec_add(vec3 a, vec3 b) { return vec(a.x + b.x, a.y + b.y, a.z + b.z); } /* prim.fut:12 */
static inline vec3 vec_sub(vec3 a, vec3 b) { return vec(a.x - b.x, a.y - b.y, a.z - b.z); } /* prim.fut:13 */
static inline vec3 vec_mul(vec3 a, vec3 b) { return vec(a.x * b.x, a.y * b.y, a.z * b.z); } /* prim.fut:14 */
static inline vec3 scale(float s, vec3 v) { return vec(s * v.x, s * v.y, s * v.z); }        /* prim.fut:17-20 */

/* prim.fut:22-24: products rounded separately, then (x+y)+z. */
static inline float dot(vec3 a, vec3 b) {
  vec3 p = vec_mul(a, b);
  return (p.x + p.y) + p.z;
}
static inline float norm(vec3 v) { return sqrtf(dot(v, v)); }                               /* prim.fut:26 */
static inline vec3 normalise(vec3 v) { return scale(1.0f / norm(v), v); }                   /* prim.fut:28 */
static inline vec3 cross(vec3 a, vec3 b) {                                                  /* prim.fut:30-33 */
  return vec(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline aabb enclosing(aabb b0, aabb b1) {                                            /* prim.fut:38-45 */
  aabb r;
  r.min = vec(fminf(b0.min.x, b1.min.x), fminf(b0.min.y, b1.min.y), fminf(b0.min.z, b1.min.z));
  r.max = vec(fmaxf(b0.max.x, b1.max.x), fmaxf(b0.max.y, b1.max.y), fmaxf(b0.max.z, b1.max.z));
  return r;
}
static inline vec3 centre(aabb b) {                                                         /* prim.fut:47-50 */
  return vec(b.min.x + 0.5f * (b.max.x - b.min.x),
             b.min.y + 0.5f * (b.max.y - b.min.y),
             b.min.z + 0.5f * (b.max.z - b.min.z));
}

/* ------------------------------------------------------------------ public types */

typedef struct { vec3 pos; vec3 colour; float radius; } orc_sphere;   /* ray.fut:22-24 */

/* `ptr = #leaf i32 | #inner i32` (bvh.fut:24) as one int32: inner i -> i (>= -1),
 * leaf i -> -2 - i.  Equality of encodings == equality of (tag, payload). */
#define PTR_INNER(i) ((int32_t)(i))
#define PTR_LEAF(i) ((int32_t)(-2 - (i)))
#define PTR_IS_LEAF(p) ((p) <= -2)
#define PTR_LEAF_IDX(p) (-2 - (p))

/* inner = {aabb, left, right, parent} (bvh.fut:26), stored SoA like Futhark does. */
typedef struct {
  int64_t n;           /* leaves; n-1 inner nodes */
  orc_sphere *L;       /* [n] spheres sorted by Morton code (bvh.fut:59) */
  uint32_t *morton;    /* [n] sorted keys (diagnostic) */
  float *bmin;         /* [n-1][3] */
  float *bmax;         /* [n-1][3] */
  int32_t *left;       /* [n-1] encoded ptr */
  int32_t *right;      /* [n-1] encoded ptr */
  int32_t *parent;     /* [n-1] */
} orc_bvh;

typedef struct { vec3 origin, llc, horizontal, vertical; } orc_camera; /* ray.fut:88-91 */

typedef struct {
  vec3 look_from, look_at;
  float fov;
  int64_t n;
  orc_sphere *spheres;
} orc_scene;                                                            /* ray.fut:171-174 */

typedef struct {
  uint64_t rays;        /* objs_hit calls (ray.fut:130) */
  uint64_t steps;       /* bvh_fold loop iterations (bvh.fut:63) */
  uint64_t box_tests;   /* aabb_hit calls (ray.fut:77) */
  uint64_t leaf_tests;  /* sphere_hit calls inside the fold (ray.fut:79) */
  uint64_t max_steps;   /* longest single fold */
} orc_counters;

/* ------------------------------------------------------------------ scenes (ray.fut:176-237) */

static void wall(orc_sphere *out, int n, float k, int kind, vec3 colour) {
  /* kind 0: x fixed at -k/2, (a,b) = (y,z)   leftwall   ray.fut:180-187
   * kind 1: z fixed at -k/2, (a,b) = (x,y)   midwall    ray.fut:189-196
   * kind 2: x fixed at +k/2, (a,b) = (y,z)   rightwall  ray.fut:198-205
   * kind 3: y fixed at -k/2, (a,b) = (x,z)   bottom     ray.fut:208-215 */
  float fn = (float)n;
  float radius = k / (fn * 2.0f);
  for (int a = 0; a < n; a++)
    for (int b = 0; b < n; b++) {
      float pa = -k / 2.0f + (k / fn) * (float)a;
      float pb = -k / 2.0f + (k / fn) * (float)b;
      orc_sphere s;
      switch (kind) {
      case 0: s.pos = vec(-k / 2.0f, pa, pb); break;
      case 1: s.pos = vec(pa, pb, -k / 2.0f); break;
      case 2: s.pos = vec(k / 2.0f, pa, pb); break;
      default: s.pos = vec(pa, -k / 2.0f, pb); break;
      }
      s.colour = colour;
      s.radius = radius;
      out[a * n + b] = s;
    }
}

/* ray.fut:176-221 */
int orc_scene_rgbbox(orc_scene *sc) {
  const int n = 10;
  const float k = 60.0f;
  sc->n = 4 * n * n;
  sc->spheres = (orc_sphere *)malloc(sizeof(orc_sphere) * (size_t)sc->n);
  if (!sc->spheres) return 1;
  wall(sc->spheres + 0 * n * n, n, k, 0, vec(1.0f, 0.0f, 0.0f));
  wall(sc->spheres + 1 * n * n, n, k, 1, vec(1.0f, 1.0f, 0.0f));
  wall(sc->spheres + 2 * n * n, n, k, 2, vec(0.0f, 0.0f, 1.0f));
  wall(sc->spheres + 3 * n * n, n, k, 3, vec(1.0f, 1.0f, 1.0f));
  sc->look_from = vec(0.0f, 30.0f, 30.0f);
  sc->look_at = vec(0.0f, -1.0f, -1.0f);
  sc->fov = 75.0f;
  return 0;
}

/* The irreg generator (ray.fut:223-237) with its two constants exposed: the
 * reference scene is (n=100, k=600); SURVEY.md 8(d) config C5 "big" is the same
 * generator at (n=1000, k=6000). */
int orc_scene_floor(orc_scene *sc, int n, float k) {
  float fn = (float)n;
  sc->n = (int64_t)n * n;
  sc->spheres = (orc_sphere *)malloc(sizeof(orc_sphere) * (size_t)sc->n);
  if (!sc->spheres) return 1;
  for (int x = 0; x < n; x++)
    for (int z = 0; z < n; z++) {
      orc_sphere s;
      s.pos = vec(-k / 2.0f + (k / fn) * (float)x, 0.0f, -k / 2.0f + (k / fn) * (float)z);
      s.colour = vec(1.0f, 1.0f, 1.0f);
      s.radius = k / (fn * 2.0f);
      sc->spheres[(size_t)x * n + z] = s;
    }
  sc->look_from = vec(0.0f, 12.0f, 30.0f);
  sc->look_at = vec(0.0f, 10.0f, -1.0f);
  sc->fov = 75.0f;
  return 0;
}

int orc_scene_irreg(orc_scene *sc) { return orc_scene_floor(sc, 100, 600.0f); }

void orc_scene_free(orc_scene *sc) { free(sc->spheres); sc->spheres = NULL; sc->n = 0; }

/* ------------------------------------------------------------------ bvh.fut: Morton codes */

static inline uint32_t expand_bits(uint32_t v) {        /* bvh.fut:8-13 */
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

static inline uint32_t morton_3D(vec3 p) {              /* bvh.fut:15-22 */
  float x = fminf(fmaxf(p.x * 1024.0f, 0.0f), 1023.0f);
  float y = fminf(fmaxf(p.y * 1024.0f, 0.0f), 1023.0f);
  float z = fminf(fmaxf(p.z * 1024.0f, 0.0f), 1023.0f);
  uint32_t xx = expand_bits((uint32_t)x);
  uint32_t yy = expand_bits((uint32_t)y);
  uint32_t zz = expand_bits((uint32_t)z);
  return xx * 4u + yy * 2u + zz;
}

static inline aabb sphere_aabb(const orc_sphere *s) {   /* ray.fut:28-30 */
  aabb b;
  vec3 r = vec(s->radius, s->radius, s->radius);
  b.min = vec_sub(s->pos, r);
  b.max = vec_add(s->pos, r);
  return b;
}

/* ------------------------------------------------------------------ radixtree.fut */

static inline int32_t clz32(uint32_t v) { return v ? (int32_t)__builtin_clz(v) : 32; }

/* radixtree.fut:13-21 */
static inline int32_t delta(const uint32_t *L, int32_t n, int32_t i, int32_t j) {
  if (j >= 0 && j < n) {
    uint32_t Li = L[i], Lj = L[j];
    if (Li == Lj) return 32 + clz32((uint32_t)i ^ (uint32_t)j);
    return clz32(Li ^ Lj);
  }
  return -1;
}

static inline int32_t sgn32(int32_t v) { return (v > 0) - (v < 0); }
static inline int32_t imin32(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax32(int32_t a, int32_t b) { return a > b ? a : b; }

/* radixtree.fut:23-64 (one inner node) and :66-72 (parent scatter). */
static void mk_radix_tree(const uint32_t *L, int32_t n, int32_t *left, int32_t *right, int32_t *parent) {
  for (int32_t i = 0; i < n - 1; i++) parent[i] = -1;              /* replicate (n-1) (-1) */
  for (int32_t i = 0; i < n - 1; i++) {
    int32_t d = sgn32(delta(L, n, i, i + 1) - delta(L, n, i, i - 1));   /* :27 */
    int32_t delta_min = delta(L, n, i, i - d);                          /* :30 */
    int32_t l_max = 2;
    while (delta(L, n, i, i + l_max * d) > delta_min) l_max *= 2;       /* :31-33 */
    int32_t l = 0;
    for (int32_t t = l_max / 2; t > 0; t /= 2)                          /* :36-40 */
      if (delta(L, n, i, i + (l + t) * d) > delta_min) l += t;
    int32_t j = i + l * d;                                              /* :41 */
    int32_t delta_node = delta(L, n, i, j);                             /* :44 */
    int32_t s = 0;
    for (int32_t q = 1; q <= l; q *= 2) {                               /* :45-50 */
      int32_t t = (l + q * 2 - 1) / (q * 2);                            /* div_rounding_up :4 */
      if (delta(L, n, i, i + (s + t) * d) > delta_node) s += t;
    }
    int32_t gamma = i + s * d + imin32(d, 0);                           /* :51 */
    if (imin32(i, j) == gamma) {                                        /* :54-57 */
      left[i] = PTR_LEAF(gamma);
    } else {
      left[i] = PTR_INNER(gamma);
      parent[gamma] = i;
    }
    if (imax32(i, j) == gamma + 1) {                                    /* :59-62 */
      right[i] = PTR_LEAF(gamma + 1);
    } else {
      right[i] = PTR_INNER(gamma + 1);
      parent[gamma + 1] = i;
    }
  }
}

/* ------------------------------------------------------------------ bvh.fut:30-59 bvh_mk */

/* Stable LSD radix sort of indices by u32 key, 8 bits per pass.  The reference uses
 * diku-dk/sorts radix_sort_by_key (2 bits per pass, radix_sort.fut:14-68; call site
 * bvh.fut:43); any stable sort by the same key yields the same permutation. */
static void stable_sort_by_key(const uint32_t *keys, int64_t n, int64_t *perm) {
  int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) perm[i] = i;
  for (int pass = 0; pass < 4; pass++) {
    int64_t count[257];
    memset(count, 0, sizeof count);
    for (int64_t i = 0; i < n; i++) count[((keys[perm[i]] >> (8 * pass)) & 0xFF) + 1]++;
    for (int b = 0; b < 256; b++) count[b + 1] += count[b];
    for (int64_t i = 0; i < n; i++) tmp[count[(keys[perm[i]] >> (8 * pass)) & 0xFF]++] = perm[i];
    memcpy(perm, tmp, sizeof(int64_t) * (size_t)n);
  }
  free(tmp);
}

void orc_bvh_free(orc_bvh *b) {
  free(b->L); free(b->morton); free(b->bmin); free(b->bmax);
  free(b->left); free(b->right); free(b->parent);
  memset(b, 0, sizeof *b);
}

int orc_bvh_build(const orc_sphere *ts, int64_t n, orc_bvh *out) {
  memset(out, 0, sizeof *out);
  if (n < 2) return 1;
  size_t ni = (size_t)(n - 1);
  vec3 *centers = (vec3 *)malloc(sizeof(vec3) * (size_t)n);
  uint32_t *keys = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
  int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  out->n = n;
  out->L = (orc_sphere *)malloc(sizeof(orc_sphere) * (size_t)n);
  out->morton = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
  out->bmin = (float *)malloc(sizeof(float) * 3 * ni);
  out->bmax = (float *)malloc(sizeof(float) * 3 * ni);
  out->left = (int32_t *)malloc(sizeof(int32_t) * ni);
  out->right = (int32_t *)malloc(sizeof(int32_t) * ni);
  out->parent = (int32_t *)malloc(sizeof(int32_t) * ni);

  /* bvh.fut:31-37 */
  float x_max = -INFINITY, y_max = -INFINITY, z_max = -INFINITY;
  float x_min = INFINITY, y_min = INFINITY, z_min = INFINITY;
  for (int64_t i = 0; i < n; i++) {
    centers[i] = centre(sphere_aabb(&ts[i]));
    x_max = fmaxf(x_max, centers[i].x); y_max = fmaxf(y_max, centers[i].y); z_max = fmaxf(z_max, centers[i].z);
    x_min = fminf(x_min, centers[i].x); y_min = fminf(y_min, centers[i].y); z_min = fminf(z_min, centers[i].z);
  }
  /* bvh.fut:38-41 */
  for (int64_t i = 0; i < n; i++) {
    vec3 q = vec((centers[i].x - x_min) / (x_max - x_min),
                 (centers[i].y - y_min) / (y_max - y_min),
                 (centers[i].z - z_min) / (z_max - z_min));
    keys[i] = morton_3D(q);
  }
  /* bvh.fut:43 */
  stable_sort_by_key(keys, n, perm);
  for (int64_t i = 0; i < n; i++) { out->L[i] = ts[perm[i]]; out->morton[i] = keys[perm[i]]; }
  /* bvh.fut:46 */
  mk_radix_tree(out->morton, (int32_t)n, out->left, out->right, out->parent);
  /* bvh.fut:44-45: every inner box starts as {(0,0,0),(0,0,0)} */
  memset(out->bmin, 0, sizeof(float) * 3 * ni);
  memset(out->bmax, 0, sizeof(float) * 3 * ni);
  /* bvh.fut:47-58: exactly `depth` Jacobi sweeps, each reading the previous array. */
  int depth = (int)log2f((float)n) + 2;
  float *pmin = (float *)malloc(sizeof(float) * 3 * ni), *pmax = (float *)malloc(sizeof(float) * 3 * ni);
  for (int it = 0; it < depth; it++) {
    memcpy(pmin, out->bmin, sizeof(float) * 3 * ni);
    memcpy(pmax, out->bmax, sizeof(float) * 3 * ni);
    for (size_t i = 0; i < ni; i++) {
      aabb c[2];
      int32_t p[2] = {out->left[i], out->right[i]};
      for (int k = 0; k < 2; k++) {
        if (PTR_IS_LEAF(p[k])) {
          c[k] = sphere_aabb(&out->L[PTR_LEAF_IDX(p[k])]);
        } else {
          const float *a = pmin + 3 * (size_t)p[k], *b = pmax + 3 * (size_t)p[k];
          c[k].min = vec(a[0], a[1], a[2]);
          c[k].max = vec(b[0], b[1], b[2]);
        }
      }
      aabb e = enclosing(c[0], c[1]);
      out->bmin[3 * i + 0] = e.min.x; out->bmin[3 * i + 1] = e.min.y; out->bmin[3 * i + 2] = e.min.z;
      out->bmax[3 * i + 0] = e.max.x; out->bmax[3 * i + 1] = e.max.y; out->bmax[3 * i + 2] = e.max.z;
    }
  }
  free(pmin); free(pmax); free(centers); free(keys); free(perm);
  return 0;
}

/* ------------------------------------------------------------------ ray.fut: intersection */

typedef struct { vec3 origin, dir; } ray;               /* ray.fut:11-12 */
typedef struct { float t; vec3 p, normal, colour; } hit; /* ray.fut:17-20 */

static inline vec3 point_at_param(ray r, float t) { return vec_add(r.origin, scale(t, r.dir)); } /* ray.fut:14-15 */

/* ray.fut:32-51.  Returns 1 and fills *h on #some. */
static inline int sphere_hit(const orc_sphere *s, ray r, float t_min, float t_max, hit *h) {
  vec3 oc = vec_sub(r.origin, s->pos);
  float a = dot(r.dir, r.dir);
  float b = dot(oc, r.dir);
  float c = dot(oc, oc) - s->radius * s->radius;
  float discriminant = b * b - a * c;
  if (discriminant <= 0.0f) return 0;
  float temp = (-b - sqrtf(b * b - a * c)) / a;
  if (!(temp < t_max && temp > t_min)) {
    temp = (-b + sqrtf(b * b - a * c)) / a;
    if (!(temp < t_max && temp > t_min)) return 0;
  }
  h->t = temp;
  h->p = point_at_param(r, temp);
  h->normal = scale(1.0f / s->radius, vec_sub(point_at_param(r, temp), s->pos));
  h->colour = s->colour;
  return 1;
}

/* ray.fut:53-70 */
static inline int aabb_hit(aabb box, ray r, float tmin0, float tmax0) {
#define ITER(mn, mx, o, d, tmin_in, tmax_in, tmin_out, tmax_out)                 \
  do {                                                                           \
    float invD = 1.0f / (d);                                                     \
    float t0 = ((mn) - (o)) * invD;                                              \
    float t1 = ((mx) - (o)) * invD;                                              \
    float t0s = invD < 0.0f ? t1 : t0;                                           \
    float t1s = invD < 0.0f ? t0 : t1;                                           \
    tmin_out = fmaxf(t0s, tmin_in);                                              \
    tmax_out = fminf(t1s, tmax_in);                                              \
  } while (0)
  float tmin1, tmax1, tmin2, tmax2, tmin3, tmax3;
  ITER(box.min.x, box.max.x, r.origin.x, r.dir.x, tmin0, tmax0, tmin1, tmax1);
  if (tmax1 <= tmin1) return 0;
  ITER(box.min.y, box.max.y, r.origin.y, r.dir.y, tmin1, tmax1, tmin2, tmax2);
  if (tmax2 <= tmin2) return 0;
  ITER(box.min.z, box.max.z, r.origin.z, r.dir.z, tmin2, tmax2, tmin3, tmax3);
  return !(tmax3 <= tmin3);
#undef ITER
}

static inline aabb node_aabb(const orc_bvh *t, int32_t i) {
  aabb b;
  b.min = vec(t->bmin[3 * i], t->bmin[3 * i + 1], t->bmin[3 * i + 2]);
  b.max = vec(t->bmax[3 * i], t->bmax[3 * i + 1], t->bmax[3 * i + 2]);
  return b;
}

static const float scene_epsilon = 0.1f;                /* ray.fut:3 */

/* objs_hit (ray.fut:76-86) with bvh_fold (bvh.fut:61-84) inlined as the literal
 * parent-pointer walk: state (acc=(j,tmax), cur, prev). */
static inline int objs_hit(const orc_bvh *bvh, ray r, float t_min, float t_max, hit *out, orc_counters *cnt) {
  int32_t j = -1;
  float tbest = t_max;
  int32_t cur = 0, prev = PTR_INNER(-1);
  uint64_t steps = 0;
  cnt->rays++;
  while (cur != -1) {
    steps++;
    int32_t nl = bvh->left[cur], nr = bvh->right[cur];
    int from_left = prev == nl, from_right = prev == nr;
    int rec = 0;
    int32_t ptr = 0;
    if (from_left) { rec = 1; ptr = nr; }
    else if (!from_right) {
      cnt->box_tests++;
      if (aabb_hit(node_aabb(bvh, cur), r, t_min, t_max)) { rec = 1; ptr = nl; }   /* contains closes over the OUTER t_max (ray.fut:77) */
    }
    if (!rec) {
      prev = PTR_INNER(cur);
      cur = bvh->parent[cur];
    } else if (!PTR_IS_LEAF(ptr)) {
      prev = PTR_INNER(cur);
      cur = ptr;
    } else {
      int32_t i = PTR_LEAF_IDX(ptr);
      hit h;
      cnt->leaf_tests++;
      if (sphere_hit(&bvh->L[i], r, scene_epsilon, tbest, &h)) { j = i; tbest = h.t; }  /* closest_hit ray.fut:78-81 */
      prev = ptr;
    }
  }
  cnt->steps += steps;
  if (steps > cnt->max_steps) cnt->max_steps = steps;
  if (j >= 0) return sphere_hit(&bvh->L[j], r, t_min, tbest + 1.0f, out);              /* ray.fut:83-85 */
  return 0;
}

/* ------------------------------------------------------------------ ray.fut: camera, scatter, colour */

/* ray.fut:93-107 */
void orc_camera_make(orc_camera *cam, vec3 lookfrom, vec3 lookat, vec3 vup, float vfov, float aspect) {
  const float f32_pi = 3.14159265358979323846f;   /* f32.pi */
  float theta = vfov * f32_pi / 180.0f;
  float half_height = tanf(theta / 2.0f);
  float half_width = aspect * half_height;
  vec3 origin = lookfrom;
  vec3 w = normalise(vec_sub(lookfrom, lookat));
  vec3 u = normalise(cross(vup, w));
  vec3 v = cross(w, u);
  cam->origin = lookfrom;
  cam->llc = vec_sub(vec_sub(vec_sub(origin, scale(half_width, u)), scale(half_height, v)), w);
  cam->horizontal = scale(2.0f * half_width, u);
  cam->vertical = scale(2.0f * half_height, v);
}

/* prepare_scene's camera (ray.fut:243-244): vup = (0,1,0), aspect = f32 w / f32 h. */
void orc_scene_camera(const orc_scene *sc, int64_t h, int64_t w, orc_camera *cam) {
  orc_camera_make(cam, sc->look_from, sc->look_at, vec(0.0f, 1.0f, 0.0f), sc->fov, (float)w / (float)h);
}

static inline ray get_ray(const orc_camera *cam, float s, float t) {   /* ray.fut:109-114 */
  ray r;
  r.origin = cam->origin;
  r.dir = vec_sub(vec_add(vec_add(cam->llc, scale(s, cam->horizontal)), scale(t, cam->vertical)), cam->origin);
  return r;
}

static inline vec3 reflect(vec3 v, vec3 n) { return vec_sub(v, scale(2.0f * dot(v, n), n)); }  /* ray.fut:116-117 */

/* ray.fut:119-124 */
static inline int scatter(ray r, const hit *h, ray *scattered, vec3 *attenuation) {
  vec3 reflected = reflect(normalise(r.dir), h->normal);
  scattered->origin = h->p;
  scattered->dir = reflected;
  if (dot(scattered->dir, h->normal) > 0.0f) { *attenuation = h->colour; return 1; }
  return 0;
}

/* ray.fut:126-148 */
static inline vec3 ray_colour(const orc_bvh *objs, ray r, int32_t max_depth, orc_counters *cnt) {
  int32_t depth = 0;
  vec3 light = vec(1.0f, 1.0f, 1.0f), colour = vec(0.0f, 0.0f, 0.0f);
  while (depth < max_depth) {
    hit h;
    if (objs_hit(objs, r, 0.000f, 1000000000.0f, &h, cnt)) {
      ray scattered;
      vec3 attenuation;
      if (scatter(r, &h, &scattered, &attenuation)) {
        r = scattered;
        depth = depth + 1;
        colour = vec_mul(light, colour);     /* uses the OLD light, as the tuple update does */
        light = vec_mul(light, attenuation);
      } else {
        depth = max_depth;
        colour = vec_mul(light, colour);
      }
    } else {
      vec3 unit_dir = normalise(r.dir);
      float t = 0.5f * (unit_dir.y + 1.0f);
      vec3 bg = vec(0.5f, 0.7f, 1.0f);
      depth = max_depth;
      colour = vec_mul(light, vec_add(scale(1.0f - t, vec(1.0f, 1.0f, 1.0f)), scale(t, bg)));
    }
  }
  return colour;
}

/* ray.fut:156-162 */
static inline int32_t colour_to_pixel(vec3 p) {
  int32_t ir = (int32_t)(255.99f * p.x);
  int32_t ig = (int32_t)(255.99f * p.y);
  int32_t ib = (int32_t)(255.99f * p.z);
  return (ir << 16) | (ig << 8) | ib;
}

/* render_image (ray.fut:166-169) + trace_ray (ray.fut:150-154), restricted to rows
 * [row_begin, row_end) of the height x width image; out is the packed band
 * (row_end-row_begin) x width.  max_depth is 50 in the reference (ray.fut:154).
 * threads <= 0: all OpenMP threads (dynamic schedule over rows). */
int orc_render_rows(const orc_bvh *objs, const orc_camera *cam, int64_t width, int64_t height,
                    int64_t row_begin, int64_t row_end, int32_t max_depth, int threads,
                    int32_t *out, orc_counters *counters) {
  orc_counters total;
  memset(&total, 0, sizeof total);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  else omp_set_num_threads(omp_get_num_procs());
#else
  (void)threads;
#endif
  /* rows are cut into column chunks so that a many-core host (the GPU box has 256 hardware
   * threads) still sees thousands of tasks; dynamic schedule because work per pixel varies a lot */
  const int64_t chunk = width >= 256 ? 128 : width;
  const int64_t nchunk = (width + chunk - 1) / chunk;
  const int64_t ntask = (row_end - row_begin) * nchunk;
#pragma omp parallel
  {
    orc_counters c;
    memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 1)
    for (int64_t task = 0; task < ntask; task++) {
      const int64_t j = row_begin + task / nchunk;
      const int64_t i0 = (task % nchunk) * chunk;
      const int64_t i1 = i0 + chunk < width ? i0 + chunk : width;
      for (int64_t i = i0; i < i1; i++) {
        float u = (float)i / (float)width;
        float v = (float)(height - j) / (float)height;     /* pixel j i -> trace_ray ... (height-j) i */
        ray r = get_ray(cam, u, v);
        out[(j - row_begin) * width + i] = colour_to_pixel(ray_colour(objs, r, max_depth, &c));
      }
    }
#pragma omp critical
    {
      total.rays += c.rays; total.steps += c.steps;
      total.box_tests += c.box_tests; total.leaf_tests += c.leaf_tests;
      if (c.max_steps > total.max_steps) total.max_steps = c.max_steps;
    }
  }
  if (counters) *counters = total;
  return 0;
}

int orc_render(const orc_bvh *objs, const orc_camera *cam, int64_t width, int64_t height,
               int32_t max_depth, int threads, int32_t *out, orc_counters *counters) {
  return orc_render_rows(objs, cam, width, height, 0, height, max_depth, threads, out, counters);
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

/* Convenience checksum used by SURVEY.md 8(c): c = c*31 + pixel (u32 wrap). */
uint32_t orc_checksum(const int32_t *px, int64_t n) {
  uint32_t c = 0;
  for (int64_t i = 0; i < n; i++) c = c * 31u + (uint32_t)px[i];
  return c;
}
