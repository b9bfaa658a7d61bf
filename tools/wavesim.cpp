// wavesim.cpp -- host-side DESIGN TOOL (not a product path, not the oracle).
//
// Executes the persistent kernel's wave-level state machine (render_kernels.hip,
// persistent_kernel) with 64 emulated lanes per wave, using the very same per-lane code
// (lane_core.h) and the product's host BVH builder.  It answers, without a GPU:
//   * does the scheduling logic terminate and produce the same pixels as the simple
//     per-pixel traversal (checksum printed; tests compare it with the oracle's);
//   * how many wave-level BOX / LEAF / SHADE phase executions a frame needs under a given
//     voting policy, and the SIMT lane efficiency of each phase.
//
//   build/wavesim <rgbbox|irreg|floor:n:k> <h> <w> [thr_shade thr_leaf lmax nwaves policy]
#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lane_core.h"
#include "rt_host.hpp"

using namespace rtk;

struct F4 { float x, y, z, w; };

struct SceneData {
  std::vector<rt::TravNode> nodes;
  std::vector<F4> sph, col;
  Cam cam;
  int w, h, tiles_x, nchunks, max_depth;
  unsigned long long *depth_hist = nullptr;   // [64] pixels by number of scatters
};

struct Lane {
  Ray r{};
  float lr = 1, lg = 1, lb = 1;
  int depth = 0, pix = -1;
  float best = kTMax;
  int bestj = -1, cur = -1, sp = 0, nl = 0;
  int stack[64];
  int leaf[64];
};

struct Counters {
  unsigned long long ops[3] = {0, 0, 0};        // wave-level phase executions
  unsigned long long lanes[3] = {0, 0, 0};      // participating lanes
  unsigned long long rays = 0, box = 0, sphere = 0;
  unsigned long long fetches = 0;
  int max_sp = 0, max_nl = 0;
};

static uint32_t checksum(const std::vector<int32_t> &px) {
  uint32_t c = 0;
  for (int32_t p : px) c = c * 31u + (uint32_t)p;
  return c;
}

// reference for the simulator itself: per-pixel depth-first traversal (pixel_kernel's loop)
static void render_simple(const SceneData &S, std::vector<int32_t> &out, Counters &C) {
  unsigned long long rays = 0, box = 0, sph = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays, box, sph)
  for (int row = 0; row < S.h; ++row)
    for (int col = 0; col < S.w; ++col) {
      Ray r = primary_ray(S.cam, col, row, S.w, S.h);
      float lr = 1, lg = 1, lb = 1;
      int depth = 0;
      int32_t pixel = 0;
      for (;;) {
        float best = kTMax;
        int bestj = -1;
        int stack[64], sp = 0;
        stack[sp++] = 0;
        rays++;
        while (sp > 0) {
          const rt::TravNode &nd = S.nodes[stack[--sp]];
          box++;
          if (!box_hit(r, nd.lo[0], nd.lo[1], nd.lo[2], nd.hi[0], nd.hi[1], nd.hi[2])) continue;
          const int kids[2] = {nd.left, nd.right};
          for (int c : kids) {
            if (c < 0) {
              const F4 &s = S.sph[~c];
              sph++;
              closest_update(sphere_root(r, s.x, s.y, s.z, s.w), ~c, best, bestj);
            } else
              stack[sp++] = c;
          }
        }
        F4 s{0, 0, 0, 1}, c{0, 0, 0, 0};
        if (bestj >= 0) { s = S.sph[bestj]; c = S.col[bestj]; }
        if (!finish_ray(r, best, bestj, s.x, s.y, s.z, s.w, c.x, c.y, c.z, c.w, lr, lg, lb, depth, S.max_depth, &pixel)) break;
      }
      out[(size_t)row * S.w + col] = pixel;
      if (S.depth_hist) {
#pragma omp atomic
        S.depth_hist[depth < 63 ? depth : 63]++;
      }
    }
  C.rays = rays; C.box = box; C.sphere = sph;
}

struct Policy {
  int thr_shade = 24, thr_leaf = 24, lmax = 8, kind = 0;
};

// One emulated wave; `next_ticket` is the shared tile queue (the simulator runs waves
// round-robin one phase at a time, so tickets are handed out in simulated-time order).
struct Wave {
  Lane lane[64];
  unsigned q_next = 0, q_end = 0;
  bool exhausted = false, done = false;
};

static bool wave_step(Wave &W, const SceneData &S, const Policy &P, unsigned &next_ticket, std::vector<int32_t> &out,
                      Counters &C) {
  bool can_box[64], can_leaf[64], idle[64], want_shade[64];
  int nb = 0, nlv = 0, ns = 0;
  for (int l = 0; l < 64; ++l) {
    Lane &L = W.lane[l];
    const bool has_node = (L.cur >= 0) || (L.sp > 0);
    can_box[l] = has_node && (L.nl <= P.lmax - 2);
    can_leaf[l] = L.nl > 0;
    idle[l] = !has_node && L.nl == 0;
    want_shade[l] = idle[l] && (L.pix >= 0 || !W.exhausted);
    nb += can_box[l]; nlv += can_leaf[l]; ns += want_shade[l];
  }
  if (nb + nlv + ns == 0) { W.done = true; return false; }
  int op;
  if (P.kind == 0) {
    if (ns >= P.thr_shade || (nb == 0 && nlv == 0)) op = 2;
    else if (nlv >= P.thr_leaf || nb == 0) op = 1;
    else op = 0;
  } else {
    // plurality vote weighted by nothing: the phase with the most ready lanes
    op = 0;
    int bestn = nb;
    if (nlv > bestn) { op = 1; bestn = nlv; }
    if (ns > bestn) { op = 2; bestn = ns; }
  }
  C.ops[op]++;
  if (op == 0) {
    for (int l = 0; l < 64; ++l) {
      if (!can_box[l]) continue;
      Lane &L = W.lane[l];
      C.lanes[0]++;
      int ni = L.cur;
      if (ni < 0) ni = L.stack[--L.sp];
      const rt::TravNode &nd = S.nodes[ni];
      C.box++;
      int next = -1;
      if (box_hit(L.r, nd.lo[0], nd.lo[1], nd.lo[2], nd.hi[0], nd.hi[1], nd.hi[2])) {
        const int cl = nd.left, cr = nd.right;
        if (cl < 0) L.leaf[L.nl++] = ~cl; else next = cl;
        if (cr < 0) L.leaf[L.nl++] = ~cr;
        else if (next < 0) next = cr;
        else L.stack[L.sp++] = cr;
      }
      L.cur = next;
      C.max_sp = std::max(C.max_sp, L.sp);
      C.max_nl = std::max(C.max_nl, L.nl);
    }
  } else if (op == 1) {
    for (int l = 0; l < 64; ++l) {
      if (!can_leaf[l]) continue;
      Lane &L = W.lane[l];
      C.lanes[1]++;
      const int j = L.leaf[--L.nl];
      const F4 &s = S.sph[j];
      C.sphere++;
      closest_update(sphere_root(L.r, s.x, s.y, s.z, s.w), j, L.best, L.bestj);
    }
  } else {
    bool want[64];
    int slot[64];
    for (int l = 0; l < 64; ++l) {
      Lane &L = W.lane[l];
      slot[l] = -1;
      if (want_shade[l]) C.lanes[2]++;
      if (idle[l] && L.pix >= 0) {
        F4 s{0, 0, 0, 1}, c{0, 0, 0, 0};
        if (L.bestj >= 0) { s = S.sph[L.bestj]; c = S.col[L.bestj]; }
        int32_t pixel;
        if (finish_ray(L.r, L.best, L.bestj, s.x, s.y, s.z, s.w, c.x, c.y, c.z, c.w, L.lr, L.lg, L.lb, L.depth, S.max_depth,
                       &pixel)) {
          L.cur = 0; L.best = kTMax; L.bestj = -1;
          C.rays++;
        } else {
          out[L.pix] = pixel;
          L.pix = -1;
        }
      }
      want[l] = idle[l] && L.pix < 0 && !W.exhausted;
    }
    for (;;) {
      int cnt = 0;
      for (int l = 0; l < 64; ++l) cnt += want[l];
      if (cnt == 0) break;
      if (W.q_next == W.q_end) {
        const unsigned t = next_ticket++;
        C.fetches++;
        if (t >= (unsigned)S.nchunks) { W.exhausted = true; break; }
        W.q_next = t * 64u;
        W.q_end = W.q_next + 64u;
      }
      const unsigned avail = W.q_end - W.q_next;
      unsigned rank = 0;
      for (int l = 0; l < 64; ++l) {
        if (!want[l]) continue;
        if (rank < avail) {
          const unsigned sidx = W.q_next + rank;
          const int tile = (int)(sidx >> 6), within = (int)(sidx & 63u);
          const int tx = tile % S.tiles_x, ty = tile / S.tiles_x;
          const int col = tx * 8 + (within & 7), row = ty * 8 + (within >> 3);
          if (col < S.w && row < S.h) { slot[l] = row * S.w + col; want[l] = false; }
        }
        rank++;
      }
      W.q_next += std::min<unsigned>((unsigned)cnt, avail);
    }
    for (int l = 0; l < 64; ++l) {
      if (slot[l] < 0) continue;
      Lane &L = W.lane[l];
      const int row = slot[l] / S.w, col = slot[l] - row * S.w;
      L.r = primary_ray(S.cam, col, row, S.w, S.h);
      L.lr = L.lg = L.lb = 1.0f;
      L.depth = 0; L.pix = slot[l]; L.cur = 0; L.best = kTMax; L.bestj = -1;
      C.rays++;
    }
  }
  return true;
}


// ---------------------------------------------------------------------------------------
// Pooled design: the wave owns 64 ray slots and two LDS work lists shared by all lanes --
// a LIFO stack of (slot, inner node) items and a list of (slot, leaf) items.  Any lane
// processes any item, so SIMT efficiency no longer depends on per-ray traversal lengths.
// Per-slot completion is tracked by an outstanding-item counter; the closest hit by a
// 64-bit min over (t bits << 32 | leaf index).
// ---------------------------------------------------------------------------------------
struct Slot {
  Ray r{};
  float lr = 1, lg = 1, lb = 1;
  int depth = 0, pix = -1;
  bool active = false;
  unsigned long long key = 0;
  int cnt = 0;
};
constexpr unsigned long long kKeyInit = ((unsigned long long)0x4e6e6b28u << 32) | 0xffffffffu;  // (1e9, none)

constexpr int kMaxSlots = 256;
static int g_nslots = 64;   // ray slots per wave (the kernel has 64; what would more buy?)
struct PWave {
  Slot slot[kMaxSlots];
  std::vector<unsigned> box, leaf;
  unsigned q_next = 0, q_end = 0;
  bool exhausted = false, done = false;
};

struct PCounters {
  unsigned long long ops[3] = {0, 0, 0}, lanes[3] = {0, 0, 0};
  unsigned long long rays = 0, box = 0, sphere = 0, fetches = 0;
  size_t max_box = 0, max_leaf = 0;
};

static inline unsigned f2u(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }

static bool pwave_step(PWave &W, const SceneData &S, int width, int thr_shade, unsigned &next_ticket,
                       std::vector<int32_t> &out, PCounters &C) {
  const size_t nbox = W.box.size(), nleaf = W.leaf.size();
  int op;
  if (nbox >= (size_t)width) op = 0;
  else if (nleaf >= (size_t)width) op = 1;
  else {
    int ndone = 0, nfree = 0;
    for (int l = 0; l < g_nslots; ++l) {
      auto &s = W.slot[l];
      if (s.active && s.cnt == 0) ndone++;
      if (!s.active && !W.exhausted) nfree++;
    }
    if (ndone + nfree >= thr_shade || (nbox == 0 && nleaf == 0)) {
      if (ndone + nfree == 0) { W.done = true; return false; }
      op = 2;
    } else if (nbox > 0) op = 0;
    else op = 1;
  }
  C.ops[op]++;
  if (op == 0) {
    const size_t n = std::min(nbox, (size_t)width);
    std::vector<unsigned> items(W.box.end() - n, W.box.end());
    W.box.resize(nbox - n);
    C.lanes[0] += n;
    for (size_t k = 0; k < n; ++k) {
      const unsigned it = items[n - 1 - k];
      const int sl = it >> 24, ni = it & 0xffffff;
      Slot &s = W.slot[sl];
      const rt::TravNode &nd = S.nodes[ni];
      C.box++;
      if (box_hit(s.r, nd.lo[0], nd.lo[1], nd.lo[2], nd.hi[0], nd.hi[1], nd.hi[2])) {
        const int kids[2] = {nd.left, nd.right};
        for (int c : kids) {
          if (c < 0) W.leaf.push_back(((unsigned)sl << 24) | (unsigned)~c);
          else W.box.push_back(((unsigned)sl << 24) | (unsigned)c);
        }
        s.cnt += 1;
      } else
        s.cnt -= 1;
    }
    C.max_box = std::max(C.max_box, W.box.size());
    C.max_leaf = std::max(C.max_leaf, W.leaf.size());
  } else if (op == 1) {
    const size_t n = std::min(nleaf, (size_t)width);
    C.lanes[1] += n;
    for (size_t k = 0; k < n; ++k) {
      const unsigned it = W.leaf.back();
      W.leaf.pop_back();
      const int sl = it >> 24, j = it & 0xffffff;
      Slot &s = W.slot[sl];
      const F4 &sp = S.sph[j];
      C.sphere++;
      const float g = sphere_root(s.r, sp.x, sp.y, sp.z, sp.w);
      if (g < kTMax) {
        const unsigned long long key = ((unsigned long long)f2u(g) << 32) | (unsigned)j;
        if (key < s.key) s.key = key;
      }
      s.cnt -= 1;
    }
  } else {
    for (int l = 0; l < g_nslots; ++l) {
      Slot &s = W.slot[l];
      if (s.active && s.cnt == 0) {
        C.lanes[2]++;
        const float best = u2f((unsigned)(s.key >> 32));
        const int bestj = (s.key == kKeyInit) ? -1 : (int)(unsigned)(s.key & 0xffffffffu);
        F4 sp{0, 0, 0, 1}, c{0, 0, 0, 0};
        if (bestj >= 0) { sp = S.sph[bestj]; c = S.col[bestj]; }
        int32_t pixel;
        if (finish_ray(s.r, best, bestj, sp.x, sp.y, sp.z, sp.w, c.x, c.y, c.z, c.w, s.lr, s.lg, s.lb, s.depth, S.max_depth, &pixel)) {
          s.key = kKeyInit; s.cnt = 1;
          W.box.push_back(((unsigned)l << 24) | 0u);
          C.rays++;
        } else {
          out[s.pix] = pixel;
          s.active = false; s.pix = -1;
        }
      }
    }
    for (int l = 0; l < g_nslots && !W.exhausted; ++l) {
      Slot &s = W.slot[l];
      if (s.active) continue;
      for (;;) {
        if (W.q_next == W.q_end) {
          const unsigned t = next_ticket++;
          C.fetches++;
          if (t >= (unsigned)S.nchunks) { W.exhausted = true; break; }
          W.q_next = t * 64u; W.q_end = W.q_next + 64u;
        }
        const unsigned sidx = W.q_next++;
        const int tile = (int)(sidx >> 6), within = (int)(sidx & 63u);
        const int tx = tile % S.tiles_x, ty = tile / S.tiles_x;
        const int col = tx * 8 + (within & 7), row = ty * 8 + (within >> 3);
        if (col < S.w && row < S.h) {
          C.lanes[2]++;
          s.r = primary_ray(S.cam, col, row, S.w, S.h);
          s.lr = s.lg = s.lb = 1.0f; s.depth = 0; s.pix = row * S.w + col; s.active = true;
          s.key = kKeyInit; s.cnt = 1;
          W.box.push_back(((unsigned)l << 24) | 0u);
          C.rays++;
          break;
        }
      }
    }
  }
  return true;
}

static int run_pooled(const SceneData &S, const std::vector<int32_t> &ref, int nwaves, int width, int thr_shade) {
  std::vector<int32_t> out(ref.size(), -1);
  std::vector<PWave> waves(nwaves);
  PCounters C;
  unsigned next_ticket = 0;
  bool any = true;
  unsigned long long rounds = 0;
  while (any) {
    any = false;
    for (auto &W : waves)
      if (!W.done) any |= pwave_step(W, S, width, thr_shade, next_ticket, out, C);
    rounds++;
  }
  size_t diff = 0;
  for (size_t i = 0; i < ref.size(); ++i) diff += ref[i] != out[i];
  std::printf("pooled(width %d, thr_shade %d, %d waves): checksum %08x diff_vs_simple %zu rays %llu box %llu sphere %llu fetches %llu\n",
              width, thr_shade, nwaves, checksum(out), diff, C.rays, C.box, C.sphere, C.fetches);
  const char *names[3] = {"BOX", "LEAF", "SHADE"};
  for (int i = 0; i < 3; ++i)
    std::printf("  %-5s wave-ops %10llu lanes %12llu efficiency %.3f\n", names[i], C.ops[i], C.lanes[i],
                C.ops[i] ? (double)C.lanes[i] / ((i == 2 ? (double)g_nslots : (double)width) * C.ops[i]) : 0.0);
  std::printf("  rounds (longest wave, in phases) %llu  max box stack %zu  max leaf list %zu\n", rounds, C.max_box, C.max_leaf);
  return diff ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <rgbbox|irreg|floor:n:k> h w [thr_shade thr_leaf lmax nwaves policy max_depth]\n", argv[0]);
    return 2;
  }
  const std::string name = argv[1];
  const int h = std::atoi(argv[2]), w = std::atoi(argv[3]);
  Policy P;
  if (argc > 4) P.thr_shade = std::atoi(argv[4]);
  if (argc > 5) P.thr_leaf = std::atoi(argv[5]);
  if (argc > 6) P.lmax = std::atoi(argv[6]);
  const int nwaves = argc > 7 ? std::atoi(argv[7]) : 4096;
  if (argc > 8) P.kind = std::atoi(argv[8]);
  const int max_depth = argc > 9 ? std::atoi(argv[9]) : 50;

  rt::SceneDesc sc;
  if (name == "rgbbox") sc = rt::make_rgbbox();
  else if (name == "irreg") sc = rt::make_floor(100, 600.0f);
  else if (name.rfind("floor:", 0) == 0) {
    int n = 0; float k = 0;
    if (std::sscanf(name.c_str(), "floor:%d:%f", &n, &k) != 2) return 2;
    sc = rt::make_floor(n, k);
  } else return 2;
  const rt::Lbvh bvh = rt::build_lbvh(sc.spheres);
  const rt::TravLayout tl = rt::make_trav_layout(bvh);
  const rt::Camera cam = rt::scene_camera(sc, h, w);

  SceneData S;
  S.nodes = tl.nodes;
  S.sph.resize(bvh.n); S.col.resize(bvh.n);
  std::memcpy(S.sph.data(), tl.sph.data(), sizeof(F4) * bvh.n);
  std::memcpy(S.col.data(), tl.col.data(), sizeof(F4) * bvh.n);
  std::memcpy(&S.cam, &cam, sizeof cam);
  S.w = w; S.h = h; S.tiles_x = (w + 7) / 8; S.nchunks = S.tiles_x * ((h + 7) / 8);
  S.max_depth = max_depth;

  std::vector<int32_t> ref((size_t)h * w, -1), out((size_t)h * w, -1);
  Counters C0, C;
  unsigned long long dh[64] = {0};
  S.depth_hist = dh;
  render_simple(S, ref, C0);
  S.depth_hist = nullptr;
  std::printf("pixels by bounce count (scatters):");
  for (int d = 0; d < 64; ++d) if (dh[d]) std::printf(" %d:%llu", d, dh[d]);
  std::printf("\n");
  std::printf("scene %s %dx%d spheres %lld height %d sweeps %d\n", name.c_str(), h, w, (long long)bvh.n, tl.height, bvh.sweeps);
  std::printf("simple: checksum %08x rays %llu box %llu sphere %llu\n", checksum(ref), C0.rays, C0.box, C0.sphere);

  if (P.kind >= 10) {
    // pooled design: kind 10 -> 64 items per op, kind 11 -> 128 items per op
    if (P.lmax >= 64) g_nslots = P.lmax;   // reuse the lmax argument: slots per wave
    return run_pooled(S, ref, nwaves, P.kind == 11 ? 128 : 64, P.thr_shade);
  }
  std::vector<Wave> waves(nwaves);
  unsigned next_ticket = 0;
  bool any = true;
  unsigned long long rounds = 0;
  while (any) {
    any = false;
    for (auto &W : waves)
      if (!W.done) any |= wave_step(W, S, P, next_ticket, out, C);
    rounds++;
  }
  size_t diff = 0;
  for (size_t i = 0; i < ref.size(); ++i) diff += ref[i] != out[i];
  std::printf("persistent: checksum %08x diff_vs_simple %zu rays %llu box %llu sphere %llu fetches %llu (expected %d)\n",
              checksum(out), diff, C.rays, C.box, C.sphere, C.fetches, S.nchunks + nwaves);
  const char *names[3] = {"BOX", "LEAF", "SHADE"};
  for (int i = 0; i < 3; ++i)
    std::printf("  %-5s wave-ops %10llu lanes %12llu efficiency %.3f\n", names[i], C.ops[i], C.lanes[i],
                C.ops[i] ? (double)C.lanes[i] / (64.0 * C.ops[i]) : 0.0);
  std::printf("  rounds (longest wave, in phases) %llu  max_sp %d max_nl %d\n", rounds, C.max_sp, C.max_nl);
  return diff ? 1 : 0;
}
