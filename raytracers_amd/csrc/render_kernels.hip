// render_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the render hot
// path: render_image / trace_ray / ray_colour / objs_hit / bvh_fold
// (futhark/ray.fut:126-169, :76-86; futhark/bvh.fut:61-84).
//
// Three kernel families, bit-identical output:
//   pixel_kernel       one thread per pixel, 8x8 pixel tile per wave, BVH read from
//                      HBM/L2 (no LDS staging of the scene)          -> BASELINE configs[1]
//   persistent_kernel  persistent waves pulling 8x8 tiles from a global ticket counter,
//                      finished lanes refilled in place (ballot + mbcnt prefix), per-wave
//                      phase voting {box, sphere, shade}, breadth-first BVH prefix and
//                      sphere table staged in LDS, per-lane traversal stack + deferred-leaf
//                      list in LDS
//   pooled_kernel      (the default) persistent waves whose 64 lanes share LDS work lists of
//                      (ray slot, node) items: any lane tests any ray's node; one launch renders
//                      one frame, one device's row tiles of it, or a batch of frames
//                                                                    -> BASELINE configs[2..4]
//
// No MFMA: there is no dense contraction on this path.  Build: -ffp-contract=off and
// the default correctly rounded fp32 divide/sqrt (parity is bit-exact).
#include <hip/hip_runtime.h>

#include <mutex>
#include <type_traits>

#include "lane_core.h"
#include "rt_device.hpp"
#include "treelet.h"

// tools/isa_budget.py compiles this file with -DRT_ISA_MARKS and counts the instructions between the marks (assembler
// comments: no instruction, no effect on the product build, where the macro is empty)
#ifdef RT_ISA_MARKS
#define RT_MARK(name) asm volatile("; RT_MARK " name)
#else
#define RT_MARK(name)
#endif

namespace rtk {

__device__ __forceinline__ int f2i(float f) { return __float_as_int(f); }

// 16-byte global load through a buffer resource (buffer_load_dwordx4, base in SGPRs, 32-bit
// byte offset per lane).  Used for the part of the scene that is NOT staged in LDS: an
// ordinary pointer load next to an LDS load gets if-converted by hipcc into one flat_load
// with a selected address, which is far slower than ds_read / buffer_load.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  const v4i v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

// local (packed) row of this part -> row of the full image, cyclic row tiles
__device__ __forceinline__ int global_row(const KParams &p, int lrow) {
  const int k = lrow / p.rows_per_tile;
  return (k * p.nparts + p.part) * p.rows_per_tile + (lrow - k * p.rows_per_tile);
}

// where a pixel of this part is stored: packed rows, or (out_skip != 0) its place in the full image
__device__ __forceinline__ size_t out_index(const KParams &p, int lrow, int col) {
  return (size_t)lrow * p.w + col + (size_t)(lrow / p.rows_per_tile) * (size_t)p.out_skip;
}

// ---------------------------------------------------------------------------------
// Family 1: one thread per pixel.
// ---------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(64) void pixel_kernel(KParams p) {
  // Traversal stack: [entry][lane] so a wave's accesses to one entry hit 64 distinct
  // dwords.  Depth bound: a Karras tree over 32-bit keys + 32-bit index tie-break has
  // height <= 64, and depth-first order keeps at most one pending sibling per level.
  __shared__ int stack[kStackPixel][64];
  const int lane = threadIdx.x;
  const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x;
  const int col = tx * 8 + (lane & 7), lrow = ty * 8 + (lane >> 3);
  if (col >= p.w || lrow >= p.rows_local) return;
  const __amdgpu_buffer_rsrc_t rs_nodes = make_rsrc(p.nodes, (unsigned)p.n_nodes * 32u);
  const __amdgpu_buffer_rsrc_t rs_sph = make_rsrc(p.sph, (unsigned)p.n_sph * 16u);
  Ray r = primary_ray(p.cam, col, global_row(p, lrow), p.w, p.h);
  float lr = 1.0f, lg = 1.0f, lb = 1.0f;
  int depth = 0;
  int32_t pixel = 0;
  unsigned long long n_rays = 0, n_box = 0, n_sph = 0;
  for (;;) {
    float best = kTMax;
    int bestj = -1;
    int sp = 0;
    stack[sp++][lane] = 0;
    if (STATS) n_rays++;
    while (sp > 0) {
      const int ni = stack[--sp][lane];
      const float4 lo = buf_load16(rs_nodes, ni * 32), hi = buf_load16(rs_nodes, ni * 32 + 16);
      const int kids[2] = {f2i(lo.w), f2i(hi.w)};
      asm volatile("" ::"v"(kids[0]), "v"(kids[1]));   // keep the child refs in the first load (no sunk re-load)
      if (STATS) n_box++;
      if (!box_hit(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z)) continue;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = kids[k];
        if (c < 0) {
          const int j = ~c;
          const float4 s = buf_load16(rs_sph, j * 16);
          if (STATS) n_sph++;
          closest_update(sphere_root(r, s.x, s.y, s.z, s.w), j, best, bestj);
        } else {
          stack[sp++][lane] = c;
        }
      }
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 1.f), c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bestj >= 0) {
      s = p.sph[bestj];
      c = p.col[bestj];
    }
    if (!finish_ray(r, best, bestj, s.x, s.y, s.z, s.w, c.x, c.y, c.z, c.w, lr, lg, lb, depth, p.max_depth, &pixel)) break;
  }
  p.out[out_index(p, lrow, col)] = pixel;
  if (STATS) {
    atomicAdd(&p.stats[0], n_rays);
    atomicAdd(&p.stats[1], n_box);
    atomicAdd(&p.stats[2], n_sph);
  }
}

// ---------------------------------------------------------------------------------
// Family 2: persistent waves.
// ---------------------------------------------------------------------------------
// ballot of a bool: the v_cmp result IS the 64-bit lane mask (HIP's __ballot(int) would go
// bool -> int -> compare again)
__device__ __forceinline__ unsigned long long bal(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// v_cndmask_b32 driven directly by a 64-bit lane mask held in SGPRs: bit set -> b, else a.
// (Written as asm because hipcc materialises `mask -> bool -> select` through VGPR 0/1 values.)
__device__ __forceinline__ int sel_mask(unsigned long long m, int a, int b) {
  int out;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(a), "v"(b), "s"(m));
  return out;
}

// 32-bit store to an LDS byte address
__device__ __forceinline__ void lds_store(int lds_byte_addr, unsigned v) {
  *reinterpret_cast<__attribute__((address_space(3))) unsigned *>((unsigned)lds_byte_addr) = v;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }   // wave-uniform value -> SGPR

__device__ __forceinline__ int lane_rank(unsigned long long m) {   // # set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ int lane_rank_from(unsigned long long m, int base) {   // base + lane_rank(m): mbcnt's own addend
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}

// ds_add_u32 executed by exactly the lanes of a 64-bit mask held in SGPRs (no VALU compare, no branch): the
// exec mask is narrowed around the one instruction and restored.
__device__ __forceinline__ void lds_add_masked(unsigned long long m, int lds_byte_addr, int v) {
  unsigned long long saved;
  asm volatile("s_and_saveexec_b64 %0, %1\n\tds_add_u32 %2, %3\n\ts_mov_b64 exec, %0"
               : "=&s"(saved)
               : "s"(m), "v"(lds_byte_addr), "v"(v)
               : "memory");
}

// A wave leaves the tile queue (it has failed once on every counter): the last one to leave zeroes the counters,
// so every launch starts from ticket 0 with no host-side bookkeeping.  (All other waves' draws precede their own
// arrival here; the resets are device-scope atomics like the draws.)
__device__ __forceinline__ void queue_leave(const KParams &p, unsigned nwaves, int lane) {
  if (lane == 0 && atomicAdd(&p.queue[kQueueExit], 1u) == nwaves - 1u) {
    for (int s = 0; s < kMaxShards; ++s) atomicExch(&p.queue[kQueueStride * s], 0u);
    atomicExch(&p.queue[kQueueExit], 0u);
  }
}

template <int THREADS, bool STATS>
__global__ __launch_bounds__(THREADS) void persistent_kernel(KParams p) {
  const int SMAX = p.smax, LMAX = p.lmax;
  extern __shared__ float4 smem[];
  float4 *const lnodes = smem;                          // [2 * lds_nodes]  breadth-first BVH prefix
  float4 *const lsph = smem + 2 * p.lds_nodes;          // [lds_sph]        {pos, radius}
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // per-wave scratch: stack[SMAX][64] then leaves[LMAX][64]; entry-major so that one
  // entry of all 64 lanes is 64 consecutive dwords (bank = lane, conflict-free)
  int *const wstack = reinterpret_cast<int *>(lsph + p.lds_sph) + wave * ((SMAX + 1 + LMAX) * 64) + lane;
  int *const wleaf = wstack + (SMAX + 1) * 64;

  const int sph_base = 2 * p.lds_nodes;
  const __amdgpu_buffer_rsrc_t rs_nodes = make_rsrc(p.nodes, (unsigned)p.n_nodes * 32u);
  const __amdgpu_buffer_rsrc_t rs_sph = make_rsrc(p.sph, (unsigned)p.n_sph * 16u);
  for (int i = threadIdx.x; i < 2 * p.lds_nodes; i += THREADS) lnodes[i] = p.nodes[i];
  for (int i = threadIdx.x; i < p.lds_sph; i += THREADS) lsph[i] = p.sph[i];
  __syncthreads();

  // ---- lane state ----
  Ray r = {};
  float lr = 1.0f, lg = 1.0f, lb = 1.0f;
  int depth = 0;
  int pix = -1;          // local pixel offset (lrow * w + col), or -1: no pixel
  float best = kTMax;
  int bestj = -1;
  int cur = -1;          // inner node held in a register (or -1)
  int sp = 0, nl = 0;    // stack / deferred-leaf counts
  unsigned long long n_rays = 0, n_box = 0, n_sph = 0;
  // ---- wave state (uniform) ----
  unsigned q_next = 0, q_end = 0;
  bool exhausted = false;

  for (;;) {
    const bool has_node = (cur >= 0) | (sp > 0);
    const bool can_box = has_node & (nl <= LMAX - 2);
    const bool can_leaf = nl > 0;
    const bool idle = !has_node & (nl == 0);
    const bool want_shade = idle & ((pix >= 0) | !exhausted);
    const unsigned long long mb = bal(can_box), ml = bal(can_leaf), ms = bal(want_shade);
    if ((mb | ml | ms) == 0ull) break;
    const int nb = __popcll(mb), nlv = __popcll(ml), ns = __popcll(ms);

    int op;   // 0 box, 1 leaf, 2 shade  (wave-uniform)
    if (ns >= p.thr_shade || (nb == 0 && nlv == 0)) op = 2;
    else if (nlv >= p.thr_leaf || nb == 0) op = 1;
    else op = 0;

    if (op == 0) {
      // ---- BOX: one inner node per lane ----
      // Written predicated rather than branchy: both float4 of the node are loaded and
      // consumed unconditionally (hipcc otherwise sinks the child-reference dwords into the
      // hit branch = a second dependent memory round trip), and the LDS / global choice is a
      // value merge of two separate loads (a pointer select would become flat_load).
      if (can_box) {
        const bool pop = cur < 0;
        sp -= pop ? 1 : 0;
        const int popped = wstack[sp * 64];          // slot sp exists (SMAX + 1 entries per lane)
        const int ni = pop ? popped : cur;
        const int li = ni < p.lds_nodes ? ni : 0;
        float4 lo = smem[2 * li], hi = smem[2 * li + 1];
        if (ni >= p.lds_nodes) {
          lo = buf_load16(rs_nodes, ni * 32);
          hi = buf_load16(rs_nodes, ni * 32 + 16);
        }
        const int cl = f2i(lo.w), cr = f2i(hi.w);
        if (STATS) n_box++;
        const bool hit = box_hit(r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
        const bool lfl = hit & (cl < 0), lfr = hit & (cr < 0);
        const bool inl = hit & (cl >= 0), inr = hit & (cr >= 0);
        if (lfl) wleaf[nl * 64] = ~cl;
        nl += lfl ? 1 : 0;
        if (lfr) wleaf[nl * 64] = ~cr;
        nl += lfr ? 1 : 0;
        if (inl & inr) wstack[sp * 64] = cr;
        sp += (inl & inr) ? 1 : 0;
        cur = inl ? cl : (inr ? cr : -1);
      }
    } else if (op == 1) {
      // ---- LEAF: one deferred sphere test per lane ----
      if (can_leaf) {
        const int j = wleaf[(--nl) * 64];
        float4 s = smem[sph_base + (j < p.lds_sph ? j : 0)];
        if (j >= p.lds_sph) s = buf_load16(rs_sph, j * 16);
        if (STATS) n_sph++;
        closest_update(sphere_root(r, s.x, s.y, s.z, s.w), j, best, bestj);
      }
    } else {
      // ---- SHADE: finish rays, then refill empty lanes from the tile queue ----
      if (idle & (pix >= 0)) {
        float4 s = make_float4(0.f, 0.f, 0.f, 1.f), c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bestj >= 0) {
          s = p.sph[bestj];
          c = p.col[bestj];
        }
        int32_t pixel;
        if (finish_ray(r, best, bestj, s.x, s.y, s.z, s.w, c.x, c.y, c.z, c.w, lr, lg, lb, depth, p.max_depth, &pixel)) {
          cur = 0;
          best = kTMax;
          bestj = -1;
          if (STATS) n_rays++;
        } else {
          const int lr_ = pix / p.w;
          p.out[out_index(p, lr_, pix - lr_ * p.w)] = pixel;
          pix = -1;
        }
      }
      bool want = idle & (pix < 0) & !exhausted;
      int slot = -1;                 // tile-queue slot assigned to this lane
      unsigned long long m = bal(want);
      while (m != 0ull) {            // wave-uniform loop
        if (q_next == q_end) {
          unsigned t = 0;
          if (lane == 0) t = atomicAdd(p.queue, 1u);
          t = __builtin_amdgcn_readfirstlane(t);
          if (t >= (unsigned)p.nchunks) {
            exhausted = true;
            break;
          }
          q_next = t * 64u;
          q_end = q_next + 64u;
        }
        const unsigned avail = q_end - q_next;
        const unsigned rank = (unsigned)lane_rank(m);
        const unsigned cnt = (unsigned)__popcll(m);
        if (want & (rank < avail)) {
          const unsigned sidx = q_next + rank;
          const int tile = (int)(sidx >> 6), within = (int)(sidx & 63u);
          const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
          const int col = tx * 8 + (within & 7), lrow = ty * 8 + (within >> 3);
          if (col < p.w && lrow < p.rows_local) {
            slot = lrow * p.w + col;
            want = false;
          }
        }
        q_next += (cnt < avail) ? cnt : avail;
        m = bal(want);
      }
      if (slot >= 0) {
        const int lrow = slot / p.w, col = slot - lrow * p.w;
        r = primary_ray(p.cam, col, global_row(p, lrow), p.w, p.h);
        lr = 1.0f; lg = 1.0f; lb = 1.0f;
        depth = 0;
        pix = slot;
        cur = 0;
        best = kTMax;
        bestj = -1;
        if (STATS) n_rays++;
      }
    }
  }
  queue_leave(p, gridDim.x * (THREADS / 64), lane);
  if (STATS) {
    atomicAdd(&p.stats[0], n_rays);
    atomicAdd(&p.stats[1], n_box);
    atomicAdd(&p.stats[2], n_sph);
  }
}

// ---------------------------------------------------------------------------------
// Family 3: pooled work items.
//
// A wave owns 64 ray SLOTS and two work lists in LDS that all lanes share: a LIFO stack of
// (slot, inner node) items and a list of (slot, leaf) items.  ANY lane processes ANY item, so a
// traversal step is one dense wave-wide operation whatever the per-ray traversal lengths are,
// and one ray's subtree is searched by many lanes at once (the fixed (0, 1e9) box interval makes
// the box tests of a ray independent of its hits, so any order gives the fold's result).
// Children are appended with ballot + mbcnt prefix sums.  Per slot, a counter of outstanding
// INNER-node items (LDS atomic add) tells when the box part of the fold is complete; the leaf
// list is always drained before folds are finished, so `counter == 0` then means the whole fold
// is.  The closest hit is an LDS 64-bit atomic min over (bits(t) << 32 | leaf index): smallest
// t, ties to the lowest leaf index -- exactly closest_hit's accumulator (ray.fut:78-81).
//
// The LDS pipeline is the resource this kernel saturates (tools/issue_peak.hip: a ds_bpermute
// costs ~3x a ds_read_b32, a ds_read_b128 of 64-byte records at random indices ~6x a
// conflict-free one, same-address LDS atomics serialise), so the layout is built around it:
//   * the staged node records are split into four PLANES of 16-byte quarters (record r's
//     quarter k at plane k + 16 r): the 16 lanes one ds_read_b128 cycle serves then spread over
//     16 bank groups instead of the 4 a 64-byte stride gives;
//   * a slot's ray lives in an LDS ray table (three float4 per slot: {o, a} {1/d} {d}) written
//     once per ray by the owning lane; an item reads its ray with two ds_read_b128 (neighbouring
//     items mostly share the slot = broadcast) instead of six or seven ds_bpermute;
//   * one append per child (box stack, leaf list or a dump dword), and the per-slot counter is
//     touched only by items whose number of inner children differs from one.
//
// Bound on the box stack (H = tree height): each operation pops the <= 64 newest items and
// pushes their <= 128 children, whose depth exceeds their parents'; remainders of at most
// 64 items per "generation" with strictly increasing depth labels => size <= 64*H + 128.
// The leaf list is drained first whenever it holds >= 64 items => size <= 63 + 128.
// ---------------------------------------------------------------------------------
constexpr int kOrderClasses = 8;   // cost classes of the adaptive tile order (tile_order_kernel)
constexpr unsigned long long kKeyInit = ((unsigned long long)0x4e6e6b28u << 32) | 0xffffffffull;   // (1e9, no leaf)

__device__ __forceinline__ float pull(int lane_byte, float v) {   // v of the lane at byte address lane_byte (ds_bpermute)
  return __int_as_float(__builtin_amdgcn_ds_bpermute(lane_byte, __float_as_int(v)));
}

// ---------------------------------------------------------------------------------
// SOLO: one ray, the whole wave.
//
// A frame cannot end before its longest bounce chain does (irreg 1000x1000: 10 pixels of 45-49 scatters, each scatter a
// fold that depends on the one before), and in the pooled loop a wave that serves ONE ray still pays every operation's
// general machinery: ~5 traversal operations + LEAF + SHADE per bounce at 1700-3900 shader cycles each = ~7 us per bounce
// (profiles/r03/exp/e10_trace.txt).  solo_trace finishes ONE pixel with the whole wave behind it -- called from the SOLO
// instantiation's prologue for single-pixel tickets (the deepest tiles of an ordered view, all of them first tickets) and, in
// the COLD instantiation, from inside the loop for the last rays of a wave that cannot refill, and in the DONATE instantiation behind
// the loop for rays given away by sibling waves.  The ray lives in registers (the
// same value in every lane), the fold is
// a loop of TREELET operations -- 2^D lanes per popped treelet root, every lane tests the boxes of both children of one
// node of the treelet speculatively, a node counts iff the boxes on its path inside the treelet passed (treelet.h: the
// fixed (0, 1e9) box interval makes a box test independent of when it is made) -- and of sphere tests folded with one LDS
// atomic min on the key format of the pooled loop; then the same shade_ray.  Same arithmetic, same (t, lowest leaf)
// winner: bit-identical pixels.  A separate, never-inlined function: its registers are allocated on their own and the
// pooled loop's are not disturbed (the pooled kernel measurably slows down when code is added to its loop).
// Uses the wave's LDS region (hit key 0, box stack, leaf list, dump dword), which is idle when it is called.
// ---------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) KParams *KParamsArg;   // the kernel's own argument block (scalar loads)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_load16(unsigned a) {
  const v4f v = *reinterpret_cast<const __attribute__((address_space(3))) v4f *>(a);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned lds_load4(unsigned a) { return *reinterpret_cast<const __attribute__((address_space(3))) unsigned *>(a); }

// (CULL exists here and is NOT used, kSoloCull: a lone ray's box items are all expanded before its first sphere test -- the leaves sit at
// the bottom of the tree, and the leaf list of one ray never reaches a batch of 64 -- so its best root is still 1e9 while the boxes
// are walked: the limit never bites and costs its instructions on the frame's critical path.  Measured with it on, round 6:
// irreg 1000 x 1000 one frame at a time 0.249 -> 0.256 ms, 500 x 500 0.218 -> 0.226; profiles/r06/exp/e1_cull_ab.txt.)
constexpr bool kSoloCull = false;
// TRACE (the instrumented launch, rt_render_trace with option trace_solo): shader cycles inside treelet operations, sphere-test operations and the
// rest of a bounce (ray_derive, root box, winner's loads, shade), with their counts, added to words 13 .. 15 of the wave's trace record.
template <bool CULL, bool TRACE = false>
__device__ __attribute__((noinline)) void solo_trace(KParamsArg pp_v, unsigned smem_lds, unsigned wbase_lds, float ox, float oy, float oz,
                                                     float dx, float dy, float dz, float lr, float lg, float lb, int pix, int depth,
                                                     int ptile) {
  // (arguments arrive in vector registers: say that they are wave-uniform, or every load through `pp` is a vector load and
  // every buffer_load a waterfall loop over its descriptor)
  const unsigned long long pp_bits = (unsigned long long)pp_v;
  KParamsArg pp = (KParamsArg)(((unsigned long long)(unsigned)uni((int)(pp_bits >> 32)) << 32) | (unsigned)uni((int)pp_bits));
  smem_lds = (unsigned)uni((int)smem_lds); wbase_lds = (unsigned)uni((int)wbase_lds);
  pix = uni(pix); depth = uni(depth); ptile = uni(ptile);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int plane16 = 16 * pp->lds_nodes;             // bytes of one node plane
  const int n_nodes16 = 16 * pp->n_nodes, lds_sph = pp->lds_sph;
  const unsigned sph_lds = smem_lds + 4u * (unsigned)plane16;
  const unsigned key_lds = wbase_lds, dump_lds = wbase_lds + 768u;
  const unsigned box_lds = wbase_lds + 4u * (unsigned)(kPooledWaveFixedDw + 256 * pp->ray_planes);
  const unsigned leaf_lds = box_lds + 4u * (unsigned)pp->capb;
  const __amdgpu_buffer_rsrc_t rs_nodes = make_rsrc(pp->nodes64, (unsigned)pp->n_nodes * 64u);
  const __amdgpu_buffer_rsrc_t rs_sph = make_rsrc(pp->sph, (unsigned)pp->n_sph * 16u);
  const __amdgpu_buffer_rsrc_t rs_col = make_rsrc(pp->col, (unsigned)pp->n_sph * 16u);
  const int max_depth = pp->max_depth;
  const float cull_c2 = pp->cull_c2, cull_kappa = pp->cull_kappa;
  const float rlx = pp->root_lo[0], rly = pp->root_lo[1], rlz = pp->root_lo[2];
  const float rhx = pp->root_hi[0], rhy = pp->root_hi[1], rhz = pp->root_hi[2];
  auto key_ptr = reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>(key_lds);
  Ray r;
  r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
  unsigned long long tc_tre = 0, tc_leaf = 0, tc_all = 0, tn_tre = 0, tn_leaf = 0, tn_ray = 0;   // (TRACE)
  const unsigned long long tc_begin = TRACE ? clock64() : 0ull;
  auto trace_flush = [&]() {
    if constexpr (TRACE) {
      __builtin_amdgcn_s_waitcnt(0);
      tc_all = clock64() - tc_begin;
      unsigned long long *const trace_rec = pp->trace == nullptr ? nullptr : pp->trace + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kTraceWords;
      if (lane == 0 && trace_rec != nullptr) {      // (this wave is the record's only writer; its words 13 .. 15 start at zero)
        trace_rec[13] += (tc_tre & 0xffffffffffull) | (tn_tre << 40);
        trace_rec[14] += (tc_leaf & 0xffffffffffull) | (tn_leaf << 40);
        trace_rec[15] += ((tc_all - tc_tre - tc_leaf) & 0xffffffffffull) | (tn_ray << 40);
      }
    }
  };
  int cand_j = -1;                 // (per lane) the leaf this lane tested in the bounce's last sphere-test operation, its sphere and colour
  float4 cand_s = make_float4(0.f, 0.f, 0.f, 1.f), cand_c = make_float4(0.f, 0.f, 0.f, 0.f);
  for (;;) {   // one ray of the pixel's chain per iteration
    ray_derive(r);
    if (TRACE) tn_ray++;
    unsigned long long key = kKeyInit;
    const float w2 = CULL ? cull_weight(r, cull_c2) : 0.0f;
    if (box_hit(r, rlx, rly, rlz, rhx, rhy, rhz)) {   // (uniform: every lane holds the same ray)
      if (lane == 0) {
        *key_ptr = kKeyInit;
        lds_store((int)box_lds, 0u);                  // the root: a treelet root whose box passed
      }
      int nbox = 1, nleaf = 0;
      while ((nbox | nleaf) != 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned long long tc_op = TRACE ? clock64() : 0ull;
        if (nleaf >= 64 || nbox == 0) {
          // ---- sphere tests: up to 64 leaves ----
          const int top = nleaf - 1 - lane;
          const unsigned item = lds_load4(leaf_lds + 4u * (unsigned)(top < 0 ? 0 : top));
          const bool act = top >= 0;
          nleaf = uni(nleaf > 64 ? nleaf - 64 : 0);
          const int jj = act ? ~((int)item >> 8) : 0;
          float4 s = lds_load16(sph_lds + 16u * (unsigned)(jj < lds_sph ? jj : 0));
          if (jj >= lds_sph) s = buf_load16(rs_sph, jj * 16);
          // (the lane keeps its sphere and fetches the sphere's colour alongside: if this leaf wins the fold, the shading below takes both
          // from this lane's registers instead of two dependent loads behind the fold -- ~600 cycles of every bounce, round 6)
          cand_j = act ? jj : -1;
          cand_c = buf_load16(rs_col, jj * 16);
          cand_s = s;
          bool near_root;
          const float g = sphere_root_flag(r, s.x, s.y, s.z, s.w, &near_root);
          if (act & (g < kTMax))
            __hip_atomic_fetch_min(key_ptr, ((unsigned long long)__float_as_uint(g) << 32) | ((unsigned)jj << 1) | (near_root ? 1u : 0u),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (TRACE) { __builtin_amdgcn_s_waitcnt(0); tc_leaf += clock64() - tc_op; tn_leaf++; }
        } else {
          // ---- treelet operation (treelet.h, cut of depth kTreeletDepth): up to 64 >> D roots, 2^D lanes each ----
          // lane `pos` of a group reads the record of the treelet's node at that position -- whatever record is there: its
          // masks tell -- and tests the boxes of BOTH its children, speculatively; the node was REACHED iff every box on its
          // path from the treelet's root passed (the masks name the lanes that tested them).  Reached nodes append their leaf
          // children to the leaf list and, on the treelet's last level, their passing inner children (the next roots).
          constexpr int D = kTreeletDepth;
          const int pos = lane & ((1 << D) - 1), gsh = lane & ~((1 << D) - 1);
          const int top = nbox - 1 - (lane >> D);
          const unsigned item = lds_load4(box_lds + 4u * (unsigned)(top < 0 ? 0 : top));
          const bool has = top >= 0;
          nbox = uni(nbox > (64 >> D) ? nbox - (64 >> D) : 0);
          int ni16 = (int)((item >> 4) & 0xfffffff0u) + 16 * pos;
          ni16 = ni16 < n_nodes16 ? ni16 : 0;
          const bool res = ni16 < plane16;
          const unsigned a0 = smem_lds + (unsigned)(res ? ni16 : 0);
          float4 q0 = lds_load16(a0), q1 = lds_load16(a0 + (unsigned)plane16), q2 = lds_load16(a0 + 2u * (unsigned)plane16),
                 q3 = lds_load16(a0 + 3u * (unsigned)plane16);
          if (!res) {
            q0 = buf_load16(rs_nodes, ni16 * 4);
            q1 = buf_load16(rs_nodes, ni16 * 4 + 16);
            q2 = buf_load16(rs_nodes, ni16 * 4 + 32);
            q3 = buf_load16(rs_nodes, ni16 * 4 + 48);
          }
          const int cl8 = f2i(q0.w), cr8 = f2i(q1.w);
          const unsigned ml = (unsigned)f2i(q2.w), mr = (unsigned)f2i(q3.w);
          // (CULL: the boxes are tested against the ray's best root so far -- the key's high word -- instead of 1e9: lane_core.h, cull_limit.
          // A node of the treelet that fails only against the limit is not reached, and neither is anything below it)
          const float limc = CULL ? cull_limit(__uint_as_float(lds_load4(key_lds + 4u)), w2, cull_kappa) : kTMax;
          const unsigned long long m_hl = bal(box_hit_clamped(r, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, limc));
          const unsigned long long m_hr = bal(box_hit_clamped(r, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, limc));
          const bool reach = has && tl_reached(ml, mr, pos, (unsigned)(m_hl >> gsh), (unsigned)(m_hr >> gsh));
          const unsigned long long m_reach = bal(reach), m_exit = m_reach & bal(tl_frontier(ml));
          const unsigned long long m_ln = bal(cl8 < 0), m_rn = bal(cr8 < 0);
          const unsigned long long m_inl = m_exit & ~m_ln & m_hl, m_inr = m_exit & ~m_rn & m_hr;
          const unsigned long long m_lfl = m_reach & m_ln, m_lfr = m_reach & m_rn;
          const int c_inl = __popcll(m_inl), c_lfl = __popcll(m_lfl);
          const int b_box = (int)box_lds + 4 * nbox, b_leaf = (int)leaf_lds + 4 * nleaf, dump = (int)dump_lds;
          const int a_l = sel_mask(m_lfl, sel_mask(m_inl, dump, b_box + 4 * lane_rank(m_inl)), b_leaf + 4 * lane_rank(m_lfl));
          const int a_r = sel_mask(m_lfr, sel_mask(m_inr, dump, b_box + 4 * lane_rank_from(m_inr, c_inl)),
                                   b_leaf + 4 * lane_rank_from(m_lfr, c_lfl));
          lds_store(a_l, (unsigned)cl8);
          lds_store(a_r, (unsigned)cr8);
          nbox = uni(nbox + c_inl + __popcll(m_inr));
          nleaf = uni(nleaf + c_lfl + __popcll(m_lfr));
          if (TRACE) { __builtin_amdgcn_s_waitcnt(0); tc_tre += clock64() - tc_op; tn_tre++; }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      key = *key_ptr;
    }
    // ---- the rest of ray_colour's iteration, as in the pooled loop's SHADE ----
    const float best = __uint_as_float((unsigned)(key >> 32));
    const bool hit = key != kKeyInit;
    const int wj = hit ? (int)((unsigned)key >> 1) : 0;
    float4 c, s;
    const unsigned long long m_win = bal(hit && cand_j == wj);
    if (m_win != 0ull) {          // (uniform) a lane of the last sphere-test operation holds the winner's sphere and colour
      const int src = uni((int)__builtin_ctzll(m_win));
      auto rl = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
      s = make_float4(rl(cand_s.x), rl(cand_s.y), rl(cand_s.z), rl(cand_s.w));
      c = make_float4(rl(cand_c.x), rl(cand_c.y), rl(cand_c.z), rl(cand_c.w));
    } else {
      c = buf_load16(rs_col, wj * 16);
      s = lds_load16(sph_lds + 16u * (unsigned)(wj < lds_sph ? wj : 0));
      if (wj >= lds_sph) s = buf_load16(rs_sph, wj * 16);
    }
    cand_j = -1;
    if (!hit) {
      s = make_float4(0.f, 0.f, 0.f, 1.f);
      c = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool have = hit;
    float t = best;
    if (hit && !rehit_is_best(best, ((unsigned)key & 1u) != 0u)) have = rehit_full(r, best, s.x, s.y, s.z, s.w, &t);
    int32_t pixel;
    if (!shade_ray<false>(r, have, t, s.x, s.y, s.z, c.x, c.y, c.z, c.w, lr, lg, lb, depth, max_depth, &pixel)) {
      if (lane == 0) {
        pp->out[pix] = pixel;
        if (pp->cost != nullptr) {
          if (pp->cost_px != nullptr) pp->cost_px[pix] = (unsigned char)(depth < 254 ? depth + 1 : 255);
          if (depth >= 2) atomicMax(&pp->cost[ptile], depth + 1);
        }
      }
      trace_flush();
      return;
    }
  }
}

// Work items are one dword: (reference << 8) | (slot * 4); `reference` is an inner node index
// (box stack) or ~leaf index (leaf list), both < 2^23.
//
// ALL_LDS: the whole traversal copy (nodes + sphere table) is staged in LDS, so the global
// (buffer_load) path is compiled out.
// COLD: the instantiation for SMALL single frames (~360 x 360 .. 800 x 800 pixels), whose time is what their last bounce chains
// take.  A wave that cannot refill any more (the queue is dry, or it holds a deep tile) and is left with at most p.cold live
// rays, all at a bounce boundary, hands them to solo_trace one after the other from INSIDE the loop.  The call costs this
// instantiation 4-7 % in every regime -- which is why the kernels of larger frames and batches do not have
// it -- and a small frame gets it back: rgbbox 500 x 500, first frame 0.395 -> 0.333 ms (one ray), later frames 0.280 -> 0.225
// (three); at 1000 x 1000 it is neutral to +2 % (profiles/r04/exp/e7, e8).  (TAIL == 1)
// DONATE (TAIL == 2): the instantiation for UNORDERED single frames (a view's first).  Their long chains start whenever the raster
// reaches them, so the launch drains for a long time after the first waves have run dry (rgbbox 1000 x 1000: the queue is empty
// at 0.6 of the span).  A wave that has left the pooled loop does not end: it offers itself in a workgroup word and sleeps; a
// wave that cannot refill gives the rays it has just scattered, one each, to the waves on offer (the ray's 12 dwords
// into the waiting wave's idle ray table, then its inbox flag); the receiver walks the chain in solo_trace -- 2.5-4.5 us per
// bounce instead of 7-16 -- and offers itself again.  Everything is LDS and workgroup scope (the hand-over through device
// memory of round 3 failed on device-scope coherence); the only code inside the loop is the donor's block in SHADE, the call sits
// behind the loop.  First frames -11 .. -28 %, ordered frames unchanged -- they keep their kernels (profiles/r04/exp/e13, e14).
// ORD: the instantiation for ORDERED single frames whose queue runs over the view's PIXEL LIST (rt_device.hpp: pixel tickets) instead
// of its tiles: pixels sorted by the length of their bounce chains, longest first; the longest go out one per ticket and are walked
// by solo_trace in the prologue, the next ones 8 / 16 / 32 per ticket to waves that do not refill while they trace them, the bulk 64
// per ticket.  A tile's 64 pixels mix one or two long chains with dozens of short ones, so a wave that holds a deep TILE parks most of
// its slots, and one that does not walks the long chain at a full wave's cadence; a ticket of pixels with EQUAL chain lengths keeps
// its wave exactly as full as the chain's deadline allows, and the last tickets of the queue are all one-ray pixels (no drain).
// CULL: boxes are tested against the slot's best root so far instead of the fixed 1e9 (lane_core.h: cull_limit) -- a
// subtree whose every root is proven larger than a root already found is not walked.  Same pixels (the fold's RESULT is the contract,
// ray.fut:76-86), fewer tests: irreg 1000 x 1000 -15 % box tests, -36 % sphere tests in this kernel's order (tools/cull_pooled.cpp).
// THREADS == 256 (round 6): the shape of FIVE workgroups of four waves per CU -- five waves per SIMD instead of four -- for scenes that are
// read from L2: their frames are bound by how much latency the resident waves hide.  The second launch-bounds argument makes the register
// allocator stay within the 96 VGPRs five waves per SIMD leave each (the 16-wave kernels hold 99-101; the difference is one dword spilled
// in SHADE).  (Two workgroups of ten waves do not do it: a workgroup's waves go to the SIMDs round robin from SIMD 0, ten waves are 3 3 2 2,
// and a second workgroup would need six wave slots of 96 VGPRs on SIMD 0 -- it never becomes resident: measured, profiles/r06/exp/e11.)
// SPILL: the instantiation of that shape for trees TALLER than 15 levels, whose box stacks may outgrow the LDS a twentieth of a CU leaves them (below); its
// value is the stack size beyond which a full BOX operation spills first -- a literal: p.capb - 64 with p.capb = 1 088 (production) or 192 (stack_cap, testing);
// compared with p.capb itself the check was a scalar load and its wait in every BOX operation of a well-filled stack: the 10^6-sphere frame 7 % slower.
template <int THREADS, bool ALL_LDS, bool STATS, bool SOLO, int TAIL = 0, bool ORD = false, bool CULL = false, int SPILL = 0>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 5 : (THREADS / 64 + 3) / 4) void pooled_kernel(KParams p) {
  constexpr bool COLD = TAIL == 1, DONATE = TAIL == 2;
  static_assert(!ORD || TAIL != 1, "ORD: no COLD variant");   // (ORD + DONATE: a frame rendered through a pixel list BORROWED from a neighbouring view, round 6)
  static_assert(!CULL || !ALL_LDS, "CULL: instantiated for the general scene path only");
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int plane = p.lds_nodes;          // float4 per node plane
  const int sph_base = 4 * plane;
  // per-wave region: key[64] (u64) | cnt[64] | dump[4] | rays[ray_planes][64] float4 | box stack[capb] | leaf list[capl]
  const int per_wave_dw = pooled_wave_dw(p.ray_planes, p.capb, p.capl);
  unsigned *const wbase = reinterpret_cast<unsigned *>(smem + sph_base + p.lds_sph) + wave * per_wave_dw;
  unsigned long long *const wkey = reinterpret_cast<unsigned long long *>(wbase);
  int *const wcnt = reinterpret_cast<int *>(wbase + 128);
  unsigned *const wdump = wbase + 192;    // where lanes with nothing to append write
  float4 *const wray = reinterpret_cast<float4 *>(wbase + kPooledWaveFixedDw);   // [0..63] {o.xyz, a}  [64..127] {1/d, 0}  ([128..191] {d, 0})
  unsigned *const wbox = wbase + kPooledWaveFixedDw + 256 * p.ray_planes;
  unsigned *const wleaf = wbox + p.capb;
  const __amdgpu_buffer_rsrc_t rs_nodes = make_rsrc(p.nodes64, (unsigned)p.n_nodes * 64u);
  const __amdgpu_buffer_rsrc_t rs_sph = make_rsrc(p.sph, (unsigned)p.n_sph * 16u);
  const __amdgpu_buffer_rsrc_t rs_col = make_rsrc(p.col, (unsigned)p.n_sph * 16u);

  // stage the node prefix as four planes of quarters, then the sphere prefix
  for (int i = threadIdx.x; i < 4 * plane; i += THREADS) smem[(i & 3) * plane + (i >> 2)] = p.nodes64[i];
  for (int i = threadIdx.x; i < p.lds_sph; i += THREADS) smem[sph_base + i] = p.sph[i];
  // Zero the wave's region: lanes without an item read a stale entry and compute on it with
  // their results masked off, so every stale entry must decode to valid indices.
  for (int i = lane; i < per_wave_dw; i += 64) wbase[i] = 0u;
  // DONATE: two words of the workgroup in wave 0's spare dump dwords -- [0] the waves that wait for a ray (bit = wave), [1] the
  // waves that are still in the pooled loop -- and every wave's own inbox flag in its dump[3]; the inbox itself is the first
  // three float4 of the waiting wave's ray table
  unsigned *const wg_words = reinterpret_cast<unsigned *>(smem + sph_base + p.lds_sph) + 193;
  if (DONATE && threadIdx.x == 0) wg_words[1] = THREADS / 64;
  __syncthreads();
  wkey[lane] = kKeyInit;

  // ---- slot state, owned by this lane ----
  Ray r = {};
  float lr = 1.0f, lg = 1.0f, lb = 1.0f;
  int depth = 0;
  int pix = -1;            // -1: slot empty
  unsigned long long n_rays = 0, n_box = 0, n_sph = 0;
  // ---- wave state (uniform) ----
  int nbox = 0, nleaf = 0;
  // SPILL (four-wave workgroups in the shape of twenty waves per CU, tall trees): p.capb is smaller than the stack's bound (trees taller than 15 levels do not
  // leave a wave 64 H + 63 dwords there).  A full BOX operation that could push beyond it first moves the OLDEST half of the stack -- its bottom -- to the wave's
  // region of p.spill; the newest spilled chunk comes back when the LDS stack has run empty; no SHADE while anything is spilled.  The order of the operations is
  // that of the unbounded LIFO (its bottom is what it would reach last anyway), so the bound on LDS + memory together is the proven 64 H + 63.  Scenes met so
  // far stay far below their capacity (high-water marks 437 .. 588 items, tools/trace_waves.py): the path is the guarantee, not the rule.
  static_assert(!SPILL || (THREADS == 256 && !ALL_LDS && !SOLO && TAIL == 0 && !ORD), "SPILL: the plain and CULL kernels of the four-wave workgroups");
  // (the number of spilled items lives in the wave's second dump dword, the region's base is recomputed where it is needed: the loop has no scalar register to
  // spare -- with both kept in registers the kernel ran 4-5 % slower whether or not anything spilled, profiles/r06/exp/e13)
  unsigned *const wspill = wbase + 193;
  auto spill_region = [&]() { return p.spill + (size_t)(blockIdx.x * (THREADS / 64) + wave) * (size_t)p.spill_stride; };
  unsigned q_next = 0, q_end = 0;
  int q_tile = 0;          // tile the current ticket maps to
  int q_col0 = 0, q_row0 = 0;   // its first pixel column / local row
  int q_frame_off = 0;     // batch launch: first pixel of the ticket's frame in p.out
  Cam q_cam = p.cam;       // ... and that frame's camera
  int ptile = 0;           // (per lane) tile of the pixel in this slot, for the cost record
  bool exhausted = false;
  // A wave that draws a DEEP tile (one whose longest bounce chain was long in the recorded frame)
  // stops refilling until that tile is finished and raises its issue priority: the frame cannot
  // end before its longest chain does, and a chain advances ~3x faster in a wave that is not
  // busy with 63 other rays' work.
  bool hold = false;
  // ... and a deep tile is handed out in 2^deep_split pieces (tickets), to as many waves: its 64 rays in one wave are 30-40
  // operations per bounce while most of them are alive, a quarter of them in each of four waves about ten.  The first
  // n_split tiles of the order are the deep ones: tickets [0, n_split << deep_split) are their pieces.
  // Only the deepest of them, though: the pieces may occupy a 32nd of the launch's waves (rgbbox 1000x1000 has 158 tiles
  // with chains of >= 32 bounces -- as quarters they would park 15 % of the waves on 16 rays each).
  // (per shard of the queue: a 32nd of the shard's home waves)
  const bool deep_on = p.nframes == 1 && p.order != nullptr && p.deep_class > 0;
  // The queue is sharded (rt_device.hpp): this wave's home shard is its workgroup's XCD; its first ticket is its own
  // number among the home shard's waves (consecutive tickets -- the deepest tiles -- land on different CUs), the
  // counters hand out the tickets behind those.
  const unsigned nwaves = gridDim.x * (THREADS / 64);
  const int ns_log2 = p.nshards > 1 ? 3 : 0;
  auto queue_const = [&]() {   // (built where it is used: nothing of it stays live across the render loop)
    QueueConst qc;
    qc.ns_log2 = ns_log2; qc.tiles_x = p.tiles_x; qc.tiles_y = p.tiles_y; qc.nframes = p.nframes;
    qc.interleave = p.interleave;
    qc.ds = p.deep_split; qc.tpt = p.tpt_log2; qc.cap_log2 = p.deep_cap_log2; qc.ntiles = p.nchunks;
    qc.order = (!ORD && deep_on) ? p.order : nullptr; qc.deep_class = p.deep_class;
    qc.px = ORD ? p.px_hdr : nullptr;
    qc.home_waves = nwaves >> ns_log2;                       // the same for every shard (the grid is a multiple of nshards)
    qc.q_static = p.static_first ? qc.home_waves : 0u;
    return qc;
  };
  unsigned q_state = queue_state_init((int)(blockIdx.x & ((1u << ns_log2) - 1u)), p.static_first != 0);
  bool q_enter = false;    // a ticket was drawn: (re)derive the tile's position
  // instrumented build only: per-wave timeline (rt_render_trace)
  unsigned long long tr_t0 = 0, tr_exh = 0, tr_c0 = 0, tr_ops[3] = {0, 0, 0}, tr_items[2] = {0, 0};
  unsigned long long tr_cyc[5] = {0, 0, 0, 0, 0}, tr_nt = 0, tr_nb2 = 0;   // shader cycles inside BOX / BOX2 / BOXT / LEAF / SHADE operations, treelet operations
  int tr_maxdepth = 0, tr_maxbox = 0, tr_maxleaf = 0;
  if (STATS) {
    tr_t0 = wall_clock64();   // 100 MHz, one counter for the whole chip (clock64 is per XCD)
    tr_c0 = clock64();
  }

  // SOLO pixels: while this wave's tickets are single pixels of the deepest tiles (rt_device.hpp: ticket_span; the first
  // tickets of the launch, most of them static), each is traced by solo_trace with the whole wave behind it.  HERE, ahead
  // of the pooled loop: a call inside the loop cost the whole kernel 4-7 % (registers saved around it, scalar spills).
  // (its own instantiation: with the call compiled in, the loop behind it runs 1-5 % slower -- scalar registers saved around
  // the call stay spilled -- so launches without single-pixel tickets use the kernel without it)
  unsigned q_cls = kPxClasses - 1;   // (ORD) class of the ticket in hand
  unsigned q_base = 0, q_ent = 0;    // (ORD) its first list position; (per lane) the list entry at q_base + lane
  if constexpr (ORD) {
    // the wave's one-pixel tickets (all of them first tickets of the list: the longest chains of the view), each walked by the
    // solo loop; the first ticket of another class goes to the pooled loop
    // (SOLO false: the instantiation for lists without a one-pixel class -- launches too large for the solo loop to matter: the
    // call's saved scalar registers cost the loop behind it 1-5 %)
    const bool solo_ok = SOLO && p.solo && p.tl_log2 == kTreeletDepth;
    for (;;) {
      TicketSpan sp;
      const QueueConst qc = queue_const();
      const unsigned wave_rank = (unsigned)uni((int)((unsigned)wave * (gridDim.x >> ns_log2) + (blockIdx.x >> ns_log2)));
      const bool got = queue_draw(q_state, qc, wave_rank, [&](int shard) {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(&p.queue[kQueueStride * shard], 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane(v);
      }, &sp);
      if (!got) {
        exhausted = true;
        break;
      }
      if (sp.cls != 0u || !solo_ok) {
        q_next = sp.q_next;
        q_end = sp.q_end;
        q_cls = sp.cls;
        q_enter = true;
        break;
      }
      if constexpr (SOLO) {
        const unsigned e = p.px_list[sp.q_next];      // uniform (scalar) load
        const int col = (int)(e & 0xffffu), lrow = (int)(e >> 16);
        const int k = lrow >> p.rpt_log2;
        const int grow = ((k * p.nparts + p.part) << p.rpt_log2) + (lrow & ((1 << p.rpt_log2) - 1));
        Ray pr;
        primary_dir_uv(p.cam, p.u_tab[col], p.v_tab[grow], pr);
        solo_trace<kSoloCull, STATS>((KParamsArg)__builtin_amdgcn_kernarg_segment_ptr(), (unsigned)(size_t)smem, (unsigned)(size_t)wbase, pr.ox, pr.oy, pr.oz,
                   pr.dx, pr.dy, pr.dz, 1.0f, 1.0f, 1.0f, lrow * p.w + col + k * p.out_skip, 0, (lrow >> 3) * p.tiles_x + (col >> 3));
      }
    }
  }
  if (!ORD && SOLO && deep_on && p.deep_split == 6 && p.tl_log2 == kTreeletDepth) {
    for (;;) {
      TicketSpan sp;
      const QueueConst qc = queue_const();
      const unsigned wave_rank = (unsigned)uni((int)((unsigned)wave * (gridDim.x >> ns_log2) + (blockIdx.x >> ns_log2)));
      const bool got = queue_draw(q_state, qc, wave_rank, [&](int shard) {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(&p.queue[kQueueStride * shard], 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane(v);
      }, &sp);
      if (!got) {
        exhausted = true;
        break;
      }
      if (sp.q_end - sp.q_next != 1u) {   // a tile (or a larger piece): the pooled loop's
        q_next = sp.q_next;
        q_end = sp.q_end;
        q_enter = true;
        break;
      }
      const int tile = p.order[sp.q_next >> 6], within = (int)(sp.q_next & 63u);
      const int ty = tile / p.tiles_x;
      const int col = (tile - ty * p.tiles_x) * 8 + (within & 7), lrow = ty * 8 + (within >> 3);
      if (col < p.w && lrow < p.rows_local) {
        const int k = lrow >> p.rpt_log2;
        const int grow = ((k * p.nparts + p.part) << p.rpt_log2) + (lrow & ((1 << p.rpt_log2) - 1));
        Ray pr;
        primary_dir_uv(p.cam, p.u_tab[col], p.v_tab[grow], pr);
        solo_trace<kSoloCull, STATS>((KParamsArg)__builtin_amdgcn_kernarg_segment_ptr(), (unsigned)(size_t)smem, (unsigned)(size_t)wbase, pr.ox, pr.oy, pr.oz,
                   pr.dx, pr.dy, pr.dz, 1.0f, 1.0f, 1.0f, lrow * p.w + col + k * p.out_skip, 0, tile);
      }
    }
  }
  for (;;) {
    RT_MARK("CHOICE_BEGIN");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if constexpr (SPILL) {
      if (nbox == 0) {
        const int nspill = uni((int)*wspill);
        if (__builtin_expect(nspill != 0, 0)) {      // the LDS stack ran empty: the newest spilled chunk comes back
          const int n = nspill < 128 ? nspill : 128;
          const unsigned *const src = spill_region() + (nspill - n);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          for (int i = lane; i < n; i += 64) wbox[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (lane == 0) *wspill = (unsigned)(nspill - n);
          nbox = uni(n);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
    }
    // (both counters are wave-uniform by construction -- ballot popcounts -- and every update goes
    // through uni(): hipcc's divergence analysis otherwise carries them in VGPRs and predicates
    // the phases)
    if (STATS) {
      tr_maxbox = nbox > tr_maxbox ? nbox : tr_maxbox;
      tr_maxleaf = nleaf > tr_maxleaf ? nleaf : tr_maxleaf;
    }
    // Which operation?  The two common cases first, on two scalar compares: a full leaf batch is drained, else a
    // full box batch is expanded (the bound on the leaf list needs that order).  Only a wave with less than a
    // batch in both lists looks at finished folds and vacant slots.
    bool leaf_op = nleaf >= 64;
    if (nbox < 64 && nleaf < 64) {
     bool drain = false;    // the leaf list must be emptied before folds can be finished
     const int thr = p.thr_shade;
     // Not a full wave of work in either list: look at completed folds / vacant slots -- unless even
     // ALL live slots being finished could not reach the threshold (a wave nursing a few deep bounce
     // chains: the look costs an LDS round trip per operation on the frame's critical path).
     const unsigned long long m_live = bal(pix >= 0);
     if (hold && m_live == 0ull) {   // the deep tile is finished: back to normal service
       hold = false;
       __builtin_amdgcn_s_setprio(0);
     }
     const bool vacant = (pix < 0) & !exhausted & !hold;
     // (... and not while the box stack still holds look_max items or more: there is box work for at least half a wave, the
     // folds that are finished can wait one more operation, and the look's LDS round trip is saved -- 1-2.5 % in the throughput
     // regimes, profiles/r04/exp/e10; a single frame of <= 32 768 tiles looks whenever fewer than 64 items are left, as before)
     // (SPILL: no SHADE while a part of the stack is in memory -- the bound on the stack rests on a SHADE finding at most 63 items)
     const bool spilled = SPILL && uni((int)*wspill) != 0;
     if (nbox == 0 || (!spilled && nbox < p.look_max && (int)__popcll(m_live | bal(vacant)) >= thr)) {
      // With leaf items pending `done` over-estimates (the counter covers inner-node items only): it
      // then only decides whether to drain the leaf list now.
      const bool done = (pix >= 0) & (wcnt[lane] == 0);
      const int ns = __popcll(bal(done | vacant));
      if (ns >= thr || nbox == 0) {
        if (nleaf > 0) {
          drain = true;
        } else {
          if (ns == 0) break;
          // ---- SHADE: finish completed folds, refill vacant slots, push the new roots ----
          RT_MARK("SHADE_BEGIN");
          bool root = false;
          if (STATS) tr_ops[2]++;
          const unsigned long long tr_s0 = STATS ? clock64() : 0ull;
          if (done) {
            if (STATS) tr_maxdepth = depth > tr_maxdepth ? depth : tr_maxdepth;
            const unsigned long long key = wkey[lane];
            const float best = __uint_as_float((unsigned)(key >> 32));
            const bool hit = key != kKeyInit;
            const int bestj = (int)((unsigned)key >> 1);
            // winner's sphere and colour: both loads are issued before either is used (two
            // dependent global latencies in a row were ~10 % of a lone wave's bounce); the sphere
            // comes from the LDS copy when it is staged
            const int wj = hit ? bestj : 0;
            float4 c = buf_load16(rs_col, wj * 16), s;
            if (ALL_LDS) {
              s = smem[sph_base + wj];
            } else {
              s = smem[sph_base + (wj < p.lds_sph ? wj : 0)];
              if (wj >= p.lds_sph) s = buf_load16(rs_sph, wj * 16);
            }
            if (!hit) {
              s = make_float4(0.f, 0.f, 0.f, 1.f);
              c = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // the (0.0, t+1) re-intersection returns t = best unless the fold's root was
            // displaced (near_root) or best+1 rounds to best: only then redo it literally
            bool have = hit;
            float t = best;
            if (hit && !rehit_is_best(best, ((unsigned)key & 1u) != 0u)) have = rehit_full(r, best, s.x, s.y, s.z, s.w, &t);
            int32_t pixel;
            if (shade_ray<false>(r, have, t, s.x, s.y, s.z, c.x, c.y, c.z, c.w, lr, lg, lb, depth, p.max_depth, &pixel)) {
              root = true;
            } else {
              p.out[pix] = pixel;
              // cost record for the adaptive order: the longest bounce chain seen in the tile, and the pixel's own
              if (p.cost != nullptr) {
                if (p.cost_px != nullptr) p.cost_px[pix] = (unsigned char)(depth < 254 ? depth + 1 : 255);
                if (depth >= 2) atomicMax(&p.cost[ptile], depth + 1);
              }
              pix = -1;
            }
          }
          bool want = (pix < 0) & !exhausted & !hold;
          int slot = -1;
          unsigned long long m = bal(want);
          while (ORD && m != 0ull) {     // pixel tickets: the next pixels of the view's list
            if (__builtin_expect(q_next == q_end, 0)) {
              if (hold) break;           // a ticket of a held class is in flight: no further tickets for now
              TicketSpan sp;
              const QueueConst qc = queue_const();
              const unsigned wave_rank = (unsigned)uni((int)((unsigned)wave * (gridDim.x >> ns_log2) + (blockIdx.x >> ns_log2)));
              const bool got = queue_draw(q_state, qc, wave_rank, [&](int shard) {
                unsigned v = 0;
                if (lane == 0) v = atomicAdd(&p.queue[kQueueStride * shard], 1u);
                return (unsigned)__builtin_amdgcn_readfirstlane(v);
              }, &sp);
              if (!got) {
                exhausted = true;
                if (STATS) tr_exh = wall_clock64();
                break;
              }
              q_next = sp.q_next;
              q_end = sp.q_end;
              q_cls = sp.cls;
              q_enter = true;
            }
            if (__builtin_expect(q_enter, 0)) {
              q_enter = false;
              if ((p.px_hold >> q_cls) & 1) {
                hold = true;
                // (the issue priority of a wave that holds a ticket: s_setprio takes an immediate)
                if (p.px_prio >= 3) __builtin_amdgcn_s_setprio(3);
                else if (p.px_prio == 2) __builtin_amdgcn_s_setprio(2);
                else if (p.px_prio == 1) __builtin_amdgcn_s_setprio(1);
              }
              // the ticket's list entries, one per lane, in ONE coalesced load: the refills that consume the ticket take
              // theirs from the lane that holds it (ds_bpermute) instead of a dependent global load each
              q_base = q_next;
              q_ent = (unsigned)lane < q_end - q_next ? p.px_list[q_next + (unsigned)lane] : 0u;
            }
            const unsigned avail = q_end - q_next;
            const unsigned rank = (unsigned)lane_rank(m);
            const unsigned cnt = (unsigned)__popcll(m);
            const unsigned e = (unsigned)__builtin_amdgcn_ds_bpermute((int)(((q_next - q_base + rank) & 63u) << 2), (int)q_ent);
            if (want & (rank < avail)) {
              const int col = (int)(e & 0xffffu), lrow = (int)(e >> 16);
              const int k = lrow >> p.rpt_log2;
              slot = lrow * p.w + col + k * p.out_skip;
              ptile = (lrow >> 3) * p.tiles_x + (col >> 3);
              const int grow = ((k * p.nparts + p.part) << p.rpt_log2) + (lrow & ((1 << p.rpt_log2) - 1));
              // (u, v from the tables, as the tile tickets take them; computing them here -- two IEEE divisions instead of two dependent
              // loads -- measured 0 .. +4 % SLOWER: a refill is bound by its VALU work, profiles/r05/exp/e9)
              primary_dir_uv(p.cam, p.u_tab[col], p.v_tab[grow], r);
              want = false;
            }
            q_next += (cnt < avail) ? cnt : avail;
            m = bal(want);
          }
          while (!ORD && m != 0ull) {    // wave-uniform loop
            if (__builtin_expect(q_next == q_end, 0)) {   // (cold: the register allocator must not favour the draw's values over the hot loop's)
              if (hold) break;           // a deep tile is in flight: no further tickets for now
              // A ticket: 1 << tpt_log2 consecutive positions of a shard's segment (a batch launch and a large frame draw
              // four tiles at a time: 4096 waves on one counter otherwise saturate it -- ~90 atomics per microsecond --
              // before they saturate the chip), except that the first tickets of a shard are pieces of its deepest tiles
              // (64 >> deep_split consecutive pixels: whole rows).  The wave's very first ticket costs no atomic.
              TicketSpan sp;
              const QueueConst qc = queue_const();
              const unsigned wave_rank = (unsigned)uni((int)((unsigned)wave * (gridDim.x >> ns_log2) + (blockIdx.x >> ns_log2)));   // (uni: `wave` comes from threadIdx)
              const bool got = queue_draw(q_state, qc, wave_rank, [&](int shard) {
                unsigned v = 0;
                if (lane == 0) v = atomicAdd(&p.queue[kQueueStride * shard], 1u);
                return (unsigned)__builtin_amdgcn_readfirstlane(v);
              }, &sp);
              if (!got) {
                exhausted = true;
                if (STATS) tr_exh = wall_clock64();
                break;
              }
              q_next = sp.q_next;
              q_end = sp.q_end;
              q_enter = true;
            }
            if (__builtin_expect(q_enter || (q_next & 63u) == 0u, 0)) {
              q_enter = false;
              // entering a tile: where it is.  A batch launch hands out the tiles of frame 0, then of frame 1, ...:
              // frame f's pixels go to out + f * frame_stride and (when the batch carries cameras) through cams[f]
              unsigned t = q_next >> 6;
              if (p.nframes > 1) {
                unsigned f;
                if (p.order != nullptr) {
                  // one view, ordered tiles: CLASS-major over the batch -- every frame's tiles of the most expensive
                  // class (longest bounce chains) first, so the last frame's chains do not start last
                  const int *cb = p.order + p.nchunks;   // first ticket of each cost class, then the tile count
                  unsigned c = 0;
                  while (c + 1 < (unsigned)kOrderClasses && t >= (unsigned)p.nframes * (unsigned)cb[c + 1]) ++c;
                  const unsigned base = (unsigned)cb[c], size = (unsigned)cb[c + 1] - base;
                  const unsigned within = t - (unsigned)p.nframes * base;
                  f = within / size;
                  t = base + (within - f * size);
                } else {
                  f = t / (unsigned)p.nchunks;
                  t -= f * (unsigned)p.nchunks;
                }
                q_frame_off = (int)(f * (unsigned)p.frame_stride);
                if (p.cams != nullptr) q_cam = p.cams[f];
              }
              bool deep_tile;
              if (p.nshards == 1 || p.interleave) {   // one queue over all tiles: a position is a ticket of the one order
                q_tile = p.order != nullptr ? p.order[t] : (int)t;   // uniform (scalar) load
                deep_tile = deep_on && (int)t < p.order[p.nchunks + p.deep_class];
              } else {                                // strips: the position lies in the segment of the shard the ticket came from
                const QueueConst qc = queue_const();
                const Shard q_s = shard_of(queue_shard(q_state), ns_log2, qc.tiles_x, qc.tiles_y);
                q_tile = p.order != nullptr ? p.order[t] : shard_tile(q_s, (int)t - q_s.seg, p.tiles_x);
                deep_tile = deep_on && (int)t - q_s.seg < queue_ndeep(qc, queue_shard(q_state));
              }
              if (deep_tile) {
                hold = true;
                __builtin_amdgcn_s_setprio(3);
              }
              const int ty = q_tile / p.tiles_x;                    // once per tile, scalar
              q_col0 = (q_tile - ty * p.tiles_x) * 8;
              q_row0 = ty * 8;
            }
            const unsigned rest = 64u - (q_next & 63u), avail = rest < q_end - q_next ? rest : q_end - q_next;   // rest of the current tile / piece
            const unsigned rank = (unsigned)lane_rank(m);
            const unsigned cnt = (unsigned)__popcll(m);
            if (want & (rank < avail)) {
              const int within = (int)((q_next + rank) & 63u);
              const int col = q_col0 + (within & 7), lrow = q_row0 + (within >> 3);
              if (col < p.w && lrow < p.rows_local) {
                // cyclic row tiles with rows_per_tile = 1 << rpt_log2 (division-free global_row)
                const int k = lrow >> p.rpt_log2;
                slot = q_frame_off + lrow * p.w + col + k * p.out_skip;   // (out_skip: 0 packed, else the pixel's place in the full image)
                ptile = q_tile;
                const int grow = ((k * p.nparts + p.part) << p.rpt_log2) + (lrow & ((1 << p.rpt_log2) - 1));
                primary_dir_uv(q_cam, p.u_tab[col], p.v_tab[grow], r);
                want = false;
              }
            }
            q_next += (cnt < avail) ? cnt : avail;
            m = bal(want);
          }
          if (slot >= 0) {
            lr = 1.0f; lg = 1.0f; lb = 1.0f;
            depth = 0;
            pix = slot;
            root = true;
          }
          if constexpr (COLD) {
            // (the lists are empty here -- SHADE runs behind the drained leaf list, and nbox == 0 is asked for: every live ray then
            // stands at a bounce boundary -- so the wave's LDS region is free for the solo loop)
            unsigned long long m_l = bal(pix >= 0);
            if ((hold || exhausted) && nbox == 0 && m_l != 0ull && (int)__popcll(m_l) <= p.cold && bal(root) == m_l && p.tl_log2 == kTreeletDepth) {
              while (m_l != 0ull) {          // the last few rays, one after the other
                const int src = uni((int)__builtin_ctzll(m_l));
                m_l &= m_l - 1ull;
                auto rl = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
                solo_trace<kSoloCull, STATS>((KParamsArg)__builtin_amdgcn_kernarg_segment_ptr(), (unsigned)(size_t)smem, (unsigned)(size_t)wbase, rl(r.ox), rl(r.oy),
                           rl(r.oz), rl(r.dx), rl(r.dy), rl(r.dz), rl(lr), rl(lg), rl(lb), __builtin_amdgcn_readlane(pix, src),
                           __builtin_amdgcn_readlane(depth, src), __builtin_amdgcn_readlane(ptile, src));
              }
              wkey[lane] = kKeyInit;     // (the solo loop used key 0 and the lists)
              pix = -1;                  // the live slots are done: their pixels are stored
              root = false;
            }
          }
          if constexpr (DONATE) {
            // A wave that cannot refill gives the rays that stand at a bounce boundary now (just scattered: their slots have no item
            // left in either list, whatever the wave's other rays are doing) to sibling waves of its workgroup that have left the loop
            // and wait: each then walks its chain in the solo loop.  LDS only.
            if ((hold || exhausted) && p.donate > 0 && p.tl_log2 == kTreeletDepth) {
              unsigned long long m_l = bal(root);
              if (m_l != 0ull && (int)__popcll(bal(pix >= 0)) <= p.donate) {
                unsigned idle = (unsigned)uni((int)__hip_atomic_load(&wg_words[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                while (m_l != 0ull && idle != 0u) {
                  const int w = uni((int)__builtin_ctz(idle));
                  unsigned old = 0u;
                  if (lane == 0) old = __hip_atomic_fetch_and(&wg_words[0], ~(1u << w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                  idle = old & ~(1u << w);
                  if ((old >> w) & 1u) {       // the wave is ours: fill its inbox
                    const int src = uni((int)__builtin_ctzll(m_l));
                    m_l &= m_l - 1ull;
                    unsigned *const ob = wbase + (w - wave) * per_wave_dw;
                    float4 *const ib = reinterpret_cast<float4 *>(ob + kPooledWaveFixedDw);
                    if (lane == src) {
                      ib[0] = make_float4(r.ox, r.oy, r.oz, r.dx);
                      ib[1] = make_float4(r.dy, r.dz, lr, lg);
                      ib[2] = make_float4(lb, __int_as_float(pix), __int_as_float(depth), __int_as_float(ptile));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == src) {
                      __hip_atomic_store(&ob[195], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                      pix = -1;
                      root = false;
                    }
                  }
                }
              }
            }
          }
          // A new fold starts with the ROOT's box test (items are nodes whose own box passed).
          if (root) ray_derive(r);   // one place for both scattered and primary rays
          const bool root_hit = root && box_hit(r, p.root_lo[0], p.root_lo[1], p.root_lo[2], p.root_hi[0], p.root_hi[1], p.root_hi[2]);
          if (root) {
            wkey[lane] = kKeyInit;
            wcnt[lane] = root_hit ? 1 : 0;    // 0: the fold is already complete (a miss), shaded next time
            wray[lane] = make_float4(r.ox, r.oy, r.oz, r.a);
            wray[64 + lane] = make_float4(r.ix, r.iy, r.iz, CULL ? cull_weight(r, p.cull_c2) : 0.0f);   // (CULL: the ray's W2)
            if (p.ray_planes == 3) wray[128 + lane] = make_float4(r.dx, r.dy, r.dz, 0.0f);
            if (STATS) { n_rays++; n_box++; }
          }
          const unsigned long long m_root = bal(root_hit);
          if (root_hit) wbox[nbox + lane_rank(m_root)] = (unsigned)lane << 2;   // (node 0, slot = lane)
          nbox = uni(nbox + __popcll(m_root));
          // Issue priority follows the deepest bounce chain this wave carries: the frame cannot
          // end before its longest chain (up to 50 dependent folds) does, and a wave that shares
          // its SIMD's issue slots evenly with 3 others walks that chain 4x slower.
          if (p.prio_depth > 0 && !hold) {
            const bool live = pix >= 0;
            if (bal(live && depth >= 4 * p.prio_depth) != 0ull) __builtin_amdgcn_s_setprio(3);
            else if (bal(live && depth >= 2 * p.prio_depth) != 0ull) __builtin_amdgcn_s_setprio(2);
            else if (bal(live && depth >= p.prio_depth) != 0ull) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
          if (STATS) tr_cyc[4] += clock64() - tr_s0;
          RT_MARK("SHADE_END");
          continue;
        }
      }
     }
     leaf_op = drain | (nbox == 0);
    }
    if (leaf_op) {
      // ---- LEAF: up to 64 (slot, sphere) items ----
      RT_MARK("LEAF_BEGIN");
      if (STATS) { tr_ops[1]++; tr_items[1] += nleaf < 64 ? nleaf : 64; }
      const unsigned long long tr_l0 = STATS ? clock64() : 0ull;
      const int top = nleaf - 1 - lane;
      const unsigned item = wleaf[top < 0 ? 0 : top];
      const bool act = top >= 0;
      nleaf = uni(nleaf > 64 ? nleaf - 64 : 0);
      const int sl = (int)(item & 0xfcu) >> 2;
      const int j = ~((int)item >> 8);        // stale zero entry -> ~0 = -1: masked below
      const float4 ra = wray[sl];
      Ray q;
      q.ox = ra.x; q.oy = ra.y; q.oz = ra.z; q.a = ra.w;
      if (p.ray_planes == 3) {                // wave-uniform
        const float4 rd = wray[128 + sl];
        asm volatile("" ::"v"(rd.w));         // keep it a 16-byte read (ds_read_b96 is slower)
        q.dx = rd.x; q.dy = rd.y; q.dz = rd.z;
      } else {                                // the direction straight from the owning lane's registers
        const int sl4 = (int)(item & 0xfcu);
        q.dx = pull(sl4, r.dx); q.dy = pull(sl4, r.dy); q.dz = pull(sl4, r.dz);
      }
      const int jj = act ? j : 0;
      float4 s;
      if (ALL_LDS) {
        s = smem[sph_base + jj];
      } else {
        s = smem[sph_base + (jj < p.lds_sph ? jj : 0)];
        if (jj >= p.lds_sph) s = buf_load16(rs_sph, jj * 16);
      }
      if (STATS) n_sph += act ? 1 : 0;
      bool near_root;
      const float g = sphere_root_flag(q, s.x, s.y, s.z, s.w, &near_root);
      // key = (bits(t), leaf << 1 | near_root): min = smallest t, ties to the lowest leaf
      if (act & (g < kTMax))
        atomicMin(&wkey[sl],
                  ((unsigned long long)__float_as_uint(g) << 32) | ((unsigned)jj << 1) | (near_root ? 1u : 0u));
      if (STATS) { __builtin_amdgcn_s_waitcnt(0); tr_cyc[3] += clock64() - tr_l0; }
      RT_MARK("LEAF_END");
    } else {
      // ---- BOX: up to 64 (slot, node) items; each tests the boxes of BOTH children ----
      // (two instantiations: a FULL batch -- every lane has an item: no clamp, no activity mask -- and the general one)
      auto box = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        if constexpr (FULL) RT_MARK("BOXFULL_BEGIN"); else RT_MARK("BOXPART_BEGIN");
        if (STATS) { tr_ops[0]++; tr_items[0] += FULL ? 64 : nbox; }
        const int top = nbox - 1 - lane;
        const unsigned item = wbox[FULL ? top : (top < 0 ? 0 : top)];
        const unsigned long long m_act = FULL ? ~0ull : bal(top >= 0);
        nbox = uni(FULL ? nbox - 64 : 0);
        const int sl4 = (int)(item & 0xfcu);
        const int ni16 = (int)((item >> 4) & 0xfffffff0u);   // node index * 16: its byte offset within a plane
        const float4 *const rayp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wray) + 4 * sl4);
        const float4 ra = rayp[0], ri = rayp[64];
        Ray q;
        q.ox = ra.x; q.oy = ra.y; q.oz = ra.z;
        q.ix = ri.x; q.iy = ri.y; q.iz = ri.z;
        float4 q0, q1, q2, q3;
        {
          const int lo16 = ALL_LDS ? ni16 : (ni16 < 16 * plane ? ni16 : 0);
          const float4 *const np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(smem) + lo16);
          q0 = np[0]; q1 = np[plane]; q2 = np[2 * plane]; q3 = np[3 * plane];
          if (!ALL_LDS && ni16 >= 16 * plane) {
            q0 = buf_load16(rs_nodes, ni16 * 4);
            q1 = buf_load16(rs_nodes, ni16 * 4 + 16);
            q2 = buf_load16(rs_nodes, ni16 * 4 + 32);
            q3 = buf_load16(rs_nodes, ni16 * 4 + 48);
          }
        }
        asm volatile("" ::"v"(q2.w), "v"(q3.w), "v"(ra.w), "v"(ri.w));   // 16-byte reads throughout
        const int cl8 = f2i(q0.w), cr8 = f2i(q1.w);   // child references, stored pre-shifted by 8 (sign = leaf)
        // (CULL: the interval's upper end is the slot's best root so far -- the high word of its hit key, one ds_read_b32 -- widened by
        // the proven margin; the ray's weight W2 travels in the spare dword of the {1/d} entry.  Two v_fma + one v_min per item.)
        float limc = kTMax;
        if constexpr (CULL) limc = cull_limit(__uint_as_float(*reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(wkey) + 2 * sl4 + 4)), ri.w, p.cull_kappa);
        // lane masks straight from the compares; the rest is 64-bit scalar logic
        const unsigned long long m_hl = bal(box_hit_clamped(q, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, limc));
        const unsigned long long m_hr = bal(box_hit_clamped(q, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, limc));
        const unsigned long long m_ln = bal(cl8 < 0), m_rn = bal(cr8 < 0);
        // an inner child continues iff its box passes; a leaf child is tested because this node passed
        const unsigned long long m_inl = m_act & ~m_ln & m_hl, m_inr = m_act & ~m_rn & m_hr;
        const unsigned long long m_lfl = m_act & m_ln, m_lfr = m_act & m_rn;
        if (STATS) n_box += __popcll(m_act & ~m_ln & (1ull << lane)) + __popcll(m_act & ~m_rn & (1ull << lane));
        // append: left children first, then right children (two independent prefix ranks per list; the right
        // children's ranks start at the left children's count: mbcnt's addend).  ONE store per child: to the box
        // stack, to the leaf list, or to the dump dword
        const int c_inl = __popcll(m_inl), c_lfl = __popcll(m_lfl);
        const int dump = (int)(size_t)(wdump);   // LDS byte address (low 32 bits of the flat address)
        const int b_box = (int)(size_t)(wbox + nbox), b_leaf = (int)(size_t)(wleaf + nleaf);
        const int a_l = sel_mask(m_lfl, sel_mask(m_inl, dump, b_box + 4 * lane_rank(m_inl)), b_leaf + 4 * lane_rank(m_lfl));
        const int a_r = sel_mask(m_lfr, sel_mask(m_inr, dump, b_box + 4 * lane_rank_from(m_inr, c_inl)),
                                 b_leaf + 4 * lane_rank_from(m_lfr, c_lfl));
        lds_store(a_l, (unsigned)cl8 | (unsigned)sl4);
        lds_store(a_r, (unsigned)cr8 | (unsigned)sl4);
        nbox = uni(nbox + c_inl + __popcll(m_inr));
        nleaf = uni(nleaf + c_lfl + __popcll(m_lfr));
        // outstanding inner-node items of the slot: one consumed, k in {0, 1, 2} appended.  Only items with k != 1
        // touch the counter (same-address LDS atomics serialise): the ds_add runs under exactly their lane mask.
        const unsigned long long m_two = m_inl & m_inr, m_none = m_act & ~(m_inl | m_inr);
        lds_add_masked(m_two | m_none, (int)(size_t)wcnt + sl4, sel_mask(m_two, -1, 1));
        if constexpr (FULL) RT_MARK("BOXFULL_END"); else RT_MARK("BOXPART_END");
      };
      // ---- BOX2: at most 32 items, TWO lanes and TWO tree levels each.  A wave with a nearly empty stack is on some
      // frame's critical path (a long bounce chain advances one operation per tree level): lane 2k handles the LEFT child
      // of item k, lane 2k+1 the RIGHT one -- tests the child's box (from the item's record) and, if it passes, reads the
      // CHILD's record and tests the grandchildren's boxes, appending those.  Exactly the tests of two consecutive BOX
      // operations (a grandchild is tested iff its parent's box passed) for one operation's fixed latencies plus one LDS
      // round trip.
      auto box2 = [&]() {
        RT_MARK("BOX2_BEGIN");
        if (STATS) { tr_ops[0]++; tr_items[0] += nbox; }
        const int role = lane & 1;
        const int top = nbox - 1 - (lane >> 1);
        const unsigned item = wbox[top < 0 ? 0 : top];
        const bool act = top >= 0;
        nbox = uni(0);
        const int sl4 = (int)(item & 0xfcu);
        const int ni16 = (int)((item >> 4) & 0xfffffff0u);
        const float4 *const rayp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(wray) + 4 * sl4);
        const float4 ra = rayp[0], ri = rayp[64];
        Ray q;
        q.ox = ra.x; q.oy = ra.y; q.oz = ra.z;
        q.ix = ri.x; q.iy = ri.y; q.iz = ri.z;
        // this lane's child: its box is quarters (2 role, 2 role + 1) of the item's record, its reference the .w of quarter `role`
        float4 lo, hi;
        int ref;
        {
          const bool res = ALL_LDS || ni16 < 16 * plane;
          const int lo16 = res ? ni16 : 0;
          const char *const np = reinterpret_cast<const char *>(smem) + lo16;
          lo = *reinterpret_cast<const float4 *>(np + 32 * role * plane);
          hi = *reinterpret_cast<const float4 *>(np + (32 * role + 16) * plane);
          ref = f2i(reinterpret_cast<const float4 *>(np + 16 * role * plane)->w);
          if (!ALL_LDS && !res) {
            lo = buf_load16(rs_nodes, ni16 * 4 + 32 * role);
            hi = buf_load16(rs_nodes, ni16 * 4 + 32 * role + 16);
            ref = f2i(buf_load16(rs_nodes, ni16 * 4 + 16 * role).w);
          }
        }
        float limc = kTMax;
        if constexpr (CULL) limc = cull_limit(__uint_as_float(*reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(wkey) + 2 * sl4 + 4)), ri.w, p.cull_kappa);
        const bool child_leaf = act & (ref < 0);
        const bool pass = act & (ref >= 0) && box_hit_clamped(q, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, limc);
        if (STATS) n_box += (act & (ref >= 0)) ? 1 : 0;
        // second level: the child's own record (a virtual item `ref | sl4`)
        const int ci16 = pass ? (int)(((unsigned)ref >> 4) & 0xfffffff0u) : 0;
        float4 q0, q1, q2, q3;
        {
          const int lo16 = ALL_LDS ? ci16 : (ci16 < 16 * plane ? ci16 : 0);
          const float4 *const np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(smem) + lo16);
          q0 = np[0]; q1 = np[plane]; q2 = np[2 * plane]; q3 = np[3 * plane];
          if (!ALL_LDS && ci16 >= 16 * plane) {
            q0 = buf_load16(rs_nodes, ci16 * 4);
            q1 = buf_load16(rs_nodes, ci16 * 4 + 16);
            q2 = buf_load16(rs_nodes, ci16 * 4 + 32);
            q3 = buf_load16(rs_nodes, ci16 * 4 + 48);
          }
        }
        asm volatile("" ::"v"(q2.w), "v"(q3.w), "v"(ra.w), "v"(ri.w), "v"(lo.w), "v"(hi.w));
        const int cl8 = f2i(q0.w), cr8 = f2i(q1.w);
        const unsigned long long m_pass = bal(pass), m_cleaf = bal(child_leaf);
        const unsigned long long m_hl = bal(box_hit_clamped(q, q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, limc));
        const unsigned long long m_hr = bal(box_hit_clamped(q, q2.x, q2.y, q2.z, q3.x, q3.y, q3.z, limc));
        const unsigned long long m_ln = bal(cl8 < 0), m_rn = bal(cr8 < 0);
        const unsigned long long m_inl = m_pass & ~m_ln & m_hl, m_inr = m_pass & ~m_rn & m_hr;
        // leaf appends: the child itself (lanes whose child is a leaf), or its leaf children -- never both for one lane
        const unsigned long long m_lfl = (m_pass & m_ln) | m_cleaf, m_lfr = m_pass & m_rn;
        if (STATS) n_box += __popcll(m_pass & ~m_ln & (1ull << lane)) + __popcll(m_pass & ~m_rn & (1ull << lane));
        const int c_inl = __popcll(m_inl), c_lfl = __popcll(m_lfl);
        const int dump = (int)(size_t)(wdump);
        const int b_box = (int)(size_t)(wbox), b_leaf = (int)(size_t)(wleaf + nleaf);
        const int a_l = sel_mask(m_lfl, sel_mask(m_inl, dump, b_box + 4 * lane_rank(m_inl)), b_leaf + 4 * lane_rank(m_lfl));
        const int a_r = sel_mask(m_lfr, sel_mask(m_inr, dump, b_box + 4 * lane_rank_from(m_inr, c_inl)),
                                 b_leaf + 4 * lane_rank_from(m_lfr, c_lfl));
        lds_store(a_l, (unsigned)sel_mask(m_cleaf, cl8, ref) | (unsigned)sl4);
        lds_store(a_r, (unsigned)cr8 | (unsigned)sl4);
        nbox = uni(c_inl + __popcll(m_inr));
        nleaf = uni(nleaf + c_lfl + __popcll(m_lfr));
        // the slot's counter of outstanding inner-node items: this lane's appended inner grandchildren, minus the item
        // itself (counted once, by the pair's even lane)
        const int d = sel_mask(m_inl, 0, 1) + sel_mask(m_inr, 0, 1) - sel_mask(bal(act & (role == 0)), 0, 1);
        lds_add_masked(bal(d != 0), (int)(size_t)wcnt + sl4, d);
        RT_MARK("BOX2_END");
      };
      const unsigned long long tr_b0 = STATS ? clock64() : 0ull;
      int tr_kind = 0;
      if constexpr (SPILL) {
        if (__builtin_expect(nbox > SPILL, 0)) {                        // a full batch may push 128 behind its 64: the oldest half of the stack goes to memory
          const int S = (nbox >> 1) & ~63;                               // (whole chunks, at least one)
          const int nspill = uni((int)*wspill);
          unsigned *const dst = spill_region() + nspill;
          for (int i = lane; i < S; i += 64) __hip_atomic_store(&dst[i], wbox[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int i = lane; i < nbox - S; i += 64) {                    // the rest moves down (a wave's LDS operations execute in order)
            const unsigned v = wbox[S + i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            wbox[i] = v;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          if (lane == 0) *wspill = (unsigned)(nspill + S);
          nbox = uni(nbox - S);
        }
      }
      if (nbox >= 64) box(std::true_type{});
      else if (nbox <= 32 && p.box2) { box2(); tr_kind = 1; }
      else box(std::false_type{});
      if (STATS) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long dt = clock64() - tr_b0;
        if (tr_kind == 0) tr_cyc[0] += dt;
        else if (tr_kind == 1) { tr_cyc[1] += dt; tr_nb2++; }
        else tr_cyc[2] += dt;
      }
    }
  }
  if constexpr (DONATE) {
    // This wave is finished: it waits for rays of its siblings until every wave of the workgroup has left the loop.
    if (p.donate > 0 && p.tl_log2 == kTreeletDepth) {
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(&wg_words[1], ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      bool offered = false;
      for (;;) {      // (ends: every wave of the workgroup leaves the pooled loop, and decrements the count when it does)
        if (!offered) {
          if (lane == 0) __hip_atomic_fetch_or(&wg_words[0], 1u << wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          offered = true;
        }
        // the count first, the flag second: a donor fills the inbox before it leaves the loop itself
        const unsigned active = (unsigned)uni((int)__hip_atomic_load(&wg_words[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned full = (unsigned)uni((int)__hip_atomic_load(&wbase[195], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (full != 0u) {
          const float4 a = wray[0], b = wray[1], c = wray[2];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          if (lane == 0) __hip_atomic_store(&wbase[195], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          offered = false;
          __builtin_amdgcn_s_setprio(2);      // (a ray that arrives here is one of the workgroup's last)
          solo_trace<kSoloCull, STATS>((KParamsArg)__builtin_amdgcn_kernarg_segment_ptr(), (unsigned)(size_t)smem, (unsigned)(size_t)wbase, a.x, a.y, a.z, a.w,
                     b.x, b.y, b.z, b.w, c.x, __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
          __builtin_amdgcn_s_setprio(0);
          continue;
        }
        if (active == 0u) break;
        __builtin_amdgcn_s_sleep(16);
      }
    }
  }
  queue_leave(p, nwaves, lane);
  if (STATS) {
    atomicAdd(&p.stats[0], n_rays);
    atomicAdd(&p.stats[1], n_box);
    atomicAdd(&p.stats[2], n_sph);
    if (p.trace != nullptr) {
      const unsigned long long t_end = wall_clock64(), c_end = clock64();
      int md = tr_maxdepth;
      for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(md, o); md = other > md ? other : md; }
      if (lane == 0) {
        unsigned long long *rec = p.trace + (size_t)(blockIdx.x * (THREADS / 64) + wave) * kTraceWords;
        rec[0] = tr_t0; rec[1] = tr_exh; rec[2] = t_end;
        rec[3] = tr_ops[0] | (tr_ops[1] << 21) | (tr_ops[2] << 42);
        rec[4] = c_end - tr_c0;   // shader cycles of this wave's life
        rec[5] = tr_nt | (tr_nb2 << 32);
        rec[8] = tr_cyc[0]; rec[9] = tr_cyc[1]; rec[10] = tr_cyc[2]; rec[11] = tr_cyc[3]; rec[12] = tr_cyc[4];
        rec[6] = (tr_items[0] << 32) | tr_items[1];
        rec[7] = (unsigned long long)md | ((unsigned long long)tr_maxbox << 16) | ((unsigned long long)tr_maxleaf << 32);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Adaptive tile order.  The frame time at 1000x1000 is bounded below by the longest bounce
// chain (a 50-bounce pixel is ~50 x (tree height + 3) dependent wave operations), so chains
// must START early.  Every frame records per tile the longest chain it saw (p.cost); this
// kernel turns the record into the ticket -> tile table of the NEXT frame of the same
// prepared scene: a stable counting sort by chain length, descending (64 bins: the exact length
// up to 62 bounces), and clears the record.  Behind the table it stores the first ticket of each
// of the kOrderClasses coarse classes (floor(log2) of the length) -- the deep-tile threshold and
// the class-major ticket order of a batch are expressed in those.  It only permutes the order in
// which independent pixels are traced.
// ---------------------------------------------------------------------------------
constexpr int kOrderThreads = 128;
constexpr int kOrderBins = 64;

__device__ __forceinline__ int order_bin(int v) { return kOrderBins - 1 - min(kOrderBins - 1, max(v, 0)); }   // 0 = longest chains

// Per shard of the tile queue (one, or eight strips) a stable counting sort of the strip's tiles (row-major within the
// strip) into the strip's segment of the table, by up to kOrderBlocksMax workgroups: every workgroup counts its chunk of
// the tiles per bin, one workgroup per shard turns the counts into starts (bin-major, then chunk by chunk: stable) and writes
// the shard's class table, every workgroup places its chunk.  (One workgroup for everything took ~0.09 ms at 1000x1000 and
// 1.4 ms at 4000x4000 -- behind every frame that records costs, i.e. every first frame of a view.)
struct OrderChunk {
  Shard sh;
  int begin, end;      // this thread's tiles [begin, end) of the shard (thread order = tile order)
};
__device__ __forceinline__ OrderChunk order_chunk(int ntiles, int tiles_x, int nshards, int nblocks) {
  OrderChunk c;
  const int s = (int)blockIdx.x / nblocks, b = (int)blockIdx.x - s * nblocks;
  c.sh = shard_of(s, nshards > 1 ? 3 : 0, tiles_x, ntiles / tiles_x);
  const int n = c.sh.ntiles;
  const int per_block = (n + nblocks - 1) / nblocks, b0 = min(n, b * per_block), b1 = min(n, b0 + per_block);
  const int per = (b1 - b0 + kOrderThreads - 1) / kOrderThreads;
  c.begin = min(b1, b0 + (int)threadIdx.x * per);
  c.end = min(b1, c.begin + per);
  return c;
}
// hist[bin][thread] = this thread's tiles of that bin (eight independent loads at a time)
__device__ __forceinline__ void order_count(const OrderChunk &c, const int *cost, int tiles_x, int (*hist)[kOrderThreads]) {
  const int t = threadIdx.x;
  for (int b = 0; b < kOrderBins; ++b) hist[b][t] = 0;
  for (int i0 = c.begin; i0 < c.end; i0 += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u < c.end ? cost[shard_tile(c.sh, i0 + u, tiles_x)] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < c.end) hist[order_bin(v[u])][t] += 1;   // a thread touches its own column only
  }
  __syncthreads();
}
// counts[(shard * kOrderBins + bin) * nblocks + block]
__global__ __launch_bounds__(kOrderThreads) void tile_count_kernel(const int *cost, int ntiles, int tiles_x, int nshards, int nblocks, int *counts) {
  __shared__ int hist[kOrderBins][kOrderThreads];
  const OrderChunk c = order_chunk(ntiles, tiles_x, nshards, nblocks);
  order_count(c, cost, tiles_x, hist);
  const int t = threadIdx.x;
  if (t < kOrderBins) {
    int acc = 0;
    for (int k = 0; k < kOrderThreads; ++k) acc += hist[t][k];
    const int s = (int)blockIdx.x / nblocks, b = (int)blockIdx.x - s * nblocks;
    counts[(s * kOrderBins + t) * nblocks + b] = acc;
  }
}
// one workgroup per shard: counts -> starts within the shard's segment; the shard's class table
__global__ __launch_bounds__(kOrderThreads) void tile_scan_kernel(int *counts, int *order, int ntiles, int nblocks) {
  __shared__ int bin_tot[kOrderBins + 1];
  const int t = threadIdx.x, s = (int)blockIdx.x;
  int *const cs = counts + s * kOrderBins * nblocks;
  if (t < kOrderBins) {      // exclusive scan of bin t's counts over the chunks
    int acc = 0;
    for (int b = 0; b < nblocks; ++b) {
      const int c = cs[t * nblocks + b];
      cs[t * nblocks + b] = acc;
      acc += c;
    }
    bin_tot[t] = acc;
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int b = 0; b < kOrderBins; ++b) {
      const int c = bin_tot[b];
      bin_tot[b] = acc;
      acc += c;
    }
    bin_tot[kOrderBins] = acc;
    // first ticket of each coarse class (class c = chains of 2^(7-c) .. 2^(8-c) - 1 bounces): the render kernel
    // treats the tickets below order[ntiles + deep_class] as deep tiles, a batch hands tickets out class-major
    int *const table = order + ntiles + kOrderTableDw * s;
    for (int c = 0; c < kOrderClasses; ++c) table[c] = bin_tot[order_bin((1 << (kOrderClasses - c)) - 1)];
    table[kOrderClasses] = acc;
  }
  __syncthreads();
  if (t < kOrderBins)
    for (int b = 0; b < nblocks; ++b) cs[t * nblocks + b] += bin_tot[t];
}
__global__ __launch_bounds__(kOrderThreads) void tile_place_kernel(int *cost, int *order, int ntiles, int tiles_x, int nshards, int nblocks,
                                                                  const int *starts) {
  __shared__ int hist[kOrderBins][kOrderThreads];   // [bin][thread], 32 KB: counts, then exclusive positions
  const OrderChunk c = order_chunk(ntiles, tiles_x, nshards, nblocks);
  order_count(c, cost, tiles_x, hist);
  const int t = threadIdx.x;
  const int s = (int)blockIdx.x / nblocks, b = (int)blockIdx.x - s * nblocks;
  // exclusive scan of each bin's row across the threads (thread order = tile order: the sort is stable), from the chunk's start
  if (t < kOrderBins) {
    int acc = starts[(s * kOrderBins + t) * nblocks + b];
    for (int k = 0; k < kOrderThreads; ++k) {
      const int n = hist[t][k];
      hist[t][k] = acc;
      acc += n;
    }
  }
  __syncthreads();
  for (int i0 = c.begin; i0 < c.end; i0 += 8) {
    int tile[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      tile[u] = shard_tile(c.sh, i0 + u < c.end ? i0 + u : c.begin, tiles_x);
      v[u] = cost[tile[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < c.end) {
        const int bin = order_bin(v[u]);
        order[c.sh.seg + hist[bin][t]] = tile[u];
        hist[bin][t] += 1;
        cost[tile[u]] = 0;
      }
  }
}

// `scratch`: kOrderScratchInts ints of device memory (the chunks' counts)
hipError_t launch_tile_order(int *cost, int *order, int ntiles, int tiles_x, int nshards, int *scratch, hipStream_t stream) {
  if (ntiles <= 0) return hipSuccess;
  const int per_shard = (ntiles + nshards - 1) / nshards;
  int nblocks = (per_shard + 2047) / 2048;
  nblocks = nblocks < 1 ? 1 : (nblocks > kOrderBlocksMax ? kOrderBlocksMax : nblocks);
  hipLaunchKernelGGL(tile_count_kernel, dim3(nshards * nblocks), dim3(kOrderThreads), 0, stream, cost, ntiles, tiles_x, nshards, nblocks, scratch);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(nshards), dim3(kOrderThreads), 0, stream, scratch, order, ntiles, nblocks);
  hipLaunchKernelGGL(tile_place_kernel, dim3(nshards * nblocks), dim3(kOrderThreads), 0, stream, cost, order, ntiles, tiles_x, nshards, nblocks,
                     scratch);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Pixel list of a view (rt_device.hpp: pixel tickets).  The view's first frame stores, next to every pixel, the number of
// rays its chain took (p.cost_px, one byte, indexed like the framebuffer); these kernels turn that record into the list the
// view's later single frames draw their tickets from: the part's pixels as (local row << 16 | column), sorted by chain length,
// longest first (a counting sort over 64 bins; a workgroup's pixels of one bin stay together, tile by tile, so a ticket's pixels are
// neighbours wherever a region has enough pixels of one length), and the header that cuts the list into the ticket classes.
// (Measured and not kept, profiles/r05/README.md: the bulk of the list in tile order instead of sorted by its own lengths -- a wave's
// rays then differ in length as a tile's do -- rendered irreg the same and rgbbox 1000 x 1000 5-7 % slower than the plain sort.)
// Which pixel goes where changes the ORDER in which independent pixels are traced, nothing else.
// ---------------------------------------------------------------------------------
constexpr int kPxThreads = 64;        // count / place: ONE wave per workgroup, a tile per step; lane b holds bin b's count / cursor in a register
constexpr int kPxScanThreads = 256;
constexpr int kPxBins = 64;
__device__ __forceinline__ int px_bin(int rays) { return rays < kPxBins - 1 ? rays : kPxBins - 1; }   // bin = rays traced (saturating)

// The workgroup's tiles [t0, t1), one after the other, one pixel per lane.  For every tile and every bin that occurs in it (a handful:
// neighbouring pixels have similar chains) f(bin, m, col, lrow) is called by the whole wave with the lane mask m of the bin's pixels
// -- no atomics: 64 lanes adding to one LDS word serialise, and the list's order would depend on who wins.
template <class F>
__device__ __forceinline__ void px_for_each_bin(const unsigned char *cost_px, const PxGeom &g, int tiles_per_block, F &&f) {
  const int ntiles = g.tiles_x * g.tiles_y;
  const int t0 = (int)blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
  const int within = (int)threadIdx.x;
  constexpr int U = 16;          // tiles whose records are loaded before the first is consumed (a dependent load per tile: ~1.5 us each)
  for (int tb = t0; tb < t1; tb += U) {
    int bins[U], cols[U], rows[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tile = tb + u;
      const int ty = tile / g.tiles_x;
      cols[u] = (tile - ty * g.tiles_x) * 8 + (within & 7);
      rows[u] = ty * 8 + (within >> 3);
      const bool in = tile < t1 && cols[u] < g.w && rows[u] < g.rows_local;
      const size_t idx = (size_t)rows[u] * g.w + cols[u] + (size_t)(rows[u] >> g.rpt_log2) * (size_t)g.out_skip;
      bins[u] = in ? (int)cost_px[idx] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int bin = bins[u] < 0 ? -1 : px_bin(bins[u]);
      unsigned long long todo = bal(bin >= 0);
      while (todo != 0ull) {       // wave-uniform
        const int b = __builtin_amdgcn_readlane(bin, (int)__builtin_ctzll(todo));
        const unsigned long long m = bal(bin == b);
        f(b, m, cols[u], rows[u]);
        todo &= ~m;
      }
    }
  }
}
// counts[bin * nblocks + block]
__global__ __launch_bounds__(kPxThreads) void px_count_kernel(const unsigned char *cost_px, PxGeom g, int tiles_per_block, int nblocks, int *counts) {
  int mine = 0;                  // lane b: the workgroup's pixels of bin b
  px_for_each_bin(cost_px, g, tiles_per_block, [&](int b, unsigned long long m, int, int) {
    if ((int)threadIdx.x == b) mine += (int)__popcll(m);
  });
  counts[threadIdx.x * nblocks + blockIdx.x] = mine;
}
// one workgroup per bin: its counts over the workgroups of the count pass -> exclusive prefix (in place), the bin's total
__global__ __launch_bounds__(kPxScanThreads) void px_scan_kernel(int *counts, int nblocks, int *totals) {
  __shared__ int wsum[kPxScanThreads / 64];
  __shared__ int carry;
  int *const c = counts + (size_t)blockIdx.x * nblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += kPxScanThreads) {
    const int i = base + (int)threadIdx.x;
    const int v = i < nblocks ? c[i] : 0;
    int incl = v;                // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int k = 0; k < wave; ++k) before += wsum[k];
    if (i < nblocks) c[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == kPxScanThreads - 1) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}
// one wave, lane l = bin l: the bins' first list positions (DESCENDING chain length) -> totals[kPxBins ..), and the header: the model
// evaluated on the histogram (rt_device.hpp: PxPolicy), the classes' cuts
__device__ __forceinline__ long long wave_sum(long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__global__ __launch_bounds__(64) void px_header_kernel(int *totals, PxPolicy pol, int *hdr) {
  const int l = (int)threadIdx.x;
  const int cnt = totals[l];
  int suf = cnt;                                   // pixels of >= l rays: inclusive suffix sum over the bins
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_down(suf, o);
    if (l + o < 64) suf += u;
  }
  const int all = __builtin_amdgcn_readlane(suf, 0);
  totals[kPxBins + l] = suf - cnt;                 // the bin's first position: the pixels of longer chains
  auto at_least = [&](int t) { return t <= 0 ? all : (t > kPxBins - 1 ? 0 : __builtin_amdgcn_readlane(suf, t)); };   // (t: wave-uniform)
  int thr[kPxClasses - 1] = {pol.thr[0], pol.thr[1], pol.thr[2], pol.thr[3]};
  if (pol.thr[0] <= 0) {
    const unsigned long long occ = bal(cnt > 0 && l >= 1);
    const int maxlen = occ ? 63 - (int)__builtin_clzll(occ) : 1;
    const bool solo = pol.solo_cap > 0;
    long long T = (long long)maxlen * pol.g[solo ? 0 : 1];     // 0.1 us: the longest chain in the narrowest class there is
    for (int it = 0; it < 4; ++it) {
      for (int k = 0; k < kPxClasses - 1; ++k) thr[k] = (int)min((long long)kPxBins, T / pol.g[k + 1] + 1);   // class k: chains too long for class k + 1
      if (!solo) thr[0] = kPxBins;
      int k = kPxClasses - 1;                                  // this lane's bin: its class, its rays' share of the waves' time (0.1 us)
      while (k > 0 && l >= thr[k - 1]) --k;
      const long long mine = l == 0 ? 0ll : (k == kPxClasses - 1 ? (long long)cnt * l * pol.ray_ns / 100 : (long long)cnt * l * pol.g[k] / (1 << px_log2(k)));
      const long long Tn = max(T, wave_sum(mine) / max(1, pol.nwaves));
      if (Tn <= T) break;                                      // (uniform)
      T = Tn;
    }
  }
  // the one-pixel class: at most solo_cap pixels (suf is non-increasing in l: the first bin that fits)
  int t0 = kPxBins;
  if (pol.solo_cap > 0) {
    const unsigned long long fits = bal(suf <= pol.solo_cap && l >= max(thr[0], 1));
    t0 = fits ? (int)__builtin_ctzll(fits) : kPxBins;
  }
  const int t1 = min(thr[1], t0), t2 = min(thr[2], t1), t3 = min(thr[3], t2);
  const int p1 = at_least(t0), p2 = at_least(t1), p3 = at_least(t2), p4 = at_least(t3);
  if (l == 0) {
    // class k holds the chains of >= its cut that no earlier class holds: its first position is the number of pixels with longer chains
    int pos[kPxClasses + 1] = {0, p1, p2, p3, p4, all};
    px_make_header(pos, hdr);
    hdr[6] = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24);          // (for diagnostics: the cuts that were used)
    hdr[7] = pol.zip;
  }
}
__global__ __launch_bounds__(kPxThreads) void px_place_kernel(const unsigned char *cost_px, PxGeom g, int tiles_per_block, int nblocks, const int *starts,
                                                            const int *bin_start, unsigned *list) {
  int cursor = starts[threadIdx.x * nblocks + blockIdx.x] + bin_start[threadIdx.x];   // lane b: where the workgroup's next pixel of bin b goes
  px_for_each_bin(cost_px, g, tiles_per_block, [&](int b, unsigned long long m, int col, int lrow) {
    const int base = __builtin_amdgcn_readlane(cursor, b);
    if ((m >> threadIdx.x) & 1ull) list[base + lane_rank(m)] = ((unsigned)lrow << 16) | (unsigned)col;
    if ((int)threadIdx.x == b) cursor += (int)__popcll(m);
  });
}

// `scratch`: px_scratch_ints() ints -- [bin][workgroup] counts, then the bins' totals and first positions
hipError_t launch_px_order(const unsigned char *cost_px, const PxGeom &g, const PxPolicy &pol, unsigned *list, int *hdr, int *scratch, hipStream_t stream) {
  const int ntiles = g.tiles_x * g.tiles_y;
  if (ntiles <= 0) return hipSuccess;
  int tpb = 16;                                     // tiles per workgroup: 1024 pixels, or more for a large frame
  while ((ntiles + tpb - 1) / tpb > kPxBlocksMax) tpb *= 2;
  const int nblocks = (ntiles + tpb - 1) / tpb;
  int *const totals = scratch + (size_t)kPxBins * kPxBlocksMax;
  hipLaunchKernelGGL(px_count_kernel, dim3(nblocks), dim3(kPxThreads), 0, stream, cost_px, g, tpb, nblocks, scratch);
  hipLaunchKernelGGL(px_scan_kernel, dim3(kPxBins), dim3(kPxScanThreads), 0, stream, scratch, nblocks, totals);
  hipLaunchKernelGGL(px_header_kernel, dim3(1), dim3(64), 0, stream, totals, pol, hdr);
  hipLaunchKernelGGL(px_place_kernel, dim3(nblocks), dim3(kPxThreads), 0, stream, cost_px, g, tpb, nblocks, scratch, totals + kPxBins, list);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Framebuffer assembly: scatter one part's packed rows into the full image.
// ---------------------------------------------------------------------------------
__global__ void place_part_kernel(const int32_t *part, int32_t *image, int w, int rows_local, int rows_per_tile,
                                  int part_id, int nparts) {
  const size_t total = (size_t)rows_local * w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int lrow = (int)(i / w), col = (int)(i - (size_t)lrow * w);
    const int k = lrow / rows_per_tile;
    const int row = (k * nparts + part_id) * rows_per_tile + (lrow - k * rows_per_tile);
    image[(size_t)row * w + col] = part[i];
  }
}

// All parts at once, of one frame or of a batch: part p's packed rows of frame f start at
// stacked + p * part_stride + f * frame_stride_in (what a gather of the ranks' send buffers to rank 0 delivers; a send
// buffer may carry several frames, or several scenes' frames); frame f's image starts at image + f * frame_stride_out.
// One workgroup per image row at a time: where the row comes from is scalar arithmetic once per row, the copy itself is
// 16-byte loads and stores when the row length and the strides allow it (an element-wise version with four integer
// divisions per pixel moved 2.7 TB/s: 49 us per 4000x4000 frame on rank 0's critical path behind every gather).
__global__ __launch_bounds__(256) void place_all_kernel(const int32_t *stacked, int32_t *image, int w, int h, int rows_per_tile, int nparts,
                                                        size_t part_stride, int nframes, size_t frame_stride_in, size_t frame_stride_out, int vec) {
  const size_t nrows = (size_t)h * (size_t)nframes;
  for (size_t rid = blockIdx.x; rid < nrows; rid += gridDim.x) {
    const int f = (int)(rid / (size_t)h), row = (int)(rid - (size_t)f * h);
    const int t = row / rows_per_tile;
    const int part = t % nparts, k = t / nparts;
    const int lrow = k * rows_per_tile + (row - t * rows_per_tile);
    const int32_t *const src = stacked + (size_t)part * part_stride + (size_t)f * frame_stride_in + (size_t)lrow * w;
    int32_t *const dst = image + (size_t)f * frame_stride_out + (size_t)row * w;
    if (vec) {
      const int4 *const s4 = reinterpret_cast<const int4 *>(src);
      int4 *const d4 = reinterpret_cast<int4 *>(dst);
      for (int i = threadIdx.x; i < (w >> 2); i += 256) d4[i] = s4[i];
    } else {
      for (int i = threadIdx.x; i < w; i += 256) dst[i] = src[i];
    }
  }
}

// ---------------------------------------------------------------------------------
// Host-side launchers
// ---------------------------------------------------------------------------------
hipError_t launch_pixel(const KParams &p, bool stats, hipStream_t stream) {
  const int tiles_y = (p.rows_local + 7) / 8;
  const unsigned grid = (unsigned)(p.tiles_x * tiles_y);
  if (grid == 0) return hipSuccess;
  if (stats) hipLaunchKernelGGL(pixel_kernel<true>, dim3(grid), dim3(64), 0, stream, p);
  else hipLaunchKernelGGL(pixel_kernel<false>, dim3(grid), dim3(64), 0, stream, p);
  return hipGetLastError();
}

// Dynamic LDS above 64 KB is an opt-in per kernel AND per device: remembered per (kernel, device) --
// a process may hold contexts on several GPUs, from several host threads.
static hipError_t allow_full_lds(const void *kfn) {
  constexpr int kMaxDev = 64, kMaxFn = 32;
  static std::mutex mu;
  static const void *fns[kMaxFn];
  static unsigned long long done[kMaxFn];   // bit d: set on device d
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  int slot = -1;
  for (int i = 0; i < kMaxFn && slot < 0; ++i) {
    if (fns[i] == kfn) slot = i;
    else if (fns[i] == nullptr) { fns[i] = kfn; slot = i; }
  }
  if (slot >= 0 && dev < kMaxDev && (done[slot] >> dev & 1ull)) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
  if (slot >= 0 && dev < kMaxDev) done[slot] |= 1ull << dev;
  return hipSuccess;
}

size_t persistent_lds_bytes(int lds_nodes, int lds_sph, int smax, int lmax, int waves_per_wg) {
  return (size_t)lds_nodes * 32 + (size_t)lds_sph * 16 + (size_t)waves_per_wg * (smax + 1 + lmax) * 64 * sizeof(int);
}

template <int THREADS, bool STATS>
static hipError_t launch_persistent_t(const KParams &p, int grid, hipStream_t stream) {
  const size_t lds = persistent_lds_bytes(p.lds_nodes, p.lds_sph, p.smax, p.lmax, THREADS / 64);
  auto kfn = persistent_kernel<THREADS, STATS>;
  if (hipError_t e = allow_full_lds(reinterpret_cast<const void *>(kfn)); e != hipSuccess) return e;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), lds, stream, p);
  return hipGetLastError();
}

hipError_t launch_persistent(const KParams &p, bool stats, int grid, int waves_per_wg, hipStream_t stream) {
  if (grid <= 0) return hipSuccess;
  if (stats) return launch_persistent_t<512, true>(p, grid, stream);
  switch (waves_per_wg) {
  case 4: return launch_persistent_t<256, false>(p, grid, stream);
  case 8: return launch_persistent_t<512, false>(p, grid, stream);
  case 12: return launch_persistent_t<768, false>(p, grid, stream);
  case 16: return launch_persistent_t<1024, false>(p, grid, stream);
  default: return hipErrorInvalidValue;
  }
}

size_t pooled_lds_bytes(int lds_nodes, int lds_sph, int capb, int capl, int ray_planes, int waves_per_wg) {
  return (size_t)lds_nodes * 64 + (size_t)lds_sph * 16 + (size_t)waves_per_wg * pooled_wave_dw(ray_planes, capb, capl) * sizeof(unsigned);
}

template <int THREADS, bool ALL_LDS, bool STATS, bool SOLO = false, int TAIL = 0, bool ORD = false, bool CULL = false, int SPILL = 0>
static hipError_t launch_pooled_t(const KParams &p, int grid, hipStream_t stream) {
  const size_t lds = pooled_lds_bytes(p.lds_nodes, p.lds_sph, p.capb, p.capl, p.ray_planes, THREADS / 64);
  auto kfn = pooled_kernel<THREADS, ALL_LDS, STATS, SOLO, TAIL, ORD, CULL, SPILL>;
  if (hipError_t e = allow_full_lds(reinterpret_cast<const void *>(kfn)); e != hipSuccess) return e;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(THREADS), lds, stream, p);
  return hipGetLastError();
}

// The workgroups of 16 waves: every flavour, with and without CULL (a CULL launch always takes the general scene path: ALL_LDS is
// an instantiation of the un-culled kernel only).
template <bool STATS, bool SOLO, int TAIL, bool ORD>
static hipError_t launch_pooled_16(const KParams &p, bool all_lds, int grid, hipStream_t stream) {
  if (p.cull) return launch_pooled_t<1024, false, STATS, SOLO, TAIL, ORD, true>(p, grid, stream);
  if constexpr (!STATS)
    if (all_lds) return launch_pooled_t<1024, true, STATS, SOLO, TAIL, ORD>(p, grid, stream);
  return launch_pooled_t<1024, false, STATS, SOLO, TAIL, ORD>(p, grid, stream);
}

// (p.cull is honoured by the workgroups of 16 waves -- what every scene within the pooled kernel's limits runs with unless an option
// says otherwise; api.cpp clears it for the other shapes)
hipError_t launch_pooled(const KParams &p, bool stats, int grid, int waves_per_wg, hipStream_t stream) {
  if (grid <= 0) return hipSuccess;
  const bool all_lds = p.lds_nodes == p.n_nodes && p.lds_sph == p.n_sph;
  if (p.cull && waves_per_wg != 16 && waves_per_wg != 4) return hipErrorInvalidValue;
  // (workgroups of four waves, CULL: the plain kernel -- batches and large frames in the shape of five workgroups per CU, api.cpp: make_plan;
  // p.spill: that shape for a tree taller than 15 levels -- the kernels whose box stack may overflow into device memory)
  if (p.spill != nullptr) {
    if (waves_per_wg != 4 || stats || p.px_hdr != nullptr || (p.capb != kSpillCapb && p.capb != kSpillCapbTest)) return hipErrorInvalidValue;
    if (p.capb == kSpillCapbTest)
      return p.cull ? launch_pooled_t<256, false, false, false, 0, false, true, kSpillCapbTest - 64>(p, grid, stream) : launch_pooled_t<256, false, false, false, 0, false, false, kSpillCapbTest - 64>(p, grid, stream);
    return p.cull ? launch_pooled_t<256, false, false, false, 0, false, true, kSpillCapb - 64>(p, grid, stream) : launch_pooled_t<256, false, false, false, 0, false, false, kSpillCapb - 64>(p, grid, stream);
  }
  if (p.cull && waves_per_wg == 4) {
    if (stats || p.px_hdr != nullptr) return hipErrorInvalidValue;
    return launch_pooled_t<256, false, false, false, 0, false, true>(p, grid, stream);
  }
  if (stats && p.px_hdr != nullptr && waves_per_wg == 16)   // (the instrumented launch of a view that renders through its pixel list)
    return p.solo ? launch_pooled_16<true, true, 0, true>(p, all_lds, grid, stream) : launch_pooled_16<true, false, 0, true>(p, all_lds, grid, stream);
  if (stats) return waves_per_wg == 16 ? launch_pooled_16<true, false, 0, false>(p, all_lds, grid, stream) : launch_pooled_t<512, false, true>(p, grid, stream);
  // (SOLO: the instantiation with the solo prologue, for launches whose first tickets are single pixels)
  const bool solo = p.solo && p.nframes == 1 && p.order != nullptr && p.deep_class > 0 && p.deep_split == 6 && p.tl_log2 == kTreeletDepth;
  // (ORD: pixel tickets; workgroups of 16 waves only)
  if (p.px_hdr != nullptr) {
    if (waves_per_wg != 16 || p.nframes != 1) return hipErrorInvalidValue;
    // (p.solo clear: a list without a one-pixel class; p.donate: with the DONATE tail -- a list borrowed from another view)
    if (p.donate) return p.solo ? launch_pooled_16<false, true, 2, true>(p, all_lds, grid, stream) : launch_pooled_16<false, false, 2, true>(p, all_lds, grid, stream);
    return p.solo ? launch_pooled_16<false, true, 0, true>(p, all_lds, grid, stream) : launch_pooled_16<false, false, 0, true>(p, all_lds, grid, stream);
  }
  // (COLD: small ordered single frames; DONATE: the first frame of a view; workgroups of 16 waves only -- other shapes render them with the ordinary kernels)
  if (p.cold && waves_per_wg == 16)
    return solo ? launch_pooled_16<false, true, 1, false>(p, all_lds, grid, stream) : launch_pooled_16<false, false, 1, false>(p, all_lds, grid, stream);
  if (p.donate && waves_per_wg == 16)
    return solo ? launch_pooled_16<false, true, 2, false>(p, all_lds, grid, stream) : launch_pooled_16<false, false, 2, false>(p, all_lds, grid, stream);
  if (waves_per_wg == 16)
    return solo ? launch_pooled_16<false, true, 0, false>(p, all_lds, grid, stream) : launch_pooled_16<false, false, 0, false>(p, all_lds, grid, stream);
#define RT_POOLED_CASE(W)                                                                                               \
  case W:                                                                                                               \
    return all_lds ? (solo ? launch_pooled_t<64 * W, true, false, true>(p, grid, stream) : launch_pooled_t<64 * W, true, false>(p, grid, stream)) \
                   : (solo ? launch_pooled_t<64 * W, false, false, true>(p, grid, stream) : launch_pooled_t<64 * W, false, false>(p, grid, stream));
  switch (waves_per_wg) {
    RT_POOLED_CASE(4)
    RT_POOLED_CASE(8)
    RT_POOLED_CASE(12)
  default: return hipErrorInvalidValue;
  }
#undef RT_POOLED_CASE
}

// Loads this file's code object and resolves the default kernels (HIP loads modules lazily, at
// the first launch: ~0.5 ms that would otherwise land in the first timed frame).
// The instantiations that CALL solo_trace (a real function: SOLO, COLD, DONATE) need a private segment -- 8 bytes per lane, the callee's
// saved register -- and the first dispatch of a queue that needs scratch waits for the runtime to allocate it: ~0.1 ms inside a context's first
// frame (its first launch is a DONATE instantiation).  warm_scratch puts that wait into context creation: a do-nothing kernel with a larger
// private segment, as many waves as the device holds, on the context's stream.
__global__ __launch_bounds__(64) void scratch_warm_kernel(int *sink, int k) {
  volatile int a[8];
  a[k & 7] = k;
  a[(k + 1) & 7] = 1;
  if (a[(k + 2) & 7] == 0x5ca7c4) *sink = 1;
}
hipError_t warm_scratch(hipStream_t stream, int *sink_dev) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipLaunchKernelGGL(scratch_warm_kernel, dim3(cus * 32), dim3(64), 0, stream, sink_dev, 1);
  return hipGetLastError();
}
void warm_render_kernels() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, false>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false>);
  // (the instantiations a view's first frames and its policy may switch to: a first use inside somebody's timed loop is 0.2-0.3 ms)
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, false, 1>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 1>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, false, 2>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 2>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, true, 2>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 2>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, true, 1>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 1>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, true, 0, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 0, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, false, 0, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 0, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, true, 2, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 2, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, true, false, false, 2, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 2, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<256, false, false, false>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<256, false, false, false, 0, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<256, false, false, false, 0, false, false, kSpillCapb - 64>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<256, false, false, false, 0, false, true, kSpillCapb - 64>);
  // ... and their CULL flavours
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 2, true, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 2, true, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 0, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 0, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 1, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 1, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 2, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 2, false, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, false, 0, true, true>);
  (void)hipFuncGetAttributes(&a, (const void *)pooled_kernel<1024, false, false, true, 0, true, true>);
  (void)hipFuncGetAttributes(&a, (const void *)px_count_kernel);
  (void)hipFuncGetAttributes(&a, (const void *)px_scan_kernel);
  (void)hipFuncGetAttributes(&a, (const void *)px_place_kernel);
  (void)hipFuncGetAttributes(&a, (const void *)px_header_kernel);
  (void)hipFuncGetAttributes(&a, (const void *)tile_count_kernel);
  (void)hipFuncGetAttributes(&a, (const void *)tile_scan_kernel);
  (void)hipFuncGetAttributes(&a, (const void *)tile_place_kernel);
}

hipError_t launch_place_part(const int32_t *part, int32_t *image, int w, int rows_local, int rows_per_tile, int part_id,
                             int nparts, hipStream_t stream) {
  const size_t total = (size_t)rows_local * w;
  if (total == 0) return hipSuccess;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(place_part_kernel, dim3(grid), dim3(256), 0, stream, part, image, w, rows_local, rows_per_tile,
                     part_id, nparts);
  return hipGetLastError();
}

hipError_t launch_place_all(const int32_t *stacked, int32_t *image, int w, int h, int rows_per_tile, int nparts,
                            size_t part_stride, hipStream_t stream, int nframes, size_t frame_stride_in, size_t frame_stride_out) {
  const size_t nrows = (size_t)h * (size_t)nframes;
  if (nrows == 0 || w <= 0) return hipSuccess;
  const unsigned grid = (unsigned)(nrows < 16384 ? nrows : 16384);
  const bool vec = w % 4 == 0 && part_stride % 4 == 0 && frame_stride_in % 4 == 0 && frame_stride_out % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(stacked) | reinterpret_cast<uintptr_t>(image)) % 16 == 0;
  hipLaunchKernelGGL(place_all_kernel, dim3(grid), dim3(256), 0, stream, stacked, image, w, h, rows_per_tile, nparts,
                     part_stride, nframes, frame_stride_in, frame_stride_out, vec ? 1 : 0);
  return hipGetLastError();
}

}  // namespace rtk
