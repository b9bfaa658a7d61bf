// bvh_build.hip -- prepare_scene's BVH construction on the GPU: the LBVH of
// futhark/bvh.fut:30-59 + futhark/radixtree.fut:11-72, bit-identical to the reference's
// {L, I} arrays, plus the derived traversal copy the render kernels read.
//
//   centres + min/max   bvh.fut:31-37     (fmin/fmax reductions: order-free for non-NaN input)
//   Morton keys         bvh.fut:38-41, :8-22   (IEEE division, fmax(NaN, 0) = 0 on a flat axis)
//   stable sort by key  bvh.fut:43        = the sort by (key, index): by RANKING in one launch (mid sizes), else LSD radix passes of 4 bits
//   radix tree          radixtree.fut:23-72    one thread per inner node, pure integer
//   AABB propagation    bvh.fut:44-58     EXACTLY floor(log2 n)+2 double-buffered Jacobi sweeps (three composed per launch where launches are the cost)
//   traversal copy      nodes renumbered by depth, treelet by treelet (root levels first = the LDS-staged prefix)
//
// Four size classes: n <= kSmallUse one workgroup, one launch; <= kRankMaxN the RANKED chain of 11 launches; <= kSweepFusedMaxN radix
// passes without count launches + fused sweeps; beyond, a launch per phase and sweep.  Everything is enqueued on the caller's stream; the one host
// round trip is the final synchronisation (tree height and root record arrive in the pinned report block, stored by the last kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include <cmath>
#include <cstdint>

#include "rt_device.hpp"
#include "treelet.h"

namespace rtk {
namespace {

constexpr int kBT = 256;          // threads per block everywhere in this file
constexpr int kSortE = 4;         // sort: elements per thread (tile = 1024 per block)
constexpr int kSortTile = kBT * kSortE;
constexpr int kSortBits = 4, kSortDigits = 1 << kSortBits, kSortWords = kSortDigits / 4;   // sort: bits per pass
constexpr int kRankMaxN = 24576;  // the RANKED sizes (sorts by ranking, one launch each): up to this many spheres

__device__ __forceinline__ float f_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float f_max(float a, float b) { return fmaxf(a, b); }

// ---- centres and their bounds -----------------------------------------------------------
// sphere7 = {pos.xyz, colour.rgb, radius}.  centre = min + 0.5 * (max - min) of sphere_aabb
// (ray.fut:28-30, prim.fut:47-50).
__device__ __forceinline__ void sphere_centre(const float *s, float c[3]) {
  const float r = s[6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mn = s[a] - r, mx = s[a] + r;
    c[a] = mn + 0.5f * (mx - mn);
  }
}

// (also zeroes the nzero counters of the chained sort passes, see sort_scatter_kernel)
__global__ __launch_bounds__(kBT) void centres_minmax_kernel(const float *sph7, int n, float *centres, float *partial, unsigned *zero,
                                                             int nzero, int *flags) {
  __shared__ float red[6][kBT];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * kBT + threadIdx.x; i < nzero; i += gridDim.x * kBT) zero[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < 4) flags[threadIdx.x] = 0;      // (the depth walk's maximum)
  for (int i = blockIdx.x * kBT + threadIdx.x; i < n; i += gridDim.x * kBT) {
    float c[3];
    sphere_centre(sph7 + 7 * (size_t)i, c);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      centres[3 * (size_t)i + a] = c[a];
      lo[a] = f_min(lo[a], c[a]);
      hi[a] = f_max(hi[a], c[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    red[a][threadIdx.x] = lo[a];
    red[3 + a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int s = kBT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = f_min(red[a][threadIdx.x], red[a][threadIdx.x + s]);
        red[3 + a][threadIdx.x] = f_max(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

// the threads' bounds -> red[0..5][0] (min.xyz, max.xyz); ends with a barrier
__device__ __forceinline__ void block_reduce_bounds(const float lo[3], const float hi[3], float (*red)[kBT]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    red[a][threadIdx.x] = lo[a];
    red[3 + a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int s = kBT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = f_min(red[a][threadIdx.x], red[a][threadIdx.x + s]);
        red[3 + a][threadIdx.x] = f_max(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
      }
    }
    __syncthreads();
  }
}
// the blocks' partial bounds -> red[0..5][0], by one block
__device__ __forceinline__ void reduce_partial_bounds(const float *partial, int nblocks, float (*red)[kBT]) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int b = threadIdx.x; b < nblocks; b += kBT)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = f_min(lo[a], partial[b * 6 + a]);
      hi[a] = f_max(hi[a], partial[b * 6 + 3 + a]);
    }
  block_reduce_bounds(lo, hi, red);
}
__global__ __launch_bounds__(kBT) void minmax_final_kernel(const float *partial, int nblocks, float *bounds) {
  __shared__ float red[6][kBT];
  reduce_partial_bounds(partial, nblocks, red);
  if (threadIdx.x < 6) bounds[threadIdx.x] = red[threadIdx.x][0];   // min.xyz, max.xyz
}

// ---- Morton keys (bvh.fut:8-22, :38-41) --------------------------------------------------
__device__ __forceinline__ unsigned spread10(unsigned v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__device__ __forceinline__ unsigned quantise10(float q) { return (unsigned)f_min(f_max(q * 1024.0f, 0.0f), 1023.0f); }

__device__ __forceinline__ unsigned morton_code(float cx, float cy, float cz, const float *bounds) {
  const float c[3] = {cx, cy, cz};
  unsigned code[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mn = bounds[a], mx = bounds[3 + a];
    const float q = (c[a] - mn) / (mx - mn);   // 0/0 = NaN on a flat axis -> 0 below
    code[a] = spread10(quantise10(q));
  }
  return code[0] * 4u + code[1] * 2u + code[2];
}
__device__ __forceinline__ void morton_elem(int i, const float *centres, const float *bounds, unsigned *keys, int *vals) {
  keys[i] = morton_code(centres[3 * (size_t)i], centres[3 * (size_t)i + 1], centres[3 * (size_t)i + 2], bounds);
  vals[i] = i;
}
__global__ __launch_bounds__(kBT) void morton_kernel(const float *centres, const float *bounds, int n, unsigned *keys,
                                                     int *vals) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i < n) morton_elem(i, centres, bounds, keys, vals);
}
// CHAINED flavour for scenes whose sort runs without count launches (sort_scatter_kernel): every block reduces the partial bounds
// itself (the same reduction tree as minmax_final_kernel: the same six floats) and leaves the first pass's digit counts of its
// 256 keys -- a quarter of a sort tile -- in block_counts.
__device__ __forceinline__ void chain_count_block(unsigned digit, bool live, unsigned *block_counts, int nblocks, unsigned *hist) {
  if (threadIdx.x < kSortDigits) hist[threadIdx.x] = 0u;
  __syncthreads();
  if (live) atomicAdd(&hist[digit], 1u);
  __syncthreads();
  if (threadIdx.x < kSortDigits && hist[threadIdx.x])
    atomicAdd(&block_counts[threadIdx.x * nblocks + (int)(blockIdx.x / (kSortTile / kBT))], hist[threadIdx.x]);
}
__global__ __launch_bounds__(kBT) void morton_chain_kernel(const float *centres, const float *partial, int red_blocks, int n,
                                                           unsigned *keys, int *vals, unsigned *block_counts, int nblocks) {
  __shared__ float red[6][kBT];
  __shared__ float s_bounds[6];
  __shared__ unsigned hist[kSortDigits];
  reduce_partial_bounds(partial, red_blocks, red);
  if (threadIdx.x < 6) s_bounds[threadIdx.x] = red[threadIdx.x][0];
  __syncthreads();
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i < n) morton_elem(i, centres, s_bounds, keys, vals);
  chain_count_block(i < n ? keys[i] & (unsigned)(kSortDigits - 1) : 0u, i < n, block_counts, nblocks, hist);
}

// The RANKED sizes' first launch: every block takes the bounds of ALL centres itself (n <= kRankMaxN spheres: 4 floats of each, from L2) and
// writes its 256 keys -- no centres array, no launch for the bounds.  (fmin / fmax: the same six floats in whatever order they are folded.)
__global__ __launch_bounds__(kBT) void morton_all_kernel(const float *__restrict__ sph7, int n, unsigned *keys, int *flags) {
  __shared__ float red[6][kBT];
  __shared__ float s_bounds[6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 8
  for (int j = threadIdx.x; j < n; j += kBT) {
    float c[3];
    sphere_centre(sph7 + 7 * (size_t)j, c);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = f_min(lo[a], c[a]);
      hi[a] = f_max(hi[a], c[a]);
    }
  }
  block_reduce_bounds(lo, hi, red);
  if (threadIdx.x < 6) s_bounds[threadIdx.x] = red[threadIdx.x][0];
  if (blockIdx.x == 0 && threadIdx.x < 4) flags[threadIdx.x] = 0;      // (the depth walk's maximum)
  __syncthreads();
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= n) return;
  float c[3];
  sphere_centre(sph7 + 7 * (size_t)i, c);
  keys[i] = morton_code(c[0], c[1], c[2], s_bounds);
}

// ---- stable LSD radix sort of (key, val), 4 bits per pass ---------------------------------
// Thread t of a block owns kSortE CONSECUTIVE elements, so thread order == element order and a
// block-wide exclusive scan of per-thread digit counts gives stable ranks.  Counts of the 16
// digit values travel packed 16 bits each in four u64 words.
template <int NT = kBT>
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long *total) {
  __shared__ unsigned long long wave_sum[NT / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    if (w < wave) base += wave_sum[w];
    tot += wave_sum[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// digit counts of kSortE elements, 16 bits per digit value, four values per u64 word
struct DigitCounts {
  unsigned long long w[kSortWords];
};
__device__ __forceinline__ void digit_add(DigitCounts &c, unsigned d) {
#pragma unroll
  for (int j = 0; j < kSortWords; ++j) c.w[j] += (d >> 2) == (unsigned)j ? 1ull << (16 * (d & 3u)) : 0ull;
}
__device__ __forceinline__ unsigned digit_get(const DigitCounts &c, unsigned d) {
  unsigned long long v = 0;
#pragma unroll
  for (int j = 0; j < kSortWords; ++j) v = (d >> 2) == (unsigned)j ? c.w[j] : v;
  return (unsigned)((v >> (16 * (d & 3u))) & 0xffffull);
}

__global__ __launch_bounds__(kBT) void sort_count_kernel(const unsigned *keys, int n, int shift, unsigned *block_counts,
                                                         int nblocks) {
  const int base = (blockIdx.x * kBT + threadIdx.x) * kSortE;
  DigitCounts cnt = {};
#pragma unroll
  for (int e = 0; e < kSortE; ++e)
    if (base + e < n) digit_add(cnt, (keys[base + e] >> shift) & (kSortDigits - 1));
  DigitCounts tot;
#pragma unroll
  for (int j = 0; j < kSortWords; ++j) (void)block_excl_scan_u64(cnt.w[j], &tot.w[j]);
  if (threadIdx.x < kSortDigits) block_counts[threadIdx.x * nblocks + blockIdx.x] = digit_get(tot, threadIdx.x);
}

// exclusive scan of m counters by one block (m = 16 * nblocks, digit-major = the order the
// sorted array is laid out in)
// (1024 threads, 8 consecutive counters each per round, in and out of global memory through LDS so that both are coalesced -- one CU serves this
// kernel: 16 strided loads per thread cost it 14 us for the 15 600 counters of a pass over 10^6 keys, a 256-thread block 16 us)
constexpr int kScanNT = 1024, kScanPer = 8, kScanRound = kScanNT * kScanPer;
__global__ __launch_bounds__(kScanNT) void scan_small_kernel(unsigned *data, int m) {
  __shared__ unsigned s_buf[kScanRound + kScanNT];      // counter c of a round at c + c / kScanPer: a thread's 8 start 9 words apart (no bank conflicts)
  unsigned carry = 0;
  for (int start = 0; start < m; start += kScanRound) {
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
      const int c = (int)threadIdx.x + e * kScanNT;
      s_buf[c + c / kScanPer] = start + c < m ? data[start + c] : 0u;
    }
    __syncthreads();
    unsigned v[kScanPer], sum = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
      v[e] = s_buf[(int)threadIdx.x * (kScanPer + 1) + e];
      sum += v[e];
    }
    unsigned long long tot;
    unsigned run = carry + (unsigned)block_excl_scan_u64<kScanNT>(sum, &tot);
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
      s_buf[(int)threadIdx.x * (kScanPer + 1) + e] = run;
      run += v[e];
    }
    carry += (unsigned)tot;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
      const int c = (int)threadIdx.x + e * kScanNT;
      if (start + c < m) data[start + c] = s_buf[c + c / kScanPer];
    }
    __syncthreads();
  }
}

// RAW: block_offsets holds the blocks' digit COUNTS as sort_count_kernel wrote them (no scan launch in between); every block
// derives its own 16 offsets from them -- digit-major exclusive prefix: the elements of smaller digits in all blocks, then
// those of the same digit in the blocks before this one.  For scenes of up to kSortRawBlocks blocks (131 072 elements), where
// a dispatch (~3.5 us of dependent-launch gap) costs more than 16 x nblocks loads per block.
// CHAIN (with RAW): the pass also leaves the NEXT pass's block counts (next_counts, zero before the launch; next_shift its digit):
// the elements of one digit land in consecutive places, at most kSortTile of them, so in at most two tiles of the output; the
// block counts {tile half, digit, next digit} in LDS and adds what is not zero to the output tiles' counters.  Integer adds: any
// order, the same counts.  A chained sort is one launch per pass, its first counts come from the kernel that writes the keys.
constexpr int kSortRawBlocks = 128;
template <bool RAW, bool CHAIN = false>
__global__ __launch_bounds__(kBT) void sort_scatter_kernel(const unsigned *keys_in, const int *vals_in, int n, int shift,
                                                           const unsigned *block_offsets, int nblocks, unsigned *keys_out,
                                                           int *vals_out, unsigned *next_counts = nullptr, int next_shift = 0) {
  static_assert(RAW || !CHAIN, "a chained pass derives its offsets from the raw counts");
  __shared__ unsigned s_all[kSortDigits][kBT / kSortDigits], s_before[kSortDigits][kBT / kSortDigits], s_off[kSortDigits];
  __shared__ unsigned s_next[CHAIN ? 2 * kSortDigits * kSortDigits : 1];
  if (CHAIN)
    for (int j = threadIdx.x; j < 2 * kSortDigits * kSortDigits; j += kBT) s_next[j] = 0u;
  if (RAW) {
    const int d = threadIdx.x & (kSortDigits - 1), c = threadIdx.x / kSortDigits;   // digit, chunk of the block list
    unsigned all = 0, before = 0;
    for (int b = c; b < nblocks; b += kBT / kSortDigits) {
      const unsigned v = block_offsets[d * nblocks + b];
      all += v;
      before += b < (int)blockIdx.x ? v : 0u;
    }
    s_all[d][c] = all;
    s_before[d][c] = before;
    __syncthreads();
    if (threadIdx.x < kSortDigits) {
      unsigned smaller = 0, mine = 0;
      for (int dd = 0; dd < kSortDigits; ++dd)
        for (int cc = 0; cc < kBT / kSortDigits; ++cc) {
          smaller += dd < (int)threadIdx.x ? s_all[dd][cc] : 0u;
          mine += dd == (int)threadIdx.x ? s_before[dd][cc] : 0u;
        }
      s_off[threadIdx.x] = smaller + mine;
    }
    __syncthreads();
  }
  const int base = (blockIdx.x * kBT + threadIdx.x) * kSortE;
  unsigned k[kSortE];
  int v[kSortE];
  DigitCounts cnt = {};
#pragma unroll
  for (int e = 0; e < kSortE; ++e) {
    if (base + e < n) {
      k[e] = keys_in[base + e];
      v[e] = vals_in[base + e];
      digit_add(cnt, (k[e] >> shift) & (kSortDigits - 1));
    }
  }
  DigitCounts rank, tot;   // per digit: elements of this block before this thread
#pragma unroll
  for (int j = 0; j < kSortWords; ++j) rank.w[j] = block_excl_scan_u64(cnt.w[j], &tot.w[j]);
#pragma unroll
  for (int e = 0; e < kSortE; ++e) {
    if (base + e < n) {
      const unsigned d = (k[e] >> shift) & (kSortDigits - 1);
      const unsigned pos = (RAW ? s_off[d] : block_offsets[d * nblocks + blockIdx.x]) + digit_get(rank, d);
      keys_out[pos] = k[e];
      vals_out[pos] = v[e];
      digit_add(rank, d);
      if (CHAIN && next_counts) {
        const unsigned half = pos / kSortTile - s_off[d] / kSortTile, dn = (k[e] >> next_shift) & (kSortDigits - 1);
        atomicAdd(&s_next[(half * kSortDigits + d) * kSortDigits + dn], 1u);
      }
    }
  }
  if (CHAIN && next_counts) {
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * kSortDigits * kSortDigits; j += kBT) {
      const unsigned c = s_next[j];
      if (c) {
        const unsigned half = j / (kSortDigits * kSortDigits), d = (j / kSortDigits) % kSortDigits, dn = j % kSortDigits;
        atomicAdd(&next_counts[dn * nblocks + s_off[d] / kSortTile + half], c);
      }
    }
  }
}

// ---- gather sorted spheres ----------------------------------------------------------------
// (... and the traversal copy's sphere tables {pos, radius} {colour, 1 / radius} in the same launch: they are a function of L alone)
__global__ __launch_bounds__(kBT) void gather_spheres_kernel(const float *sph7, const int *order, int n, float *L7, float4 *sph, float4 *col) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= n) return;
  const float *s = sph7 + 7 * (size_t)order[i];
  float *d = L7 + 7 * (size_t)i;
  float v[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { v[k] = s[k]; d[k] = v[k]; }
  sph[i] = make_float4(v[0], v[1], v[2], v[6]);
  col[i] = make_float4(v[3], v[4], v[5], 1.0f / v[6]);
}

// ---- radix tree (radixtree.fut:13-72) -----------------------------------------------------
__device__ __forceinline__ int delta(const unsigned *L, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned a = L[i], b = L[j];
  if (a == b) return 32 + __clz((unsigned)i ^ (unsigned)j);   // __clz(0) == 32, as u32.clz
  return __clz(a ^ b);
}

// ptr encoding of the canonical arrays: inner i -> i, leaf i -> -2 - i
__device__ __forceinline__ void radix_tree_node(int i, const unsigned *L, int n, int *left, int *right, int *parent) {
  const int diff = delta(L, n, i, i + 1) - delta(L, n, i, i - 1);
  const int d = (diff > 0) - (diff < 0);
  const int dmin = delta(L, n, i, i - d);
  int lmax = 2;
  while (delta(L, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t > 0; t /= 2)
    if (delta(L, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(L, n, i, j);
  int s = 0;
  for (int q = 1; q <= l; q *= 2) {
    const int t = (l + 2 * q - 1) / (2 * q);
    if (delta(L, n, i, i + (s + t) * d) > dnode) s += t;
  }
  const int gamma = i + s * d + min(d, 0);
  if (min(i, j) == gamma) {
    left[i] = -2 - gamma;
  } else {
    left[i] = gamma;
    parent[gamma] = i;
  }
  if (max(i, j) == gamma + 1) {
    right[i] = -2 - (gamma + 1);
  } else {
    right[i] = gamma + 1;
    parent[gamma + 1] = i;
  }
}
// the four fills ahead of the tree and the sweeps in one dispatch: parent = -1, both "previous" box buffers = 0, fin = 0x7f7f7f7f
__global__ __launch_bounds__(kBT) void build_fills_kernel(int *parent, float *pmin, float *pmax, int *fin, int ni) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= ni) return;
  parent[i] = -1;
  fin[i] = 0x7f7f7f7f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pmin[3 * i + k] = 0.0f;
    pmax[3 * i + k] = 0.0f;
  }
}

// The sorted spheres AND the radix tree in one launch (the fused-sweep sizes): thread i gathers sphere i and builds inner node i.  Nothing is
// filled ahead of it: every inner node but the root is some node's child and gets its parent link from it, the root is node 0.
// LDSKEYS (the ranked sizes): all sorted keys staged in LDS first -- a node's ~2 log2(n) dependent key reads cost LDS latency, not L2's.
template <bool LDSKEYS>
__global__ __launch_bounds__(kBT) void tree_kernel(const unsigned *__restrict__ L, const int *__restrict__ order, const float *__restrict__ sph7,
                                                   int n, float *__restrict__ L7, float4 *__restrict__ sph, float4 *__restrict__ col,
                                                   int *left, int *right, int *parent) {
  extern __shared__ unsigned s_L[];
  const int i = blockIdx.x * kBT + threadIdx.x;
  float v[7];
  if (i < n) {
    const float *src = sph7 + 7 * (size_t)order[i];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = src[k];
  }
  if (LDSKEYS) {
    constexpr int kPer = kRankMaxN / kBT;
    unsigned w[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int j = (int)threadIdx.x + u * kBT;
      if (j < n) w[u] = L[j];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int j = (int)threadIdx.x + u * kBT;
      if (j < n) s_L[j] = w[u];
    }
    __syncthreads();
  }
  if (i >= n) return;
  float *d = L7 + 7 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 7; ++k) d[k] = v[k];
  sph[i] = make_float4(v[0], v[1], v[2], v[6]);
  col[i] = make_float4(v[3], v[4], v[5], 1.0f / v[6]);
  if (i == 0) parent[0] = -1;
  if (i < n - 1) radix_tree_node(i, LDSKEYS ? s_L : L, n, left, right, parent);
}
__global__ __launch_bounds__(kBT) void radix_tree_kernel(const unsigned *L, int n, int *left, int *right, int *parent) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i < n - 1) radix_tree_node(i, L, n, left, right, parent);
}

// ---- AABB propagation: one Jacobi sweep (bvh.fut:48-58) -----------------------------------
__device__ __forceinline__ void aabb_sweep_node(int i, const float *L7, const int *left, const int *right,
                                                const float *pmin, const float *pmax, float *cmin, float *cmax) {
  float mn[2][3], mx[2][3];
  const int kid[2] = {left[i], right[i]};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (kid[k] <= -2) {
      const float *s = L7 + 7 * (size_t)(-2 - kid[k]);
      const float r = s[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mn[k][a] = s[a] - r;
        mx[k][a] = s[a] + r;
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mn[k][a] = pmin[3 * (size_t)kid[k] + a];
        mx[k][a] = pmax[3 * (size_t)kid[k] + a];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    cmin[3 * (size_t)i + a] = f_min(mn[0][a], mn[1][a]);   // enclosing, prim.fut:38-45
    cmax[3 * (size_t)i + a] = f_max(mx[0][a], mx[1][a]);
  }
}
// fin[i] = the sweep in which node i first computed its FINAL box (children final one sweep
// earlier, or leaves); INT_MAX until then.  Once that value sits in both ping-pong buffers
// (sweeps fin and fin + 1) recomputing it would store the same bits again: the thread returns.
__global__ __launch_bounds__(kBT) void aabb_sweep_kernel(const float *L7, const int *left, const int *right, int ni,
                                                         const float *pmin, const float *pmax, float *cmin, float *cmax,
                                                         int *fin, int sweep) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= ni) return;
  const int mine = fin[i];
  if (mine <= sweep - 2) return;
  aabb_sweep_node(i, L7, left, right, pmin, pmax, cmin, cmax);
  if (mine > sweep) {
    // (a child finishing in this very sweep stores `sweep`, which is not < sweep: no race)
    const int kl = left[i], kr = right[i];
    const bool fl = kl <= -2 || fin[kl] < sweep, fr = kr <= -2 || fin[kr] < sweep;
    if (fl && fr) fin[i] = sweep;
  }
}

// Several Jacobi sweeps in ONE launch (mid-size scenes, where the build is a chain of ~4.4 us dispatches and 15-19 of them are sweeps):
// box_{s+K}(i) is a pure function of the boxes K sweeps earlier at most K levels below i -- enclosing(box_{s+K-1}(left), box_{s+K-1}(right)), each of
// those the same one level down, a leaf child contributing its sphere's box at every level -- and enclosing is componentwise fmin / fmax: no rounding,
// so composing K levels in one thread gives the very bits K launches would.  Reads the buffer of sweep s only (complete: written by an earlier launch),
// writes sweep s + K for every node.
struct Box3 { float lo[3], hi[3]; };
// (FIRST: the launch that starts the sweeps -- "the boxes of sweep 0" are all zero, nothing is read and nothing had to be filled)
template <int K, bool FIRST>
__device__ __forceinline__ Box3 box_after(int ptr, const float *L7, const int *left, const int *right, const float *pmin, const float *pmax) {
  Box3 b;
  if (ptr <= -2) {                      // a leaf: sphere_aabb (ray.fut:28-30), whatever the sweep
    const float *s = L7 + 7 * (size_t)(-2 - ptr);
    const float r = s[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) { b.lo[a] = s[a] - r; b.hi[a] = s[a] + r; }
    return b;
  }
  if constexpr (K == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      b.lo[a] = FIRST ? 0.0f : pmin[3 * (size_t)ptr + a];
      b.hi[a] = FIRST ? 0.0f : pmax[3 * (size_t)ptr + a];
    }
    return b;
  } else {
    const Box3 l = box_after<K - 1, FIRST>(left[ptr], L7, left, right, pmin, pmax), r = box_after<K - 1, FIRST>(right[ptr], L7, left, right, pmin, pmax);
#pragma unroll
    for (int a = 0; a < 3; ++a) { b.lo[a] = f_min(l.lo[a], r.lo[a]); b.hi[a] = f_max(l.hi[a], r.hi[a]); }   // enclosing, prim.fut:38-45
    return b;
  }
}
// (... the FIRST launch also walks every node up to the root -- depth_walk_kernel's work, which needs the parent links and nothing of the sweeps)
template <int K, bool FIRST>
__global__ __launch_bounds__(kBT) void aabb_sweepk_kernel(const float *L7, const int *left, const int *right, int ni, const float *pmin, const float *pmax,
                                                          float *cmin, float *cmax, const int *parent, unsigned *depth_keys, int *maxdepth) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i < ni) {
    const Box3 b = box_after<K, FIRST>(i, L7, left, right, pmin, pmax);
#pragma unroll
    for (int a = 0; a < 3; ++a) { cmin[3 * (size_t)i + a] = b.lo[a]; cmax[3 * (size_t)i + a] = b.hi[a]; }
  }
  if (FIRST && parent) {
    int d = 0;
    if (i < ni) {
      for (int p = parent[i]; p >= 0; p = parent[p]) ++d;
      depth_keys[i] = (unsigned)d;
    }
    for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o));
    if ((threadIdx.x & 63) == 0) atomicMax(maxdepth, d);
  }
}
constexpr int kSweepLevels = 3;        // sweeps per launch of the mid-size path (measured at 10^4 spheres: 3 levels 8.4 us a launch, 5 levels 24-29 us)
constexpr int kSweepFusedMaxN = 131072;   // ... used up to this many spheres (beyond: one sweep per launch, with the nodes that are final dropping out)

// ---- node depths (for the traversal numbering) ----------------------------------------------
// depth of every inner node = number of ancestors (walk up the parent links), written as the
// sort key of the traversal numbering; also the maximum depth
// (block_counts: the first digit counts of a chained sort, see morton_chain_kernel; nullptr = none)
__global__ __launch_bounds__(kBT) void depth_walk_kernel(const int *parent, int ni, unsigned *keys, int *vals, int *maxdepth,
                                                         unsigned *block_counts, int nblocks) {
  __shared__ unsigned hist[kSortDigits];
  const int i = blockIdx.x * kBT + threadIdx.x;
  int d = 0;
  if (i < ni) {
    for (int p = parent[i]; p >= 0; p = parent[p]) ++d;
    keys[i] = (unsigned)d;
    vals[i] = i;
  }
  if (block_counts) chain_count_block((unsigned)d & (unsigned)(kSortDigits - 1), i < ni, block_counts, nblocks, hist);
  for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o));
  if ((threadIdx.x & 63) == 0) atomicMax(maxdepth, d);
}

// ---- treelet-major numbering (treelet.h, cut of kTreeletDepth levels) ---------------------------
// From the numbering by depth (order[t] = canonical node, depth_sorted[t] its depth, trav_of its inverse): a node whose depth is a
// multiple of D and its descendants of relative depth < D become neighbours, in the order of their heap indices inside the treelet
// (root 0, children of h at 2h + 1 and 2h + 2), compacted; the treelets keep the order of their roots, so the nodes nearest the root
// still form a prefix.  size[t] = nodes of the treelet rooted at the t-th node (0 for a node that is no root); its exclusive scan is
// where each treelet starts.  A treelet has at most 2^D - 1 <= 15 nodes: every node recomputes what it needs of its own treelet
// from the child links (no per-treelet arrays).
// occupancy of the treelet rooted at canonical node r: bit h = an inner node sits at heap index h
__device__ __forceinline__ uint32_t treelet_occ(int r, const int *left, const int *right) {
  constexpr int D = kTreeletDepth, kInner = (1 << (D - 1)) - 1;   // heap indices whose children lie inside the treelet
  uint32_t occ = 1u;
  int node[kInner > 0 ? kInner : 1];
  node[0] = r;
#pragma unroll
  for (int h = 0; h < kInner; ++h) {
    if (!((occ >> h) & 1u)) continue;
    const int c = node[h], l = left[c], rr = right[c];
    if (l >= 0) {
      occ |= 1u << (2 * h + 1);
      if (2 * h + 1 < kInner) node[2 * h + 1] = l;
    }
    if (rr >= 0) {
      occ |= 1u << (2 * h + 2);
      if (2 * h + 2 < kInner) node[2 * h + 2] = rr;
    }
  }
  return occ;
}
// canonical node c at relative depth `rel` of its treelet: the treelet's root and c's heap index (the step nearest the root is the most
// significant path bit -- host_build.cpp: make_trav_layout)
__device__ __forceinline__ int treelet_root(int c, int rel, const int *parent, const int *right, int *heap) {
  int a = c, path = 0;
  for (int s = 0; s < rel; ++s) {
    const int p = parent[a];
    path |= (right[p] == a ? 1 : 0) << s;
    a = p;
  }
  *heap = (1 << rel) - 1 + path;
  return a;
}
// (... and the inverse of the numbering by depth, trav_of[order[t]] = t, in the same launch)
__global__ __launch_bounds__(kBT) void treelet_size_kernel(const unsigned *depth_sorted, const int *order, const int *left,
                                                           const int *right, int ni, unsigned *size, int *trav_of) {
  const int t = blockIdx.x * kBT + threadIdx.x;
  if (t >= ni) return;
  const int c = order[t];
  trav_of[c] = t;
  size[t] = (depth_sorted[t] % (unsigned)kTreeletDepth) ? 0u : (unsigned)tl_popc(treelet_occ(c, left, right));
}
constexpr int kScanE = 4;   // elements per thread of the multi-block scan
__global__ __launch_bounds__(kBT) void scan_sums_kernel(const unsigned *v, int n, unsigned *sums) {
  const int i0 = (blockIdx.x * kBT + threadIdx.x) * kScanE;
  unsigned sum = 0;
#pragma unroll
  for (int e = 0; e < kScanE; ++e) sum += i0 + e < n ? v[i0 + e] : 0u;
  unsigned long long tot;
  (void)block_excl_scan_u64(sum, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = (unsigned)tot;
}
__global__ __launch_bounds__(kBT) void scan_apply_kernel(unsigned *v, int n, const unsigned *sums_excl) {   // in place, exclusive
  const int i0 = (blockIdx.x * kBT + threadIdx.x) * kScanE;
  unsigned x[kScanE], sum = 0;
#pragma unroll
  for (int e = 0; e < kScanE; ++e) {
    x[e] = i0 + e < n ? v[i0 + e] : 0u;
    sum += x[e];
  }
  unsigned long long tot;
  unsigned run = sums_excl[blockIdx.x] + (unsigned)block_excl_scan_u64(sum, &tot);
#pragma unroll
  for (int e = 0; e < kScanE; ++e) {
    if (i0 + e < n) v[i0 + e] = run;
    run += x[e];
  }
}
// a node's new index (+ its heap index in its treelet, packed: treelet.h) from the treelet starts
// (rel: the node's depth inside its treelet, depth % kTreeletDepth)
__device__ __forceinline__ unsigned treelet_place_rel(int c, int rel, const unsigned *start, const int *trav_by_depth, const int *parent,
                                                     const int *left, const int *right) {
  int heap;
  const int r = treelet_root(c, rel, parent, right, &heap);
  return tl_pack_place(start[trav_by_depth[r]] + (unsigned)tl_pos(treelet_occ(r, left, right), heap), heap);
}
__device__ __forceinline__ unsigned treelet_place(int c, const unsigned *depth_sorted, const unsigned *start, const int *trav_by_depth,
                                                 const int *parent, const int *left, const int *right) {
  return treelet_place_rel(c, (int)(depth_sorted[trav_by_depth[c]] % (unsigned)kTreeletDepth), start, trav_by_depth, parent, left, right);
}
// the two mask dwords of canonical node c, whose packed place is `place` (treelet.h: tl_masks)
__device__ __forceinline__ TlMasks treelet_masks(int c, unsigned place, const int *parent, const int *left, const int *right) {
  const int heap = tl_place_heap(place);
  int h2;
  const int r = treelet_root(c, tl_heap_level(heap), parent, right, &h2);
  return tl_masks(treelet_occ(r, left, right), heap, kTreeletDepth);
}
__global__ __launch_bounds__(kBT) void treelet_number_kernel(const unsigned *depth_sorted, const unsigned *start, const int *trav_by_depth,
                                                             const int *parent, const int *left, const int *right, int ni,
                                                             int *place, int *order) {
  const int c = blockIdx.x * kBT + threadIdx.x;
  if (c >= ni) return;
  const unsigned pl = treelet_place(c, depth_sorted, start, trav_by_depth, parent, left, right);
  place[c] = (int)pl;
  order[pl & kTlIndexMask] = c;
}


// ---- mid-size scenes (kSmallUse < n <= kRankMaxN): sorts by ranking, scans by every block -------------------------------------
// Here a launch is ~4.5 us of dependent-dispatch latency and a radix-sort pass ~8 us, whatever it does: the build is priced in
// LAUNCHES.  A stable sort by key is the sort by (key, index), so an element's place is the number of elements whose (key, index)
// is smaller, and that can be had in ONE launch: every block stages all n keys in LDS and buckets them by their leading bits
// (a histogram, its scan, the members' indices listed bucket by bucket -- the same work in every block, ~10^4 LDS operations);
// an element's place is then its bucket's start plus the number of smaller (key, index) among the bucket's members, counted by
// 8 lanes per element, 64 elements per block.  The bucket is a monotone function of (key, index), so any distribution is sorted
// right; one bucket holding everything (all keys equal) costs n / 8 compares per lane, ~10 us.  (All pairs against all tiles -- a v_cmp
// and the population count of its mask per element and tile -- was tried first: the scalar unit waits ~25 cycles for every mask,
// 34 us for 10^4 keys.)
// DEPTH: the keys are node depths, vals the nodes (the numbering by depth); the lane that places node c also writes the inverse
// numbering and, for a treelet root, its treelet's size (treelet_size_kernel).
constexpr int kRankNT = 512, kRankBuckets = 2 * kRankNT, kRankLanes = 8, kRankPerBlock = kRankNT / kRankLanes;
// the bucket of (key, index): monotone in that pair's order.  Morton keys: their leading 10 bits.  Depths (< 64, beyond: one last
// bucket row): 16 slices of the index range per depth -- a level of the tree holds up to half the nodes.
template <bool DEPTH>
__device__ __forceinline__ unsigned rank_bucket(unsigned key, int j, int slice_shift) {
  return DEPTH ? min(key, 63u) * 16u + ((unsigned)j >> slice_shift) : min(key >> 20, (unsigned)kRankBuckets - 1u);
}
inline size_t rank_sort_lds_bytes(int n) { return (sizeof(unsigned) + sizeof(unsigned short)) * 64 * (size_t)((n + 63) / 64); }
template <bool DEPTH>
__global__ __launch_bounds__(kRankNT) void rank_sort_kernel(const unsigned *__restrict__ keys_in, int n, unsigned *__restrict__ keys_out,
                                                            int *__restrict__ vals_out, const int *__restrict__ left,
                                                            const int *__restrict__ right, unsigned *__restrict__ size,
                                                            int *__restrict__ trav_of) {
  extern __shared__ unsigned s_keys[];          // n keys (padded to whole tiles of 64), then the bucket lists: n indices of 16 bits
  __shared__ unsigned s_hist[kRankBuckets], s_off[kRankBuckets], s_wave[kRankNT / 64];
  const int npad = 64 * ((n + 63) / 64);
  const int slice_shift = max(0, 32 - __clz(n - 1) - 4);      // index >> slice_shift < 16
  unsigned short *s_list = reinterpret_cast<unsigned short *>(s_keys + npad);
  constexpr int kPer = kRankMaxN / kRankNT;     // elements per thread of the bucketing phases: j = thread + u * kRankNT
  {   // staging: all loads in flight before the first LDS write
    unsigned v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int j = (int)threadIdx.x + u * kRankNT;
      if (j < n) v[u] = keys_in[j];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int j = (int)threadIdx.x + u * kRankNT;
      if (j < n) s_keys[j] = v[u];
    }
  }
  s_hist[threadIdx.x] = 0u;
  s_hist[threadIdx.x + kRankNT] = 0u;
  __syncthreads();
  unsigned pos[kPer];                           // the element's place among its bucket's members (any order: the count below looks at indices)
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int j = (int)threadIdx.x + u * kRankNT;
    if (j < n) pos[u] = atomicAdd(&s_hist[rank_bucket<DEPTH>(s_keys[j], j, slice_shift)], 1u);
  }
  __syncthreads();
  {                                             // exclusive scan of the histogram, two buckets per thread
    const unsigned h0 = s_hist[2 * threadIdx.x], h1 = s_hist[2 * threadIdx.x + 1];
    unsigned incl = h0 + h1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o);
      if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned base = incl - h0 - h1;
#pragma unroll
    for (int w = 0; w < kRankNT / 64; ++w) base += w < (int)(threadIdx.x >> 6) ? s_wave[w] : 0u;
    s_off[2 * threadIdx.x] = base;
    s_off[2 * threadIdx.x + 1] = base + h0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int j = (int)threadIdx.x + u * kRankNT;
    if (j < n) s_list[s_off[rank_bucket<DEPTH>(s_keys[j], j, slice_shift)] + pos[u]] = (unsigned short)j;
  }
  __syncthreads();
  const int sub = threadIdx.x & (kRankLanes - 1), i = min((int)blockIdx.x * kRankPerBlock + (int)(threadIdx.x / kRankLanes), n - 1);
  const unsigned ki = s_keys[i], bk = rank_bucket<DEPTH>(ki, i, slice_shift), beg = s_off[bk], end = beg + s_hist[bk];
  unsigned cnt = 0;
  for (unsigned m = beg + sub; m < end; m += kRankLanes) {
    const int j = s_list[m];
    const unsigned kj = s_keys[j];
    cnt += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 1; o < kRankLanes; o <<= 1) cnt += __shfl_xor(cnt, o);
  if (sub != 0 || (int)blockIdx.x * kRankPerBlock + (int)(threadIdx.x / kRankLanes) >= n) return;
  const unsigned r = beg + cnt;
  keys_out[r] = ki;
  vals_out[r] = i;
  if (DEPTH) {
    trav_of[i] = (int)r;
    size[r] = (ki % (unsigned)kTreeletDepth) ? 0u : (unsigned)tl_popc(treelet_occ(i, left, right));
  }
}
// treelet_number_kernel whose every block scans the treelet sizes itself (into LDS) instead of three scan launches ahead of it
constexpr int kNumNT = 1024;
__global__ __launch_bounds__(kNumNT) void treelet_number_scan_kernel(const unsigned *depth_by_node, const unsigned *size, const int *trav_by_depth,
                                                                     const int *parent, const int *left, const int *right, int ni, int *place,
                                                                     int *order) {
  extern __shared__ unsigned s_start[];     // exclusive scan of size[0 .. ni), padded to whole rounds of 4 * kNumNT
  constexpr int kRounds = (kRankMaxN + 4 * kNumNT - 1) / (4 * kNumNT);
  uint4 x[kRounds];
#pragma unroll
  for (int u = 0; u < kRounds; ++u) {       // (size[] is a scratch piece of >= 3 * ni floats: whole quads are readable)
    const int i0 = (u * kNumNT + (int)threadIdx.x) * 4;
    x[u] = make_uint4(0u, 0u, 0u, 0u);
    if (i0 < ni) {
      x[u] = reinterpret_cast<const uint4 *>(size)[i0 / 4];
      x[u].y = i0 + 1 < ni ? x[u].y : 0u;
      x[u].z = i0 + 2 < ni ? x[u].z : 0u;
      x[u].w = i0 + 3 < ni ? x[u].w : 0u;
    }
  }
  unsigned carry = 0;
#pragma unroll
  for (int u = 0; u < kRounds; ++u) {
    if (u * kNumNT * 4 >= ni) break;
    unsigned long long tot;
    const unsigned run = carry + (unsigned)block_excl_scan_u64<kNumNT>(x[u].x + x[u].y + x[u].z + x[u].w, &tot);
    reinterpret_cast<uint4 *>(s_start)[u * kNumNT + (int)threadIdx.x] = make_uint4(run, run + x[u].x, run + x[u].x + x[u].y, run + x[u].x + x[u].y + x[u].z);
    carry += (unsigned)tot;
  }
  __syncthreads();
  const int c = blockIdx.x * kNumNT + threadIdx.x;
  if (c >= ni) return;
  // (the depths by canonical node are still where the sweeps' first launch wrote them: one load instead of two dependent ones)
  const unsigned pl = treelet_place_rel(c, (int)(depth_by_node[c] % (unsigned)kTreeletDepth), s_start, trav_by_depth, parent, left, right);
  place[c] = (int)pl;
  order[pl & kTlIndexMask] = c;
}

// ---- traversal copy ---------------------------------------------------------------------------
__device__ __forceinline__ void trav_node(int t, const int *order, const int *trav_of, const int *parent, const int *left, const int *right,
                                          const float *bmin, const float *bmax, float4 *nodes32, float4 *nodes64) {
  const int c = order[t];
  const int kid[2] = {left[c], right[c]};
  int ref[2];
  float4 q[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                 make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (kid[k] <= -2) {
      ref[k] = ~(-2 - kid[k]);
    } else {
      ref[k] = (int)((unsigned)trav_of[kid[k]] & kTlIndexMask);
      const float *mn = bmin + 3 * (size_t)kid[k], *mx = bmax + 3 * (size_t)kid[k];
      q[2 * k] = make_float4(mn[0], mn[1], mn[2], 0.f);
      q[2 * k + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
  }
  q[0].w = __int_as_float((int)((unsigned)ref[0] << 8));   // pre-shifted: a pooled work item is (reference << 8) | (slot * 4)
  q[1].w = __int_as_float((int)((unsigned)ref[1] << 8));
  const TlMasks tm = treelet_masks(c, (unsigned)trav_of[c], parent, left, right);   // the node's place in its treelet (treelet.h)
  q[2].w = __int_as_float((int)tm.l);
  q[3].w = __int_as_float((int)tm.r);
  nodes64[4 * (size_t)t + 0] = q[0];
  nodes64[4 * (size_t)t + 1] = q[1];
  nodes64[4 * (size_t)t + 2] = q[2];
  nodes64[4 * (size_t)t + 3] = q[3];
  const float *mn = bmin + 3 * (size_t)c, *mx = bmax + 3 * (size_t)c;
  nodes32[2 * (size_t)t + 0] = make_float4(mn[0], mn[1], mn[2], __int_as_float(ref[0]));
  nodes32[2 * (size_t)t + 1] = make_float4(mx[0], mx[1], mx[2], __int_as_float(ref[1]));
}
// (report: the host-pinned block of gpu_build_pinned_bytes() -- the thread of the root's record, t == 0, writes the maximum depth and that record
// straight into it, which the two copies behind the launch used to do; nullptr = no report)
__global__ __launch_bounds__(kBT) void trav_nodes_kernel(const int *order, const int *trav_of, const int *parent, const int *left,
                                                         const int *right, const float *bmin, const float *bmax, int ni,
                                                         float4 *nodes32, float4 *nodes64, const int *maxdepth, int *report) {
  const int t = blockIdx.x * kBT + threadIdx.x;
  if (t >= ni) return;
  trav_node(t, order, trav_of, parent, left, right, bmin, bmax, nodes32, nodes64);
  if (t == 0 && report) {
    report[0] = *maxdepth;
    reinterpret_cast<float4 *>(report + 4)[0] = nodes32[0];
    reinterpret_cast<float4 *>(report + 4)[1] = nodes32[1];
  }
}


// ---- small scenes: the whole build in ONE workgroup, one launch ---------------------------------
// For n <= kSmallMax everything above runs inside a single 1024-thread workgroup with
// __syncthreads() between the phases (the reference scenes have 400 and 10 000 spheres: a chain
// of ~125 tiny launches costs more in launch gaps than in work).  Same arithmetic, same arrays.
// Sort keys/values, the sorted Morton keys (the radix tree's binary searches), parent links and
// the traversal numbering live in LDS (2 x 4n bytes <= 128 KB); boxes stay in global memory (L2).
constexpr int kSmallNT = 1024;
constexpr int kSmallMax = 16384;   // digit totals fit the packed 16-bit counters
// ... and it is USED up to kSmallUse spheres: one workgroup takes 0.062 ms at 400 spheres and 0.076 at 768 (its key sort by ranking, small_sort), the ranked
// chain of 11 launches 0.074-0.080 / 0.083; at 800 .. 1024 the two are level, beyond 1024 the one workgroup sorts by LSD passes again: 0.125 ms at 2000, 0.26 at
// 6144 against 0.091 / 0.110 (profiles/r06/exp/e8_prepare_sizes.txt; until round 6 the chain was ~45 launches, 0.25 ms, and the crossover 6144)
constexpr int kSmallUse = 768;
inline int small_use() {          // (RT_BVH_SMALL_USE: a measurement aid, the crossover between the one-workgroup build and the ranked chain)
  static const int v = [] {
    const char *e = getenv("RT_BVH_SMALL_USE");
    const int x = e ? atoi(e) : kSmallUse;
    return x < 2 ? 2 : (x > kSmallMax ? kSmallMax : x);
  }();
  return v;
}
constexpr int kSmallE = 17;        // consecutive elements a thread owns in a sort pass (odd: LDS stride)

struct SmallArgs {
  const float *sph7;      // [n][7] input spheres (device memory: a scene is device resident)
  int n, sweeps;
  GpuBvhOut o;
  float *centres;   // [n][3]
  float4 *box4;     // [2][n-1][2]: ping-pong boxes {lo.xyz, final?} {hi.xyz, -}
  int *order, *trav_of;   // [n-1] traversal numbering and its inverse
  int *result;      // host-pinned: [0] max depth, [4..11] the root's traversal record
};

// inclusive add-scan across the 64 lanes of a wave (DPP row shifts + row broadcasts)
__device__ __forceinline__ unsigned wave_incl_scan_add(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// Stable LSD sort of (lk[i], lv[i]), i < m, by key bits [0, bits): 2 bits per pass.  Thread t owns
// elements [t*E, (t+1)*E) in registers between passes; LDS is the exchange buffer.  Digit counts
// travel as four 16-bit fields (two per dword: m <= 16384, so no field ever carries).
// (m <= kSmallNT, which is every scene this kernel is USED for since round 6: by ranking -- a thread per element counts the (key, index) pairs below its
// own over the m keys in LDS, read at wave-uniform addresses; rgbbox, 400 keys: 6.8 us where 15 passes took 25, prepare_scene 0.082 -> 0.060 ms)
__device__ __forceinline__ void small_sort(unsigned *lk, unsigned *lv, int m, int bits) {
  if (m <= kSmallNT && bits > 8) {      // (a 6-bit key -- the depths -- is three cheap passes below)
    const int t = threadIdx.x;
    const unsigned k = t < m ? lk[t] : 0u, v = t < m ? lv[t] : 0u;
    unsigned r = 0;
    const int w0 = t & ~63;               // this wave's elements are [w0, w0 + 64)
    if (w0 < m) {                         // (a wave without elements only keeps the barriers)
      int j = 0;
      for (; j + 16 <= m; j += 16) {      // 16 keys per step: four 16-byte reads in flight, at wave-uniform addresses
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = reinterpret_cast<const uint4 *>(lk + j)[u];
        if (j + 16 <= w0 || j >= w0 + 64) {       // all 16 before (key <= k counts) or behind (key < k) every element of the wave: one compare per pair
          const unsigned thr = k + (j < w0 ? 1u : 0u);      // keys < 2^30: no wrap
#pragma unroll
          for (int u = 0; u < 4; ++u) r += (q[u].x < thr ? 1u : 0u) + (q[u].y < thr ? 1u : 0u) + (q[u].z < thr ? 1u : 0u) + (q[u].w < thr ? 1u : 0u);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned kk[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) r += (kk[e] < k || (kk[e] == k && j + 4 * u + e < t)) ? 1u : 0u;
          }
        }
      }
      for (; j < m; ++j) {
        const unsigned kj = lk[j];
        r += (kj < k || (kj == k && j < t)) ? 1u : 0u;
      }
    }
    __syncthreads();
    if (t < m) {
      lk[r] = k;
      lv[r] = v;
    }
    __syncthreads();
    return;
  }
  __shared__ uint2 wave_tot[2][kSmallNT / 64];
  const int E = ((m + kSmallNT - 1) / kSmallNT) | 1;   // odd: conflict-free stride
  const int base = threadIdx.x * E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned k[kSmallE], v[kSmallE];
#pragma unroll
  for (int e = 0; e < kSmallE; ++e)
    if (e < E && base + e < m) {
      k[e] = lk[base + e];
      v[e] = lv[base + e];
    }
  int par = 0;
  for (int shift = 0; shift < bits; shift += 2, par ^= 1) {
    unsigned c01 = 0, c23 = 0;   // counts of digits {0,1} and {2,3}
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < m) {
        const unsigned d = (k[e] >> shift) & 3u;
        const unsigned one = 1u << ((d & 1u) * 16);
        c01 += d < 2 ? one : 0u;
        c23 += d < 2 ? 0u : one;
      }
    const unsigned i01 = wave_incl_scan_add(c01), i23 = wave_incl_scan_add(c23);
    if (lane == 63) wave_tot[par][wave] = make_uint2(i01, i23);
    __syncthreads();   // (also: every thread's chunk is in registers before anyone scatters)
    unsigned b01 = 0, b23 = 0, t01 = 0, t23 = 0;
#pragma unroll
    for (int w = 0; w < kSmallNT / 64; ++w) {
      const uint2 t = wave_tot[par][w];
      b01 += w < wave ? t.x : 0u;
      b23 += w < wave ? t.y : 0u;
      t01 += t.x;
      t23 += t.y;
    }
    // exclusive rank of this thread's first element of each digit, plus where the digit starts
    const unsigned n0 = t01 & 0xffffu, n1 = t01 >> 16, n2 = t23 & 0xffffu;
    unsigned r01 = b01 + i01 - c01 + (n0 << 16);
    unsigned r23 = b23 + i23 - c23 + (n0 + n1) + ((n0 + n1 + n2) << 16);
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < m) {
        const unsigned d = (k[e] >> shift) & 3u;
        const unsigned sh = (d & 1u) * 16;
        const unsigned pos = ((d < 2 ? r01 : r23) >> sh) & 0xffffu;
        lk[pos] = k[e];
        lv[pos] = v[e];
        r01 += d < 2 ? 1u << sh : 0u;
        r23 += d < 2 ? 0u : 1u << sh;
      }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < m) {
        k[e] = lk[base + e];
        v[e] = lv[base + e];
      }
  }
}

__global__ __launch_bounds__(kSmallNT) void bvh_small_kernel(SmallArgs a) {
  extern __shared__ unsigned small_lds[];
  __shared__ float s_red[kSmallNT / 64][6];
  __shared__ float s_bounds[8];
  __shared__ int s_maxdepth;
  const int tid = threadIdx.x, n = a.n, ni = a.n - 1;
  unsigned *lk = small_lds, *lv = small_lds + n;
  // phase timestamps (100 MHz) for RT_BVH_STAMPS=1, written straight into the pinned result block
  unsigned long long *stamps = (unsigned long long *)(a.result + 16);
#define STAMP(k) do { if (tid == 0) stamps[k] = wall_clock64(); } while (0)
  STAMP(0);
  // 1. centres and their bounds
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < n; i += kSmallNT) {
    float c[3];
    sphere_centre(a.sph7 + 7 * (size_t)i, c);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      a.centres[3 * (size_t)i + k] = c[k];
      lo[k] = f_min(lo[k], c[k]);
      hi[k] = f_max(hi[k], c[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    for (int o = 32; o > 0; o >>= 1) {
      lo[k] = f_min(lo[k], __shfl_xor(lo[k], o));
      hi[k] = f_max(hi[k], __shfl_xor(hi[k], o));
    }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s_red[tid >> 6][k] = lo[k];
      s_red[tid >> 6][3 + k] = hi[k];
    }
  }
  if (tid == 0) s_maxdepth = 0;
  __syncthreads();
  if (tid < 6) {
    float r = s_red[0][tid];
    for (int w = 1; w < kSmallNT / 64; ++w) r = tid < 3 ? f_min(r, s_red[w][tid]) : f_max(r, s_red[w][tid]);
    s_bounds[tid] = r;
  }
  __syncthreads();
  STAMP(1);
  // 2. Morton keys (each thread reads back the centres it wrote)
  for (int i = tid; i < n; i += kSmallNT) morton_elem(i, a.centres, s_bounds, lk, (int *)lv);
  __syncthreads();
  STAMP(2);
  // 3. stable sort by the 30-bit key
  small_sort(lk, lv, n, 30);
  STAMP(3);
  // 4. sorted spheres (canonical L, and the traversal copies of the same data)
  for (int i = tid; i < n; i += kSmallNT) {
    const float *s = a.sph7 + 7 * (size_t)lv[i];
    float *d = a.o.L7 + 7 * (size_t)i;
    float f[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) f[k] = s[k];
#pragma unroll
    for (int k = 0; k < 7; ++k) d[k] = f[k];
    a.o.sph[i] = make_float4(f[0], f[1], f[2], f[6]);
    a.o.col[i] = make_float4(f[3], f[4], f[5], 1.0f / f[6]);
  }
  __syncthreads();
  STAMP(4);
  // 5. radix tree over the sorted keys (keys in lk; parent links into lv)
  int *lpar = (int *)lv;
  for (int i = tid; i < ni; i += kSmallNT) lpar[i] = -1;
  __syncthreads();
  for (int i = tid; i < ni; i += kSmallNT) radix_tree_node(i, lk, n, a.o.left, a.o.right, lpar);
  __syncthreads();
  STAMP(5);
  // 6. depth of every inner node = number of ancestors; keys for the traversal numbering
  {
    const int E = ((ni + kSmallNT - 1) / kSmallNT) | 1;
    const int base = tid * E;
    int dep[kSmallE], md = 0;
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < ni) {
        int d = 0;
        for (int p = lpar[base + e]; p >= 0; p = lpar[p]) ++d;
        dep[e] = d;
        md = max(md, d);
        a.o.parent[base + e] = lpar[base + e];
      }
    for (int o = 32; o > 0; o >>= 1) md = max(md, __shfl_xor(md, o));
    if ((tid & 63) == 0) atomicMax(&s_maxdepth, md);
    __syncthreads();   // every walk is done: lk / lv can be reused
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < ni) {
        lk[base + e] = (unsigned)dep[e];
        lv[base + e] = (unsigned)(base + e);
      }
    __syncthreads();
    if (tid == 0) a.result[0] = s_maxdepth;
  }
  STAMP(6);
  // 7. traversal numbering: stable sort of the inner nodes by depth (< 64), then its inverse
  small_sort(lk, lv, ni, 6);
  __syncthreads();   // (the sort's last read-back is done)
  int *order = a.order, *trav_of = a.trav_of;   // global: LDS is about to hold the work lists
  for (int t = tid; t < ni; t += kSmallNT) trav_of[(int)lv[t]] = t;
  // ... then treelet by treelet (treelet.h; the multi-kernel path's treelet_size / scan / treelet_number in one block):
  // lk[t] becomes the start of the treelet rooted at the t-th node of the order by depth, the depth's residue mod D kept in bits 0..2
  {
    const int E = ((ni + kSmallNT - 1) / kSmallNT) | 1;
    const int base = tid * E;
    unsigned sz[kSmallE], sum = 0;
#pragma unroll
    for (int e = 0; e < kSmallE; ++e) {
      sz[e] = 0;
      if (e < E && base + e < ni) {
        const int c = (int)lv[base + e];
        const unsigned rel = lk[base + e] % (unsigned)kTreeletDepth;
        sz[e] = rel ? 0u : (unsigned)tl_popc(treelet_occ(c, a.o.left, a.o.right));
        lk[base + e] = rel;          // (only this thread touches its own entries until the sync below)
        sum += sz[e];
      }
    }
    unsigned long long tot;
    unsigned run = (unsigned)block_excl_scan_u64<kSmallNT>(sum, &tot);   // (syncs: the by-depth inverse above is visible after it)
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < ni) {
        lk[base + e] = (run << 3) | lk[base + e];
        run += sz[e];
      }
    __syncthreads();
    unsigned pl[kSmallE];
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < ni) {   // canonical node base + e
        const int c = base + e;
        int heap;
        const int r = treelet_root(c, (int)(lk[trav_of[c]] & 7u), a.o.parent, a.o.right, &heap);
        pl[e] = tl_pack_place((lk[trav_of[r]] >> 3) + (unsigned)tl_pos(treelet_occ(r, a.o.left, a.o.right), heap), heap);
      }
    __syncthreads();   // every by-depth index has been read
#pragma unroll
    for (int e = 0; e < kSmallE; ++e)
      if (e < E && base + e < ni) {
        trav_of[base + e] = (int)pl[e];                 // index + heap index, packed
        order[pl[e] & kTlIndexMask] = base + e;
      }
  }
  __syncthreads();
  STAMP(7);
  // 8. AABB propagation: exactly `sweeps` Jacobi sweeps from all-zero boxes (bvh.fut:47-58).
  // A node whose children are final (leaves, or inner nodes that were final one sweep earlier)
  // is final itself; once it has written that value into both ping-pong buffers, recomputing it
  // would store the same bits again, so it leaves the work list.  Lists hold node | once << 15.
  float4 *buf[2] = {a.box4, a.box4 + 2 * (size_t)ni};
  unsigned short *list[2] = {(unsigned short *)small_lds, (unsigned short *)small_lds + ni};
  __shared__ int s_cnt[3];
  for (int i = tid; i < 2 * ni; i += kSmallNT) buf[0][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < ni; i += kSmallNT) list[0][i] = (unsigned short)i;
  if (tid == 0) {
    s_cnt[0] = ni;
    s_cnt[1] = 0;
    s_cnt[2] = 0;
  }
  __syncthreads();
  int rd = 0;
  for (int s = 0; s < a.sweeps; ++s, rd ^= 1) {
    const float4 *__restrict__ prev = buf[rd];
    float4 *__restrict__ cur = buf[rd ^ 1];
    const unsigned short *lin = list[rd];
    unsigned short *lout = list[rd ^ 1];
    const int count = s_cnt[s % 3];
    int *cnt_out = &s_cnt[(s + 1) % 3];
    if (tid == 0) s_cnt[(s + 2) % 3] = 0;   // (last read one sweep ago)
    for (int i0 = 0; i0 < count; i0 += 2 * kSmallNT) {
      // two list entries per lane, so that eight box loads are in flight
      int node[2], kid[2][2];
      bool act[2], once[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = i0 + u * kSmallNT + tid;
        act[u] = idx < count;
        const unsigned ent = act[u] ? lin[idx] : 0u;
        node[u] = (int)(ent & 0x7fffu);
        once[u] = (ent >> 15) != 0;
        kid[u][0] = a.o.left[node[u]];
        kid[u][1] = a.o.right[node[u]];
      }
      float4 A[2][2], B[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool leaf = kid[u][k] <= -2;
          const float4 *pa = leaf ? a.o.sph + (-2 - kid[u][k]) : prev + 2 * (size_t)kid[u][k];
          A[u][k] = pa[0];
          B[u][k] = leaf ? A[u][k] : pa[1];
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float mn[2][3], mx[2][3];
        bool fin = true;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool leaf = kid[u][k] <= -2;
          const float4 p = A[u][k], q = B[u][k];
          mn[k][0] = leaf ? p.x - p.w : p.x;   // sphere_aabb: pos -/+ radius (ray.fut:157-160)
          mn[k][1] = leaf ? p.y - p.w : p.y;
          mn[k][2] = leaf ? p.z - p.w : p.z;
          mx[k][0] = leaf ? p.x + p.w : q.x;
          mx[k][1] = leaf ? p.y + p.w : q.y;
          mx[k][2] = leaf ? p.z + p.w : q.z;
          fin = fin && (leaf || p.w != 0.0f);
        }
        if (act[u]) {
          cur[2 * (size_t)node[u]] = make_float4(f_min(mn[0][0], mn[1][0]), f_min(mn[0][1], mn[1][1]),
                                                 f_min(mn[0][2], mn[1][2]), fin ? 1.0f : 0.0f);
          cur[2 * (size_t)node[u] + 1] = make_float4(f_max(mx[0][0], mx[1][0]), f_max(mx[0][1], mx[1][1]),
                                                     f_max(mx[0][2], mx[1][2]), 0.0f);
        }
        // survivors go to the next list (one LDS atomic per wave)
        const bool keep = act[u] && !(fin && once[u]);
        const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
        if (km) {
          int basep = 0;
          if ((tid & 63) == 0) basep = atomicAdd(cnt_out, __popcll(km));
          basep = __builtin_amdgcn_readfirstlane(basep);
          const int r = __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
          if (keep) lout[basep + r] = (unsigned short)(node[u] | (fin ? 0x8000 : 0));
        }
      }
    }
    __syncthreads();
  }
  STAMP(8);
  // 9. canonical boxes and the traversal records, from the newest buffer
  const float4 *fb = buf[rd];
  for (int i = tid; i < ni; i += kSmallNT) {
    const float4 l = fb[2 * (size_t)i], h = fb[2 * (size_t)i + 1];
    a.o.bmin[3 * (size_t)i + 0] = l.x; a.o.bmin[3 * (size_t)i + 1] = l.y; a.o.bmin[3 * (size_t)i + 2] = l.z;
    a.o.bmax[3 * (size_t)i + 0] = h.x; a.o.bmax[3 * (size_t)i + 1] = h.y; a.o.bmax[3 * (size_t)i + 2] = h.z;
  }
  for (int t = tid; t < ni; t += kSmallNT) {
    const int c = order[t];
    const int kid[2] = {a.o.left[c], a.o.right[c]};
    float4 q[4];
    int ref[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool leaf = kid[k] <= -2;
      const size_t at = leaf ? 0 : (size_t)kid[k];
      const float4 l = fb[2 * at], h = fb[2 * at + 1];
      ref[k] = leaf ? ~(-2 - kid[k]) : (int)((unsigned)trav_of[at] & kTlIndexMask);
      q[2 * k] = leaf ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(l.x, l.y, l.z, 0.f);
      q[2 * k + 1] = leaf ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(h.x, h.y, h.z, 0.f);
    }
    q[0].w = __int_as_float((int)((unsigned)ref[0] << 8));   // pre-shifted (see trav_node)
    q[1].w = __int_as_float((int)((unsigned)ref[1] << 8));
    const TlMasks tm = treelet_masks(c, (unsigned)trav_of[c], a.o.parent, a.o.left, a.o.right);
    q[2].w = __int_as_float((int)tm.l);
    q[3].w = __int_as_float((int)tm.r);
    a.o.nodes64[4 * (size_t)t + 0] = q[0];
    a.o.nodes64[4 * (size_t)t + 1] = q[1];
    a.o.nodes64[4 * (size_t)t + 2] = q[2];
    a.o.nodes64[4 * (size_t)t + 3] = q[3];
    const float4 l = fb[2 * (size_t)c], h = fb[2 * (size_t)c + 1];
    const float4 r0 = make_float4(l.x, l.y, l.z, __int_as_float(ref[0])), r1 = make_float4(h.x, h.y, h.z, __int_as_float(ref[1]));
    a.o.nodes32[2 * (size_t)t + 0] = r0;
    a.o.nodes32[2 * (size_t)t + 1] = r1;
    if (t == 0) {   // the root's box goes straight to the host (tested when a ray starts)
      float *rr = (float *)(a.result + 4);
      rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w;
      rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
    }
  }
  __syncthreads();
  STAMP(9);
#undef STAMP
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// one stable LSD pass of kSortBits bits over (keys, vals): in -> out
hipError_t sort_pass(const unsigned *kin, const int *vin, unsigned *kout, int *vout, int n, int shift, unsigned *counts,
                     hipStream_t st) {
  const int nblocks = cdiv(n, kBT * kSortE);
  hipLaunchKernelGGL(sort_count_kernel, dim3(nblocks), dim3(kBT), 0, st, kin, n, shift, counts, nblocks);
  if (nblocks <= kSortRawBlocks) {
    hipLaunchKernelGGL(sort_scatter_kernel<true>, dim3(nblocks), dim3(kBT), 0, st, kin, vin, n, shift, counts, nblocks, kout, vout);
  } else {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kScanNT), 0, st, counts, kSortDigits * nblocks);
    hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3(nblocks), dim3(kBT), 0, st, kin, vin, n, shift, counts, nblocks, kout, vout);
  }
  return hipGetLastError();
}
// ... and a chained one: the counts are there already, the pass leaves the next pass's (next == nullptr: the last pass)
hipError_t sort_pass_chained(const unsigned *kin, const int *vin, unsigned *kout, int *vout, int n, int shift, const unsigned *counts,
                             unsigned *next, hipStream_t st) {
  const int nblocks = cdiv(n, kSortTile);
  hipLaunchKernelGGL((sort_scatter_kernel<true, true>), dim3(nblocks), dim3(kBT), 0, st, kin, vin, n, shift, counts, nblocks, kout, vout,
                     next, shift + kSortBits);
  return hipGetLastError();
}

}  // namespace

#define BVH_HIP(call)                      \
  do {                                     \
    hipError_t e_ = (call);                \
    if (e_ != hipSuccess) return e_;       \
  } while (0)

namespace {
constexpr int kKeyPasses = (30 + kSortBits - 1) / kSortBits, kDepthPasses = (6 + kSortBits - 1) / kSortBits, kChainPasses = kKeyPasses + kDepthPasses;
// device scratch of one build, carved from one caller-provided block (256-byte aligned pieces)
struct ScratchLayout {
  size_t centres, partial, bounds, k0, k1, v0, v1, counts, chain, bufmin, bufmax, depth, trav, flags, box4, total;
};
ScratchLayout scratch_layout(int n) {
  const size_t ni = (size_t)n - 1;
  const int nb_n = cdiv(n, kBT);
  const size_t sort_blocks = (size_t)cdiv(n, kBT * kSortE), red_blocks = nb_n < 1024 ? nb_n : 1024;
  ScratchLayout l{};
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~size_t(255); return at; };
  l.centres = carve(sizeof(float) * 3 * (size_t)n);
  if (n <= small_use()) {
    l.box4 = carve(sizeof(float4) * 4 * ni);
    l.v0 = carve(sizeof(int) * ni);
    l.trav = carve(sizeof(int) * ni);
  } else {
    l.partial = carve(sizeof(float) * 6 * red_blocks);
    l.bounds = carve(sizeof(float) * 8);
    l.k0 = carve(sizeof(unsigned) * (size_t)n);
    l.k1 = carve(sizeof(unsigned) * (size_t)n);
    l.v0 = carve(sizeof(int) * (size_t)n);
    l.v1 = carve(sizeof(int) * (size_t)n);
    l.counts = carve(sizeof(unsigned) * kSortDigits * sort_blocks + 16);
    l.chain = carve(sizeof(unsigned) * kSortDigits * (sort_blocks <= kSortRawBlocks ? sort_blocks : 0) * kChainPasses);
    l.bufmin = carve(sizeof(float) * 3 * ni);
    l.bufmax = carve(sizeof(float) * 3 * ni);
    l.depth = carve(sizeof(int) * ni);
    l.trav = carve(sizeof(int) * ni);
    l.flags = carve(sizeof(int) * 64);
  }
  l.total = off;
  return l;
}
}  // namespace

size_t gpu_build_scratch_bytes(int n) { return scratch_layout(n).total; }
void warm_build_kernels() {   // see warm_render_kernels; with the context's device current
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void *)bvh_small_kernel);
  // (above 64 KB of LDS a kernel has to say so, per device; here and not between two launches of a build: the host is barely ahead of the device there)
  (void)hipFuncSetAttribute((const void *)rank_sort_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_sort_lds_bytes(kRankMaxN));
  (void)hipFuncSetAttribute((const void *)rank_sort_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_sort_lds_bytes(kRankMaxN));
  (void)hipFuncSetAttribute((const void *)tree_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(unsigned) * kRankMaxN);
  (void)hipFuncSetAttribute((const void *)treelet_number_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(unsigned) * 4 * kNumNT * ((kRankMaxN + 4 * kNumNT - 1) / (4 * kNumNT)));
}
// host-pinned block the kernels report through: [0] max depth, [1] "depth changed" flag,
// [4..11] the root's traversal record, [16..] phase timestamps of the small-scene kernel
size_t gpu_build_pinned_bytes() { return 64 * sizeof(int); }

namespace {
__global__ __launch_bounds__(kBT) void copy_words_kernel(const uint4 *src, uint4 *dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)kBT + threadIdx.x; i < n16; i += (size_t)gridDim.x * kBT) dst[i] = src[i];
}
}  // namespace
// Device copy of a host-PINNED buffer by a kernel that reads it over PCIe (bytes rounded up to 16;
// both buffers padded accordingly).  hipMemcpy of a few hundred KB of pageable memory takes 7 ms
// on this stack, its pinned/DMA variant 0.3 ms, this ~20 us.
hipError_t gpu_copy_from_pinned(void *dst_dev, const void *src_pinned, size_t bytes, hipStream_t st) {
  const size_t n16 = (bytes + 15) / 16;
  if (n16 == 0) return hipSuccess;
  const size_t blocks = (n16 + kBT - 1) / kBT;
  hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(kBT), 0, st,
                     (const uint4 *)src_pinned, (uint4 *)dst_dev, n16);
  return hipGetLastError();
}

// Builds everything from n spheres in device memory.  All output arrays, the scratch block
// (gpu_build_scratch_bytes(n)) and the host-pinned block (gpu_build_pinned_bytes()) are allocated by
// the caller (sizes in rt_device.hpp: GpuBvhOut).  Returns after the stream has drained, with the
// tree height and the root's box.
hipError_t gpu_build_bvh(const float *sph7_dev, int n, const GpuBvhOut &o, char *scratch, char *pinned, hipStream_t st,
                         int *height_out, float root_lo[3], float root_hi[3]) {
  const int ni = n - 1;
  const int nb_n = cdiv(n, kBT), nb_ni = cdiv(ni, kBT);
  const int red_blocks = nb_n < 1024 ? nb_n : 1024;
  const ScratchLayout l = scratch_layout(n);
  float *centres = (float *)(scratch + l.centres);
  int *result = (int *)pinned;
  const float *root = (const float *)(result + 4);

  if (n <= small_use()) {
    // the whole build in one workgroup / one launch
    SmallArgs a{sph7_dev, n, (int)log2f((float)n) + 2, o, centres, (float4 *)(scratch + l.box4),
                (int *)(scratch + l.v0), (int *)(scratch + l.trav), result};
    BVH_HIP(hipFuncSetAttribute((const void *)bvh_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * (int)sizeof(unsigned) * kSmallMax));
    hipLaunchKernelGGL(bvh_small_kernel, dim3(1), dim3(kSmallNT), 2 * sizeof(unsigned) * (size_t)n, st, a);
    BVH_HIP(hipGetLastError());
    BVH_HIP(hipStreamSynchronize(st));
    *height_out = result[0] + 1;
    for (int k = 0; k < 3; ++k) {
      root_lo[k] = root[k];
      root_hi[k] = root[4 + k];
    }
    if (getenv("RT_BVH_STAMPS")) {
      const unsigned long long *t = (const unsigned long long *)(result + 16);
      fprintf(stderr, "bvh_small n=%d height=%d: centres %.1f morton %.1f sort %.1f gather %.1f tree %.1f depth %.1f numbering %.1f boxes %.1f records %.1f us\n",
              n, *height_out, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
              (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01, (t[7] - t[6]) * 0.01, (t[8] - t[7]) * 0.01, (t[9] - t[8]) * 0.01);
    }
    return hipSuccess;
  }
  float *partial = (float *)(scratch + l.partial), *bounds = (float *)(scratch + l.bounds);
  unsigned *keys[2] = {(unsigned *)(scratch + l.k0), (unsigned *)(scratch + l.k1)}, *counts = (unsigned *)(scratch + l.counts);
  int *vals[2] = {(int *)(scratch + l.v0), (int *)(scratch + l.v1)};
  float *bufmin = (float *)(scratch + l.bufmin), *bufmax = (float *)(scratch + l.bufmax);
  int *depth = (int *)(scratch + l.depth), *trav_of = (int *)(scratch + l.trav), *flags = (int *)(scratch + l.flags);
  // 1. centres, bounds, Morton keys
  // (a CHAINED sort, for scenes of up to kSortRawBlocks sort tiles: no count launches, no launch for the bounds either;
  // RANKED, up to kRankMaxN spheres: one launch per sort)
  const int sort_blocks = cdiv(n, kSortTile), sort_blocks_i = cdiv(ni, kSortTile);
  const bool ranked = n <= kRankMaxN, chained = !ranked && sort_blocks <= kSortRawBlocks;
  unsigned *chain = (unsigned *)(scratch + l.chain);
  auto chain_counts = [&](int pass) { return chain + (size_t)pass * kSortDigits * sort_blocks; };
  if (ranked) {
    hipLaunchKernelGGL(morton_all_kernel, dim3(nb_n), dim3(kBT), 0, st, sph7_dev, n, keys[0], flags);
  } else {
    hipLaunchKernelGGL(centres_minmax_kernel, dim3(red_blocks), dim3(kBT), 0, st, sph7_dev, n, centres, partial, chain,
                       chained ? kChainPasses * kSortDigits * sort_blocks : 0, flags);
    if (chained) {
      hipLaunchKernelGGL(morton_chain_kernel, dim3(nb_n), dim3(kBT), 0, st, centres, partial, red_blocks, n, keys[0], vals[0], chain_counts(0),
                         sort_blocks);
    } else {
      hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(kBT), 0, st, partial, red_blocks, bounds);
      hipLaunchKernelGGL(morton_kernel, dim3(nb_n), dim3(kBT), 0, st, centres, bounds, n, keys[0], vals[0]);
    }
  }
  // 2. stable sort by the 30-bit key: 8 passes of 4 bits
  int cur = 0;
  if (ranked) {
    hipLaunchKernelGGL(rank_sort_kernel<false>, dim3(cdiv(n, kRankPerBlock)), dim3(kRankNT), rank_sort_lds_bytes(n), st, keys[0], n,
                       keys[1], vals[1], (const int *)nullptr, (const int *)nullptr, (unsigned *)nullptr, (int *)nullptr);
    cur = 1;
  } else {
    for (int shift = 0, pass = 0; shift < 30; shift += kSortBits, ++pass) {
      if (chained)
        BVH_HIP(sort_pass_chained(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, chain_counts(pass),
                                  pass + 1 < kKeyPasses ? chain_counts(pass + 1) : nullptr, st));
      else
        BVH_HIP(sort_pass(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, counts, st));
      cur ^= 1;
    }
  }
  // 3. sorted spheres, radix tree over the sorted keys; 4. AABB propagation: exactly floor(log2 n) + 2 sweeps from all-zero boxes
  const int sweeps = (int)log2f((float)n) + 2;
  float *pmin = bufmin, *pmax = bufmax, *cmin = o.bmin, *cmax = o.bmax;
  const bool fused = n <= kSweepFusedMaxN;
  const int nlaunch = fused ? (sweeps + kSweepLevels - 1) / kSweepLevels : sweeps;
  if (nlaunch % 2 == 0) {   // the last launch must land in o.bmin / o.bmax
    pmin = o.bmin; pmax = o.bmax; cmin = bufmin; cmax = bufmax;
  }
  if (fused) {
    if (ranked)
      hipLaunchKernelGGL(tree_kernel<true>, dim3(nb_n), dim3(kBT), sizeof(unsigned) * (size_t)n, st, keys[cur], vals[cur], sph7_dev, n, o.L7, o.sph,
                         o.col, o.left, o.right, o.parent);
    else
      hipLaunchKernelGGL(tree_kernel<false>, dim3(nb_n), dim3(kBT), 0, st, keys[cur], vals[cur], sph7_dev, n, o.L7, o.sph, o.col, o.left, o.right,
                         o.parent);
  } else {
    hipLaunchKernelGGL(gather_spheres_kernel, dim3(nb_n), dim3(kBT), 0, st, sph7_dev, vals[cur], n, o.L7, o.sph, o.col);
    hipLaunchKernelGGL(build_fills_kernel, dim3(nb_ni), dim3(kBT), 0, st, o.parent, pmin, pmax, depth, ni);   // (fin[] borrows depth[], which is set later)
    hipLaunchKernelGGL(radix_tree_kernel, dim3(nb_ni), dim3(kBT), 0, st, keys[cur], n, o.left, o.right, o.parent);
  }
  // (ranked: the first sweep launch also writes the depths, the key of the numbering below; the sorted keys are in keys[1] there)
  const int *walk = ranked ? o.parent : nullptr;
  for (int s = 0; s < sweeps;) {
    const int k = fused ? (sweeps - s < kSweepLevels ? sweeps - s : kSweepLevels) : 1;
#define RT_SWEEPK(K)                                                                                                                         \
  case K:                                                                                                                                    \
    if (s == 0)                                                                                                                              \
      hipLaunchKernelGGL((aabb_sweepk_kernel<K, true>), dim3(nb_ni), dim3(kBT), 0, st, o.L7, o.left, o.right, ni, pmin, pmax, cmin, cmax,    \
                         walk, keys[0], flags + 1);                                                                                          \
    else                                                                                                                                     \
      hipLaunchKernelGGL((aabb_sweepk_kernel<K, false>), dim3(nb_ni), dim3(kBT), 0, st, o.L7, o.left, o.right, ni, pmin, pmax, cmin, cmax,   \
                         (const int *)nullptr, (unsigned *)nullptr, (int *)nullptr);                                                         \
    break;
    if (!fused) {
      hipLaunchKernelGGL(aabb_sweep_kernel, dim3(nb_ni), dim3(kBT), 0, st, o.L7, o.left, o.right, ni, pmin, pmax, cmin, cmax, depth, s);
    } else {
      switch (k) {
        RT_SWEEPK(1) RT_SWEEPK(2) RT_SWEEPK(3) RT_SWEEPK(4) RT_SWEEPK(5)
      }
    }
#undef RT_SWEEPK
    s += k;
    float *t0 = pmin, *t1 = pmax;
    pmin = cmin; pmax = cmax; cmin = t0; cmax = t1;
  }
  // (after the loop pmin/pmax point at the newest boxes == o.bmin/o.bmax by the parity choice above)
  // 5./6. depths by walking up the parent links; traversal numbering: stable sort of the inner
  // nodes by depth
  if (!ranked)
    hipLaunchKernelGGL(depth_walk_kernel, dim3(nb_ni), dim3(kBT), 0, st, o.parent, ni, keys[0], vals[0], flags + 1,
                       chained ? chain_counts(kKeyPasses) : nullptr, sort_blocks_i);
  if (ranked) {
    // ... one launch: the numbering by depth, its inverse, the treelet sizes; then the places (every block scans the sizes itself)
    unsigned *size = (unsigned *)bufmin;      // (free since the sweeps: the newest boxes are in o.bmin / o.bmax)
    int *place = depth, *order2 = vals[0];
    hipLaunchKernelGGL(rank_sort_kernel<true>, dim3(cdiv(ni, kRankPerBlock)), dim3(kRankNT), rank_sort_lds_bytes(ni), st, keys[0], ni,
                       keys[1], vals[1], o.left, o.right, size, trav_of);
    hipLaunchKernelGGL(treelet_number_scan_kernel, dim3(cdiv(ni, kNumNT)), dim3(kNumNT), sizeof(unsigned) * 4 * kNumNT * (size_t)cdiv(ni, 4 * kNumNT), st,
                       keys[0], size, trav_of, o.parent, o.left, o.right, ni, place, order2);
    hipLaunchKernelGGL(trav_nodes_kernel, dim3(nb_ni), dim3(kBT), 0, st, order2, place, o.parent, o.left, o.right, o.bmin, o.bmax, ni, o.nodes32,
                       o.nodes64, flags + 1, result);
  } else {
    cur = 0;
    for (int shift = 0, pass = 0; shift < 6; shift += kSortBits, ++pass) {   // depth <= 30 key bits + 26 index bits < 64: 2 passes
      if (chained)
        BVH_HIP(sort_pass_chained(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], ni, shift, chain_counts(kKeyPasses + pass),
                                  pass + 1 < kDepthPasses ? chain_counts(kKeyPasses + pass + 1) : nullptr, st));
      else
        BVH_HIP(sort_pass(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], ni, shift, counts, st));
      cur ^= 1;
    }
    // ... then treelet by treelet (treelet.h): sizes of the treelets in the order of their roots, their starts, the nodes' places
    unsigned *start = keys[cur ^ 1];      // (free since the sort)
    int *place = depth, *order2 = vals[cur ^ 1];
    const int sb = cdiv(ni, kBT * kScanE);
    hipLaunchKernelGGL(treelet_size_kernel, dim3(nb_ni), dim3(kBT), 0, st, keys[cur], vals[cur], o.left, o.right, ni, start, trav_of);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(sb), dim3(kBT), 0, st, start, ni, counts);
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kScanNT), 0, st, counts, sb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(sb), dim3(kBT), 0, st, start, ni, counts);
    hipLaunchKernelGGL(treelet_number_kernel, dim3(nb_ni), dim3(kBT), 0, st, keys[cur], start, trav_of, o.parent, o.left, o.right, ni,
                       place, order2);
    hipLaunchKernelGGL(trav_nodes_kernel, dim3(nb_ni), dim3(kBT), 0, st, order2, place, o.parent, o.left, o.right, o.bmin, o.bmax, ni, o.nodes32,
                       o.nodes64, flags + 1, result);
  }
  BVH_HIP(hipGetLastError());
  BVH_HIP(hipStreamSynchronize(st));
  *height_out = result[0] + 1;   // levels of inner nodes == edges on the longest root -> leaf path
  for (int k = 0; k < 3; ++k) {
    root_lo[k] = root[k];
    root_hi[k] = root[4 + k];
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------
// The visiting order of a frame nothing is known about (api.cpp: get_first_order): tile rows in bit-reversed order and, inside a
// row, blocks of 8 tiles in bit-reversed order of the blocks.  Built on the device (a pageable hipMemcpy of the 60 KB table took
// milliseconds of the view's first frame on this stack).  rank[0 .. tiles_y): a row's place among the rows; rank[tiles_y ..): a
// block's first column in the new order -- brute force, a thread per row / block (api.cpp caps the sizes).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int fo_bitrev(int i, int bits) {
  int r = 0;
  for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
  return r;
}
__global__ __launch_bounds__(256) void first_order_rank_kernel(int tiles_x, int tiles_y, int *rank) {
  // (the reversed indices once, in LDS: the counting loops below read them instead of reversing r again for every pair -- 31 us -> 6 us at 125 rows)
  __shared__ unsigned short s_rev[1024];
  const int nb = (tiles_x + 7) / 8;
  int by = 0, bx = 0;
  while ((1 << by) < tiles_y) ++by;
  while ((1 << bx) < nb) ++bx;
  const bool tabled = tiles_y + nb <= 1024;
  if (tabled) {
    for (int i = threadIdx.x; i < tiles_y + nb; i += blockDim.x) s_rev[i] = (unsigned short)(i < tiles_y ? fo_bitrev(i, by) : fo_bitrev(i - tiles_y, bx));
    __syncthreads();
  }
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < tiles_y) {
    const int mine = fo_bitrev(t, by);
    int n = 0;
    for (int r = 0; r < tiles_y; ++r) n += (tabled ? (int)s_rev[r] : fo_bitrev(r, by)) < mine ? 1 : 0;
    rank[t] = n;
  } else if (t < tiles_y + nb) {
    const int q = t - tiles_y, mine = fo_bitrev(q, bx);
    int off = 0;
    for (int o = 0; o < nb; ++o)
      if ((tabled ? (int)s_rev[tiles_y + o] : fo_bitrev(o, bx)) < mine) off += min(8, tiles_x - 8 * o);
    rank[tiles_y + q] = off;
  }
}
// (... and zeroes the class tables behind the permutation: order[ntiles .. total) -- a memset's two or three fill dispatches ahead of a context's first frame)
__global__ void first_order_fill_kernel(int tiles_x, int tiles_y, const int *rank, int *order, int total) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int ntiles = tiles_x * tiles_y;
  for (int z = ntiles + t; z < total; z += gridDim.x * blockDim.x) order[z] = 0;
  if (t >= ntiles) return;
  const int r = t / tiles_x, x = t - r * tiles_x;
  order[rank[r] * tiles_x + rank[tiles_y + (x >> 3)] + (x & 7)] = t;
}
// `order`: order_table_ints(tiles) ints (the permutation, then zeroed class tables); `rank`: tiles_y + ceil(tiles_x / 8) ints of scratch
hipError_t launch_first_order(int *order, int *rank, int tiles_x, int tiles_y, hipStream_t stream) {
  const int ntiles = tiles_x * tiles_y, nb = (tiles_x + 7) / 8;
  hipLaunchKernelGGL(first_order_rank_kernel, dim3((tiles_y + nb + 255) / 256), dim3(256), 0, stream, tiles_x, tiles_y, rank);
  hipLaunchKernelGGL(first_order_fill_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, stream, tiles_x, tiles_y, rank, order, order_table_ints(ntiles));
  return hipGetLastError();
}

// The primary rays' u = i / w and v = (h - row) / h (trace_ray, ray.fut:150-154; lane_core.h: pixel_u / pixel_v) per column / row, computed
// on the device: the same correctly rounded binary32 divisions the host would make -- no host table, no blocking copy inside a view's
// first render call.
__global__ void uv_tables_kernel(float *u, float *v, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < w) u[i] = pixel_u(i, w);
  if (i < h) v[i] = pixel_v(i, h);
}
hipError_t launch_uv_tables(float *u, float *v, int w, int h, hipStream_t stream) {
  const int n = w > h ? w : h;
  hipLaunchKernelGGL(uv_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, u, v, w, h);
  return hipGetLastError();
}

}  // namespace rtk
