// bvh_build.hip -- prepare_scene's BVH construction on the GPU: the LBVH of
// futhark/bvh.fut:30-59 + futhark/radixtree.fut:11-72, bit-identical to the reference's
// {L, I} arrays, plus the derived traversal copy the render kernels read.
//
//   centres + min/max   bvh.fut:31-37     (fmin/fmax reductions: order-free for non-NaN input)
//   Morton keys         bvh.fut:38-41, :8-22   (IEEE division, fmax(NaN, 0) = 0 on a flat axis)
//   stable sort by key  bvh.fut:43        LSD radix, 2 bits per pass like radix_sort.fut:14-32
//   radix tree          radixtree.fut:23-72    one thread per inner node, pure integer
//   AABB propagation    bvh.fut:44-58     EXACTLY floor(log2 n)+2 double-buffered Jacobi sweeps
//   traversal copy      nodes renumbered by depth (root levels first = the LDS-staged prefix)
//
// Everything is enqueued on the caller's stream; the only host round trips are the
// "did any depth change" flag (once per 8 sweeps) and the final tree height.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "rt_device.hpp"

namespace rtk {
namespace {

constexpr int kBT = 256;          // threads per block everywhere in this file
constexpr int kSortE = 4;         // sort: elements per thread (tile = 1024 per block)

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

__global__ __launch_bounds__(kBT) void centres_minmax_kernel(const float *sph7, int n, float *centres, float *partial) {
  __shared__ float red[6][kBT];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
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

__global__ __launch_bounds__(kBT) void minmax_final_kernel(const float *partial, int nblocks, float *bounds) {
  __shared__ float red[6][kBT];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int b = threadIdx.x; b < nblocks; b += kBT)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = f_min(lo[a], partial[b * 6 + a]);
      hi[a] = f_max(hi[a], partial[b * 6 + 3 + a]);
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

__global__ __launch_bounds__(kBT) void morton_kernel(const float *centres, const float *bounds, int n, unsigned *keys,
                                                     int *vals) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= n) return;
  unsigned code[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mn = bounds[a], mx = bounds[3 + a];
    const float q = (centres[3 * (size_t)i + a] - mn) / (mx - mn);   // 0/0 = NaN on a flat axis -> 0 below
    code[a] = spread10(quantise10(q));
  }
  keys[i] = code[0] * 4u + code[1] * 2u + code[2];
  vals[i] = i;
}

// ---- stable LSD radix sort of (key, val), 2 bits per pass ---------------------------------
// Thread t of a block owns kSortE CONSECUTIVE elements, so thread order == element order and a
// block-wide exclusive scan of per-thread digit counts gives stable ranks.  Counts of the four
// digit values travel packed in one u64 (16 bits each).
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long v, unsigned long long *total) {
  __shared__ unsigned long long wave_sum[kBT / 64];
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
  for (int w = 0; w < kBT / 64; ++w) {
    if (w < wave) base += wave_sum[w];
    tot += wave_sum[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kBT) void sort_count_kernel(const unsigned *keys, int n, int shift, unsigned *block_counts,
                                                         int nblocks) {
  const int base = (blockIdx.x * kBT + threadIdx.x) * kSortE;
  unsigned long long cnt = 0;
#pragma unroll
  for (int e = 0; e < kSortE; ++e)
    if (base + e < n) cnt += 1ull << (16 * ((keys[base + e] >> shift) & 3u));
  unsigned long long tot;
  (void)block_excl_scan_u64(cnt, &tot);
  if (threadIdx.x < 4) block_counts[threadIdx.x * nblocks + blockIdx.x] = (unsigned)((tot >> (16 * threadIdx.x)) & 0xffffull);
}

// exclusive scan of m counters by one block (m = 4 * nblocks, digit-major = the order the
// sorted array is laid out in)
__global__ __launch_bounds__(kBT) void scan_small_kernel(unsigned *data, int m) {
  __shared__ unsigned carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int start = 0; start < m; start += kBT * 4) {
    const int i0 = start + threadIdx.x * 4;
    unsigned v[4], sum = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = i0 + e < m ? data[i0 + e] : 0u;
      sum += v[e];
    }
    unsigned long long tot;
    const unsigned excl = (unsigned)block_excl_scan_u64(sum, &tot);
    unsigned run = carry_s + excl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i0 + e < m) data[i0 + e] = run;
      run += v[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += (unsigned)tot;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBT) void sort_scatter_kernel(const unsigned *keys_in, const int *vals_in, int n, int shift,
                                                           const unsigned *block_offsets, int nblocks, unsigned *keys_out,
                                                           int *vals_out) {
  const int base = (blockIdx.x * kBT + threadIdx.x) * kSortE;
  unsigned k[kSortE];
  int v[kSortE];
  unsigned long long cnt = 0;
#pragma unroll
  for (int e = 0; e < kSortE; ++e) {
    if (base + e < n) {
      k[e] = keys_in[base + e];
      v[e] = vals_in[base + e];
      cnt += 1ull << (16 * ((k[e] >> shift) & 3u));
    }
  }
  unsigned long long tot;
  unsigned long long rank = block_excl_scan_u64(cnt, &tot);   // per digit: elements of this block before this thread
#pragma unroll
  for (int e = 0; e < kSortE; ++e) {
    if (base + e < n) {
      const unsigned d = (k[e] >> shift) & 3u;
      const unsigned pos = block_offsets[d * nblocks + blockIdx.x] + (unsigned)((rank >> (16 * d)) & 0xffffull);
      keys_out[pos] = k[e];
      vals_out[pos] = v[e];
      rank += 1ull << (16 * d);
    }
  }
}

// ---- gather sorted spheres ----------------------------------------------------------------
__global__ __launch_bounds__(kBT) void gather_spheres_kernel(const float *sph7, const int *order, int n, float *L7) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= n) return;
  const float *s = sph7 + 7 * (size_t)order[i];
  float *d = L7 + 7 * (size_t)i;
#pragma unroll
  for (int k = 0; k < 7; ++k) d[k] = s[k];
}

// ---- radix tree (radixtree.fut:13-72) -----------------------------------------------------
__device__ __forceinline__ int delta(const unsigned *L, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned a = L[i], b = L[j];
  if (a == b) return 32 + __clz((unsigned)i ^ (unsigned)j);   // __clz(0) == 32, as u32.clz
  return __clz(a ^ b);
}

// ptr encoding of the canonical arrays: inner i -> i, leaf i -> -2 - i
__global__ __launch_bounds__(kBT) void radix_tree_kernel(const unsigned *L, int n, int *left, int *right, int *parent) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= n - 1) return;
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

// ---- AABB propagation: one Jacobi sweep (bvh.fut:48-58) -----------------------------------
__global__ __launch_bounds__(kBT) void aabb_sweep_kernel(const float *L7, const int *left, const int *right, int ni,
                                                         const float *pmin, const float *pmax, float *cmin, float *cmax) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= ni) return;
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

// ---- node depths (for the traversal numbering) ----------------------------------------------
__global__ __launch_bounds__(kBT) void depth_sweep_kernel(const int *parent, int ni, int *depth, int *changed) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= ni || depth[i] >= 0) return;
  const int pd = depth[parent[i]];
  if (pd >= 0) {
    depth[i] = pd + 1;
    *changed = 1;
  }
}

__global__ __launch_bounds__(kBT) void depth_keys_kernel(const int *depth, int ni, unsigned *keys, int *vals, int *maxdepth) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= ni) return;
  keys[i] = (unsigned)depth[i];
  vals[i] = i;
  atomicMax(maxdepth, depth[i]);
}

__global__ __launch_bounds__(kBT) void invert_kernel(const int *order, int ni, int *trav_of) {
  const int t = blockIdx.x * kBT + threadIdx.x;
  if (t < ni) trav_of[order[t]] = t;
}

// ---- traversal copy ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBT) void trav_nodes_kernel(const int *order, const int *trav_of, const int *left,
                                                         const int *right, const float *bmin, const float *bmax, int ni,
                                                         float4 *nodes32, float4 *nodes64) {
  const int t = blockIdx.x * kBT + threadIdx.x;
  if (t >= ni) return;
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
      ref[k] = trav_of[kid[k]];
      const float *mn = bmin + 3 * (size_t)kid[k], *mx = bmax + 3 * (size_t)kid[k];
      q[2 * k] = make_float4(mn[0], mn[1], mn[2], 0.f);
      q[2 * k + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
  }
  q[0].w = __int_as_float(ref[0]);
  q[1].w = __int_as_float(ref[1]);
  nodes64[4 * (size_t)t + 0] = q[0];
  nodes64[4 * (size_t)t + 1] = q[1];
  nodes64[4 * (size_t)t + 2] = q[2];
  nodes64[4 * (size_t)t + 3] = q[3];
  const float *mn = bmin + 3 * (size_t)c, *mx = bmax + 3 * (size_t)c;
  nodes32[2 * (size_t)t + 0] = make_float4(mn[0], mn[1], mn[2], __int_as_float(ref[0]));
  nodes32[2 * (size_t)t + 1] = make_float4(mx[0], mx[1], mx[2], __int_as_float(ref[1]));
}

__global__ __launch_bounds__(kBT) void trav_spheres_kernel(const float *L7, int n, float4 *sph, float4 *col) {
  const int i = blockIdx.x * kBT + threadIdx.x;
  if (i >= n) return;
  const float *s = L7 + 7 * (size_t)i;
  sph[i] = make_float4(s[0], s[1], s[2], s[6]);
  col[i] = make_float4(s[3], s[4], s[5], 1.0f / s[6]);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// one stable 2-bit LSD pass over (keys, vals): in -> out
hipError_t sort_pass(const unsigned *kin, const int *vin, unsigned *kout, int *vout, int n, int shift, unsigned *counts,
                     hipStream_t st) {
  const int nblocks = cdiv(n, kBT * kSortE);
  hipLaunchKernelGGL(sort_count_kernel, dim3(nblocks), dim3(kBT), 0, st, kin, n, shift, counts, nblocks);
  hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kBT), 0, st, counts, 4 * nblocks);
  hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(kBT), 0, st, kin, vin, n, shift, counts, nblocks, kout, vout);
  return hipGetLastError();
}

}  // namespace

#define BVH_HIP(call)                      \
  do {                                     \
    hipError_t e_ = (call);                \
    if (e_ != hipSuccess) return e_;       \
  } while (0)

// Builds everything from n spheres already on the device.  All output arrays are allocated by
// the caller (sizes in rt_device.hpp: GpuBvhOut).  Scratch is allocated and freed here.
hipError_t gpu_build_bvh(const float *sph7_dev, int n, const GpuBvhOut &o, hipStream_t st, int *height_out) {
  const int ni = n - 1;
  const int nb_n = cdiv(n, kBT), nb_ni = cdiv(ni, kBT);
  const int sort_blocks = cdiv(n, kBT * kSortE);
  const int red_blocks = nb_n < 1024 ? nb_n : 1024;
  // one scratch allocation, carved up (256-byte aligned pieces)
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~size_t(255); return at; };
  const size_t o_centres = carve(sizeof(float) * 3 * (size_t)n), o_partial = carve(sizeof(float) * 6 * (size_t)red_blocks),
               o_bounds = carve(sizeof(float) * 8), o_k0 = carve(sizeof(unsigned) * (size_t)n),
               o_k1 = carve(sizeof(unsigned) * (size_t)n), o_v0 = carve(sizeof(int) * (size_t)n),
               o_v1 = carve(sizeof(int) * (size_t)n), o_counts = carve(sizeof(unsigned) * 4 * (size_t)sort_blocks + 16),
               o_bufmin = carve(sizeof(float) * 3 * (size_t)ni), o_bufmax = carve(sizeof(float) * 3 * (size_t)ni),
               o_depth = carve(sizeof(int) * (size_t)ni), o_trav = carve(sizeof(int) * (size_t)ni), o_flags = carve(sizeof(int) * 4);
  char *scratch = nullptr;
  BVH_HIP(hipMalloc((void **)&scratch, off));
  float *centres = (float *)(scratch + o_centres), *partial = (float *)(scratch + o_partial), *bounds = (float *)(scratch + o_bounds);
  unsigned *keys[2] = {(unsigned *)(scratch + o_k0), (unsigned *)(scratch + o_k1)}, *counts = (unsigned *)(scratch + o_counts);
  int *vals[2] = {(int *)(scratch + o_v0), (int *)(scratch + o_v1)};
  float *bufmin = (float *)(scratch + o_bufmin), *bufmax = (float *)(scratch + o_bufmax);
  int *depth = (int *)(scratch + o_depth), *trav_of = (int *)(scratch + o_trav), *flags = (int *)(scratch + o_flags);

  // 1. centres, bounds, Morton keys
  hipLaunchKernelGGL(centres_minmax_kernel, dim3(red_blocks), dim3(kBT), 0, st, sph7_dev, n, centres, partial);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(kBT), 0, st, partial, red_blocks, bounds);
  hipLaunchKernelGGL(morton_kernel, dim3(nb_n), dim3(kBT), 0, st, centres, bounds, n, keys[0], vals[0]);
  // 2. stable sort by the 30-bit key (15 passes of 2 bits)
  int cur = 0;
  for (int shift = 0; shift < 30; shift += 2) {
    BVH_HIP(sort_pass(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift, counts, st));
    cur ^= 1;
  }
  hipLaunchKernelGGL(gather_spheres_kernel, dim3(nb_n), dim3(kBT), 0, st, sph7_dev, vals[cur], n, o.L7);
  // 3. radix tree over the sorted keys
  BVH_HIP(hipMemsetAsync(o.parent, 0xFF, sizeof(int) * (size_t)ni, st));   // -1
  hipLaunchKernelGGL(radix_tree_kernel, dim3(nb_ni), dim3(kBT), 0, st, keys[cur], n, o.left, o.right, o.parent);
  // 4. AABB propagation: exactly floor(log2 n) + 2 sweeps from all-zero boxes
  const int sweeps = (int)log2f((float)n) + 2;
  float *pmin = bufmin, *pmax = bufmax, *cmin = o.bmin, *cmax = o.bmax;
  if (sweeps % 2 == 0) {   // the last sweep must land in o.bmin / o.bmax
    pmin = o.bmin; pmax = o.bmax; cmin = bufmin; cmax = bufmax;
  }
  BVH_HIP(hipMemsetAsync(pmin, 0, sizeof(float) * 3 * (size_t)ni, st));
  BVH_HIP(hipMemsetAsync(pmax, 0, sizeof(float) * 3 * (size_t)ni, st));
  for (int s = 0; s < sweeps; ++s) {
    hipLaunchKernelGGL(aabb_sweep_kernel, dim3(nb_ni), dim3(kBT), 0, st, o.L7, o.left, o.right, ni, pmin, pmax, cmin, cmax);
    float *t0 = pmin, *t1 = pmax;
    pmin = cmin; pmax = cmax; cmin = t0; cmax = t1;
  }
  // (after the loop pmin/pmax point at the newest boxes == o.bmin/o.bmax by the parity choice above)
  // 5. depths: top-down sweeps until nothing changes
  BVH_HIP(hipMemsetAsync(depth, 0xFF, sizeof(int) * (size_t)ni, st));
  BVH_HIP(hipMemsetAsync(depth, 0, sizeof(int), st));   // root
  for (int round = 0; round < 16; ++round) {
    BVH_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, st));
    for (int s = 0; s < 8; ++s)
      hipLaunchKernelGGL(depth_sweep_kernel, dim3(nb_ni), dim3(kBT), 0, st, o.parent, ni, depth, flags);
    int changed = 0;
    BVH_HIP(hipMemcpyAsync(&changed, flags, sizeof(int), hipMemcpyDeviceToHost, st));
    BVH_HIP(hipStreamSynchronize(st));
    if (!changed) break;
  }
  // 6. traversal numbering: stable sort of the inner nodes by depth (6 bits)
  BVH_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, st));
  hipLaunchKernelGGL(depth_keys_kernel, dim3(nb_ni), dim3(kBT), 0, st, depth, ni, keys[0], vals[0], flags + 1);
  cur = 0;
  for (int shift = 0; shift < 8; shift += 2) {
    BVH_HIP(sort_pass(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], ni, shift, counts, st));
    cur ^= 1;
  }
  hipLaunchKernelGGL(invert_kernel, dim3(nb_ni), dim3(kBT), 0, st, vals[cur], ni, trav_of);
  hipLaunchKernelGGL(trav_nodes_kernel, dim3(nb_ni), dim3(kBT), 0, st, vals[cur], trav_of, o.left, o.right, o.bmin, o.bmax,
                     ni, o.nodes32, o.nodes64);
  hipLaunchKernelGGL(trav_spheres_kernel, dim3(nb_n), dim3(kBT), 0, st, o.L7, n, o.sph, o.col);
  BVH_HIP(hipGetLastError());
  int maxdepth = 0;
  BVH_HIP(hipMemcpyAsync(&maxdepth, flags + 1, sizeof(int), hipMemcpyDeviceToHost, st));
  BVH_HIP(hipStreamSynchronize(st));
  *height_out = maxdepth + 1;   // levels of inner nodes == edges on the longest root -> leaf path
  (void)hipFree(scratch);
  return hipSuccess;
}

}  // namespace rtk
