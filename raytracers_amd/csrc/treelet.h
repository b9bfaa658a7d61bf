// treelet.h -- the treelet cut of the traversal copy: numbering and per-node masks, shared by the host builder
// (host_build.cpp), the GPU builder (bvh_build.hip), the pooled kernel's treelet operation (render_kernels.hip) and
// the CPU checks (tools/treelet_check.cpp).  No HIP in this header.
//
// Why.  bvh_fold (futhark/bvh.fut:61-84) tests a leaf iff every ancestor's box passes aabb_hit with the FIXED interval
// (0, 1e9) (futhark/ray.fut:77): whether a box passes does not depend on when, or by whom, it is tested.  A wave that
// serves only a few rays (the long bounce chains that bound a frame's time) can therefore test the boxes of several tree
// LEVELS at once, speculatively, and decide afterwards which nodes were reached -- one dependent wave operation per D levels
// instead of one per level (or two).
//
// The cut.  Inner nodes at depths 0, D, 2D, ... are treelet ROOTS; a treelet is its root plus the root's descendants of
// relative depth < D (at most 2^D - 1 nodes).  The traversal copy numbers the nodes treelet by treelet -- treelets ordered by
// (depth of the root, canonical index of the root), nodes inside a treelet by their heap index (root 0, children of h are
// 2h + 1 and 2h + 2), compacted -- so the node at position p of the treelet rooted at traversal index R is R + p, and
// the part staged in LDS (a prefix) still holds the levels nearest the root.  A treelet operation gives 2^D lanes to an item
// (a treelet root whose own box passed): lane p reads record R + p and tests the boxes of BOTH children of that node.
//
// Per node, in the two spare dwords of its 64-byte record ({R.lo, mask_l} {R.hi, mask_r}, see rt_host.hpp):
//   mask_l = (positions of the ancestors inside the treelet whose LEFT child is on the path to this node) | 1 << own position
//            | bit 31 when the node's children lie outside the treelet (relative depth D - 1: passing inner children are
//            the next treelets' roots)
//   mask_r = (... whose RIGHT child is on the path) | 1 << own position
// The own position sits in BOTH masks (an ancestor's in exactly one): a lane at position p that reads a record of another
// treelet -- the treelet has fewer than 2^D - 1 nodes -- finds (mask_l & mask_r) != 1 << p and stays out.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_TL_HD __host__ __device__ __forceinline__
#else
#define RT_TL_HD inline
#endif

namespace rtk {

constexpr int kTreeletMaxDepth = 5;             // 2^5 - 1 = 31 nodes: positions fit bits 0..30
#ifndef RT_TREELET_DEPTH
#define RT_TREELET_DEPTH 4   // (round 6: 4 levels -- a lone ray's fold is 4 dependent treelet operations instead of 7; 2: rounds 3-5.  -DRT_TREELET_DEPTH=n builds another cut, 1 .. 4)
#endif
constexpr int kTreeletDepth = RT_TREELET_DEPTH;   // the cut both builders make: a node whose depth is a multiple of this, and its descendants of relative depth < this
constexpr uint32_t kTlFrontier = 0x80000000u;   // in mask_l
constexpr uint32_t kTlPosBits = 0x7fffffffu;

RT_TL_HD int tl_popc(uint32_t v) { return __builtin_popcount(v); }

// position of the node with heap index h in a treelet whose nodes' heap indices are the set bits of occ
RT_TL_HD int tl_pos(uint32_t occ, int h) { return tl_popc(occ & ((1u << h) - 1u)); }

struct TlMasks { uint32_t l, r; };
// builder side: the two mask dwords of the node at heap index h (relative depth = floor(log2(h + 1)))
RT_TL_HD TlMasks tl_masks(uint32_t occ, int h, int D) {
  const uint32_t self = 1u << tl_pos(occ, h);
  TlMasks m{self, self};
  for (int c = h; c > 0;) {
    const int par = (c - 1) >> 1;
    const uint32_t bit = 1u << tl_pos(occ, par);
    if (c & 1) m.l |= bit;   // odd heap index: a left child
    else m.r |= bit;
    c = par;
  }
  if (h >= (1 << (D - 1)) - 1) m.l |= kTlFrontier;
  return m;
}

// The GPU builder (bvh_build.hip) carries a node's place in its treelet next to its traversal index: index in bits 0..26, the node's HEAP
// index inside its treelet in bits 27..30 (cuts of at most 4 levels: heap index <= 14); bit 31 stays clear (the place is also carried in
// signed ints).  rt_scene_from_spheres admits at most 2^kMaxSpheresLog2 spheres: every inner-node index then fits the index field.
constexpr int kMaxSpheresLog2 = 26;
constexpr int kTlIndexBits = 27;
static_assert(kTlIndexBits > kMaxSpheresLog2 && kTlIndexBits + 4 <= 31, "a node's place: index field + 4 bits of heap index in a non-negative int");
static_assert(kTreeletDepth >= 1 && kTreeletDepth <= 4, "the GPU builder packs a heap index of at most 4 bits");
constexpr uint32_t kTlIndexMask = (1u << kTlIndexBits) - 1u;
RT_TL_HD uint32_t tl_pack_place(uint32_t index, int heap) { return index | ((uint32_t)heap << kTlIndexBits); }
RT_TL_HD int tl_place_heap(uint32_t place) { return (int)(place >> kTlIndexBits); }
RT_TL_HD int tl_heap_level(int heap) { return 31 - __builtin_clz((unsigned)heap + 1u); }   // relative depth of a heap index

// kernel side: lane at position `pos` of its group read masks (ml, mr); hl / hr are the group's bits (bit p = the lane at
// position p found the box of its node's left / right child passing).  true: this lane holds a node of the item's
// treelet and every box on the path from the treelet's root to it passed.
RT_TL_HD bool tl_valid(uint32_t ml, uint32_t mr, int pos) { return ((ml & mr) & kTlPosBits) == (1u << pos); }
RT_TL_HD bool tl_reached(uint32_t ml, uint32_t mr, int pos, uint32_t hl, uint32_t hr) {
  const uint32_t self = 1u << pos;
  const uint32_t al = (ml & kTlPosBits) ^ self, ar = mr ^ self;
  return tl_valid(ml, mr, pos) && (hl & al) == al && (hr & ar) == ar;
}
RT_TL_HD bool tl_frontier(uint32_t ml) { return (ml & kTlFrontier) != 0u; }

}  // namespace rtk
