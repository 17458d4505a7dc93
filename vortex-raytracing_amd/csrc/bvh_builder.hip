// BLAS construction on the GPU, in the reference's node format.
//
// Reference: tests/regression/raytracing/bvh.cpp:30-264 -- BVH::build (binned SAH, binary), the collapse to 4-wide nodes and
// the quantiser, all host code run once per mesh at scene load; triangles are reordered in place so that a leaf is a range
// (bvh.cpp:126-128).  csrc/scene_builder.cpp is this package's CPU counterpart (threaded SAH, the quality builder).  This file
// is the builder for geometry that changes per frame: some hundred and fifty short launches over data that never leaves HBM.
//
//   1. centroid bounds              one pass, wavefront reduction + 6 atomics per workgroup
//   2. 63-bit Morton keys           21 bits per axis of the triangle's box centre (extended order)
//   3. radix sort (key, index)      rocPRIM device sort, 8 passes over 12 bytes per triangle
//   4. clustering                   PLOC: rounds of "merge the pairs that are each other's nearest neighbour within 8 positions of
//                                   the Morton order" (nearest = smallest surface area of the union), boxes and the SAH dynamic
//                                   programme of the 4-wide collapse computed as the nodes are made; the last 1,024 clusters in
//                                   one workgroup.  (Rounds 1-2: Karras' binary radix tree + a bottom-up box pass.)
//   4b. the binary tree optimised by parallel reinsertion (round 5; Meister & Bittner 2018): four iterations of "every node looks for the
//      place where it would cost least, the largest gains that do not touch each other's links are applied, boxes refitted bottom-up",
//      the last refit recomputing the dynamic programme.  6.3 ms per million triangles in all against 1.5 ms without; the headline
//      frame on the tree 8.4 Grays/s against 7.95, 8.9 with the children of a wide node sorted along its widest axis (step 5; the CPU
//      builder's optimised tree: 8.8).  VXRT_BVH_REINSERT=0 switches it off.
//   5. collapse to 4-wide + quantise + emit, level by level, following the dynamic programme's choices; every child gets its
//      range of the final triangle order from its parent; a subtree marked as a leaf (<= leaf_max triangles, and cheaper as
//      a leaf) lists its triangles there.  Children are allocated after their parent, which is what vxrt_accel_build's
//      validation asks of any tree.
//   6. triangles (and their shading records) gathered into the final order.
//
// Quantisation follows the format (decode = origin + ldexp(q, e), rt_traversal.cpp:61-67) and is conservative by
// construction: every q is checked against the decode's own rounding and the exponent is raised until all children fit.
// Parity for a builder is what it is for scene_builder.cpp (the reference builder reads uninitialised bounds, bvh.cpp:79-86,
// so its tree is not reproducible from its algorithm): structural invariants + every ray finding the brute-force distance.
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <mutex>
#include <vector>
#include "rt_types.h"
#include "../../include/vortex_hip.h"

#ifndef BB_EXTENDED_MORTON
#define BB_EXTENDED_MORTON 1
#endif

namespace {

constexpr int BB_RI_MAX_ITERS = 32;  // reinsertion iterations at most (step 4b)
constexpr int BB_RI_STEPS = 1 << 15;  // nodes one mover's search visits at most (the usual search: some tens)
constexpr int BB_RI_LISTS = 512;     // heights the refit of step 4b follows (a binary tree deeper than that is reported, not emitted)
#ifndef BB_CHILD_ORDER
#define BB_CHILD_ORDER 3          // slots of a wide node: children by centre along the node's widest axis, ascending (see bb_collapse_kernel)
#endif
#ifndef BB_RI_ITERS
#define BB_RI_ITERS 4
#endif
constexpr int BB_MAX_LEVELS = 34;   // launches of the collapse pass; a tree deeper than RT_MAX_LEVELS is reported, not emitted half-way

struct Box3 { float lx, ly, lz, hx, hy, hz; };

__device__ __forceinline__ int f2ord(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

__device__ __forceinline__ Box3 tri_box(const float* __restrict__ t) {
  Box3 b;
  b.lx = fminf(fminf(t[0], t[3]), t[6]); b.hx = fmaxf(fmaxf(t[0], t[3]), t[6]);
  b.ly = fminf(fminf(t[1], t[4]), t[7]); b.hy = fmaxf(fmaxf(t[1], t[4]), t[7]);
  b.lz = fminf(fminf(t[2], t[5]), t[8]); b.hz = fmaxf(fmaxf(t[2], t[5]), t[8]);
  return b;
}

// bounds to +-inf, counters to zero except: one node allocated (the root, output slot 0), one item on level 0 = (binary root, slot 0)
// primitive i of the build: a triangle (stride 9) or, for the TLAS, an instance's world-space box (stride 6)
__device__ __forceinline__ Box3 prim_box(const float* __restrict__ prims, uint32_t i, bool boxes) {
  if (!boxes) return tri_box(prims + (size_t)i * 9);
  const float* p = prims + (size_t)i * 6;
  Box3 b; b.lx = p[0]; b.ly = p[1]; b.lz = p[2]; b.hx = p[3]; b.hy = p[4]; b.hz = p[5];
  return b;
}

__global__ void bb_init_kernel(int* cb, uint32_t* counters, uint32_t n_counters, uint4* level0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3) cb[i] = 0x7fffffff;
  else if (i < 6) cb[i] = (int)0x80000000;
  if (i < n_counters) counters[i] = (i == 0 || i == 8) ? 1u : 0u;
  if (i == 0) level0[0] = make_uint4(0u, 0u, 0u, 0u);   // (binary node 0 -- the root, or the only leaf when n == 1 -- into slot 0, triangles from 0)
}

// ---- 1. bounds of the box centres ----
__global__ __launch_bounds__(256) void bb_bounds_kernel(const float* __restrict__ tri, uint32_t n, int* __restrict__ cb, bool boxes) {
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const Box3 b = prim_box(tri, i, boxes);
    const float c[3] = {0.5f * (b.lx + b.hx), 0.5f * (b.ly + b.hy), 0.5f * (b.lz + b.hz)};
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], c[a]); hi[a] = fmaxf(hi[a], c[a]); }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_down(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off)); }
  // one set of atomics per workgroup (six contended addresses: per wavefront they cost more than the pass itself)
  __shared__ float s_lo[4][3], s_hi[4][3];
  if ((threadIdx.x & 63u) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const float l = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
    const float h = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
    if (l <= h) { atomicMin(cb + a, f2ord(l)); atomicMax(cb + 3 + a, f2ord(h)); }
  }
}

// ---- 2. Morton keys ----
__device__ __forceinline__ uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x001f00000000ffffull;
  x = (x | x << 16) & 0x001f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// axis and bit of every key position (one thread: 63 steps over three numbers)
__global__ void bb_axis_sequence_kernel(const int* __restrict__ cb, uint32_t* __restrict__ seq) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float e3[3] = {ord2f(cb[3]) - ord2f(cb[0]), ord2f(cb[4]) - ord2f(cb[1]), ord2f(cb[5]) - ord2f(cb[2])};
  int used[3] = {0, 0, 0};
  for (int b = 0; b < 63; ++b) {
    int a = -1; float best = -1.0f;
    for (int k = 0; k < 3; ++k) if (used[k] < 21 && (a < 0 || e3[k] > best)) { best = e3[k]; a = k; }   // (a NaN extent: the first axis with bits left)
    seq[b] = (uint32_t)a | ((uint32_t)(20 - used[a]) << 2);
    used[a]++; e3[a] *= 0.5f;
  }
}

__global__ __launch_bounds__(256) void bb_morton_kernel(const float* __restrict__ tri, uint32_t n, const int* __restrict__ cb, const uint32_t* __restrict__ seq,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, bool boxes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Box3 b = prim_box(tri, i, boxes);
  const float c[3] = {0.5f * (b.lx + b.hx), 0.5f * (b.ly + b.hy), 0.5f * (b.lz + b.hz)};
  uint32_t q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = ord2f(cb[a]), hi = ord2f(cb[3 + a]);
    const float ext = hi - lo;
    float t = ext > 0.0f ? (c[a] - lo) / ext * 2097152.0f : 0.0f;
    if (!(t >= 0.0f)) t = 0.0f;            // (also a NaN centre)
    if (t > 2097151.0f) t = 2097151.0f;
    q[a] = (uint32_t)t;
  }
#if BB_EXTENDED_MORTON
  // extended Morton order (Vinkler et al. 2017): the next bit always comes from the axis whose cell is still the longest, so an
  // elongated scene is not cut across its short axes as often as along its long one (plain interleaving gives every axis 21 bits
  // whatever its extent).  The axis sequence depends on the bounds only: bb_axis_sequence_kernel computed it once
  // (seq[b] = axis | shift << 2 for key bit b, most significant first; 63 scalar loads here).
  uint64_t key = 0;
#pragma unroll 9
  for (int b = 0; b < 63; ++b) {
    const uint32_t sq = seq[b], a = sq & 3u, sh = sq >> 2;
    const uint32_t qa = a == 0 ? q[0] : (a == 1 ? q[1] : q[2]);
    key = (key << 1) | ((qa >> sh) & 1u);
  }
  keys[i] = key;
#else
  keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
#endif
  vals[i] = i;
}

// ---- 4. clustering: PLOC (Meister & Bittner, "Parallel locally-ordered clustering for BVH construction", 2018) ----
// The clusters -- at first one per triangle, in Morton order -- are merged bottom-up in rounds: every cluster looks at the
// BB_RADIUS clusters on either side of it in the array for the one whose union with it has the smallest surface area; two clusters
// that choose each other become a node; the array is compacted in place order and the round repeats.  Unlike the binary radix tree
// of rounds 1-2 (which splits the Morton range at its highest differing bit, a spatial median) the pairs are chosen by the area of
// the result, which is what the traversal pays for.  Everything is deterministic: ties go to the lower position, node ids and the
// compacted positions come from prefix sums, nothing from the order in which atomics land.
//   node ids: leaf j (sorted position) is n-1+j; internal ids are handed out downwards from n-2, so the last merge -- the root -- is 0.
// A round that merges fewer than 1/16 of the clusters is followed by one of forced pairing (positions 2k, 2k+1): inputs where
// nearly nothing is mutual (a thousand coincident triangles: every union has the same area) still finish in O(log n) rounds.
//
// The SAH-optimal collapse to 4-wide nodes (the dynamic programme of Ylitie, Karras & Laine, "Efficient incoherent ray traversal on
// GPUs through compressed wide BVHs", 2017, for width 4) rides in the merge: f[j] = least cost of covering the new node's subtree
// with at most j child slots of a wide node, cost = expected bytes fetched per random ray (52 B per node record, 36 B per triangle,
// times surface area); `plan` records the choices for the top-down pass (step 6).  A subtree of <= leaf_max triangles may become
// a leaf where that is cheaper than a node over it.
#ifndef BB_RADIUS_N
#define BB_RADIUS_N 8
#endif
constexpr int BB_RADIUS = BB_RADIUS_N;   // search radius.  The CPU prototype's surface-area cost ranks 4 < 8 < 16 (tools/ploc_prototype.py); the frame on the GPU says
                                         // 8: 7.43-7.55 Grays/s against 7.26-7.35 at 4, 7.25 at 6, 7.32 at 10, 7.43 at 12, 7.42 at 16 (profiles/r03_v_ploc_radius.txt); +0.13 ms
constexpr int BB_TAIL = 1024;         // clusters the single-workgroup tail takes over at
constexpr float BB_NODE_COST = 52.0f, BB_TRI_COST = 36.0f;   // bytes

// node of the binary tree: box, children, the dynamic programme's costs and choices
struct __attribute__((aligned(16))) BNode {
  float lx, ly, lz; uint32_t left;
  float hx, hy, hz; uint32_t right;
  float f1, f2, f3, f4;
  uint32_t count, plan, pad0, pad1;
};
static_assert(sizeof(BNode) == 64, "record size");
// active cluster of a round
struct __attribute__((aligned(16))) Cluster {
  float lx, ly, lz; uint32_t id;
  float hx, hy, hz; uint32_t count;
  float f1, f2, f3, f4;
};
static_assert(sizeof(Cluster) == 48, "cluster size");

// plan bits: [1:0] a2, [3:2] a3, [5:4] a4 (slots given to the left child when the node's two children share 2 / 3 / 4 slots),
// [6] [7] [8] "stays one child" with 2 / 3 / 4 slots offered, [9] leaf
constexpr uint32_t PLAN_LEAF = 1u << 9;
__device__ __forceinline__ uint32_t plan_a(uint32_t plan, uint32_t j) { return (plan >> (2u * (j - 2u))) & 3u; }
__device__ __forceinline__ bool plan_self(uint32_t plan, uint32_t j) { return (plan >> (4u + j)) & 1u; }

__device__ __forceinline__ float box_area(float lx, float ly, float lz, float hx, float hy, float hz) {
  const float x = hx - lx, y = hy - ly, z = hz - lz;
  return x * y + y * z + z * x;
}

__device__ __forceinline__ Cluster bb_merge(const Cluster& A, const Cluster& B, uint32_t id, uint32_t leaf_max, float tri_cost, uint32_t& plan) {
  Cluster c;
  c.lx = fminf(A.lx, B.lx); c.ly = fminf(A.ly, B.ly); c.lz = fminf(A.lz, B.lz);
  c.hx = fmaxf(A.hx, B.hx); c.hy = fmaxf(A.hy, B.hy); c.hz = fmaxf(A.hz, B.hz);
  c.id = id; c.count = A.count + B.count;
  const float ar = box_area(c.lx, c.ly, c.lz, c.hx, c.hy, c.hz);
  const float g2 = A.f1 + B.f1;
  float g3 = A.f1 + B.f2; uint32_t a3 = 1;
  if (A.f2 + B.f1 < g3) { g3 = A.f2 + B.f1; a3 = 2; }
  float g4 = A.f1 + B.f3; uint32_t a4 = 1;
  if (A.f2 + B.f2 < g4) { g4 = A.f2 + B.f2; a4 = 2; }
  if (A.f3 + B.f1 < g4) { g4 = A.f3 + B.f1; a4 = 3; }
  const float c_node = ar * BB_NODE_COST + g4;
  const float c_leaf = c.count <= leaf_max ? ar * (BB_NODE_COST + tri_cost * (float)c.count) : __builtin_inff();
  const bool leaf = c_leaf <= c_node;
  c.f1 = leaf ? c_leaf : c_node;
  const bool s2 = c.f1 <= g2, s3 = c.f1 <= g3, s4 = c.f1 <= g4;
  c.f2 = s2 ? c.f1 : g2; c.f3 = s3 ? c.f1 : g3; c.f4 = s4 ? c.f1 : g4;
  plan = 1u | (a3 << 2) | (a4 << 4) | ((uint32_t)s2 << 6) | ((uint32_t)s3 << 7) | ((uint32_t)s4 << 8) | (leaf ? PLAN_LEAF : 0u);
  return c;
}

__device__ __forceinline__ void bb_store_node(BNode* __restrict__ rec, const Cluster& c, uint32_t left, uint32_t right, uint32_t plan) {
  float4* o = (float4*)(rec + c.id);
  o[0] = make_float4(c.lx, c.ly, c.lz, __uint_as_float(left));
  o[1] = make_float4(c.hx, c.hy, c.hz, __uint_as_float(right));
  o[2] = make_float4(c.f1, c.f2, c.f3, c.f4);
  o[3] = make_float4(__uint_as_float(c.count), __uint_as_float(plan), 0.0f, 0.0f);
}
__device__ __forceinline__ Cluster bb_load_cluster(const Cluster* __restrict__ p) {
  const float4* q = (const float4*)p;
  const float4 a = q[0], b = q[1], c = q[2];
  Cluster r;
  r.lx = a.x; r.ly = a.y; r.lz = a.z; r.id = __float_as_uint(a.w);
  r.hx = b.x; r.hy = b.y; r.hz = b.z; r.count = __float_as_uint(b.w);
  r.f1 = c.x; r.f2 = c.y; r.f3 = c.z; r.f4 = c.w;
  return r;
}
__device__ __forceinline__ void bb_store_cluster(Cluster* __restrict__ p, const Cluster& c) {
  float4* q = (float4*)p;
  q[0] = make_float4(c.lx, c.ly, c.lz, __uint_as_float(c.id));
  q[1] = make_float4(c.hx, c.hy, c.hz, __uint_as_float(c.count));
  q[2] = make_float4(c.f1, c.f2, c.f3, c.f4);
}

// state of the clustering on the device: [0] clusters, [1] next internal id + 1, [2] forced pairing in this round, [3] rounds done,
// [4] [5] clusters / next id at the start of the round being applied
enum { ST_N = 0, ST_NEXT = 1, ST_FORCED = 2, ST_ROUNDS = 3, ST_N_OLD = 4, ST_NEXT_OLD = 5, ST_WORDS = 8 };

__global__ __launch_bounds__(256) void bb_leaves_kernel(const float* __restrict__ prims, const uint32_t* __restrict__ vals, uint32_t n, bool boxes,
                                                          float tri_cost, Cluster* __restrict__ cl, BNode* __restrict__ rec, uint32_t* __restrict__ st) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) { st[ST_N] = n; st[ST_NEXT] = n - 1; st[ST_FORCED] = 0; st[ST_ROUNDS] = 0; }
  if (j >= n) return;
  const Box3 b = prim_box(prims, vals[j], boxes);
  Cluster c;
  c.lx = b.lx; c.ly = b.ly; c.lz = b.lz; c.hx = b.hx; c.hy = b.hy; c.hz = b.hz;
  c.id = n - 1 + j; c.count = 1;
  c.f1 = c.f2 = c.f3 = c.f4 = box_area(b.lx, b.ly, b.lz, b.hx, b.hy, b.hz) * (BB_NODE_COST + tri_cost);
  bb_store_cluster(cl + j, c);
  bb_store_node(rec, c, 0xffffffffu, 0xffffffffu, PLAN_LEAF);
}

// nearest neighbour of position g among [g - R, g + R] of boxes held in LDS (entry k of the tile = position s0 + k); -1 = none
__device__ __forceinline__ int bb_nearest(const float (*sb)[6], int k, int g, int m, int s0, bool forced) {
  if (forced) { const int j = g ^ 1; return j < m ? j : -1; }
  float best = __builtin_inff(); int bj = -1;
  const float lx = sb[k][0], ly = sb[k][1], lz = sb[k][2], hx = sb[k][3], hy = sb[k][4], hz = sb[k][5];
#pragma unroll
  for (int d = -BB_RADIUS; d <= BB_RADIUS; ++d) {
    if (d == 0) continue;
    const int j = g + d;
    if (j < 0 || j >= m) continue;
    const float* o = sb[j - s0];
    const float a = box_area(fminf(lx, o[0]), fminf(ly, o[1]), fminf(lz, o[2]), fmaxf(hx, o[3]), fmaxf(hy, o[4]), fmaxf(hz, o[5]));
    if (bj < 0 || a < best || (best != best && a == a)) { best = a; bj = j; }   // (ascending j, strict <: ties go to the lower position; a NaN area never wins over a number, first candidate or not)
  }
  return bj;
}

// decision of a round per position: bits [1:0] 0 stays, 1 leads a merge, 2 absorbed; bits [7:2] partner - position + BB_RADIUS
enum { DEC_STAY = 0, DEC_LEAD = 1, DEC_GONE = 2 };
constexpr int BB_TILE = 256, BB_HALO = 2 * BB_RADIUS;

__global__ __launch_bounds__(256) void bb_ploc_decide_kernel(const Cluster* __restrict__ cl, const uint32_t* __restrict__ st,
                                                               uint8_t* __restrict__ dec, uint2* __restrict__ tile_counts) {
  const int m = (int)st[ST_N];
  const bool forced = st[ST_FORCED] != 0u;
  const int n_tiles = (m + BB_TILE - 1) / BB_TILE;
  __shared__ float sb[BB_TILE + 2 * BB_HALO][6];
  __shared__ int s_nn[BB_TILE + 2 * BB_HALO];
  __shared__ uint32_t s_cnt[4][2];
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int s0 = t * BB_TILE - BB_HALO;   // position of LDS entry 0
    for (int k = threadIdx.x; k < BB_TILE + 2 * BB_HALO; k += 256) {
      const int g = s0 + k;
      if (g >= 0 && g < m) {
        const float4* q = (const float4*)(cl + g);
        const float4 a = q[0], b = q[1];
        sb[k][0] = a.x; sb[k][1] = a.y; sb[k][2] = a.z; sb[k][3] = b.x; sb[k][4] = b.y; sb[k][5] = b.z;
      }
    }
    __syncthreads();
    // neighbours of the tile's positions and of BB_RADIUS positions either side (their choice decides whether a pair is mutual)
    for (int k = BB_RADIUS + threadIdx.x; k < BB_TILE + 2 * BB_HALO - BB_RADIUS; k += 256) {
      const int g = s0 + k;
      s_nn[k] = (g >= 0 && g < m) ? bb_nearest(sb, k, g, m, s0, forced) : -1;
    }
    __syncthreads();
    const int k = BB_HALO + threadIdx.x, g = s0 + k;
    uint32_t state = DEC_STAY; int j = -1;
    if (g < m) {
      j = s_nn[k];
      if (j >= 0 && s_nn[j - s0] == g) state = g < j ? DEC_LEAD : DEC_GONE;
      dec[g] = (uint8_t)(state | ((uint32_t)(j >= 0 ? j - g + BB_RADIUS : 0) << 2));
    }
    const unsigned long long keep = __ballot(g < m && state != DEC_GONE), lead = __ballot(g < m && state == DEC_LEAD);
    if ((threadIdx.x & 63u) == 0) { s_cnt[threadIdx.x >> 6][0] = (uint32_t)__popcll(keep); s_cnt[threadIdx.x >> 6][1] = (uint32_t)__popcll(lead); }
    __syncthreads();
    if (threadIdx.x == 0)
      tile_counts[t] = make_uint2(s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0], s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1]);
    __syncthreads();
  }
}

// exclusive prefix of the tiles' (kept, merging) counts, and the state of the next round; one workgroup
__global__ __launch_bounds__(1024) void bb_ploc_scan_kernel(const uint2* __restrict__ tile_counts, uint2* __restrict__ tile_base, uint32_t* __restrict__ st) {
  const uint32_t m = st[ST_N];
  const uint32_t n_tiles = (m + BB_TILE - 1) / BB_TILE;
  __shared__ uint2 s_w[16];
  __shared__ uint2 s_carry;
  if (threadIdx.x == 0) s_carry = make_uint2(0u, 0u);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (uint32_t base = 0; base < n_tiles; base += 1024u) {
    const uint32_t t = base + threadIdx.x;
    const uint2 v = t < n_tiles ? tile_counts[t] : make_uint2(0u, 0u);
    uint2 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t x = __shfl_up(inc.x, off), y = __shfl_up(inc.y, off);
      if (lane >= (uint32_t)off) { inc.x += x; inc.y += y; }
    }
    if (lane == 63u) s_w[wv] = inc;
    __syncthreads();
    uint2 woff = s_carry;
    for (uint32_t k = 0; k < wv; ++k) { woff.x += s_w[k].x; woff.y += s_w[k].y; }
    if (t < n_tiles) tile_base[t] = make_uint2(woff.x + inc.x - v.x, woff.y + inc.y - v.y);
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry = make_uint2(woff.x + inc.x, woff.y + inc.y);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint2 tot = s_carry;
    st[ST_N_OLD] = m; st[ST_NEXT_OLD] = st[ST_NEXT];
    st[ST_N] = tot.x; st[ST_NEXT] -= tot.y;
    st[ST_FORCED] = (st[ST_FORCED] == 0u && tot.y < m / 16u) ? 1u : 0u;
    st[ST_ROUNDS] += 1u;
  }
}

__global__ __launch_bounds__(256) void bb_ploc_apply_kernel(const Cluster* __restrict__ cl, Cluster* __restrict__ out, const uint8_t* __restrict__ dec,
                                                              const uint2* __restrict__ tile_base, const uint32_t* __restrict__ st,
                                                              BNode* __restrict__ rec, uint32_t leaf_max, float tri_cost) {
  const int m = (int)st[ST_N_OLD];
  const uint32_t next = st[ST_NEXT_OLD];
  const int n_tiles = (m + BB_TILE - 1) / BB_TILE;
  __shared__ uint32_t s_cnt[4][2];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int g = t * BB_TILE + (int)threadIdx.x;
    const uint32_t d = g < m ? dec[g] : (uint32_t)DEC_GONE;
    const uint32_t state = d & 3u;
    const unsigned long long keep = __ballot(state != DEC_GONE), lead = __ballot(state == DEC_LEAD);
    if (lane == 0) { s_cnt[wv][0] = (uint32_t)__popcll(keep); s_cnt[wv][1] = (uint32_t)__popcll(lead); }
    __syncthreads();
    uint2 base = tile_base[t];
    for (uint32_t k = 0; k < wv; ++k) { base.x += s_cnt[k][0]; base.y += s_cnt[k][1]; }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t p = base.x + (uint32_t)__popcll(keep & below);
    if (state == DEC_STAY) {
      bb_store_cluster(out + p, bb_load_cluster(cl + g));
    } else if (state == DEC_LEAD) {
      const int j = g + (int)(d >> 2) - BB_RADIUS;
      const Cluster A = bb_load_cluster(cl + g), B = bb_load_cluster(cl + j);
      uint32_t plan;
      const Cluster c = bb_merge(A, B, next - 1u - (base.y + (uint32_t)__popcll(lead & below)), leaf_max, tri_cost, plan);
      bb_store_node(rec, c, A.id, B.id, plan);
      bb_store_cluster(out + p, c);
    }
  }
}

// the last BB_TAIL clusters: all remaining rounds in one workgroup, the clusters in LDS
__global__ __launch_bounds__(1024) void bb_ploc_tail_kernel(const Cluster* __restrict__ cl, uint32_t* __restrict__ st, BNode* __restrict__ rec,
                                                              uint32_t leaf_max, float tri_cost) {
  __shared__ float sb[BB_TAIL][6];
  __shared__ float sf[BB_TAIL][4];
  __shared__ uint32_t s_id[BB_TAIL], s_count[BB_TAIL];
  __shared__ int s_nn[BB_TAIL];
  __shared__ uint32_t s_w[16][2];
  int m = (int)st[ST_N];
  uint32_t next = st[ST_NEXT];
  bool forced = st[ST_FORCED] != 0u;
  uint32_t rounds = st[ST_ROUNDS];
  if (m > BB_TAIL) return;   // (the host only launches it below that)
  const int i = (int)threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  if (i < m) {
    const Cluster c = bb_load_cluster(cl + i);
    sb[i][0] = c.lx; sb[i][1] = c.ly; sb[i][2] = c.lz; sb[i][3] = c.hx; sb[i][4] = c.hy; sb[i][5] = c.hz;
    sf[i][0] = c.f1; sf[i][1] = c.f2; sf[i][2] = c.f3; sf[i][3] = c.f4;
    s_id[i] = c.id; s_count[i] = c.count;
  }
  __syncthreads();
  while (m > 1) {
    s_nn[i] = i < m ? bb_nearest(sb, i, i, m, 0, forced) : -1;
    __syncthreads();
    uint32_t state = DEC_GONE; int j = -1;
    if (i < m) {
      j = s_nn[i];
      state = (j >= 0 && s_nn[j] == i) ? (i < j ? DEC_LEAD : DEC_GONE) : DEC_STAY;
    }
    Cluster c;
    uint32_t left = 0, right = 0;
    if (state != DEC_GONE) {
      c.lx = sb[i][0]; c.ly = sb[i][1]; c.lz = sb[i][2]; c.hx = sb[i][3]; c.hy = sb[i][4]; c.hz = sb[i][5];
      c.f1 = sf[i][0]; c.f2 = sf[i][1]; c.f3 = sf[i][2]; c.f4 = sf[i][3];
      c.id = s_id[i]; c.count = s_count[i];
    }
    Cluster B;
    if (state == DEC_LEAD) {
      B.lx = sb[j][0]; B.ly = sb[j][1]; B.lz = sb[j][2]; B.hx = sb[j][3]; B.hy = sb[j][4]; B.hz = sb[j][5];
      B.f1 = sf[j][0]; B.f2 = sf[j][1]; B.f3 = sf[j][2]; B.f4 = sf[j][3];
      B.id = s_id[j]; B.count = s_count[j];
    }
    const unsigned long long keep = __ballot(state != DEC_GONE), lead = __ballot(state == DEC_LEAD);
    if (lane == 0) { s_w[wv][0] = (uint32_t)__popcll(keep); s_w[wv][1] = (uint32_t)__popcll(lead); }
    __syncthreads();   // (also: every read of the old arrays is done)
    uint32_t bk = 0, bl = 0, tk = 0, tl = 0;
    for (uint32_t k = 0; k < 16u; ++k) { if (k < wv) { bk += s_w[k][0]; bl += s_w[k][1]; } tk += s_w[k][0]; tl += s_w[k][1]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t p = bk + (uint32_t)__popcll(keep & below);
    if (state == DEC_LEAD) {
      left = c.id; right = B.id;
      uint32_t plan;
      c = bb_merge(c, B, next - 1u - (bl + (uint32_t)__popcll(lead & below)), leaf_max, tri_cost, plan);
      bb_store_node(rec, c, left, right, plan);
    }
    if (state != DEC_GONE) {
      sb[p][0] = c.lx; sb[p][1] = c.ly; sb[p][2] = c.lz; sb[p][3] = c.hx; sb[p][4] = c.hy; sb[p][5] = c.hz;
      sf[p][0] = c.f1; sf[p][1] = c.f2; sf[p][2] = c.f3; sf[p][3] = c.f4;
      s_id[p] = c.id; s_count[p] = c.count;
    }
    forced = !forced && tl < (uint32_t)m / 16u;
    m = (int)tk; next -= tl; ++rounds;
    __syncthreads();
  }
  if (threadIdx.x == 0) { st[ST_N] = (uint32_t)m; st[ST_NEXT] = next; st[ST_FORCED] = forced ? 1u : 0u; st[ST_ROUNDS] = rounds; }
}

// ---- 4b. the binary tree optimised by reinsertion, every node at once ----
// (Meister & Bittner, "Parallel reinsertion for bounding volume hierarchy optimization", 2018; csrc/scene_builder.cpp runs the serial form
// of Bittner et al. 2013 on the CPU tree.)  One iteration:
//   search   every node `in` (not the root, not a child of the root) looks for the node `out` next to which it would cost least: cut out
//            together with its parent p (p's other child takes p's place), p re-used as the new parent of (out, in).  The gain is what the
//            cut saves -- p's area and the shrinking of the ancestors below the common ancestor ("pivot") -- minus what the insertion adds:
//            area(out + in) and the growth of out's ancestors below the pivot.  The walk goes up from p pivot by pivot and searches the
//            subtree on the other side of each with branch and bound (a subtree whose best case cannot beat the best gain so far is left).
//   lock     a node with a positive gain claims the six nodes whose child / parent links its move rewrites (in, p, p's other child, p's
//            parent, out, out's parent) with atomicMax of (gain, in): the larger gain wins a contested node, nothing depends on timing.
//   resolve  a claimant that holds all six is a winner -- unless an ancestor of its `out` is a winner with a larger key: two disjoint
//            subtrees that each move into the other would close a cycle cut off from the root; of any such ring the member below
//            the largest key stands back, so no ring closes.
//   apply    winners rewrite their links (disjoint by the locks).
//   refit    boxes bottom-up (one thread per leaf walks up; the second to arrive at a node computes it); the last refit also recomputes
//            the counts and the dynamic programme of step 4 for the collapse.
// Node ids keep their classes (root 0, internal < n-1, leaves >= n-1): a move re-uses p.
constexpr int BB_RI_STACK = 48;
constexpr uint32_t BB_NONE = 0xffffffffu;

struct RiArgs {
  BNode* rec;
  uint32_t* parent;               // 2n-1
  unsigned long long* lock;       // 2n-1: (gain bits << 32) | claimant
  unsigned long long* best;       // 2n-1: (gain bits << 32) | out; 0 = no move
  uint8_t* win0; uint8_t* win;    // 2n-1 each
  uint32_t* arrived;              // n-1: refit counters
  uint32_t* moves;                // [it]: moves applied in iteration it
  uint32_t n;                     // triangles
  uint32_t it, mod;
  uint32_t leaf_max; float tri_cost;
};

__device__ __forceinline__ void ri_load(const BNode* __restrict__ rec, uint32_t id, Box3& b, uint32_t& l, uint32_t& r) {
  const float4* q = (const float4*)(rec + id);
  const float4 a = q[0], c = q[1];
  b.lx = a.x; b.ly = a.y; b.lz = a.z; l = __float_as_uint(a.w);
  b.hx = c.x; b.hy = c.y; b.hz = c.z; r = __float_as_uint(c.w);
}
__device__ __forceinline__ float ri_area(const Box3& b) { return box_area(b.lx, b.ly, b.lz, b.hx, b.hy, b.hz); }
__device__ __forceinline__ Box3 ri_union(const Box3& a, const Box3& b) {
  Box3 u;
  u.lx = fminf(a.lx, b.lx); u.ly = fminf(a.ly, b.ly); u.lz = fminf(a.lz, b.lz);
  u.hx = fmaxf(a.hx, b.hx); u.hy = fmaxf(a.hy, b.hy); u.hz = fmaxf(a.hz, b.hz);
  return u;
}

__global__ __launch_bounds__(256) void bb_ri_parents_kernel(RiArgs A) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) A.parent[0] = BB_NONE;
  if (i >= A.n - 1u) return;
  const float4* q = (const float4*)(A.rec + i);
  A.parent[__float_as_uint(q[0].w)] = i;
  A.parent[__float_as_uint(q[1].w)] = i;
}

__global__ __launch_bounds__(256) void bb_ri_search_kernel(RiArgs A) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_nodes = 2u * A.n - 1u;
  if (x >= n_nodes) return;
  A.lock[x] = 0ull;
  unsigned long long res = 0ull;
  const uint32_t p = x ? A.parent[x] : 0u;
  if (x != 0u && p != 0u && (A.mod <= 1u || x % A.mod == A.it % A.mod)) {
    Box3 bi; uint32_t t0, t1;
    ri_load(A.rec, x, bi, t0, t1);
    const float a_in = ri_area(bi);
    Box3 bp; uint32_t pl, pr;
    ri_load(A.rec, p, bp, pl, pr);
    float d_rem = ri_area(bp);        // saved by the cut so far: p itself, then the shrinking of the path below the pivot
    float best = 0.0f; uint32_t best_out = BB_NONE;
    uint32_t sn[BB_RI_STACK]; float si[BB_RI_STACK];
    uint32_t prev = x, pivot = p, pvl = pl, pvr = pr;
    Box3 pbox = bp;                   // the pivot's box as it is
    Box3 path; bool have_path = false;   // the box of the path's node below the pivot after the cut
    // (both walks are bounded whatever the links say: BB_RI_LISTS pivots, BB_RI_STEPS nodes searched per mover -- a tree that is one,
    // as the ring guard keeps it, needs neither bound; a search cut short only finds a smaller gain)
    uint32_t steps = 0;
    for (uint32_t up = 0; up < (uint32_t)BB_RI_LISTS; ++up) {
      const uint32_t sib = pvl == prev ? pvr : pvl;
      Box3 bs; uint32_t sl, sr;
      ri_load(A.rec, sib, bs, sl, sr);
      // the subtree on the other side of the pivot
      if (d_rem - a_in > best) {
        int sp = 0;
        uint32_t node = sib; float ind = 0.0f; Box3 bn = bs; uint32_t nl = sl, nr = sr;
        for (;;) {
          const float direct = ri_area(ri_union(bn, bi));
          const float gain = d_rem - ind - direct;
          if (gain > best && !(pivot == p && node == sib)) { best = gain; best_out = node; }   // (next to its own sibling: where it is)
          const float ind2 = ind + direct - ri_area(bn);
          if (++steps > (uint32_t)BB_RI_STEPS) break;
          if (nl != BB_NONE && d_rem - ind2 - a_in > best) {
            if (sp < BB_RI_STACK) { sn[sp] = nr; si[sp] = ind2; ++sp; }
            node = nl; ind = ind2;
            ri_load(A.rec, node, bn, nl, nr);
            continue;
          }
          bool got = false;
          while (sp > 0) {
            --sp;
            if (d_rem - si[sp] - a_in > best) { node = sn[sp]; ind = si[sp]; got = true; break; }
          }
          if (!got) break;
          ri_load(A.rec, node, bn, nl, nr);
        }
      }
      // one level up: the pivot becomes part of the path
      const Box3 shrunk = have_path ? ri_union(path, bs) : bs;   // (p itself is replaced by its other child)
      if (pivot != p) d_rem += ri_area(pbox) - ri_area(shrunk);
      path = shrunk; have_path = true;
      prev = pivot;
      pivot = A.parent[pivot];
      if (pivot == BB_NONE) break;
      ri_load(A.rec, pivot, pbox, pvl, pvr);
    }
    if (best_out != BB_NONE && best > 0.0f) res = ((unsigned long long)__float_as_uint(best) << 32) | best_out;
  }
  A.best[x] = res;
}

// the six nodes whose links the move of x rewrites
struct RiMove { uint32_t p, sib, gp, out, po; };
__device__ __forceinline__ RiMove ri_move(const RiArgs& A, uint32_t x, unsigned long long b) {
  RiMove m;
  m.p = A.parent[x];
  const float4* q = (const float4*)(A.rec + m.p);
  const uint32_t l = __float_as_uint(q[0].w), r = __float_as_uint(q[1].w);
  m.sib = l == x ? r : l;
  m.gp = A.parent[m.p];
  m.out = (uint32_t)b;
  m.po = A.parent[m.out];
  return m;
}

__global__ __launch_bounds__(256) void bb_ri_lock_kernel(RiArgs A) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2u * A.n - 1u) return;
  const unsigned long long b = A.best[x];
  if (A.moves) { const unsigned long long want = __ballot(b != 0ull); if ((threadIdx.x & 63u) == 0u && want) atomicAdd(A.moves + BB_RI_MAX_ITERS + A.it, (uint32_t)__popcll(want)); }
  if (!b) return;
  const RiMove m = ri_move(A, x, b);
  const unsigned long long key = (b & 0xffffffff00000000ull) | x;
  atomicMax(A.lock + x, key); atomicMax(A.lock + m.p, key); atomicMax(A.lock + m.sib, key);
  atomicMax(A.lock + m.gp, key); atomicMax(A.lock + m.out, key); atomicMax(A.lock + m.po, key);
}

__global__ __launch_bounds__(256) void bb_ri_resolve_kernel(RiArgs A, int pass) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2u * A.n - 1u) return;
  const unsigned long long b = A.best[x];
  if (pass == 0) {
    bool w = false;
    if (b) {
      const RiMove m = ri_move(A, x, b);
      const unsigned long long key = (b & 0xffffffff00000000ull) | x;
      w = A.lock[x] == key && A.lock[m.p] == key && A.lock[m.sib] == key && A.lock[m.gp] == key && A.lock[m.out] == key && A.lock[m.po] == key;
    }
    A.win0[x] = w ? 1 : 0;
    return;
  }
  bool w = A.win0[x] != 0;
  if (w) {
    const unsigned long long key = (b & 0xffffffff00000000ull) | x;
    uint32_t up = 0;
    for (uint32_t a = A.parent[(uint32_t)b]; a != BB_NONE; a = A.parent[a]) {
      if (A.win0[a] && ((A.best[a] & 0xffffffff00000000ull) | a) > key) { w = false; break; }
      if (++up > (uint32_t)BB_RI_LISTS) { w = false; break; }     // (deeper than the refit follows: such a tree is reported, not emitted)
    }
  }
  A.win[x] = w ? 1 : 0;
}

__global__ __launch_bounds__(256) void bb_ri_apply_kernel(RiArgs A) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  const bool w = x < 2u * A.n - 1u && A.win[x] != 0;
  if (w) {
    const RiMove m = ri_move(A, x, A.best[x]);
    uint32_t* g = (uint32_t*)(A.rec + m.gp);            // words 3 / 7: left / right
    if (g[3] == m.p) g[3] = m.sib; else g[7] = m.sib;
    A.parent[m.sib] = m.gp;
    uint32_t* o = (uint32_t*)(A.rec + m.po);            // (po may be gp: read after the write above)
    if (o[3] == m.out) o[3] = m.p; else o[7] = m.p;
    A.parent[m.p] = m.po;
    uint32_t* pp = (uint32_t*)(A.rec + m.p);
    pp[3] = m.out; pp[7] = x;
    A.parent[m.out] = m.p;
  }
  const unsigned long long any = __ballot(w);
  if (A.moves && (threadIdx.x & 63u) == 0u && any) atomicAdd(A.moves + A.it, (uint32_t)__popcll(any));   // (VXRT_BVH_VERBOSE only: 25,000 atomics on one word cost 0.15 ms)
}

// Refit, bottom-up by height: list k holds the nodes whose two children are done after pass k-1 (a node is put on a list by the second of
// its children to finish: one atomic counter per node).  Children are computed in an earlier LAUNCH than their parent -- the kernel boundary
// is what makes their boxes visible (device-scope fences inside one walk-up kernel cost 5.8 ms per refit of a million triangles: an L2
// write-back per step).  The first passes are launches of their own; once the lists are short one workgroup finishes all remaining heights
// with workgroup barriers between them.  FULL: counts and the dynamic programme of step 4 as well as the boxes.
__device__ __forceinline__ void ri_append(uint32_t* __restrict__ list, uint32_t* __restrict__ count, bool ready, uint32_t node) {
  const unsigned long long m = __ballot(ready);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63u;
  const int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  if (ready) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = node;
}

__global__ __launch_bounds__(256) void bb_ri_refit_leaves_kernel(RiArgs A, uint32_t* __restrict__ out, uint32_t* __restrict__ counts) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t p = BB_NONE; bool ready = false;
  if (j < A.n) {
    p = A.parent[A.n - 1u + j];
    if (p != BB_NONE) ready = atomicAdd(A.arrived + p, 1u) == 1u;
  }
  ri_append(out, counts, ready, p);
}

template <bool FULL>
__device__ __forceinline__ void ri_refit_node(const RiArgs& A, uint32_t cur) {
  const float4* me = (const float4*)(A.rec + cur);
  const uint32_t l = __float_as_uint(me[0].w), r = __float_as_uint(me[1].w);
  const float4* pl = (const float4*)(A.rec + l);
  const float4* pr = (const float4*)(A.rec + r);
  if (FULL) {
    const float4 a0 = pl[0], a1 = pl[1], a2 = pl[2], a3 = pl[3], b0 = pr[0], b1 = pr[1], b2 = pr[2], b3 = pr[3];
    Cluster a, b;
    a.lx = a0.x; a.ly = a0.y; a.lz = a0.z; a.hx = a1.x; a.hy = a1.y; a.hz = a1.z; a.f1 = a2.x; a.f2 = a2.y; a.f3 = a2.z; a.f4 = a2.w; a.count = __float_as_uint(a3.x); a.id = l;
    b.lx = b0.x; b.ly = b0.y; b.lz = b0.z; b.hx = b1.x; b.hy = b1.y; b.hz = b1.z; b.f1 = b2.x; b.f2 = b2.y; b.f3 = b2.z; b.f4 = b2.w; b.count = __float_as_uint(b3.x); b.id = r;
    uint32_t plan;
    const Cluster c = bb_merge(a, b, cur, A.leaf_max, A.tri_cost, plan);
    bb_store_node(A.rec, c, l, r, plan);
  } else {
    const float4 a0 = pl[0], a1 = pl[1], b0 = pr[0], b1 = pr[1];
    float4* o = (float4*)(A.rec + cur);
    o[0] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), __uint_as_float(l));
    o[1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), __uint_as_float(r));
  }
}

template <bool FULL>
__global__ __launch_bounds__(256) void bb_ri_refit_pass_kernel(RiArgs A, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ counts) {
  const uint32_t m = counts[0];
  for (uint32_t base = blockIdx.x * blockDim.x; base < m; base += gridDim.x * blockDim.x) {   // (whole wavefronts run every iteration: ri_append votes)
    const uint32_t i = base + threadIdx.x;
    uint32_t p = BB_NONE; bool ready = false;
    if (i < m) {
      const uint32_t cur = in[i];
      ri_refit_node<FULL>(A, cur);
      p = A.parent[cur];
      if (p != BB_NONE) ready = atomicAdd(A.arrived + p, 1u) == 1u;
    }
    ri_append(out, counts + 1, ready, p);
  }
}

// all remaining heights in one workgroup; lists alternate between the two buffers; counts[k] = length of list k
template <bool FULL>
__global__ __launch_bounds__(1024) void bb_ri_refit_tail_kernel(RiArgs A, uint32_t* __restrict__ l0, uint32_t* __restrict__ l1, uint32_t* __restrict__ counts, uint32_t first, uint32_t max_lists) {
  uint32_t k = first;
  for (;;) {
    const uint32_t m = counts[k];
    if (m == 0u || k + 1u >= max_lists) break;     // (every thread reads the same value: counts[k] was complete at the last barrier)
    const uint32_t* in = (k & 1u) ? l1 : l0;
    uint32_t* out = (k & 1u) ? l0 : l1;
    for (uint32_t base = 0; base < m; base += blockDim.x) {
      const uint32_t i = base + threadIdx.x;
      uint32_t p = BB_NONE; bool ready = false;
      if (i < m) {
        const uint32_t cur = in[i];
        ri_refit_node<FULL>(A, cur);
        p = A.parent[cur];
        if (p != BB_NONE) ready = atomicAdd(A.arrived + p, 1u) == 1u;
      }
      ri_append(out, counts + k + 1u, ready, p);
    }
    __threadfence_block();
    __syncthreads();
    ++k;
  }
}

// ---- 5. collapse + quantise + emit ----
// smallest e with extent / 255 <= 2^e (bvh.cpp:215-264 picks ceil(log2(extent / 255))), from the float's own exponent: exact
__device__ __forceinline__ int bb_pick_exp(float extent) {
  if (!(extent > 0.0f) || extent > 3.0e38f) return 0;
  int k;
  const float m = frexpf(extent / 255.0f, &k);   // extent / 255 = m * 2^k, m in [0.5, 1)
  int e = m == 0.5f ? k - 1 : k;
  return max(-126, min(126, e));
}

// q_lo, q_hi of one axis of one child at scale s = 2^e (inv = 2^-e, both exact: |e| <= 126); false if the child does not fit 8 bits there
__device__ __forceinline__ bool bb_quant_axis(float origin, float s, float inv, float cmin, float cmax, uint32_t& qlo, uint32_t& qhi) {
  float fl = floorf((cmin - origin) * inv), fh = ceilf((cmax - origin) * inv);
  if (!(fl >= 0.0f)) fl = 0.0f;
  if (!(fh >= fl)) fh = fl;
  if (fh > 255.0f) return false;
  int lo = (int)fl, hi = (int)fh;
  if (lo > 255) return false;
  // conservative after the decode's own rounding (origin + q * 2^e rounds once; q * 2^e itself is exact)
  while (lo > 0 && origin + (float)lo * s > cmin) --lo;
  while (hi < 255 && origin + (float)hi * s < cmax) ++hi;
  if (origin + (float)hi * s < cmax) return false;
  qlo = (uint32_t)lo; qhi = (uint32_t)hi;
  return true;
}
__device__ __forceinline__ float bb_pow2(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }   // e in [-126, 127]

struct CollapseArgs {
  const BNode* rec;
  uint32_t n, tri_offset, node_capacity;
  uint32_t* nodes;          // 13 dwords per node
  uint32_t* counters;       // [0] nodes allocated, [1] leaves, [2] largest leaf, [3] deepest level, [4] error flags, [8 + L] items of level L
  const uint4* in; uint4* out;   // items, one per internal node of the level: (binary node, output slot, first position of its triangles in the final order, -)
  uint32_t level;
  const uint32_t* vals;     // sorted position -> primitive
  uint32_t* order;          // BLAS build: final position -> primitive (the gather's index); nullptr = TLAS build (a leaf names its instance)
  uint32_t* chunk_tot; uint2* chunk_base;   // per chunk of 256 items of the level: (children | internal children << 16); (first node slot, first queue position)
  uint32_t child_order;     // 0: slots as the binary tree hands them out, 1: largest surface area first, 2: smallest first, 3 / 4: by centre along the parent's widest axis, ascending / descending
};

struct BRec { Box3 box; uint32_t left, right, count, plan; };
__device__ __forceinline__ BRec bb_load_rec(const BNode* __restrict__ rec, uint32_t id) {
  const float4* p = (const float4*)(rec + id);
  const float4 a = p[0], b = p[1], c = p[3];
  BRec r;
  r.box.lx = a.x; r.box.ly = a.y; r.box.lz = a.z; r.left = __float_as_uint(a.w);
  r.box.hx = b.x; r.box.hy = b.y; r.box.hz = b.z; r.right = __float_as_uint(b.w);
  r.count = __float_as_uint(c.x); r.plan = __float_as_uint(c.y);
  return r;
}

// a leaf record in slot `out`: the subtree of binary node b (<= 15 triangles), its triangles listed from `start` of the final order
__device__ __forceinline__ void bb_emit_leaf(const CollapseArgs& A, uint32_t b, const BRec& r, uint32_t out, uint32_t start) {
  // (binary leaf j has id n-1+j: a child id says whether it is one, no load needed)
  uint32_t first_prim = 0;
  const uint32_t nl = A.n - 1u;
  if (r.left == 0xffffffffu) {
    first_prim = A.vals[b - nl];
    if (A.order) A.order[start] = first_prim;
  } else if (r.count == 2u) {   // (the usual leaf: two triangles, both children are binary leaves)
    first_prim = A.vals[r.left - nl];
    if (A.order) { A.order[start] = first_prim; A.order[start + 1u] = A.vals[r.right - nl]; }
  } else {
    // left to right: the t-th triangle is found by walking down with the counts
    for (uint32_t t = 0; t < r.count; ++t) {
      uint32_t x = b, rem = t;
      while (x < nl) {
        const uint32_t l = A.rec[x].left;
        const uint32_t cl = l >= nl ? 1u : A.rec[l].count;
        if (rem < cl) x = l; else { rem -= cl; x = A.rec[x].right; }
      }
      const uint32_t prim = A.vals[x - nl];
      if (t == 0) first_prim = prim;
      if (A.order) A.order[start + t] = prim;
    }
  }
  uint32_t* o = A.nodes + (size_t)out * RT_NODE_DWORDS;
  o[0] = __float_as_uint(r.box.lx); o[1] = __float_as_uint(r.box.ly); o[2] = __float_as_uint(r.box.lz);
  const int e0 = bb_pick_exp(r.box.hx - r.box.lx), e1 = bb_pick_exp(r.box.hy - r.box.ly), e2 = bb_pick_exp(r.box.hz - r.box.lz);
  o[3] = (uint32_t)(uint8_t)(int8_t)e0 | ((uint32_t)(uint8_t)(int8_t)e1 << 8) | ((uint32_t)(uint8_t)(int8_t)e2 << 16) | (A.order ? 0u : 1u << 24);   // imask: 1 = TLAS node
  if (!A.order) { o[4] = 0; o[5] = first_prim; }                 // TLAS leaf (bvh.cpp:325-328): leafData = blasIdx
  else { o[4] = start + A.tri_offset; o[5] = r.count; }          // bvh.cpp:260: already offset by the mesh's first triangle
#pragma unroll
  for (int k = 6; k < RT_NODE_DWORDS; ++k) o[k] = 0u;
}

// Top-down, one launch per level, one item per INTERNAL node of the 4-wide tree: the node takes the child slots the dynamic
// programme chose (step 4) -- its two binary children share four slots as plan.a4 says; a child offered j > 1 slots either stays
// one child or hands them on to its own two children (plan.self / plan.a of that child) -- gives every child its range of the final
// triangle order (its own range, cut up in slot order), writes its own record and the records of the children that are leaves
// (two thirds of all nodes: as items of their own they would idle through the internal nodes' work in the same wavefront), and
// queues the others for the next level.
//
// The slots are handed out by PREFIX SUMS in queue order, not by atomics (round 5: with one atomic per workgroup the records' order
// differed from build to build; the tree did not): a level is three launches -- COUNT (this kernel without its writes: the children
// and internal children of every chunk of 256 items), bb_collapse_scan_kernel (one workgroup: the chunks' bases, the level's totals),
// and this kernel again.  Two builds of the same triangles give the same bytes.
template <bool COUNT>
__global__ __launch_bounds__(256) void bb_collapse_kernel(CollapseArgs A) {
  const uint32_t n_items = A.counters[8 + A.level];
  const uint32_t lane = threadIdx.x & 63u;
  // (every lane of a wavefront runs every iteration: node slots and queue positions come from prefix sums over the wavefront's lanes,
  // the workgroup's wavefronts and -- through the scan kernel -- the level's chunks)
  for (uint32_t base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {
    const uint32_t it = base + threadIdx.x;
    bool act = it < n_items;
    const uint4 item = act ? A.in[it] : make_uint4(A.n - 1, 0u, 0u, 0u);
    const uint32_t b = item.x, out = item.y, start = item.z;
    const BRec me = bb_load_rec(A.rec, b);
    const Box3 bx = me.box;
    if (act && (me.plan & PLAN_LEAF) != 0u) {   // only the root can arrive here as a leaf (a mesh of a few triangles)
      if (!COUNT) {
        bb_emit_leaf(A, b, me, out, start);
        atomicAdd(A.counters + 1, 1u); atomicMax(A.counters + 2, me.count);
      }
      act = false;
    }
    // (all arrays below are indexed with compile-time constants only -- unrolled loops, selects on k == pick -- so that they
    // live in registers: with dynamic indices they went to scratch and a level took as long as ~250 dependent scratch accesses)
    uint32_t c[4] = {0, 0, 0, 0}, slots[4] = {0, 0, 0, 0};
    uint32_t nc = 0;
    BRec cr[4];
    cr[0] = cr[1] = cr[2] = cr[3] = me;
    if (act) {
      c[0] = me.left; c[1] = me.right;
      slots[0] = plan_a(me.plan, 4u); slots[1] = 4u - slots[0];
      nc = 2;
      cr[0] = bb_load_rec(A.rec, c[0]); cr[1] = bb_load_rec(A.rec, c[1]);
      // the two children's own splits are known now: their four children are fetched together (one latency, not two)
      const bool sp0 = slots[0] > 1u && cr[0].left != 0xffffffffu && !plan_self(cr[0].plan, slots[0]);
      const bool sp1 = slots[1] > 1u && cr[1].left != 0xffffffffu && !plan_self(cr[1].plan, slots[1]);
      BRec g0l = me, g0r = me, g1l = me, g1r = me;
      if (sp0) { g0l = bb_load_rec(A.rec, cr[0].left); g0r = bb_load_rec(A.rec, cr[0].right); }
      if (sp1) { g1l = bb_load_rec(A.rec, cr[1].left); g1r = bb_load_rec(A.rec, cr[1].right); }
      if (sp0) {
        const uint32_t j = slots[0], ja = plan_a(cr[0].plan, j), gl = cr[0].left, gr = cr[0].right;
        c[0] = gl; cr[0] = g0l; slots[0] = ja;
        c[2] = gr; cr[2] = g0r; slots[2] = j - ja;
        nc = 3;
      }
      if (sp1) {
        const uint32_t j = slots[1], ja = plan_a(cr[1].plan, j), gl = cr[1].left, gr = cr[1].right;
        c[1] = gl; cr[1] = g1l; slots[1] = ja;
#pragma unroll
        for (int k = 2; k < 4; ++k) if ((uint32_t)k == nc) { c[k] = gr; cr[k] = g1r; slots[k] = j - ja; }
        ++nc;
      }
      // at most one more split: a grandchild that was handed two of three slots
      if (nc == 3u) {
        int pick = -1;
#pragma unroll
        for (int k = 2; k >= 0; --k)
          if (slots[k] > 1u && cr[k].left != 0xffffffffu && !plan_self(cr[k].plan, slots[k])) pick = k;
        if (pick >= 0) {
          uint32_t gl = 0, gr = 0, j = 0, pl = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) if (k == pick) { gl = cr[k].left; gr = cr[k].right; j = slots[k]; pl = cr[k].plan; }
          const uint32_t ja = plan_a(pl, j);
          const BRec L = bb_load_rec(A.rec, gl), R = bb_load_rec(A.rec, gr);
#pragma unroll
          for (int k = 0; k < 3; ++k) if (k == pick) { c[k] = gl; cr[k] = L; slots[k] = ja; }
          c[3] = gr; cr[3] = R; slots[3] = j - ja;
          nc = 4;
        }
      }
    }
    // slot order = the order in which the frame's occlusion rays visit the children (any-hit, rt_kernels.hip: slot order, no sorting by
    // distance); closest-hit rays sort by distance, for them the slot order only decides ties.  Default (3): by the child's centre along
    // this node's widest axis, ascending -- the order a top-down builder's "left = below the plane" produces and csrc/scene_builder.cpp
    // emits; clustering and reinsertion leave the two children of a binary node in no particular order.  Measured at three light
    // positions against the order as built (profiles/r05_g_gpu_reinsertion.txt, section 8): occlusion rays' node fetches -7 % / +2 % / -17 %,
    // the headline frame on the GPU-built tree 8.4 -> 9.0 Grays/s.  1 / 2: largest / smallest surface area first (Nah & Manocha's SATO
    // and its opposite), 4: descending centre -- each wins at one light and loses at another; kept for measurements (VXRT_BVH_CHILD_ORDER).
    if (!COUNT && A.child_order != 0u) {
      float key[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ar = box_area(cr[k].box.lx, cr[k].box.ly, cr[k].box.lz, cr[k].box.hx, cr[k].box.hy, cr[k].box.hz);
        // 3 / 4: by the child's centre along the parent's widest axis, ascending / descending (what a top-down builder's "left = below the plane" gives)
        const float ex = bx.hx - bx.lx, ey = bx.hy - bx.ly, ez = bx.hz - bx.lz;
        const float cen = ex >= ey && ex >= ez ? cr[k].box.lx + cr[k].box.hx : (ey >= ez ? cr[k].box.ly + cr[k].box.hy : cr[k].box.lz + cr[k].box.hz);
        const float kk = A.child_order == 1u ? ar : (A.child_order == 2u ? -ar : (A.child_order == 3u ? -cen : cen));
        key[k] = (uint32_t)k < nc ? kk : -__builtin_inff();
      }
#define BB_CSWAP(i, j) do { if (key[j] > key[i]) { const float tk = key[i]; key[i] = key[j]; key[j] = tk; const uint32_t tc = c[i]; c[i] = c[j]; c[j] = tc; \
        const uint32_t ts = slots[i]; slots[i] = slots[j]; slots[j] = ts; const BRec tr = cr[i]; cr[i] = cr[j]; cr[j] = tr; } } while (0)
      BB_CSWAP(0, 1); BB_CSWAP(2, 3); BB_CSWAP(0, 2); BB_CSWAP(1, 3); BB_CSWAP(1, 2);
#undef BB_CSWAP
    }
    uint32_t ni = 0, nl = 0, lmax = 0;   // children that go on: internal ones; leaves among them and their largest
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((uint32_t)k < nc) { if (cr[k].plan & PLAN_LEAF) { ++nl; lmax = max(lmax, cr[k].count); } else ++ni; }
    // wavefront prefix sums: node slots over all children, next-level queue positions over the internal ones
    uint32_t incl = nc | (ni << 16);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += t; }
    const uint32_t total = __shfl(incl, 63);
    // ... the workgroup's chunk of 256 items: its totals out (COUNT), its bases in (from the scan over the chunks)
    __shared__ uint32_t s_tot[4], s_base[2];
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 0) s_tot[wv] = total;
    __syncthreads();
    if (COUNT) {
      if (threadIdx.x == 0) A.chunk_tot[base >> 8] = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];   // (children | internal children << 16: at most 1,024 each)
      __syncthreads();
      continue;
    }
    if (threadIdx.x == 0) { const uint2 cb = A.chunk_base[base >> 8]; s_base[0] = cb.x; s_base[1] = cb.y; }
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < wv; ++k) woff += s_tot[k];
    const uint32_t excl = woff + incl - (nc | (ni << 16));
    const uint32_t first = s_base[0] + (excl & 0xffffu), pos = s_base[1] + (excl >> 16);
    __syncthreads();   // (s_tot / s_base are rewritten by the next iteration)
    uint32_t nls = nl, lm = lmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nls += __shfl_down(nls, off); lm = max(lm, (uint32_t)__shfl_down(lm, off)); }
    if (lane == 0 && nls != 0u) {
      atomicAdd(A.counters + 1, nls);
      atomicMax(A.counters + 2, lm);
      atomicMax(A.counters + 3, A.level + 1u);
    }
    if (!act) continue;
    if (first + nc > A.node_capacity) { atomicOr(A.counters + 4, 1u); continue; }
    uint32_t w[13];
    w[0] = __float_as_uint(bx.lx); w[1] = __float_as_uint(bx.ly); w[2] = __float_as_uint(bx.lz);
    int e[3] = {bb_pick_exp(bx.hx - bx.lx), bb_pick_exp(bx.hy - bx.ly), bb_pick_exp(bx.hz - bx.lz)};
    uint32_t ql[4][3] = {}, qh[4][3] = {};
    const float org[3] = {bx.lx, bx.ly, bx.lz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      for (;;) {
        bool ok = true;
        const float sc = bb_pow2(e[a]), inv = bb_pow2(-e[a]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((uint32_t)k < nc && ok) {
            const float cmin = a == 0 ? cr[k].box.lx : (a == 1 ? cr[k].box.ly : cr[k].box.lz);
            const float cmax = a == 0 ? cr[k].box.hx : (a == 1 ? cr[k].box.hy : cr[k].box.hz);
            ok = bb_quant_axis(org[a], sc, inv, cmin, cmax, ql[k][a], qh[k][a]);
          }
        }
        if (ok) break;
        if (e[a] >= 126) { atomicOr(A.counters + 4, 2u); break; }
        ++e[a];
      }
    }
    w[4] = first;     // relative to this BLAS's first node (rt_traversal.cpp:92,119); TLAS: to its node 0
    w[5] = A.order ? 0u : 0xffffffffu;   // internal TLAS nodes carry UINT32_MAX (bvh.cpp:417)
    w[3] = (uint32_t)(uint8_t)(int8_t)e[0] | ((uint32_t)(uint8_t)(int8_t)e[1] << 8) | ((uint32_t)(uint8_t)(int8_t)e[2] << 16) | (A.order ? 0u : 1u << 24);   // imask: 1 = TLAS node
    // children: 4 x { meta, lo x y z, hi x y z } = 28 bytes from dword 6 on
    uint64_t cbits[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool on = (uint32_t)k < nc;
      cbits[k] = on ? (1ull | ((uint64_t)ql[k][0] << 8) | ((uint64_t)ql[k][1] << 16) | ((uint64_t)ql[k][2] << 24) |
                       ((uint64_t)qh[k][0] << 32) | ((uint64_t)qh[k][1] << 40) | ((uint64_t)qh[k][2] << 48)) : 0ull;   // 7 bytes
    }
    // pack the four 7-byte groups back to back
    const unsigned __int128 lo128 = (unsigned __int128)cbits[0] | ((unsigned __int128)cbits[1] << 56) | ((unsigned __int128)cbits[2] << 112);
    const unsigned __int128 hi128 = ((unsigned __int128)cbits[2] >> 16) | ((unsigned __int128)cbits[3] << 40);
    w[6] = (uint32_t)lo128; w[7] = (uint32_t)(lo128 >> 32); w[8] = (uint32_t)(lo128 >> 64); w[9] = (uint32_t)(lo128 >> 96);
    w[10] = (uint32_t)hi128; w[11] = (uint32_t)(hi128 >> 32); w[12] = (uint32_t)(hi128 >> 64);
    uint32_t* o = A.nodes + (size_t)out * RT_NODE_DWORDS;
#pragma unroll
    for (int k = 0; k < RT_NODE_DWORDS; ++k) o[k] = w[k];
    // the children: leaves are written now, the others queued
    uint32_t cstart = start, qpos = pos;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if ((uint32_t)k < nc) {
        if (cr[k].plan & PLAN_LEAF) bb_emit_leaf(A, c[k], cr[k], first + k, cstart);
        else A.out[qpos++] = make_uint4(c[k], first + k, cstart, 0u);
        cstart += cr[k].count;
      }
    }
  }
}

// the chunks' bases of a level: node slots from the nodes allocated so far, queue positions from 0; the level's totals into the counters
__global__ __launch_bounds__(1024) void bb_collapse_scan_kernel(CollapseArgs A) {
  const uint32_t n_items = A.counters[8 + A.level];
  const uint32_t n_chunks = (n_items + 255u) >> 8;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  __shared__ uint32_t s_w[16][2], s_carry[2];
  if (threadIdx.x == 0) { s_carry[0] = A.counters[0]; s_carry[1] = 0u; }
  __syncthreads();
  for (uint32_t t0 = 0; t0 < n_chunks; t0 += 1024u) {
    const uint32_t c = t0 + threadIdx.x;
    const uint32_t v = c < n_chunks ? A.chunk_tot[c] : 0u;
    const uint32_t a = v & 0xffffu, b = v >> 16;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t ta = __shfl_up(ia, off), tb = __shfl_up(ib, off);
      if (lane >= (uint32_t)off) { ia += ta; ib += tb; }
    }
    if (lane == 63u) { s_w[wv][0] = ia; s_w[wv][1] = ib; }
    __syncthreads();
    uint32_t wa = 0, wb = 0, ta = 0, tb = 0;
    for (uint32_t k = 0; k < 16u; ++k) { if (k < wv) { wa += s_w[k][0]; wb += s_w[k][1]; } ta += s_w[k][0]; tb += s_w[k][1]; }
    const uint32_t ca = s_carry[0], cb = s_carry[1];
    if (c < n_chunks) A.chunk_base[c] = make_uint2(ca + wa + ia - a, cb + wb + ib - b);
    __syncthreads();
    if (threadIdx.x == 0) { s_carry[0] = ca + ta; s_carry[1] = cb + tb; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { A.counters[0] = s_carry[0]; A.counters[8 + A.level + 1] = s_carry[1]; }
}

// ---- 6. gather into the final order ----
__global__ __launch_bounds__(256) void bb_gather_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ vals, uint32_t n, uint32_t dwords,
                                                          uint32_t* __restrict__ dst) {
  const uint64_t total = (uint64_t)n * dwords;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t j = (uint32_t)(t / dwords), k = (uint32_t)(t % dwords);
    dst[t] = src[(size_t)vals[j] * dwords + k];
  }
}

// Scratch: one grow-only device allocation per device, kept between builds (a scene rebuilt every frame must not pay a dozen
// hipMalloc / hipFree pairs per build); vxrt_bvh_release_scratch() returns it.
struct Arena {
  void* base = nullptr; size_t cap = 0; size_t used = 0; int dev = -1;
  uint32_t* pinned = nullptr;   // 256 host words for the read-backs (cluster count between chunks of rounds, counters at the end)
  template <class T> T* get(size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (used + bytes > cap) return nullptr;
    T* p = (T*)((char*)base + used);
    used += bytes;
    return p;
  }
};
std::mutex g_arena_mu;
Arena g_arena;

bool arena_reserve(size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (g_arena.base && (g_arena.dev != dev || g_arena.cap < bytes)) { (void)hipFree(g_arena.base); g_arena.base = nullptr; g_arena.cap = 0; }
  if (!g_arena.pinned && hipHostMalloc((void**)&g_arena.pinned, 256 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { g_arena.pinned = nullptr; return false; }
  if (!g_arena.base) {
    if (hipMalloc(&g_arena.base, bytes) != hipSuccess) { g_arena.base = nullptr; g_arena.cap = 0; return false; }
    g_arena.cap = bytes; g_arena.dev = dev;
  }
  g_arena.used = 0;
  return true;
}

}  // namespace

static int build_common(void* d_tri, void* d_triEx, uint32_t n_tris, uint32_t tri_offset, uint32_t leaf_max,
                        void* d_nodes, uint32_t node_capacity, vxrt_bvh_info_t* info, void* stream, bool boxes) {
  if (!d_tri || !d_nodes || n_tris == 0 || n_tris > 0x0fffffffu) return -1;
  if (leaf_max == 0) leaf_max = 2;
  if (leaf_max > 15) leaf_max = 15;
  if (boxes) { leaf_max = 1; d_triEx = nullptr; }
  if ((uint64_t)node_capacity < 2ull * n_tris - 1ull) return -1;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n = n_tris;
  std::lock_guard<std::mutex> lk(g_arena_mu);   // (builds on one device are serialised on the scratch arena)
  const uint32_t n_counters = 8 + BB_MAX_LEVELS + 2 + 2 * BB_RI_MAX_ITERS;   // (... then, VXRT_BVH_VERBOSE: the moves of each reinsertion iteration and the nodes that wanted one)
  size_t tmp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 63u, s) != hipSuccess) return -1;
  const uint32_t blocks = (n + 255u) / 256u;
  const size_t per_tri = 2 * 8 + 2 * 4 + 2 * sizeof(BNode) + 2 * sizeof(Cluster) + 1 + 2 * 16 + 4 + (d_triEx ? 64 : 36) + (2 * 4 + 2 * 8 + 2 * 8 + 4 + 4);
  if (!arena_reserve((size_t)n * per_tri + (size_t)blocks * 16 + tmp_bytes + 64 * 1024)) return -1;
  Arena& sc = g_arena;
  int* cb = sc.get<int>(8);
  uint32_t* counters = sc.get<uint32_t>(n_counters);
  uint32_t* st = sc.get<uint32_t>(ST_WORDS);
  uint32_t* seq = sc.get<uint32_t>(64);
  uint64_t* keys0 = sc.get<uint64_t>(n);
  uint64_t* keys1 = sc.get<uint64_t>(n);
  uint32_t* vals0 = sc.get<uint32_t>(n);
  uint32_t* vals1 = sc.get<uint32_t>(n);
  BNode* rec = sc.get<BNode>(2 * (size_t)n);
  Cluster* cl0 = sc.get<Cluster>(n);
  Cluster* cl1 = sc.get<Cluster>(n);
  uint8_t* dec = sc.get<uint8_t>(n);
  uint2* tile_counts = sc.get<uint2>(blocks);
  uint2* tile_base = sc.get<uint2>(blocks);
  uint4* q0 = sc.get<uint4>(n);
  uint4* q1 = sc.get<uint4>(n);
  uint32_t* order = sc.get<uint32_t>(n);
  uint32_t* gather = sc.get<uint32_t>((size_t)n * (d_triEx ? 16 : 9));
  uint32_t* ri_parent = sc.get<uint32_t>(2 * (size_t)n);
  unsigned long long* ri_lock = sc.get<unsigned long long>(2 * (size_t)n);
  unsigned long long* ri_best = sc.get<unsigned long long>(2 * (size_t)n);
  uint8_t* ri_win = sc.get<uint8_t>(4 * (size_t)n);
  uint32_t* ri_arrived = sc.get<uint32_t>((size_t)n + BB_RI_LISTS);   // refit: one counter per internal node, then the lengths of the lists
  void* tmp = sc.get<uint8_t>(tmp_bytes ? tmp_bytes : 16);
  if (!cb || !counters || !st || !seq || !keys0 || !keys1 || !vals0 || !vals1 || !rec || !cl0 || !cl1 || !dec || !tile_counts || !tile_base || !q0 || !q1 ||
      !order || !gather || !tmp || !sc.pinned || !ri_parent || !ri_lock || !ri_best || !ri_win || !ri_arrived) return -1;
  const uint32_t wide = blocks < 4096u ? blocks : 4096u;
  // (VXRT_BVH_TRI_COST: the triangle's cost against the node record's 52, for measurements; the default is the format's byte ratio)
  static const float tri_cost_env = [] { const char* e = getenv("VXRT_BVH_TRI_COST"); return e ? (float)atof(e) : BB_TRI_COST; }();
  const float tri_cost = boxes ? 0.0f : tri_cost_env;

  hipLaunchKernelGGL(bb_init_kernel, dim3(1), dim3(256), 0, s, cb, counters, n_counters, q0);
  hipLaunchKernelGGL(bb_bounds_kernel, dim3(wide < 512u ? wide : 512u), dim3(256), 0, s, (const float*)d_tri, n, cb, boxes);
  hipLaunchKernelGGL(bb_axis_sequence_kernel, dim3(1), dim3(64), 0, s, (const int*)cb, seq);
  hipLaunchKernelGGL(bb_morton_kernel, dim3(blocks), dim3(256), 0, s, (const float*)d_tri, n, (const int*)cb, (const uint32_t*)seq, keys0, vals0, boxes);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys0, keys1, vals0, vals1, (size_t)n, 0u, 63u, s) != hipSuccess) return -1;

  // clustering: rounds of (decide, scan, apply) over the whole array while it is long, eight at a time between looks at the
  // cluster count (the count only falls, so the last one read bounds the grid); the single-workgroup tail takes the rest
  hipLaunchKernelGGL(bb_leaves_kernel, dim3(blocks), dim3(256), 0, s, (const float*)d_tri, vals1, n, boxes, tri_cost, cl0, rec, st);
  Cluster* cur = cl0; Cluster* nxt = cl1;
  uint32_t m_upper = n;
  for (int chunk = 0; m_upper > (uint32_t)BB_TAIL; ++chunk) {
    if (chunk == 64) return -1;   // (512 rounds: every second round at least halves the array or merges a sixteenth of it)
    const uint32_t tiles = (m_upper + BB_TILE - 1) / BB_TILE, g = tiles < 2048u ? tiles : 2048u;
    for (int r = 0; r < 8; ++r) {
      hipLaunchKernelGGL(bb_ploc_decide_kernel, dim3(g), dim3(256), 0, s, (const Cluster*)cur, (const uint32_t*)st, dec, tile_counts);
      hipLaunchKernelGGL(bb_ploc_scan_kernel, dim3(1), dim3(1024), 0, s, (const uint2*)tile_counts, tile_base, st);
      hipLaunchKernelGGL(bb_ploc_apply_kernel, dim3(g), dim3(256), 0, s, (const Cluster*)cur, nxt, (const uint8_t*)dec, (const uint2*)tile_base,
                         (const uint32_t*)st, rec, leaf_max, tri_cost);
      Cluster* t = cur; cur = nxt; nxt = t;
    }
    if (hipMemcpyAsync(sc.pinned, st, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    if (sc.pinned[0] > m_upper || sc.pinned[0] == 0u) return -1;
    m_upper = sc.pinned[0];
  }
  hipLaunchKernelGGL(bb_ploc_tail_kernel, dim3(1), dim3(1024), 0, s, (const Cluster*)cur, st, rec, leaf_max, tri_cost);

  // the binary tree optimised by reinsertion (step 4b).  VXRT_BVH_REINSERT = "iterations[:mod]" (0 = the PLOC tree as it is)
  static const int ri_iters = [] { const char* e = getenv("VXRT_BVH_REINSERT"); return e ? atoi(e) : BB_RI_ITERS; }();
  static const int ri_mod = [] { const char* e = getenv("VXRT_BVH_REINSERT"); const char* c = e ? strchr(e, ':') : nullptr; return c ? atoi(c + 1) : 1; }();
  bool ri_ran = false;
  if (!boxes && ri_iters > 0 && n >= 16u) {
    RiArgs R;
    R.rec = rec; R.parent = ri_parent; R.lock = ri_lock; R.best = ri_best; R.win0 = ri_win; R.win = ri_win + 2 * (size_t)n; R.arrived = ri_arrived;
    R.moves = getenv("VXRT_BVH_VERBOSE") ? counters + 8 + BB_MAX_LEVELS + 2 : nullptr; R.n = n; R.it = 0; R.mod = (uint32_t)(ri_mod > 1 ? ri_mod : 1); R.leaf_max = leaf_max; R.tri_cost = tri_cost;
    const uint32_t nb2 = (2u * n + 255u) / 256u;
    uint32_t* wl0 = (uint32_t*)cl0; uint32_t* wl1 = (uint32_t*)cl1;   // (the cluster arrays are free after the clustering)
    uint32_t ri_wide = 2;                                             // passes as launches of their own: until a list is at most ~2,048 nodes long in a balanced tree
    while (ri_wide < 24u && (n >> ri_wide) > 2048u) ++ri_wide;
    hipLaunchKernelGGL(bb_ri_parents_kernel, dim3(blocks), dim3(256), 0, s, R);
    const int iters = ri_iters < BB_RI_MAX_ITERS ? ri_iters : BB_RI_MAX_ITERS;
    for (int it = 0; it < iters; ++it) {
      R.it = (uint32_t)it;
      hipLaunchKernelGGL(bb_ri_search_kernel, dim3(nb2), dim3(256), 0, s, R);
      hipLaunchKernelGGL(bb_ri_lock_kernel, dim3(nb2), dim3(256), 0, s, R);
      hipLaunchKernelGGL(bb_ri_resolve_kernel, dim3(nb2), dim3(256), 0, s, R, 0);
      hipLaunchKernelGGL(bb_ri_resolve_kernel, dim3(nb2), dim3(256), 0, s, R, 1);
      hipLaunchKernelGGL(bb_ri_apply_kernel, dim3(nb2), dim3(256), 0, s, R);
      if (hipMemsetAsync(ri_arrived, 0, ((size_t)n + BB_RI_LISTS) * 4, s) != hipSuccess) return -1;
      const bool full = it + 1 == iters;
      uint32_t* counts = ri_arrived + n;             // counts[k]: nodes on list k (height k + 1)
      hipLaunchKernelGGL(bb_ri_refit_leaves_kernel, dim3(blocks), dim3(256), 0, s, R, wl0, counts);
      uint32_t k = 0;
      for (; k < ri_wide; ++k) {                     // list k holds at most n / (k + 2) nodes
        const uint32_t cap = n / (k + 2u) + 1u, g = (cap + 255u) / 256u < 2048u ? (cap + 255u) / 256u : 2048u;
        const uint32_t* in = (k & 1u) ? wl1 : wl0; uint32_t* out = (k & 1u) ? wl0 : wl1;
        if (full) hipLaunchKernelGGL(bb_ri_refit_pass_kernel<true>, dim3(g), dim3(256), 0, s, R, in, out, counts + k);
        else hipLaunchKernelGGL(bb_ri_refit_pass_kernel<false>, dim3(g), dim3(256), 0, s, R, in, out, counts + k);
      }
      if (full) hipLaunchKernelGGL(bb_ri_refit_tail_kernel<true>, dim3(1), dim3(1024), 0, s, R, wl0, wl1, counts, k, (uint32_t)BB_RI_LISTS);
      else hipLaunchKernelGGL(bb_ri_refit_tail_kernel<false>, dim3(1), dim3(1024), 0, s, R, wl0, wl1, counts, k, (uint32_t)BB_RI_LISTS);
    }
    ri_ran = true;
  }

  // collapse, level by level; sixteen levels, then as many more as the item counts ask for
  CollapseArgs A;
  A.rec = rec; A.n = n; A.tri_offset = tri_offset; A.node_capacity = node_capacity;
  A.nodes = (uint32_t*)d_nodes; A.counters = counters; A.vals = vals1; A.order = boxes ? nullptr : order;
  static const uint32_t child_order_env = [] { const char* e = getenv("VXRT_BVH_CHILD_ORDER"); return e ? (uint32_t)atoi(e) : (uint32_t)BB_CHILD_ORDER; }();
  A.child_order = child_order_env;
  A.chunk_tot = (uint32_t*)tile_counts; A.chunk_base = tile_base;   // (the clustering's per-tile arrays are free now: one entry per 256 items, at most n items per level)
  const uint32_t cwide = blocks < 1024u ? blocks : 1024u;
  uint32_t L = 0;
  for (;;) {
    const uint32_t stop = L + 16u < (uint32_t)BB_MAX_LEVELS ? L + 16u : (uint32_t)BB_MAX_LEVELS;
    for (; L < stop; ++L) {
      A.in = (L & 1u) ? q1 : q0; A.out = (L & 1u) ? q0 : q1; A.level = L;
      // (level L holds at most 4^L items)
      uint32_t g = cwide;
      if (L < 8) { const uint32_t items = 1u << (2 * L); g = (items + 255u) / 256u < cwide ? (items + 255u) / 256u : cwide; }
      hipLaunchKernelGGL(bb_collapse_kernel<true>, dim3(g), dim3(256), 0, s, A);
      hipLaunchKernelGGL(bb_collapse_scan_kernel, dim3(1), dim3(1024), 0, s, A);
      hipLaunchKernelGGL(bb_collapse_kernel<false>, dim3(g), dim3(256), 0, s, A);
    }
    if (L >= (uint32_t)BB_MAX_LEVELS) break;
    if (hipMemcpyAsync(sc.pinned, counters + 8 + L, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
    if (sc.pinned[0] == 0u) break;
  }
  // triangles (and shading records) into the final order, in place through a scratch copy (bvh.cpp:126-128 reorders in place);
  // instances stay where they are (a TLAS leaf names its instance)
  if (!boxes) {
    hipLaunchKernelGGL(bb_gather_kernel, dim3(wide), dim3(256), 0, s, (const uint32_t*)d_tri, (const uint32_t*)order, n, 9u, gather);
    if (hipMemcpyAsync(d_tri, gather, (size_t)n * 36, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
  }
  if (d_triEx) {
    hipLaunchKernelGGL(bb_gather_kernel, dim3(wide), dim3(256), 0, s, (const uint32_t*)d_triEx, (const uint32_t*)order, n, 16u, gather);
    if (hipMemcpyAsync(d_triEx, gather, (size_t)n * 64, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
  }
  uint32_t* hc = sc.pinned;   // counters, then the clustering state, then the root's record
  if (hipMemcpyAsync(hc, counters, n_counters * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipMemcpyAsync(hc + 192, st, ST_WORDS * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipMemcpyAsync(hc + 208, rec, sizeof(BNode), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;   // record 0: the root (or the only leaf)
  hc[191] = 2u;
  if (ri_ran && hipMemcpyAsync(hc + 191, ri_arrived, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;   // the root's counter of the last refit
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  if (hipGetLastError() != hipSuccess) return -1;
  BNode hroot;
  memcpy(&hroot, hc + 208, sizeof hroot);
  static const bool verbose = getenv("VXRT_BVH_VERBOSE") != nullptr;
  if (verbose && !boxes) {
    fprintf(stderr, "[bvh_builder] n %u, reinsertion moves per iteration (made / wanted):", n);
    for (int it = 0; it < BB_RI_MAX_ITERS && it < ri_iters; ++it) fprintf(stderr, " %u/%u", hc[8 + BB_MAX_LEVELS + 2 + it], hc[8 + BB_MAX_LEVELS + 2 + BB_RI_MAX_ITERS + it]);
    fprintf(stderr, "; nodes %u, depth %u\n", hc[0], hc[3]);
  }
  if (info) {
    info->n_nodes = hc[0]; info->n_leaves = hc[1]; info->max_leaf = hc[2]; info->max_depth = hc[3];
    info->bounds[0] = hroot.lx; info->bounds[1] = hroot.ly; info->bounds[2] = hroot.lz;
    info->bounds[3] = hroot.hx; info->bounds[4] = hroot.hy; info->bounds[5] = hroot.hz;
  }
  if (hc[191] != 2u) return -2;                                 // the last refit did not reach the root: a binary tree of more than BB_RI_LISTS levels
  if (hc[192 + ST_N] != 1u || hc[192 + ST_NEXT] != 0u) return -1;   // the clustering did not end in one root with every id used
  if (hc[4] != 0u) return -1;                                  // capacity or exponent range exhausted
  if (hc[8 + BB_MAX_LEVELS] != 0u) return -2;                  // deeper than the collapse pass goes
  if (hc[3] >= (uint32_t)RT_MAX_LEVELS) return -2;             // deeper than the reference's trail (rt_traversal.h:8): use the SAH builder
  return 0;
}

extern "C" int vxrt_bvh_build(void* d_tri, void* d_triEx, uint32_t n_tris, uint32_t tri_offset, uint32_t leaf_max,
                              void* d_nodes, uint32_t node_capacity, vxrt_bvh_info_t* info, void* stream) {
  return build_common(d_tri, d_triEx, n_tris, tri_offset, leaf_max, d_nodes, node_capacity, info, stream, false);
}

// TLAS over instances (reference: BVH::buildTLAS, bvh.cpp:266-421, host code there): the same pipeline over the instances'
// world-space boxes, one instance per leaf (leafData = blasIdx, imask = 1), internal nodes marked UINT32_MAX.
extern "C" int vxrt_tlas_build(const float* d_instance_boxes, uint32_t n_instances, void* d_nodes, uint32_t node_capacity,
                               vxrt_bvh_info_t* info, void* stream) {
  return build_common((void*)d_instance_boxes, nullptr, n_instances, 0, 1, d_nodes, node_capacity, info, stream, true);
}

extern "C" void vxrt_bvh_release_scratch(void) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  if (g_arena.base) (void)hipFree(g_arena.base);
  if (g_arena.pinned) (void)hipHostFree(g_arena.pinned);
  g_arena = Arena();
}
