// BLAS construction on the GPU, in the reference's node format.
//
// Reference: tests/regression/raytracing/bvh.cpp:30-264 -- BVH::build (binned SAH, binary), the collapse to 4-wide nodes and
// the quantiser, all host code run once per mesh at scene load; triangles are reordered in place so that a leaf is a range
// (bvh.cpp:126-128).  csrc/scene_builder.cpp is this package's CPU counterpart (threaded SAH, the quality builder).  This file
// is the builder for geometry that changes per frame: the whole build is a dozen launches over data that never leaves HBM.
//
//   1. centroid bounds              one pass, wavefront reduction + 6 atomics per workgroup
//   2. 63-bit Morton keys           21 bits per axis of the triangle's box centre
//   3. radix sort (key, index)      rocPRIM device sort, 8 passes over 12 bytes per triangle
//   4. binary radix tree            Karras 2012: every internal node finds its range and split independently (no recursion,
//                                   no dependence between nodes); equal keys are told apart by their index
//   5. boxes bottom-up              one thread per triangle climbs; the second thread to reach a node owns it
//   6. collapse to 4-wide + quantise + emit, level by level: a node adopts its binary children and then, twice, replaces the
//      adopted subtree of largest surface area by that subtree's two children; subtrees of <= leaf_max triangles become
//      leaves (a subtree of the radix tree is a contiguous range of the sorted order).  Children are allocated after their
//      parent, which is what vxrt_accel_build's validation asks of any tree.
//   7. triangles (and their shading records) gathered into the sorted order.
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
#include <stdio.h>
#include <mutex>
#include <vector>
#include "rt_types.h"
#include "../../include/vortex_hip.h"

#ifndef BB_EXTENDED_MORTON
#define BB_EXTENDED_MORTON 1
#endif

namespace {

constexpr int BB_MAX_LEVELS = 34;   // launches of the collapse pass; a tree deeper than RT_MAX_LEVELS is reported, not emitted half-way

struct Box3 { float lx, ly, lz, hx, hy, hz; };

// One 48-byte record per node of the binary radix tree (internal i in [0, n-1), leaf j as n-1+j): everything the later passes
// read about a node in three aligned 16-byte loads of ONE cache line (separate child / range / box arrays cost the collapse pass
// some fifteen scattered lines per item).
struct __attribute__((aligned(16))) BNode {
  float lx, ly, lz; uint32_t left;
  float hx, hy, hz; uint32_t right;
  uint32_t first, last, pad0, pad1;
};
static_assert(sizeof(BNode) == 48, "record size");

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

__global__ void bb_init_kernel(int* cb, uint32_t* counters, uint32_t n_counters, uint2* level0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3) cb[i] = 0x7fffffff;
  else if (i < 6) cb[i] = (int)0x80000000;
  if (i < n_counters) counters[i] = (i == 0 || i == 8) ? 1u : 0u;
  if (i == 0) level0[0] = make_uint2(0u, 0u);   // (node id 0 is the binary root, or the only leaf when n == 1)
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

__global__ __launch_bounds__(256) void bb_morton_kernel(const float* __restrict__ tri, uint32_t n, const int* __restrict__ cb,
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
  // whatever its extent).  The axis sequence depends on the bounds only: the same for every thread.
  float e3[3] = {ord2f(cb[3]) - ord2f(cb[0]), ord2f(cb[4]) - ord2f(cb[1]), ord2f(cb[5]) - ord2f(cb[2])};
  int used[3] = {0, 0, 0};
  uint64_t key = 0;
  for (int b = 0; b < 63; ++b) {
    int a = -1; float best = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (used[k] < 21 && e3[k] > best) { best = e3[k]; a = k; }
    const uint32_t qa = a == 0 ? q[0] : (a == 1 ? q[1] : q[2]);
    const int ua = a == 0 ? used[0] : (a == 1 ? used[1] : used[2]);
    key = (key << 1) | ((qa >> (20 - ua)) & 1u);
#pragma unroll
    for (int k = 0; k < 3; ++k) if (k == a) { used[k]++; e3[k] *= 0.5f; }
  }
  keys[i] = key;
#else
  keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
#endif
  vals[i] = i;
}

// ---- 4. binary radix tree (Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", 2012) ----
// node ids: internal i in [0, n-1), leaf j as (n-1) + j; internal 0 is the root
__device__ __forceinline__ int bb_delta(const uint64_t* __restrict__ k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = k[i], b = k[j];
  return a == b ? 64 + __clz(i ^ j) : __clzll((long long)(a ^ b));
}

__global__ __launch_bounds__(256) void bb_tree_kernel(const uint64_t* __restrict__ keys, int n, BNode* __restrict__ rec,
                                                        uint32_t* __restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = bb_delta(keys, n, i, i + 1) - bb_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = bb_delta(keys, n, i, i - d);
  int lmax = 2;
  while (bb_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (bb_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = bb_delta(keys, n, i, j);
  int s = 0;
  for (int t = l;;) {
    t = (t + 1) >> 1;
    if (bb_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const uint32_t left = first == gamma ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
  const uint32_t right = last == gamma + 1 ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  rec[i].left = left; rec[i].right = right;
  rec[i].first = (uint32_t)first; rec[i].last = (uint32_t)last;
  parent[left] = (uint32_t)i;
  parent[right] = (uint32_t)i;
  if (i == 0) parent[0] = 0xffffffffu;
}

// ---- 5. boxes, bottom-up ----
// Two threads meet at every internal node; the second one needs the first one's box.  A release/acquire pair at agent scope
// (__threadfence) costs an L2 write-back + invalidate per use on this chip (the XCDs' L2s are not coherent with each other):
// 6 ms for a million triangles.  The boxes are therefore exchanged with relaxed agent-scope atomics -- stores that write through
// to the coherence point, loads that read there -- and the only ordering needed, "box complete before the counter moves", is the
// wavefront waiting for its own stores (s_waitcnt) before it issues the counter's atomic.  Both sides of that ordering are spelled
// out for the COMPILER as well: the wait is an asm statement with a memory clobber (no memory operation moves across it), and
// the second arrival reads its sibling's box behind another one (relaxed atomics to different addresses may otherwise be
// reordered).  The hardware part -- stores acknowledged at the coherence point before vmcnt reaches 0, loads issued in order
// behind the returned atomic -- is what tests/test_gpu_bvh_builder.py checks exactly on the 1M-triangle tree (check_tree_fast).
__device__ __forceinline__ float coherent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void bb_fit_kernel(const float* __restrict__ tri, const uint32_t* __restrict__ vals, uint32_t n,
                                                       const uint32_t* __restrict__ parent, BNode* __restrict__ rec, uint32_t* __restrict__ flag, bool boxes) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  Box3 b = prim_box(tri, vals[j], boxes);
  uint32_t id = n - 1 + j;
  rec[id].left = rec[id].right = 0xffffffffu; rec[id].first = rec[id].last = j;   // (a leaf: read back only by later launches)
  for (;;) {
    BNode* o = rec + id;
    coherent_store(&o->lx, b.lx); coherent_store(&o->ly, b.ly); coherent_store(&o->lz, b.lz);
    coherent_store(&o->hx, b.hx); coherent_store(&o->hy, b.hy); coherent_store(&o->hz, b.hz);
    if (n == 1) return;
    const uint32_t p = id == 0 ? 0xffffffffu : parent[id];
    if (p == 0xffffffffu) return;                     // the root's box is written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wavefront's stores have completed; nothing is moved across
    if (__hip_atomic_fetch_add(flag + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;   // the sibling subtree is not finished: its last thread continues
    asm volatile("" ::: "memory");                                // the sibling's box is read AFTER the counter said it is complete
    const uint32_t l = rec[p].left, r = rec[p].right;   // (written by the previous launch)
    const BNode* s = rec + (l == id ? r : l);
    b.lx = fminf(b.lx, coherent_load(&s->lx)); b.ly = fminf(b.ly, coherent_load(&s->ly)); b.lz = fminf(b.lz, coherent_load(&s->lz));
    b.hx = fmaxf(b.hx, coherent_load(&s->hx)); b.hy = fmaxf(b.hy, coherent_load(&s->hy)); b.hz = fmaxf(b.hz, coherent_load(&s->hz));
    id = p;
  }
}

// ---- 6. collapse + quantise + emit ----
// smallest e with extent / 255 <= 2^e (bvh.cpp:215-264 picks ceil(log2(extent / 255))), from the float's own exponent: exact
__device__ __forceinline__ int bb_pick_exp(float extent) {
  if (!(extent > 0.0f) || extent > 3.0e38f) return 0;
  int k;
  const float m = frexpf(extent / 255.0f, &k);   // extent / 255 = m * 2^k, m in [0.5, 1)
  int e = m == 0.5f ? k - 1 : k;
  return max(-126, min(126, e));
}

// q_lo, q_hi of one axis of one child at exponent e; false if the child does not fit 8 bits there
__device__ __forceinline__ bool bb_quant_axis(float origin, int e, float cmin, float cmax, uint32_t& qlo, uint32_t& qhi) {
  const float s = ldexpf(1.0f, e);
  float fl = floorf((cmin - origin) / s), fh = ceilf((cmax - origin) / s);
  if (!(fl >= 0.0f)) fl = 0.0f;
  if (!(fh >= fl)) fh = fl;
  if (fh > 255.0f) return false;
  int lo = (int)fl, hi = (int)fh;
  if (lo > 255) return false;
  // conservative after the decode's own rounding (origin + q * 2^e rounds once)
  while (lo > 0 && origin + ldexpf((float)lo, e) > cmin) --lo;
  while (hi < 255 && origin + ldexpf((float)hi, e) < cmax) ++hi;
  if (origin + ldexpf((float)hi, e) < cmax) return false;
  qlo = (uint32_t)lo; qhi = (uint32_t)hi;
  return true;
}

struct CollapseArgs {
  const BNode* rec;
  uint32_t n, leaf_max, tri_offset, node_capacity;
  uint32_t* nodes;          // 13 dwords per node
  uint32_t* counters;       // [0] nodes allocated, [1] leaves, [2] largest leaf, [3] deepest level, [4] error flags, [8 + L] items of level L
  const uint2* in; uint2* out;
  uint32_t level;
  const uint32_t* prim_ids;   // TLAS build: sorted position -> instance (blasIdx); nullptr = BLAS build
};

struct BRec { Box3 box; uint32_t left, right, first, last; };
__device__ __forceinline__ BRec bb_load_rec(const BNode* __restrict__ rec, uint32_t id) {
  const float4* p = (const float4*)(rec + id);
  const float4 a = p[0], b = p[1], c = p[2];
  BRec r;
  r.box.lx = a.x; r.box.ly = a.y; r.box.lz = a.z; r.left = __float_as_uint(a.w);
  r.box.hx = b.x; r.box.hy = b.y; r.box.hz = b.z; r.right = __float_as_uint(b.w);
  r.first = __float_as_uint(c.x); r.last = __float_as_uint(c.y);
  return r;
}
__device__ __forceinline__ float bb_area(const Box3& b) {
  const float x = b.hx - b.lx, y = b.hy - b.ly, z = b.hz - b.lz;
  return x * y + y * z + z * x;
}

__global__ __launch_bounds__(256) void bb_collapse_kernel(CollapseArgs A) {
  const uint32_t n_items = A.counters[8 + A.level];
  const uint32_t lane = threadIdx.x & 63u;
  // (every lane of a wavefront runs every iteration: node slots and queue positions are handed out per wavefront, one atomic
  // each, from a prefix sum over its lanes -- per item they would be 600,000 atomics on one address per level)
  for (uint32_t base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {
    const uint32_t it = base + threadIdx.x;
    const bool act = it < n_items;
    const uint2 item = act ? A.in[it] : make_uint2(A.n - 1, 0u);
    const uint32_t b = item.x, out = item.y;
    const BRec me = bb_load_rec(A.rec, b);
    const uint2 rg = make_uint2(me.first, me.last);
    const uint32_t count = rg.y - rg.x + 1;
    const Box3 bx = me.box;
    const bool leaf = count <= A.leaf_max;
    // (all arrays below are indexed with compile-time constants only -- unrolled loops, selects on k == pick -- so that they
    // live in registers: with dynamic indices they went to scratch and a level took as long as ~250 dependent scratch accesses)
    uint32_t c[4] = {0, 0, 0, 0};
    uint32_t nc = 0;
    BRec cr[4];
    cr[0] = cr[1] = cr[2] = cr[3] = me;
    if (act && !leaf) {
      c[0] = me.left; c[1] = me.right;
      nc = 2;
      cr[0] = bb_load_rec(A.rec, c[0]); cr[1] = bb_load_rec(A.rec, c[1]);
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        int pick = -1; float best = -1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((uint32_t)k < nc && cr[k].last - cr[k].first + 1 > A.leaf_max) {   // (else it becomes a leaf as it is; a single triangle always does)
            const float ar = bb_area(cr[k].box);
            if (ar > best) { best = ar; pick = k; }
          }
        }
        if (pick >= 0) {
          uint32_t gl = 0, gr = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) if (k == pick) { gl = cr[k].left; gr = cr[k].right; }
          const BRec L = bb_load_rec(A.rec, gl), R = bb_load_rec(A.rec, gr);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k == pick) { c[k] = gl; cr[k] = L; }
            if ((uint32_t)k == nc) { c[k] = gr; cr[k] = R; }
          }
          ++nc;
        }
      }
    }
    // wavefront prefix sum of the child counts -> node slots and next-level queue positions
    uint32_t incl = nc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += t; }
    const uint32_t total = __shfl(incl, 63);
    // ... and one pair of atomics per WORKGROUP (same-address atomics from every wavefront of a level are what the level waits for)
    __shared__ uint32_t s_tot[4], s_base[2];
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 0) s_tot[wv] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t bt = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
      s_base[0] = bt ? atomicAdd(A.counters + 0, bt) : 0u;
      s_base[1] = bt ? atomicAdd(A.counters + 8 + A.level + 1, bt) : 0u;
    }
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t k = 0; k < wv; ++k) woff += s_tot[k];
    const uint32_t first = s_base[0] + woff + incl - nc, pos = s_base[1] + woff + incl - nc;
    __syncthreads();   // (s_tot / s_base are rewritten by the next iteration)
    const unsigned long long leafm = __ballot(act && leaf);
    uint32_t lmax = act && leaf ? count : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = max(lmax, (uint32_t)__shfl_down(lmax, off));
    if (lane == 0 && leafm != 0ull) {
      atomicAdd(A.counters + 1, (uint32_t)__popcll(leafm));
      atomicMax(A.counters + 2, lmax);
      atomicMax(A.counters + 3, A.level);
    }
    if (!act) continue;
    uint32_t w[13];
    w[0] = __float_as_uint(bx.lx); w[1] = __float_as_uint(bx.ly); w[2] = __float_as_uint(bx.lz);
    int e[3] = {bb_pick_exp(bx.hx - bx.lx), bb_pick_exp(bx.hy - bx.ly), bb_pick_exp(bx.hz - bx.lz)};
    uint32_t ql[4][3] = {}, qh[4][3] = {};
    if (leaf) {
      if (A.prim_ids) { w[4] = 0; w[5] = A.prim_ids[rg.x]; }   // TLAS leaf (bvh.cpp:325-328): leafData = blasIdx
      else { w[4] = rg.x + A.tri_offset; w[5] = count; }       // bvh.cpp:260: already offset by the mesh's first triangle
    } else {
      if (first + nc > A.node_capacity) { atomicOr(A.counters + 4, 1u); continue; }
      const float org[3] = {bx.lx, bx.ly, bx.lz};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if ((uint32_t)k < nc && ok) {
              const float cmin = a == 0 ? cr[k].box.lx : (a == 1 ? cr[k].box.ly : cr[k].box.lz);
              const float cmax = a == 0 ? cr[k].box.hx : (a == 1 ? cr[k].box.hy : cr[k].box.hz);
              ok = bb_quant_axis(org[a], e[a], cmin, cmax, ql[k][a], qh[k][a]);
            }
          }
          if (ok) break;
          if (e[a] >= 126) { atomicOr(A.counters + 4, 2u); break; }
          ++e[a];
        }
      }
      w[4] = first;     // relative to this BLAS's first node (rt_traversal.cpp:92,119); TLAS: to its node 0
      w[5] = A.prim_ids ? 0xffffffffu : 0u;   // internal TLAS nodes carry UINT32_MAX (bvh.cpp:417)
#pragma unroll
      for (int k = 0; k < 4; ++k) if ((uint32_t)k < nc) A.out[pos + k] = make_uint2(c[k], first + k);
    }
    w[3] = (uint32_t)(uint8_t)(int8_t)e[0] | ((uint32_t)(uint8_t)(int8_t)e[1] << 8) | ((uint32_t)(uint8_t)(int8_t)e[2] << 16) | (A.prim_ids ? 1u << 24 : 0u);   // imask: 1 = TLAS node
    // children: 4 x { meta, lo x y z, hi x y z } = 28 bytes from dword 6 on
    uint64_t cbits[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool on = !leaf && (uint32_t)k < nc;
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
  }
}

// ---- 7. gather into the sorted order ----
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
  if (g_arena.base && (g_arena.dev != dev || g_arena.cap < bytes)) { (void)hipFree(g_arena.base); g_arena = Arena(); }
  if (!g_arena.base) {
    if (hipMalloc(&g_arena.base, bytes) != hipSuccess) { g_arena = Arena(); return false; }
    g_arena.cap = bytes; g_arena.dev = dev;
  }
  g_arena.used = 0;
  return true;
}

}  // namespace

static int build_common(void* d_tri, void* d_triEx, uint32_t n_tris, uint32_t tri_offset, uint32_t leaf_max,
                        void* d_nodes, uint32_t node_capacity, vxrt_bvh_info_t* info, void* stream, bool boxes) {
  if (!d_tri || !d_nodes || n_tris == 0 || n_tris > 0x0fffffffu) return -1;   // (the radix-tree search probes positions up to 3 n in 32-bit arithmetic)
  if (leaf_max == 0) leaf_max = 2;
  if (leaf_max > 15) leaf_max = 15;
  if (boxes) { leaf_max = 1; d_triEx = nullptr; }
  if ((uint64_t)node_capacity < 2ull * n_tris - 1ull) return -1;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n = n_tris;
  std::lock_guard<std::mutex> lk(g_arena_mu);   // (builds on one device are serialised on the scratch arena)
  const uint32_t n_counters = 8 + BB_MAX_LEVELS + 2;
  size_t tmp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 63u, s) != hipSuccess) return -1;
  const size_t per_tri = 2 * 8 + 2 * 4 + 2 * 48 + 2 * 4 + 4 + 2 * 8 + (d_triEx ? 64 : 36);
  if (!arena_reserve((size_t)n * per_tri + tmp_bytes + 64 * 1024)) return -1;
  Arena& sc = g_arena;
  int* cb = sc.get<int>(8);
  uint32_t* counters = sc.get<uint32_t>(n_counters);
  uint64_t* keys0 = sc.get<uint64_t>(n);
  uint64_t* keys1 = sc.get<uint64_t>(n);
  uint32_t* vals0 = sc.get<uint32_t>(n);
  uint32_t* vals1 = sc.get<uint32_t>(n);
  BNode* rec = sc.get<BNode>(2 * (size_t)n);
  uint32_t* parent = sc.get<uint32_t>(2 * (size_t)n);
  uint32_t* flag = sc.get<uint32_t>(n);
  uint2* q0 = sc.get<uint2>(n);
  uint2* q1 = sc.get<uint2>(n);
  uint32_t* gather = sc.get<uint32_t>((size_t)n * (d_triEx ? 16 : 9));
  void* tmp = sc.get<uint8_t>(tmp_bytes ? tmp_bytes : 16);
  if (!cb || !counters || !keys0 || !keys1 || !vals0 || !vals1 || !rec || !parent || !flag || !q0 || !q1 || !gather || !tmp) return -1;
  const uint32_t blocks = (n + 255u) / 256u;
  const uint32_t wide = blocks < 4096u ? blocks : 4096u;

  hipLaunchKernelGGL(bb_init_kernel, dim3(1), dim3(256), 0, s, cb, counters, n_counters, q0);
  if (hipMemsetAsync(flag, 0, (size_t)n * 4, s) != hipSuccess) return -1;
  hipLaunchKernelGGL(bb_bounds_kernel, dim3(wide < 512u ? wide : 512u), dim3(256), 0, s, (const float*)d_tri, n, cb, boxes);
  hipLaunchKernelGGL(bb_morton_kernel, dim3(blocks), dim3(256), 0, s, (const float*)d_tri, n, cb, keys0, vals0, boxes);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys0, keys1, vals0, vals1, (size_t)n, 0u, 63u, s) != hipSuccess) return -1;
  if (n > 1) hipLaunchKernelGGL(bb_tree_kernel, dim3((n - 1 + 255u) / 256u), dim3(256), 0, s, keys1, (int)n, rec, parent);
  hipLaunchKernelGGL(bb_fit_kernel, dim3(blocks), dim3(256), 0, s, (const float*)d_tri, vals1, n, parent, rec, flag, boxes);

  CollapseArgs A;
  A.rec = rec; A.n = n; A.leaf_max = leaf_max; A.tri_offset = tri_offset; A.node_capacity = node_capacity;
  A.nodes = (uint32_t*)d_nodes; A.counters = counters; A.prim_ids = boxes ? vals1 : nullptr;
  for (uint32_t L = 0; L < (uint32_t)BB_MAX_LEVELS; ++L) {
    A.in = (L & 1u) ? q1 : q0; A.out = (L & 1u) ? q0 : q1; A.level = L;
    // (level L holds at most 4^L items)
    uint32_t g = wide;
    if (L < 8) { const uint32_t items = 1u << (2 * L); g = (items + 255u) / 256u < wide ? (items + 255u) / 256u : wide; }
    hipLaunchKernelGGL(bb_collapse_kernel, dim3(g), dim3(256), 0, s, A);
  }
  // triangles (and shading records) into the sorted order, in place through a scratch copy (bvh.cpp:126-128 reorders in place);
  // instances stay where they are (a TLAS leaf names its instance)
  if (!boxes) {
    hipLaunchKernelGGL(bb_gather_kernel, dim3(wide), dim3(256), 0, s, (const uint32_t*)d_tri, vals1, n, 9u, gather);
    if (hipMemcpyAsync(d_tri, gather, (size_t)n * 36, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
  }
  if (d_triEx) {
    hipLaunchKernelGGL(bb_gather_kernel, dim3(wide), dim3(256), 0, s, (const uint32_t*)d_triEx, vals1, n, 16u, gather);
    if (hipMemcpyAsync(d_triEx, gather, (size_t)n * 64, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
  }
  std::vector<uint32_t> hc(n_counters);
  BNode hroot;
  if (hipMemcpyAsync(hc.data(), counters, n_counters * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipMemcpyAsync(&hroot, rec, sizeof hroot, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;   // record 0: the root (or the only leaf)
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  if (hipGetLastError() != hipSuccess) return -1;
  if (info) {
    info->n_nodes = hc[0]; info->n_leaves = hc[1]; info->max_leaf = hc[2]; info->max_depth = hc[3];
    info->bounds[0] = hroot.lx; info->bounds[1] = hroot.ly; info->bounds[2] = hroot.lz;
    info->bounds[3] = hroot.hx; info->bounds[4] = hroot.hy; info->bounds[5] = hroot.hz;
  }
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
  g_arena = Arena();
}
