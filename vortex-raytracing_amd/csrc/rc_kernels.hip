// HIP kernels of the reference's SOFTWARE ray caster (tests/regression/raycast, the "software twin" of the RTU
// test: SURVEY.md s8f-4) for gfx950, built on the machinery of the RTU path (rt_kernels.hip) instead of a lane-per-pixel
// copy of the CPU loop:
//   * an acceleration layout derived once per scene from the reference-format buffers: compact 64-byte BVH2 nodes that hold
//     BOTH children's boxes and complete child descriptors (one aligned fetch per node step tests two boxes; a leaf never costs
//     a node fetch: its triangle range is in the descriptor), and triangles in edge form laid out in leaf order, so the triIdx
//     indirection of the reference (render.h:95) is resolved at build time (the original index rides in the record);
//   * persistent wavefronts pulling 8x8 pixel tiles from a sharded queue; whole tiles traverse together;
//   * per-lane traversal stack of 4-byte descriptors in LDS (the reference re-tests nothing at pop time, so an entry is just
//     the node), hit attributes / world ray in LDS next to it, the active object-space ray in registers;
//   * an if-if loop (node step, leaf step, TLAS step), shading once the tile's rays have finished, mirror bounces and
//     samples as further rays of the same lane.
// Same boundary as the RTU path: the reference host program uploads its BVH2 / TLAS / instance / triangle / texture buffers and
// a 192-byte kernel_arg_t through vx_*, the backend resolves the addresses, caches the layout and calls vxrc_render_accel.
//
// Semantics restated from the reference (paths relative to tests/regression/raycast):
//   kernel loop      kernel.cpp:9-33 (samples summed, RGB32FtoRGB8)
//   ray generation   render.h:192-211
//   Trace            render.h:213-275 (iterative mirror bounce, per-instance texture)
//   traversal        render.h:75-190 (explicit stacks of BVH_STACK_SIZE = 64; note :110 pushes the NEARER BVH
//                    child first, i.e. visits the farther one first -- reproduced, it decides distance ties; the TLAS loop at
//                    :176 has it the right way round; a pushed child is visited even if hit.dist has shrunk since)
//   box / triangle   geometry.h:1442-1465 / :1416-1440 (libstdc++ min/max chains kept literally: NaN slabs of rays with a zero
//                    direction component resolve as on the host)
// Built -ffp-contract=off.  Pixel-exact against the reference's own -c output (tests/test_rc_twin.py, tests/golden/rc_*.npz).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "../../include/vortex_hip.h"

#define RC_LARGE_FLOAT 1e30f
#define RC_EPSILON 1e-6f
#define RC_STACK 64             // BVH_STACK_SIZE of the reference (deeper = undefined behaviour there, status bit here)
#ifndef RC_LDS_STACK
#define RC_LDS_STACK 8          // stack entries kept in LDS per lane; deeper ones in scratch (8 + 7 wavefronts per SIMD: +1 % over 12 + 6, +3 % over 16 + 5)
#endif
#ifndef RC_WIDE
#define RC_WIDE 1               // fast-domain rays walk the BVH2 two levels per fetch through 128-byte wide nodes (see the wide step); 0 = the two-wide walk only
#endif
#ifndef RC_WAVES
#define RC_WAVES 7              // wavefronts per SIMD the kernel is compiled for
#endif
#ifndef RC_LEAF_LANES
#define RC_LEAF_LANES 1         // leaves are tested once this many lanes hold one (or no lane holds a node): a lane with a leaf WAITS -- it cannot go on, the order of
#endif                          // its triangle tests is the reference's.  Measured: 8 / 16 / 24 / 32 lanes all 3-5 % slower than 1 (profiles/r04_w_twin_leaf_lanes.txt)
// deepest tree (internal nodes on a root-to-leaf path) whose wide walk stays inside RC_STACK entries: three entries per two levels
#define RC_WIDE_MAX_DEPTH (2u * (RC_STACK / 3u))
#define RC_STATUS_STACK 1u      // same bits as the RTU path's status word
#define RC_STATUS_ITER 2u
#define RC_STATUS_BAD_SCENE 4u
#define RC_STATUS_SLOW_BOXES 16u  // build-time only: a reachable box is not lo <= hi, finite and within 2^60: the scene keeps the libstdc++ min/max slab form
#define RC_STATUS_NO_WIDE 32u     // build-time only: some internal node's box is not exactly the union of its children's: the scene keeps the two-wide walk
#define RC_TLAS_ITER_LIMIT (1u << 20)   // the TLAS is taken as uploaded (not re-laid out), so its walk is bounded
#define RC_QUEUE_SHARDS 8u
#define RC_QUEUE_STRIDE 32u
#define RC_CTL_DWORDS (RC_QUEUE_SHARDS * RC_QUEUE_STRIDE)

extern "C" uint32_t* vxrt_status_word_device(void);   // rt_kernels.hip
extern "C" int vxrt_internal_lpt_sort(const uint32_t* cost, uint32_t* order, uint32_t n_tiles, uint32_t tiles_per_shard, uint32_t* clear, uint32_t clear_dwords, void* stream);   // rt_kernels.hip

namespace {

// child / work descriptor: bit 31 clear = internal node (compact index); bit 31 set = leaf: bit 30 clear -> inline,
// (count - 1) in [29:25], first wide triangle in [24:0]; bit 30 set -> by reference (index of the reference leaf node)
#define RCD_LEAF 0x80000000u
#define RCD_LEAF_REF 0x40000000u
#define RCD_FIRST_MASK 0x01FFFFFFu
#define RC_CUR_TLAS 0xFFFFFFF0u    // BVH stack exhausted: next step is a TLAS pop
#define RC_CUR_SHADE 0xFFFFFFF1u   // ray finished: shade
#define RC_CUR_IDLE 0xFFFFFFF2u
__device__ __forceinline__ bool rc_is_node(uint32_t d) { return d < RCD_LEAF; }
__device__ __forceinline__ bool rc_is_leaf(uint32_t d) { return d >= RCD_LEAF && d < RC_CUR_TLAS; }

struct RcDev {
  const uint32_t* tlas; uint32_t n_tlas;   // reference format, 8 dwords per node: aabbMin, leftRight, aabbMax, blasIdx
  const uint32_t* blas; uint32_t n_blas;   // 40 dwords per record
  const uint32_t* bvh; uint32_t n_bvh;     // reference nodes (leaves by reference only)
  const float* triEx;                      // 15 floats
  const uint8_t* tex; uint64_t tex_bytes;
  uint32_t tlas_root;
  const uint4* nodes_c;                    // compact nodes, one per reference node slot (only internal ones are filled)
  const float4* tri_w;                     // wide triangles in triIdx order: (v0, e1.x) (e1.yz, e2.xy) (e2.z, triIdx, -, -)
  const uint32_t* blas_root;               // per instance record: descriptor of its BVH root
  uint32_t n_tri_idx;
  uint32_t fast_boxes;                     // 1: every box of the compact nodes is lo <= hi, finite, within 2^60 (checked by the build): the sign-selected slab form is exact for rays in the fast domain
  const uint4* nodes_w;                    // wide nodes (128 B, one per internal reference node): the boxes of its GRANDchildren, two tree levels per fetch; nullptr = off
};
#define RCD_NONE 0xFFFFFFFEu               // wide node: empty slot

struct RcParams {
  float cpos[3], cfwd[3], cright[3], cup[3], viewplane[2];
  uint32_t spp, max_depth;
  float lpos[3], lcol[3], amb[3], bg[3];
};

__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

// geometry.h:1442-1465 with 1/dir hoisted (it is recomputed per test there: same value)
__device__ __forceinline__ float ray_box(float ox, float oy, float oz, float ix, float iy, float iz,
                                         float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  const float tx1 = (mnx - ox) * ix, tx2 = (mxx - ox) * ix;
  float tmin = std_min(tx1, tx2), tmax = std_max(tx1, tx2);
  const float ty1 = (mny - oy) * iy, ty2 = (mxy - oy) * iy;
  tmin = std_max(tmin, std_min(ty1, ty2)); tmax = std_min(tmax, std_max(ty1, ty2));
  const float tz1 = (mnz - oz) * iz, tz2 = (mxz - oz) * iz;
  tmin = std_max(tmin, std_min(tz1, tz2)); tmax = std_min(tmax, std_max(tz1, tz2));
  if (tmax < tmin || tmax <= 0) return RC_LARGE_FLOAT;
  return tmin;
}

// geometry.h:1416-1440 on a wide triangle (v0, edge1, edge2: the subtractions of :1418-1419 done once at build time)
__device__ __forceinline__ bool ray_tri(float ox, float oy, float oz, float dx, float dy, float dz, float4 t0, float4 t1, float4 t2,
                                        float& dist, float& bx, float& by, float& bz) {
  const float v0x = t0.x, v0y = t0.y, v0z = t0.z;
  const float e1x = t0.w, e1y = t1.x, e1z = t1.y;
  const float e2x = t1.z, e2y = t1.w, e2z = t2.x;
  const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
  const float a = e1x * hx + e1y * hy + e1z * hz;
  if (fabsf(a) < RC_EPSILON) return false;
  const float f = 1 / a;
  const float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
  const float w1 = f * (sx * hx + sy * hy + sz * hz);
  if (w1 < 0 || w1 > 1) return false;
  const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
  const float w2 = f * (dx * qx + dy * qy + dz * qz);
  if (w2 < 0 || w1 + w2 > 1) return false;
  const float tt = f * (e2x * qx + e2y * qy + e2z * qz);
  if (tt <= RC_EPSILON) return false;
  dist = tt; bx = w1; by = w2; bz = 1 - w1 - w2;
  return true;
}

__device__ __forceinline__ uint32_t f2u_x86(float f) { return (uint32_t)(long long)f; }   // uint32_t(float) as x86-64 g++ lowers it

// ---------------------------------------------------------------------------------------------
// acceleration-layout build (validates every index the BVH walk follows; a malformed scene fails on the host)
// ---------------------------------------------------------------------------------------------
// an internal node is well formed if its two children (adjacent, render.h:103-104; indices relative to the instance) lie inside
// the instance's node range and AFTER the node: both builders allocate children after their parent, and requiring it makes
// every accepted tree acyclic
__device__ __forceinline__ bool rc_node_ok(const uint32_t* ref, uint32_t i, uint32_t base, uint32_t end) {
  const uint64_t l64 = (uint64_t)base + ref[(size_t)i * 8 + 3];
  return l64 > i && l64 + 1 < end && l64 + 1 < 0x3FFFFFF0ull;
}
// descriptor of child / root node ci (reachable, so what it names must exist: violations raise the status)
__device__ uint32_t rc_child_desc(const uint32_t* ref, uint32_t ci, uint32_t base, uint32_t end, uint32_t n_tri_idx, uint32_t* status) {
  const uint32_t* cw = ref + (size_t)ci * 8;
  const uint32_t lf = cw[3], tc = cw[7];
  if (tc == 0u) {                                                // internal
    if (!rc_node_ok(ref, ci, base, end)) atomicOr(status, RC_STATUS_BAD_SCENE);
    return ci;
  }
  if ((uint64_t)lf + tc > n_tri_idx) { atomicOr(status, RC_STATUS_BAD_SCENE); return RCD_LEAF | RCD_LEAF_REF | ci; }
  if (tc <= 32u && lf <= RCD_FIRST_MASK) return RCD_LEAF | ((tc - 1u) << 25) | lf;
  return RCD_LEAF | RCD_LEAF_REF | ci;
}

// one thread per reference node; bases/ends: sorted node ranges of the instances
__global__ void rc_accel_nodes_kernel(const uint32_t* __restrict__ ref, uint32_t n_nodes, uint4* __restrict__ out,
                                      const uint32_t* __restrict__ bases, const uint32_t* __restrict__ ends, uint32_t nb,
                                      uint32_t n_tri_idx, uint32_t* status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  uint32_t base = 0, end = 0;
  bool in = false;
  for (uint32_t j = 0; j < nb; ++j) if (i >= bases[j] && i < ends[j]) { base = bases[j]; end = ends[j]; in = true; }
  if (!in) return;
  const uint32_t* w = ref + (size_t)i * 8;
  if (w[7] != 0u) return;                                        // leaf: described by its parent (or by the root descriptor)
  // a slot that is not a well-formed internal node (the unused tail and slot 1 of the reference's 2N-node buffer are zero) is left
  // alone: if it is reachable, its parent -- or the root check -- has raised the status
  if (!rc_node_ok(ref, i, base, end)) return;
  const uint32_t l = base + w[3], r = l + 1;
  const uint32_t* lw = ref + (size_t)l * 8;
  const uint32_t* rw = ref + (size_t)r * 8;
  // the fast slab form (rc_persistent_kernel) takes the plane on the side the ray comes from as the near one: exact only for lo <= hi,
  // and NaN-free only for bounded planes
  for (int k = 0; k < 3; ++k) {
    const float l0 = __uint_as_float(lw[k]), l1 = __uint_as_float(lw[4 + k]), r0 = __uint_as_float(rw[k]), r1 = __uint_as_float(rw[4 + k]);
    if (!(l0 <= l1) || !(r0 <= r1) || !(fabsf(l0) <= 0x1p+60f) || !(fabsf(l1) <= 0x1p+60f) || !(fabsf(r0) <= 0x1p+60f) || !(fabsf(r1) <= 0x1p+60f))
      atomicOr(status, RC_STATUS_SLOW_BOXES);
  }
  uint4* o = out + (size_t)i * 4;
  o[0] = make_uint4(lw[0], lw[1], lw[2], lw[4]);                 // L.min.xyz, L.max.x
  o[1] = make_uint4(lw[5], lw[6], rw[0], rw[1]);                 // L.max.yz, R.min.xy
  o[2] = make_uint4(rw[2], rw[4], rw[5], rw[6]);                 // R.min.z, R.max.xyz
  o[3] = make_uint4(rc_child_desc(ref, l, base, end, n_tri_idx, status), rc_child_desc(ref, r, base, end, n_tri_idx, status), 0u, 0u);
}

// Wide node of internal node i, 128 B: slots 0,1 = the children of its LEFT child (or, if that is a leaf, the leaf itself in slot 0), slots
// 2,3 = those of its RIGHT child; per slot a box (6 floats) and a descriptor.  One fetch then carries the walk over TWO levels of the BVH2
// (see the wide step of rc_persistent_kernel for why the visiting order, and with it every hit, stays the reference's).  What the step
// needs of the skipped level is its boxes' entry distances, to order the two sides (render.h:110): a box that is exactly the union of its
// children's has per-axis slab values that are the min / max of theirs, so the distance is recomputed from the grandchildren -- the build
// checks the union property per node and the scene keeps the two-wide walk where it fails.
//   q0..q5: 24 floats, slot k at [6k .. 6k+5] = lo.xyz, hi.xyz      q6: the four descriptors      q7.x: bit 0 = left child internal, bit 1 = right
__global__ void rc_accel_wide_kernel(const uint32_t* __restrict__ ref, uint32_t n_nodes, uint4* __restrict__ out,
                                     const uint32_t* __restrict__ bases, const uint32_t* __restrict__ ends, uint32_t nb,
                                     uint32_t n_tri_idx, uint32_t* status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  uint32_t base = 0, end = 0;
  bool in = false;
  for (uint32_t j = 0; j < nb; ++j) if (i >= bases[j] && i < ends[j]) { base = bases[j]; end = ends[j]; in = true; }
  if (!in) return;
  const uint32_t* w = ref + (size_t)i * 8;
  if (w[7] != 0u || !rc_node_ok(ref, i, base, end)) return;
  uint32_t f[24], desc[4], shape = 0u;
  for (int k = 0; k < 24; ++k) f[k] = 0u;
  for (int k = 0; k < 4; ++k) desc[k] = RCD_NONE;
  for (int side = 0; side < 2; ++side) {
    const uint32_t c = base + w[3] + (uint32_t)side;
    const uint32_t* cw = ref + (size_t)c * 8;
    if (cw[7] == 0u && rc_node_ok(ref, c, base, end)) {          // internal child: its two children fill the side's slots
      shape |= 1u << side;
      const uint32_t g0 = base + cw[3];
      for (int g = 0; g < 2; ++g) {
        const uint32_t* gw = ref + (size_t)(g0 + g) * 8;
        uint32_t* o = f + 6 * (2 * side + g);
        o[0] = gw[0]; o[1] = gw[1]; o[2] = gw[2]; o[3] = gw[4]; o[4] = gw[5]; o[5] = gw[6];
        desc[2 * side + g] = rc_child_desc(ref, g0 + g, base, end, n_tri_idx, status);
      }
      // the child's box must be exactly the union of the grandchildren's (float equality, per plane)
      const uint32_t* a = ref + (size_t)g0 * 8; const uint32_t* b = a + 8;
      for (int k = 0; k < 3; ++k) {
        const float lo = fminf(__uint_as_float(a[k]), __uint_as_float(b[k])), hi = fmaxf(__uint_as_float(a[4 + k]), __uint_as_float(b[4 + k]));
        if (!(lo == __uint_as_float(cw[k])) || !(hi == __uint_as_float(cw[4 + k]))) atomicOr(status, RC_STATUS_NO_WIDE);
      }
    } else {                                                      // leaf child (or a malformed one: flagged by the two-wide build): itself, in the side's first slot
      uint32_t* o = f + 6 * (2 * side);
      o[0] = cw[0]; o[1] = cw[1]; o[2] = cw[2]; o[3] = cw[4]; o[4] = cw[5]; o[5] = cw[6];
      desc[2 * side] = rc_child_desc(ref, c, base, end, n_tri_idx, status);
    }
  }
  uint4* o = out + (size_t)i * 8;
  for (int q = 0; q < 6; ++q) o[q] = make_uint4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
  o[6] = make_uint4(desc[0], desc[1], desc[2], desc[3]);
  o[7] = make_uint4(shape, 0u, 0u, 0u);
}

// Depth of the BVH2s in INTERNAL nodes on a root-to-leaf path -- what bounds a walk's stack.  The reference's walk (render.h:99-121) leaves at
// most one entry per internal node of its path; the wide step takes two levels per fetch and leaves up to THREE (the side it does not enter,
// as its two children, and the sibling of the grandchild it does enter), about 1.5 per level: a chain-like tree 43 to 63 levels deep that the
// reference walks inside its 64 entries would overflow the wide walk.  The build therefore measures the depth and keeps the two-wide walk for
// a scene whose wide walk could need more than RC_STACK entries.  Level by level (a node's children lie after it, so the trees are acyclic):
// pass t gives every child of a node of level t the level t + 1; `deepest` ends as the last level any node reached.
__global__ void rc_accel_depth_kernel(const uint32_t* __restrict__ ref, uint32_t n_nodes, const uint32_t* __restrict__ bases, const uint32_t* __restrict__ ends,
                                      uint32_t nb, uint32_t level, uint32_t* __restrict__ depth, uint32_t* __restrict__ deepest) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  uint32_t base = 0, end = 0;
  bool in = false;
  for (uint32_t j = 0; j < nb; ++j) if (i >= bases[j] && i < ends[j]) { base = bases[j]; end = ends[j]; in = true; }
  if (!in) return;
  const uint32_t* w = ref + (size_t)i * 8;
  if (w[7] != 0u || !rc_node_ok(ref, i, base, end)) return;      // leaves (and malformed slots: flagged elsewhere) need no entry
  if (level == 1u) {                                              // the roots: node 0 of every instance (render.h:84)
    if (i == base) { depth[i] = 1u; *deepest = 1u; }
    return;
  }
  if (depth[i] != level - 1u) return;
  const uint32_t l = base + w[3];
  for (uint32_t c = l; c <= l + 1u; ++c) {
    const uint32_t* cw = ref + (size_t)c * 8;
    if (cw[7] == 0u && rc_node_ok(ref, c, base, end)) { atomicMax(&depth[c], level); *deepest = level; }   // (every writer of a pass stores the same value)
  }
}

__global__ void rc_accel_tris_kernel(const float* __restrict__ tri, const uint32_t* __restrict__ triIdx, uint32_t n_idx, uint32_t n_tris,
                                     float4* __restrict__ out, uint32_t* status) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_idx) return;
  const uint32_t ti = triIdx[j];
  if (ti >= n_tris) { atomicOr(status, RC_STATUS_BAD_SCENE); return; }
  const float* t = tri + (size_t)ti * 9;
  const float v0x = t[0], v0y = t[1], v0z = t[2];
  out[(size_t)j * 3 + 0] = make_float4(v0x, v0y, v0z, t[3] - v0x);
  out[(size_t)j * 3 + 1] = make_float4(t[4] - v0y, t[5] - v0z, t[6] - v0x, t[7] - v0y);
  out[(size_t)j * 3 + 2] = make_float4(t[8] - v0z, __uint_as_float(ti), 0.f, 0.f);
}

__global__ void rc_accel_roots_kernel(const uint32_t* __restrict__ ref, const uint32_t* __restrict__ blas, uint32_t n_blas, uint32_t n_bvh,
                                      const uint32_t* __restrict__ bases, const uint32_t* __restrict__ ends, uint32_t nb,
                                      uint32_t n_tri_idx, uint32_t* __restrict__ roots, uint32_t* status) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_blas) return;
  const uint32_t base = blas[(size_t)j * 40 + 32];
  roots[j] = RC_CUR_IDLE;
  if (base >= n_bvh || base >= 0x3FFFFFF0u) { atomicOr(status, RC_STATUS_BAD_SCENE); return; }
  uint32_t end = n_bvh;
  for (uint32_t k = 0; k < nb; ++k) if (bases[k] == base) end = ends[k];
  roots[j] = rc_child_desc(ref, base, base, end, n_tri_idx, status);       // bstack[0] = node 0 of the instance (render.h:84)
}

// ---------------------------------------------------------------------------------------------
// persistent render kernel
// ---------------------------------------------------------------------------------------------
struct RcArgs {
  uint32_t W, H, y0, y1, tiles_x, n_tiles, per_shard;
  uint32_t* dst; float* colors; uint32_t* status; uint32_t* queue;
  // optional: queue position -> tile (within each shard's band of the frame, most expensive first: learned from the context's previous
  // frame of the same window) and where this frame's cost per tile goes (loop iterations of the wavefront that traced it; a leaf-body
  // run counts three) -- the RTU path's longest-first order (rt_kernels.hip), which is worth +18 % on its serial frames
  const uint32_t* tile_order; uint32_t* tile_cost;
};

__global__ __launch_bounds__(256, RC_WAVES) void rc_persistent_kernel(RcDev sc, RcParams p, RcArgs A) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __shared__ uint32_t s_stk[4][RC_LDS_STACK][64];
  // 0-2 world origin, 3-5 world direction, 6-8 object direction, 9-11 hit bx/by/bz, 12 hit blasIdx, 13 hit triIdx
  __shared__ uint32_t s_ctx[4][14][64];
  uint32_t* const lstk = &s_stk[wave][0][lane];
  uint32_t* const ctx = &s_ctx[wave][0][lane];
#define CTX(i) ctx[(i) * 64]
#define CTXF(i) __uint_as_float(ctx[(i) * 64])
  uint32_t ovf[RC_STACK];            // stack entries past the LDS part
  uint32_t tstack[RC_STACK];         // TLAS stack (scratch; an instance list is walked once per ray)
  // active object-space ray and traversal state in registers
  float rox = 0, roy = 0, roz = 0, rix = 0, riy = 0, riz = 0, hitd = 0;
  uint32_t cur = RC_CUR_IDLE, cur_blas = 0, sp = 0, tsp = 0, titer = 0;
  bool lfast = false;                // the lane's object-space ray is in the fast domain (and the scene's boxes allow the fast slab form)
  // pixel / path state
  uint32_t px = 0, py = 0, smp = 0, bounce = 0;
  float cr = 0, cg = 0, cb = 0, rr = 0, rg = 0, rb = 0, thr = 1.0f;
  // queue: the home shard is the wavefront's physical XCD (HW_REG_XCC_ID: band s of the frame is traced by the same XCD, and found in its
  // L2, frame after frame); shards one of the workgroup's wavefronts found handed out are not polled again by the others (s_dry)
  bool queue_empty = false;
  const uint32_t shard = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) % RC_QUEUE_SHARDS;
  uint32_t tries = 0, work = 0, cost_tile = 0xFFFFFFFFu;
  __shared__ uint32_t s_dry;
  if (threadIdx.x == 0) s_dry = 0u;
  __syncthreads();

  auto push = [&](uint32_t d) {
    if (sp >= RC_STACK) { atomicOr(A.status, RC_STATUS_STACK); return; }
    if (sp < RC_LDS_STACK) lstk[sp * 64] = d; else ovf[sp] = d;
    ++sp;
  };
  auto pop = [&]() {
    if (sp == 0) { cur = RC_CUR_TLAS; return; }
    --sp;
    cur = sp < RC_LDS_STACK ? lstk[sp * 64] : ovf[sp];
  };
  // a new ray of this lane (render.h:143-150: hit reset, TLAS root on the stack)
  auto start_ray = [&](float ox, float oy, float oz, float dx, float dy, float dz) {
    CTX(0) = __float_as_uint(ox); CTX(1) = __float_as_uint(oy); CTX(2) = __float_as_uint(oz);
    CTX(3) = __float_as_uint(dx); CTX(4) = __float_as_uint(dy); CTX(5) = __float_as_uint(dz);
    hitd = RC_LARGE_FLOAT;
    CTX(9) = 0; CTX(10) = 0; CTX(11) = 0; CTX(12) = 0; CTX(13) = 0;
    tsp = 0; sp = 0; titer = 0;
    tstack[tsp++] = sc.tlas_root;
    cur = RC_CUR_TLAS;
  };
  // render.h:192-211 GenerateRay
  auto primary_ray = [&]() {
    const float x_ndc = (float)((double)(((float)px + 0.5f) / (float)A.W) - 0.5);
    const float y_ndc = (float)((double)(((float)py + 0.5f) / (float)A.H) - 0.5);
    const float x_vp = x_ndc * p.viewplane[0], y_vp = y_ndc * p.viewplane[1];
    const float cx = x_vp * p.cright[0] + y_vp * p.cup[0] + p.cfwd[0];
    const float cy = x_vp * p.cright[1] + y_vp * p.cup[1] + p.cfwd[1];
    const float cz = x_vp * p.cright[2] + y_vp * p.cup[2] + p.cfwd[2];
    const float wx = cx + p.cpos[0], wy = cy + p.cpos[1], wz = cz + p.cpos[2];
    const float vx = wx - p.cpos[0], vy = wy - p.cpos[1], vz = wz - p.cpos[2];
    const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
    rr = 0.f; rg = 0.f; rb = 0.f; thr = 1.0f; bounce = 0;
    start_ray(p.cpos[0], p.cpos[1], p.cpos[2], vx * inv, vy * inv, vz * inv);
  };

  for (;;) {
    // ================= fetch: one 8x8 tile per wavefront once every lane is idle =================
    if (__ballot(cur != RC_CUR_IDLE) == 0ull) {
      uint32_t tile = 0xFFFFFFFFu;
      uint32_t dry = *(volatile uint32_t*)&s_dry;
      while (!queue_empty && tries < RC_QUEUE_SHARDS) {
        const uint32_t sid = (shard + tries) % RC_QUEUE_SHARDS;
        if ((dry >> sid) & 1u) { ++tries; continue; }
        const uint32_t s_lo = sid * A.per_shard;
        // a shard past the end of the tile range is skipped by an explicit test the optimiser cannot fold away (the RTU kernel's guard: as a
        // select feeding `base < s_n`, this compiler dropped the `s_lo < n_tiles` half and a shard past the end handed out tiles >= n_tiles)
        uint32_t in_range;
        asm volatile("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0 ; RTGUARD shard_range" : "=s"(in_range)
                     : "s"(__builtin_amdgcn_readfirstlane(s_lo)), "s"(__builtin_amdgcn_readfirstlane(A.n_tiles)) : "scc");
        if (!in_range) { ++tries; continue; }
        const uint32_t s_n = min(A.per_shard, A.n_tiles - s_lo);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A.queue + sid * RC_QUEUE_STRIDE, 1u);
        base = __shfl(base, 0);
        if (base < s_n) { tile = s_lo + base; break; }
        if (lane == 0) atomicOr(&s_dry, 1u << sid);
        dry |= 1u << sid;
        ++tries;
      }
      if (A.tile_cost && lane == 0 && cost_tile != 0xFFFFFFFFu) A.tile_cost[cost_tile] = work;
      if (tile == 0xFFFFFFFFu) { queue_empty = true; break; }
      if (A.tile_order) tile = A.tile_order[tile];
      cost_tile = tile; work = 0;
      px = (tile % A.tiles_x) * 8u + (lane & 7u);
      py = A.y0 + (tile / A.tiles_x) * 8u + (lane >> 3);
      if (px < A.W && py < A.y1) {
        smp = 0; cr = 0.f; cg = 0.f; cb = 0.f;
        primary_ray();
      }
    }

    // ================= traverse: one step of whatever each lane holds, per iteration =================
    for (;;) {
      ++work;
      if (RC_WIDE && sc.nodes_w && __all(!rc_is_node(cur) || lfast)) {
        if (rc_is_node(cur)) {
          // ---- WIDE step: two levels of the BVH2 per fetch (fast-domain rays only, wave-uniform choice) ----
          // The reference (render.h:99-121) tests a node's two children, visits the FARTHER of the two hit ones first (:110 as written; equal
          // distances: the left one) and leaves the other on its stack; the child it visits next is popped at once -- nothing happens
          // in between -- so its own step can be taken right here, from the grandchildren's boxes in this record.  The child left on the stack
          // is replaced by ITS two children, in the order the reference will give them when it pops it (same ray, same boxes: same order).
          // They are culled against the hit distance of NOW instead of that of the pop: too little culling only visits subtrees whose every
          // triangle fails the strict `dist < hit.dist` (box distance <= triangle distance), so the hit -- index included, the order of the
          // subtrees that matter is unchanged -- is the reference's.  The distances of the skipped level, which order its two sides, are
          // recomputed from the grandchildren: for a box that is exactly the union of two others (checked by the build) every per-axis slab
          // value is the min / max of theirs, and (lo - o) * (1/d) is monotone in lo.
          const uint4* np = sc.nodes_w + (size_t)cur * 8;
          const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4], w5 = np[5], w6 = np[6];
          const uint32_t shape = np[7].x;
          const uint32_t fw[24] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w, w5.x, w5.y, w5.z, w5.w};
          const uint32_t dsc[4] = {w6.x, w6.y, w6.z, w6.w};
          const bool nx = rix < 0.0f, ny = riy < 0.0f, nz = riz < 0.0f;
          float tn[4][3], tf[4][3], d[4];
          bool v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float lx = __uint_as_float(fw[6 * k]), ly = __uint_as_float(fw[6 * k + 1]), lz = __uint_as_float(fw[6 * k + 2]);
            const float hx = __uint_as_float(fw[6 * k + 3]), hy = __uint_as_float(fw[6 * k + 4]), hz = __uint_as_float(fw[6 * k + 5]);
            tn[k][0] = ((nx ? hx : lx) - rox) * rix; tn[k][1] = ((ny ? hy : ly) - roy) * riy; tn[k][2] = ((nz ? hz : lz) - roz) * riz;
            tf[k][0] = ((nx ? lx : hx) - rox) * rix; tf[k][1] = ((ny ? ly : hy) - roy) * riy; tf[k][2] = ((nz ? lz : hz) - roz) * riz;
            const float a = fmaxf(fmaxf(tn[k][0], tn[k][1]), tn[k][2]), b = fminf(fminf(tf[k][0], tf[k][1]), tf[k][2]);
            d[k] = (b < a || b <= 0) ? RC_LARGE_FLOAT : a;
            v[k] = dsc[k] != RCD_NONE && d[k] != RC_LARGE_FLOAT && d[k] < hitd;
          }
          // entry distance of a side: the child's own box (a leaf child sits in the side's first slot) or the union of its two children's
          float dside[2];
#pragma unroll
          for (int sd = 0; sd < 2; ++sd) {
            const int k0 = 2 * sd, k1 = 2 * sd + 1;
            if ((shape >> sd) & 1u) {
              const float a = fmaxf(fmaxf(fminf(tn[k0][0], tn[k1][0]), fminf(tn[k0][1], tn[k1][1])), fminf(tn[k0][2], tn[k1][2]));
              const float b = fminf(fminf(fmaxf(tf[k0][0], tf[k1][0]), fmaxf(tf[k0][1], tf[k1][1])), fmaxf(tf[k0][2], tf[k1][2]));
              dside[sd] = (b < a || b <= 0) ? RC_LARGE_FLOAT : a;
            } else dside[sd] = d[k0];
          }
          // each side's children in visiting order (the farther of two hit ones first; equal: the first)
          uint32_t sq[2][2]; uint32_t ns[2];
#pragma unroll
          for (int sd = 0; sd < 2; ++sd) {
            const int k0 = 2 * sd, k1 = 2 * sd + 1;
            const bool both = v[k0] && v[k1];
            const bool sw = both && d[k0] < d[k1];
            sq[sd][0] = (v[k0] && !sw) ? dsc[k0] : dsc[k1];
            sq[sd][1] = sw ? dsc[k0] : dsc[k1];
            ns[sd] = (v[k0] ? 1u : 0u) + (v[k1] ? 1u : 0u);
          }
          const bool rfirst = ns[0] != 0u && ns[1] != 0u && dside[0] < dside[1];   // :110 one level up
          const uint32_t a0 = rfirst ? sq[1][0] : sq[0][0], a1 = rfirst ? sq[1][1] : sq[0][1], na = rfirst ? ns[1] : ns[0];
          const uint32_t b0 = rfirst ? sq[0][0] : sq[1][0], b1 = rfirst ? sq[0][1] : sq[1][1], nbb = rfirst ? ns[0] : ns[1];
          const uint32_t n = na + nbb;
          const uint32_t o0 = na ? a0 : b0;
          const uint32_t o1 = na == 2u ? a1 : (na == 1u ? b0 : b1);
          const uint32_t o2 = na == 2u ? b0 : b1;
          const uint32_t o3 = b1;
          if (n > 3u) push(o3);
          if (n > 2u) push(o2);
          if (n > 1u) push(o1);
          if (n) cur = o0; else pop();
        }
      } else
      if (rc_is_node(cur)) {
        // ---- BVH internal node (render.h:99-121): both children's boxes in one 64-byte record ----
        const uint4* np = sc.nodes_c + (size_t)cur * 4;
        const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        float dLeft, dRight;
        if (__all(!rc_is_node(cur) || lfast)) {
          // Fast slab form (wave-uniform choice; the other form is always valid).  For a box with lo <= hi and a ray whose slab
          // products cannot be NaN or overflow, (lo - o) * (1/d) and (hi - o) * (1/d) are ordered by the sign of 1/d alone, so the
          // std::min / std::max of each pair (geometry.h:1445-1458) IS the product with the plane on the near / far side: six
          // selects per box replace twelve compare + select pairs, and the chains collapse into v_max3 / v_min3.  Same values
          // (a +0 / -0 difference cannot reach a comparison's outcome), so the same decisions and the same hit.
          const bool nx = rix < 0.0f, ny = riy < 0.0f, nz = riz < 0.0f;
          const float lx0 = __uint_as_float(q0.x), ly0 = __uint_as_float(q0.y), lz0 = __uint_as_float(q0.z);
          const float lx1 = __uint_as_float(q0.w), ly1 = __uint_as_float(q1.x), lz1 = __uint_as_float(q1.y);
          const float rx0 = __uint_as_float(q1.z), ry0 = __uint_as_float(q1.w), rz0 = __uint_as_float(q2.x);
          const float rx1 = __uint_as_float(q2.y), ry1 = __uint_as_float(q2.z), rz1 = __uint_as_float(q2.w);
          {
            const float tn = fmaxf(fmaxf(((nx ? lx1 : lx0) - rox) * rix, ((ny ? ly1 : ly0) - roy) * riy), ((nz ? lz1 : lz0) - roz) * riz);
            const float tf = fminf(fminf(((nx ? lx0 : lx1) - rox) * rix, ((ny ? ly0 : ly1) - roy) * riy), ((nz ? lz0 : lz1) - roz) * riz);
            dLeft = (tf < tn || tf <= 0) ? RC_LARGE_FLOAT : tn;
          }
          {
            const float tn = fmaxf(fmaxf(((nx ? rx1 : rx0) - rox) * rix, ((ny ? ry1 : ry0) - roy) * riy), ((nz ? rz1 : rz0) - roz) * riz);
            const float tf = fminf(fminf(((nx ? rx0 : rx1) - rox) * rix, ((ny ? ry0 : ry1) - roy) * riy), ((nz ? rz0 : rz1) - roz) * riz);
            dRight = (tf < tn || tf <= 0) ? RC_LARGE_FLOAT : tn;
          }
        } else {
          dLeft = ray_box(rox, roy, roz, rix, riy, riz, __uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z),
                          __uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
          dRight = ray_box(rox, roy, roz, rix, riy, riz, __uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x),
                           __uint_as_float(q2.y), __uint_as_float(q2.z), __uint_as_float(q2.w));
        }
        uint32_t left = q3.x, right = q3.y;
        const bool hitLeft = (dLeft != RC_LARGE_FLOAT) && (dLeft < hitd);
        const bool hitRight = (dRight != RC_LARGE_FLOAT) && (dRight < hitd);
        if (hitLeft && hitRight) {
          if (dLeft < dRight) { const uint32_t t = left; left = right; right = t; }   // :110 as written: the farther child is visited first
          push(right);
          cur = left;
        } else if (hitLeft) cur = left;
        else if (hitRight) cur = right;
        else pop();
      }
      const unsigned long long leaf_lanes = __ballot(rc_is_leaf(cur));
      if (leaf_lanes != 0ull && (RC_LEAF_LANES <= 1 || __popcll(leaf_lanes) >= RC_LEAF_LANES || __ballot(rc_is_node(cur)) == 0ull)) {
        work += 2u;
        if (rc_is_leaf(cur)) {
          // ---- BVH leaf (render.h:88-98): triangles in triIdx order, strict '<' ----
          uint32_t first, count;
          if (cur & RCD_LEAF_REF) { const uint32_t* rn = sc.bvh + (size_t)(cur & 0x3FFFFFFFu) * 8; first = rn[3]; count = rn[7]; }
          else { first = cur & RCD_FIRST_MASK; count = ((cur >> 25) & 31u) + 1u; }
          const float cdx = CTXF(6), cdy = CTXF(7), cdz = CTXF(8);
          for (uint32_t i = 0; i < count; ++i) {
            const float4* tp = sc.tri_w + (size_t)(first + i) * 3;
            const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            float d, b0, b1, b2;
            if (ray_tri(rox, roy, roz, cdx, cdy, cdz, t0, t1, t2, d, b0, b1, b2) && d < hitd) {
              hitd = d;
              CTX(9) = __float_as_uint(b0); CTX(10) = __float_as_uint(b1); CTX(11) = __float_as_uint(b2);
              CTX(12) = cur_blas; CTX(13) = __float_as_uint(t2.y);
            }
          }
          pop();
        }
      }
      if (__ballot(cur == RC_CUR_TLAS) != 0ull) {
        if (cur == RC_CUR_TLAS) {
          // ---- TLAS step (render.h:152-187), on the buffers as uploaded ----
          if (tsp == 0) cur = RC_CUR_SHADE;
          else if (++titer > RC_TLAS_ITER_LIMIT) { atomicOr(A.status, RC_STATUS_ITER); cur = RC_CUR_SHADE; }
          else {
            const uint32_t nodeIdx = tstack[--tsp];
            if (nodeIdx >= sc.n_tlas) { atomicOr(A.status, RC_STATUS_BAD_SCENE); cur = RC_CUR_SHADE; }
            else {
              const uint32_t* nd = sc.tlas + (size_t)nodeIdx * 8;
              const uint32_t leftRight = nd[3];
              const float ox = CTXF(0), oy = CTXF(1), oz = CTXF(2), dx = CTXF(3), dy = CTXF(4), dz = CTXF(5);
              if (leftRight == 0u) {
                const uint32_t blasIdx = nd[7];
                if (blasIdx >= sc.n_blas) { atomicOr(A.status, RC_STATUS_BAD_SCENE); cur = RC_CUR_SHADE; }
                else {
                  const uint32_t* bp = sc.blas + (size_t)blasIdx * 40;
                  const float* M = (const float*)bp + 16;   // invTransform
                  // ray_t::transform (geometry.h:1411-1414): float4(v, w) * M, direction (w = 0) first, then origin (w = 1)
                  const float bdx = M[0] * dx + M[1] * dy + M[2] * dz + M[3] * 0.0f;
                  const float bdy = M[4] * dx + M[5] * dy + M[6] * dz + M[7] * 0.0f;
                  const float bdz = M[8] * dx + M[9] * dy + M[10] * dz + M[11] * 0.0f;
                  rox = M[0] * ox + M[1] * oy + M[2] * oz + M[3] * 1.0f;
                  roy = M[4] * ox + M[5] * oy + M[6] * oz + M[7] * 1.0f;
                  roz = M[8] * ox + M[9] * oy + M[10] * oz + M[11] * 1.0f;
                  rix = 1.0f / bdx; riy = 1.0f / bdy; riz = 1.0f / bdz;
                  // fast domain of the slab test (as on the RTU path): 1/d finite, non-zero, at most 2^64; origin at most 2^60
                  lfast = sc.fast_boxes != 0u && fabsf(rix) <= 0x1p+64f && fabsf(riy) <= 0x1p+64f && fabsf(riz) <= 0x1p+64f && rix != 0.0f && riy != 0.0f && riz != 0.0f &&
                          fabsf(rox) <= 0x1p+60f && fabsf(roy) <= 0x1p+60f && fabsf(roz) <= 0x1p+60f;
                  CTX(6) = __float_as_uint(bdx); CTX(7) = __float_as_uint(bdy); CTX(8) = __float_as_uint(bdz);
                  cur_blas = blasIdx;
                  sp = 0;
                  cur = sc.blas_root[blasIdx];
                  if (cur == RC_CUR_IDLE) { atomicOr(A.status, RC_STATUS_BAD_SCENE); cur = RC_CUR_TLAS; }
                }
              } else {
                uint32_t left = leftRight & 0xFFFFu, right = leftRight >> 16;
                if (left >= sc.n_tlas || right >= sc.n_tlas) { atomicOr(A.status, RC_STATUS_BAD_SCENE); cur = RC_CUR_SHADE; }
                else {
                  const float* ln = (const float*)(sc.tlas + (size_t)left * 8);
                  const float* rn = (const float*)(sc.tlas + (size_t)right * 8);
                  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
                  const float dLeft = ray_box(ox, oy, oz, ix, iy, iz, ln[0], ln[1], ln[2], ln[4], ln[5], ln[6]);
                  const float dRight = ray_box(ox, oy, oz, ix, iy, iz, rn[0], rn[1], rn[2], rn[4], rn[5], rn[6]);
                  const bool hitLeft = (dLeft != RC_LARGE_FLOAT) && (dLeft < hitd);
                  const bool hitRight = (dRight != RC_LARGE_FLOAT) && (dRight < hitd);
                  if (hitLeft && hitRight) {
                    if (dLeft > dRight) { const uint32_t t = left; left = right; right = t; }   // :176
                    if (tsp + 2 > RC_STACK) atomicOr(A.status, RC_STATUS_STACK);
                    else { tstack[tsp++] = right; tstack[tsp++] = left; }
                  } else if (hitLeft) {
                    if (tsp + 1 > RC_STACK) atomicOr(A.status, RC_STATUS_STACK); else tstack[tsp++] = left;
                  } else if (hitRight) {
                    if (tsp + 1 > RC_STACK) atomicOr(A.status, RC_STATUS_STACK); else tstack[tsp++] = right;
                  }
                }
              }
            }
          }
        }
      }
      if (__ballot(cur < RC_CUR_SHADE) == 0ull) break;   // no lane has traversal work left: shade the tile's finished rays
    }

    // ================= shade: render.h:213-275 Trace, one bounce per pass =================
    if (cur == RC_CUR_SHADE) {
      const float ox = CTXF(0), oy = CTXF(1), oz = CTXF(2), dx = CTXF(3), dy = CTXF(4), dz = CTXF(5);
      bool next_ray = false;
      if (hitd == RC_LARGE_FLOAT) {
        rr = rr + p.bg[0] * thr; rg = rg + p.bg[1] * thr; rb = rb + p.bg[2] * thr;   // :230
      } else {
        const float hbx = CTXF(9), hby = CTXF(10), hbz = CTXF(11);
        const uint32_t hblas = CTX(12), htri = CTX(13);
        const uint32_t* bp = sc.blas + (size_t)hblas * 40;
        const float* te = sc.triEx + (size_t)htri * 15;   // N0 N1 N2 uv0 uv1 uv2
        const float Ix = ox + dx * hitd, Iy = oy + dy * hitd, Iz = oz + dz * hitd;   // :239
        float Nx = te[3] * hbx + te[6] * hby + te[0] * hbz;                             // :242 N1*bx + N2*by + N0*bz
        float Ny = te[4] * hbx + te[7] * hby + te[1] * hbz;
        float Nz = te[5] * hbx + te[8] * hby + te[2] * hbz;
        const float* m = (const float*)bp + 16;
        const float z0 = 0.0f * 0.0f;
        const float Tx = m[0] * Nx + m[4] * Ny + m[8] * Nz + z0;     // float4(N,0) * transposed 3x3 (geometry.h:1141-1147)
        const float Ty = m[1] * Nx + m[5] * Ny + m[9] * Nz + z0;
        const float Tz = m[2] * Nx + m[6] * Ny + m[10] * Nz + z0;
        const float inv = 1.0f / sqrtf(Tx * Tx + Ty * Ty + Tz * Tz);
        Nx = Tx * inv; Ny = Ty * inv; Nz = Tz * inv;
        const float u = te[11] * hbx + te[13] * hby + te[9] * hbz;    // :247 uv1*bx + uv2*by + uv0*bz
        const float v = te[12] * hbx + te[14] * hby + te[10] * hbz;
        const unsigned long long tex_offset = (unsigned long long)bp[34] | ((unsigned long long)bp[35] << 32);
        const uint32_t tw = bp[36], th = bp[37];
        float cr_ = 0.f, cg_ = 0.f, cb_ = 0.f;
        if (tw == 0u || th == 0u || tex_offset > sc.tex_bytes || (unsigned long long)tw * th > (sc.tex_bytes - tex_offset) / 4ull) {
          atomicOr(A.status, RC_STATUS_BAD_SCENE);
        } else {
          uint32_t iu = f2u_x86(u * (float)tw), iv = f2u_x86(v * (float)th);
          iu %= tw; iv %= th;
          const uint32_t texel = ((const uint32_t*)(sc.tex + tex_offset))[iu + iv * tw];
          const float s256 = 1 / 256.0f;
          cr_ = (float)(int)((texel >> 16) & 255) * s256; cg_ = (float)(int)((texel >> 8) & 255) * s256; cb_ = (float)(int)(texel & 255) * s256;
        }
        // diffuseLighting :59-71
        float Lx = p.lpos[0] - Ix, Ly = p.lpos[1] - Iy, Lz = p.lpos[2] - Iz;
        const float dist = sqrtf(Lx * Lx + Ly * Ly + Lz * Lz);
        const float il = 1.0f / dist;
        Lx *= il; Ly *= il; Lz *= il;
        const float att = 1.0f / (1.0f + dist * 0.1f);
        const float NdotL = std_max(0.0f, Nx * Lx + Ny * Ly + Nz * Lz);
        const float dr = cr_ * (p.amb[0] + att * p.lcol[0] * NdotL);
        const float dg = cg_ * (p.amb[1] + att * p.lcol[1] * NdotL);
        const float db = cb_ * (p.amb[2] + att * p.lcol[2] * NdotL);
        const float refl = __uint_as_float(bp[38]);
        rr = rr + thr * dr * (1 - refl); rg = rg + thr * dg * (1 - refl); rb = rb + thr * db * (1 - refl);   // :257
        thr *= refl;                                                                                           // :260
        if (refl > 0.0f && bounce + 1 < p.max_depth) {                                                          // :263-268
          const float nd = Nx * dx + Ny * dy + Nz * dz;
          const float vx = dx - (2.0f * Nx) * nd, vy = dy - (2.0f * Ny) * nd, vz = dz - (2.0f * Nz) * nd;
          const float rinv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
          const float Rx = vx * rinv, Ry = vy * rinv, Rz = vz * rinv;
          ++bounce;
          start_ray(Ix + Rx * 0.001f, Iy + Ry * 0.001f, Iz + Rz * 0.001f, Rx, Ry, Rz);
          next_ray = true;
        } else {
          rr = rr + thr * p.bg[0]; rg = rg + thr * p.bg[1]; rb = rb + thr * p.bg[2];                           // :271
        }
      }
      if (!next_ray) {
        // (a sample whose loop ran out of depth ends here too: `for bounce < max_depth` leaves with the radiance so far)
        cr = cr + rr; cg = cg + rg; cb = cb + rb;          // kernel.cpp:24
        if (++smp < p.spp) primary_ray();
        else {
          const size_t idx = (size_t)px + (size_t)py * A.W;
          const int ir = (int)(std_min(cr, 1.f) * 255), ig = (int)(std_min(cg, 1.f) * 255), ib = (int)(std_min(cb, 1.f) * 255);   // common.h:107-112
          A.dst[idx] = (uint32_t)((ir << 16) + (ig << 8) + ib);
          if (A.colors) { A.colors[3 * idx] = cr; A.colors[3 * idx + 1] = cg; A.colors[3 * idx + 2] = cb; }
          cur = RC_CUR_IDLE;
        }
      }
    }
  }
#undef CTX
#undef CTXF
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
// Per-frame state: queue counters, the cost of the last frame's tiles and the order derived from it.  Two of them, handed out by stream, so
// that frames issued alternately on two streams overlap (the tail of one launch is filled by the head of the next, as on the RTU path).
struct RcFrameCtx {
  uint32_t* ctl = nullptr; uint32_t* cost = nullptr; uint32_t* order = nullptr;
  uint32_t cap = 0, key[4] = {0, 0, 0, 0};
  bool valid = false, ctl_dirty = true, busy = false, done_recorded = false;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};
struct vxrc_accel {
  vxrc_scene_t ref{};
  void* nodes_c = nullptr; void* tri_w = nullptr; void* blas_root = nullptr; void* nodes_w = nullptr;
#ifndef RC_CTXS
#define RC_CTXS 4             // frame contexts of a layout: frames issued round robin on up to 4 streams overlap (1 / 2 / 3 / 4 in flight: 3.68 / 4.08 / 4.17 / 4.19 Grays/s, profiles/r04_af_twin_frames_in_flight.txt)
#endif
  RcFrameCtx ctx[RC_CTXS];
  uint32_t next_ctx = 0;
  bool multi_stream = false; hipStream_t first_stream = nullptr; bool stream_seen = false;
  uint32_t fast_boxes = 0;   // see RcDev
  uint32_t depth = 0;        // internal nodes on the longest root-to-leaf path (measured when the wide layout was built; 0 = not measured)
};

extern "C" int vxrc_accel_destroy(vxrc_accel_t* a) {
  if (!a) return 0;
  (void)hipDeviceSynchronize();
  (void)hipFree(a->nodes_c); (void)hipFree(a->tri_w); (void)hipFree(a->blas_root); (void)hipFree(a->nodes_w);
  for (RcFrameCtx& c : a->ctx) {
    (void)hipFree(c.ctl); (void)hipFree(c.cost); (void)hipFree(c.order);
    if (c.done) (void)hipEventDestroy(c.done);
  }
  delete a;
  return 0;
}

extern "C" int vxrc_accel_info(const vxrc_accel_t* a, uint32_t which, uint64_t* value) {
  if (!a || !value) return -1;
  switch (which) {
  case 0: *value = a->nodes_w ? 1u : 0u; return 0;
  case 1: *value = a->depth; return 0;
  }
  return -1;
}

// frame context for a call on stream s: the one this stream used last, else an unused one, else the other one behind its completion event
static RcFrameCtx* rc_acquire_ctx(vxrc_accel* a, hipStream_t s) {
  if (!a->stream_seen) { a->stream_seen = true; a->first_stream = s; } else if (s != a->first_stream) a->multi_stream = true;
  RcFrameCtx* c = nullptr;
  for (RcFrameCtx& k : a->ctx) if (!c && k.busy && k.stream == s) c = &k;
  for (RcFrameCtx& k : a->ctx) if (!c && !k.busy) c = &k;
  if (!c) c = &a->ctx[a->next_ctx++ % RC_CTXS];
  if (!c->ctl) {
    if (hipMalloc((void**)&c->ctl, RC_CTL_DWORDS * 4) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) return nullptr;
    c->ctl_dirty = true;
  }
  if (c->busy && c->stream != s) {
    if (c->done_recorded) { if (hipStreamWaitEvent(s, c->done, 0) != hipSuccess) return nullptr; }
    else if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); if (hipDeviceSynchronize() != hipSuccess) return nullptr; }
  }
  return c;
}
static int rc_release_ctx(vxrc_accel* a, RcFrameCtx* c, hipStream_t s) {
  c->done_recorded = a->multi_stream;
  if (c->done_recorded && hipEventRecord(c->done, s) != hipSuccess) return -1;
  c->busy = true; c->stream = s;
  return 0;
}

extern "C" int vxrc_accel_build(const vxrc_scene_t* s, void* stream, vxrc_accel_t** out) {
  if (!s || !out) return -1;
  if (!s->tlas || !s->blas || !s->bvh || !s->tri || !s->triEx || !s->triIdx || !s->tex) return -1;
  if (s->n_tlas_nodes == 0 || s->n_blas == 0 || s->n_bvh_nodes == 0 || s->n_tris == 0 || s->n_tri_idx == 0) return -1;
  if (s->tlas_root >= s->n_tlas_nodes) return -1;
  if (s->n_bvh_nodes >= 0x3FFFFFF0u) return -1;   // node indices share the descriptor space with the state markers
  hipStream_t st = (hipStream_t)stream;
  std::vector<uint32_t> recs((size_t)s->n_blas * 40);
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  if (hipMemcpy(recs.data(), s->blas, recs.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  std::vector<uint32_t> bases;
  for (uint32_t j = 0; j < s->n_blas; ++j) {
    const uint32_t off = recs[(size_t)j * 40 + 32];
    if (off >= s->n_bvh_nodes) return -1;
    bases.push_back(off);
  }
  std::sort(bases.begin(), bases.end());
  bases.erase(std::unique(bases.begin(), bases.end()), bases.end());
  std::vector<uint32_t> ends(bases.size());
  for (size_t j = 0; j < bases.size(); ++j) ends[j] = j + 1 < bases.size() ? bases[j + 1] : s->n_bvh_nodes;
  auto a = new (std::nothrow) vxrc_accel();
  if (!a) return -1;
  a->ref = *s;
  uint32_t* d_ranges = nullptr; uint32_t* d_status = nullptr; uint32_t* d_depth = nullptr;
  static const bool wide_on = [] { const char* e = getenv("VXRC_WIDE"); return !(e && e[0] == '0'); }();
  bool ok = hipMalloc(&a->nodes_c, (size_t)s->n_bvh_nodes * 64) == hipSuccess &&
            (!(RC_WIDE && wide_on) || hipMalloc(&a->nodes_w, (size_t)s->n_bvh_nodes * 128) == hipSuccess) &&
            hipMalloc(&a->tri_w, (size_t)s->n_tri_idx * 48) == hipSuccess &&
            hipMalloc(&a->blas_root, (size_t)s->n_blas * 4) == hipSuccess &&
            hipMalloc((void**)&d_ranges, bases.size() * 8) == hipSuccess && hipMalloc((void**)&d_status, 8) == hipSuccess &&
            (!a->nodes_w || hipMalloc((void**)&d_depth, (size_t)s->n_bvh_nodes * 4) == hipSuccess);
  uint32_t hstatus = 0, deepest = 0;
  if (ok) {
    ok = hipMemcpy(d_ranges, bases.data(), bases.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_ranges + bases.size(), ends.data(), ends.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(d_status, 0, 8) == hipSuccess && hipMemsetAsync(a->tri_w, 0, (size_t)s->n_tri_idx * 48, st) == hipSuccess &&
         (!d_depth || hipMemsetAsync(d_depth, 0, (size_t)s->n_bvh_nodes * 4, st) == hipSuccess);
  }
  if (ok) {
    const uint32_t nb = (uint32_t)bases.size();
    hipLaunchKernelGGL(rc_accel_nodes_kernel, dim3((s->n_bvh_nodes + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->bvh, s->n_bvh_nodes, (uint4*)a->nodes_c,
                       d_ranges, d_ranges + nb, nb, s->n_tri_idx, d_status);
    if (a->nodes_w) hipLaunchKernelGGL(rc_accel_wide_kernel, dim3((s->n_bvh_nodes + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->bvh, s->n_bvh_nodes, (uint4*)a->nodes_w,
                                       d_ranges, d_ranges + nb, nb, s->n_tri_idx, d_status);
    hipLaunchKernelGGL(rc_accel_tris_kernel, dim3((s->n_tri_idx + 255) / 256), dim3(256), 0, st, (const float*)s->tri, (const uint32_t*)s->triIdx, s->n_tri_idx,
                       s->n_tris, (float4*)a->tri_w, d_status);
    hipLaunchKernelGGL(rc_accel_roots_kernel, dim3((s->n_blas + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->bvh, (const uint32_t*)s->blas, s->n_blas,
                       s->n_bvh_nodes, d_ranges, d_ranges + nb, nb, s->n_tri_idx, (uint32_t*)a->blas_root, d_status);
    // depth of the trees, for the wide walk's stack (see rc_accel_depth_kernel): one pass per level, up to the first level the wide walk
    // could not hold any more (no host round trip in between: a pass past the deepest level finds nothing to do)
    if (d_depth)
      for (uint32_t level = 1; level <= RC_WIDE_MAX_DEPTH + 1u; ++level)
        hipLaunchKernelGGL(rc_accel_depth_kernel, dim3((s->n_bvh_nodes + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->bvh, s->n_bvh_nodes,
                           d_ranges, d_ranges + nb, nb, level, d_depth, d_status + 1);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
         hipMemcpy(&hstatus, d_status, 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(&deepest, d_status + 1, 4, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d_ranges); (void)hipFree(d_status); (void)hipFree(d_depth);
  if (!ok || (hstatus & ~(RC_STATUS_SLOW_BOXES | RC_STATUS_NO_WIDE)) != 0) { vxrc_accel_destroy(a); return -1; }
  a->fast_boxes = (hstatus & RC_STATUS_SLOW_BOXES) ? 0u : 1u;
  if (a->nodes_w && (hstatus & (RC_STATUS_SLOW_BOXES | RC_STATUS_NO_WIDE))) { (void)hipFree(a->nodes_w); a->nodes_w = nullptr; }   // (the wide walk needs both properties)
  // ... and a stack that holds it: up to three entries per two levels.  A deeper tree keeps the reference's own walk, one entry per level.
  if (a->nodes_w && deepest > RC_WIDE_MAX_DEPTH) { (void)hipFree(a->nodes_w); a->nodes_w = nullptr; }
  a->depth = deepest;
  *out = a;
  return 0;
}

extern "C" int vxrc_render_accel(vxrc_accel_t* a, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                                 const vxrc_params_t* prm, uint32_t* dst, float* colors, void* stream) {
  if (!a || !prm || !dst) return -1;
  if (width == 0 || height == 0 || y0 > y1 || y1 > height) return -1;
  if (prm->samples_per_pixel == 0) return -1;   // leaves every pixel black in the reference; refuse like the RTU path
  if (y0 == y1) return 0;
  uint32_t* st = vxrt_status_word_device();
  if (!st) return -1;
  const vxrc_scene_t* s = &a->ref;
  RcDev d{};
  d.tlas = (const uint32_t*)s->tlas; d.n_tlas = s->n_tlas_nodes;
  d.blas = (const uint32_t*)s->blas; d.n_blas = s->n_blas;
  d.bvh = (const uint32_t*)s->bvh; d.n_bvh = s->n_bvh_nodes;
  d.triEx = (const float*)s->triEx;
  d.tex = (const uint8_t*)s->tex; d.tex_bytes = s->tex_bytes;
  d.tlas_root = s->tlas_root;
  d.nodes_c = (const uint4*)a->nodes_c; d.tri_w = (const float4*)a->tri_w; d.blas_root = (const uint32_t*)a->blas_root;
  d.n_tri_idx = s->n_tri_idx;
  d.fast_boxes = a->fast_boxes;
  d.nodes_w = (const uint4*)a->nodes_w;
  RcParams p{};
  for (int i = 0; i < 3; ++i) {
    p.cpos[i] = prm->camera_pos[i]; p.cfwd[i] = prm->camera_forward[i]; p.cright[i] = prm->camera_right[i]; p.cup[i] = prm->camera_up[i];
    p.lpos[i] = prm->light_pos[i]; p.lcol[i] = prm->light_color[i]; p.amb[i] = prm->ambient_color[i]; p.bg[i] = prm->background_color[i];
  }
  p.viewplane[0] = prm->viewplane[0]; p.viewplane[1] = prm->viewplane[1];
  p.spp = prm->samples_per_pixel; p.max_depth = prm->max_depth;
  hipStream_t hs = (hipStream_t)stream;
  RcArgs A{};
  A.W = width; A.H = height; A.y0 = y0; A.y1 = y1;
  A.tiles_x = (width + 7) / 8;
  const uint64_t nt = (uint64_t)A.tiles_x * ((y1 - y0 + 7) / 8);
  if (nt > 0x3ffffffull) return -1;
  A.n_tiles = (uint32_t)nt;
  A.per_shard = (A.n_tiles + RC_QUEUE_SHARDS - 1) / RC_QUEUE_SHARDS;
  RcFrameCtx* c = rc_acquire_ctx(a, hs);
  if (!c) return -1;
  auto fail = [&]() -> int { c->ctl_dirty = true; (void)rc_release_ctx(a, c, hs); return -1; };
  A.dst = dst; A.colors = colors; A.status = st; A.queue = c->ctl;
  if (c->ctl_dirty && hipMemsetAsync(c->ctl, 0, RC_CTL_DWORDS * 4, hs) != hipSuccess) return fail();
  c->ctl_dirty = true;    // until the sort launch that clears the counters again is enqueued
  // longest tile first inside each band, learned from this context's previous frame of the same window (VXRC_LPT=0: off)
  static const bool lpt_on = [] { const char* e = getenv("VXRC_LPT"); return !(e && e[0] == '0'); }();
  const bool lpt = lpt_on && A.n_tiles >= 4096u;
  if (lpt) {
    if (c->cap < A.n_tiles) {
      if (hipStreamSynchronize(hs) != hipSuccess) return fail();
      (void)hipFree(c->cost); (void)hipFree(c->order);
      c->cost = c->order = nullptr; c->cap = 0; c->valid = false;
      if (hipMalloc((void**)&c->cost, (size_t)A.n_tiles * 4) != hipSuccess || hipMalloc((void**)&c->order, (size_t)A.n_tiles * 4) != hipSuccess) return fail();
      c->cap = A.n_tiles;
    }
    const uint32_t key[4] = {width, height, y0, y1};
    if (memcmp(key, c->key, sizeof key) != 0) { c->valid = false; memcpy(c->key, key, sizeof key); }
    A.tile_cost = c->cost;
    A.tile_order = c->valid ? c->order : nullptr;
  }
  static int per_cu = 0, cus = 0;
  if (!per_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rc_persistent_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  }
  const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * cus, (nt + 3) / 4);
  hipLaunchKernelGGL(rc_persistent_kernel, dim3(grid ? grid : 1), dim3(256), 0, hs, d, p, A);
  if (hipGetLastError() != hipSuccess) return fail();
  if (lpt) {
    // the order for this context's next frame, and the queue counters cleared behind the frame (one launch instead of the fill)
    if (vxrt_internal_lpt_sort(c->cost, c->order, A.n_tiles, A.per_shard, c->ctl, RC_CTL_DWORDS, hs) != 0) return fail();
    c->valid = true; c->ctl_dirty = false;
  }
  return rc_release_ctx(a, c, hs);
}

// one-shot form: layout built, frame rendered, layout freed (tests, callers that render a scene once)
extern "C" int vxrc_render(const vxrc_scene_t* s, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                           const vxrc_params_t* prm, uint32_t* dst, float* colors, void* stream) {
  if (!s || !prm || !dst) return -1;
  if (width == 0 || height == 0 || y0 > y1 || y1 > height) return -1;
  if (prm->samples_per_pixel == 0) return -1;
  vxrc_accel_t* a = nullptr;
  if (vxrc_accel_build(s, stream, &a) != 0) return -1;
  const int rc = vxrc_render_accel(a, width, height, y0, y1, prm, dst, colors, stream);
  vxrc_accel_destroy(a);   // (synchronises the device)
  return rc;
}
