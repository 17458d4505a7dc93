// HIP kernels of the ray-tracing hot path for gfx950 (MI355X): 4-wide quantized-BVH traversal
// (TLAS -> BLAS), Moller-Trumbore, Lambert shade, RGB8 pack.  Hand-written for CDNA4 wave64.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off   (contraction OFF is part of the
// contract: SURVEY.md s7 "FP contraction"; division and sqrt are the correctly rounded forms,
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).  The only fused operation is the
// explicit fma in the child-box decode, where the product is exact (see eval_children).
//
// Semantics restated from the reference (paths relative to the reference repo):
//   traversal     sim/simx/rt_traversal.cpp:26-213 + sim/simx/rt_unit.cpp:98-116,199-202
//   box / tri     sim/simx/rt_traversal.cpp:318-339 / :263-316 ; instance transform :231-261
//   ray gen       tests/regression/raytracing/kernel.cpp:28-39
//   shading       shaders/closest.cpp:57-127, shaders/miss.cpp:9-14, rtx_shading.h:5-18,55-67
//   pixel pack    common.h:149-154, kernel.cpp:95-106
// The trail/short-stack/restart machinery of the simulator is replaced by one pass over a full
// per-lane stack whose entries carry m = max(entry distance along the path); DESIGN.md s3 proves
// this returns the same hit (index included) as the reference's accept-and-re-descend loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_types.h"
#include "../../include/vortex_hip.h"

#define STATUS_STACK_OVERFLOW 1u
#define STATUS_ITER_LIMIT 2u
#define STATUS_BAD_SCENE 4u
#define STATUS_FMA_DECODE_DIFFERS 8u   // build-time only: selects the ldexp decode for the scene

// ---------------------------------------------------------------------------------------------
// Device-side acceleration layout, derived once per scene from the reference-format buffers by the
// accel_* kernels below (the reference bytes stay the source of truth).
//
// Work descriptor, 32 bit: [31:30] kind, [29:0] payload
//     kind 0  TLAS internal node   payload = compact node index (TLAS nodes come first)
//     kind 1  BLAS internal node   payload = compact node index (n_tlas + index in the bvh buffer)
//     kind 2  BLAS leaf            payload = count<<26 | firstTriangle   (count 1..15; count 0:
//                                  payload = index of the reference leaf node, range read from it)
//     kind 3  instance (TLAS leaf) payload = blasIdx
//     0xFFFFFFFF = empty child slot, 0xFFFFFFFE = ray finished, 0xFFFFFFFD = lane idle
//
// Compact node, 64 B = half a cache line, one per INTERNAL node, TLAS and BLAS nodes in one index space:
//     q0 = origin.xyz, 2^ex as float
//     q1 = plane words lo.x, lo.y, lo.z, hi.x     one byte per child: byte k of word j = plane j of child k
//     q2 = plane words hi.y, hi.z, desc0, desc1   desc k = complete work descriptor of child k
//     q3 = desc2, desc3, 2^ey, 2^ez
//   -> a node visit is four 16-byte loads per lane (the vector-memory return path, not HBM, is what
//      saturates first on MI355X for wider nodes: profiles/r01_b_*), no index arithmetic on children,
//      a leaf or instance never costs a node fetch of its own, and stack entries are 2 dwords.
// Wide triangle, 48 B: v0, edge1 = v1 - v0, edge2 = v2 - v0 (the subtractions of
//     rt_traversal.cpp:272-278 done once), three aligned 16-byte loads.
// ---------------------------------------------------------------------------------------------
#define DK_TLAS 0u
#define DK_BLAS 1u
#define DK_LEAF 2u
#define DK_INST 3u
#define DESC(kind, payload) (((kind) << 30) | (payload))
#define DESC_DONE 0xFFFFFFFEu   // traversal of the lane's ray has ended
#define DESC_IDLE 0xFFFFFFFDu   // lane has no ray
#define DESC_NONE 0xFFFFFFFFu   // compact node: empty child slot
#define PAYLOAD_MASK 0x3FFFFFFFu
#define DESC_TOP_FLAG 0x20000000u   // node descriptor: payload = slot of the LDS-staged top-of-tree image (kernels with USE_TOP only)
#define DESC_TOP_SLOT 0x0000FFFFu
#define LEAF_FIRST_BITS 26
#define LEAF_FIRST_MASK 0x03FFFFFFu
#define LEAF_MAX_INLINE 15u
#define CNODE_VEC4 4
#define WTRI_FLOATS 12

struct SceneDev {
  const uint4* nodes_c;      // compact nodes: the TLAS nodes, then the BLAS nodes (one index space, no per-lane base select)
  const uint32_t* ref_tlas;  // reference TLAS nodes (13 dwords each): exponents for the ldexp decode
  uint32_t n_tlas;           // compact index of BLAS node j = n_tlas + j
  const float4* tri_w;       // wide triangles
  const uint32_t* blas_root; // per instance record: descriptor of its BLAS root
  uint32_t tlas_root;        // descriptor of the TLAS root
  uint32_t exact_decode;     // 1: decode child boxes with ldexp instead of the exact-product fma
  const uint32_t* ref_bvh;   // reference bvh nodes (13 dwords each): ranges of leaves > 15 triangles
  const uint32_t* blas;      // reference blas_node_t records (40 dwords each)
  const rt_triex_t* triEx;
  const rt_material_t* mat;
  const uint8_t* tex;
  // top of the tree for LDS staging (accel_top_kernel): the first n_top internal nodes in breadth-first order from the
  // TLAS root, as four planes of n_top uint4 (q0[], q1[], q2[], q3[]: conflict-free ds_read_b128 for neighbouring slots);
  // child descriptors inside the image and the *_top roots address staged nodes by slot (DESC_TOP_FLAG)
  const uint4* top_img;
  uint32_t n_top;
  uint32_t tlas_root_top;
  const uint32_t* blas_root_top;
  uint32_t ident_root;       // 1: the TLAS root is an instance leaf whose inverse transform is the identity (see start_ray)
};

struct HitRec { float dist, bx, by, bz; uint32_t blasIdx, triIdx; };

// libstdc++ std::min / std::max (rt_traversal.cpp:327-337 use them; NaN behaviour is part of parity)
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

// rt_traversal.cpp:318-339 with idir hoisted (1.0f/rd is recomputed per child there; same value).
// EXACT selects the libstdc++ min/max forms; the fast form uses v_min/v_max, which differs only
// when a NaN is present (0*inf), i.e. only if some ray direction component is 0/inf/NaN.
template <bool EXACT>
__device__ __forceinline__ float slab_interval(float tx1, float tx2, float ty1, float ty2, float tz1, float tz2) {
  float tmin, tmax;
  if (EXACT) {
    tmin = std_min(tx1, tx2);
    tmax = std_max(tx1, tx2);
    tmin = std_max(tmin, std_min(ty1, ty2));
    tmax = std_min(tmax, std_max(ty1, ty2));
    tmin = std_max(tmin, std_min(tz1, tz2));
    tmax = std_min(tmax, std_max(tz1, tz2));
  } else {
    tmin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));   // v_min x3, v_max3
    tmax = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
  }
  return (tmax < tmin || tmax <= 0) ? RT_LARGE_FLOAT : tmin;
}

template <bool EXACT>
__device__ __forceinline__ float ray_box(float ox, float oy, float oz, float ix, float iy, float iz,
                                         float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  float tx1 = (mnx - ox) * ix, tx2 = (mxx - ox) * ix;
  float ty1 = (mny - oy) * iy, ty2 = (mxy - oy) * iy;
  float tz1 = (mnz - oz) * iz, tz2 = (mxz - oz) * iz;
  return slab_interval<EXACT>(tx1, tx2, ty1, ty2, tz1, tz2);
}

// rt_traversal.cpp:263-316 on a wide triangle (v0, edge1, edge2)
__device__ __forceinline__ float ray_tri(float ox, float oy, float oz, float dx, float dy, float dz,
                                         float4 t0, float4 t1, float4 t2, float& bx, float& by, float& bz) {
  const float v0x = t0.x, v0y = t0.y, v0z = t0.z;
  const float e1x = t0.w, e1y = t1.x, e1z = t1.y;
  const float e2x = t1.z, e2y = t1.w, e2z = t2.x;
  float hx = dy * e2z - dz * e2y;
  float hy = dz * e2x - dx * e2z;
  float hz = dx * e2y - dy * e2x;
  float a = e1x * hx + e1y * hy + e1z * hz;
  if (fabsf(a) < RT_EPSILON) return RT_LARGE_FLOAT;
  float f = 1 / a;
  float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
  float w1 = f * (sx * hx + sy * hy + sz * hz);
  if (w1 < 0 || w1 > 1) return RT_LARGE_FLOAT;
  float qx = sy * e1z - sz * e1y;
  float qy = sz * e1x - sx * e1z;
  float qz = sx * e1y - sy * e1x;
  float w2 = f * (dx * qx + dy * qy + dz * qz);
  if (w2 < 0 || w1 + w2 > 1) return RT_LARGE_FLOAT;
  float tf = f * (e2x * qx + e2y * qy + e2z * qz);
  if (tf <= RT_EPSILON) return RT_LARGE_FLOAT;
  bx = w1;
  by = w2;
  bz = 1 - w1 - w2;
  return tf;
}

struct Cand { float d; uint32_t desc; };
// visit order: nearer first; equal distance -> higher child index first (stable far->near sort of
// rt_traversal.cpp:76-78 read from the back).  Filtered children carry d = +inf.
// Adjacent compare-exchange with a strict '<' never reorders equal keys, so a 6-comparator bubble
// network over the children laid out [3, 2, 1, 0] gives exactly that order without carrying the index.
__device__ __forceinline__ void cmpx(Cand& x, Cand& y) {
  const bool sw = y.d < x.d;
  const Cand tx = x, ty = y;
  x.d = sw ? ty.d : tx.d; x.desc = sw ? ty.desc : tx.desc;
  y.d = sw ? tx.d : ty.d; y.desc = sw ? tx.desc : ty.desc;
}
__device__ __forceinline__ void order_children(Cand* c) {   // in: c[k] = child k; out: c[0] nearest ... c[3] farthest
  Cand s0 = c[3], s1 = c[2], s2 = c[1], s3 = c[0];
  cmpx(s0, s1); cmpx(s1, s2); cmpx(s2, s3); cmpx(s0, s1); cmpx(s1, s2); cmpx(s0, s1);
  c[0] = s0; c[1] = s1; c[2] = s2; c[3] = s3;
}

template <int K>
__device__ __forceinline__ float qbyte(uint32_t w) { return (float)((w >> (8 * K)) & 0xffu); }   // v_cvt_f32_ubyteK

// Box test of child K of an internal node (rt_traversal.cpp:59-74).  pl[0..2] = the lo planes of x, y, z
// and pl[3..5] the hi planes, one byte per child.
// Decode: the reference computes origin + ldexp(float(q), e) (:61-67).  float(q) * 2^e is exact for
// an 8-bit q whenever 2^e is representable, so fma(float(q), 2^e, origin) rounds the same exact sum
// once and yields the identical float with one instruction less per coordinate.
// Fast form (!EXACT && !LDEXP): with q_lo <= q_hi the decoded planes, the differences to the origin and
// the products with 1/d are monotone, so min(t_lo, t_hi) IS the plane on the side the ray comes from:
// the caller selects the near/far plane words by the sign of 1/d once per node (6 selects for the four
// children) and the twelve v_min/v_max per child collapse into one v_max3 and one v_min3.  Both
// preconditions are verified per scene by the accel build, which sets exact_decode otherwise (LDEXP form).
template <int K, bool EXACT, bool LDEXP>
__device__ __forceinline__ float child_box(const uint32_t* pl, float px, float py, float pz, float sx, float sy, float sz,
                                           int ex, int ey, int ez, float rox, float roy, float roz, float rix, float riy, float riz) {
  float ax, ay, az, bx, by, bz;
  if (LDEXP) {
    ax = px + ldexpf(qbyte<K>(pl[0]), ex); ay = py + ldexpf(qbyte<K>(pl[1]), ey); az = pz + ldexpf(qbyte<K>(pl[2]), ez);
    bx = px + ldexpf(qbyte<K>(pl[3]), ex); by = py + ldexpf(qbyte<K>(pl[4]), ey); bz = pz + ldexpf(qbyte<K>(pl[5]), ez);
  } else {
    ax = __fmaf_rn(qbyte<K>(pl[0]), sx, px); ay = __fmaf_rn(qbyte<K>(pl[1]), sy, py); az = __fmaf_rn(qbyte<K>(pl[2]), sz, pz);
    bx = __fmaf_rn(qbyte<K>(pl[3]), sx, px); by = __fmaf_rn(qbyte<K>(pl[4]), sy, py); bz = __fmaf_rn(qbyte<K>(pl[5]), sz, pz);
  }
  const float tx1 = (ax - rox) * rix, tx2 = (bx - rox) * rix;
  const float ty1 = (ay - roy) * riy, ty2 = (by - roy) * riy;
  const float tz1 = (az - roz) * riz, tz2 = (bz - roz) * riz;
  if (EXACT || LDEXP) return slab_interval<EXACT>(tx1, tx2, ty1, ty2, tz1, tz2);
  const float tmin = fmaxf(fmaxf(tx1, ty1), tz1);   // pl[0..2] already hold the near planes
  const float tmax = fminf(fminf(tx2, ty2), tz2);
  return (tmax < tmin || tmax <= 0) ? RT_LARGE_FLOAT : tmin;
}

// Box tests of the <=4 children of an internal node.
template <bool EXACT, bool LDEXP>
__device__ __forceinline__ void eval_children(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, const uint32_t* __restrict__ ref_node,
                                              float rox, float roy, float roz, float rix, float riy, float riz,
                                              float hit_dist, Cand* c) {
  const float px = __uint_as_float(q0.x), py = __uint_as_float(q0.y), pz = __uint_as_float(q0.z);
  // plane scales 2^e as floats (fma decode); the ldexp decode takes the exponents from the reference node
  const float sx = __uint_as_float(q0.w), sy = __uint_as_float(q3.z), sz = __uint_as_float(q3.w);
  int ex = 0, ey = 0, ez = 0;
  if (LDEXP) {
    const uint32_t ew = ref_node[3];
    ex = (int)(int8_t)(ew & 0xff); ey = (int)(int8_t)((ew >> 8) & 0xff); ez = (int)(int8_t)((ew >> 16) & 0xff);
  }
  uint32_t pl[6] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
  if (!EXACT && !LDEXP) {
    const bool nx = rix < 0, ny = riy < 0, nz = riz < 0;
    pl[0] = nx ? q1.w : q1.x; pl[3] = nx ? q1.x : q1.w;
    pl[1] = ny ? q2.x : q1.y; pl[4] = ny ? q1.y : q2.x;
    pl[2] = nz ? q2.y : q1.z; pl[5] = nz ? q1.z : q2.y;
  }
  const uint32_t desc[4] = {q2.z, q2.w, q3.x, q3.y};   // complete work descriptors, DESC_NONE for an empty slot (:60)
  float d[4];
  d[0] = child_box<0, EXACT, LDEXP>(pl, px, py, pz, sx, sy, sz, ex, ey, ez, rox, roy, roz, rix, riy, riz);
  d[1] = child_box<1, EXACT, LDEXP>(pl, px, py, pz, sx, sy, sz, ex, ey, ez, rox, roy, roz, rix, riy, riz);
  d[2] = child_box<2, EXACT, LDEXP>(pl, px, py, pz, sx, sy, sz, ex, ey, ez, rox, roy, roz, rix, riy, riz);
  d[3] = child_box<3, EXACT, LDEXP>(pl, px, py, pz, sx, sy, sz, ex, ey, ez, rox, roy, roz, rix, riy, riz);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = (desc[k] != DESC_NONE) && (d[k] < hit_dist);     // :60, :71
    c[k].d = ok ? d[k] : __builtin_inff();
    c[k].desc = desc[k];
  }
}

// per-lane fetch counters of the STATS build (algorithmic bytes): what the
// reference logs in RT_mem_accesses (rt_traversal.cpp:54,116,148,158), without restart re-reads.
// A leaf or instance child still counts as one 52-byte node fetch: the reference reads that node.
struct Fetches { unsigned node = 0, inst = 0, tri = 0; };

#define ITER_LIMIT (1u << 22)   // backstop; accepted trees are acyclic (children stored after parents)

// ---------------------------------------------------------------------------------------------
// shading (closest.cpp:57-127 / miss.cpp:9-14)
// ---------------------------------------------------------------------------------------------
struct ShadeParams { float amb[3], lcol[3], lpos[3], bg[3]; uint32_t max_depth; };

__device__ __forceinline__ uint32_t f2u_x86(float f) { return (uint32_t)(long long)f; } // rtx_shading.h:7-8 as x86-64 g++ lowers it

// Occlusion ray of the shadow extension (no reference counterpart): from the hit point toward the
// light, origin pushed 1e-3 along L like the reference's mirror bounce (closest.cpp:104), tmax = |L|.
// I, L and dist are computed exactly as shade_eval computes them.
__device__ __forceinline__ void shadow_ray(float lpx, float lpy, float lpz, float ox, float oy, float oz, float dx, float dy, float dz,
                                           float hit_dist, float& sox, float& soy, float& soz, float& sdx, float& sdy, float& sdz, float& sdist) {
  const float Ix = ox + dx * hit_dist, Iy = oy + dy * hit_dist, Iz = oz + dz * hit_dist;
  float Lx = lpx - Ix, Ly = lpy - Iy, Lz = lpz - Iz;
  const float dist = sqrtf(Lx * Lx + Ly * Ly + Lz * Lz);
  const float il = 1.0f / dist;
  Lx *= il; Ly *= il; Lz *= il;
  sox = Ix + Lx * 0.001f; soy = Iy + Ly * 0.001f; soz = Iz + Lz * 0.001f;
  sdx = Lx; sdy = Ly; sdz = Lz;
  sdist = dist;
}

__device__ __forceinline__ void shadow_ray(const ShadeParams& p, float ox, float oy, float oz, float dx, float dy, float dz,
                                           float hit_dist, float& sox, float& soy, float& soz, float& sdx, float& sdy, float& sdz, float& sdist) {
  shadow_ray(p.lpos[0], p.lpos[1], p.lpos[2], ox, oy, oz, dx, dy, dz, hit_dist, sox, soy, soz, sdx, sdy, sdz, sdist);
}

// closest.cpp:57-90 for one hit: the non-reflected diffuse contribution `throughput * diffuse * (1 - reflectivity)`
// with throughput = 1 (:87), the reflectivity (:84), the hit point I and the shading normal N.
// occluded: result of the shadow extension (false = reference).
template <bool STATS = false>
__device__ void shade_terms(const SceneDev& sc, const ShadeParams& p, float ox, float oy, float oz,
                            float dx, float dy, float dz, const HitRec& hit, bool occluded,
                            float& r, float& g, float& b, float& refl_out,
                            float& Ix_o, float& Iy_o, float& Iz_o, float& Nx_o, float& Ny_o, float& Nz_o,
                            unsigned* textured = nullptr, float* albedo3 = nullptr) {
  const uint32_t* bp = sc.blas + (size_t)hit.blasIdx * (RT_BLAS_STRIDE / 4);
  const rt_triex_t te = sc.triEx[hit.triIdx];
  const rt_material_t* mat = sc.mat + te.texId;
  // I = orig + dir * dist (:61)
  const float Ix = ox + dx * hit.dist, Iy = oy + dy * hit.dist, Iz = oz + dz * hit.dist;
  // N = N1*bx + N2*by + N0*bz (:64)
  float Nx = te.N1[0] * hit.bx + te.N2[0] * hit.by + te.N0[0] * hit.bz;
  float Ny = te.N1[1] * hit.bx + te.N2[1] * hit.by + te.N0[1] * hit.bz;
  float Nz = te.N1[2] * hit.bx + te.N2[2] * hit.by + te.N0[2] * hit.bz;
  // transposed 3x3 of invTransform, TransformVector with w = 0 (:65-66, geometry.h:1141-1147,1280-1293)
  const float m0 = __uint_as_float(bp[1]), m1 = __uint_as_float(bp[2]), m2 = __uint_as_float(bp[3]);
  const float m4 = __uint_as_float(bp[5]), m5 = __uint_as_float(bp[6]), m6 = __uint_as_float(bp[7]);
  const float m8 = __uint_as_float(bp[9]), m9 = __uint_as_float(bp[10]), m10 = __uint_as_float(bp[11]);
  const float z0 = 0.0f * 0.0f;
  float Tx = m0 * Nx + m4 * Ny + m8 * Nz + z0;
  float Ty = m1 * Nx + m5 * Ny + m9 * Nz + z0;
  float Tz = m2 * Nx + m6 * Ny + m10 * Nz + z0;
  float inv = 1.0f / sqrtf(Tx * Tx + Ty * Ty + Tz * Tz);
  Nx = Tx * inv; Ny = Ty * inv; Nz = Tz * inv;
  // uv (:69)
  const float u = te.uv1[0] * hit.bx + te.uv2[0] * hit.by + te.uv0[0] * hit.bz;
  const float v = te.uv1[1] * hit.bx + te.uv2[1] * hit.by + te.uv0[1] * hit.bz;
  float cr, cg, cb;
  if (mat->diffuse_tex_id >= 0) {  // :72-77, texSample rtx_shading.h:5-18, RGB8toRGB32F common.h:156-162
    if (STATS) *textured += 1;
    const uint32_t tw = mat->tex_width, th = mat->tex_height;
    uint32_t iu = f2u_x86(u * (float)tw), iv = f2u_x86(v * (float)th);
    iu %= tw; iv %= th;
    const uint32_t texel = ((const uint32_t*)(sc.tex + mat->tex_offset))[iu + iv * tw];
    const float s = 1 / 256.0f;
    cr = (float)(int)((texel >> 16) & 255) * s;
    cg = (float)(int)((texel >> 8) & 255) * s;
    cb = (float)(int)(texel & 255) * s;
  } else {
    cr = mat->diffuse[0]; cg = mat->diffuse[1]; cb = mat->diffuse[2];
  }
  // diffuseLighting (rtx_shading.h:55-67)
  float Lx = p.lpos[0] - Ix, Ly = p.lpos[1] - Iy, Lz = p.lpos[2] - Iz;
  const float dist = sqrtf(Lx * Lx + Ly * Ly + Lz * Lz);
  const float il = 1.0f / dist;
  Lx *= il; Ly *= il; Lz *= il;
  const float att = 1.0f / (1.0f + dist * 0.1f);
  float NdotL = std_max(0.0f, Nx * Lx + Ny * Ly + Nz * Lz);
  if (occluded) NdotL = 0.0f;   // shadow extension: occluded -> no direct term
  const float dr = cr * (p.amb[0] + att * p.lcol[0] * NdotL);
  const float dg = cg * (p.amb[1] + att * p.lcol[1] * NdotL);
  const float db = cb * (p.amb[2] + att * p.lcol[2] * NdotL);
  const float refl = __uint_as_float(bp[38]);   // blas_node_t::reflectivity @152
  const float thr = 1.0f;
  r = 0.0f + thr * dr * (1 - refl);             // :87
  g = 0.0f + thr * dg * (1 - refl);
  b = 0.0f + thr * db * (1 - refl);
  refl_out = refl;
  Ix_o = Ix; Iy_o = Iy; Iz_o = Iz; Nx_o = Nx; Ny_o = Ny; Nz_o = Nz;
  if (albedo3) { albedo3[0] = cr; albedo3[1] = cg; albedo3[2] = cb; }   // texColor (:72-77)
}

// closest.cpp:57-127 without a secondary ray (reflectivity <= 0 or bounce + 1 >= max_depth) / miss.cpp:9-14
template <bool STATS = false>
__device__ void shade_eval(const SceneDev& sc, const ShadeParams& p, float ox, float oy, float oz,
                           float dx, float dy, float dz, const HitRec& hit, bool found, bool occluded,
                           float& r, float& g, float& b, unsigned* textured = nullptr) {
  if (!found) { r = p.bg[0]; g = p.bg[1]; b = p.bg[2]; return; }
  float refl, Ix, Iy, Iz, Nx, Ny, Nz;
  shade_terms<STATS>(sc, p, ox, oy, oz, dx, dy, dz, hit, occluded, r, g, b, refl, Ix, Iy, Iz, Nx, Ny, Nz, textured);
  float thr = 1.0f;
  thr *= refl;                                  // :90
  r = r + p.bg[0] * thr;                        // :123
  g = g + p.bg[1] * thr;
  b = b + p.bg[2] * thr;
}

// closest.cpp:96-99: the mirror ray leaving a hit.  R = normalize(dir - 2.0f * N * dot(N, dir)), origin I + R * 0.001f
__device__ __forceinline__ void mirror_ray(float dx, float dy, float dz, float Ix, float Iy, float Iz, float Nx, float Ny, float Nz,
                                           float* out6) {
  const float nd = Nx * dx + Ny * dy + Nz * dz;
  const float vx = dx - (2.0f * Nx) * nd, vy = dy - (2.0f * Ny) * nd, vz = dz - (2.0f * Nz) * nd;
  const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  const float Rx = vx * inv, Ry = vy * inv, Rz = vz * inv;
  out6[0] = Ix + Rx * 0.001f; out6[1] = Iy + Ry * 0.001f; out6[2] = Iz + Rz * 0.001f;
  out6[3] = Rx; out6[4] = Ry; out6[5] = Rz;
}

__device__ __forceinline__ uint32_t pack_rgb8(float r, float g, float b) {  // common.h:149-154
  int ir = (int)(std_min(r, 1.f) * 255);
  int ig = (int)(std_min(g, 1.f) * 255);
  int ib = (int)(std_min(b, 1.f) * 255);
  return (uint32_t)((ir << 16) + (ig << 8) + ib);
}

// kernel.cpp:28-39.  u = (x*2.0 - W)/H and v = (y*2.0 - H)/H are evaluated in double and rounded to
// f32 there; they depend on x (resp. y) only, so the host evaluates exactly that expression once per
// column / row (IEEE double division is correctly rounded on both sides) and the kernels read the
// two small tables instead of running an f64 divide per ray.
__device__ __forceinline__ void generate_ray(float u, float v,
                                             float& ox, float& oy, float& oz, float& dx, float& dy, float& dz) {
  // front=(1,0,0); right=cross(front,(0,1,0))=(0,0,1); up=cross(right,front)=(0,1,0)
  const float rx = 0.0f * 0.0f - 0.0f * 1.0f, ry = 0.0f * 0.0f - 1.0f * 0.0f, rz = 1.0f * 1.0f - 0.0f * 0.0f;
  const float ux = ry * 0.0f - rz * 0.0f, uy = rz * 1.0f - rx * 0.0f, uz = rx * 0.0f - ry * 1.0f;
  const float FOV = 1.0f;
  float vx = u * rx + v * ux + FOV * 1.0f;
  float vy = u * ry + v * uy + FOV * 0.0f;
  float vz = u * rz + v * uz + FOV * 0.0f;
  const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  ox = 0.0f; oy = 100.0f; oz = 0.0f;
  dx = vx * inv; dy = vy * inv; dz = vz * inv;
}

__device__ __forceinline__ uint32_t wang_hash(uint32_t s) {   // common.h:129-135
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u; s = s ^ (s >> 4);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15);
  return s;
}
__device__ __forceinline__ float random_float(uint32_t& s) {   // common.h:137-147
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return (float)s * 2.3283064365387e-10f;
}

// The occlusion / bounce ray of sample `smp` of pixel (x, y) leaving the hit point I with shading normal N (view direction vd), as
// oracle/rt_oracle.c:orc_ao_ray defines it, operation by operation: o = I + N' * 1e-3, d = cosine-weighted about the normal N'
// that faces the viewer (rejection-sampled disk, Duff et al. basis; only IEEE add / mul / div / sqrt).
__device__ __forceinline__ void ao_sample_ray(uint32_t x, uint32_t y, uint32_t W, uint32_t spp, uint32_t smp, uint32_t user_seed,
                                              float Ix, float Iy, float Iz, float nx, float ny, float nz, float vdx, float vdy, float vdz, float* o6) {
  uint32_t seed = wang_hash((x + y * W) * spp + smp + 1u + user_seed * 0x9E3779B9u);
  if (seed == 0u) seed = 1u;
  float u = 0.0f, v = 0.0f, r2 = 0.0f;
  bool ok = false;
  for (int k = 0; k < 8 && !ok; ++k) {
    const float a = 2.0f * random_float(seed) - 1.0f;
    const float b = 2.0f * random_float(seed) - 1.0f;
    const float q = a * a + b * b;
    if (q < 1.0f) { u = a; v = b; r2 = q; ok = true; }
  }
  const float z = sqrtf(1.0f - r2);
  if (nx * vdx + ny * vdy + nz * vdz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
  const float sign = nz >= 0.0f ? 1.0f : -1.0f;
  const float a = -1.0f / (sign + nz);
  const float b = nx * ny * a;
  const float tx = 1.0f + sign * nx * nx * a, ty = sign * b, tz = -sign * nx;
  const float bx = b, by = sign + ny * ny * a, bz = -ny;
  o6[0] = Ix + nx * 0.001f; o6[1] = Iy + ny * 0.001f; o6[2] = Iz + nz * 0.001f;
  o6[3] = tx * u + bx * v + nx * z;
  o6[4] = ty * u + by * v + ny * z;
  o6[5] = tz * u + bz * v + nz * z;
}

#ifndef RT_TRI_PREFETCH
#define RT_TRI_PREFETCH 1
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 6
#endif

// ---------------------------------------------------------------------------------------------
// Persistent traversal kernel.
//
// A fixed grid of wavefronts pulls jobs (pixels of 8x8 tiles, or rays of a ray buffer) from a
// sharded queue, so the launch ends within one job batch of the last job instead of within the
// slowest tile of a static tile->wavefront map (per-tile clocks showed half-empty CUs for the
// second half of a frame).  Idle lanes take new jobs once RT_DEAD_MAX (rendering) / RT_TRACE_DEAD_MAX
// (ray buffers) lanes of the wavefront are not traversing: 64 = whole tiles for rendering, because
// coherent camera rays lose more from sharing a wavefront with another tile than they gain from refilled
// lanes; 16 for incoherent ray buffers.  A tile's lanes trace the primary ray, then (shadow jobs) the
// occlusion ray of the same pixel; any-hit is a per-lane flag.
// Rendering is deferred: this kernel leaves 24-byte hit records, rt_shade_kernel makes pixels.
// Per-ray semantics -- and therefore results -- do not depend on the schedule.
// ---------------------------------------------------------------------------------------------
#ifndef RT_DEAD_MAX
#define RT_DEAD_MAX 64      // render jobs: leave the traversal loop (finish rays, fetch jobs) once this many lanes
#endif                      // are not traversing (finished or idle); 64 = whole-tile batches
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 1          // render jobs: lanes of a tile reach their leaves together anyway
#endif
#ifndef RT_TRACE_LEAF_MIN
#define RT_TRACE_LEAF_MIN 24   // incoherent rays: +3.5 % at 16, another 1 % at 24
#endif
#ifndef RT_TRI_LDS
#define RT_TRI_LDS 0        // > 0: triangles of the leaf most lanes of the wavefront hold are staged through LDS (north_star: "triangle vertices staged
#endif                      // through LDS"): leaves of up to RT_TRI_LDS triangles, when at least RT_TRI_LDS_MIN lanes hold the same one.  Measured, off: DESIGN.md s5
#ifndef RT_TRI_LDS_MIN
#define RT_TRI_LDS_MIN 8
#endif
#ifndef RT_OCCLUSION_ORDER
#define RT_OCCLUSION_ORDER 0      // 0: slot order (the default); 1 / 2: measurement variants (profiles/r05_g_gpu_reinsertion.txt, section 7)
#endif
#ifndef RT_OCCLUSION_SLOT_ORDER
#define RT_OCCLUSION_SLOT_ORDER 0
#endif
#ifndef RT_UNORDERED_OCCLUSION
#define RT_UNORDERED_OCCLUSION 1
#endif
#ifndef RT_SHADOW_FINISH_MIN
#define RT_SHADOW_FINISH_MIN 65   // > 64: off
#endif
#ifndef RT_TRACE_DEAD_MAX
#define RT_TRACE_DEAD_MAX 16   // ray-buffer jobs (incoherent rays): refill early
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 64         // jobs reserved per global atomic (one 8x8 tile)
#endif
#ifndef RT_TRACE_NT
#define RT_TRACE_NT 0       // ray buffers: rays loaded and hit records stored with the streaming hint
#endif
#ifndef RT_TRACE_CHUNK
#define RT_TRACE_CHUNK 64   // ray buffers: jobs reserved per global atomic (rays have no screen neighbours to keep together)
#endif
#ifndef RT_XCC_HOME
#define RT_XCC_HOME 1         // a wavefront's home queue shard is its physical XCD (0 = blockIdx % 8, which names a group of blocks that share an XCD, not the XCD)
#endif
#ifndef RT_QUEUE_DRY_MASK
#define RT_QUEUE_DRY_MASK 1   // shards a wavefront found handed out are not polled again by the other wavefronts of its workgroup (LDS mask)
#endif
#ifndef RT_STEAL_SPREAD
#define RT_STEAL_SPREAD 0     // order in which a wavefront visits the other shards once its home shard is handed out: +1, +2, ... (0) or bit-reversed distance (1:
                              // the helpers of a drained band spread over the remaining ones; measured -1.4 % serial, +-0 elsewhere: profiles/r04_m_steal_spread_ab.txt)
#endif
#ifndef QUEUE_SHARDS
#define QUEUE_SHARDS 8u     // one device-scope counter saturates near 90 dequeues/us
#endif
#define QUEUE_STRIDE 32u    // one 128-byte line per shard counter
// per-frame control block: [0] deferral count (own 128-byte line), then the queue counters of the main
// launch, of the EXACT launch over the deferred list and of the a-priori EXACT launch
#define CTL_QUEUE_DWORDS (QUEUE_SHARDS * QUEUE_STRIDE)
#define CTL_DWORDS (32u + 3u * CTL_QUEUE_DWORDS)
// Frames (camera tiles + their occlusion rays): 7 wavefronts per SIMD (72 VGPRs, 6 stack levels in LDS).  With frames traced in
// batches -- many tiles per wavefront, so ramp and tail of a launch no longer decide -- occupancy pays: 7 / 8 wavefronts are +5.3 /
// +5.6 % on the headline frame (one frame per launch: +-1 %, measured in round 2), 8 loses 3 % on serial frames, 7 gains 2 %
// there.  Ray buffers (JOB_TRACE): 7 wavefronts with 7 LDS levels -- they need one context slot less, which pays for the seventh
// level -- +2.5 % on random rays, hairball AO and diffuse bounce unchanged (with 6 levels AO lost 2 %): profiles/r02_o_occupancy.txt.
#ifndef RT_WAVES_TRACE
#define RT_WAVES_TRACE 7
#endif
#ifndef RT_LDS_STACK_TRACE
#define RT_LDS_STACK_TRACE 7
#endif
#ifndef RT_WAVES_RENDER
#define RT_WAVES_RENDER 7
#endif
#ifndef RT_LDS_STACK_RENDER
#define RT_LDS_STACK_RENDER 6
#endif
#ifndef RT_WAVES_RENDER_PACKED
#define RT_WAVES_RENDER_PACKED 8
#endif
#ifndef RT_LDS_STACK_RENDER_PACKED
#define RT_LDS_STACK_RENDER_PACKED 5
#endif
#ifndef LDS_STACK
#define LDS_STACK 8         // stack levels kept in LDS per lane (4 KiB per wavefront); deeper ones go to scratch
#endif
#ifndef RT_WG_WAVES
#define RT_WG_WAVES 4       // wavefronts per workgroup of the persistent kernels (the staged top of the tree is shared by them)
#endif
#define RT_WG_THREADS (64 * RT_WG_WAVES)
#ifndef RT_SHALLOW_LEVELS
#define RT_SHALLOW_LEVELS 16  // internal levels (TLAS + BLAS) up to which a scene takes the SHALLOW instantiations: 48 stack entries instead of 96 + the LDS levels
#endif
#ifndef RT_TOP_NODES
#define RT_TOP_NODES 0      // internal nodes of the top of the tree staged in LDS per workgroup (64 B each); 0 = off
#endif
#define RT_TOP_MAX 1024
static_assert(RT_TOP_NODES <= RT_TOP_MAX, "top-of-tree image");

// JOB_TRACE_UNORDERED: a ray buffer of any-hit rays whose caller only wants "blocked or not" (ambient occlusion, the occlusion rays of a
// bounce level): JOB_TRACE's kernel with the children visited in slot order, as the frame's occlusion rays are (no sorting by distance,
// no path maxima).  vxrt_trace's VXRT_MODE_ANY returns the reference's FIRST accepted candidate and keeps JOB_TRACE.
enum { JOB_RENDER = 0, JOB_RENDER_SHADOW = 1, JOB_TRACE = 2, JOB_RENDER_GI = 3, JOB_TRACE_UNORDERED = 4 };
__host__ __device__ constexpr bool is_trace_job(int job) { return job == JOB_TRACE || job == JOB_TRACE_UNORDERED; }
// JOB_RENDER_GI: the whole "one diffuse bounce" frame (BASELINE configs[2] as worded; recipe: oracle/rt_oracle.c:orc_render_gi) in ONE
// persistent launch -- a lane traces its pixel's primary ray, shades the hit (closest.cpp's else arm), draws the pixel's bounce ray
// (ao_sample_ray, sample 0 of 1), traces it for its closest hit in the same lane, shades that hit and writes the pixel:
// colour = Lambert(primary) + albedo(primary) * Lambert(bounce hit | background).  Before, the frame was a primary launch, a pass that
// listed the hit pixels, a ray-generation pass, a 2 M-ray trace launch (a short launch with a long tail: 0.61 of the frame's 1.17 ms),
// an accumulation pass and a final pass.
#ifndef RT_WAVES_GI
#define RT_WAVES_GI 7
#endif
#ifndef RT_LDS_STACK_GI
#define RT_LDS_STACK_GI 6
#endif
#ifndef RT_GI_DEAD_MAX
#define RT_GI_DEAD_MAX 64    // (lanes refilled one by one: 16 -> 1.50 ms, 32 -> 1.45, 8 -> 1.72 against 1.18 with whole tiles: profiles/r03_f_gi_fused_ab.txt)
#endif

// -DRT_ISA_MARKS: comment-only markers in the listing (hipcc -S) that delimit the regions of the traversal loop for
// tools/isa_regions.py; never set for a build that is run
#ifdef RT_ISA_MARKS
#define RT_MARK(name) asm volatile("; RTMARK " name)
#else
#define RT_MARK(name)
#endif

// Hit records of a frame window are kept TILE-MAJOR between the traversal and the shading pass: record of pixel (x, y) =
// tile * 64 + lane of the 8x8 tile grid that starts at row y0, i.e. the job id of the traversal kernel.  A wavefront
// therefore writes the 64 records of its tile as one contiguous, 128-byte aligned 1,536-byte block, once (the occlusion
// result is folded into bit 31 of blasIdx before the record is written): no cache line is shared between wavefronts, so no
// XCD writes a partial line back (round 1 wrote pixel-major records + an atomicOr per occluded pixel: 121 MB of HBM
// writes per 1080p frame for 50 MB of records, profiles/r01_k_pmc.txt).
// `lr` = local row of the window: rows are counted through the window's tile rows (8 each) in order.
__device__ __forceinline__ size_t hit_index(uint32_t x, uint32_t lr, uint32_t tiles_x) {
  return ((size_t)(lr >> 3) * tiles_x + (x >> 3)) * 64u + ((lr & 7u) << 3) + (x & 7u);
}
// frame row of local row lr: the window's k-th tile row is frame rows y0 + k * row_step ... + 7 (row_step = 8 for a contiguous
// window, 8 * stride for the interleaved tile rows of vxrt_render_interleaved)
__device__ __forceinline__ uint32_t frame_row(uint32_t lr, uint32_t y0, uint32_t row_step) { return y0 + (lr >> 3) * row_step + (lr & 7u); }

// Division of a tile index (< 2^25: render_common refuses larger launches) by a divisor that is the same for the whole launch -- tiles per row,
// tiles per frame of a batch -- as a multiply-high, an add and a shift with constants the host derives once (Granlund & Montgomery 1994:
// L = ceil(log2 d), m = floor(2^32 (2^L - d) / d) + 1, x / d = (mulhi(x, m) + x) >> L; the sum cannot overflow for x < 2^31).  The compiler's
// own expansion of x / d for a runtime d is a float reciprocal + two correction steps, ~20 VALU instructions per division and a hoisted
// reciprocal per divisor held in a VGPR for the whole kernel; a lane derives its pixel from its job three times per pixel.
struct FastDiv { uint32_t d, m, s; };
static FastDiv fast_div_make(uint32_t d) {
  FastDiv f{d ? d : 1u, 1u, 0u};
  while ((1ull << f.s) < f.d) ++f.s;
  f.m = (uint32_t)((((1ull << f.s) - f.d) << 32) / f.d) + 1u;
  return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t x, const FastDiv& f) { return (__umulhi(x, f.m) + x) >> f.s; }

struct PersistArgs {
  uint32_t W, H, y0, y1, tiles_x;
  FastDiv div_tiles_x, div_frame_tiles;   // (tiles_x, frame_tiles as divisors: see FastDiv)
  uint32_t row_step;              // frame rows between two consecutive tile rows of the window: 8, or 8 * stride (interleaved)
  uint32_t total;                 // number of jobs (tiles*64 pixels, or rays); an upper bound when total_dev is set
  const uint32_t* total_dev;      // optional: the job count lives in device memory (produced by an earlier kernel of the stream)
  HitRec* hits;                   // render: W*H hit records (occlusion in bit 31 of blasIdx); trace: n records
  const float* rays; const float* tmax; int any_hit;   // trace inputs
  const uint32_t* order;          // trace, optional: queue position -> ray id (rays binned by origin cell and direction octant)
  unsigned long long* counters;   // [0] rays (+ STATS: [1..4])
  uint32_t* status;
  uint32_t* queue;                // QUEUE_SHARDS counters (QUEUE_STRIDE dwords apart), zeroed by the host before the launch
  uint32_t per_shard;             // jobs per shard (multiple of 64)
  const float* utab; const float* vtab;   // camera u per column, v per row (see generate_ray)
  // rays whose slab products can be NaN (a zero / non-finite direction component) are not traced by
  // the main launch: their job id (bit 31 = occlusion phase) is appended here and a second, small
  // launch of the EXACT variant (libstdc++ min/max forms) traces them
  uint32_t* defer_count; uint32_t* defer_list; uint32_t defer_cap;
  // render jobs, optional: longest-processing-time-first order learned from the previous frame of this context
  // (tile_order[queue position] = tile, sorted by cost within each shard's band) and where this frame's cost goes
  const uint32_t* tile_order; uint32_t* tile_cost;
  uint32_t shard_rot;             // diagnostic (VXRT_SHARD_ROT): home shard of block b = (b + shard_rot) % QUEUE_SHARDS
  unsigned long long* wave_log;   // STATS only, optional: 16 u64 per wavefront (see vxrt_render_wave_log in the header)
  // optional, every build (the TIMED kernels too: one store per wavefront when it ends, nothing inside the loop): 2 u64 per wavefront of the main
  // launch -- [0] the constant 100 MHz clock at its end, [1] rays it started | physical XCD << 56 (vxrt_debug_end_log; tools/xcd_tail.py)
  unsigned long long* end_log;
  // batch of frames in one launch (vxrt_render_interleaved_batch): the window's tiles repeat `frame_tiles` apart, frame f = tile /
  // frame_tiles is shaded and lit with pbatch[f]; nullptr = one frame
  const ShadeParams* pbatch; uint32_t frame_tiles;
  // JOB_RENDER_GI: the frame itself (pixel (x, y) at dst[x + y * W]), optional f32 colours, seed of the bounce rays
  uint32_t* dst; float* colors; uint32_t gi_seed;
};

// Domain of the fast (non-EXACT) traversal: every component of 1/d finite, non-zero and at most 2^64 in magnitude, every origin
// component at most 2^60.  The accel build holds node planes to 2^60 as well (else the scene runs the LDEXP instantiation), so
// a slab value (plane - o) * (1/d) stays below 2^125: finite, never NaN.  Rays outside go to the EXACT launch, which evaluates
// the reference's min/max chains literally.  (Comparisons with NaN are false, so NaN / inf components fail these tests.)
#define RT_FAST_INV_MAX 0x1p+64f
#define RT_FAST_POS_MAX 0x1p+60f
__device__ __forceinline__ bool ray_in_fast_domain(float ox, float oy, float oz, float ix, float iy, float iz) {
  return fabsf(ix) <= RT_FAST_INV_MAX && fabsf(iy) <= RT_FAST_INV_MAX && fabsf(iz) <= RT_FAST_INV_MAX && ix != 0.0f && iy != 0.0f && iz != 0.0f &&
         fabsf(ox) <= RT_FAST_POS_MAX && fabsf(oy) <= RT_FAST_POS_MAX && fabsf(oz) <= RT_FAST_POS_MAX;
}

// max of two values neither of which is a NaN: one v_max_f32 (fmaxf adds a canonicalising v_max_f32 x, x for a signalling NaN the
// compiler cannot rule out)
__device__ __forceinline__ float vmax_nonan(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ bool is_node_desc(uint32_t d) { return d < 0x80000000u; }
__device__ __forceinline__ bool is_leaf_desc(uint32_t d) { return (d >> 30) == DK_LEAF; }
__device__ __forceinline__ bool is_inst_desc(uint32_t d) { return d >= 0xC0000000u && d < DESC_IDLE; }
__device__ __forceinline__ bool is_work_desc(uint32_t d) { return d < DESC_IDLE; }

// per-lane flag bits
#define F_FOUND 1u
#define F_ANYHIT 2u
#define F_SHADOW 8u      // render job is in its occlusion-ray phase
#define F_WORLD 16u      // the active ray registers hold the world-space ray (TLAS level)
#define F_RESUMED 32u    // EXACT launch: occlusion ray handed over by the main launch (its primary hit record is in memory)

// Register budget is the lever here (profiles/r01_c_*: at 4 waves/SIMD the VALU pipe idles 58 % of
// the time waiting on dependent loads), so a lane keeps in VGPRs only what every step touches: the
// ACTIVE ray (origin + reciprocal direction; world space at TLAS level, object space inside an
// instance), hit distance, path_m, the current work item and the register-cached stack top.  The ray
// direction (triangle tests only), barycentrics / indices of the best hit and blasIdx live in LDS
// next to the stack; the world ray is not stored at all - it is re-derived from the job (ray buffer,
// camera tables, or the pixel's primary hit record) on the rare TLAS-level steps.
// STATS: 0 = the timed kernel; 1 = counting build in the reference's order (ordered occlusion, no leaf helpers: its fetch counts equal
// the canonical restatement's); 2 = counting build of the traversal the timed kernel actually performs (unordered occlusion, helpers)
// PACKED: the instantiation for frames traced in sets that overlap on two streams (bench.py's pipelined mode, batches): 8 wavefronts
// per SIMD (64 VGPRs, 5 stack levels in LDS) instead of 7 -- +1.6 % there, where many tiles per wavefront hide the few spilled
// registers, and -8 % on a serial frame, which keeps 7 (profiles/r03_c_flag_variants.txt)
// SHALLOW: the scene's trees are at most RT_SHALLOW_LEVELS internal levels deep on any root-to-leaf path, TLAS and BLAS together (measured by the
// accel build, accel_depth_kernel), so a lane's stack never holds more than 3 x RT_SHALLOW_LEVELS entries and the part of it that lives in
// scratch is sized for that instead of for the reference's 32 levels: 344 instead of 768 bytes per lane for the 8-wavefront instantiation.
// (The 1,048,576-triangle atrium is 13 levels deep, the 10 M-triangle hairball 15.)  Timed builds only; deeper scenes take the full-size form.
template <int JOB, int STATS, bool LDEXP, bool EXACT, bool PACKED = false, bool SHALLOW = false>
__global__ __launch_bounds__(EXACT ? 256 : RT_WG_THREADS, EXACT ? 4 : (is_trace_job(JOB) ? RT_WAVES_TRACE : (JOB == JOB_RENDER_GI ? RT_WAVES_GI : (PACKED ? RT_WAVES_RENDER_PACKED : RT_WAVES_RENDER)))) void rt_persistent_kernel(SceneDev sc, ShadeParams p, PersistArgs A) {
  // stack levels in LDS: what the instantiation's occupancy leaves room for (160 KB per CU)
  constexpr int LSTK = EXACT ? LDS_STACK : (is_trace_job(JOB) ? RT_LDS_STACK_TRACE : (JOB == JOB_RENDER_GI ? RT_LDS_STACK_GI : (PACKED ? RT_LDS_STACK_RENDER_PACKED : RT_LDS_STACK_RENDER)));
  constexpr int WG_WAVES = EXACT ? 4 : RT_WG_WAVES;
  constexpr bool USE_TOP = RT_TOP_NODES > 0 && !EXACT && !LDEXP;   // (the ldexp decode reads exponents from the reference node by index)
  constexpr uint32_t DEAD_MAX = is_trace_job(JOB) ? RT_TRACE_DEAD_MAX : (JOB == JOB_RENDER_GI ? RT_GI_DEAD_MAX : RT_DEAD_MAX);
  // render-with-shadow jobs: retire finished primary rays (their lanes continue with the occlusion ray
  // of the same pixel - same traversal code, so no phase mixing) before the whole tile is done
  constexpr uint32_t FINISH_MIN = JOB == JOB_RENDER_SHADOW ? RT_SHADOW_FINISH_MIN : 65u;
  const uint32_t lane = threadIdx.x & 63u;
  // EXACT launch: the jobs are the entries of the deferral list the main launch left behind
  const uint32_t n_jobs = EXACT ? min(*A.defer_count, A.defer_cap) : (A.total_dev ? min(*A.total_dev, A.total) : A.total);
  const uint32_t per_shard = (EXACT || A.total_dev) ? (((n_jobs + QUEUE_SHARDS - 1) / QUEUE_SHARDS + 63u) & ~63u) : A.per_shard;

  __shared__ uint2 s_stk[WG_WAVES][LSTK][64];   // stack levels below the register top, 8 B entries, conflict-free rows
  // 0-2 active dir, 3-4 hit bx/by (bz = 1 - bx - by is re-derived when the record is written), 5 distance of the pixel's
  // primary hit while its occlusion ray is traced, 6 hit blasIdx, 7 hit triIdx, 8 blasIdx
  // (ray buffers have no "primary hit kept while the occlusion ray runs": slot 5 is dropped there, 8 slots + 7 stack levels fit 7 workgroups per CU)
  constexpr int NCTX = (!EXACT && is_trace_job(JOB)) ? 8 : 9;
  __shared__ uint32_t s_ctx[WG_WAVES][NCTX][64];
  __shared__ uint32_t s_dry;      // bit s: a wavefront of this workgroup found queue shard s handed out
  if (threadIdx.x == 0) s_dry = 0u;
  __syncthreads();
  uint2* const lstk = &s_stk[threadIdx.x >> 6][0][lane];
  uint32_t* const ctx = &s_ctx[threadIdx.x >> 6][0][lane];
#define CTX(i) ctx[((NCTX == 8 && (i) > 5) ? (i) - 1 : (i)) * 64]
  // top of the tree staged in LDS (north_star: "BVH nodes staged through LDS"): the first levels are what every ray of every
  // tile walks, and a ds_read_b128 does not queue behind the CU's vector-memory pipeline
  __shared__ uint4 s_top[USE_TOP ? 4 : 1][USE_TOP ? RT_TOP_NODES : 1];
  constexpr bool TRI_LDS = RT_TRI_LDS > 0 && !EXACT && !is_trace_job(JOB);
  __shared__ float4 s_tri[TRI_LDS ? WG_WAVES : 1][TRI_LDS ? 3 * RT_TRI_LDS : 1];
  const uint32_t n_top = USE_TOP ? min(sc.n_top, (uint32_t)RT_TOP_NODES) : 0u;
  if (USE_TOP && n_top) {
    for (uint32_t i = threadIdx.x; i < 4u * n_top; i += (uint32_t)(64 * WG_WAVES)) s_top[i / n_top][i % n_top] = sc.top_img[(size_t)(i / n_top) * RT_TOP_NODES + (i % n_top)];
    __syncthreads();
  }
  const uint32_t root_desc = (USE_TOP && n_top) ? sc.tlas_root_top : sc.tlas_root;
  const uint32_t* const blas_roots = (USE_TOP && n_top) ? sc.blas_root_top : sc.blas_root;
  // single-instance scenes: the BLAS root every ray starts at, fetched once per wavefront (a scalar) instead of once per ray -- a dependent
  // load less on the way from a job to its first node step
  const uint32_t root_blas_desc = is_inst_desc(root_desc) ? blas_roots[root_desc & PAYLOAD_MASK] : DESC_DONE;

  // ---- per-lane ray state in registers ----
  float arx = 0, ary = 0, arz = 0, aix = 0, aiy = 0, aiz = 0;   // active ray: origin, 1/direction
  float hitd = 0, path_m = 0, tos_m = 0;
  uint32_t cur = DESC_IDLE, tos_d = DESC_DONE, job = 0, flags = 0;
  // JOB_RENDER_GI: colour and albedo of the pixel's primary hit and the pixel's bounce ray (world space), kept while that ray is traced
  float g_col[3] = {0.f, 0.f, 0.f}, g_alb[3] = {0.f, 0.f, 0.f}, g_ray[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int sp = 0;                     // entries below the register top (LDS, then scratch)
  // entries past the LDS levels (scratch): the whole stack holds STACK_CAP entries below the register top
  constexpr int STACK_CAP = SHALLOW ? 3 * RT_SHALLOW_LEVELS : LSTK + RT_STACK_ENTRIES;
  static_assert(STACK_CAP > LSTK, "stack");
  uint32_t ovf_d[STACK_CAP - LSTK];
  float ovf_m[STACK_CAP - LSTK];
  // wave-uniform job-queue state
  bool queue_empty = false;
  // Home shard = the PHYSICAL XCD the wavefront runs on (HW_REG_XCC_ID, 0..7), so that band s of the frame is traced by the same XCD in every
  // launch and finds its part of the BVH in that XCD's L2 from the frame before.  Rounds 1-3 took blockIdx % 8: blocks b and b + 8 do share an
  // XCD, but WHICH one block 0 lands on changes from launch to launch (per-wavefront logs: the group of XCDs a band's wavefronts run on moves
  // by four between consecutive launches, profiles/r04_l_xcd.txt), so an XCD met another band's working set at every launch.  Speed only:
  // any wavefront may take any shard's jobs.  (shard_rot: diagnostic rotation of the XCD -> band map.)
  const uint32_t xcc_id = RT_XCC_HOME ? (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) : blockIdx.x;   // XCC_ID[3:0]
  const uint32_t shard = (xcc_id + A.shard_rot) % QUEUE_SHARDS;
  uint32_t tries = 0, loc_next = 0, loc_end = 0;
  uint32_t loc_off = 0;           // job id = queue position + loc_off (tile order indirection of render jobs)
  // tile whose cost is being taken for A.tile_cost.  The cost is WORK, not time: loop iterations of the wavefront while it held the tile (a leaf-body
  // run counts three).  Round 3 took the 100 MHz clocks the tile occupied its wavefront -- but a tile's duration says when it ran, not what it
  // is: tiles started as the queue runs dry take 2.5x as long as the same tiles earlier in a launch, the duration of a tile in one set of frames
  // correlates at -0.1 .. -0.2 with its duration in the set before, and an order learned from it puts last set's late (= "slow") tiles first and
  // the truly expensive ones last, every other set (profiles/r04_f_tile_tail.txt).  Iterations are a property of the tile alone.
  uint32_t lpt_tile = 0xFFFFFFFFu, lpt_work = 0; unsigned long long lpt_t0 = 0;
  Fetches fx;
  unsigned nrays = 0, nhit = 0;
  unsigned long long t_first = 0;
  unsigned long long wl_tn = 0, wl_tl = 0, wl_t0 = 0;   // wave_log: shader clocks inside the node body / the leaf body
  unsigned wl_no23 = 0, wl_no3 = 0, wl_iter = 0, wl_node_x = 0, wl_node_l = 0, wl_leaf_x = 0, wl_leaf_l = 0;   // wave_log: lane occupancy of the two bodies
  unsigned long long wl_tstart = 0, wl_tf = 0, wl_tfin = 0, wl_tmark = 0;   // wave_log: shader clocks in the fetch / finish sections
  unsigned long long wl_tq = 0;   // wave_log: 100 MHz clock at which this wavefront found every queue shard empty
  if (STATS && A.wave_log) { t_first = wall_clock64(); wl_tstart = __builtin_readcyclecounter(); }

  auto pixel_of = [&](uint32_t r, uint32_t& x, uint32_t& y) {
    // (the job id goes through an empty asm statement: the compiler then derives the pixel again wherever it is asked for -- a dozen
    // instructions -- instead of keeping x, y and the table addresses of every lane alive, in scratch, from the start of a ray to its end)
    asm volatile("" : "+v"(r));
    uint32_t tile = r >> 6;
    const uint32_t l = r & 63u;
    if (A.pbatch) tile -= fast_div(tile, A.div_frame_tiles) * A.frame_tiles;   // (a batch of frames: same window, frame_tiles tiles apart)
    const uint32_t ty = fast_div(tile, A.div_tiles_x);
    x = (tile - ty * A.tiles_x) * 8u + (l & 7u);
    y = A.y0 + ty * A.row_step + (l >> 3);
  };
  // where the lane's hit record goes (same reason for the empty asm statement as in pixel_of: the address is formed where it is used -- one
  // multiply-add -- instead of living in two registers, or two scratch slots, for the length of the ray)
  auto hit_slot = [&]() -> HitRec* {
    uint32_t j = job;
    asm volatile("" : "+v"(j));
    return A.hits + j;
  };
  // the lane's world-space ray, re-derived from its job (deterministic: same bits every time)
  auto world_ray = [&](float& ox, float& oy, float& oz, float& dx, float& dy, float& dz, float& tmax_) {
    tmax_ = RT_LARGE_FLOAT;
    if (is_trace_job(JOB)) {
      const float* rp = A.rays + (size_t)job * 6;
      if (RT_TRACE_NT) {   // (a ray is read once, by one lane: streamed past the caches that hold the tree)
        ox = __builtin_nontemporal_load(rp); oy = __builtin_nontemporal_load(rp + 1); oz = __builtin_nontemporal_load(rp + 2);
        dx = __builtin_nontemporal_load(rp + 3); dy = __builtin_nontemporal_load(rp + 4); dz = __builtin_nontemporal_load(rp + 5);
      } else { ox = rp[0]; oy = rp[1]; oz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5]; }
      if (A.tmax) tmax_ = A.tmax[job];
    } else {
      uint32_t x, y;
      pixel_of(job, x, y);
      generate_ray(A.utab[x], A.vtab[y], ox, oy, oz, dx, dy, dz);
      if (JOB == JOB_RENDER_GI && (flags & F_SHADOW)) {   // bounce phase: the ray drawn when the primary ray finished
        ox = g_ray[0]; oy = g_ray[1]; oz = g_ray[2]; dx = g_ray[3]; dy = g_ray[4]; dz = g_ray[5];
      }
      if (JOB == JOB_RENDER_SHADOW && (flags & F_SHADOW)) {
        const float pd = __uint_as_float(CTX(5));   // distance of this pixel's primary hit
        float sox, soy, soz, sdx, sdy, sdz, sdist;
        float lpx = p.lpos[0], lpy = p.lpos[1], lpz = p.lpos[2];
        if (A.pbatch) { const ShadeParams* q = A.pbatch + fast_div(job >> 6, A.div_frame_tiles); lpx = q->lpos[0]; lpy = q->lpos[1]; lpz = q->lpos[2]; }
        shadow_ray(lpx, lpy, lpz, ox, oy, oz, dx, dy, dz, pd, sox, soy, soz, sdx, sdy, sdz, sdist);
        ox = sox; oy = soy; oz = soz; dx = sdx; dy = sdy; dz = sdz;
        tmax_ = sdist;
      }
    }
  };
  // main launch only: hand this lane's ray (in its current phase) over to the EXACT launch
  auto defer = [&](bool counted) {
    const uint32_t slot = atomicAdd(A.defer_count, 1u);
    // (JOB_RENDER_GI: a bounce ray outside the fast domain sends the whole PIXEL to the EXACT launch, which traces its primary ray again
    // -- same hit by construction -- and goes on from there; the primary ray this launch counted is taken back)
    if (JOB == JOB_RENDER_GI && (flags & F_SHADOW)) nrays--;
    if (slot < A.defer_cap) A.defer_list[slot] = job | ((flags & F_SHADOW) && JOB != JOB_RENDER_GI ? 0x80000000u : 0u);
    if (JOB == JOB_RENDER_SHADOW && (flags & F_SHADOW)) {
      // the EXACT launch resumes this pixel's occlusion ray from the primary hit record in memory
      HitRec h;
      h.bx = __uint_as_float(CTX(3)); h.by = __uint_as_float(CTX(4)); h.bz = 1 - h.bx - h.by;
      h.dist = __uint_as_float(CTX(5)); h.blasIdx = CTX(6); h.triIdx = CTX(7);
      *hit_slot() = h;
    }
    if (counted) nrays--;   // the EXACT launch counts the ray when it starts it again
    cur = DESC_IDLE;
  };
  // TLAS leaf (rt_traversal.cpp:109-121): fetch the instance record, move the ray to object space
  auto enter_instance = [&](uint32_t blasIdx, float ox, float oy, float oz, float dx, float dy, float dz) {
    const uint32_t* bp = sc.blas + (size_t)blasIdx * (RT_BLAS_STRIDE / 4);
    uint32_t bw[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) bw[i] = bp[i];
    if (STATS) { fx.node++; fx.inst++; }
    const float m00 = __uint_as_float(bw[1]), m01 = __uint_as_float(bw[2]), m02 = __uint_as_float(bw[3]), m03 = __uint_as_float(bw[4]);
    const float m10 = __uint_as_float(bw[5]), m11 = __uint_as_float(bw[6]), m12 = __uint_as_float(bw[7]), m13 = __uint_as_float(bw[8]);
    const float m20 = __uint_as_float(bw[9]), m21 = __uint_as_float(bw[10]), m22 = __uint_as_float(bw[11]), m23 = __uint_as_float(bw[12]);
    arx = m00 * ox + m01 * oy + m02 * oz + m03;   // :231-261
    ary = m10 * ox + m11 * oy + m12 * oz + m13;
    arz = m20 * ox + m21 * oy + m22 * oz + m23;
    const float cdx = m00 * dx + m01 * dy + m02 * dz;
    const float cdy = m10 * dx + m11 * dy + m12 * dz;
    const float cdz = m20 * dx + m21 * dy + m22 * dz;
    aix = 1.0f / cdx; aiy = 1.0f / cdy; aiz = 1.0f / cdz;
    const bool s2 = ray_in_fast_domain(arx, ary, arz, aix, aiy, aiz);
    if (!EXACT && !s2) { defer(true); return; }   // object-space ray can produce NaN slabs: restart it in the EXACT launch
    flags &= ~F_WORLD;
    CTX(0) = __float_as_uint(cdx); CTX(1) = __float_as_uint(cdy); CTX(2) = __float_as_uint(cdz);
    CTX(8) = blasIdx;
    cur = blas_roots[blasIdx];   // BLAS root: same level, path_m unchanged
  };
  // (re)start the lane's traversal at the TLAS root (rt_traversal.cpp:39-40) with world ray (o, d)
  auto start_ray = [&](float ox, float oy, float oz, float dx, float dy, float dz, float tmax_, bool any_) {
    arx = ox; ary = oy; arz = oz;
    aix = 1.0f / dx; aiy = 1.0f / dy; aiz = 1.0f / dz;
    // the fast slab forms are exact only if no slab product can be NaN or overflow (see ray_in_fast_domain)
    const bool safe = ray_in_fast_domain(ox, oy, oz, aix, aiy, aiz);
    flags = (flags & (F_SHADOW | F_RESUMED)) | F_WORLD | (any_ ? F_ANYHIT : 0u);
    if (!EXACT && !safe) {
      // camera rays with a zero direction component are known before the launch (u == 0 or v == 0): the
      // host lists them and a concurrent EXACT launch traces them; everything else is deferred
      if (!is_trace_job(JOB) && !(flags & F_SHADOW)) cur = DESC_IDLE; else defer(false);
      return;
    }
    hitd = tmax_ > RT_LARGE_FLOAT ? RT_LARGE_FLOAT : tmax_;   // (a bound above 1e30 is 1e30: a missed box reports 1e30, rt_traversal.cpp:338, and must stay filtered by `d < hit.dist`)
    path_m = -__builtin_inff(); sp = 0; tos_d = DESC_DONE;
    cur = root_desc;
    nrays++;
    // single-instance scenes (the reference's default): the TLAS root is the instance leaf, enter it
    // right away with the ray at hand instead of re-deriving it in the instance step
    if (is_inst_desc(root_desc)) {
      // ... and when that instance's inverse transform is the identity (checked by the accel build: ones on the diagonal, zeros of either sign
      // elsewhere), the object-space ray IS the world ray, bit for bit, so the record fetch, the 18 multiply-adds, three divisions and the
      // second domain check of the instance step are skipped.  Why the bits agree: 1*x is x; adding products that are +-0 leaves a non-zero x
      // alone and turns a zero sum into +0 -- so the only component the arithmetic would change is an origin component that is -0 (it becomes
      // +0); directions have no zero component inside the fast domain.  Rays with a -0 origin component take the general step.
      const bool no_neg_zero = __float_as_uint(ox) != 0x80000000u && __float_as_uint(oy) != 0x80000000u && __float_as_uint(oz) != 0x80000000u;
      if (sc.ident_root && safe && no_neg_zero) {   // (an EXACT launch's rays outside the fast domain -- zero, infinite or NaN components -- take the general step)
        if (STATS) { fx.node++; fx.inst++; }
        flags &= ~F_WORLD;
        CTX(0) = __float_as_uint(dx); CTX(1) = __float_as_uint(dy); CTX(2) = __float_as_uint(dz);
        CTX(8) = root_desc & PAYLOAD_MASK;
        cur = root_blas_desc;
      } else enter_instance(root_desc & PAYLOAD_MASK, ox, oy, oz, dx, dy, dz);
    }
  };
  // (the scratch part of the stack through volatile pointers: the compiler must not speculate its loads into the common path)
  volatile uint32_t* const vovf_d = ovf_d;
  volatile float* const vovf_m = ovf_m;
  auto push = [&](uint32_t d, float m) {
    if (tos_d != DESC_DONE) {
      if (sp < LSTK) lstk[sp * 64] = make_uint2(tos_d, __float_as_uint(tos_m));
      else { ovf_d[sp - LSTK] = tos_d; ovf_m[sp - LSTK] = tos_m; }
      ++sp;
    }
    tos_d = d; tos_m = m;
  };
  // next pending work item of this lane (m < hit.dist: the reference's re-filtering),
  // or DESC_DONE when its stack is exhausted.  The refill of the register top from LDS is not waited for.
  auto pop_next = [&]() {
    cur = DESC_DONE;
    while (tos_d != DESC_DONE) {
      const uint32_t d = tos_d;
      const float m = tos_m;
      if (sp > 0) {
        --sp;
        if (sp < LSTK) { const uint2 e = lstk[sp * 64]; tos_d = e.x; tos_m = __uint_as_float(e.y); }
        else { tos_d = ovf_d[sp - LSTK]; tos_m = ovf_m[sp - LSTK]; }
      } else {
        tos_d = DESC_DONE;
      }
      if (m < hitd) { cur = d; path_m = m; break; }
    }
  };

  for (;;) {
    RT_MARK("fetch");
    if (STATS && A.wave_log) wl_tmark = __builtin_readcyclecounter();
    // ================= fetch: hand new jobs to idle lanes =================
    // Jobs are reserved per wavefront in chunks from one of the queue shards (one global atomic per
    // RT_CHUNK jobs); lanes then draw from the wavefront's private range.
    {
      const unsigned long long idle = __ballot(cur == DESC_IDLE);
      if (!queue_empty && idle != 0ull && (is_trace_job(JOB) || FINISH_MIN > 64u || idle == ~0ull || RT_DEAD_MAX < 64)) {
        const uint32_t wl_tries0 = tries; const unsigned long long wl_tpoll0 = (STATS && A.wave_log) ? __builtin_readcyclecounter() : 0ull;
        if (loc_next == loc_end) {   // wave-uniform: reserve the next chunk, stealing from other shards when the home shard is dry
          // shards a wavefront of this WORKGROUP has found handed out (LDS: no memory traffic): not polled again by its other three.  Every
          // wavefront used to poll every shard once before it ends -- 8 failing read-modify-writes each on the eight hottest lines of the
          // system, more than the launch's successful ones, all within its last third: serial frames +6 %, ray buffers +4 %
          // (profiles/r04_h_dry_mask_ab.txt).  Anything that ADDS traffic next to these counters loses, whatever it saves: the same mask
          // published through memory (a word read with agent scope and OR-ed into) -30 %, a look at the counter before the read-modify-write
          // -20 % (agent-scope load) / -47 % (non-temporal load, which turns every queue atomic into a round trip to memory), several tiles per
          // reservation -7 % (neighbouring tiles traced one after the other by ONE wavefront share less than the same tiles traced at the same
          // time by four: the queue's order is what keeps a CU's L1 warm) -- profiles/r04_h_*.txt.
          uint32_t dry = RT_QUEUE_DRY_MASK ? *(volatile uint32_t*)&s_dry : 0u;
          while (tries < QUEUE_SHARDS) {
            // the shards in the order this wavefront visits them: its home shard (its XCD's band), then RT_STEAL_SPREAD ? the others in
            // bit-reversed distance (+4, +2, +6, +1, +5, +3, +7: the helpers of a drained band spread over the remaining ones) : +1, +2, ...
            const uint32_t step = RT_STEAL_SPREAD ? (((tries & 1u) << 2) | (tries & 2u) | ((tries >> 2) & 1u)) : tries;
            const uint32_t sid = (shard + step) % QUEUE_SHARDS;
            if (RT_QUEUE_DRY_MASK && ((dry >> sid) & 1u)) { ++tries; continue; }
            const uint32_t s_lo = sid * per_shard;
            // a shard past the end of the job range costs no atomic (an EXACT launch with nothing deferred used to pay eight per
            // wavefront to find eight empty shards).  Written as an explicit range test: folded into `s_n == 0` on a select, this
            // compiler dropped the `s_lo < n_jobs` half of the condition and the wavefronts ran past the end of the job list.
            // The test is an asm statement the optimiser cannot look into, and its marker comment is what the build check greps for
            // in every instantiation's listing (tests/test_build_guards.py: one RTGUARD before the kernel's first queue atomic).
            uint32_t in_range;
            asm volatile("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0 ; RTGUARD shard_range" : "=s"(in_range)
                         : "s"(__builtin_amdgcn_readfirstlane(s_lo)), "s"(__builtin_amdgcn_readfirstlane(n_jobs)) : "scc");   // (both wave-uniform)
            if (!in_range) { ++tries; continue; }
            const uint32_t s_n = min(per_shard, n_jobs - s_lo);
            uint32_t base = 0;
            constexpr uint32_t CHUNK = (is_trace_job(JOB) && !EXACT) ? (uint32_t)RT_TRACE_CHUNK : (uint32_t)RT_CHUNK;
            if (lane == 0) base = atomicAdd(A.queue + sid * QUEUE_STRIDE, CHUNK);
            base = __shfl(base, 0);
            if (base < s_n) {
              loc_next = s_lo + base; loc_end = s_lo + min(base + CHUNK, s_n);
              break;
            }
            // handed out: tell the workgroup's other wavefronts
            if (RT_QUEUE_DRY_MASK) {
              if (lane == 0) atomicOr(&s_dry, 1u << sid);
              dry |= 1u << sid;
            }
            ++tries;
          }
          if (tries >= QUEUE_SHARDS) { queue_empty = true; if (STATS && A.wave_log && !wl_tq) wl_tq = wall_clock64(); }
          if (STATS && A.wave_log && !USE_TOP && tries != wl_tries0 && lane == 0) wl_no23 += (unsigned)(__builtin_readcyclecounter() - wl_tpoll0);   // (diagnostic: shader clocks of the reservations that met a dry shard)
        }
        uint32_t avail = loc_end - loc_next;
        if (!is_trace_job(JOB) && !EXACT) {
          avail = min(avail, 64u - (loc_next & 63u));     // lanes draw from ONE tile at a time (a reservation may span several)
          if (avail != 0u && (A.tile_order || A.tile_cost)) {
            // the tile these jobs belong to: queue position -> tile through the order of the launch; its cost is taken from here to the next tile's start
            const uint32_t pos = loc_next >> 6;
            const uint32_t tile = A.tile_order ? A.tile_order[pos] : pos;
            loc_off = __builtin_amdgcn_readfirstlane((tile << 6) - (loc_next & ~63u));   // (wave-uniform: a scalar register)
            if (A.tile_cost && tile != lpt_tile) {
              if (lane == 0 && lpt_tile != 0xFFFFFFFFu) A.tile_cost[lpt_tile] = lpt_work;
              if (STATS && A.wave_log) {   // (diagnostic: when the tile was started and how long the one before it took, in 100 MHz clocks, behind the costs)
                const unsigned long long now = wall_clock64();
                if (lane == 0) {
                  if (lpt_tile != 0xFFFFFFFFu) A.tile_cost[2u * (A.total >> 6) + lpt_tile] = (uint32_t)min(now - lpt_t0, 0xFFFFFFFFull);
                  A.tile_cost[(A.total >> 6) + tile] = (uint32_t)now;
                  A.tile_cost[3u * (A.total >> 6) + tile] = tries;   // 0 = taken from the wavefront's home shard, else stolen from the tries-th shard after it
                }
                lpt_t0 = now;
              }
              lpt_tile = tile; lpt_work = 0;
            }
          }
        }
        if (avail != 0u && cur == DESC_IDLE) {
          const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
          if (rank < avail) {
            job = loc_next + rank + loc_off;
            flags = 0;
            if (!EXACT && is_trace_job(JOB) && A.order) job = A.order[job];
            if (EXACT) {
              const uint32_t wd = A.defer_list[job];
              job = wd & 0x7fffffffu;
              if (wd >> 31) { flags = F_SHADOW | F_RESUMED; if (JOB == JOB_RENDER_SHADOW) CTX(5) = __float_as_uint(hit_slot()->dist); }
            }
            float ox, oy, oz, dx, dy, dz, tm;
            if (is_trace_job(JOB)) {
              world_ray(ox, oy, oz, dx, dy, dz, tm);
              start_ray(ox, oy, oz, dx, dy, dz, tm, A.any_hit != 0);
            } else {
              uint32_t x, y;
              pixel_of(job, x, y);
              if (x < A.W && y < A.y1) {   // kernel.cpp:62,101
                world_ray(ox, oy, oz, dx, dy, dz, tm);
                start_ray(ox, oy, oz, dx, dy, dz, tm, (flags & F_SHADOW) != 0u);
              }
            }
          }
        }
        loc_next += min(avail, (uint32_t)__popcll(idle));
      }
      if (__ballot(cur != DESC_IDLE) == 0ull) {
        if (queue_empty && loc_next == loc_end) {
          if (!is_trace_job(JOB) && !EXACT && A.tile_cost && lane == 0 && lpt_tile != 0xFFFFFFFFu) {
            A.tile_cost[lpt_tile] = lpt_work;
            if (STATS && A.wave_log) A.tile_cost[2u * (A.total >> 6) + lpt_tile] = (uint32_t)min(wall_clock64() - lpt_t0, 0xFFFFFFFFull);
          }
          break;
        }
        if (STATS && A.wave_log) wl_tf += __builtin_readcyclecounter() - wl_tmark;
        continue;
      }
    }
    if (STATS && A.wave_log) wl_tf += __builtin_readcyclecounter() - wl_tmark;

    // ================= traverse: one step of whatever each lane holds, per iteration =================
    for (;;) {
      RT_MARK("loop_top");
      if (!is_trace_job(JOB) && !EXACT) ++lpt_work;   // (wave-uniform: one scalar add per iteration)
      if (STATS && A.wave_log) {
        const unsigned long long nm = __ballot(is_node_desc(cur));
        ++wl_iter;
        if (nm) {
          ++wl_node_x; wl_node_l += (unsigned)__popcll(nm);
          const uint32_t c0 = __shfl(cur, __ffsll((long long)nm) - 1);
          if (__ballot(is_node_desc(cur) && cur == c0) == nm) ++wl_no3;   // every node lane of the wavefront is at the same node
        }
      }
      if (STATS && A.wave_log) wl_t0 = __builtin_readcyclecounter();
      RT_MARK("node");
      if (is_node_desc(cur)) {
        // ---- internal node: 4 box tests, order, push the far ones, continue with the nearest ----
        const bool top = (cur >> 30) == DK_TLAS;
        if (top && !(flags & F_WORLD)) {   // back at TLAS level after an instance (multi-instance scenes only)
          float ox, oy, oz, dx, dy, dz, tm;
          world_ray(ox, oy, oz, dx, dy, dz, tm);
          arx = ox; ary = oy; arz = oz; aix = 1.0f / dx; aiy = 1.0f / dy; aiz = 1.0f / dz;
          flags |= F_WORLD;
        }
        const uint32_t ni = cur & PAYLOAD_MASK;
        uint4 q0, q1, q2, q3;
        if (USE_TOP && (cur & DESC_TOP_FLAG)) {
          // (volatile, LDS-typed pointer: through plain pointers the compiler merges the two arms into FLAT loads of a
          // selected address -- thirteen flat_load instead of four ds_read_b128 / global_load_dwordx4)
          typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) const volatile u32x4_t lds_u32x4_t;
          lds_u32x4_t* tp = (lds_u32x4_t*)&s_top[0][0] + (cur & DESC_TOP_SLOT);
          const u32x4_t t0 = tp[0], t1 = tp[RT_TOP_NODES], t2 = tp[2 * RT_TOP_NODES], t3 = tp[3 * RT_TOP_NODES];
          q0 = make_uint4(t0.x, t0.y, t0.z, t0.w); q1 = make_uint4(t1.x, t1.y, t1.z, t1.w);
          q2 = make_uint4(t2.x, t2.y, t2.z, t2.w); q3 = make_uint4(t3.x, t3.y, t3.z, t3.w);
          if (STATS && A.wave_log) ++wl_no23;
        } else {
          const uint4* np = sc.nodes_c + (size_t)ni * CNODE_VEC4;
          q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
        }
        const uint32_t* ref_node = nullptr;
        if (LDEXP) ref_node = top ? sc.ref_tlas + (size_t)ni * RT_NODE_DWORDS : sc.ref_bvh + (size_t)(ni - sc.n_tlas) * RT_NODE_DWORDS;
        if (STATS) fx.node++;
        Cand c[4];
        eval_children<EXACT, LDEXP>(q0, q1, q2, q3, ref_node, arx, ary, arz, aix, aiy, aiz, hitd, c);
        if (((JOB == JOB_RENDER_SHADOW && RT_UNORDERED_OCCLUSION) || JOB == JOB_TRACE_UNORDERED) && STATS != 1 && __all((flags & F_ANYHIT) != 0u)) {   // STATS keeps the reference's order, hence its fetch counts
          // occlusion rays of a frame only feed a boolean (is anything hit before the light?): the set
          // of triangles an any-hit traversal can reach does not depend on the visiting order, so the
          // ordering network and the path_m bookkeeping are skipped (vxrt_trace's MODE_ANY, which
          // returns the reference's FIRST accepted candidate, keeps the ordered path)
          const bool v0 = c[0].d < __builtin_inff(), v1 = c[1].d < __builtin_inff(), v2 = c[2].d < __builtin_inff(), v3 = c[3].d < __builtin_inff();
#if RT_OCCLUSION_ORDER == 1
          // measurement variant: the child the ray enters LAST first (an occlusion ray starts on a surface: the boxes around its origin hold
          // that surface's own neighbourhood, which cannot block it), the others in slot order
          if (v0 || v1 || v2 || v3) {
            bool more = true;
            if (sp + 4 > STACK_CAP) { atomicOr(A.status, STATUS_STACK_OVERFLOW); more = false; }
            const float ninf = -__builtin_inff();
            const float e0 = v0 ? c[0].d : ninf, e1 = v1 ? c[1].d : ninf, e2 = v2 ? c[2].d : ninf, e3 = v3 ? c[3].d : ninf;
            const float m = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
            const int pick = e0 == m ? 0 : (e1 == m ? 1 : (e2 == m ? 2 : 3));
            cur = pick == 0 ? c[0].desc : (pick == 1 ? c[1].desc : (pick == 2 ? c[2].desc : c[3].desc));
            if (more) {
              if (v0 && pick != 0) push(c[0].desc, c[0].d);
              if (v1 && pick != 1) push(c[1].desc, c[1].d);
              if (v2 && pick != 2) push(c[2].desc, c[2].d);
              if (v3 && pick != 3) push(c[3].desc, c[3].d);
            }
          } else {
            pop_next();
          }
#elif RT_OCCLUSION_ORDER == 2
          // measurement variant: all children by descending entry distance
          if (v0 || v1 || v2 || v3) {
            bool more = true;
            if (sp + 4 > STACK_CAP) { atomicOr(A.status, STATUS_STACK_OVERFLOW); more = false; }
            order_children(c);      // valid first, nearest in c[0]
            const int nv = (int)v0 + (int)v1 + (int)v2 + (int)v3;
            cur = nv == 1 ? c[0].desc : (nv == 2 ? c[1].desc : (nv == 3 ? c[2].desc : c[3].desc));
            if (more) {
              if (nv > 1) push(c[0].desc, c[0].d);
              if (nv > 2) push(c[1].desc, c[1].d);
              if (nv > 3) push(c[2].desc, c[2].d);
            }
          } else {
            pop_next();
          }
#else
          if (v0 || v1 || v2 || v3) {
            bool more = true;
            if (sp + 4 > STACK_CAP) { atomicOr(A.status, STATUS_STACK_OVERFLOW); more = false; }
            cur = v0 ? c[0].desc : (v1 ? c[1].desc : (v2 ? c[2].desc : c[3].desc));
            if (more) {
#if RT_OCCLUSION_SLOT_ORDER
              // pushed last = visited next: the children are visited in slot order, which a builder may choose (largest surface area first)
              if (v3 && (v0 || v1 || v2)) push(c[3].desc, c[3].d);
              if (v2 && (v0 || v1)) push(c[2].desc, c[2].d);
              if (v1 && v0) push(c[1].desc, c[1].d);
#else
              if (v1 && v0) push(c[1].desc, c[1].d);
              if (v2 && (v0 || v1)) push(c[2].desc, c[2].d);
              if (v3 && (v0 || v1 || v2)) push(c[3].desc, c[3].d);
#endif
            }
          } else {
            pop_next();
          }
#endif
        } else {
          order_children(c);   // valid children first (d < inf), nearest in c[0]
          // (path_m and the candidates' distances are never NaN -- a filtered child carries +inf -- so the maxima need no
          // canonicalising v_max x, x in front of them)
          if (c[0].d < __builtin_inff()) {
            bool more = true;
            if (sp + 4 > STACK_CAP) { atomicOr(A.status, STATUS_STACK_OVERFLOW); more = false; }
            // far first so that the nearest pending sibling is on top (:98-103)
            if (more && c[3].d < __builtin_inff()) push(c[3].desc, vmax_nonan(path_m, c[3].d));
            if (more && c[2].d < __builtin_inff()) push(c[2].desc, vmax_nonan(path_m, c[2].d));
            if (more && c[1].d < __builtin_inff()) push(c[1].desc, vmax_nonan(path_m, c[1].d));
            cur = c[0].desc;
            path_m = vmax_nonan(path_m, c[0].d);
          } else {
            pop_next();
          }
        }
      }
      if (STATS && A.wave_log) { const unsigned long long t1 = __builtin_readcyclecounter(); wl_tn += t1 - wl_t0; wl_t0 = t1; }
      RT_MARK("inst");
      if (__any(is_inst_desc(cur))) {
        if (is_inst_desc(cur)) {
          float ox, oy, oz, dx, dy, dz, tm;
          world_ray(ox, oy, oz, dx, dy, dz, tm);
          enter_instance(cur & PAYLOAD_MASK, ox, oy, oz, dx, dy, dz);
        }
      }
      // leaves are postponed until RT_LEAF_MIN lanes hold one (or no lane has a node left): the leaf
      // body then runs for many lanes at once instead of once per iteration for a few
      RT_MARK("leaf");
      const unsigned long long leafm = __ballot(is_leaf_desc(cur));
      if (leafm != 0ull && ((uint32_t)__popcll(leafm) >= (is_trace_job(JOB) ? RT_TRACE_LEAF_MIN : RT_LEAF_MIN) || __ballot(is_node_desc(cur) || is_inst_desc(cur)) == 0ull)) {
        // ---- BLAS leaf (:123-161): triangles in index order, strict '<' ----
        if (STATS && A.wave_log) { ++wl_leaf_x; wl_leaf_l += (unsigned)__popcll(leafm); }
        if (!is_trace_job(JOB) && !EXACT) lpt_work += 2u;   // (a leaf-body run costs about 2.5 node-body runs: tools/wave_balance.py)
        {
          // triangles through LDS (RT_TRI_LDS): the lanes of a tile reach the same leaves, and every one of them loads the leaf's
          // triangles for itself.  The leaf of the first leaf lane is loaded ONCE, by 3 lanes per triangle, and handed to the lanes
          // that hold it by broadcast reads; the others load theirs as before.
          bool tri_from_lds = false;
          if (TRI_LDS) {
            const uint32_t L0 = __shfl(cur, __ffsll((long long)leafm) - 1);
            const uint32_t c0 = (L0 >> LEAF_FIRST_BITS) & LEAF_MAX_INLINE;
            const unsigned long long same = __ballot(cur == L0);
            if (c0 != 0u && c0 <= (uint32_t)RT_TRI_LDS && (uint32_t)__popcll(same) >= (uint32_t)RT_TRI_LDS_MIN) {   // (wave-uniform)
              if (lane < 3u * c0) s_tri[threadIdx.x >> 6][lane] = sc.tri_w[(size_t)(L0 & LEAF_FIRST_MASK) * 3 + lane];
              tri_from_lds = cur == L0;
              if (STATS && A.wave_log) wl_no23 += (unsigned)__popcll(same);
            }
          }
          if (is_leaf_desc(cur)) {
            if (STATS) fx.node++;
            uint32_t leftFirst = cur & LEAF_FIRST_MASK, triCount = (cur >> LEAF_FIRST_BITS) & LEAF_MAX_INLINE;
            if (triCount == 0u) {   // leaf with more than 15 triangles: range kept in the reference node
              const uint32_t* rn = sc.ref_bvh + (size_t)leftFirst * RT_NODE_DWORDS;
              leftFirst = rn[4]; triCount = rn[5];
            }
            const float cdx = __uint_as_float(CTX(0)), cdy = __uint_as_float(CTX(1)), cdz = __uint_as_float(CTX(2));
            bool stop = false;
            // ray buffers (incoherent rays, latency-bound leaves): the next triangle's 48 bytes are requested before the
            // current one is tested, +3 %; camera tiles lose 1.5 % to the extra registers, so they load in place
            constexpr bool PREFETCH = is_trace_job(JOB) && RT_TRI_PREFETCH;
            float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
            if (PREFETCH) { const float4* tp0 = sc.tri_w + (size_t)leftFirst * 3; n0 = tp0[0]; n1 = tp0[1]; n2 = tp0[2]; }
            for (uint32_t i = 0; i < triCount; ++i) {
              const uint32_t triIdx = leftFirst + i;
              float4 t0, t1, t2;
              if (TRI_LDS && tri_from_lds) {
                // (typed LDS pointer: through a plain pointer the compiler would merge this arm and the global one into flat loads)
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) const volatile f32x4_t lds_f32x4_t;
                lds_f32x4_t* lp = (lds_f32x4_t*)&s_tri[threadIdx.x >> 6][0] + 3u * i;
                const f32x4_t a0 = lp[0], a1 = lp[1], a2 = lp[2];
                t0 = make_float4(a0.x, a0.y, a0.z, a0.w); t1 = make_float4(a1.x, a1.y, a1.z, a1.w); t2 = make_float4(a2.x, a2.y, a2.z, a2.w);
              } else
              if (PREFETCH) {
                t0 = n0; t1 = n1; t2 = n2;
                if (i + 1u < triCount) { const float4* tn = sc.tri_w + (size_t)(triIdx + 1u) * 3; n0 = tn[0]; n1 = tn[1]; n2 = tn[2]; }
              } else {
                const float4* tp = sc.tri_w + (size_t)triIdx * 3;
                t0 = tp[0]; t1 = tp[1]; t2 = tp[2];
              }
              if (STATS) fx.tri++;
              float bx, by, bz;
              const float d = ray_tri(arx, ary, arz, cdx, cdy, cdz, t0, t1, t2, bx, by, bz);
              if (d < hitd) {
                hitd = d;
                flags |= F_FOUND;
                // (a frame's occlusion ray only feeds a boolean; slots 3-7 keep the pixel's primary hit meanwhile)
                if (!(JOB == JOB_RENDER_SHADOW && (flags & F_SHADOW))) { CTX(3) = __float_as_uint(bx); CTX(4) = __float_as_uint(by); CTX(6) = CTX(8); CTX(7) = triIdx; }
                if (flags & F_ANYHIT) { stop = true; break; }
                // the reference re-descends from the root with the shrunken hit.dist; if any box on the
                // current path no longer passes `d < hit.dist` it abandons this subtree
                if (!(path_m < hitd)) break;
              }
            }
            if (stop) { sp = 0; tos_d = DESC_DONE; cur = DESC_DONE; }
            else pop_next();
          }
        }
      }
      if (STATS && A.wave_log) { const unsigned long long t1 = __builtin_readcyclecounter(); wl_tl += t1 - wl_t0; }
      RT_MARK("loop_exit");
      // leave when nothing traverses any more, or when enough lanes are dead weight AND leaving can
      // revive them (finished rays to retire, or idle lanes while jobs remain)
      const unsigned long long work = __ballot(is_work_desc(cur));
      if (work == 0ull) break;
      const unsigned long long done = __ballot(cur == DESC_DONE);
      const uint32_t n_done = (uint32_t)__popcll(done);
      const uint32_t n_idle = queue_empty && loc_next == loc_end ? 0u : 64u - (uint32_t)__popcll(work | done);
      if (n_done + n_idle >= DEAD_MAX || n_done >= FINISH_MIN) break;
    }

    // ================= finish: rays whose traversal ended =================
    RT_MARK("finish");
    if (STATS && A.wave_log) wl_tmark = __builtin_readcyclecounter();
    if (cur == DESC_DONE) {
      const bool found = (flags & F_FOUND) != 0u;
      HitRec h; h.dist = RT_LARGE_FLOAT; h.bx = 0; h.by = 0; h.bz = 0; h.blasIdx = 0; h.triIdx = 0;
      if (is_trace_job(JOB)) {
        if (found) {
          h.dist = hitd; h.bx = __uint_as_float(CTX(3)); h.by = __uint_as_float(CTX(4)); h.bz = 1 - h.bx - h.by;   // rt_traversal.cpp:311-313
          h.blasIdx = CTX(6); h.triIdx = CTX(7);
        }
        if (RT_TRACE_NT) {
          uint32_t* hp = (uint32_t*)hit_slot();
          __builtin_nontemporal_store(__float_as_uint(h.dist), hp); __builtin_nontemporal_store(__float_as_uint(h.bx), hp + 1);
          __builtin_nontemporal_store(__float_as_uint(h.by), hp + 2); __builtin_nontemporal_store(__float_as_uint(h.bz), hp + 3);
          __builtin_nontemporal_store(h.blasIdx, hp + 4); __builtin_nontemporal_store(h.triIdx, hp + 5);
        } else *hit_slot() = h;
        cur = DESC_IDLE;
      } else if (JOB == JOB_RENDER_GI) {
        // one diffuse bounce, in the lane (see JOB_RENDER_GI above).  Every step is the code of the pass it replaces:
        // rt_ao_prepare_kernel (colour / albedo / hit point / normal of the primary hit), rt_ao_rays_kernel (the bounce ray) and, for the rest,
        // what the multi-pass form did after its trace launch: shade the bounce hit, colour += albedo * that, pack (orc_render_gi).
        uint32_t x, y;
        pixel_of(job, x, y);
        const size_t e = (size_t)x + (size_t)y * A.W;
        bool write = false;
        float cr = 0.f, cg = 0.f, cb = 0.f;
        if (found) {
          h.dist = hitd; h.bx = __uint_as_float(CTX(3)); h.by = __uint_as_float(CTX(4)); h.bz = 1 - h.bx - h.by;
          h.blasIdx = CTX(6); h.triIdx = CTX(7);
        }
        if (!(flags & F_SHADOW)) {
          float ox, oy, oz, dx, dy, dz;
          generate_ray(A.utab[x], A.vtab[y], ox, oy, oz, dx, dy, dz);
          if (!found) {
            cr = p.bg[0]; cg = p.bg[1]; cb = p.bg[2];   // miss.cpp:9-14; no bounce
            write = true;
          } else {
            float r, g, b, refl, Ix, Iy, Iz, Nx, Ny, Nz, a3[3];
            shade_terms<false>(sc, p, ox, oy, oz, dx, dy, dz, h, false, r, g, b, refl, Ix, Iy, Iz, Nx, Ny, Nz, nullptr, a3);
            float thr = 1.0f;
            thr *= refl;
            g_col[0] = r + p.bg[0] * thr; g_col[1] = g + p.bg[1] * thr; g_col[2] = b + p.bg[2] * thr;
            g_alb[0] = a3[0]; g_alb[1] = a3[1]; g_alb[2] = a3[2];
            ao_sample_ray(x, y, A.W, 1u, 0u, A.gi_seed, Ix, Iy, Iz, Nx, Ny, Nz, dx, dy, dz, g_ray);
            flags = F_SHADOW;       // (second phase of the pixel)
            start_ray(g_ray[0], g_ray[1], g_ray[2], g_ray[3], g_ray[4], g_ray[5], RT_LARGE_FLOAT, false);
          }
        } else {
          float r, g, b;
          shade_eval<false>(sc, p, g_ray[0], g_ray[1], g_ray[2], g_ray[3], g_ray[4], g_ray[5], h, found, false, r, g, b);
          cr = g_col[0] + g_alb[0] * r; cg = g_col[1] + g_alb[1] * g; cb = g_col[2] + g_alb[2] * b;
          write = true;
        }
        if (write) {
          A.dst[e] = pack_rgb8(cr, cg, cb);
          if (A.colors) { A.colors[3 * e] = cr; A.colors[3 * e + 1] = cg; A.colors[3 * e + 2] = cb; }
          cur = DESC_IDLE;
        }
      } else if (!(flags & F_SHADOW)) {
        // deferred shading: finishing a ray costs one store, not a chain of dependent loads
        if (STATS && found) nhit++;
        if (JOB == JOB_RENDER_SHADOW && found) {
          // continue this lane with the pixel's occlusion ray; the record is written when that ray has finished
          uint32_t x, y;
          pixel_of(job, x, y);
          float ox, oy, oz, dx, dy, dz, sox, soy, soz, sdx, sdy, sdz, sdist;
          generate_ray(A.utab[x], A.vtab[y], ox, oy, oz, dx, dy, dz);
          float lpx = p.lpos[0], lpy = p.lpos[1], lpz = p.lpos[2];
          if (A.pbatch) { const ShadeParams* q = A.pbatch + fast_div(job >> 6, A.div_frame_tiles); lpx = q->lpos[0]; lpy = q->lpos[1]; lpz = q->lpos[2]; }
          shadow_ray(lpx, lpy, lpz, ox, oy, oz, dx, dy, dz, hitd, sox, soy, soz, sdx, sdy, sdz, sdist);
          CTX(5) = __float_as_uint(hitd);
          flags = F_SHADOW;
          start_ray(sox, soy, soz, sdx, sdy, sdz, sdist, true);
        } else {
          if (found) {
            h.dist = hitd; h.bx = __uint_as_float(CTX(3)); h.by = __uint_as_float(CTX(4)); h.bz = 1 - h.bx - h.by;
            h.blasIdx = CTX(6); h.triIdx = CTX(7);
          }
          *hit_slot() = h;   // tile-major: job = tile * 64 + lane
          cur = DESC_IDLE;
        }
      } else if (EXACT && (flags & F_RESUMED)) {
        // occlusion ray handed over by the main launch: the record is in memory already, only the result is added
        if (found) hit_slot()->blasIdx |= 0x80000000u;
        cur = DESC_IDLE;
      } else {
        h.bx = __uint_as_float(CTX(3)); h.by = __uint_as_float(CTX(4)); h.bz = 1 - h.bx - h.by;
        h.dist = __uint_as_float(CTX(5)); h.blasIdx = CTX(6) | (found ? 0x80000000u : 0u); h.triIdx = CTX(7);   // found = occluded
        *hit_slot() = h;
        cur = DESC_IDLE;
      }
    }
    if (STATS && A.wave_log) wl_tfin += __builtin_readcyclecounter() - wl_tmark;
  }
#undef CTX

  if (STATS && A.wave_log && !EXACT) {
    unsigned s = nrays;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    for (int o = 32; o > 0; o >>= 1) wl_no23 += __shfl_down(wl_no23, o);   // node steps served from the LDS image, all lanes
    if (lane == 0) {
      unsigned long long* w = A.wave_log + 16ull * (blockIdx.x * (uint32_t)WG_WAVES + (threadIdx.x >> 6));
      w[13] = wl_tf; w[14] = wl_tfin; w[15] = wl_tq;
      w[0] = t_first; w[1] = wall_clock64(); w[2] = s; w[8] = wl_no23;
      w[9] = (unsigned long long)wl_no3 | ((unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 56);   // [63:56] physical XCD
      w[10] = wl_tn; w[11] = wl_tl; w[12] = __builtin_readcyclecounter() - wl_tstart;
      w[3] = wl_iter; w[4] = wl_node_x; w[5] = wl_node_l; w[6] = wl_leaf_x; w[7] = wl_leaf_l;
    }
  }
  if (!EXACT && A.end_log) {
    unsigned s = nrays;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) {
      unsigned long long* w = A.end_log + 2ull * (blockIdx.x * (uint32_t)WG_WAVES + (threadIdx.x >> 6));
      w[0] = wall_clock64();
      w[1] = (unsigned long long)s | ((unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 56);
    }
  }
  if (A.counters) {
    // one device atomic per WORKGROUP and counter (its wavefronts add up in LDS first): per wavefront, the 7,168 read-modify-writes a frame's
    // launch ends with -- all on one cache line -- cost the frame 40 us (8 %: profiles/r04_p_rays_counter.txt)
    __shared__ unsigned s_cnt[5];
    if (threadIdx.x < 5u) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    unsigned v[5] = {nrays, fx.node, fx.inst, fx.tri, nhit};
#pragma unroll
    for (int k = 0; k < (STATS ? 5 : 1); ++k) {
      unsigned s = v[k];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
      if (lane == 0 && s) atomicAdd(&s_cnt[k], s);
    }
    __syncthreads();
    if (threadIdx.x < (STATS ? 5u : 1u) && s_cnt[threadIdx.x]) atomicAdd(A.counters + threadIdx.x, (unsigned long long)s_cnt[threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------------
// Ray-pool trace kernel (ray buffers only; round 5, VERDICT item 2: incoherent rays compacted through LDS).
//
// The persistent kernel above gives every lane ONE ray and runs, per loop iteration, the node body for the lanes that hold a node and the
// leaf body for the lanes that hold a leaf: with incoherent rays half the lanes of every instruction are masked (lane utilisation 0.50 on the
// 16 Mi random rays: profiles/r04_zz_pmc_summary_random_rays.txt) -- rays of a wavefront are in different phases, and a refilled lane does not
// change that.  Here a wavefront owns a POOL of RT_POOL_SLOTS rays (more than it has lanes) whose whole state lives in LDS -- active ray,
// hit distance, path maximum, current work item, stack -- and every iteration it picks up to 64 slots that are in the SAME phase (a
// ballot + prefix count over the slots' work items), loads their state, runs that one body at full width and stores the state back.  Lanes
// are workers, not owners: nothing of a ray lives in registers between two iterations.  Per-ray arithmetic is the persistent kernel's, step
// by step (same eval_children / order_children / ray_tri, same push order, same `m < hit.dist` rule), so the hit records are the same bits;
// the schedule is what differs.  One wavefront per workgroup: no barrier couples wavefronts.
//   * the stack's first RT_POOL_LSTK entries of a slot are in LDS, deeper ones in a global spill area (a lane's scratch cannot follow a ray
//     from lane to lane); there is no register top;
//   * an accepted hit is written to the ray's record at once (closest hit: the last accept is the record; records of misses are written
//     when the ray ends), so barycentrics and indices need no LDS;
//   * rays outside the fast domain go to the deferral list and the EXACT launch, as in the persistent kernel.
// ---------------------------------------------------------------------------------------------
#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 96        // rays a wavefront holds (64 lanes work on them)
#endif
#ifndef RT_POOL_LSTK
#define RT_POOL_LSTK 5          // stack entries of a slot kept in LDS
#endif
#ifndef RT_POOL_WAVES
#define RT_POOL_WAVES 4         // wavefronts per SIMD the kernel is compiled for (LDS: (15 + 2 * RT_POOL_LSTK) * 4 * RT_POOL_SLOTS bytes per wavefront)
#endif
#ifndef RT_POOL_LEAF_MIN
#define RT_POOL_LEAF_MIN 48     // the leaf body runs once this many slots hold a leaf (or nothing else can run)
#endif
#ifndef RT_POOL_NODE_KEEP
#define RT_POOL_NODE_KEEP 44    // a node pass goes on with the same slots while at least this many of its lanes are still at a node (65: one step per pass)
#endif
#ifndef RT_POOL_REFILL_MIN
#define RT_POOL_REFILL_MIN 16   // finished / empty slots are serviced (records of misses written, new rays started) once there are this many
#endif
enum { PF_OX = 0, PF_OY, PF_OZ, PF_IX, PF_IY, PF_IZ, PF_HITD, PF_PATHM, PF_CUR, PF_JOB, PF_FLAGS, PF_DX, PF_DY, PF_DZ, PF_BLAS, PF_STK };
#define PF_SP_SHIFT 8           // PF_FLAGS: per-ray flag bits in [7:0], stack entries in [15:8]

template <bool LDEXP, bool SHALLOW>
__global__ __launch_bounds__(64, RT_POOL_WAVES) void rt_pool_trace_kernel(SceneDev sc, PersistArgs A, uint2* __restrict__ spill) {
  constexpr int S = RT_POOL_SLOTS, L = RT_POOL_LSTK;
  static_assert(S >= 64 && S <= 128, "a lane classifies at most two slots");
  constexpr int CAP = SHALLOW ? 3 * RT_SHALLOW_LEVELS : 3 * RT_MAX_LEVELS + L;   // entries a ray's stack may hold
  constexpr int OVF = CAP - L;
  __shared__ uint32_t pool[PF_STK + 2 * L][S];
  __shared__ uint32_t list[64];
  const uint32_t lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint2* const my_spill = spill + (size_t)blockIdx.x * S * OVF;
  const uint32_t n_jobs = A.total_dev ? min(*A.total_dev, A.total) : A.total;
  const uint32_t per_shard = A.total_dev ? (((n_jobs + QUEUE_SHARDS - 1) / QUEUE_SHARDS + 63u) & ~63u) : A.per_shard;
  const uint32_t root_desc = sc.tlas_root;
  const uint32_t xcc_id = RT_XCC_HOME ? (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) : blockIdx.x;
  const uint32_t shard = (xcc_id + A.shard_rot) % QUEUE_SHARDS;
  uint32_t tries = 0, loc_next = 0, loc_end = 0;
  bool queue_empty = false;
  unsigned nrays = 0;
  for (uint32_t s = lane; s < (uint32_t)S; s += 64u) { pool[PF_CUR][s] = DESC_IDLE; pool[PF_FLAGS][s] = 0u; }
  __syncthreads();

  // ---- per-slot state of the lane's current slot, in registers for the length of one pass ----
  uint32_t slot = 0, cur = DESC_IDLE, job = 0, flags = 0, sp = 0;
  float arx = 0, ary = 0, arz = 0, aix = 0, aiy = 0, aiz = 0, hitd = 0, path_m = 0;
  auto push = [&](uint32_t d, float m) {
    if (sp < (uint32_t)L) { pool[PF_STK + 2 * sp][slot] = d; pool[PF_STK + 2 * sp + 1][slot] = __float_as_uint(m); }
    else my_spill[(size_t)slot * OVF + (sp - L)] = make_uint2(d, __float_as_uint(m));
    ++sp;
  };
  auto pop_next = [&]() {      // next pending work item with m < hit.dist, or the end of the ray
    cur = DESC_DONE;
    while (sp > 0u) {
      --sp;
      uint32_t d; float m;
      if (sp < (uint32_t)L) { d = pool[PF_STK + 2 * sp][slot]; m = __uint_as_float(pool[PF_STK + 2 * sp + 1][slot]); }
      else { const uint2 e = my_spill[(size_t)slot * OVF + (sp - L)]; d = e.x; m = __uint_as_float(e.y); }
      if (m < hitd) { cur = d; path_m = m; break; }
    }
  };
  auto defer = [&]() {
    const uint32_t q = atomicAdd(A.defer_count, 1u);
    if (q < A.defer_cap) A.defer_list[q] = job;
    cur = DESC_IDLE;
  };
  auto world_ray = [&](float& ox, float& oy, float& oz, float& dx, float& dy, float& dz) {
    const float* rp = A.rays + (size_t)job * 6;
    ox = rp[0]; oy = rp[1]; oz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5];
  };
  // TLAS leaf (rt_traversal.cpp:109-121): the instance record, the ray in object space
  auto enter_instance = [&](uint32_t blasIdx, float ox, float oy, float oz, float dx, float dy, float dz, bool counted) {
    const uint32_t* bp = sc.blas + (size_t)blasIdx * (RT_BLAS_STRIDE / 4);
    uint32_t bw[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) bw[i] = bp[i];
    const float m00 = __uint_as_float(bw[1]), m01 = __uint_as_float(bw[2]), m02 = __uint_as_float(bw[3]), m03 = __uint_as_float(bw[4]);
    const float m10 = __uint_as_float(bw[5]), m11 = __uint_as_float(bw[6]), m12 = __uint_as_float(bw[7]), m13 = __uint_as_float(bw[8]);
    const float m20 = __uint_as_float(bw[9]), m21 = __uint_as_float(bw[10]), m22 = __uint_as_float(bw[11]), m23 = __uint_as_float(bw[12]);
    arx = m00 * ox + m01 * oy + m02 * oz + m03;   // :231-261
    ary = m10 * ox + m11 * oy + m12 * oz + m13;
    arz = m20 * ox + m21 * oy + m22 * oz + m23;
    const float cdx = m00 * dx + m01 * dy + m02 * dz;
    const float cdy = m10 * dx + m11 * dy + m12 * dz;
    const float cdz = m20 * dx + m21 * dy + m22 * dz;
    aix = 1.0f / cdx; aiy = 1.0f / cdy; aiz = 1.0f / cdz;
    if (!ray_in_fast_domain(arx, ary, arz, aix, aiy, aiz)) { if (counted) nrays--; defer(); return; }   // (the EXACT launch counts the ray when it starts it again)
    flags &= ~F_WORLD;
    pool[PF_DX][slot] = __float_as_uint(cdx); pool[PF_DY][slot] = __float_as_uint(cdy); pool[PF_DZ][slot] = __float_as_uint(cdz);
    pool[PF_BLAS][slot] = blasIdx;
    cur = sc.blas_root[blasIdx];
  };
  auto load_state = [&]() {
    cur = pool[PF_CUR][slot]; job = pool[PF_JOB][slot];
    const uint32_t f = pool[PF_FLAGS][slot]; flags = f & 0xFFu; sp = f >> PF_SP_SHIFT;
    arx = __uint_as_float(pool[PF_OX][slot]); ary = __uint_as_float(pool[PF_OY][slot]); arz = __uint_as_float(pool[PF_OZ][slot]);
    aix = __uint_as_float(pool[PF_IX][slot]); aiy = __uint_as_float(pool[PF_IY][slot]); aiz = __uint_as_float(pool[PF_IZ][slot]);
    hitd = __uint_as_float(pool[PF_HITD][slot]); path_m = __uint_as_float(pool[PF_PATHM][slot]);
  };
  auto store_ray = [&]() {
    pool[PF_OX][slot] = __float_as_uint(arx); pool[PF_OY][slot] = __float_as_uint(ary); pool[PF_OZ][slot] = __float_as_uint(arz);
    pool[PF_IX][slot] = __float_as_uint(aix); pool[PF_IY][slot] = __float_as_uint(aiy); pool[PF_IZ][slot] = __float_as_uint(aiz);
  };
  auto store_walk = [&]() {
    pool[PF_CUR][slot] = cur; pool[PF_FLAGS][slot] = flags | (sp << PF_SP_SHIFT);
    pool[PF_HITD][slot] = __float_as_uint(hitd); pool[PF_PATHM][slot] = __float_as_uint(path_m);
  };

  for (;;) {
    // ---- what phase is every slot in?  (a lane looks at slots lane and lane + 64) ----
    const uint32_t c0 = pool[PF_CUR][lane];
    const uint32_t c1 = lane + 64u < (uint32_t)S ? pool[PF_CUR][lane + 64u] : DESC_IDLE;
    enum { P_NODE, P_LEAF, P_INST, P_SERVICE };
    int pass = -1;
    unsigned long long m0, m1;
    // (the masks of the phases are formed only as far as the decision needs them: a full node pass, the common case, costs two compares)
    const unsigned long long node0 = __ballot(is_node_desc(c0)), node1 = __ballot(is_node_desc(c1));
    const uint32_t n_node = (uint32_t)(__popcll(node0) + __popcll(node1));
    if (n_node >= 64u) { pass = P_NODE; m0 = node0; m1 = node1; }
    else {
      const unsigned long long leaf0 = __ballot(is_leaf_desc(c0)), leaf1 = __ballot(is_leaf_desc(c1));
      const uint32_t n_leaf = (uint32_t)(__popcll(leaf0) + __popcll(leaf1));
      if (n_leaf >= (uint32_t)RT_POOL_LEAF_MIN) { pass = P_LEAF; m0 = leaf0; m1 = leaf1; }
      else {
        const bool more_jobs = !(queue_empty && loc_next == loc_end);
        const unsigned long long slots1 = S < 128 ? ((1ull << (S - 64)) - 1ull) : ~0ull;
        const unsigned long long srv0 = __ballot(c0 == DESC_DONE || (more_jobs && c0 == DESC_IDLE));
        const unsigned long long srv1 = __ballot(c1 == DESC_DONE || (more_jobs && c1 == DESC_IDLE)) & slots1;
        const uint32_t n_service = (uint32_t)(__popcll(srv0) + __popcll(srv1));
        const unsigned long long inst0 = __ballot(is_inst_desc(c0)), inst1 = __ballot(is_inst_desc(c1));
        if (n_service >= (uint32_t)RT_POOL_REFILL_MIN) { pass = P_SERVICE; m0 = srv0; m1 = srv1; }
        else if (inst0 | inst1) { pass = P_INST; m0 = inst0; m1 = inst1; }
        else if (n_node) { pass = P_NODE; m0 = node0; m1 = node1; }
        else if (n_leaf) { pass = P_LEAF; m0 = leaf0; m1 = leaf1; }
        else if (n_service) { pass = P_SERVICE; m0 = srv0; m1 = srv1; }
        else break;                                    // every slot empty, nothing left in the queue
      }
    }
    const uint32_t k0 = (uint32_t)__popcll(m0), n_sel = min(64u, k0 + (uint32_t)__popcll(m1));
    // (one wavefront per workgroup: its LDS instructions execute in order, so the list written here is what the reads below see -- the
    // compiler must only keep them in program order; no s_barrier and, above all, no wait for the global loads and stores in flight)
    __builtin_amdgcn_wave_barrier();
    if ((m0 >> lane) & 1ull) list[(uint32_t)__popcll(m0 & lt_mask)] = lane;
    if ((m1 >> lane) & 1ull) { const uint32_t r = k0 + (uint32_t)__popcll(m1 & lt_mask); if (r < 64u) list[r] = lane + 64u; }
    __builtin_amdgcn_wave_barrier();
    const bool act = lane < n_sel;
    slot = act ? list[lane] : 0u;

    if (pass == P_NODE) {
      if (act) load_state();
      // a lane keeps its slot while enough of the pass's lanes are still at a node: the pool's cost -- finding the slots, loading and storing
      // their state -- is paid once for several steps, at the price of the lanes that have meanwhile reached a leaf or the end waiting masked
      for (;;) {
      if (act && is_node_desc(cur)) {
        const bool top = (cur >> 30) == DK_TLAS;
        if (top && !(flags & F_WORLD)) {               // back at TLAS level after an instance (multi-instance scenes only)
          float dx, dy, dz;
          world_ray(arx, ary, arz, dx, dy, dz);
          aix = 1.0f / dx; aiy = 1.0f / dy; aiz = 1.0f / dz;
          flags |= F_WORLD;
          store_ray();
        }
        const uint32_t ni = cur & PAYLOAD_MASK;
        const uint4* np = sc.nodes_c + (size_t)ni * CNODE_VEC4;
        const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
        const uint32_t* ref_node = nullptr;
        if (LDEXP) ref_node = top ? sc.ref_tlas + (size_t)ni * RT_NODE_DWORDS : sc.ref_bvh + (size_t)(ni - sc.n_tlas) * RT_NODE_DWORDS;
        Cand c[4];
        eval_children<false, LDEXP>(q0, q1, q2, q3, ref_node, arx, ary, arz, aix, aiy, aiz, hitd, c);
        order_children(c);
        if (c[0].d < __builtin_inff()) {
          bool more = true;
          if (sp + 3u > (uint32_t)CAP) { atomicOr(A.status, STATUS_STACK_OVERFLOW); more = false; }
          if (more && c[3].d < __builtin_inff()) push(c[3].desc, vmax_nonan(path_m, c[3].d));   // far first (:98-103)
          if (more && c[2].d < __builtin_inff()) push(c[2].desc, vmax_nonan(path_m, c[2].d));
          if (more && c[1].d < __builtin_inff()) push(c[1].desc, vmax_nonan(path_m, c[1].d));
          cur = c[0].desc;
          path_m = vmax_nonan(path_m, c[0].d);
        } else pop_next();
      }
      if ((uint32_t)__popcll(__ballot(act && is_node_desc(cur))) < (uint32_t)RT_POOL_NODE_KEEP) break;
      }
      if (act) store_walk();
    } else if (pass == P_LEAF) {
      if (act) {
        load_state();
        uint32_t leftFirst = cur & LEAF_FIRST_MASK, triCount = (cur >> LEAF_FIRST_BITS) & LEAF_MAX_INLINE;
        if (triCount == 0u) { const uint32_t* rn = sc.ref_bvh + (size_t)leftFirst * RT_NODE_DWORDS; leftFirst = rn[4]; triCount = rn[5]; }
        const float cdx = __uint_as_float(pool[PF_DX][slot]), cdy = __uint_as_float(pool[PF_DY][slot]), cdz = __uint_as_float(pool[PF_DZ][slot]);
        const uint32_t blasIdx = pool[PF_BLAS][slot];
        bool stop = false;
        float4 n0, n1, n2;
        { const float4* tp0 = sc.tri_w + (size_t)leftFirst * 3; n0 = tp0[0]; n1 = tp0[1]; n2 = tp0[2]; }
        for (uint32_t i = 0; i < triCount; ++i) {
          const uint32_t triIdx = leftFirst + i;
          const float4 t0 = n0, t1 = n1, t2 = n2;
          if (i + 1u < triCount) { const float4* tn = sc.tri_w + (size_t)(triIdx + 1u) * 3; n0 = tn[0]; n1 = tn[1]; n2 = tn[2]; }
          float bx, by, bz;
          const float d = ray_tri(arx, ary, arz, cdx, cdy, cdz, t0, t1, t2, bx, by, bz);
          if (d < hitd) {
            hitd = d;
            flags |= F_FOUND;
            HitRec h; h.dist = d; h.bx = bx; h.by = by; h.bz = 1 - bx - by; h.blasIdx = blasIdx; h.triIdx = triIdx;   // rt_traversal.cpp:311-313
            A.hits[job] = h;                           // the record of the best hit so far: the last accept stands
            if (flags & F_ANYHIT) { stop = true; break; }
            if (!(path_m < hitd)) break;               // the reference abandons this subtree
          }
        }
        if (stop) { sp = 0; cur = DESC_DONE; } else pop_next();
        store_walk();
      }
    } else if (pass == P_INST) {
      if (act) {
        load_state();
        float ox, oy, oz, dx, dy, dz;
        world_ray(ox, oy, oz, dx, dy, dz);
        enter_instance(cur & PAYLOAD_MASK, ox, oy, oz, dx, dy, dz, true);
        store_ray();
        store_walk();
      }
    } else {
      // ---- service: rays that ended leave (a miss gets its record now), empty slots take new rays ----
      bool empty = false;
      if (act) {
        load_state();
        if (cur == DESC_DONE) {
          if (!(flags & F_FOUND)) { HitRec h; h.dist = RT_LARGE_FLOAT; h.bx = 0; h.by = 0; h.bz = 0; h.blasIdx = 0; h.triIdx = 0; A.hits[job] = h; }
          cur = DESC_IDLE;
        }
        empty = true;
      }
      const unsigned long long want = __ballot(empty);
      uint32_t n_want = (uint32_t)__popcll(want);
      uint32_t given = 0;                               // lanes want & ((1 << given) - 1) ... have their ray
      while (n_want > given && !queue_empty) {
        if (loc_next == loc_end) {                     // reserve the next chunk (the persistent kernel's queue: home shard = physical XCD, then the others)
          while (tries < QUEUE_SHARDS) {
            const uint32_t sid = (shard + tries) % QUEUE_SHARDS;
            const uint32_t s_lo = sid * per_shard;
            uint32_t in_range;
            asm volatile("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0 ; RTGUARD shard_range" : "=s"(in_range)
                         : "s"(__builtin_amdgcn_readfirstlane(s_lo)), "s"(__builtin_amdgcn_readfirstlane(n_jobs)) : "scc");
            if (!in_range) { ++tries; continue; }
            const uint32_t s_n = min(per_shard, n_jobs - s_lo);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(A.queue + sid * QUEUE_STRIDE, (uint32_t)RT_CHUNK);
            base = __shfl(base, 0);
            if (base < s_n) { loc_next = s_lo + base; loc_end = s_lo + min(base + (uint32_t)RT_CHUNK, s_n); break; }
            ++tries;
          }
          if (tries >= QUEUE_SHARDS) { queue_empty = true; break; }
        }
        const uint32_t take = min(n_want - given, loc_end - loc_next);
        const uint32_t rank = (uint32_t)__popcll(want & lt_mask);
        if (empty && rank >= given && rank < given + take) {
          job = loc_next + (rank - given);
          if (A.order) job = A.order[job];
          float ox, oy, oz, dx, dy, dz;
          world_ray(ox, oy, oz, dx, dy, dz);
          const float tmax_ = A.tmax ? A.tmax[job] : RT_LARGE_FLOAT;
          arx = ox; ary = oy; arz = oz;
          aix = 1.0f / dx; aiy = 1.0f / dy; aiz = 1.0f / dz;
          flags = F_WORLD | (A.any_hit ? F_ANYHIT : 0u);
          sp = 0;
          if (!ray_in_fast_domain(ox, oy, oz, aix, aiy, aiz)) defer();
          else {
            hitd = tmax_ > RT_LARGE_FLOAT ? RT_LARGE_FLOAT : tmax_;
            path_m = -__builtin_inff();
            cur = root_desc;
            nrays++;
            if (is_inst_desc(root_desc)) {
              const bool no_neg_zero = __float_as_uint(ox) != 0x80000000u && __float_as_uint(oy) != 0x80000000u && __float_as_uint(oz) != 0x80000000u;
              if (sc.ident_root && no_neg_zero) {       // (see start_ray of the persistent kernel: the object-space ray IS the world ray)
                flags &= ~F_WORLD;
                pool[PF_DX][slot] = __float_as_uint(dx); pool[PF_DY][slot] = __float_as_uint(dy); pool[PF_DZ][slot] = __float_as_uint(dz);
                pool[PF_BLAS][slot] = root_desc & PAYLOAD_MASK;
                cur = sc.blas_root[root_desc & PAYLOAD_MASK];
              } else enter_instance(root_desc & PAYLOAD_MASK, ox, oy, oz, dx, dy, dz, true);
            }
          }
          pool[PF_JOB][slot] = job;
          store_ray();
          empty = false;                               // (this lane's slot is taken -- or went to the deferral list and stays empty for the next pass)
        }
        given += take; loc_next += take;
      }
      if (act) store_walk();
    }
  }

  if (A.end_log) {
    unsigned s = nrays;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) {
      unsigned long long* w = A.end_log + 2ull * (blockIdx.x & 8191u);
      w[0] = wall_clock64();
      w[1] = (unsigned long long)s | ((unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 56);
    }
  }
  if (A.counters) {
    unsigned s = nrays;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0 && s) atomicAdd(A.counters, (unsigned long long)s);
  }
}

// ---------------------------------------------------------------------------------------------
// Two rays per lane (ray buffers only; round 5, VERDICT item 2, second form: the state stays in REGISTERS).
//
// What masks half the lanes of the persistent kernel on incoherent rays is the phase: every iteration runs the node body for the lanes whose
// ray is at a node (39.5 of 64 on the random rays) and, when enough have gathered, the leaf body for those at a leaf (22.7 of 64); the others
// wait.  The ray-pool kernel above cures that by moving every ray's state through LDS, which costs more than it saves.  Here a lane OWNS two
// rays: the active one in the registers the bodies work on, the parked one in a second set.  Before a body runs, a lane whose active ray is
// not in that body's phase but whose parked ray is exchanges the two (v_swap_b32 under the lane's execution mask: 14 instructions for the
// wavefront, whoever swaps) -- so the node body sees a lane unless NEITHER of its rays is at a node.  Nothing moves through memory: each
// ray keeps its own stack rows in LDS and scratch (selected by the half the lane is working on), an accepted hit goes straight to the
// ray's record.  Per-ray arithmetic is the persistent kernel's, step by step; only the interleaving of independent rays differs.
// ---------------------------------------------------------------------------------------------
#ifndef RT_PAIR_LSTK
#define RT_PAIR_LSTK 5          // stack entries of each of a lane's two rays kept in LDS
#endif
#ifndef RT_PAIR_WAVES
#define RT_PAIR_WAVES 5         // wavefronts per SIMD the kernel is compiled for
#endif
#ifndef RT_PAIR_LEAF_MIN
#define RT_PAIR_LEAF_MIN 32     // the leaf body runs once this many lanes hold a leaf in either ray (or no lane holds a node)
#endif
#ifndef RT_PAIR_DEAD_MAX
#define RT_PAIR_DEAD_MAX 32     // of the wavefront's 128 ray slots: finished / empty ones are serviced once there are this many
#endif

template <bool LDEXP, bool SHALLOW>
__global__ __launch_bounds__(64, RT_PAIR_WAVES) void rt_pair_trace_kernel(SceneDev sc, PersistArgs A) {
  constexpr int L = RT_PAIR_LSTK;
  constexpr int CAP = SHALLOW ? 3 * RT_SHALLOW_LEVELS : 3 * RT_MAX_LEVELS + L;
  constexpr int OVF = CAP - L;
  __shared__ uint2 s_stk[2][L][64];
  __shared__ uint32_t s_ctx[2][4][64];     // per ray: object-space direction (triangle tests), instance index
  const uint32_t lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t n_jobs = A.total_dev ? min(*A.total_dev, A.total) : A.total;
  const uint32_t per_shard = A.total_dev ? (((n_jobs + QUEUE_SHARDS - 1) / QUEUE_SHARDS + 63u) & ~63u) : A.per_shard;
  const uint32_t root_desc = sc.tlas_root;
  const uint32_t root_blas_desc = is_inst_desc(root_desc) ? sc.blas_root[root_desc & PAYLOAD_MASK] : DESC_DONE;
  const uint32_t xcc_id = RT_XCC_HOME ? (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) : blockIdx.x;
  const uint32_t shard = (xcc_id + A.shard_rot) % QUEUE_SHARDS;
  uint32_t tries = 0, loc_next = 0, loc_end = 0;
  bool queue_empty = false;
  unsigned nrays = 0;
  // the active ray (what the bodies work on) and the parked one; `half` = which of the lane's two stack / context rows the active ray owns
  float arx = 0, ary = 0, arz = 0, aix = 0, aiy = 0, aiz = 0, hitd = 0, path_m = 0;
  uint32_t cur = DESC_IDLE, job = 0, flags = 0, sp = 0, half = 0;
  float q_arx = 0, q_ary = 0, q_arz = 0, q_aix = 0, q_aiy = 0, q_aiz = 0, q_hitd = 0, q_path_m = 0;
  uint32_t q_cur = DESC_IDLE, q_job = 0, q_flags = 0, q_sp = 0, q_half = 1;
  uint2 ovf[2 * OVF];
#define PSWAPF(a, b) asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b))
  auto swap_rays = [&]() {       // (called under a divergent condition: the lanes that take the branch exchange their two rays)
    PSWAPF(arx, q_arx); PSWAPF(ary, q_ary); PSWAPF(arz, q_arz); PSWAPF(aix, q_aix); PSWAPF(aiy, q_aiy); PSWAPF(aiz, q_aiz);
    PSWAPF(hitd, q_hitd); PSWAPF(path_m, q_path_m); PSWAPF(cur, q_cur); PSWAPF(job, q_job); PSWAPF(flags, q_flags); PSWAPF(sp, q_sp); PSWAPF(half, q_half);
  };
  auto push = [&](uint32_t d, float m) {
    if (sp < (uint32_t)L) s_stk[half][sp][lane] = make_uint2(d, __float_as_uint(m));
    else ovf[half * OVF + (sp - L)] = make_uint2(d, __float_as_uint(m));
    ++sp;
  };
  auto pop_next = [&]() {
    cur = DESC_DONE;
    while (sp > 0u) {
      --sp;
      const uint2 e = sp < (uint32_t)L ? s_stk[half][sp][lane] : ovf[half * OVF + (sp - L)];
      if (__uint_as_float(e.y) < hitd) { cur = e.x; path_m = __uint_as_float(e.y); break; }
    }
  };
  auto defer = [&]() {
    const uint32_t q = atomicAdd(A.defer_count, 1u);
    if (q < A.defer_cap) A.defer_list[q] = job;
    cur = DESC_IDLE;
  };
  auto world_ray = [&](float& ox, float& oy, float& oz, float& dx, float& dy, float& dz) {
    const float* rp = A.rays + (size_t)job * 6;
    ox = rp[0]; oy = rp[1]; oz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5];
  };
  auto enter_instance = [&](uint32_t blasIdx, float ox, float oy, float oz, float dx, float dy, float dz) {   // rt_traversal.cpp:109-121, :231-261
    const uint32_t* bp = sc.blas + (size_t)blasIdx * (RT_BLAS_STRIDE / 4);
    uint32_t bw[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) bw[i] = bp[i];
    const float m00 = __uint_as_float(bw[1]), m01 = __uint_as_float(bw[2]), m02 = __uint_as_float(bw[3]), m03 = __uint_as_float(bw[4]);
    const float m10 = __uint_as_float(bw[5]), m11 = __uint_as_float(bw[6]), m12 = __uint_as_float(bw[7]), m13 = __uint_as_float(bw[8]);
    const float m20 = __uint_as_float(bw[9]), m21 = __uint_as_float(bw[10]), m22 = __uint_as_float(bw[11]), m23 = __uint_as_float(bw[12]);
    arx = m00 * ox + m01 * oy + m02 * oz + m03;
    ary = m10 * ox + m11 * oy + m12 * oz + m13;
    arz = m20 * ox + m21 * oy + m22 * oz + m23;
    const float cdx = m00 * dx + m01 * dy + m02 * dz;
    const float cdy = m10 * dx + m11 * dy + m12 * dz;
    const float cdz = m20 * dx + m21 * dy + m22 * dz;
    aix = 1.0f / cdx; aiy = 1.0f / cdy; aiz = 1.0f / cdz;
    if (!ray_in_fast_domain(arx, ary, arz, aix, aiy, aiz)) { nrays--; defer(); return; }   // (the EXACT launch counts the ray when it starts it again)
    flags &= ~F_WORLD;
    s_ctx[half][0][lane] = __float_as_uint(cdx); s_ctx[half][1][lane] = __float_as_uint(cdy); s_ctx[half][2][lane] = __float_as_uint(cdz);
    s_ctx[half][3][lane] = blasIdx;
    cur = sc.blas_root[blasIdx];
  };
  auto dead = [&](uint32_t d, bool more_jobs) { return d == DESC_DONE || (more_jobs && d == DESC_IDLE); };

  for (;;) {
    // ================= service: rays that ended leave (a miss gets its record), empty slots take new rays =================
    const bool more_jobs = !(queue_empty && loc_next == loc_end);
    const uint32_t n_dead = (uint32_t)(__popcll(__ballot(dead(cur, more_jobs))) + __popcll(__ballot(dead(q_cur, more_jobs))));
    const bool any_work = __ballot(is_work_desc(cur) || is_work_desc(q_cur)) != 0ull;
    if (n_dead >= (uint32_t)RT_PAIR_DEAD_MAX || (!any_work && n_dead)) {
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {        // the active rays, then (everything exchanged) the parked ones; two exchanges restore the order
        if (cur == DESC_DONE) {
          if (!(flags & F_FOUND)) { HitRec m; m.dist = RT_LARGE_FLOAT; m.bx = 0; m.by = 0; m.bz = 0; m.blasIdx = 0; m.triIdx = 0; A.hits[job] = m; }
          cur = DESC_IDLE;
        }
        const unsigned long long want = __ballot(cur == DESC_IDLE);
        const uint32_t n_want = (uint32_t)__popcll(want);
        uint32_t given = 0;
        while (n_want > given && !queue_empty) {
          if (loc_next == loc_end) {
            while (tries < QUEUE_SHARDS) {
              const uint32_t sid = (shard + tries) % QUEUE_SHARDS;
              const uint32_t s_lo = sid * per_shard;
              uint32_t in_range;
              asm volatile("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0 ; RTGUARD shard_range" : "=s"(in_range)
                           : "s"(__builtin_amdgcn_readfirstlane(s_lo)), "s"(__builtin_amdgcn_readfirstlane(n_jobs)) : "scc");
              if (!in_range) { ++tries; continue; }
              const uint32_t s_n = min(per_shard, n_jobs - s_lo);
              uint32_t base = 0;
              if (lane == 0) base = atomicAdd(A.queue + sid * QUEUE_STRIDE, (uint32_t)RT_TRACE_CHUNK);
              base = __shfl(base, 0);
              if (base < s_n) { loc_next = s_lo + base; loc_end = s_lo + min(base + (uint32_t)RT_TRACE_CHUNK, s_n); break; }
              ++tries;
            }
            if (tries >= QUEUE_SHARDS) { queue_empty = true; break; }
          }
          const uint32_t take = min(n_want - given, loc_end - loc_next);
          const uint32_t rank = (uint32_t)__popcll(want & lt_mask);
          if (cur == DESC_IDLE && ((want >> lane) & 1ull) && rank >= given && rank < given + take) {
            job = loc_next + (rank - given);
            if (A.order) job = A.order[job];
            float ox, oy, oz, dx, dy, dz;
            world_ray(ox, oy, oz, dx, dy, dz);
            const float tmax_ = A.tmax ? A.tmax[job] : RT_LARGE_FLOAT;
            arx = ox; ary = oy; arz = oz;
            aix = 1.0f / dx; aiy = 1.0f / dy; aiz = 1.0f / dz;
            flags = F_WORLD | (A.any_hit ? F_ANYHIT : 0u);
            sp = 0;
            if (!ray_in_fast_domain(ox, oy, oz, aix, aiy, aiz)) defer();
            else {
              hitd = tmax_ > RT_LARGE_FLOAT ? RT_LARGE_FLOAT : tmax_;
              path_m = -__builtin_inff();
              cur = root_desc;
              nrays++;
              if (is_inst_desc(root_desc)) {
                const bool no_neg_zero = __float_as_uint(ox) != 0x80000000u && __float_as_uint(oy) != 0x80000000u && __float_as_uint(oz) != 0x80000000u;
                if (sc.ident_root && no_neg_zero) {     // (see start_ray of the persistent kernel: the object-space ray IS the world ray)
                  flags &= ~F_WORLD;
                  s_ctx[half][0][lane] = __float_as_uint(dx); s_ctx[half][1][lane] = __float_as_uint(dy); s_ctx[half][2][lane] = __float_as_uint(dz);
                  s_ctx[half][3][lane] = root_desc & PAYLOAD_MASK;
                  cur = root_blas_desc;
                } else enter_instance(root_desc & PAYLOAD_MASK, ox, oy, oz, dx, dy, dz);
              }
            }
          }
          given += take; loc_next += take;
        }
        swap_rays();
      }
    }
    if (__ballot(is_work_desc(cur) || is_work_desc(q_cur)) == 0ull) {
      if (queue_empty && loc_next == loc_end) break;
      continue;
    }

    // ================= instance steps (TLAS leaves of multi-instance scenes): whichever of a lane's rays holds one =================
    if (__ballot(is_inst_desc(cur) || is_inst_desc(q_cur)) != 0ull) {
      if (!is_inst_desc(cur) && is_inst_desc(q_cur)) swap_rays();
      if (is_inst_desc(cur)) {
        float ox, oy, oz, dx, dy, dz;
        world_ray(ox, oy, oz, dx, dy, dz);
        enter_instance(cur & PAYLOAD_MASK, ox, oy, oz, dx, dy, dz);
      }
    }
    // ================= node body: a lane takes part unless NEITHER of its rays is at a node =================
    if (!is_node_desc(cur) && is_node_desc(q_cur)) swap_rays();
    if (is_node_desc(cur)) {
      const bool top = (cur >> 30) == DK_TLAS;
      if (top && !(flags & F_WORLD)) {                 // back at TLAS level after an instance (multi-instance scenes only)
        float dx, dy, dz;
        world_ray(arx, ary, arz, dx, dy, dz);
        aix = 1.0f / dx; aiy = 1.0f / dy; aiz = 1.0f / dz;
        flags |= F_WORLD;
      }
      const uint32_t ni = cur & PAYLOAD_MASK;
      const uint4* np = sc.nodes_c + (size_t)ni * CNODE_VEC4;
      const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
      const uint32_t* ref_node = nullptr;
      if (LDEXP) ref_node = top ? sc.ref_tlas + (size_t)ni * RT_NODE_DWORDS : sc.ref_bvh + (size_t)(ni - sc.n_tlas) * RT_NODE_DWORDS;
      Cand c[4];
      eval_children<false, LDEXP>(q0, q1, q2, q3, ref_node, arx, ary, arz, aix, aiy, aiz, hitd, c);
      order_children(c);
      if (c[0].d < __builtin_inff()) {
        bool more = true;
        if (sp + 3u > (uint32_t)CAP) { atomicOr(A.status, STATUS_STACK_OVERFLOW); more = false; }
        if (more && c[3].d < __builtin_inff()) push(c[3].desc, vmax_nonan(path_m, c[3].d));   // far first (:98-103)
        if (more && c[2].d < __builtin_inff()) push(c[2].desc, vmax_nonan(path_m, c[2].d));
        if (more && c[1].d < __builtin_inff()) push(c[1].desc, vmax_nonan(path_m, c[1].d));
        cur = c[0].desc;
        path_m = vmax_nonan(path_m, c[0].d);
      } else pop_next();
    }
    // ================= leaf body: once enough lanes hold a leaf in either ray, or no lane has a node left =================
    const unsigned long long leafm = __ballot(is_leaf_desc(cur) || is_leaf_desc(q_cur));
    if (leafm != 0ull && ((uint32_t)__popcll(leafm) >= (uint32_t)RT_PAIR_LEAF_MIN ||
                          __ballot(is_node_desc(cur) || is_node_desc(q_cur) || is_inst_desc(cur) || is_inst_desc(q_cur)) == 0ull)) {
      if (!is_leaf_desc(cur) && is_leaf_desc(q_cur)) swap_rays();
      if (is_leaf_desc(cur)) {
        uint32_t leftFirst = cur & LEAF_FIRST_MASK, triCount = (cur >> LEAF_FIRST_BITS) & LEAF_MAX_INLINE;
        if (triCount == 0u) { const uint32_t* rn = sc.ref_bvh + (size_t)leftFirst * RT_NODE_DWORDS; leftFirst = rn[4]; triCount = rn[5]; }
        const float cdx = __uint_as_float(s_ctx[half][0][lane]), cdy = __uint_as_float(s_ctx[half][1][lane]), cdz = __uint_as_float(s_ctx[half][2][lane]);
        const uint32_t blasIdx = s_ctx[half][3][lane];
        bool stop = false;
        float4 n0, n1, n2;
        { const float4* tp0 = sc.tri_w + (size_t)leftFirst * 3; n0 = tp0[0]; n1 = tp0[1]; n2 = tp0[2]; }
        for (uint32_t i = 0; i < triCount; ++i) {
          const uint32_t triIdx = leftFirst + i;
          const float4 t0 = n0, t1 = n1, t2 = n2;
          if (i + 1u < triCount) { const float4* tn = sc.tri_w + (size_t)(triIdx + 1u) * 3; n0 = tn[0]; n1 = tn[1]; n2 = tn[2]; }
          float bx, by, bz;
          const float d = ray_tri(arx, ary, arz, cdx, cdy, cdz, t0, t1, t2, bx, by, bz);
          if (d < hitd) {
            hitd = d;
            flags |= F_FOUND;
            HitRec hr; hr.dist = d; hr.bx = bx; hr.by = by; hr.bz = 1 - bx - by; hr.blasIdx = blasIdx; hr.triIdx = triIdx;   // rt_traversal.cpp:311-313
            A.hits[job] = hr;                          // the record of the best hit so far: the last accept stands
            if (flags & F_ANYHIT) { stop = true; break; }
            if (!(path_m < hitd)) break;               // the reference abandons this subtree
          }
        }
        if (stop) { sp = 0; cur = DESC_DONE; } else pop_next();
      }
    }
  }
#undef PSWAPF
  if (A.counters) {
    unsigned s = nrays;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0 && s) atomicAdd(A.counters, (unsigned long long)s);
  }
}

// Tile order for the next frame of a context: within each queue shard's band of tiles (a contiguous part of the
// frame, whose tiles share BVH nodes in the L2 of the XCD that works on it), most expensive first; cost = 100 MHz
// clocks the tile occupied its wavefront in the frame just traced.  A launch ends when its last tile ends, and a
// wavefront only gets about five tiles of a 1080p frame: starting the expensive ones first leaves the cheap ones for
// the tail.  One workgroup of 256 threads per shard, counting sort over 2048 monotone cost classes (5-bit exponent, 6-bit
// mantissa) with a parallel scan.  Runs as the FIRST workgroups of the shading launch (rt_shade_kernel): as a launch of its
// own between the traversal and the shading pass it sat on the critical path of a serial frame for 21 us, most of it one
// thread scanning the 2048 counters (profiles/r02_d_exact_timeline.txt).
// Measured and rejected: one global order dealt round robin to the shards (-4 %: loses the band -> XCD locality) and
// bands cut at equal cost instead of equal size (-7 %: the measured cost of a tile includes the contention on its SIMD).
// `base_order` (optional): the static order the queue positions have without learning (the band-major order of a batch of frames); the
// tiles of queue positions [lo, hi) are then base_order[lo..hi), and it is those that are sorted into order[lo..hi).
//
// Measured and rejected in round 4 (profiles/r04_k_pool_lpt_ab.txt): "cheap tiles last" -- each band's tiles split at one cost threshold, the
// cheap ones (a quarter of the frame's cost) handed out only when every band's expensive tiles are gone, so that what the wavefronts hold
// when the queue runs dry is a cheap tile: serial frames -9 %, a rank's sets of frames -2 .. -5 %.  The end of a launch is not long because of
// WHICH tiles are last: per-wavefront logs show every XCD doing the same number of loop iterations, and one half of the XCDs taking 20 % more
// clocks for each -- whatever band it traces (tools/wave_balance_batch.py, profiles/r04_l_xcd.txt).
__device__ __forceinline__ uint32_t lpt_cls(uint32_t c) {       // monotone cost class: 5-bit exponent, 6-bit mantissa
  if (c < 64u) return c;                                  // exponents 0..5 collapse onto the small values
  const uint32_t e = 31u - (uint32_t)__clz((int)c);       // 6..31
  return ((e - 5u) << 6) | ((c >> (e - 6u)) & 63u);       // 64 .. 1727
}
__device__ void lpt_order_block(uint32_t shard, const uint32_t* __restrict__ cost, uint32_t* __restrict__ order,
                                uint32_t n_tiles, uint32_t tiles_per_shard, uint32_t* hist /* LDS, 2048 + 8 words */,
                                const uint32_t* __restrict__ base_order = nullptr) {
  const uint32_t lo = shard * tiles_per_shard;
  const uint32_t hi = min(lo + tiles_per_shard, n_tiles);
  if (lo >= hi) return;   // (block-uniform)
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < 2048u; i += 256u) hist[i] = 0u;
  __syncthreads();
  for (uint32_t t = lo + threadIdx.x; t < hi; t += 256u) atomicAdd(&hist[2047u - lpt_cls(cost[base_order ? base_order[t] : t])], 1u);   // descending
  __syncthreads();
  // exclusive scan of the 2048 counters: 8 consecutive counters per thread, wavefront scan, then the four wavefront totals
  uint32_t v[8], sum = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = hist[threadIdx.x * 8u + k]; sum += v[k]; }
  uint32_t inc = sum;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += y; }
  if (lane == 63u) hist[2048u + wave] = inc;
  __syncthreads();
  uint32_t base = inc - sum;
  for (uint32_t w = 0; w < wave; ++w) base += hist[2048u + w];
#pragma unroll
  for (int k = 0; k < 8; ++k) { hist[threadIdx.x * 8u + k] = base; base += v[k]; }
  __syncthreads();
  for (uint32_t t = lo + threadIdx.x; t < hi; t += 256u) {
    const uint32_t tile = base_order ? base_order[t] : t;
    order[lo + atomicAdd(&hist[2047u - lpt_cls(cost[tile])], 1u)] = tile;
  }
}

// Deferred shading pass: one thread per pixel of rows [y0,y1), x fastest, so hit records are read
// and pixels written fully coalesced.
template <bool STATS>
__global__ __launch_bounds__(256) void rt_shade_kernel(SceneDev sc, ShadeParams p, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1, uint32_t row_step,
                                                      uint32_t n_rows, const float* __restrict__ utab, const float* __restrict__ vtab,
                                                      const HitRec* __restrict__ hb, uint32_t* __restrict__ dst,
                                                      HitRec* __restrict__ hits, float* __restrict__ colors,
                                                      unsigned long long* counters, uint32_t* __restrict__ ctl_reset,
                                                      uint32_t lpt_blocks, const uint32_t* __restrict__ lpt_cost, uint32_t* __restrict__ lpt_order,
                                                      uint32_t lpt_tiles, uint32_t lpt_per_shard,
                                                      uint32_t batch = 1, const ShadeParams* __restrict__ pbatch = nullptr, uint64_t dst_frame_stride = 0,
                                                      const uint32_t* __restrict__ lpt_base = nullptr) {
  // the first lpt_blocks workgroups sort the frame's tiles by cost for the context's next frame (see lpt_order_block)
  __shared__ uint32_t s_hist[2048 + 8];
  if (blockIdx.x < lpt_blocks) { lpt_order_block(blockIdx.x, lpt_cost, lpt_order, lpt_tiles, lpt_per_shard, s_hist, lpt_base); return; }
  const uint32_t blk = blockIdx.x - lpt_blocks;
  // last kernel of a frame: every user of the frame's control block (queue counters, deferral count)
  // has finished, so zero it here for the context's next frame instead of paying fill launches per frame
  if (ctl_reset && blk == 0)
    for (uint32_t i = threadIdx.x; i < CTL_DWORDS; i += 256u) ctl_reset[i] = 0u;
  const uint64_t t = (uint64_t)blk * 256u + threadIdx.x;
  const uint64_t n = (uint64_t)W * n_rows * batch;   // local rows of the window (its tile rows x 8; rows past y1 are skipped), per frame of the batch
  unsigned ntex = 0, npix = 0;
  const uint32_t x = (uint32_t)(t % W), vr = (uint32_t)(t / W);   // vr: row of the batch's stacked windows
  const uint32_t frame = batch > 1 ? vr / n_rows : 0u, lr = vr - frame * n_rows, y = frame_row(lr, y0, row_step);
  if (batch > 1 && t < n) { p = pbatch[frame]; dst += (size_t)frame * dst_frame_stride; }
  if (t < n && y < y1) {
    const size_t idx = (size_t)x + (size_t)y * W;
    HitRec h = hb[hit_index(x, vr, (W + 7u) >> 3)];   // tile-major, 192 contiguous bytes per 8 pixels of a row
    const uint32_t occ_bit = h.blasIdx & 0x80000000u;
    const bool occ = occ_bit != 0u;
    h.blasIdx &= 0x7fffffffu;
    const bool found = h.dist != RT_LARGE_FLOAT;
    float ox, oy, oz, dx, dy, dz;
    generate_ray(utab[x], vtab[y], ox, oy, oz, dx, dy, dz);
    float r, g, b;
    shade_eval<STATS>(sc, p, ox, oy, oz, dx, dy, dz, h, found, occ, r, g, b, &ntex);
    dst[idx] = pack_rgb8(r, g, b);
    if (hits) { HitRec o = h; o.blasIdx |= occ_bit; hits[idx] = o; }   // bit 31 of blasIdx: the pixel's occlusion ray was blocked
    if (colors) { colors[3 * idx] = r; colors[3 * idx + 1] = g; colors[3 * idx + 2] = b; }
    npix = 1;
  }
  if (STATS && counters) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned v[2] = {ntex, npix};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned s = v[k];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
      if (lane == 0 && s) atomicAdd(counters + 5 + k, (unsigned long long)s);
    }
  }
}

// closest-hit / miss shader of arbitrary (ray, hit record) pairs: what the RTU test's shaders compute for the ray a payload belongs
// to (closest.cpp:57-127 without a secondary ray, miss.cpp:9-14), e.g. for the hit records vxrt_trace returns
__global__ __launch_bounds__(256) void rt_shade_rays_kernel(SceneDev sc, ShadeParams p, uint64_t n, const float* __restrict__ rays,
                                                           const HitRec* __restrict__ hits, float* __restrict__ colors, uint32_t* __restrict__ rgb8) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float* rp = rays + i * 6;
  HitRec h = hits[i];
  h.blasIdx &= 0x7fffffffu;
  float r, g, b;
  shade_eval<false>(sc, p, rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], h, h.dist != RT_LARGE_FLOAT, false, r, g, b);
  if (colors) { colors[3 * i] = r; colors[3 * i + 1] = g; colors[3 * i + 2] = b; }
  if (rgb8) rgb8[i] = pack_rgb8(r, g, b);
}

// ---------------------------------------------------------------------------------------------
// Mirror bounce (closest.cpp:95-121) as a wavefront over depth levels.  The reference recurses inside
// the closest-hit shader: C(ray) = term + (reflectivity > 0 && bounce + 1 < max_depth ? C(mirror ray)
// : background) * reflectivity, C(miss) = background.  Here level k holds the rays of bounce k (level 0 =
// the pixels); shading a level appends the next level's rays, and the colours are folded back from the
// deepest level to the pixels in the reference's order of operations, so the result has the same bits.
// Taken only when max_depth > 1 and some instance is reflective (the shipped scene builder has none).
// ---------------------------------------------------------------------------------------------
// Shade level `level`.  LEVEL0: entry = pixel of rows [y0,y1), hit record from the traversal (occlusion
// in bit 31 of blasIdx); else entry i = ray rays[6i..] with hit hits[i] (occluded iff shits[i] hit).
// Entries that bounce leave (term, reflectivity) in term[] and append a ray; the others are final.
template <bool LEVEL0>
__global__ __launch_bounds__(256) void rt_shade_bounce_kernel(SceneDev sc, ShadeParams p, uint32_t level, uint64_t n,
    uint32_t W, uint32_t y0, const float* __restrict__ utab, const float* __restrict__ vtab,
    const HitRec* __restrict__ hb, const float* __restrict__ rays, const HitRec* __restrict__ shits,
    float4* __restrict__ term, float* __restrict__ col, uint32_t* __restrict__ dst, HitRec* __restrict__ hits_out,
    float* __restrict__ colors_out, uint32_t* next_count, float* __restrict__ next_rays, uint32_t* __restrict__ next_parent,
    uint32_t* __restrict__ ctl_reset) {
  if (ctl_reset && blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < CTL_DWORDS; i += 256u) ctl_reset[i] = 0u;
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= n) return;
  size_t e = (size_t)t;
  float ox, oy, oz, dx, dy, dz;
  HitRec h;
  bool occ;
  if (LEVEL0) {
    const uint32_t x = (uint32_t)(t % W), y = y0 + (uint32_t)(t / W);
    e = (size_t)x + (size_t)y * W;
    h = hb[hit_index(x, y - y0, (W + 7u) >> 3)];
    occ = (h.blasIdx & 0x80000000u) != 0u;
    if (hits_out) hits_out[e] = h;
    h.blasIdx &= 0x7fffffffu;
    generate_ray(utab[x], vtab[y], ox, oy, oz, dx, dy, dz);
  } else {
    const float* rp = rays + e * 6;
    ox = rp[0]; oy = rp[1]; oz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5];
    h = hb[e];
    occ = shits != nullptr && shits[e].dist != RT_LARGE_FLOAT;
  }
  float r, g, b;
  bool final_ = true;
  if (h.dist == RT_LARGE_FLOAT) {   // miss.cpp:9-14
    r = p.bg[0]; g = p.bg[1]; b = p.bg[2];
  } else {
    float refl, Ix, Iy, Iz, Nx, Ny, Nz;
    shade_terms<false>(sc, p, ox, oy, oz, dx, dy, dz, h, occ, r, g, b, refl, Ix, Iy, Iz, Nx, Ny, Nz);
    if (refl > 0.0f && level + 1u < p.max_depth) {   // :95
      final_ = false;
      term[e] = make_float4(r, g, b, refl);
      const uint32_t slot = atomicAdd(next_count, 1u);   // every level has room for one ray per entry of the level before
      mirror_ray(dx, dy, dz, Ix, Iy, Iz, Nx, Ny, Nz, next_rays + (size_t)slot * 6);
      next_parent[slot] = (uint32_t)e;
    } else {
      float thr = 1.0f;
      thr *= refl;                    // :90
      r = r + p.bg[0] * thr;          // :123
      g = g + p.bg[1] * thr;
      b = b + p.bg[2] * thr;
    }
  }
  if (final_) {
    if (LEVEL0) {
      dst[e] = pack_rgb8(r, g, b);
      if (colors_out) { colors_out[3 * e] = r; colors_out[3 * e + 1] = g; colors_out[3 * e + 2] = b; }
    } else {
      col[3 * e] = r; col[3 * e + 1] = g; col[3 * e + 2] = b;
    }
  }
}

// occlusion rays of a bounce level (shadow extension at every depth); a miss gets a ray nothing can hit
__global__ __launch_bounds__(256) void rt_bounce_shadow_rays_kernel(ShadeParams p, uint32_t n, const float* __restrict__ rays,
    const HitRec* __restrict__ hits, float* __restrict__ srays, float* __restrict__ stmax, unsigned long long* rays_traced) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  bool real = false;
  if (i < n) {
    const float* rp = rays + (size_t)i * 6;
    float* sp = srays + (size_t)i * 6;
    const float d = hits[i].dist;
    if (d == RT_LARGE_FLOAT) {
      sp[0] = 0.f; sp[1] = 0.f; sp[2] = 0.f; sp[3] = 1.f; sp[4] = 1.f; sp[5] = 1.f;
      stmax[i] = -1.0f;
    } else {
      float sdist;
      shadow_ray(p, rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], d, sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], sdist);
      stmax[i] = sdist;
      real = true;
    }
  }
  const unsigned long long m = __ballot(real);
  if (rays_traced && (threadIdx.x & 63u) == 0u && m) atomicAdd(rays_traced, (unsigned long long)__popcll(m));
}

// fold level k into level k-1 (closest.cpp:117): C[parent] = term[parent] + C_k * (1 * reflectivity[parent])
template <bool TO_PIXELS>
__global__ __launch_bounds__(256) void rt_bounce_unwind_kernel(uint32_t n, const uint32_t* __restrict__ parent, const float* __restrict__ col_k,
    const float4* __restrict__ term_prev, float* __restrict__ col_prev, uint32_t* __restrict__ dst, float* __restrict__ colors_out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const size_t q = parent[i];
  const float4 t = term_prev[q];
  float thr = 1.0f;
  thr *= t.w;
  const float r = t.x + col_k[3 * (size_t)i] * thr, g = t.y + col_k[3 * (size_t)i + 1] * thr, b = t.z + col_k[3 * (size_t)i + 2] * thr;
  if (TO_PIXELS) {
    dst[q] = pack_rgb8(r, g, b);
    if (colors_out) { colors_out[3 * q] = r; colors_out[3 * q + 1] = g; colors_out[3 * q + 2] = b; }
  } else {
    col_prev[3 * q] = r; col_prev[3 * q + 1] = g; col_prev[3 * q + 2] = b;
  }
}

// ---------------------------------------------------------------------------------------------
// Ambient occlusion (extension, BASELINE config 5; recipe defined in oracle/rt_oracle.c:orc_ao_ray and
// mirrored here operation by operation -- RNG of common.h:129-147, rejection-sampled disk, Duff basis:
// only IEEE add/mul/div/sqrt, so host and device produce the same rays).
// ---------------------------------------------------------------------------------------------
// per pixel of rows [y0,y1): Lambert colour of the primary hit (else arm of closest.cpp), hit point and
// shading normal for the occlusion rays; geo[t] = (I, hit?), nrm[t] = (N, 0), col[t] = (rgb, 0), cnt[t] = 0;
// pixels with a hit are appended to list[] (count in hdr[0]; the order is arbitrary, nothing depends on it)
#define AO_PREP_CHUNKS 4   // pixels per thread of rt_ao_prepare_kernel: one list-append atomic per 1,024 pixels
__global__ __launch_bounds__(256) void rt_ao_prepare_kernel(SceneDev sc, ShadeParams p, uint64_t n, uint32_t W, uint32_t y0,
    const float* __restrict__ utab, const float* __restrict__ vtab, const HitRec* __restrict__ hb,
    float4* __restrict__ geo, float4* __restrict__ nrm, float4* __restrict__ col, uint32_t* __restrict__ cnt,
    uint32_t* __restrict__ list, uint32_t* hdr, uint32_t* ctl_reset, float4* __restrict__ alb /* optional: albedo of the hit */) {
  if (ctl_reset && blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < CTL_DWORDS; i += 256u) ctl_reset[i] = 0u;
  // The hit pixels are appended to one list.  One atomic per wavefront on the list's counter (32,400 for a 1080p frame, all on one
  // address, ~10 ns apart) was 0.3 of the kernel's 0.37 ms: a workgroup now counts the hits of 1,024 pixels in LDS and appends once.
  __shared__ uint32_t s_cnt[AO_PREP_CHUNKS][4];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  bool hit_c[AO_PREP_CHUNKS];
  uint32_t off_c[AO_PREP_CHUNKS];
#pragma unroll
  for (int c = 0; c < AO_PREP_CHUNKS; ++c) {
    const uint64_t t = ((uint64_t)blockIdx.x * AO_PREP_CHUNKS + c) * 256u + threadIdx.x;
    bool hit = false;
    if (t < n) {
      const uint32_t x = (uint32_t)(t % W), y = y0 + (uint32_t)(t / W);
      HitRec h = hb[hit_index(x, y - y0, (W + 7u) >> 3)];
      h.blasIdx &= 0x7fffffffu;
      float ox, oy, oz, dx, dy, dz;
      generate_ray(utab[x], vtab[y], ox, oy, oz, dx, dy, dz);
      float r, g, b;
      if (h.dist == RT_LARGE_FLOAT) {
        r = p.bg[0]; g = p.bg[1]; b = p.bg[2];
        geo[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        nrm[t] = make_float4(0.f, 0.f, 1.f, 0.f);
      } else {
        float refl, Ix, Iy, Iz, Nx, Ny, Nz, a3[3];
        shade_terms<false>(sc, p, ox, oy, oz, dx, dy, dz, h, false, r, g, b, refl, Ix, Iy, Iz, Nx, Ny, Nz, nullptr, a3);
        float thr = 1.0f;
        thr *= refl;
        r = r + p.bg[0] * thr; g = g + p.bg[1] * thr; b = b + p.bg[2] * thr;
        geo[t] = make_float4(Ix, Iy, Iz, 1.0f);
        nrm[t] = make_float4(Nx, Ny, Nz, 0.f);
        if (alb) alb[t] = make_float4(a3[0], a3[1], a3[2], 0.f);
        hit = true;
      }
      col[t] = make_float4(r, g, b, 0.f);
      cnt[t] = 0u;
    }
    const unsigned long long m = __ballot(hit);
    hit_c[c] = hit;
    off_c[c] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[c][wv] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  __shared__ uint32_t s_base;
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int c = 0; c < AO_PREP_CHUNKS; ++c) for (int w = 0; w < 4; ++w) { const uint32_t v = s_cnt[c][w]; s_cnt[c][w] = tot; tot += v; }
    s_base = tot ? atomicAdd(hdr, tot) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < AO_PREP_CHUNKS; ++c)
    if (hit_c[c]) list[s_base + s_cnt[c][wv] + off_c[c]] = (uint32_t)(((uint64_t)blockIdx.x * AO_PREP_CHUNKS + c) * 256u + threadIdx.x);
}

// samples [s0, s0 + ns) of every listed pixel: ray i = (pixel list[i / ns], sample s0 + i % ns); hdr[1] = number of rays
__global__ __launch_bounds__(256) void rt_ao_rays_kernel(uint64_t cap, uint32_t W, uint32_t y0, const float* __restrict__ utab, const float* __restrict__ vtab,
    const float4* __restrict__ geo, const float4* __restrict__ nrm, const uint32_t* __restrict__ list, uint32_t* hdr,
    uint32_t spp, uint32_t s0, uint32_t ns, uint32_t user_seed, float radius, float* __restrict__ rays, float* __restrict__ tmax) {
  const uint64_t total = (uint64_t)hdr[0] * ns;
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i == 0) hdr[1] = (uint32_t)total;
  if (i >= total || i >= cap) return;
  const uint32_t t = list[i / ns], smp = s0 + (uint32_t)(i % ns);
  float* o = rays + (size_t)i * 6;
  const float4 gI = geo[t];
  const uint32_t x = t % W, y = y0 + t / W;
  float ox, oy, oz, vdx, vdy, vdz;
  generate_ray(utab[x], vtab[y], ox, oy, oz, vdx, vdy, vdz);
  const float4 gN = nrm[t];
  float r6[6];
  ao_sample_ray(x, y, W, spp, smp, user_seed, gI.x, gI.y, gI.z, gN.x, gN.y, gN.z, vdx, vdy, vdz, r6);
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = r6[k];
  tmax[i] = radius;
}

__global__ __launch_bounds__(256) void rt_ao_accumulate_kernel(uint64_t cap, const uint32_t* __restrict__ list, const uint32_t* __restrict__ hdr, uint32_t ns,
    const HitRec* __restrict__ ohits, uint32_t* __restrict__ cnt) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const bool live = i < hdr[1] && i < cap;
  // the ns samples of a pixel sit next to each other: the lanes of a wavefront that belong to one pixel add up with a ballot and
  // the first of them does the pixel's atomic (16 spp: 4 atomics per wavefront instead of up to 64)
  const unsigned long long m = __ballot(live && ohits[live ? i : 0].dist == RT_LARGE_FLOAT);
  if (!live) return;
  const uint32_t lane = threadIdx.x & 63u, k = (uint32_t)(i % ns);
  const uint32_t s0 = lane > k ? lane - k : 0u, e0 = min(63u, lane - k + ns - 1u);   // lanes of this pixel in this wavefront (lane - k may wrap: then s0 = 0)
  const uint32_t e = lane >= k ? e0 : min(63u, lane + (ns - 1u - k));
  if (lane == s0) {
    const unsigned long long seg = (~0ull >> (63u - e)) & (~0ull << s0);
    const uint32_t c = (uint32_t)__popcll(m & seg);
    if (c) atomicAdd(cnt + list[i / ns], c);
  }
}

__global__ __launch_bounds__(256) void rt_ao_final_kernel(uint64_t n, uint32_t W, uint32_t y0, const float4* __restrict__ geo, const float4* __restrict__ col,
    const uint32_t* __restrict__ cnt, uint32_t spp, uint32_t* __restrict__ dst, float* __restrict__ colors_out, uint32_t* __restrict__ unoccluded,
    unsigned long long* rays_traced) {
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  bool hit = false;
  if (t < n) {
    const uint32_t x = (uint32_t)(t % W), y = y0 + (uint32_t)(t / W);
    const size_t e = (size_t)x + (size_t)y * W;
    const float4 c = col[t];
    float r = c.x, g = c.y, b = c.z;
    uint32_t open = 0u;
    if (geo[t].w != 0.f) {
      hit = true;
      open = cnt[t];
      const float f = (float)open / (float)spp;
      r *= f; g *= f; b *= f;
    }
    dst[e] = pack_rgb8(r, g, b);
    if (colors_out) { colors_out[3 * e] = r; colors_out[3 * e + 1] = g; colors_out[3 * e + 2] = b; }
    if (unoccluded) unoccluded[e] = open;
  }
  const unsigned long long m = __ballot(hit);
  if (rays_traced && (threadIdx.x & 63u) == 0u && m) atomicAdd(rays_traced, (unsigned long long)__popcll(m) * spp);
}

// ---------------------------------------------------------------------------------------------
// Secondary rays re-sorted before they are traced (north_star: "ray packets re-sorted ... to tame divergence").
// The rays of a bounce / AO pass leave the compaction in pixel order with directions spread over a hemisphere: a wavefront of
// 64 consecutive rays shares origins but not directions.  Counting sort by key = direction octant x origin cell (the 16x16-pixel
// cell of the ray's pixel: hit points of neighbouring pixels are neighbours in space), so that 64 consecutive queue positions
// hold rays that start in one small region AND point into the same octant.  Only the ORDER in which rays are traced changes:
// the kernel reads ray order[q] and writes hit record order[q], so every result is bit-identical (tests compare them all).
// ---------------------------------------------------------------------------------------------
#define BIN_CELL_SHIFT 4u   // 16 x 16 pixels
__global__ __launch_bounds__(256) void rt_bin_count_kernel(uint64_t cap, const uint32_t* __restrict__ hdr, const float* __restrict__ rays,
    const uint32_t* __restrict__ list, uint32_t ns, uint32_t W, uint32_t cells_x, uint32_t n_cells, uint32_t* __restrict__ hist, uint32_t* __restrict__ keys) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= hdr[1] || i >= cap) return;
  const uint32_t t = list[i / ns];
  const uint32_t cell = ((t / W) >> BIN_CELL_SHIFT) * cells_x + ((t % W) >> BIN_CELL_SHIFT);
  const float* r = rays + i * 6;
  const uint32_t oct = (__float_as_uint(r[3]) >> 31) | ((__float_as_uint(r[4]) >> 31) << 1) | ((__float_as_uint(r[5]) >> 31) << 2);
  const uint32_t key = oct * n_cells + cell;
  keys[i] = key;
  atomicAdd(&hist[key], 1u);
}
// exclusive scan of hist[0..n) in place, one workgroup of 1024 threads, `per` consecutive counters per thread
__global__ __launch_bounds__(1024) void rt_bin_scan_kernel(uint32_t* __restrict__ hist, uint32_t n, uint32_t per) {
  __shared__ uint32_t wsum[16];
  const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n);
  uint32_t sum = 0;
  for (uint32_t k = lo; k < hi; ++k) sum += hist[k];
  uint32_t inc = sum;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += y; }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = inc - sum;
  for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
  for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = hist[k]; hist[k] = base; base += v; }
}
__global__ __launch_bounds__(256) void rt_bin_scatter_kernel(uint64_t cap, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ keys,
                                                            uint32_t* __restrict__ hist, uint32_t* __restrict__ order) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= hdr[1] || i >= cap) return;
  order[atomicAdd(&hist[keys[i]], 1u)] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------
// Reference-quirks traversal (opt-in; vxrt_trace_reference_quirks).  The kernels above implement the CANONICAL algorithm (DESIGN.md
// s3), which returns what the reference's RTU returns wherever the reference addresses its own data.  The RTU does not always:
// children of a TLAS internal node that is popped from the short stack are addressed relative to the LAST BLAS's base_ptr
// (rt_traversal.cpp:91-92 after :119-120), so with a TLAS deeper than one level it reads unrelated memory and loses real hits.
// What it then returns depends on what lies at those addresses, i.e. on the memory layout -- so this mode works as the
// simulator does: on ONE flat memory image addressed with 32-bit offsets (the simulated RAM) and the four base pointers of the
// RTX DCRs, and it restates BVHTraverser::traverse literally: trail[32], the 5-entry short stack that forgets its oldest entry,
// the restart from the root when the stack is dry, a full re-descent after every accepted candidate (rt_unit.cpp:199-202),
// libstdc++ min/max in the slab test, the per-test edge subtractions.  One thread per ray: this is a compatibility mode, not a
// fast path.  Reads outside the image return zeros.  rt_traversal.cpp:80-86 spins 2^32 times without effect when
// trail[level] == 4 and no child passes: closed form here.  Depth > 32 overruns trail[] in the reference: status bit, ray ends.
// ---------------------------------------------------------------------------------------------
struct QuirkImage { const uint8_t* mem; uint64_t size; uint32_t tlas_ptr, blas_ptr, bvh_ptr, tri_ptr; };

__device__ __forceinline__ void q_read(const QuirkImage& im, uint32_t* dst, uint32_t addr, uint32_t dwords) {
  if ((uint64_t)addr + 4ull * dwords > im.size || (addr & 3u)) { for (uint32_t i = 0; i < dwords; ++i) dst[i] = 0u; return; }
  const uint32_t* p = (const uint32_t*)(im.mem + addr);
  for (uint32_t i = 0; i < dwords; ++i) dst[i] = p[i];
}

__global__ __launch_bounds__(64) void rt_quirks_trace_kernel(QuirkImage im, const float* __restrict__ rays, const float* __restrict__ tmax, uint64_t n,
                                                            HitRec* __restrict__ out, int any_hit_first, uint32_t* status) {
  const uint64_t r = (uint64_t)blockIdx.x * 64u + threadIdx.x;
  if (r >= n) return;
  const float* rp = rays + r * 6;
  const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
  HitRec hit; hit.dist = tmax ? tmax[r] : RT_LARGE_FLOAT; hit.bx = 0; hit.by = 0; hit.bz = 0; hit.blasIdx = 0; hit.triIdx = 0;
  if (hit.dist > RT_LARGE_FLOAT) hit.dist = RT_LARGE_FLOAT;
  uint8_t trail[RT_MAX_TRAIL];
  for (int i = 0; i < RT_MAX_TRAIL; ++i) trail[i] = 0;
  // ShortStack<TraversalStackEntry, 5> (types.h:1808-1840)
  uint32_t ss_ptr[5]; uint8_t ss_last[5]; uint32_t ss_head = 0, ss_count = 0;
  auto ss_push = [&](uint32_t ptr, uint8_t last) {
    if (ss_count < 5u) ss_count++;            // (a full stack overwrites its oldest entry: head wraps onto it)
    ss_ptr[ss_head] = ptr; ss_last[ss_head] = last;
    ss_head = (ss_head + 1u) % 5u;
  };
  bool accepted = false;
  bool limit = true;
  for (uint32_t guard = 0; guard < (1u << 16); ++guard) {   // one pass of traverse() per accepted candidate (a ray accepts a handful)
    uint32_t level = 0, base_ptr = im.tlas_ptr, node_ptr = im.tlas_ptr, blasIdx = 0;
    float cx = ox, cy = oy, cz = oz, cdx = dx, cdy = dy, cdz = dz;   // cur_ray
    bool finished = false, pending = false;
    float pending_dist = 0.f;
    // findNextParentLevel + pop (rt_traversal.cpp:171-213); true = traversal over
    auto pop = [&]() -> bool {
      int parent = -1;
      for (int i = (int)level - 1; i >= 0; --i) if (i < RT_MAX_TRAIL && trail[i] != 4) { parent = i; break; }
      if (parent < 0) return true;
      trail[parent]++;
      for (int i = parent + 1; i < RT_MAX_TRAIL; ++i) trail[i] = 0;
      if (ss_count == 0u) { base_ptr = im.tlas_ptr; node_ptr = im.tlas_ptr; level = 0; }
      else {
        ss_head = ss_head == 0u ? 4u : ss_head - 1u;
        ss_count--;
        node_ptr = ss_ptr[ss_head];
        if (ss_last[ss_head]) trail[parent] = 4;
        level = (uint32_t)parent + 1u;
      }
      return false;
    };
    uint32_t it = 0;
    for (; it < ITER_LIMIT && !finished && !pending; ++it) {
      uint32_t w[RT_NODE_DWORDS];
      q_read(im, w, node_ptr, RT_NODE_DWORDS);
      const uint32_t imask = w[3] >> 24, leftFirst = w[4], leafData = w[5];
      const bool top = imask == 1u;
      const bool leaf = top ? (leafData != 0xffffffffu) : (leafData != 0u);
      if (!leaf) {
        const float px = __uint_as_float(w[0]), py = __uint_as_float(w[1]), pz = __uint_as_float(w[2]);
        const int ex = (int)(int8_t)(w[3] & 0xff), ey = (int)(int8_t)((w[3] >> 8) & 0xff), ez = (int)(int8_t)((w[3] >> 16) & 0xff);
        const uint8_t* cb = (const uint8_t*)w + 24;
        float dist[4]; uint32_t child[4]; int cnt = 0;
        const float rox = top ? ox : cx, roy = top ? oy : cy, roz = top ? oz : cz;
        const float rdx = top ? dx : cdx, rdy = top ? dy : cdy, rdz = top ? dz : cdz;
        const float ix = 1.0f / rdx, iy = 1.0f / rdy, iz = 1.0f / rdz;
        for (int k = 0; k < 4; ++k) {
          const uint8_t* c = cb + 7 * k;
          if (c[0] == 0) continue;
          const float d = ray_box<true>(rox, roy, roz, ix, iy, iz,
                                        px + ldexpf((float)c[1], ex), py + ldexpf((float)c[2], ey), pz + ldexpf((float)c[3], ez),
                                        px + ldexpf((float)c[4], ex), py + ldexpf((float)c[5], ey), pz + ldexpf((float)c[6], ez));
          if (d < hit.dist) {
            // std::sort(a.dist > b.dist) on <= 4 elements = insertion sort: stable, farthest first (:76-78)
            int j = cnt;
            while (j > 0 && d > dist[j - 1]) { dist[j] = dist[j - 1]; child[j] = child[j - 1]; --j; }
            dist[j] = d; child[j] = (uint32_t)k;
            cnt++;
          }
        }
        if (level >= RT_MAX_TRAIL) { atomicOr(status, STATUS_STACK_OVERFLOW); finished = true; break; }
        const uint32_t kdrop = trail[level];
        const uint32_t drop = kdrop == 4u ? (uint32_t)cnt - 1u : kdrop;     // wraps when cnt == 0 (:81)
        if (drop >= (uint32_t)cnt) cnt = 0; else cnt -= (int)drop;
        if (cnt == 0) finished = pop();
        else {
          const uint32_t nearest = child[cnt - 1];
          cnt--;
          node_ptr = base_ptr + (leftFirst + nearest) * RT_NODE_BYTES;          // base_ptr may be STALE here: the quirk
          if (cnt == 0) trail[level] = 4;
          else for (int q = 0; q < cnt; ++q) ss_push(base_ptr + (leftFirst + child[q]) * RT_NODE_BYTES, q == 0 ? 1 : 0);
          level++;
        }
      } else if (top) {
        blasIdx = leafData;
        uint32_t bw[13];
        q_read(im, bw, im.blas_ptr + blasIdx * RT_BLAS_STRIDE, 13);
        const float m00 = __uint_as_float(bw[1]), m01 = __uint_as_float(bw[2]), m02 = __uint_as_float(bw[3]), m03 = __uint_as_float(bw[4]);
        const float m10 = __uint_as_float(bw[5]), m11 = __uint_as_float(bw[6]), m12 = __uint_as_float(bw[7]), m13 = __uint_as_float(bw[8]);
        const float m20 = __uint_as_float(bw[9]), m21 = __uint_as_float(bw[10]), m22 = __uint_as_float(bw[11]), m23 = __uint_as_float(bw[12]);
        cx = m00 * ox + m01 * oy + m02 * oz + m03; cy = m10 * ox + m11 * oy + m12 * oz + m13; cz = m20 * ox + m21 * oy + m22 * oz + m23;
        cdx = m00 * dx + m01 * dy + m02 * dz; cdy = m10 * dx + m11 * dy + m12 * dz; cdz = m20 * dx + m21 * dy + m22 * dz;
        base_ptr = im.bvh_ptr + bw[0] * RT_NODE_BYTES;
        node_ptr = base_ptr;
      } else {
        for (uint32_t i = 0; i < leafData && !pending; ++i) {
          const uint32_t triIdx = leftFirst + i;
          uint32_t tw[9];
          q_read(im, tw, im.tri_ptr + triIdx * RT_TRI_BYTES, 9);
          const float v0x = __uint_as_float(tw[0]), v0y = __uint_as_float(tw[1]), v0z = __uint_as_float(tw[2]);
          const float4 t0 = make_float4(v0x, v0y, v0z, __uint_as_float(tw[3]) - v0x);
          const float4 t1 = make_float4(__uint_as_float(tw[4]) - v0y, __uint_as_float(tw[5]) - v0z, __uint_as_float(tw[6]) - v0x, __uint_as_float(tw[7]) - v0y);
          const float4 t2 = make_float4(__uint_as_float(tw[8]) - v0z, 0.f, 0.f, 0.f);
          float bx, by, bz;
          const float d = ray_tri(cx, cy, cz, cdx, cdy, cdz, t0, t1, t2, bx, by, bz);
          if (d < hit.dist) {
            pending_dist = d; hit.bx = bx; hit.by = by; hit.bz = bz; hit.blasIdx = blasIdx; hit.triIdx = triIdx;
            ss_count = 0; ss_head = 0;     // :150-153 (an emptied stack pops nothing: head position is irrelevant)
            pending = true;
          }
        }
        if (!pending) finished = pop();
      }
    }
    if (!pending) { limit = !finished; break; }   // traversal completed -- or the iteration backstop ran out (reported below)
    hit.dist = pending_dist;                   // rt_unit.cpp:199-202 COMMIT_ACCEPT, then traverse again from the root with the kept trail
    accepted = true;
    if (any_hit_first) { limit = false; break; }
  }
  if (limit) atomicOr(status, STATUS_ITER_LIMIT);   // the walk did not end within the backstops (the reference would still be spinning)
  if (!accepted) { hit.dist = RT_LARGE_FLOAT; hit.bx = 0; hit.by = 0; hit.bz = 0; hit.blasIdx = 0; hit.triIdx = 0; }
  out[r] = hit;
}

// camera rays of rows [y0, y1) as a ray buffer (kernel.cpp:28-39; u, v in double as there -- IEEE division, so the same bits as the
// host-built tables of the frame kernels): ray (x, y) at index x + (y - y0) * W
__global__ __launch_bounds__(256) void rt_camera_rays_kernel(uint32_t W, uint32_t H, uint32_t y0, uint64_t n, float* __restrict__ rays) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t x = (uint32_t)(i % W), y = y0 + (uint32_t)(i / W);
  const float u = (float)(((double)x * 2.0 - (double)W) / (double)H), v = (float)(((double)y * 2.0 - (double)H) / (double)H);
  float* o = rays + i * 6;
  generate_ray(u, v, o[0], o[1], o[2], o[3], o[4], o[5]);
}

struct ShadeBatch { ShadeParams p[VXRT_MAX_BATCH]; };
__global__ void set_batch_params_kernel(ShadeBatch b, uint32_t n, ShadeParams* __restrict__ dst) {
  if (threadIdx.x < n) dst[threadIdx.x] = b.p[threadIdx.x];
}

// Image assembly of a frame split by interleaved tile rows (vxrt_wire_pack / vxrt_wire_unpack in the header): 0x00RRGGBB pixels as 3 bytes on
// the wire.  One thread per 4 pixels = 16 bytes in, 12 bytes out (three aligned words), rows of the share contiguous on the wire.
__global__ __launch_bounds__(256) void wire_pack_kernel(const uint32_t* __restrict__ frames, uint64_t frame_stride, uint32_t W4, uint32_t per, uint32_t world, uint32_t rank,
                                                        uint64_t n_quads, uint32_t* __restrict__ wire) {
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= n_quads) return;
  const uint32_t xq = (uint32_t)(t % W4);
  const uint64_t row = t / W4;                       // row of the wire: frame * per * 8 + j * 8 + y
  const uint32_t rows_per_frame = per * 8u;
  const uint32_t f = (uint32_t)(row / rows_per_frame), lr = (uint32_t)(row - (uint64_t)f * rows_per_frame);
  const uint32_t j = lr >> 3, y = lr & 7u;
  const uint4 p = *(const uint4*)(frames + (size_t)f * frame_stride + ((size_t)(j * world + rank) * 8u + y) * (W4 * 4u) + (size_t)xq * 4u);
  uint32_t* o = wire + t * 3u;
  o[0] = (p.x & 0xFFFFFFu) | (p.y << 24);
  o[1] = ((p.y >> 8) & 0xFFFFu) | (p.z << 16);
  o[2] = ((p.z >> 16) & 0xFFu) | (p.w << 8);
}
__global__ __launch_bounds__(256) void wire_unpack_kernel(const uint32_t* __restrict__ wire_all, uint64_t wire_stride_words, uint32_t W4, uint32_t per, uint32_t world,
                                                          uint64_t quads_per_rank, uint32_t* __restrict__ frames, uint64_t frame_stride) {
  const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= quads_per_rank * world) return;
  const uint32_t r = (uint32_t)(t / quads_per_rank);
  const uint64_t q = t - (uint64_t)r * quads_per_rank;
  const uint32_t xq = (uint32_t)(q % W4);
  const uint64_t row = q / W4;
  const uint32_t rows_per_frame = per * 8u;
  const uint32_t f = (uint32_t)(row / rows_per_frame), lr = (uint32_t)(row - (uint64_t)f * rows_per_frame);
  const uint32_t j = lr >> 3, y = lr & 7u;
  const uint32_t* w = wire_all + (size_t)r * wire_stride_words + q * 3u;
  const uint32_t a = w[0], b = w[1], c = w[2];
  uint4 p;
  p.x = a & 0xFFFFFFu;
  p.y = (a >> 24) | ((b & 0xFFFFu) << 8);
  p.z = (b >> 16) | ((c & 0xFFu) << 16);
  p.w = c >> 8;
  *(uint4*)(frames + (size_t)f * frame_stride + ((size_t)(j * world + r) * 8u + y) * (W4 * 4u) + (size_t)xq * 4u) = p;
}

__global__ void add_counter_kernel(unsigned long long* c, unsigned long long v) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(c, v); }

// ---------------------------------------------------------------------------------------------
// acceleration-layout build (one pass over the reference-format buffers; validates every index the
// traversal will follow so that a malformed scene is rejected on the host instead of faulting the GPU)
// ---------------------------------------------------------------------------------------------
// one thread per reference node of one buffer.  bases/ends: sorted BLAS node ranges (nb of them).
__global__ void accel_nodes_kernel(const uint32_t* __restrict__ ref, uint32_t n_nodes, uint4* __restrict__ out, int is_tlas,
                                   const uint32_t* __restrict__ bases, const uint32_t* __restrict__ ends, uint32_t nb,
                                   uint32_t n_tris, uint32_t n_blas, uint32_t bias, uint32_t* status) {
  // bias: compact index of this buffer's node 0 (0 for the TLAS pass, n_tlas for the BLAS pass); `out` is already offset by it
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const uint32_t* w = ref + (size_t)i * RT_NODE_DWORDS;
  const uint32_t imask = w[3] >> 24, leftFirst = w[4], leafData = w[5];
  uint32_t base = 0, end = n_nodes;
  if (!is_tlas) {
    bool in = false;
    for (uint32_t j = 0; j < nb; ++j) if (i >= bases[j] && i < ends[j]) { base = bases[j]; end = ends[j]; in = true; }
    if (!in) return;   // node outside every instance's range: unreachable, leave untouched
  }
  if (imask != (is_tlas ? 1u : 0u)) return;   // not a node of this kind (e.g. unused tail): reachable nodes are checked via their parent
  const bool leaf = is_tlas ? (leafData != 0xffffffffu) : (leafData != 0u);
  if (leaf) return;
  const float px = __uint_as_float(w[0]), py = __uint_as_float(w[1]), pz = __uint_as_float(w[2]);
  const float po[3] = {px, py, pz};
  const int ev[3] = {(int)(int8_t)(w[3] & 0xff), (int)(int8_t)((w[3] >> 8) & 0xff), (int)(int8_t)((w[3] >> 16) & 0xff)};
  const uint8_t* bytes = (const uint8_t*)w;
  uint32_t pay[4] = {DESC_NONE, DESC_NONE, DESC_NONE, DESC_NONE};   // complete descriptors of the children
  uint8_t qb[24];
  for (int k = 0; k < 4; ++k) {
    const uint8_t* c = bytes + 24 + 7 * k;
    for (int j = 0; j < 6; ++j) qb[4 * j + k] = c[1 + j];   // plane-major: word j = plane j (lo xyz, hi xyz) of the four children
    if (c[0] == 0) continue;   // meta (rt_traversal.cpp:60)
    // the sign-selected slab form needs lo <= hi per axis (child_box); an inverted box falls back to min/max
    if (c[1] > c[4] || c[2] > c[5] || c[3] > c[6]) atomicOr(status, STATUS_FMA_DECODE_DIFFERS);
    // fma decode must reproduce origin + ldexp(float(q), e) bit for bit (eval_children)
    for (int j = 0; j < 6; ++j) {
      const float q = (float)c[1 + j];
      const float a = po[j % 3] + ldexpf(q, ev[j % 3]);
      const float b = __fmaf_rn(q, ldexpf(1.0f, ev[j % 3]), po[j % 3]);
      if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) atomicOr(status, STATUS_FMA_DECODE_DIFFERS);
    }
    const uint64_t ci64 = (uint64_t)base + leftFirst + (uint32_t)k;   // calcNodePtr(base_ptr, leftFirst + childIdx), :91-92
    // children are allocated after their parent by the builders (bvh.cpp:94-97, 371-402): requiring
    // that makes every accepted tree acyclic, so traversal terminates
    if (ci64 >= end || ci64 <= i || ci64 + bias > PAYLOAD_MASK) { atomicOr(status, STATUS_BAD_SCENE); continue; }
    const uint32_t ci = (uint32_t)ci64;
    const uint32_t* cw = ref + (size_t)ci * RT_NODE_DWORDS;
    const uint32_t c_imask = cw[3] >> 24, c_lf = cw[4], c_ld = cw[5];
    if (c_imask != (is_tlas ? 1u : 0u)) { atomicOr(status, STATUS_BAD_SCENE); continue; }
    if (is_tlas) {
      if (c_ld != 0xffffffffu) {
        if (c_ld >= n_blas || c_ld >= 0x3FFFFFF0u) { atomicOr(status, STATUS_BAD_SCENE); continue; }
        pay[k] = DESC(DK_INST, c_ld);
      } else pay[k] = DESC(DK_TLAS, ci + bias);
    } else {
      if (c_ld != 0u) {
        if ((uint64_t)c_lf + c_ld > n_tris) { atomicOr(status, STATUS_BAD_SCENE); continue; }
        pay[k] = DESC(DK_LEAF, (c_ld <= LEAF_MAX_INLINE && c_lf <= LEAF_FIRST_MASK) ? ((c_ld << LEAF_FIRST_BITS) | c_lf) : ci);   // else by reference
        if (!(c_ld <= LEAF_MAX_INLINE && c_lf <= LEAF_FIRST_MASK) && ci > LEAF_FIRST_MASK) { atomicOr(status, STATUS_BAD_SCENE); pay[k] = DESC_NONE; }
      } else pay[k] = DESC(DK_BLAS, ci + bias);
    }
  }
  uint32_t qw[6];
  for (int v = 0; v < 6; ++v) qw[v] = (uint32_t)qb[4 * v] | ((uint32_t)qb[4 * v + 1] << 8) | ((uint32_t)qb[4 * v + 2] << 16) | ((uint32_t)qb[4 * v + 3] << 24);
  uint4* o = out + (size_t)i * CNODE_VEC4;
  o[0] = make_uint4(w[0], w[1], w[2], __float_as_uint(ldexpf(1.0f, ev[0])));
  o[1] = make_uint4(qw[0], qw[1], qw[2], qw[3]);
  o[2] = make_uint4(qw[4], qw[5], pay[0], pay[1]);
  o[3] = make_uint4(pay[2], pay[3], __float_as_uint(ldexpf(1.0f, ev[1])), __float_as_uint(ldexpf(1.0f, ev[2])));
}

__global__ void accel_tris_kernel(const float* __restrict__ tri, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* t = tri + (size_t)i * 9;
  const float v0x = t[0], v0y = t[1], v0z = t[2];
  // edge1 = v1 - v0, edge2 = v2 - v0 exactly as rt_traversal.cpp:272-278 computes them per test
  out[(size_t)i * 3 + 0] = make_float4(v0x, v0y, v0z, t[3] - v0x);
  out[(size_t)i * 3 + 1] = make_float4(t[4] - v0y, t[5] - v0z, t[6] - v0x, t[7] - v0y);
  out[(size_t)i * 3 + 2] = make_float4(t[8] - v0z, 0.f, 0.f, 0.f);
}

// root descriptors: thread 0 -> TLAS root, thread 1+j -> BLAS root of instance record j
__global__ void accel_roots_kernel(const uint32_t* __restrict__ tlas, const uint32_t* __restrict__ bvh, const uint32_t* __restrict__ blas,
                                   uint32_t n_tlas, uint32_t n_bvh, uint32_t n_blas, uint32_t n_tris, uint32_t* tlas_root, uint32_t* blas_root,
                                   uint32_t* status) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    const uint32_t imask = tlas[3] >> 24, ld = tlas[5];
    *tlas_root = DESC_DONE;
    if (imask != 1u) atomicOr(status, STATUS_BAD_SCENE);
    else if (ld != 0xffffffffu) {
      if (ld >= n_blas || ld >= 0x3FFFFFF0u) atomicOr(status, STATUS_BAD_SCENE);
      else *tlas_root = DESC(DK_INST, ld);
    } else *tlas_root = DESC(DK_TLAS, 0u);   // compact index 0
  } else if (t - 1 < n_blas) {
    const uint32_t j = t - 1;
    const uint32_t off = blas[(size_t)j * (RT_BLAS_STRIDE / 4)];
    blas_root[j] = DESC_DONE;
    if (off >= n_bvh || off > LEAF_FIRST_MASK) { atomicOr(status, STATUS_BAD_SCENE); return; }
    const uint32_t* w = bvh + (size_t)off * RT_NODE_DWORDS;
    const uint32_t imask = w[3] >> 24, lf = w[4], ld = w[5];
    if (imask != 0u) atomicOr(status, STATUS_BAD_SCENE);
    else if (ld != 0u) {
      if ((uint64_t)lf + ld > n_tris) atomicOr(status, STATUS_BAD_SCENE);
      else blas_root[j] = DESC(DK_LEAF, (ld <= LEAF_MAX_INLINE && lf <= LEAF_FIRST_MASK) ? ((ld << LEAF_FIRST_BITS) | lf) : off);
    } else if ((uint64_t)off + n_tlas > PAYLOAD_MASK) atomicOr(status, STATUS_BAD_SCENE);
    else blas_root[j] = DESC(DK_BLAS, off + n_tlas);
  }
}

// Depth of the scene in INTERNAL levels on a root-to-leaf path, TLAS and BLAS together: what bounds a lane's stack (a node step leaves at most
// three pending siblings; instance and leaf steps leave none).  One pass per level over the compact nodes (children lie after their parents, so
// the trees are acyclic): pass t gives every internal child of a node of level t - 1 the level t; a TLAS leaf hands its level on to the root of
// its instance's BLAS.  `deepest` ends as the last level any node reached.  Only reached nodes are read (unreached slots are not initialised).
__global__ void accel_depth_kernel(const uint4* __restrict__ nodes_c, uint32_t n_nodes, uint32_t tlas_root, const uint32_t* __restrict__ blas_root,
                                   uint32_t level, uint32_t* __restrict__ depth, uint32_t* __restrict__ deepest) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  auto reach = [&](uint32_t d) {
    if (is_inst_desc(d)) d = blas_root[d & PAYLOAD_MASK];
    if (!is_node_desc(d)) return;
    const uint32_t c = d & PAYLOAD_MASK;
    if (c < n_nodes) { atomicMax(&depth[c], level); *deepest = level; }   // (every writer of a pass stores the same value)
  };
  if (level == 1u) {
    if (i == 0) reach(tlas_root);
    return;
  }
  if (i >= n_nodes || depth[i] != level - 1u) return;
  const uint4* np = nodes_c + (size_t)i * CNODE_VEC4;
  const uint4 q2 = np[2], q3 = np[3];
  reach(q2.z); reach(q2.w); reach(q3.x); reach(q3.y);
}

// Top of the tree for LDS staging: breadth-first from the TLAS root through the instance roots, the first `cap` internal
// nodes get slots 0..n-1 (so the levels every ray walks come first).  The image holds their compact nodes as four planes of
// `cap` uint4 with the child descriptors of staged children rewritten to DESC_TOP_FLAG | slot; the *_top roots likewise.
// The global compact nodes stay untouched: kernels that do not stage (EXACT, ldexp decode) start from the plain roots and
// never meet a slot descriptor.  One wavefront; a few hundred nodes, once per scene.
__global__ __launch_bounds__(64) void accel_top_kernel(const uint4* __restrict__ nodes_c, uint32_t tlas_root, const uint32_t* __restrict__ blas_root,
                                                       uint32_t n_blas, uint32_t cap, uint4* __restrict__ img, uint32_t* __restrict__ out_n,
                                                       uint32_t* __restrict__ tlas_root_top, uint32_t* __restrict__ blas_root_top) {
  __shared__ uint32_t q[RT_TOP_MAX];
  __shared__ uint32_t n_s;
  const uint32_t lane = threadIdx.x;
  auto find = [&](uint32_t idx, uint32_t n) -> uint32_t {   // wave-wide search; returns slot or 0xFFFFFFFF
    uint32_t hit = 0xFFFFFFFFu;
    for (uint32_t b = 0; b < n; b += 64u) {
      const unsigned long long m = __ballot(b + lane < n && q[b + lane] == idx);
      if (m) { hit = b + (uint32_t)__ffsll((long long)m) - 1u; break; }
    }
    return hit;
  };
  uint32_t n = 0;
  auto push = [&](uint32_t d) {   // wave-uniform d
    if (!is_node_desc(d)) return;
    const uint32_t idx = d & PAYLOAD_MASK;
    if (n >= cap || find(idx, n) != 0xFFFFFFFFu) return;
    if (lane == 0) q[n] = idx;
    ++n;
    __syncthreads();
  };
  if (is_inst_desc(tlas_root)) push(blas_root[tlas_root & PAYLOAD_MASK]); else push(tlas_root);
  for (uint32_t head = 0; head < n && n < cap; ++head) {
    const uint4* np = nodes_c + (size_t)q[head] * CNODE_VEC4;
    const uint4 q2 = np[2], q3 = np[3];
    const uint32_t d[4] = {q2.z, q2.w, q3.x, q3.y};
    for (int k = 0; k < 4; ++k) {
      if (is_inst_desc(d[k])) push(blas_root[d[k] & PAYLOAD_MASK]); else push(d[k]);
    }
  }
  __syncthreads();
  auto patch = [&](uint32_t d) -> uint32_t {
    if (!is_node_desc(d)) return d;
    const uint32_t slot = find(d & PAYLOAD_MASK, n);
    return slot == 0xFFFFFFFFu ? d : ((d & 0xC0000000u) | DESC_TOP_FLAG | slot);
  };
  for (uint32_t sidx = 0; sidx < n; ++sidx) {
    const uint4* np = nodes_c + (size_t)q[sidx] * CNODE_VEC4;
    uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    q2.z = patch(q2.z); q2.w = patch(q2.w); q3.x = patch(q3.x); q3.y = patch(q3.y);
    if (lane == 0) { img[sidx] = q0; img[cap + sidx] = q1; img[2 * (size_t)cap + sidx] = q2; img[3 * (size_t)cap + sidx] = q3; }
  }
  for (uint32_t j = 0; j < n_blas; ++j) {
    const uint32_t v = patch(blas_root[j]);
    if (lane == 0) blas_root_top[j] = v;
  }
  const uint32_t tr = patch(tlas_root);
  if (lane == 0) { *tlas_root_top = tr; *out_n = n; }
  (void)n_s;
}

// shading inputs (closest.cpp:52-77 dereferences them unchecked; here a scene that would read outside its buffers is rejected
// when the acceleration layout is built): every triangle's texId names a material, and every textured material's texels lie
// inside the texture buffer with non-zero dimensions (texSample takes `% width`, rtx_shading.h:9-10)
__global__ void accel_check_shading_kernel(const rt_triex_t* __restrict__ triEx, uint32_t n_tris, const rt_material_t* __restrict__ mat, uint32_t n_mats,
                                           uint64_t tex_bytes, int have_tex, uint32_t* status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_tris && triEx[i].texId >= n_mats) atomicOr(status, STATUS_BAD_SCENE);
  if (i < n_mats && mat[i].diffuse_tex_id >= 0) {
    const uint64_t w = mat[i].tex_width, h = mat[i].tex_height, off = mat[i].tex_offset;
    if (!have_tex || w == 0 || h == 0 || (off & 3u) != 0 || off > tex_bytes || w * h > (tex_bytes - off) / 4u) atomicOr(status, STATUS_BAD_SCENE);
  }
}

// ---------------------------------------------------------------------------------------------
// host entry points (C ABI, include/vortex_hip.h level 2)
// ---------------------------------------------------------------------------------------------
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

static uint32_t* g_status[16] = {nullptr};

static uint32_t* status_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_status[dev]) {
    if (hipMalloc((void**)&g_status[dev], sizeof(uint32_t)) != hipSuccess) return nullptr;
    (void)hipMemset(g_status[dev], 0, sizeof(uint32_t));
  }
  return g_status[dev];
}

#define LPT_MIN_TILES 20000u
#define LPT_BATCH_MAX_TILES 100000u
#ifndef EXACT_GRID
#define EXACT_GRID 128   // workgroups of the EXACT launch (it sees a fraction of a percent of the rays)
#endif

// grid of a persistent launch: what the device holds at once (occupancy x CUs, queried once per kernel
// and device), capped by the job count
template <class K>
static uint32_t persistent_grid(K kernel, uint64_t jobs, int wg_threads = RT_WG_THREADS) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, uint64_t> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  uint64_t g = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({(const void*)kernel, dev});
    if (it != cache.end()) g = it->second;
  }
  if (!g) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, wg_threads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    // measurement knob (tools/occupancy_sweep.sh): fewer resident workgroups per CU than the kernel allows
    if (const char* e = getenv("VXRT_WGS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < per_cu) per_cu = v; }
    g = (uint64_t)per_cu * (uint64_t)cus;
    if (getenv("VXRT_DEBUG")) fprintf(stderr, "[vxrt] persistent grid: %d blocks/CU x %d CUs\n", per_cu, cus);
    std::lock_guard<std::mutex> lk(mu);
    cache[{(const void*)kernel, dev}] = g;
  }
  const uint64_t need = (jobs + (uint64_t)wg_threads - 1) / (uint64_t)wg_threads;
  if (g > need) g = need;
  return (uint32_t)(g ? g : 1);
}

// Mutable per-frame state.  An accel owns up to MAX_FRAMES_IN_FLIGHT of these and hands them out round
// robin, so that renders issued on different streams overlap on the GPU (the tail of one persistent
// launch, where most wavefronts have drained, is filled by the head of the next frame's launch); a
// context is handed out again only behind the event of its previous render.
#define MAX_FRAMES_IN_FLIGHT 8
struct FrameCtx {
  void* hitbuf = nullptr;      // W*H hit records between the traversal and the shading pass
  uint64_t hitbuf_pixels = 0;
  uint32_t* defer = nullptr;   // job list of the EXACT launch
  uint64_t defer_cap = 0;
  uint32_t* ctl = nullptr;     // control block (CTL_DWORDS), zero between frames
  ShadeParams* pbatch = nullptr;   // per-frame shading parameters of a batch launch (VXRT_MAX_BATCH entries)
  bool ctl_dirty = false;      // a call failed after touching it: clear before the next use
  // mirror-bounce levels (allocated on first use; level 0 only holds `term`, one entry per pixel)
  struct Level {
    float* rays = nullptr; HitRec* hits = nullptr; uint32_t* parent = nullptr; float4* term = nullptr; float* col = nullptr;
    float* srays = nullptr; float* stmax = nullptr; HitRec* shits = nullptr;
    uint64_t cap = 0; uint32_t n = 0;
  };
  std::vector<Level> lv;
  uint32_t* bcount = nullptr;  // device: rays appended to the level being built
  // tile cost of the last frame and the order derived from it (render jobs, see lpt_order_kernel)
  // one slot per batch size (slot 1 = single frames): a frame loop that alternates batch sizes keeps what it learned for each
  struct Lpt { uint32_t* cost = nullptr; uint32_t* order = nullptr; uint32_t cap = 0; uint32_t key[6] = {0, 0, 0, 0, 0, 0}; bool valid = false; };
  Lpt lpt[VXRT_MAX_BATCH + 1];
  // ambient-occlusion pass (allocated on first use), one entry per pixel of the window
  float4* ao_geo = nullptr; float4* ao_nrm = nullptr; float4* ao_col = nullptr; uint32_t* ao_cnt = nullptr;
  uint32_t* ao_list = nullptr; uint32_t* ao_hdr = nullptr;   // pixels with a hit; [0] their number, [1] rays of the current batch
  float* ao_rays = nullptr; float* ao_tmax = nullptr; HitRec* ao_hits = nullptr; uint64_t ao_cap = 0, ao_ray_cap = 0;
  uint32_t* bin_hist = nullptr; uint32_t* bin_keys = nullptr; uint32_t* bin_order = nullptr; uint64_t bin_cap = 0, bin_ray_cap = 0;   // secondary-ray binning
  void* pool_spill = nullptr; uint64_t pool_spill_bytes = 0;   // ray-pool trace kernel: the part of the slots' stacks that does not fit LDS
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_side = nullptr, ev_done = nullptr;
  bool busy = false, inited = false, done_recorded = false;
  hipStream_t last_stream = nullptr;
};

struct vxrt_accel {
  SceneDev dev{};
  vxrt_scene_t ref{};
  void* nodes_c = nullptr; void* tri_w = nullptr; void* blas_root = nullptr;
  void* top_img = nullptr; uint32_t* top_roots = nullptr;   // LDS-staged top of the tree: image; [0] n, [1] TLAS root, [2..] BLAS roots
  FrameCtx ctx[MAX_FRAMES_IN_FLIGHT];
  uint32_t n_ctx = 1, next_ctx = 0;
  bool stream_seen = false, multi_stream = false; hipStream_t first_stream = nullptr;   // (see release_ctx)
  float* uvtab = nullptr;      // camera tables: u[W] then v[H]
  uint32_t uv_w = 0, uv_h = 0;
  // camera pixels whose primary ray has a zero direction component (u == 0 or v == 0): listed on the
  // host per (W, H, y0, y1) and traced by an EXACT launch on a side stream, concurrently with the main one
  uint32_t* apriori = nullptr; // [0] count, [1..] job ids
  uint32_t* batch_order[VXRT_MAX_BATCH + 1] = {}; uint32_t bo_tiles = 0;   // band-major tile order of a batch of k frames of bo_tiles tiles each, per k
  uint32_t ap_count = 0, ap_key[6] = {0, 0, 0, 0, 0, 0};
  uint64_t ap_cap = 0;
  float max_reflectivity = 0.0f;   // over the instance records: > 0 enables the mirror-bounce path
  unsigned long long* trace_wave_log = nullptr;   // diagnostic (vxrt_debug_trace_wave_log): per-wavefront log of the counting build's ray-buffer launches
  unsigned long long* end_log = nullptr;   // diagnostic (vxrt_debug_end_log): where the main launches leave their wavefronts' end times
  uint32_t levels = 0;             // internal levels on the longest root-to-leaf path (TLAS + BLAS), counted up to RT_SHALLOW_LEVELS + 1
  bool shallow = false;            // levels <= RT_SHALLOW_LEVELS: the timed launches take the SHALLOW instantiations
  int device = 0;
};

static void accel_free(vxrt_accel* a) {
  if (!a) return;
  (void)hipDeviceSynchronize();
  (void)hipFree(a->nodes_c); (void)hipFree(a->tri_w); (void)hipFree(a->blas_root);
  (void)hipFree(a->top_img); (void)hipFree(a->top_roots);
  (void)hipFree(a->uvtab); (void)hipFree(a->apriori);
  for (uint32_t k = 0; k <= VXRT_MAX_BATCH; ++k) (void)hipFree(a->batch_order[k]);
  for (FrameCtx& c : a->ctx) {
    (void)hipFree(c.hitbuf); (void)hipFree(c.defer); (void)hipFree(c.ctl); (void)hipFree(c.bcount);
    for (FrameCtx::Lpt& l : c.lpt) { (void)hipFree(l.cost); (void)hipFree(l.order); }
    (void)hipFree(c.ao_geo); (void)hipFree(c.ao_nrm); (void)hipFree(c.ao_col); (void)hipFree(c.ao_cnt); (void)hipFree(c.ao_rays); (void)hipFree(c.ao_tmax); (void)hipFree(c.ao_hits); (void)hipFree(c.ao_list); (void)hipFree(c.ao_hdr);
    (void)hipFree(c.bin_hist); (void)hipFree(c.bin_keys); (void)hipFree(c.bin_order);
    for (FrameCtx::Level& l : c.lv) {
      (void)hipFree(l.rays); (void)hipFree(l.hits); (void)hipFree(l.parent); (void)hipFree(l.term); (void)hipFree(l.col);
      (void)hipFree(l.srays); (void)hipFree(l.stmax); (void)hipFree(l.shits);
    }
    (void)hipFree(c.pbatch); (void)hipFree(c.pool_spill);
    if (c.side) (void)hipStreamDestroy(c.side);
    if (c.ev_in) (void)hipEventDestroy(c.ev_in);
    if (c.ev_side) (void)hipEventDestroy(c.ev_side);
    if (c.ev_done) (void)hipEventDestroy(c.ev_done);
  }
  delete a;
}

// next frame context, ordered on `s` behind its previous use
static FrameCtx* acquire_ctx(vxrt_accel* a, hipStream_t s) {
  if (!a->stream_seen) { a->stream_seen = true; a->first_stream = s; } else if (s != a->first_stream) a->multi_stream = true;
  // the context this stream used last (no event hop: the stream orders the two frames), else one never used, else round robin.
  // (Plain round robin pairs contexts with streams only while the caller's stream rotation and the call count stay in step: an odd
  // number of warm-up frames was enough to put every later frame behind a cross-stream event wait, -3 %.)
  uint32_t pick = a->n_ctx;
  for (uint32_t k = 0; k < a->n_ctx && pick == a->n_ctx; ++k) { const uint32_t i = (a->next_ctx + k) % a->n_ctx; if (a->ctx[i].busy && a->ctx[i].last_stream == s) pick = i; }
  for (uint32_t k = 0; k < a->n_ctx && pick == a->n_ctx; ++k) { const uint32_t i = (a->next_ctx + k) % a->n_ctx; if (!a->ctx[i].busy) pick = i; }
  if (pick == a->n_ctx) pick = a->next_ctx % a->n_ctx;
  a->next_ctx = pick + 1;
  FrameCtx& c = a->ctx[pick];
  if (!c.inited) {   // (a failed attempt is completed by the next one: every piece is created only if still missing)
    // the side stream carries the small EXACT launch over the a-priori list: highest priority, so that its few workgroups are
    // placed before the main launch fills every CU (an EXACT workgroup cannot co-reside with a full persistent grid: LDS)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (!c.side && hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, hi) != hipSuccess) return nullptr;
    if (!c.ev_in && hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (!c.ev_side && hipEventCreateWithFlags(&c.ev_side, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (!c.ev_done && hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (!c.ctl && hipMalloc((void**)&c.ctl, CTL_DWORDS * sizeof(uint32_t)) != hipSuccess) return nullptr;
    if (hipMemset(c.ctl, 0, CTL_DWORDS * sizeof(uint32_t)) != hipSuccess) return nullptr;
    c.inited = true;
  }
  if (c.busy && c.last_stream != s) {   // (same stream: already ordered)
    if (c.done_recorded) { if (hipStreamWaitEvent(s, c.ev_done, 0) != hipSuccess) return nullptr; }
    else if (hipStreamSynchronize(c.last_stream) != hipSuccess) {
      // (the accel's first call on a second stream: see release_ctx.)  The first stream may have been destroyed by its owner in the
      // meantime -- destroying a stream completes its work, but the handle is stale: order behind the whole device instead, and only
      // a failure of that is a failure of the call (the context must not stay unusable behind a dead handle)
      (void)hipGetLastError();
      if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    }
    c.busy = false; c.last_stream = nullptr;
  }
  if (c.ctl_dirty) {
    if (hipMemsetAsync(c.ctl, 0, CTL_DWORDS * sizeof(uint32_t), s) != hipSuccess) return nullptr;
    c.ctl_dirty = false;
  }
  return &c;
}

// The completion event orders a context's next use on ANOTHER stream.  An accel that has only ever seen one stream (serial frames,
// the vx_* sequence) does not pay for it -- an event record is a barrier packet on the stream, ~10 us between a frame's shading
// launch and the next frame's traversal -- and the first call on a second stream waits for the first stream on the host instead.
static int release_ctx(vxrt_accel* a, FrameCtx* c, hipStream_t s) {
  c->done_recorded = a->multi_stream || a->n_ctx > 1;
  if (c->done_recorded && hipEventRecord(c->ev_done, s) != hipSuccess) return -1;
  c->busy = true; c->last_stream = s;
  return 0;
}

extern "C" uint32_t* vxrt_status_word_device(void) { return status_word(); }   // shared with rc_kernels.hip (not part of the public header)

// The longest-first tile order of a frame for the software twin's launch (rc_kernels.hip; internal, not part of the public header): the same
// per-band counting sort the RTU path's shading launch carries (lpt_order_block), as a launch of QUEUE_SHARDS workgroups that also zeroes the
// `clear_dwords` words at `clear` (the twin's queue counters, for its next frame).
__global__ __launch_bounds__(256) void lpt_sort_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n_tiles, uint32_t tiles_per_shard,
                                                       uint32_t* __restrict__ clear, uint32_t clear_dwords) {
  __shared__ uint32_t s_hist[2048 + 8];
  if (clear && blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < clear_dwords; i += 256u) clear[i] = 0u;
  lpt_order_block(blockIdx.x, cost, order, n_tiles, tiles_per_shard, s_hist, nullptr);
}
extern "C" int vxrt_internal_lpt_sort(const uint32_t* cost, uint32_t* order, uint32_t n_tiles, uint32_t tiles_per_shard, uint32_t* clear, uint32_t clear_dwords, void* stream) {
  hipLaunchKernelGGL(lpt_sort_kernel, dim3(QUEUE_SHARDS), dim3(256), 0, (hipStream_t)stream, cost, order, n_tiles, tiles_per_shard, clear, clear_dwords);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" {

const char* vxrt_version(void) { return "vortex-rt-mi355x 0.3 (gfx950, compact 64-byte nodes, persistent wavefronts)"; }

int vxrt_accel_build(const vxrt_scene_t* s, void* stream, vxrt_accel_t** out) {
  if (!s || !out || !s->tlas || !s->blas || !s->bvh || !s->tri) return -1;
  if (s->n_tlas_nodes == 0 || s->n_blas == 0 || s->n_bvh_nodes == 0 || s->n_tris == 0) return -1;
  if ((uint64_t)s->n_tlas_nodes + s->n_bvh_nodes > PAYLOAD_MASK || s->n_tris >= 0x7fffffffu) return -1;   // one compact index space
  hipStream_t st = (hipStream_t)stream;
  // instance node ranges (host side, n_blas is small): sorted unique bvh_offsets
  std::vector<uint32_t> recs((size_t)s->n_blas * (RT_BLAS_STRIDE / 4));
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  if (hipMemcpy(recs.data(), s->blas, recs.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  std::vector<uint32_t> bases;
  float max_refl = 0.0f;
  for (uint32_t j = 0; j < s->n_blas; ++j) {
    const uint32_t off = recs[(size_t)j * (RT_BLAS_STRIDE / 4)];
    if (off >= s->n_bvh_nodes) return -1;
    bases.push_back(off);
    float refl;
    memcpy(&refl, &recs[(size_t)j * (RT_BLAS_STRIDE / 4) + 38], sizeof(float));   // blas_node_t::reflectivity @152
    if (refl > max_refl) max_refl = refl;
  }
  std::sort(bases.begin(), bases.end());
  bases.erase(std::unique(bases.begin(), bases.end()), bases.end());
  std::vector<uint32_t> ends(bases.size());
  for (size_t j = 0; j < bases.size(); ++j) ends[j] = j + 1 < bases.size() ? bases[j + 1] : s->n_bvh_nodes;

  auto a = new (std::nothrow) vxrt_accel();
  if (!a) return -1;
  a->ref = *s;
  a->max_reflectivity = max_refl;
  (void)hipGetDevice(&a->device);
  uint32_t* d_ranges = nullptr;
  uint32_t* d_status = nullptr;
  uint32_t* d_troot = nullptr;
  constexpr uint32_t TOP_CAP = RT_TOP_NODES;
  // slot descriptors use bit 29 of the payload: only scenes whose compact index space stays below it are staged
  const bool stage_top = TOP_CAP > 0 && (uint64_t)s->n_tlas_nodes + s->n_bvh_nodes < DESC_TOP_FLAG;
  bool ok = hipMalloc(&a->nodes_c, ((size_t)s->n_tlas_nodes + s->n_bvh_nodes) * CNODE_VEC4 * 16) == hipSuccess &&
            (!stage_top || (hipMalloc(&a->top_img, (size_t)TOP_CAP * CNODE_VEC4 * 16) == hipSuccess &&
                            hipMalloc((void**)&a->top_roots, ((size_t)s->n_blas + 2) * sizeof(uint32_t)) == hipSuccess)) &&
            hipMalloc(&a->tri_w, (size_t)s->n_tris * WTRI_FLOATS * 4) == hipSuccess &&
            hipMalloc(&a->blas_root, (size_t)s->n_blas * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc((void**)&d_ranges, bases.size() * 8) == hipSuccess &&
            hipMalloc((void**)&d_status, 4) == hipSuccess && hipMalloc((void**)&d_troot, 4) == hipSuccess;
  uint32_t hstatus = 0, troot = DESC_DONE;
  if (ok) {
    ok = hipMemcpy(d_ranges, bases.data(), bases.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_ranges + bases.size(), ends.data(), ends.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(d_status, 0, 4) == hipSuccess;
  }
  if (ok) {
    const uint32_t nb = (uint32_t)bases.size();
    hipLaunchKernelGGL(accel_nodes_kernel, dim3((s->n_tlas_nodes + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->tlas, s->n_tlas_nodes,
                       (uint4*)a->nodes_c, 1, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, s->n_tris, s->n_blas, 0u, d_status);
    hipLaunchKernelGGL(accel_nodes_kernel, dim3((s->n_bvh_nodes + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->bvh, s->n_bvh_nodes,
                       (uint4*)a->nodes_c + (size_t)s->n_tlas_nodes * CNODE_VEC4, 0, d_ranges, d_ranges + nb, nb, s->n_tris, s->n_blas, s->n_tlas_nodes, d_status);
    hipLaunchKernelGGL(accel_tris_kernel, dim3((s->n_tris + 255) / 256), dim3(256), 0, st, (const float*)s->tri, s->n_tris, (float4*)a->tri_w);
    hipLaunchKernelGGL(accel_roots_kernel, dim3((s->n_blas + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t*)s->tlas, (const uint32_t*)s->bvh,
                       (const uint32_t*)s->blas, s->n_tlas_nodes, s->n_bvh_nodes, s->n_blas, s->n_tris, d_troot, (uint32_t*)a->blas_root, d_status);
    if (s->triEx && s->mat && s->n_mats) {
      const uint32_t nchk = std::max(s->n_tris, s->n_mats);
      hipLaunchKernelGGL(accel_check_shading_kernel, dim3((nchk + 255) / 256), dim3(256), 0, st, (const rt_triex_t*)s->triEx, s->n_tris,
                         (const rt_material_t*)s->mat, s->n_mats, (uint64_t)s->tex_bytes, s->tex ? 1 : 0, d_status);
    }
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
         hipMemcpy(&hstatus, d_status, 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(&troot, d_troot, 4, hipMemcpyDeviceToHost) == hipSuccess;
  }
  uint32_t n_top = 0, troot_top = troot;
  if (ok && stage_top && (hstatus & (STATUS_BAD_SCENE | STATUS_FMA_DECODE_DIFFERS)) == 0) {
    ok = hipMemsetAsync(a->top_img, 0, (size_t)TOP_CAP * CNODE_VEC4 * 16, st) == hipSuccess;
    hipLaunchKernelGGL(accel_top_kernel, dim3(1), dim3(64), 0, st, (const uint4*)a->nodes_c, troot, (const uint32_t*)a->blas_root, s->n_blas, TOP_CAP,
                       (uint4*)a->top_img, a->top_roots, a->top_roots + 1, a->top_roots + 2);
    uint32_t hdr[2] = {0, DESC_DONE};
    ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
         hipMemcpy(hdr, a->top_roots, sizeof hdr, hipMemcpyDeviceToHost) == hipSuccess;
    n_top = hdr[0]; troot_top = hdr[1];
  }
  // depth class of the scene (see accel_depth_kernel): RT_SHALLOW_LEVELS + 1 passes, no host round trip in between -- a scene that still
  // reaches new nodes in the last one is deeper than the class
  uint32_t levels = 0;
  if (ok && (hstatus & STATUS_BAD_SCENE) == 0) {
    const uint32_t nc = s->n_tlas_nodes + s->n_bvh_nodes;
    uint32_t* d_depth = nullptr;
    ok = hipMalloc((void**)&d_depth, ((size_t)nc + 1) * 4) == hipSuccess && hipMemsetAsync(d_depth, 0, ((size_t)nc + 1) * 4, st) == hipSuccess;
    if (ok) {
      for (uint32_t level = 1; level <= RT_SHALLOW_LEVELS + 1u; ++level)
        hipLaunchKernelGGL(accel_depth_kernel, dim3(level == 1u ? 1u : (nc + 255) / 256), dim3(256), 0, st, (const uint4*)a->nodes_c, nc, troot,
                           (const uint32_t*)a->blas_root, level, d_depth, d_depth + nc);
      ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
           hipMemcpy(&levels, d_depth + nc, 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d_depth);
  }
  (void)hipFree(d_ranges); (void)hipFree(d_status); (void)hipFree(d_troot);
  if (!ok || (hstatus & STATUS_BAD_SCENE) != 0) { accel_free(a); return -1; }   // malformed tree: rejected before any traversal
  static const int shallow_env = [] { const char* e = getenv("VXRT_SHALLOW"); return e ? atoi(e) : -1; }();   // (measurement knob: 0 = full-size stacks for every scene)
  a->levels = levels;
  a->shallow = levels <= RT_SHALLOW_LEVELS && shallow_env != 0;
  a->dev.nodes_c = (const uint4*)a->nodes_c; a->dev.ref_tlas = (const uint32_t*)s->tlas; a->dev.n_tlas = s->n_tlas_nodes; a->dev.tri_w = (const float4*)a->tri_w;
  a->dev.blas_root = (const uint32_t*)a->blas_root; a->dev.tlas_root = troot;
  a->dev.ident_root = 0u;
  if (troot >= 0xC0000000u && troot < DESC_IDLE) {   // (an instance descriptor) a single instance under the TLAS root: is its inverse transform (dwords 1-12 of the record) the identity?
    float m[12];
    static const bool ident_off = [] { const char* e = getenv("VXRT_IDENT_ROOT"); return e && e[0] == '0'; }();
    if (!ident_off && hipMemcpy(m, (const uint32_t*)s->blas + (size_t)(troot & PAYLOAD_MASK) * (RT_BLAS_STRIDE / 4) + 1, sizeof m, hipMemcpyDeviceToHost) == hipSuccess) {
      bool id = true;
      for (int i = 0; i < 12; ++i) id = id && m[i] == ((i % 5 == 0) ? 1.0f : 0.0f);     // (m[0], m[5], m[10] on the diagonal; -0 == 0)
      a->dev.ident_root = id ? 1u : 0u;
    }
  }
  a->dev.exact_decode = (hstatus & STATUS_FMA_DECODE_DIFFERS) ? 1u : 0u;
  a->dev.ref_bvh = (const uint32_t*)s->bvh;
  a->dev.blas = (const uint32_t*)s->blas; a->dev.triEx = (const rt_triex_t*)s->triEx;
  a->dev.mat = (const rt_material_t*)s->mat; a->dev.tex = (const uint8_t*)s->tex;
  a->dev.top_img = (const uint4*)a->top_img; a->dev.n_top = n_top; a->dev.tlas_root_top = troot_top;
  a->dev.blas_root_top = n_top ? a->top_roots + 2 : (const uint32_t*)a->blas_root;
  if (getenv("VXRT_DEBUG")) fprintf(stderr, "[vxrt] accel: %u top-of-tree nodes staged for LDS (cap %u)\n", n_top, TOP_CAP);
  *out = a;
  return 0;
}

int vxrt_accel_destroy(vxrt_accel_t* a) {
  if (!a) return 0;
  accel_free(a);
  return 0;
}

uint64_t vxrt_accel_bytes(const vxrt_accel_t* a) {
  if (!a) return 0;
  return (uint64_t)a->ref.n_tlas_nodes * CNODE_VEC4 * 16 + (uint64_t)a->ref.n_bvh_nodes * CNODE_VEC4 * 16 +
         (uint64_t)a->ref.n_tris * WTRI_FLOATS * 4 + (uint64_t)a->ref.n_blas * 4;
}

int vxrt_accel_info(const vxrt_accel_t* a, uint32_t which, uint64_t* value) {
  if (!a || !value) return -1;
  switch (which) {
  case 0: *value = a->levels; return 0;            // internal levels on the longest root-to-leaf path, counted up to RT_SHALLOW_LEVELS + 1
  case 1: *value = a->shallow ? 1u : 0u; return 0; // the timed launches take the SHALLOW instantiations (48-entry stacks)
  case 2: *value = a->dev.ident_root; return 0;    // the TLAS root is one identity instance (rays keep their world coordinates)
  case 3: *value = a->dev.exact_decode; return 0;  // the scene takes the ldexp decode / generic slab form
  }
  return -1;
}

int vxrt_accel_frames_in_flight(vxrt_accel_t* a, uint32_t n) {
  if (!a || n < 1 || n > MAX_FRAMES_IN_FLIGHT) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  a->n_ctx = n; a->next_ctx = 0;
  return 0;
}

static int ensure_defer(FrameCtx* c, uint64_t jobs, hipStream_t s) {
  if (c->defer_cap >= jobs) return 0;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  (void)hipFree(c->defer);
  c->defer = nullptr; c->defer_cap = 0;
  if (hipMalloc((void**)&c->defer, jobs * sizeof(uint32_t)) != hipSuccess) return -1;
  c->defer_cap = jobs;
  return 0;
}

// internal mode of trace_on_ctx: any-hit rays whose hit records are only read as "blocked or not" (JOB_TRACE_UNORDERED)
constexpr int MODE_ANY_UNORDERED = 0x100;
// ray buffer -> hit records on frame context c (the body of vxrt_trace; also the bounce levels of vxrt_render)
static int trace_on_ctx(vxrt_accel_t* a, FrameCtx* c, const float* rays, uint64_t n, const float* tmax,
                        HitRec* hits, int mode, hipStream_t s, const uint32_t* n_dev = nullptr,
                        unsigned long long* stats_counters = nullptr, const uint32_t* order = nullptr) {
  uint32_t* st = status_word();
  if (!st) return -1;
  PersistArgs A{};
  const bool unordered = mode == MODE_ANY_UNORDERED;
  A.total = (uint32_t)n; A.hits = hits; A.rays = rays; A.tmax = tmax; A.any_hit = mode == VXRT_MODE_ANY || unordered;
  A.total_dev = n_dev;
  A.end_log = a->end_log;
  A.order = order;
  A.counters = stats_counters;
  A.wave_log = stats_counters ? a->trace_wave_log : nullptr;
  A.status = st;
  A.per_shard = ((A.total + QUEUE_SHARDS - 1) / QUEUE_SHARDS + 63u) & ~63u;
  if (ensure_defer(c, A.total, s) != 0) return -1;
  if (c->ctl_dirty) {
    if (hipMemsetAsync(c->ctl, 0, CTL_DWORDS * sizeof(uint32_t), s) != hipSuccess) return -1;
  }
  // no kernel follows the EXACT launch that could zero the control block again: it stays dirty and the
  // next use of this context clears it with one fill
  c->ctl_dirty = true;
  A.defer_count = c->ctl; A.defer_list = c->defer; A.defer_cap = A.total;
  A.queue = c->ctl + 32;
  PersistArgs X = A;
  X.queue = c->ctl + 32 + CTL_QUEUE_DWORDS;
  X.total_dev = nullptr;   // the EXACT launch takes its count from the deferral list
  X.order = nullptr;
  ShadeParams p{};
  // incoherent rays: the ray-pool kernel (rt_pool_trace_kernel; VXRT_POOL=0/1 forces the choice), timed builds only
  static const int pool_env = [] { const char* e = getenv("VXRT_POOL"); return e ? atoi(e) : 0; }();
  if (pool_env == 2 && !stats_counters) {     // two rays per lane (rt_pair_trace_kernel)
#define LAUNCH_PAIR(LD, SH) do { \
      hipLaunchKernelGGL((rt_pair_trace_kernel<LD, SH>), dim3(persistent_grid(rt_pair_trace_kernel<LD, SH>, n / 2, 64)), dim3(64), 0, s, a->dev, A); \
      hipLaunchKernelGGL((rt_persistent_kernel<JOB_TRACE, 0, LD, true>), dim3(std::max<uint32_t>(EXACT_GRID, persistent_grid(rt_persistent_kernel<JOB_TRACE, 0, LD, true>, n / 8, 256))), dim3(256), 0, s, a->dev, p, X); } while (0)
    if (a->shallow) { if (a->dev.exact_decode) LAUNCH_PAIR(true, true); else LAUNCH_PAIR(false, true); }
    else            { if (a->dev.exact_decode) LAUNCH_PAIR(true, false); else LAUNCH_PAIR(false, false); }
#undef LAUNCH_PAIR
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (pool_env == 1 && !stats_counters) {
#define LAUNCH_POOL(LD, SH) do { \
      const uint32_t g = persistent_grid(rt_pool_trace_kernel<LD, SH>, n, 64); \
      const uint64_t need = (uint64_t)g * RT_POOL_SLOTS * ((SH ? 3 * RT_SHALLOW_LEVELS : 3 * RT_MAX_LEVELS + RT_POOL_LSTK) - RT_POOL_LSTK) * sizeof(uint2); \
      if (c->pool_spill_bytes < need) { \
        if (hipStreamSynchronize(s) != hipSuccess) return -1; \
        (void)hipFree(c->pool_spill); c->pool_spill = nullptr; c->pool_spill_bytes = 0; \
        if (hipMalloc(&c->pool_spill, need) != hipSuccess) return -1; \
        c->pool_spill_bytes = need; \
      } \
      hipLaunchKernelGGL((rt_pool_trace_kernel<LD, SH>), dim3(g), dim3(64), 0, s, a->dev, A, (uint2*)c->pool_spill); \
      hipLaunchKernelGGL((rt_persistent_kernel<JOB_TRACE, 0, LD, true>), dim3(std::max<uint32_t>(EXACT_GRID, persistent_grid(rt_persistent_kernel<JOB_TRACE, 0, LD, true>, n / 8, 256))), dim3(256), 0, s, a->dev, p, X); } while (0)
    if (a->shallow) { if (a->dev.exact_decode) LAUNCH_POOL(true, true); else LAUNCH_POOL(false, true); }
    else            { if (a->dev.exact_decode) LAUNCH_POOL(true, false); else LAUNCH_POOL(false, false); }
#undef LAUNCH_POOL
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
#define LAUNCH_TJ(J, ST, LD, SH) do { \
    hipLaunchKernelGGL((rt_persistent_kernel<J, ST, LD, false, false, SH>), dim3(persistent_grid(rt_persistent_kernel<J, ST, LD, false, false, SH>, n)), dim3(RT_WG_THREADS), 0, s, a->dev, p, A); \
    hipLaunchKernelGGL((rt_persistent_kernel<JOB_TRACE, ST, LD, true>), dim3(std::max<uint32_t>(EXACT_GRID, persistent_grid(rt_persistent_kernel<JOB_TRACE, ST, LD, true>, n / 8, 256))), dim3(256), 0, s, a->dev, p, X); } while (0)
#define LAUNCH_T(ST, LD, SH) LAUNCH_TJ(JOB_TRACE, ST, LD, SH)
  // any-hit rays whose caller only wants "blocked or not": children in slot order (the EXACT launch keeps the ordered form: a boolean either way)
  static const bool unordered_off = [] { const char* e = getenv("VXRT_UNORDERED_ANY"); return e && atoi(e) == 0; }();
  if (unordered && !unordered_off && !stats_counters) {
    if (a->shallow) { if (a->dev.exact_decode) LAUNCH_TJ(JOB_TRACE_UNORDERED, 0, true, true); else LAUNCH_TJ(JOB_TRACE_UNORDERED, 0, false, true); }
    else            { if (a->dev.exact_decode) LAUNCH_TJ(JOB_TRACE_UNORDERED, 0, true, false); else LAUNCH_TJ(JOB_TRACE_UNORDERED, 0, false, false); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  // (the EXACT launch's grid grows with the ray buffer -- a workgroup per 2,048 rays, up to the machine: how many rays were deferred
  // is known on the device only, and a buffer of axis-parallel rays defers all of them; with nothing deferred its wavefronts find
  // every shard empty without an atomic and exit)
  if (stats_counters)  { if (a->dev.exact_decode) LAUNCH_T(1, true, false); else LAUNCH_T(1, false, false); }
  else if (a->shallow) { if (a->dev.exact_decode) LAUNCH_T(0, true, true); else LAUNCH_T(0, false, true); }
  else                 { if (a->dev.exact_decode) LAUNCH_T(0, true, false); else LAUNCH_T(0, false, false); }
#undef LAUNCH_T
#undef LAUNCH_TJ
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

static bool grow_buf(void** ptr, uint64_t have, uint64_t need, size_t bytes_per_entry) {
  if (have >= need && *ptr) return true;
  (void)hipFree(*ptr);
  *ptr = nullptr;
  return hipMalloc(ptr, (size_t)need * bytes_per_entry) == hipSuccess;
}

static bool level_reserve(FrameCtx::Level& l, uint64_t n, bool shadow, bool only_term) {
  if (l.cap >= n && l.term && (only_term || l.rays) && (!shadow || only_term || l.srays)) return true;
  const uint64_t have = l.cap;
  bool ok = grow_buf((void**)&l.term, have, n, 16);
  if (!only_term) {
    ok = ok && grow_buf((void**)&l.rays, have, n, 24) && grow_buf((void**)&l.hits, have, n, sizeof(HitRec)) &&
         grow_buf((void**)&l.parent, have, n, 4) && grow_buf((void**)&l.col, have, n, 12);
    if (shadow) ok = ok && grow_buf((void**)&l.srays, l.srays ? have : 0, n, 24) && grow_buf((void**)&l.stmax, l.stmax ? have : 0, n, 4) &&
                     grow_buf((void**)&l.shits, l.shits ? have : 0, n, sizeof(HitRec));
  }
  if (ok && l.cap < n) l.cap = n;
  return ok;
}

// Tail of a frame with reflective instances (replaces the plain shading pass): shade level 0, then per
// bounce level trace -> (occlusion rays ->) shade, then fold the colours back.  The level sizes come back
// to the host between levels, so this path synchronises the stream (it is not the benchmarked one).
static int render_bounce_tail(vxrt_accel_t* a, FrameCtx* c, const ShadeParams& p, uint32_t width, uint32_t y0, uint32_t y1,
                              int shadow, const float* utab, const float* vtab, uint32_t* dst, HitRec* hits, float* colors,
                              unsigned long long* rays_traced, hipStream_t s) {
  const SceneDev& sc = a->dev;
  const uint64_t npix = (uint64_t)width * (y1 - y0);          // entries of level 0 (addressed by pixel index)
  const uint64_t pix_span = (uint64_t)width * y1;              // term[] of level 0 is indexed by x + y*W
  if (npix > 0x7fffffffull) return -1;
  if (!c->bcount && hipMalloc((void**)&c->bcount, sizeof(uint32_t)) != hipSuccess) return -1;
  if (c->lv.size() < 2) c->lv.resize(2);
  if (!level_reserve(c->lv[0], pix_span, false, true)) return -1;
  if (!level_reserve(c->lv[1], npix, shadow != 0, false)) return -1;
  dim3 block(256);
  if (hipMemsetAsync(c->bcount, 0, sizeof(uint32_t), s) != hipSuccess) return -1;
  hipLaunchKernelGGL(rt_shade_bounce_kernel<true>, dim3((uint32_t)((npix + 255) / 256)), block, 0, s, sc, p, 0u, npix, width, y0, utab, vtab,
                     (const HitRec*)c->hitbuf, (const float*)nullptr, (const HitRec*)nullptr, c->lv[0].term, (float*)nullptr, dst, hits, colors,
                     c->bcount, c->lv[1].rays, c->lv[1].parent, c->ctl);
  if (hipGetLastError() != hipSuccess) return -1;
  c->ctl_dirty = false;
  uint32_t depth = 0;   // deepest level that holds rays
  for (uint32_t k = 1; k < p.max_depth; ++k) {
    uint32_t n = 0;
    if (hipMemcpyAsync(&n, c->bcount, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    if (n == 0) break;
    FrameCtx::Level& L = c->lv[k];
    L.n = n;
    depth = k;
    if (rays_traced) hipLaunchKernelGGL(add_counter_kernel, dim3(1), dim3(64), 0, s, rays_traced, (unsigned long long)n);
    if (trace_on_ctx(a, c, L.rays, n, nullptr, L.hits, VXRT_MODE_CLOSEST, s) != 0) return -1;
    const dim3 grid((n + 255u) / 256u);
    if (shadow) {
      hipLaunchKernelGGL(rt_bounce_shadow_rays_kernel, grid, block, 0, s, p, n, (const float*)L.rays, (const HitRec*)L.hits, L.srays, L.stmax, rays_traced);
      if (trace_on_ctx(a, c, L.srays, n, L.stmax, L.shits, MODE_ANY_UNORDERED, s) != 0) return -1;
    }
    if (c->lv.size() < (size_t)k + 2) c->lv.resize((size_t)k + 2);
    FrameCtx::Level& Nx = c->lv[k + 1];
    const bool more = k + 1 < p.max_depth;
    if (more && !level_reserve(Nx, n, shadow != 0, false)) return -1;
    FrameCtx::Level& Lk = c->lv[k];   // (resize may have moved the vector)
    if (hipMemsetAsync(c->bcount, 0, sizeof(uint32_t), s) != hipSuccess) return -1;
    hipLaunchKernelGGL(rt_shade_bounce_kernel<false>, grid, block, 0, s, sc, p, k, (uint64_t)n, width, y0, utab, vtab,
                       (const HitRec*)Lk.hits, (const float*)Lk.rays, shadow ? (const HitRec*)Lk.shits : (const HitRec*)nullptr, Lk.term, Lk.col,
                       (uint32_t*)nullptr, (HitRec*)nullptr, (float*)nullptr, c->bcount, more ? Nx.rays : (float*)nullptr, more ? Nx.parent : (uint32_t*)nullptr,
                       (uint32_t*)nullptr);
    if (hipGetLastError() != hipSuccess) return -1;
    if (!more) break;
  }
  for (uint32_t k = depth; k >= 1; --k) {
    FrameCtx::Level& L = c->lv[k];
    const dim3 grid((L.n + 255u) / 256u);
    if (k == 1) hipLaunchKernelGGL(rt_bounce_unwind_kernel<true>, grid, block, 0, s, L.n, (const uint32_t*)L.parent, (const float*)L.col, (const float4*)c->lv[0].term, (float*)nullptr, dst, colors);
    else        hipLaunchKernelGGL(rt_bounce_unwind_kernel<false>, grid, block, 0, s, L.n, (const uint32_t*)L.parent, (const float*)L.col, (const float4*)c->lv[k - 1].term, c->lv[k - 1].col, (uint32_t*)nullptr, (float*)nullptr);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Tail of an ambient-occlusion frame (replaces the plain shading pass): the pixels with a hit are listed on
// the device, their occlusion rays are generated in batches of whole samples (<= AO_BATCH_RAYS rays) and each
// batch is one any-hit launch whose job count stays in device memory.  No host synchronisation.
#define AO_BATCH_RAYS (32ull << 20)
#define VXRT_AO_MODE_DIFFUSE_BOUNCE 1u   // internal: vxrt_ao_params_t::reserved
static int render_ao_tail(vxrt_accel_t* a, FrameCtx* c, const ShadeParams& p, uint32_t width, uint32_t y0, uint32_t y1,
                          const vxrt_ao_params_t* ao, const float* utab, const float* vtab, uint32_t* dst, float* colors,
                          uint32_t* unoccluded, unsigned long long* rays_traced, hipStream_t s) {
  const SceneDev& sc = a->dev;
  const uint64_t n = (uint64_t)width * (y1 - y0);
  if (ao->reserved == VXRT_AO_MODE_DIFFUSE_BOUNCE) return -1;    // (the diffuse-bounce frame is JOB_RENDER_GI: it has no tail)
  if (n > 0x7fffffffull || ao->spp == 0) return -1;
  uint32_t ns = (uint32_t)std::min<uint64_t>(ao->spp, std::max<uint64_t>(1, AO_BATCH_RAYS / n));   // samples per batch
  const uint64_t ray_cap = n * ns;
  if (ray_cap > 0x7fffffffull) return -1;
  if (c->ao_cap < n || c->ao_ray_cap < ray_cap) {
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    const uint64_t have = c->ao_cap, rhave = c->ao_ray_cap;
    bool ok = grow_buf((void**)&c->ao_geo, have, n, 16) && grow_buf((void**)&c->ao_nrm, have, n, 16) && grow_buf((void**)&c->ao_col, have, n, 16) &&
              grow_buf((void**)&c->ao_cnt, have, n, 4) && grow_buf((void**)&c->ao_list, have, n, 4) && grow_buf((void**)&c->ao_hdr, c->ao_hdr ? 1 : 0, 1, 8) &&
              grow_buf((void**)&c->ao_rays, rhave, ray_cap, 24) && grow_buf((void**)&c->ao_tmax, rhave, ray_cap, 4) &&
              grow_buf((void**)&c->ao_hits, rhave, ray_cap, sizeof(HitRec));
    if (!ok) return -1;
    c->ao_cap = std::max(have, n); c->ao_ray_cap = std::max(rhave, ray_cap);
  }
  const dim3 block(256), grid((uint32_t)((n + 255) / 256)), rgrid((uint32_t)((ray_cap + 255) / 256));
  // VXRT_SORT_SECONDARY=1: trace the secondary rays in (direction octant, origin cell) order instead of generation order.  OFF by
  // default: measured SLOWER on both passes that use it (diffuse bounce 1.53 -> 1.95 ms, 10M-triangle hairball AO 9.2 -> 16.6 ms,
  // profiles/r02_f_secondary_sort.txt).  The traversal is bound by per-lane VALU work, which coherence does not reduce, and the
  // generation order already puts the 16 samples of one pixel (AO) / 64 neighbouring pixels (bounce) side by side.
  static const bool sort_on = [] { const char* e = getenv("VXRT_SORT_SECONDARY"); return e && e[0] == '1'; }();
  const uint32_t cells_x = (width + (1u << BIN_CELL_SHIFT) - 1) >> BIN_CELL_SHIFT, cells_y = (y1 - y0 + (1u << BIN_CELL_SHIFT) - 1) >> BIN_CELL_SHIFT;
  const uint32_t n_cells = cells_x * cells_y, n_bins = 8u * n_cells;
  if (sort_on && (c->bin_cap < n_bins || c->bin_ray_cap < ray_cap)) {
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    bool ok = grow_buf((void**)&c->bin_hist, c->bin_cap, n_bins, 4) && grow_buf((void**)&c->bin_keys, c->bin_ray_cap, ray_cap, 4) &&
              grow_buf((void**)&c->bin_order, c->bin_ray_cap, ray_cap, 4);
    if (!ok) return -1;
    c->bin_cap = std::max<uint64_t>(c->bin_cap, n_bins); c->bin_ray_cap = std::max(c->bin_ray_cap, ray_cap);
  }
  auto bin_rays = [&](uint32_t ns_batch) -> const uint32_t* {   // ao_rays of the current batch -> bin_order
    if (!sort_on) return nullptr;
    if (hipMemsetAsync(c->bin_hist, 0, (size_t)n_bins * 4, s) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(rt_bin_count_kernel, rgrid, block, 0, s, ray_cap, (const uint32_t*)c->ao_hdr, (const float*)c->ao_rays, (const uint32_t*)c->ao_list, ns_batch,
                       width, cells_x, n_cells, c->bin_hist, c->bin_keys);
    hipLaunchKernelGGL(rt_bin_scan_kernel, dim3(1), dim3(1024), 0, s, c->bin_hist, n_bins, (n_bins + 1023u) / 1024u);
    hipLaunchKernelGGL(rt_bin_scatter_kernel, rgrid, block, 0, s, ray_cap, (const uint32_t*)c->ao_hdr, (const uint32_t*)c->bin_keys, c->bin_hist, c->bin_order);
    return c->bin_order;
  };
  if (hipMemsetAsync(c->ao_hdr, 0, 8, s) != hipSuccess) return -1;
  hipLaunchKernelGGL(rt_ao_prepare_kernel, dim3((uint32_t)((n + 256u * AO_PREP_CHUNKS - 1u) / (256u * AO_PREP_CHUNKS))), block, 0, s, sc, p, n, width, y0, utab, vtab, (const HitRec*)c->hitbuf,
                     c->ao_geo, c->ao_nrm, c->ao_col, c->ao_cnt, c->ao_list, c->ao_hdr, c->ctl, (float4*)nullptr);
  if (hipGetLastError() != hipSuccess) return -1;
  c->ctl_dirty = false;
  for (uint32_t s0 = 0; s0 < ao->spp; s0 += ns) {
    const uint32_t k = std::min(ns, ao->spp - s0);
    hipLaunchKernelGGL(rt_ao_rays_kernel, rgrid, block, 0, s, ray_cap, width, y0, utab, vtab, (const float4*)c->ao_geo, (const float4*)c->ao_nrm,
                       (const uint32_t*)c->ao_list, c->ao_hdr, ao->spp, s0, k, ao->seed, ao->radius, c->ao_rays, c->ao_tmax);
    if (trace_on_ctx(a, c, c->ao_rays, n * k, c->ao_tmax, c->ao_hits, MODE_ANY_UNORDERED, s, c->ao_hdr + 1, nullptr, bin_rays(k)) != 0) return -1;
    hipLaunchKernelGGL(rt_ao_accumulate_kernel, rgrid, block, 0, s, ray_cap, (const uint32_t*)c->ao_list, (const uint32_t*)c->ao_hdr, k,
                       (const HitRec*)c->ao_hits, c->ao_cnt);
  }
  hipLaunchKernelGGL(rt_ao_final_kernel, grid, block, 0, s, n, width, y0, (const float4*)c->ao_geo, (const float4*)c->ao_col, (const uint32_t*)c->ao_cnt,
                     ao->spp, dst, colors, unoccluded, rays_traced);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int render_common(vxrt_accel_t* a, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const vxrt_shade_params_t* params, int shadow, uint32_t* dst, vxrt_hit_t* hits, float* colors,
                         unsigned long long* counters, int stats, void* stream, unsigned long long* wave_log = nullptr,
                         const vxrt_ao_params_t* ao = nullptr, uint32_t* unoccluded = nullptr, uint32_t stride = 1,
                         uint32_t batch = 1, uint64_t dst_frame_stride = 0) {
  if (!a || !params || !dst) return -1;
  // batch > 1: `params` is an array of `batch` entries, frame f goes to dst + f * dst_frame_stride; plain frames without optional outputs
  if (batch == 0 || batch > VXRT_MAX_BATCH || (batch > 1 && (hits || colors || ao || ((stats || wave_log) && !(stats == 2 && wave_log))))) return -1;   // (a batch with the wave log: diagnostic, traversal only -- its shading launch is the single-frame one)
  if (ao && (stats || shadow)) return -1;
  if (stride == 0 || (stride > 1 && ((y0 & 7u) != 0 || ao))) return -1;   // interleaved tile rows: tile-aligned start, plain frames only
  if (!a->ref.triEx || !a->ref.mat || a->ref.n_mats == 0) return -1;  // shading needs them (closest.cpp:52-55)
  if (width == 0 || height == 0 || y0 > y1 || y1 > height) return -1;
  if (stats && !counters) return -1;
  if (y0 == y1) return 0;
  uint32_t* st = status_word();
  if (!st) return -1;
  ShadeParams p;
  for (int i = 0; i < 3; ++i) {
    p.amb[i] = params->ambient[i]; p.lcol[i] = params->light_color[i];
    p.lpos[i] = params->light_pos[i]; p.bg[i] = params->background[i];
  }
  p.max_depth = params->max_depth;
  // tile rows of the window: every stride-th tile row of [y0, y1) starting with the one at y0
  const uint32_t tiles_x = (width + 7) / 8, tiles_y = ((y1 - y0 + 7) / 8 + stride - 1) / stride;
  const uint32_t row_step = 8u * stride;
  const uint64_t n_tiles64 = (uint64_t)tiles_x * tiles_y * batch;
  if (n_tiles64 > 0x1ffffffull) return -1;
  const uint32_t n_tiles = (uint32_t)n_tiles64;            // of the whole batch
  const uint32_t frame_tiles = tiles_x * tiles_y;
  dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  const SceneDev& sc = a->dev;
  // every refusal that depends on the arguments alone comes BEFORE a frame context is taken (a context taken and not released
  // would leave its next user unordered behind whatever this call had already enqueued)
  const bool mirror = p.max_depth > 1 && a->max_reflectivity > 0.0f;
  if (mirror && (stats || stride > 1 || batch > 1)) return -1;   // the mirror-bounce path: whole single frames, timed build only
  if (batch > 1)
    for (uint32_t f = 0; f < batch; ++f)
      if (params[f].max_depth > 1 && a->max_reflectivity > 0.0f) return -1;
  FrameCtx* c = acquire_ctx(a, s);
  if (!c) return -1;
  // from here on a failure releases the context the way a success does: its event is recorded behind whatever was enqueued, the
  // context is marked busy on this stream and its control block is cleared before the next use
  auto fail = [&]() -> int { c->ctl_dirty = true; (void)release_ctx(a, c, s); return -1; };
  // hit-record buffer between the two passes (one per frame in flight)
  const uint64_t pixels = batch > 1 ? (uint64_t)n_tiles * 64u : (uint64_t)tiles_x * ((height + 7) / 8 + 1) * 64u;   // tile-major records of any row window of the frame
  if (c->hitbuf_pixels < pixels) {
    if (hipStreamSynchronize(s) != hipSuccess) return fail();
    (void)hipFree(c->hitbuf);
    c->hitbuf = nullptr; c->hitbuf_pixels = 0;
    if (hipMalloc(&c->hitbuf, pixels * sizeof(HitRec)) != hipSuccess) return fail();
    c->hitbuf_pixels = pixels;
  }
  if (a->uv_w != width || a->uv_h != height) {
    // kernel.cpp:32-33 evaluated on the host in double, once per column and row
    std::vector<float> tab((size_t)width + height);
    for (uint32_t x = 0; x < width; ++x) tab[x] = (float)(((double)x * 2.0 - (double)width) / (double)height);
    for (uint32_t y = 0; y < height; ++y) tab[width + y] = (float)(((double)y * 2.0 - (double)height) / (double)height);
    if (hipDeviceSynchronize() != hipSuccess) return fail();   // frames in flight on other streams read the old table
    (void)hipFree(a->uvtab);
    a->uvtab = nullptr; a->uv_w = a->uv_h = 0;
    if (hipMalloc((void**)&a->uvtab, tab.size() * sizeof(float)) != hipSuccess) return fail();
    if (hipMemcpy(a->uvtab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail();
    a->uv_w = width; a->uv_h = height;
  }
  PersistArgs A{};
  A.W = width; A.H = height; A.y0 = y0; A.y1 = y1; A.tiles_x = tiles_x; A.row_step = row_step; A.total = n_tiles * 64u;
  A.div_tiles_x = fast_div_make(tiles_x); A.div_frame_tiles = fast_div_make(frame_tiles); A.frame_tiles = frame_tiles;
  A.hits = (HitRec*)c->hitbuf; A.counters = counters; A.status = st; A.wave_log = wave_log; A.end_log = a->end_log;
  A.utab = a->uvtab; A.vtab = a->uvtab + width;
  if (batch > 1) {
    if (!c->pbatch && hipMalloc((void**)&c->pbatch, VXRT_MAX_BATCH * sizeof(ShadeParams)) != hipSuccess) return fail();
    ShadeParams pb[VXRT_MAX_BATCH];
    for (uint32_t f = 0; f < batch; ++f) {
      for (int i = 0; i < 3; ++i) {
        pb[f].amb[i] = params[f].ambient[i]; pb[f].lcol[i] = params[f].light_color[i];
        pb[f].lpos[i] = params[f].light_pos[i]; pb[f].bg[i] = params[f].background[i];
      }
      pb[f].max_depth = params[f].max_depth;
    }
    // (by value through the kernel arguments: captured when the launch is enqueued, whatever the caller does with `params` next)
    ShadeBatch sb;
    for (uint32_t f = 0; f < VXRT_MAX_BATCH; ++f) sb.p[f] = pb[f < batch ? f : 0];
    hipLaunchKernelGGL(set_batch_params_kernel, dim3(1), dim3(64), 0, s, sb, batch, c->pbatch);
    A.pbatch = c->pbatch; A.frame_tiles = frame_tiles;
    // tile order of a batch: band-major -- queue shard s (= the XCD that works on it) gets band s of EVERY frame, so that an
    // XCD's L2 keeps holding one band's part of the BVH, as it does for a single frame; frame-major order would hand each XCD
    // whole frames (measured at 8 frames per batch: slower than no batch at all)
    if (a->bo_tiles != frame_tiles) {   // another window: drop the orders of the old one
      if (hipDeviceSynchronize() != hipSuccess) return fail();
      for (uint32_t k = 0; k <= VXRT_MAX_BATCH; ++k) { (void)hipFree(a->batch_order[k]); a->batch_order[k] = nullptr; }
      a->bo_tiles = frame_tiles;
    }
    if (!a->batch_order[batch]) {
      std::vector<uint32_t> ord;
      ord.reserve(n_tiles);
      const uint32_t band = (frame_tiles + QUEUE_SHARDS - 1) / QUEUE_SHARDS;
      for (uint32_t sh = 0; sh < QUEUE_SHARDS; ++sh)
        for (uint32_t f = 0; f < batch; ++f)
          for (uint32_t t = sh * band; t < std::min(frame_tiles, (sh + 1) * band); ++t) ord.push_back(f * frame_tiles + t);
      if (hipMalloc((void**)&a->batch_order[batch], ord.size() * sizeof(uint32_t)) != hipSuccess) return fail();
      if (hipMemcpy(a->batch_order[batch], ord.data(), ord.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return fail();
    }
    A.tile_order = a->batch_order[batch];
  }
  if (ensure_defer(c, A.total, s) != 0) return fail();
  A.defer_count = c->ctl; A.defer_list = c->defer; A.defer_cap = A.total;
  A.queue = c->ctl + 32;
  A.per_shard = ((A.total + QUEUE_SHARDS - 1) / QUEUE_SHARDS + 63u) & ~63u;
  static const uint32_t shard_rot_env = [] { const char* e = getenv("VXRT_SHARD_ROT"); return e ? (uint32_t)atoi(e) : 0u; }();
  A.shard_rot = shard_rot_env;
  // longest tile first, learned from this context's previous frame of the same window (VXRT_LPT=0 disables).  Only
  // with one frame in flight: overlapped frames fill each other's tails already
  static const bool lpt_on = [] { const char* e = getenv("VXRT_LPT"); return !(e && e[0] == '0'); }();
  // and only for frames of more than LPT_MIN_TILES tiles: below, the sort launch costs more than the shorter tail saves
  // (1024x1024, 86 % background: -5 %; 1920x1080: +9 %; 3840x2160: +4 %; the sort on a side stream instead: worse, the
  // two extra event hops cost more than the kernel)
  // Single frames in flight on several streams: no difference, measured (round 2).  BATCHES of frames: a rank's share of a frame split
  // over GPUs makes short launches -- at 8 ranks a 20-step run is two launches of ~5 tiles per wavefront, whose tails nothing
  // fills -- and the batches of a frame loop repeat: the order is learned from the context's previous batch of the same size
  // (VXRT_LPT_BATCH=0 disables; profiles/r03_h_lpt_batch.txt).
  static const bool lpt_batch_on = [] { const char* e = getenv("VXRT_LPT_BATCH"); return !(e && e[0] == '0'); }();
  // ... for batches of at most LPT_BATCH_MAX_TILES tiles (about a dozen per resident wavefront): measured on one box, driver-sized
  // runs, rank 0's pipeline of 8 / 4 / 2 ranks (40.8 K / 81.6 K / 162 K tiles per batch): +5.5 % / +2 % / 0; one GPU's batches of
  // five whole frames (162 K tiles, sets overlapping on two streams): -5 % -- sorted by cost, a band's tiles are no longer
  // traced next to their screen neighbours, and there the tails are filled anyway.
  static const uint32_t lpt_batch_max = [] { const char* e = getenv("VXRT_LPT_BATCH_MAX"); return e ? (uint32_t)atoll(e) : LPT_BATCH_MAX_TILES; }();   // (measurement knob)
  // (a set issued on its own -- one frame context: the samples of one vx_start -- has nothing behind it to fill its tail, whatever its size)
  // (longest tile first also in a large set then: the samples of `rt_host -s 5`, 162 K tiles, 2.25 -> 2.00 ms per vx_start; VXRT_LPT_BATCH_ALONE=0: off)
  static const int lpt_alone_env = [] { const char* e = getenv("VXRT_LPT_BATCH_ALONE"); return e ? atoi(e) : 1; }();
  const bool lpt = lpt_on && (!stats || wave_log) && n_tiles >= LPT_MIN_TILES &&
                   (batch == 1 ? a->n_ctx == 1 : (lpt_batch_on && (n_tiles <= lpt_batch_max || (lpt_alone_env && a->n_ctx == 1))));
  FrameCtx::Lpt& L = c->lpt[batch];
  if (lpt) {
    if (L.cap < n_tiles) {
      if (hipStreamSynchronize(s) != hipSuccess) return fail();
      (void)hipFree(L.cost); (void)hipFree(L.order);
      L.cost = L.order = nullptr; L.cap = 0; L.valid = false;
      // (cost: n entries + n start clocks + n durations + n steal distances behind them, the latter three written by the wave-log build only)
      if (hipMalloc((void**)&L.cost, (size_t)n_tiles * 4 * 4) != hipSuccess || hipMalloc((void**)&L.order, (size_t)n_tiles * 4) != hipSuccess) return fail();
      L.cap = n_tiles;
    }
    const uint32_t key[6] = {width, height, y0, y1, (uint32_t)shadow | (stride << 1), (ao ? 1u : 0u) | (batch << 1)};
    if (memcmp(key, L.key, sizeof key) != 0) { L.valid = false; memcpy(L.key, key, sizeof key); }
    if (!L.cost || !L.order) return fail();   // (whatever happened above: no launch with a missing table)
    A.tile_cost = L.cost;
    if (L.valid) A.tile_order = L.order;      // else: the static order (identity, or the batch's band-major order set above)
    else if (batch == 1) A.tile_order = nullptr;
  }
  // a-priori EXACT list (camera rays with u == 0 or v == 0), rebuilt only when the window changes.  The list of a batch is the
  // frames' lists one after the other, so a list built for F frames serves every batch <= F: the launch takes a prefix.
  if (a->ap_key[0] != width || a->ap_key[1] != height || a->ap_key[2] != y0 || a->ap_key[3] != y1 || a->ap_key[4] != stride || a->ap_key[5] < batch || !a->apriori) {
    std::vector<uint32_t> list(1, 0u);
    for (uint32_t t = 0; t < frame_tiles; ++t)
      for (uint32_t l = 0; l < 64; ++l) {
        const uint32_t x = (t % tiles_x) * 8u + (l & 7u), y = y0 + (t / tiles_x) * row_step + (l >> 3);
        if (x >= width || y >= y1) continue;
        const float u = (float)(((double)x * 2.0 - (double)width) / (double)height);
        const float v = (float)(((double)y * 2.0 - (double)height) / (double)height);
        if (u == 0.0f || v == 0.0f) list.push_back(t * 64u + l);
      }
    const size_t per_frame = list.size() - 1;
    for (uint32_t f = 1; f < batch; ++f)
      for (size_t i = 0; i < per_frame; ++i) list.push_back(list[1 + i] + f * frame_tiles * 64u);
    list[0] = (uint32_t)(list.size() - 1);
    if (hipDeviceSynchronize() != hipSuccess) return fail();
    if (a->ap_cap < list.size()) {
      (void)hipFree(a->apriori);
      a->apriori = nullptr; a->ap_cap = 0;
      if (hipMalloc((void**)&a->apriori, list.size() * sizeof(uint32_t)) != hipSuccess) return fail();
      a->ap_cap = list.size();
    }
    if (hipMemcpy(a->apriori, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) return fail();
    a->ap_count = (uint32_t)per_frame;      // per frame
    a->ap_key[0] = width; a->ap_key[1] = height; a->ap_key[2] = y0; a->ap_key[3] = y1; a->ap_key[4] = stride; a->ap_key[5] = batch;
  }
  const uint32_t ap_count = a->ap_count * batch;   // of this launch: the first `batch` frames of the list
  // EXACT launch over the a-priori list on the side stream (ordered after everything already queued on
  // `s`: it writes hit records the previous frame's shading pass may still be reading), concurrent with
  // the main launch; then the main launch and the EXACT launch over whatever the main one deferred
  PersistArgs X = A, X0 = A;
  X.queue = c->ctl + 32 + CTL_QUEUE_DWORDS;
  c->ctl_dirty = true;   // until the shading pass that zeroes the block again is enqueued
  const bool side_launch = ap_count != 0;
  // the a-priori EXACT launch needs a few workgroups (3,000 rays of a 1080p frame = 12); the main launch leaves that many slots
  // free: a persistent grid that fills every CU (LDS) would otherwise keep them waiting until its first workgroups retire, and
  // the frame would end on them (measured: 33 us after the main launch, profiles/r02_d_exact_timeline.txt)
  const uint32_t side_wgs = side_launch ? std::min<uint32_t>(EXACT_GRID, (ap_count + 255u) / 256u) : 0u;
  // ... on a serial frame.  With sets of frames overlapping on several streams the a-priori launch cannot get those slots anyway --
  // an EXACT workgroup needs more registers than one retired main workgroup frees, so it only finds room in a tail (measured:
  // it ends when its own main launch begins to drain, profiles/r03_a_pipeline_timeline.txt) -- and nothing waits for it before
  // the other stream's tail: the main launch takes the whole machine then, +1 % (serial: -7 %).  VXRT_SIDE_RESERVE=0/1 forces it.
  static const int side_reserve_env = [] { const char* e = getenv("VXRT_SIDE_RESERVE"); return e ? atoi(e) : -1; }();
  const bool side_reserve = side_reserve_env >= 0 ? side_reserve_env != 0 : a->n_ctx == 1;
  // frames packed in overlapping sets: the 8-wavefront instantiation (see rt_persistent_kernel); VXRT_PACKED=0/1 forces it
  static const int packed_env = [] { const char* e = getenv("VXRT_PACKED"); return e ? atoi(e) : -1; }();
  static const int packed_batch_env = [] { const char* e = getenv("VXRT_PACKED_BATCH"); return e ? atoi(e) : 0; }();   // (measurement knob: sets of frames take it even with one frame context)
  const bool packed = packed_env >= 0 ? packed_env != 0 : ((a->n_ctx > 1 || (packed_batch_env && batch > 1)) && n_tiles >= LPT_MIN_TILES);
  hipStream_t side = c->side;
  if (side_launch) {
    // the side stream starts behind what is queued on `s` (the previous frame's shading pass reads the hit records this launch writes) --
    // unless nothing is: a caller that waits for every frame before it starts the next (the vx_* sequence: vx_ready_wait, then vx_start)
    // finds the stream drained, and the fork's event record -- a barrier packet in front of the main launch, ~12 us -- is not needed
    const bool drained = hipStreamQuery(s) == hipSuccess;
    if (!drained && (hipEventRecord(c->ev_in, s) != hipSuccess || hipStreamWaitEvent(side, c->ev_in, 0) != hipSuccess)) return fail();
    X0.queue = c->ctl + 32 + 2 * CTL_QUEUE_DWORDS;
    X0.defer_count = a->apriori; X0.defer_list = a->apriori + 1; X0.defer_cap = ap_count;
  }
  // A window that is small against the machine (one rank's share of a frame split N ways: at 1080p / 8 GPUs 4,080 tiles for 6,096
  // resident wavefronts) makes one launch a single round of tiles -- as long as its slowest tile, about half a full frame -- and a
  // second frame's launch only gets the slots the first one leaves.  With several frames in flight each launch therefore takes
  // its share of the machine (capacity / frames in flight): the frames run side by side, each wavefront working through several
  // tiles, and the machine stays full.  (A full frame, many tiles per wavefront, keeps the whole grid: measured better.)
  static const int grid_div_env = [] { const char* e = getenv("VXRT_GRID_DIV"); return e ? atoi(e) : 0; }();
#define MAIN_GRID(K) [&]() -> uint32_t { \
    const uint32_t side_wgs_r = side_reserve ? side_wgs : 0u; \
    uint32_t g = persistent_grid(K, A.total + (uint64_t)side_wgs_r * RT_WG_THREADS); \
    const uint32_t cap = persistent_grid(K, ~0ull >> 8); \
    uint32_t div = grid_div_env > 0 ? (uint32_t)grid_div_env : ((grid_div_env == 0 && a->n_ctx > 1 && (uint64_t)A.total < 2ull * 64ull * RT_WG_WAVES * cap) ? a->n_ctx : 1u); \
    if (div > 1u) g = std::min<uint32_t>(g, std::max<uint32_t>(cap / div, 1u)); \
    return std::max<uint32_t>(1u, g > side_wgs_r ? g - side_wgs_r : 1u); }()
#define LAUNCH_P(J, ST, LD, PK, SH) do { \
    if (side_launch) hipLaunchKernelGGL((rt_persistent_kernel<J, ST, LD, true>), dim3(side_wgs), block, 0, side, sc, p, X0); \
    hipLaunchKernelGGL((rt_persistent_kernel<J, ST, LD, false, PK, SH>), dim3(MAIN_GRID((rt_persistent_kernel<J, ST, LD, false, PK, SH>))), dim3(RT_WG_THREADS), 0, s, sc, p, A); \
    hipLaunchKernelGGL((rt_persistent_kernel<J, ST, LD, true>), dim3(EXACT_GRID), block, 0, s, sc, p, X); } while (0)
#define LAUNCH_PD(J, ST, PK) do { if (sc.exact_decode) LAUNCH_P(J, ST, true, PK, false); else LAUNCH_P(J, ST, false, PK, false); } while (0)
  // (timed builds: the scene's depth class picks the size of the scratch part of the stack)
#define LAUNCH_PDS(J, PK) do { if (a->shallow) { if (sc.exact_decode) LAUNCH_P(J, 0, true, PK, true); else LAUNCH_P(J, 0, false, PK, true); } else LAUNCH_PD(J, 0, PK); } while (0)
  // one diffuse bounce: the whole frame in the persistent launches (JOB_RENDER_GI).  (The multi-pass form it replaced -- list the hit
  // pixels, generate the rays, a 2 M-ray trace launch, accumulate, final -- took the same 1.18 ms: profiles/r03_f_gi_fused_ab.txt)
  const bool gi_fused = ao && ao->reserved == VXRT_AO_MODE_DIFFUSE_BOUNCE;
  if (gi_fused && (stats || shadow || unoccluded)) return fail();
  if (gi_fused) {
    A.dst = dst; A.colors = colors; A.gi_seed = ao->seed;
    X.dst = dst; X.colors = colors; X.gi_seed = ao->seed;
    X0.dst = dst; X0.colors = colors; X0.gi_seed = ao->seed;
    LAUNCH_PDS(JOB_RENDER_GI, false);
  } else
  if (stats == 2)  { if (shadow) LAUNCH_PD(JOB_RENDER_SHADOW, 2, false); else LAUNCH_PD(JOB_RENDER, 2, false); }
  else if (stats)  { if (shadow) LAUNCH_PD(JOB_RENDER_SHADOW, 1, false); else LAUNCH_PD(JOB_RENDER, 1, false); }
  else if (packed) { if (shadow) LAUNCH_PDS(JOB_RENDER_SHADOW, true); else LAUNCH_PDS(JOB_RENDER, true); }
  else             { if (shadow) LAUNCH_PDS(JOB_RENDER_SHADOW, false); else LAUNCH_PDS(JOB_RENDER, false); }
#undef LAUNCH_PDS
#undef LAUNCH_PD
#undef LAUNCH_P
#undef MAIN_GRID
  // (the tile sort for the next frame rides in the shading launch; the AO / bounce tails have no such launch and skip it)
  const bool lpt_sort = lpt && !ao && !(p.max_depth > 1 && a->max_reflectivity > 0.0f);
  if (lpt && !lpt_sort) L.valid = false;
  if (side_launch) {
    if (hipEventRecord(c->ev_side, side) != hipSuccess || hipStreamWaitEvent(s, c->ev_side, 0) != hipSuccess) return fail();
  }
  if (gi_fused) {
    // nothing follows: the pixels are written.  The control block stays as the launches left it; the context's next call clears it.
    if (hipGetLastError() != hipSuccess) return fail();
    return release_ctx(a, c, s);
  }
  if (ao) {
    if (render_ao_tail(a, c, p, width, y0, y1, ao, A.utab, A.vtab, dst, colors, unoccluded, counters, s) != 0) return fail();
    return release_ctx(a, c, s);
  }
  if (p.max_depth > 1 && a->max_reflectivity > 0.0f) {
    // reflective instances: the shading pass becomes the level-0 step of the mirror-bounce wavefront
    if (render_bounce_tail(a, c, p, width, y0, y1, shadow, A.utab, A.vtab, dst, (HitRec*)hits, colors, counters, s) != 0) return fail();
    return release_ctx(a, c, s);
  }
  const uint64_t npx = (uint64_t)width * tiles_y * 8u * batch;
  const uint32_t lpt_blocks = lpt_sort ? QUEUE_SHARDS : 0u;
  dim3 sgrid((uint32_t)((npx + 255) / 256) + lpt_blocks);
  if (stats) hipLaunchKernelGGL(rt_shade_kernel<true>, sgrid, block, 0, s, sc, p, width, height, y0, y1, row_step, tiles_y * 8u, A.utab, A.vtab, (const HitRec*)c->hitbuf, dst, (HitRec*)hits, colors, counters, c->ctl,
                                lpt_blocks, (const uint32_t*)L.cost, L.order, n_tiles, A.per_shard >> 6, 1u, (const ShadeParams*)nullptr, (uint64_t)0,
                                batch > 1 ? (const uint32_t*)a->batch_order[batch] : (const uint32_t*)nullptr);
  else       hipLaunchKernelGGL(rt_shade_kernel<false>, sgrid, block, 0, s, sc, p, width, height, y0, y1, row_step, tiles_y * 8u, A.utab, A.vtab, (const HitRec*)c->hitbuf, dst, (HitRec*)hits, colors, counters, c->ctl,
                                lpt_blocks, (const uint32_t*)L.cost, L.order, n_tiles, A.per_shard >> 6, batch, (const ShadeParams*)c->pbatch, dst_frame_stride,
                                batch > 1 ? (const uint32_t*)a->batch_order[batch] : (const uint32_t*)nullptr);
  if (hipGetLastError() != hipSuccess) return fail();
  if (lpt_sort) L.valid = true;
  c->ctl_dirty = false;
  return release_ctx(a, c, s);
}

int vxrt_render(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                const vxrt_shade_params_t* params, int shadow, uint32_t* dst, vxrt_hit_t* hits,
                float* colors, unsigned long long* rays_traced, void* stream) {
  return render_common(accel, width, height, y0, y1, params, shadow, dst, hits, colors, rays_traced, false, stream);
}

// Tile rows phase, phase + stride, phase + 2 stride, ... of the frame (8 rows each): what rank `phase` of `stride` ranks renders when one
// frame is split over GPUs (bench.py --gpus N; DCR 0x7F3 through the vx_* boundary).  Interleaving balances the ranks -- the
// cost of a tile varies 4x across the frame, mostly with height -- where contiguous bands do not.
int vxrt_render_interleaved(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t phase, uint32_t stride,
                            const vxrt_shade_params_t* params, int shadow, uint32_t* dst, vxrt_hit_t* hits,
                            float* colors, unsigned long long* rays_traced, void* stream) {
  if (stride == 0 || phase >= stride) return -1;
  if ((uint64_t)phase * 8u >= height) return 0;   // more ranks than tile rows: nothing for this one
  return render_common(accel, width, height, phase * 8u, height, params, shadow, dst, hits, colors, rays_traced, false, stream, nullptr, nullptr, nullptr, stride);
}

// n_frames frames of the same window in ONE set of launches: frame f is lit and shaded with params[f] and written to dst + f *
// dst_frame_stride.  One rank's share of a frame split N ways is a single round of tiles -- as long as its slowest tile, about half
// a full frame's time however small the share -- so a sequence of frames is traced side by side instead.
int vxrt_render_interleaved_batch(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t phase, uint32_t stride, uint32_t n_frames,
                                  const vxrt_shade_params_t* params, int shadow, uint32_t* dst, uint64_t dst_frame_stride,
                                  unsigned long long* rays_traced, void* stream) {
  if (stride == 0 || phase >= stride || n_frames == 0) return -1;
  if ((uint64_t)phase * 8u >= height) return 0;
  return render_common(accel, width, height, phase * 8u, height, params, shadow, dst, nullptr, nullptr, rays_traced, false, stream, nullptr, nullptr, nullptr, stride,
                       n_frames, dst_frame_stride);
}

// n_frames frames of the row window [y0, y1) in one set of launches: what a rank renders when the frame is split into contiguous
// bands (bench.py --shard bands).  dst addresses each frame as a FULL frame does (pixel (x, y) of frame f at dst[f * dst_frame_stride
// + x + y * width]): only rows [y0, y1) are touched, so a caller that keeps just its band passes (band buffer - y0 * width) and a
// frame stride of (y1 - y0) * width.
int vxrt_render_rows_batch(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t n_frames,
                           const vxrt_shade_params_t* params, int shadow, uint32_t* dst, uint64_t dst_frame_stride,
                           unsigned long long* rays_traced, void* stream) {
  if (n_frames == 0) return -1;
  return render_common(accel, width, height, y0, y1, params, shadow, dst, nullptr, nullptr, rays_traced, false, stream, nullptr, nullptr, nullptr, 1,
                       n_frames, dst_frame_stride);
}

// diagnostic: vxrt_render_interleaved_batch's traversal launch with the per-wavefront log of vxrt_render_wave_log (counting build of the
// timed traversal; the pixels are NOT produced: the shading launch of this build knows single frames only)
int vxrt_render_interleaved_batch_wave_log(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t phase, uint32_t stride, uint32_t n_frames,
                                           const vxrt_shade_params_t* params, int shadow, uint32_t* dst, uint64_t dst_frame_stride,
                                           unsigned long long* counters, unsigned long long* wave_log, void* stream) {
  if (stride == 0 || phase >= stride || n_frames == 0 || !wave_log || !counters) return -1;
  return render_common(accel, width, height, phase * 8u, height, params, shadow, dst, nullptr, nullptr, counters, 2, stream, wave_log, nullptr, nullptr, stride,
                       n_frames, dst_frame_stride);
}

// n_frames whole frames in one set of launches (vxrt_render_interleaved_batch with a single rank)
int vxrt_render_batch(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t n_frames, const vxrt_shade_params_t* params, int shadow,
                      uint32_t* dst, uint64_t dst_frame_stride, unsigned long long* rays_traced, void* stream) {
  return vxrt_render_interleaved_batch(accel, width, height, 0, 1, n_frames, params, shadow, dst, dst_frame_stride, rays_traced, stream);
}

// Same launches as vxrt_render with the fetch counters compiled in (slower; never the timed path).
// counters: device u64[7] = rays, node fetches, instance fetches, triangle fetches, shaded hits,
// textured hits, pixels written -- the inputs of the algorithmic-bytes formula.
int vxrt_render_stats(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                      const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                      unsigned long long* counters, void* stream) {
  return render_common(accel, width, height, y0, y1, params, shadow, dst, nullptr, nullptr, counters, true, stream);
}

// vxrt_render_stats for the traversal the TIMED kernel performs: occlusion rays of a frame visit children in slot order (the
// result is a boolean, the order cannot change it) and idle lanes test a leaf's second triangle, so node / triangle fetch
// counts differ from the reference-order counts of vxrt_render_stats; both are reported next to each other (bench.py)
int vxrt_render_stats_timed(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                            const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                            unsigned long long* counters, void* stream) {
  return render_common(accel, width, height, y0, y1, params, shadow, dst, nullptr, nullptr, counters, 2, stream);
}

// diagnostic: vxrt_render_stats that also logs, per wavefront of the main traversal launch, the first
// and last 100 MHz clock and the number of rays it started (wave_log: device u64[13 * waves], waves =
// 4 * blocks of the launch; 13 * 4 * 8 * 256 entries are always enough)
int vxrt_render_wave_log(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                         const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                         unsigned long long* counters, unsigned long long* wave_log, void* stream) {
  return render_common(accel, width, height, y0, y1, params, shadow, dst, nullptr, nullptr, counters, 2, stream, wave_log);   // the timed traversal
}

int vxrt_render_diffuse_bounce(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                               const vxrt_shade_params_t* params, uint32_t seed, uint32_t* dst, float* colors,
                               unsigned long long* rays_traced, void* stream) {
  vxrt_ao_params_t gi{};
  gi.spp = 1; gi.radius = RT_LARGE_FLOAT; gi.seed = seed; gi.reserved = VXRT_AO_MODE_DIFFUSE_BOUNCE;
  return render_common(accel, width, height, y0, y1, params, 0, dst, nullptr, colors, rays_traced, false, stream, nullptr, &gi, nullptr);
}

int vxrt_render_ao(vxrt_accel_t* accel, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                   const vxrt_shade_params_t* params, const vxrt_ao_params_t* ao, uint32_t* dst, float* colors,
                   uint32_t* unoccluded, unsigned long long* rays_traced, void* stream) {
  if (!ao || ao->spp == 0 || ao->spp > 4096 || !(ao->radius > 0.0f) || ao->reserved != 0) return -1;
  return render_common(accel, width, height, y0, y1, params, 0, dst, nullptr, colors, rays_traced, false, stream, nullptr, ao, unoccluded);
}

// vxrt_trace with the fetch counters compiled in (slower; never the timed path): counters = device u64[8],
// [0..3] = {rays, node fetches, instance fetches, triangle fetches}, the inputs of SURVEY s8d's per-ray formula
int vxrt_trace_stats(vxrt_accel_t* a, const float* rays, uint64_t n, const float* tmax,
                     vxrt_hit_t* hits, int mode, unsigned long long* counters, void* stream) {
  if (!a || !counters || (n && (!rays || !hits))) return -1;
  if (mode != VXRT_MODE_CLOSEST && mode != VXRT_MODE_ANY) return -1;
  if (n == 0) return 0;
  if (n > 0x7fffffffull) return -1;
  hipStream_t s = (hipStream_t)stream;
  FrameCtx* c = acquire_ctx(a, s);
  if (!c) return -1;
  // (a failure releases the context the way a success does: its event is recorded behind whatever was enqueued, see render_common)
  if (trace_on_ctx(a, c, rays, n, tmax, (HitRec*)hits, mode, s, nullptr, counters) != 0) { c->ctl_dirty = true; (void)release_ctx(a, c, s); return -1; }
  return release_ctx(a, c, s);
}

int vxrt_trace(vxrt_accel_t* a, const float* rays, uint64_t n, const float* tmax,
               vxrt_hit_t* hits, int mode, void* stream) {
  if (!a || (n && (!rays || !hits))) return -1;
  if (mode != VXRT_MODE_CLOSEST && mode != VXRT_MODE_ANY) return -1;
  if (n == 0) return 0;
  if (n > 0x7fffffffull) return -1;
  hipStream_t s = (hipStream_t)stream;
  FrameCtx* c = acquire_ctx(a, s);
  if (!c) return -1;
  if (trace_on_ctx(a, c, rays, n, tmax, (HitRec*)hits, mode, s) != 0) { c->ctl_dirty = true; (void)release_ctx(a, c, s); return -1; }
  return release_ctx(a, c, s);
}

// Opt-in compatibility mode: the reference RTU's traversal restated literally, quirks included, on a flat memory image (see
// rt_quirks_trace_kernel).  image: device memory of image_size bytes; the four offsets are what the RTX DCRs 0x6..0x9 hold in the
// simulator (32-bit addresses into its RAM).  hits: n records; a miss has dist 1e30.  mode: VXRT_MODE_CLOSEST = the fixed point of
// the accept loop, VXRT_MODE_ANY = the first accepted candidate.
int vxrt_trace_reference_quirks(const void* image, uint64_t image_size, uint32_t tlas_off, uint32_t blas_off, uint32_t bvh_off, uint32_t tri_off,
                                const float* rays, uint64_t n, const float* tmax, vxrt_hit_t* hits, int mode, void* stream) {
  if (!image || image_size < 64 || (n && (!rays || !hits))) return -1;
  if (mode != VXRT_MODE_CLOSEST && mode != VXRT_MODE_ANY) return -1;
  if (n == 0) return 0;
  if (n > 0x7fffffffull) return -1;
  uint32_t* st = status_word();
  if (!st) return -1;
  QuirkImage im{(const uint8_t*)image, image_size, tlas_off, blas_off, bvh_off, tri_off};
  hipLaunchKernelGGL(rt_quirks_trace_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, im, rays, tmax, n, (HitRec*)hits,
                     mode == VXRT_MODE_ANY ? 1 : 0, st);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Camera rays of rows [y0, y1) of a W x H frame as a ray buffer (6 floats per ray, ray (x, y) at x + (y - y0) * W): what the frame
// kernels trace, for callers that trace them through vxrt_trace / vxrt_trace_reference_quirks and shade with vxrt_shade_rays.
int vxrt_camera_rays(uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, float* rays, void* stream) {
  if (!rays || width == 0 || height == 0 || y0 > y1 || y1 > height) return -1;
  const uint64_t n = (uint64_t)width * (y1 - y0);
  if (n == 0) return 0;
  hipLaunchKernelGGL(rt_camera_rays_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, width, height, y0, n, rays);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int vxrt_shade_rays(vxrt_accel_t* a, const float* rays, const vxrt_hit_t* hits, uint64_t n, const vxrt_shade_params_t* params,
                    float* colors, uint32_t* rgb8, void* stream) {
  if (!a || !params || (n && (!rays || !hits)) || (!colors && !rgb8)) return -1;
  if (!a->ref.triEx || !a->ref.mat || a->ref.n_mats == 0) return -1;
  if (n == 0) return 0;
  if (n > 0x7fffffffull) return -1;
  // a hit record names a triangle and an instance: both must exist (records a caller made up are not trusted)
  ShadeParams p;
  for (int i = 0; i < 3; ++i) {
    p.amb[i] = params->ambient[i]; p.lcol[i] = params->light_color[i];
    p.lpos[i] = params->light_pos[i]; p.bg[i] = params->background[i];
  }
  p.max_depth = 1;
  hipLaunchKernelGGL(rt_shade_rays_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a->dev, p, n, rays, (const HitRec*)hits, colors, rgb8);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// diagnostic (tests): the control block of frame context `ctx` as the last call left it -- [0] deferral count, [32 + 32 k] the
// main launch's queue shard k, [32 + 256 + 32 k] the shards of the EXACT launch over the deferred list, [32 + 512 + 32 k] those of
// the a-priori EXACT launch.  vxrt_trace leaves the block for the next call to clear, so it can be inspected after a trace.
// diagnostic (tools/xcd_tail.py): from now on every main traversal launch on this layout -- the timed kernels included -- leaves, per wavefront,
// the constant 100 MHz clock at its end and its ray count | physical XCD << 56 in `log` (device memory, 2 u64 per wavefront, room for 8,192
// wavefronts: 16 x 8,192 u64; the caller zeroes it between the launches it wants to tell apart, and keeps it alive until they have run); nullptr
// switches it off again.  The EXACT launches do not write.  Costs the timed kernel one store per wavefront at its end.
int vxrt_wire_pack(const uint32_t* frames, uint64_t frame_stride, uint32_t width, uint32_t per, uint32_t world, uint32_t rank, uint32_t n_frames,
                   uint8_t* wire, void* stream) {
  if (!frames || !wire || width == 0 || (width & 3u) != 0 || per == 0 || world == 0 || rank >= world || n_frames == 0) return -1;
  if (((uintptr_t)frames & 15u) != 0 || ((uintptr_t)wire & 3u) != 0 || (frame_stride & 3u) != 0) return -1;   // (16-byte pixel quads, word stores)
  const uint64_t n = (uint64_t)n_frames * per * 8u * (width / 4u);
  if (n > 0xFFFFFFFFull * 256ull) return -1;
  hipLaunchKernelGGL(wire_pack_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames, frame_stride, width / 4u, per, world, rank, n, (uint32_t*)wire);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int vxrt_wire_unpack(const uint8_t* wire_all, uint64_t wire_stride_bytes, uint32_t width, uint32_t per, uint32_t world, uint32_t n_frames,
                     uint32_t* frames, uint64_t frame_stride, void* stream) {
  if (!frames || !wire_all || width == 0 || (width & 3u) != 0 || per == 0 || world == 0 || n_frames == 0) return -1;
  if (((uintptr_t)frames & 15u) != 0 || ((uintptr_t)wire_all & 3u) != 0 || (frame_stride & 3u) != 0 || (wire_stride_bytes & 3u) != 0) return -1;
  const uint64_t per_rank = (uint64_t)n_frames * per * 8u * (width / 4u);
  if (wire_stride_bytes < per_rank * 12u) return -1;
  const uint64_t n = per_rank * world;
  if (n > 0xFFFFFFFFull * 256ull) return -1;
  hipLaunchKernelGGL(wire_unpack_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)wire_all, wire_stride_bytes / 4u, width / 4u, per, world,
                     per_rank, frames, frame_stride);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// diagnostic (tools/trace_phases.py): the counting build's ray-buffer launches (vxrt_trace_stats) keep vxrt_render_wave_log's 16 u64 per wavefront in
// `log` from now on (device memory, 16 x 8,192 u64; nullptr: off) -- loop iterations, runs of the node / leaf body and the lanes active in them
int vxrt_debug_trace_wave_log(vxrt_accel_t* a, unsigned long long* log) {
  if (!a) return -1;
  a->trace_wave_log = log;
  return 0;
}

int vxrt_debug_end_log(vxrt_accel_t* a, unsigned long long* log) {
  if (!a) return -1;
  a->end_log = log;   // (captured by the launches enqueued from now on; launches already enqueued keep what they were given)
  return 0;
}

int vxrt_debug_read_control(vxrt_accel_t* a, uint32_t ctx, uint32_t* out, uint32_t n_dwords, void* stream) {
  if (!a || !out || ctx >= MAX_FRAMES_IN_FLIGHT || n_dwords > CTL_DWORDS || !a->ctx[ctx].ctl) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  return hipMemcpy(out, a->ctx[ctx].ctl, (size_t)n_dwords * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

// diagnostic (tools/tile_tail.py): what frame context `ctx` learned for sets of `batch` frames -- cost[n] (loop iterations of the wavefront
// that traced the tile in the last launch; after a wave-log launch n start clocks, n durations in 100 MHz clocks and n steal distances
// follow) and the order[n] derived from it.  Returns the capacity in tiles, -1 on error.
int vxrt_debug_read_lpt(vxrt_accel_t* a, uint32_t ctx, uint32_t batch, uint32_t* cost, uint32_t cost_cap, uint32_t* order, uint32_t order_cap, void* stream) {
  if (!a || ctx >= MAX_FRAMES_IN_FLIGHT || batch > VXRT_MAX_BATCH) return -1;
  FrameCtx::Lpt& L = a->ctx[ctx].lpt[batch];
  if (!L.cost || !L.order) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (cost && hipMemcpy(cost, L.cost, (size_t)std::min<uint32_t>(cost_cap, 4u * L.cap) * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (order && hipMemcpy(order, L.order, (size_t)std::min(order_cap, L.cap) * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)L.cap;
}

int vxrt_status(void* stream, uint32_t* status) {
  uint32_t* st = status_word();
  if (!st || !status) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (hipMemcpy(status, st, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  // read-and-clear: a failed run must not poison the runs after it (the word is per device, shared by all launches)
  if (*status != 0u && hipMemset(st, 0, sizeof(uint32_t)) != hipSuccess) return -1;
  return 0;
}

}  // extern "C"
