// HIP kernels of the ray-tracing hot path for gfx950 (MI355X): 4-wide quantized-BVH traversal
// (TLAS -> BLAS), Moller-Trumbore, Lambert shade, RGB8 pack.  Hand-written for CDNA4 wave64:
// one 8x8 pixel tile (the reference's block, kernel.cpp:128-133) == one wavefront.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off   (contraction OFF is part of the
// contract: SURVEY.md s7 "FP contraction"; division and sqrt are the correctly rounded forms,
// hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
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

#define TLAS_FLAG 0x80000000u
#define STATUS_STACK_OVERFLOW 1u

struct SceneDev {
  const uint32_t* tlas;   // 13 dwords per node
  const uint32_t* bvh;
  const uint32_t* blas;   // 40 dwords per record
  const float* tri;       // 9 floats per triangle
  const rt_triex_t* triEx;
  const rt_material_t* mat;
  const uint8_t* tex;
};

struct HitRec { float dist, bx, by, bz; uint32_t blasIdx, triIdx; };

// libstdc++ std::min / std::max (rt_traversal.cpp:327-337 use them; NaN behaviour is part of parity)
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

__device__ __forceinline__ float ubyte_f(const uint32_t* w, int byte_off) {
  return (float)((w[byte_off >> 2] >> ((byte_off & 3) * 8)) & 0xffu);
}
__device__ __forceinline__ uint32_t ubyte_u(const uint32_t* w, int byte_off) {
  return (w[byte_off >> 2] >> ((byte_off & 3) * 8)) & 0xffu;
}

// rt_traversal.cpp:318-339 with idir hoisted (1.0f/rd is recomputed per child there; same value).
// EXACT selects the libstdc++ min/max forms; the fast form uses v_min/v_max, which differs only
// when a NaN is present (0*inf), i.e. only if some ray direction component is 0/inf/NaN.
template <bool EXACT>
__device__ __forceinline__ float ray_box(float ox, float oy, float oz, float ix, float iy, float iz,
                                         float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
  float tx1 = (mnx - ox) * ix, tx2 = (mxx - ox) * ix;
  float ty1 = (mny - oy) * iy, ty2 = (mxy - oy) * iy;
  float tz1 = (mnz - oz) * iz, tz2 = (mxz - oz) * iz;
  float tmin, tmax;
  if (EXACT) {
    tmin = std_min(tx1, tx2);
    tmax = std_max(tx1, tx2);
    tmin = std_max(tmin, std_min(ty1, ty2));
    tmax = std_min(tmax, std_max(ty1, ty2));
    tmin = std_max(tmin, std_min(tz1, tz2));
    tmax = std_min(tmax, std_max(tz1, tz2));
  } else {
    tmin = fminf(tx1, tx2);
    tmax = fmaxf(tx1, tx2);
    tmin = fmaxf(tmin, fminf(ty1, ty2));
    tmax = fminf(tmax, fmaxf(ty1, ty2));
    tmin = fmaxf(tmin, fminf(tz1, tz2));
    tmax = fminf(tmax, fmaxf(tz1, tz2));
  }
  return (tmax < tmin || tmax <= 0) ? RT_LARGE_FLOAT : tmin;
}

// rt_traversal.cpp:263-316
__device__ __forceinline__ float ray_tri(float ox, float oy, float oz, float dx, float dy, float dz,
                                         const float* __restrict__ t, float& bx, float& by, float& bz) {
  float v0x = t[0], v0y = t[1], v0z = t[2];
  float e1x = t[3] - v0x, e1y = t[4] - v0y, e1z = t[5] - v0z;
  float e2x = t[6] - v0x, e2y = t[7] - v0y, e2z = t[8] - v0z;
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

struct Cand { float d; uint32_t idx; };
// visit order: nearer first; equal distance -> higher child index first (stable far->near sort of
// rt_traversal.cpp:76-78 read from the back).  Filtered children carry d = +inf.
__device__ __forceinline__ void cmpx(Cand& a, Cand& b) {
  bool sw = (b.d < a.d) || (b.d == a.d && b.idx > a.idx);
  Cand ta = a, tb = b;
  a.d = sw ? tb.d : ta.d; a.idx = sw ? tb.idx : ta.idx;
  b.d = sw ? ta.d : tb.d; b.idx = sw ? ta.idx : tb.idx;
}

// Box tests of the <=4 children of an internal node (rt_traversal.cpp:59-74).
template <bool EXACT>
__device__ __forceinline__ void eval_children(const uint32_t* w, float px, float py, float pz, int ex, int ey, int ez,
                                              float rox, float roy, float roz, float rix, float riy, float riz,
                                              float hit_dist, Cand* c) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int b = 24 + 7 * k;
    const uint32_t meta = ubyte_u(w, b);
    // :61-67  origin + ldexp(float(q), e)
    float mnx = px + ldexpf(ubyte_f(w, b + 1), ex);
    float mny = py + ldexpf(ubyte_f(w, b + 2), ey);
    float mnz = pz + ldexpf(ubyte_f(w, b + 3), ez);
    float mxx = px + ldexpf(ubyte_f(w, b + 4), ex);
    float mxy = py + ldexpf(ubyte_f(w, b + 5), ey);
    float mxz = pz + ldexpf(ubyte_f(w, b + 6), ez);
    float d = ray_box<EXACT>(rox, roy, roz, rix, riy, riz, mnx, mny, mnz, mxx, mxy, mxz);
    bool ok = (meta != 0u) && (d < hit_dist);             // :60, :71
    c[k].d = ok ? d : __builtin_inff();
    c[k].idx = (uint32_t)k;
  }
}

// per-lane fetch counters of the STATS build (algorithmic bytes): what the
// reference logs in RT_mem_accesses (rt_traversal.cpp:54,116,148,158), without restart re-reads
struct Fetches { unsigned node = 0, inst = 0, tri = 0; };

// One closest-hit (or any-hit) query.  Returns true if a candidate was accepted.
template <bool ANY_HIT, bool STATS = false>
__device__ bool trace_ray(const SceneDev& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                          float tmax, HitRec& hit, uint32_t* status, Fetches* fx = nullptr) {
  // world-space ray (TLAS nodes) and object-space ray (BLAS nodes), with reciprocals
  const float wix = 1.0f / dx, wiy = 1.0f / dy, wiz = 1.0f / dz;
  float cox = ox, coy = oy, coz = oz, cdx = dx, cdy = dy, cdz = dz;
  float cix = wix, ciy = wiy, ciz = wiz;

  // a lane may use v_min/v_max only if no slab product can be NaN
  bool lane_fast = (wix - wix == 0.0f) && (wiy - wiy == 0.0f) && (wiz - wiz == 0.0f) &&
                   (ox - ox == 0.0f) && (oy - oy == 0.0f) && (oz - oz == 0.0f);

  uint32_t stk_node[RT_STACK_ENTRIES];
  float stk_m[RT_STACK_ENTRIES];
  int sp = 0;

  hit.dist = tmax; hit.bx = 0; hit.by = 0; hit.bz = 0; hit.blasIdx = 0; hit.triIdx = 0;
  bool found = false;
  uint32_t blasIdx = 0;
  uint32_t bvh_off = 0;          // node offset of the instance being traversed
  uint32_t cur = TLAS_FLAG | 0u; // TLAS root (rt_traversal.cpp:39-40)
  float path_m = -__builtin_inff();
  bool have = true;

  while (have) {
    const bool top_addr = (cur & TLAS_FLAG) != 0;
    const uint32_t* np = top_addr ? sc.tlas + (size_t)(cur & ~TLAS_FLAG) * RT_NODE_DWORDS
                                  : sc.bvh + (size_t)(bvh_off + cur) * RT_NODE_DWORDS;
    uint32_t w[RT_NODE_DWORDS];
#pragma unroll
    for (int i = 0; i < RT_NODE_DWORDS; ++i) w[i] = np[i];
    if (STATS) fx->node++;

    const float px = __uint_as_float(w[0]), py = __uint_as_float(w[1]), pz = __uint_as_float(w[2]);
    const int ex = (int)(int8_t)(w[3] & 0xff), ey = (int)(int8_t)((w[3] >> 8) & 0xff), ez = (int)(int8_t)((w[3] >> 16) & 0xff);
    const bool istop = (w[3] >> 24) == 1u;                   // isTopLevel (:219-221)
    const uint32_t leftFirst = w[4], leafData = w[5];
    const bool leaf = istop ? (leafData != 0xffffffffu) : (leafData != 0u); // isLeaf (:223-225)
    bool descend = false;

    if (!leaf) {
      const float rox = istop ? ox : cox, roy = istop ? oy : coy, roz = istop ? oz : coz;
      const float rix = istop ? wix : cix, riy = istop ? wiy : ciy, riz = istop ? wiz : ciz;
      Cand c[4];
      // wave-uniform choice: v_min/v_max slabs unless some active lane could see a NaN product
      if (__all(lane_fast)) eval_children<false>(w, px, py, pz, ex, ey, ez, rox, roy, roz, rix, riy, riz, hit.dist, c);
      else                  eval_children<true>(w, px, py, pz, ex, ey, ez, rox, roy, roz, rix, riy, riz, hit.dist, c);
      int n = (c[0].d < __builtin_inff()) + (c[1].d < __builtin_inff()) + (c[2].d < __builtin_inff()) + (c[3].d < __builtin_inff());
      cmpx(c[0], c[1]); cmpx(c[2], c[3]); cmpx(c[0], c[2]); cmpx(c[1], c[3]); cmpx(c[1], c[2]);
      if (n > 0) {
        const uint32_t fl = cur & TLAS_FLAG;
        if (sp + 3 > RT_STACK_ENTRIES) { atomicOr(status, STATUS_STACK_OVERFLOW); n = 1; }
        // far first so that the nearest pending sibling is on top (:98-103)
        if (n > 3) { stk_node[sp] = fl | (leftFirst + c[3].idx); stk_m[sp] = fmaxf(path_m, c[3].d); ++sp; }
        if (n > 2) { stk_node[sp] = fl | (leftFirst + c[2].idx); stk_m[sp] = fmaxf(path_m, c[2].d); ++sp; }
        if (n > 1) { stk_node[sp] = fl | (leftFirst + c[1].idx); stk_m[sp] = fmaxf(path_m, c[1].d); ++sp; }
        cur = fl | (leftFirst + c[0].idx);
        path_m = fmaxf(path_m, c[0].d);
        descend = true;
      }
    } else if (istop) {
      // TLAS leaf (:109-121): fetch the instance record, move the ray to object space
      blasIdx = leafData;
      const uint32_t* bp = sc.blas + (size_t)blasIdx * (RT_BLAS_STRIDE / 4);
      uint32_t bw[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) bw[i] = bp[i];
      if (STATS) fx->inst++;
      const float m00 = __uint_as_float(bw[1]), m01 = __uint_as_float(bw[2]), m02 = __uint_as_float(bw[3]), m03 = __uint_as_float(bw[4]);
      const float m10 = __uint_as_float(bw[5]), m11 = __uint_as_float(bw[6]), m12 = __uint_as_float(bw[7]), m13 = __uint_as_float(bw[8]);
      const float m20 = __uint_as_float(bw[9]), m21 = __uint_as_float(bw[10]), m22 = __uint_as_float(bw[11]), m23 = __uint_as_float(bw[12]);
      cox = m00 * ox + m01 * oy + m02 * oz + m03;   // :231-261
      coy = m10 * ox + m11 * oy + m12 * oz + m13;
      coz = m20 * ox + m21 * oy + m22 * oz + m23;
      cdx = m00 * dx + m01 * dy + m02 * dz;
      cdy = m10 * dx + m11 * dy + m12 * dz;
      cdz = m20 * dx + m21 * dy + m22 * dz;
      cix = 1.0f / cdx; ciy = 1.0f / cdy; ciz = 1.0f / cdz;
      const bool s2 = (cix - cix == 0.0f) && (ciy - ciy == 0.0f) && (ciz - ciz == 0.0f) &&
                      (cox - cox == 0.0f) && (coy - coy == 0.0f) && (coz - coz == 0.0f);
      lane_fast = lane_fast && s2;
      bvh_off = bw[0];
      cur = 0u;  // BLAS root; same level, path_m unchanged
      descend = true;
    } else {
      // BLAS leaf (:123-161): triangles in index order, strict '<'
      const uint32_t triCount = leafData;
      for (uint32_t i = 0; i < triCount; ++i) {
        const uint32_t triIdx = leftFirst + i;
        const float* tp = sc.tri + (size_t)triIdx * 9;
        float t[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) t[j] = tp[j];
        if (STATS) fx->tri++;
        float bx, by, bz;
        float d = ray_tri(cox, coy, coz, cdx, cdy, cdz, t, bx, by, bz);
        if (d < hit.dist) {
          hit.dist = d; hit.bx = bx; hit.by = by; hit.bz = bz;
          hit.blasIdx = blasIdx; hit.triIdx = triIdx;
          found = true;
          if (ANY_HIT) break;
          // the reference re-descends from the root with the shrunken hit.dist; if any box on the
          // current path no longer passes `d < hit.dist` it abandons this subtree
          if (!(path_m < hit.dist)) break;
        }
      }
      if (ANY_HIT && found) break;
    }

    if (!descend) {
      have = false;
      while (sp > 0) {
        --sp;
        const float m = stk_m[sp];
        if (m < hit.dist) { cur = stk_node[sp]; path_m = m; have = true; break; }
      }
    }
  }
  if (!found) hit.dist = RT_LARGE_FLOAT;
  return found;
}

// ---------------------------------------------------------------------------------------------
// shading (closest.cpp:57-127 / miss.cpp:9-14)
// ---------------------------------------------------------------------------------------------
struct ShadeParams { float amb[3], lcol[3], lpos[3], bg[3]; uint32_t max_depth; };

__device__ __forceinline__ uint32_t f2u_x86(float f) { return (uint32_t)(long long)f; } // rtx_shading.h:7-8 as x86-64 g++ lowers it

template <bool SHADOW, bool STATS = false>
__device__ void shade(const SceneDev& sc, const ShadeParams& p, float ox, float oy, float oz,
                      float dx, float dy, float dz, const HitRec& hit, bool found,
                      float& r, float& g, float& b, uint32_t* status, unsigned& extra_rays,
                      Fetches* fx = nullptr, unsigned* textured = nullptr) {
  if (!found) { r = p.bg[0]; g = p.bg[1]; b = p.bg[2]; return; }
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
  if (SHADOW) {
    // extension (no reference counterpart): one occlusion ray toward the light; occluded -> no
    // direct term.  Origin pushed 1e-3 along L like the reference's mirror bounce (closest.cpp:104).
    HitRec sh;
    bool occ = trace_ray<true, STATS>(sc, Ix + Lx * 0.001f, Iy + Ly * 0.001f, Iz + Lz * 0.001f, Lx, Ly, Lz, dist, sh, status, fx);
    extra_rays += 1;
    if (occ) NdotL = 0.0f;
  }
  const float dr = cr * (p.amb[0] + att * p.lcol[0] * NdotL);
  const float dg = cg * (p.amb[1] + att * p.lcol[1] * NdotL);
  const float db = cb * (p.amb[2] + att * p.lcol[2] * NdotL);
  const float refl = __uint_as_float(bp[38]);   // blas_node_t::reflectivity @152
  float thr = 1.0f;
  r = 0.0f + thr * dr * (1 - refl);             // :87
  g = 0.0f + thr * dg * (1 - refl);
  b = 0.0f + thr * db * (1 - refl);
  thr *= refl;                                  // :90
  r = r + p.bg[0] * thr;                        // :123 (no secondary ray: scene.cpp:96 sets reflectivity 0)
  g = g + p.bg[1] * thr;
  b = b + p.bg[2] * thr;
}

__device__ __forceinline__ uint32_t pack_rgb8(float r, float g, float b) {  // common.h:149-154
  int ir = (int)(std_min(r, 1.f) * 255);
  int ig = (int)(std_min(g, 1.f) * 255);
  int ib = (int)(std_min(b, 1.f) * 255);
  return (uint32_t)((ir << 16) + (ig << 8) + ib);
}

// kernel.cpp:28-39 -- u and v are evaluated in double, then rounded to f32
__device__ __forceinline__ void generate_ray(uint32_t x, uint32_t y, uint32_t W, uint32_t H,
                                             float& ox, float& oy, float& oz, float& dx, float& dy, float& dz) {
  const float u = (float)(((double)x * 2.0 - (double)W) / (double)H);
  const float v = (float)(((double)y * 2.0 - (double)H) / (double)H);
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

// One wavefront == one 8x8 tile (block of the reference grid); 4 tiles per 256-thread workgroup.
template <bool SHADOW, bool STATS = false>
__global__ __launch_bounds__(256) void rt_render_kernel(SceneDev sc, ShadeParams p, uint32_t W, uint32_t H,
                                                        uint32_t y0, uint32_t tiles_x, uint32_t n_tiles,
                                                        uint32_t y1, uint32_t* __restrict__ dst,
                                                        HitRec* __restrict__ hits, float* __restrict__ colors,
                                                        unsigned long long* rays_traced, uint32_t* status) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t x = tx * 8u + (lane & 7u);
  const uint32_t y = y0 + ty * 8u + (lane >> 3);
  const bool active = (x < W) && (y < y1);   // kernel.cpp:62,101
  unsigned nrays = 0, nhit = 0, ntex = 0;
  Fetches fx;
  if (active) {
    float ox, oy, oz, dx, dy, dz;
    generate_ray(x, y, W, H, ox, oy, oz, dx, dy, dz);
    HitRec hit;
    bool found = trace_ray<false, STATS>(sc, ox, oy, oz, dx, dy, dz, RT_LARGE_FLOAT, hit, status, &fx);
    nrays = 1;
    nhit = found ? 1u : 0u;
    float r, g, b;
    shade<SHADOW, STATS>(sc, p, ox, oy, oz, dx, dy, dz, hit, found, r, g, b, status, nrays, &fx, &ntex);
    const size_t idx = (size_t)x + (size_t)y * W;
    dst[idx] = pack_rgb8(r, g, b);
    if (hits) hits[idx] = hit;
    if (colors) { colors[3 * idx] = r; colors[3 * idx + 1] = g; colors[3 * idx + 2] = b; }
  }
  if (rays_traced) {
    // wave-level reduction, one atomic per wavefront and counter.  STATS build: rays_traced[0..6] =
    // rays, node fetches, instance fetches, triangle fetches, shaded hits, textured hits, pixels
    unsigned v[7] = {nrays, fx.node, fx.inst, fx.tri, nhit, ntex, active ? 1u : 0u};
#pragma unroll
    for (int k = 0; k < (STATS ? 7 : 1); ++k) {
      unsigned s = v[k];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
      if (lane == 0 && s) atomicAdd(rays_traced + k, (unsigned long long)s);
    }
  }
}

template <bool ANY_HIT>
__global__ __launch_bounds__(256) void rt_trace_kernel(SceneDev sc, const float* __restrict__ rays, uint64_t n,
                                                       const float* __restrict__ tmax, HitRec* __restrict__ hits,
                                                       uint32_t* status) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float* rp = rays + i * 6;
  float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
  HitRec hit;
  trace_ray<ANY_HIT>(sc, ox, oy, oz, dx, dy, dz, tmax ? tmax[i] : RT_LARGE_FLOAT, hit, status);
  hits[i] = hit;
}

// ---------------------------------------------------------------------------------------------
// host entry points (C ABI, include/vortex_hip.h level 2)
// ---------------------------------------------------------------------------------------------
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

static int check_scene(const vxrt_scene_t* s, SceneDev* d) {
  if (!s || !s->tlas || !s->blas || !s->bvh || !s->tri) return -1;
  if (s->n_tlas_nodes == 0 || s->n_blas == 0 || s->n_bvh_nodes == 0 || s->n_tris == 0) return -1;
  if (s->n_tlas_nodes >= 0x80000000u || s->n_bvh_nodes >= 0x80000000u) return -1;
  d->tlas = (const uint32_t*)s->tlas;
  d->bvh = (const uint32_t*)s->bvh;
  d->blas = (const uint32_t*)s->blas;
  d->tri = (const float*)s->tri;
  d->triEx = (const rt_triex_t*)s->triEx;
  d->mat = (const rt_material_t*)s->mat;
  d->tex = (const uint8_t*)s->tex;
  return 0;
}

extern "C" {

const char* vxrt_version(void) { return "vortex-rt-mi355x 0.1 (gfx950)"; }

int vxrt_render(const vxrt_scene_t* scene, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                const vxrt_shade_params_t* params, int shadow, uint32_t* dst, vxrt_hit_t* hits,
                float* colors, unsigned long long* rays_traced, void* stream) {
  SceneDev sc;
  if (check_scene(scene, &sc) != 0 || !params || !dst) return -1;
  if (!scene->triEx || !scene->mat || scene->n_mats == 0) return -1;  // shading needs them (closest.cpp:52-55)
  if (width == 0 || height == 0 || y0 > y1 || y1 > height) return -1;
  if (y0 == y1) return 0;
  uint32_t* st = status_word();
  if (!st) return -1;
  ShadeParams p;
  for (int i = 0; i < 3; ++i) {
    p.amb[i] = params->ambient[i]; p.lcol[i] = params->light_color[i];
    p.lpos[i] = params->light_pos[i]; p.bg[i] = params->background[i];
  }
  p.max_depth = params->max_depth;
  const uint32_t tiles_x = (width + 7) / 8, tiles_y = (y1 - y0 + 7) / 8;
  const uint64_t n_tiles64 = (uint64_t)tiles_x * tiles_y;
  if (n_tiles64 > 0x7fffffffull) return -1;
  const uint32_t n_tiles = (uint32_t)n_tiles64;
  dim3 grid((n_tiles + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (shadow)
    hipLaunchKernelGGL(rt_render_kernel<true>, grid, block, 0, s, sc, p, width, height, y0, tiles_x, n_tiles, y1,
                       dst, (HitRec*)hits, colors, rays_traced, st);
  else
    hipLaunchKernelGGL(rt_render_kernel<false>, grid, block, 0, s, sc, p, width, height, y0, tiles_x, n_tiles, y1,
                       dst, (HitRec*)hits, colors, rays_traced, st);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Same launch as vxrt_render with the fetch counters compiled in (slower; never the timed path).
// counters: device u64[7] = rays, node fetches, instance fetches, triangle fetches, shaded hits,
// textured hits, pixels written -- the inputs of the algorithmic-bytes formula.
int vxrt_render_stats(const vxrt_scene_t* scene, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                      const vxrt_shade_params_t* params, int shadow, uint32_t* dst,
                      unsigned long long* counters, void* stream) {
  SceneDev sc;
  if (check_scene(scene, &sc) != 0 || !params || !dst || !counters) return -1;
  if (!scene->triEx || !scene->mat || scene->n_mats == 0) return -1;
  if (width == 0 || height == 0 || y0 > y1 || y1 > height) return -1;
  if (y0 == y1) return 0;
  uint32_t* st = status_word();
  if (!st) return -1;
  ShadeParams p;
  for (int i = 0; i < 3; ++i) {
    p.amb[i] = params->ambient[i]; p.lcol[i] = params->light_color[i];
    p.lpos[i] = params->light_pos[i]; p.bg[i] = params->background[i];
  }
  p.max_depth = params->max_depth;
  const uint32_t tiles_x = (width + 7) / 8, tiles_y = (y1 - y0 + 7) / 8;
  const uint64_t n_tiles64 = (uint64_t)tiles_x * tiles_y;
  if (n_tiles64 > 0x7fffffffull) return -1;
  const uint32_t n_tiles = (uint32_t)n_tiles64;
  dim3 grid((n_tiles + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (shadow)
    hipLaunchKernelGGL((rt_render_kernel<true, true>), grid, block, 0, s, sc, p, width, height, y0, tiles_x, n_tiles, y1,
                       dst, (HitRec*)nullptr, (float*)nullptr, counters, st);
  else
    hipLaunchKernelGGL((rt_render_kernel<false, true>), grid, block, 0, s, sc, p, width, height, y0, tiles_x, n_tiles, y1,
                       dst, (HitRec*)nullptr, (float*)nullptr, counters, st);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int vxrt_trace(const vxrt_scene_t* scene, const float* rays, uint64_t n, const float* tmax,
               vxrt_hit_t* hits, int mode, void* stream) {
  SceneDev sc;
  if (check_scene(scene, &sc) != 0 || (n && (!rays || !hits))) return -1;
  if (mode != VXRT_MODE_CLOSEST && mode != VXRT_MODE_ANY) return -1;
  if (n == 0) return 0;
  if ((n + 255) / 256 > 0x7fffffffull) return -1;
  uint32_t* st = status_word();
  if (!st) return -1;
  dim3 grid((uint32_t)((n + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (mode == VXRT_MODE_ANY)
    hipLaunchKernelGGL(rt_trace_kernel<true>, grid, block, 0, s, sc, rays, n, tmax, (HitRec*)hits, st);
  else
    hipLaunchKernelGGL(rt_trace_kernel<false>, grid, block, 0, s, sc, rays, n, tmax, (HitRec*)hits, st);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int vxrt_status(void* stream, uint32_t* status) {
  uint32_t* st = status_word();
  if (!st || !status) return -1;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (hipMemcpy(status, st, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return 0;
}

}  // extern "C"
