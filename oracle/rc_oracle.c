/* TEST INFRASTRUCTURE -- see rc_oracle.h.  Every function cites the reference lines it follows
 * (paths relative to tests/regression/raycast unless noted). */
#include "rc_oracle.h"
#include <math.h>
#include <string.h>

#define RC_LARGE_FLOAT 1e30f     /* geometry.h:15 */
#define RC_EPSILON 1e-6f         /* geometry.h:17 */
#define RC_STACK 64              /* render.h:5 BVH_STACK_SIZE */

typedef struct { float x, y, z; } v3;
static inline v3 v3m(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3p(const float* p) { return v3m(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return v3m(a.x + b.x, a.y + b.y, a.z + b.z); }          /* geometry.h:346 */
static inline v3 sub(v3 a, v3 b) { return v3m(a.x - b.x, a.y - b.y, a.z - b.z); }          /* :542 */
static inline v3 mulvs(v3 a, float b) { return v3m(a.x * b, a.y * b, a.z * b); }           /* :721 */
static inline v3 mulsv(float b, v3 a) { return v3m(b * a.x, b * a.y, b * a.z); }           /* :722 */
static inline v3 mulvv(v3 a, v3 b) { return v3m(a.x * b.x, a.y * b.y, a.z * b.z); }        /* :715 */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           /* :888 */
static inline v3 cross(v3 a, v3 b) { return v3m(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }   /* :897-899 */
static inline v3 normalize(v3 v) { float invLen = 1.0f / sqrtf(dot(v, v)); return mulvs(v, invLen); }                    /* :180,913-916 */
/* libstdc++ std::min / std::max */
static inline float std_min(float a, float b) { return (b < a) ? b : a; }
static inline float std_max(float a, float b) { return (a < b) ? b : a; }

/* float4(a, w) * M (geometry.h:1281-1286): row r of M dotted with (a, w) */
static inline v3 xform(v3 a, float w, const float* M) {
  return v3m(M[0] * a.x + M[1] * a.y + M[2] * a.z + M[3] * w,
             M[4] * a.x + M[5] * a.y + M[6] * a.z + M[7] * w,
             M[8] * a.x + M[9] * a.y + M[10] * a.z + M[11] * w);
}

/* geometry.h:1416-1440 ray_t::intersect(tri) */
static int ray_tri(v3 orig, v3 dir, const rc_tri_t* t, float* dist, v3* bc) {
  v3 v0 = v3p(t->v0), edge1 = sub(v3p(t->v1), v0), edge2 = sub(v3p(t->v2), v0);
  v3 h = cross(dir, edge2);
  float a = dot(edge1, h);
  if (fabsf(a) < RC_EPSILON) return 0;
  float f = 1 / a;
  v3 s = sub(orig, v0);
  float w1 = f * dot(s, h);
  if (w1 < 0 || w1 > 1) return 0;
  v3 q = cross(s, edge1);
  float w2 = f * dot(dir, q);
  if (w2 < 0 || w1 + w2 > 1) return 0;
  float tt = f * dot(edge2, q);
  if (tt <= RC_EPSILON) return 0;
  *dist = tt;
  bc->x = w1; bc->y = w2; bc->z = 1 - w1 - w2;
  return 1;
}

/* geometry.h:1442-1465 ray_t::intersect(aabb) */
static float ray_box(v3 orig, v3 dir, const float* mn, const float* mx) {
  float idir_x = 1.0f / dir.x, idir_y = 1.0f / dir.y, idir_z = 1.0f / dir.z;
  float tx1 = (mn[0] - orig.x) * idir_x, tx2 = (mx[0] - orig.x) * idir_x;
  float tmin = std_min(tx1, tx2), tmax = std_max(tx1, tx2);
  float ty1 = (mn[1] - orig.y) * idir_y, ty2 = (mx[1] - orig.y) * idir_y;
  tmin = std_max(tmin, std_min(ty1, ty2)); tmax = std_min(tmax, std_max(ty1, ty2));
  float tz1 = (mn[2] - orig.z) * idir_z, tz2 = (mx[2] - orig.z) * idir_z;
  tmin = std_max(tmin, std_min(tz1, tz2)); tmax = std_min(tmax, std_max(tz1, tz2));
  if (tmax < tmin || tmax <= 0) return RC_LARGE_FLOAT;
  return tmin;
}

/* render.h:75-124 BVHIntersect.  Note the comparison at :110: with both children hit and dLeft < dRight the
 * indices are swapped, so the FARTHER child is popped first (the TLAS loop at :176 uses '>'). */
static int bvh_intersect(v3 orig, v3 dir, uint32_t blasIdx, const rc_bvh_node_t* bvh, const uint32_t* triIdx,
                         const rc_tri_t* tri, rc_hit_t* hit) {
  uint32_t stack[RC_STACK];
  uint32_t sp = 0;
  stack[sp++] = 0;
  while (sp != 0) {
    uint32_t nodeIdx = stack[--sp];
    const rc_bvh_node_t* node = &bvh[nodeIdx];
    if (node->triCount != 0) {
      for (uint32_t i = 0; i < node->triCount; ++i) {
        uint32_t ti = triIdx[node->leftFirst + i];
        float dist; v3 bc;
        if (ray_tri(orig, dir, &tri[ti], &dist, &bc) && dist < hit->dist) {
          hit->dist = dist; hit->bx = bc.x; hit->by = bc.y; hit->bz = bc.z;
          hit->blasIdx = blasIdx; hit->triIdx = ti;
        }
      }
    } else {
      uint32_t left = node->leftFirst, right = left + 1;
      float dLeft = ray_box(orig, dir, bvh[left].aabbMin, bvh[left].aabbMax);
      float dRight = ray_box(orig, dir, bvh[right].aabbMin, bvh[right].aabbMax);
      int hitLeft = (dLeft != RC_LARGE_FLOAT) && (dLeft < hit->dist);
      int hitRight = (dRight != RC_LARGE_FLOAT) && (dRight < hit->dist);
      if (hitLeft && hitRight) {
        if (dLeft < dRight) { uint32_t t = left; left = right; right = t; }
        if (sp + 2 > RC_STACK) return -1;
        stack[sp++] = right;
        stack[sp++] = left;
      } else if (hitLeft) {
        if (sp + 1 > RC_STACK) return -1;
        stack[sp++] = left;
      } else if (hitRight) {
        if (sp + 1 > RC_STACK) return -1;
        stack[sp++] = right;
      }
    }
  }
  return 0;
}

/* render.h:126-141 BLASIntersect + :143-190 TLASIntersect */
int rc_trace(const rc_args_t* a, const float ray6[6], rc_hit_t* hit) {
  v3 orig = v3p(ray6), dir = v3p(ray6 + 3);
  hit->dist = RC_LARGE_FLOAT; hit->bx = hit->by = hit->bz = 0; hit->blasIdx = 0; hit->triIdx = 0;   /* common.h:24-29 */
  uint32_t stack[RC_STACK];
  uint32_t sp = 0;
  stack[sp++] = a->tlas_root;
  while (sp != 0) {
    uint32_t nodeIdx = stack[--sp];
    const rc_tlas_node_t* node = &a->tlas[nodeIdx];
    if (node->leftRight == 0) {
      const rc_blas_t* b = &a->blas[node->blasIdx];
      /* ray_t::transform (geometry.h:1411-1414): direction first, then origin */
      v3 d2 = xform(dir, 0.0f, b->invTransform);
      v3 o2 = xform(orig, 1.0f, b->invTransform);
      if (bvh_intersect(o2, d2, node->blasIdx, a->bvh + b->bvh_offset, a->triIdx, a->tri, hit) != 0) return -1;
    } else {
      uint32_t left = node->leftRight & 0xFFFF, right = node->leftRight >> 16;
      float dLeft = ray_box(orig, dir, a->tlas[left].aabbMin, a->tlas[left].aabbMax);
      float dRight = ray_box(orig, dir, a->tlas[right].aabbMin, a->tlas[right].aabbMax);
      int hitLeft = (dLeft != RC_LARGE_FLOAT) && (dLeft < hit->dist);
      int hitRight = (dRight != RC_LARGE_FLOAT) && (dRight < hit->dist);
      if (hitLeft && hitRight) {
        if (dLeft > dRight) { uint32_t t = left; left = right; right = t; }
        if (sp + 2 > RC_STACK) return -1;
        stack[sp++] = right;
        stack[sp++] = left;
      } else if (hitLeft) {
        if (sp + 1 > RC_STACK) return -1;
        stack[sp++] = left;
      } else if (hitRight) {
        if (sp + 1 > RC_STACK) return -1;
        stack[sp++] = right;
      }
    }
  }
  return 0;
}

/* render.h:192-211 GenerateRay */
void rc_generate_ray(const rc_args_t* a, uint32_t x, uint32_t y, float out6[6]) {
  float x_ndc = (float)((double)(((float)x + 0.5f) / (float)a->dst_width) - 0.5);
  float y_ndc = (float)((double)(((float)y + 0.5f) / (float)a->dst_height) - 0.5);
  float x_vp = x_ndc * a->viewplane[0];
  float y_vp = y_ndc * a->viewplane[1];
  v3 pos = v3p(a->camera_pos);
  v3 pt_cam = add(add(mulsv(x_vp, v3p(a->camera_right)), mulsv(y_vp, v3p(a->camera_up))), v3p(a->camera_forward));
  v3 pt_w = add(pt_cam, pos);
  v3 d = normalize(sub(pt_w, pos));
  out6[0] = pos.x; out6[1] = pos.y; out6[2] = pos.z; out6[3] = d.x; out6[4] = d.y; out6[5] = d.z;
}

static inline uint32_t f2u_x86(float f) { return (uint32_t)(int64_t)f; }   /* uint32_t(float) as x86-64 g++ lowers it */

/* render.h:213-275 Trace */
int rc_radiance(const rc_args_t* a, const float ray6[6], float out3[3]) {
  v3 orig = v3p(ray6), dir = v3p(ray6 + 3);
  v3 radiance = v3m(0, 0, 0);
  float throughput = 1.0f;
  v3 bg = v3p(a->background_color);
  for (uint32_t bounce = 0; bounce < a->max_depth; ++bounce) {
    rc_hit_t hit;
    float r6[6] = {orig.x, orig.y, orig.z, dir.x, dir.y, dir.z};
    if (rc_trace(a, r6, &hit) != 0) return -1;
    if (hit.dist == RC_LARGE_FLOAT) {
      radiance = add(radiance, mulvs(bg, throughput));                                   /* :230 */
      break;
    }
    const rc_blas_t* blas = &a->blas[hit.blasIdx];
    const rc_triex_t* te = &a->triEx[hit.triIdx];
    v3 I = add(orig, mulvs(dir, hit.dist));                                               /* :239 */
    v3 N = add(add(mulvs(v3p(te->N1), hit.bx), mulvs(v3p(te->N2), hit.by)), mulvs(v3p(te->N0), hit.bz));   /* :242 */
    /* :243-244  transposed() copies the 3x3 block into an identity (geometry.h:1141-1147); float4(N,0) * M */
    const float* m = blas->invTransform;
    v3 Nt = v3m(m[0] * N.x + m[4] * N.y + m[8] * N.z + 0.0f * 0.0f,
                m[1] * N.x + m[5] * N.y + m[9] * N.z + 0.0f * 0.0f,
                m[2] * N.x + m[6] * N.y + m[10] * N.z + 0.0f * 0.0f);
    N = normalize(Nt);
    float uvx = te->uv1[0] * hit.bx + te->uv2[0] * hit.by + te->uv0[0] * hit.bz;         /* :247 */
    float uvy = te->uv1[1] * hit.bx + te->uv2[1] * hit.by + te->uv0[1] * hit.bz;
    const uint32_t* px = (const uint32_t*)(a->tex + blas->tex_offset);                   /* :250-251, texSample :8-22 */
    uint32_t iu = f2u_x86(uvx * (float)blas->tex_width), iv = f2u_x86(uvy * (float)blas->tex_height);
    iu %= blas->tex_width; iv %= blas->tex_height;
    uint32_t texel = px[iu + iv * blas->tex_width];
    float sc = 1 / 256.0f;                                                                /* common.h:114-120 */
    v3 texColor = v3m((float)(int)((texel >> 16) & 255) * sc, (float)(int)((texel >> 8) & 255) * sc, (float)(int)(texel & 255) * sc);
    /* diffuseLighting :59-71 */
    v3 L = sub(v3p(a->light_pos), I);
    float dist = sqrtf(dot(L, L));
    L = mulvs(L, 1.0f / dist);
    float att = 1.0f / (1.0f + dist * 0.1f);
    float NdotL = std_max(0.0f, dot(N, L));
    v3 diffuse = mulvv(texColor, add(v3p(a->ambient_color), mulvs(mulsv(att, v3p(a->light_color)), NdotL)));
    float reflectivity = blas->reflectivity;
    radiance = add(radiance, mulvs(mulsv(throughput, diffuse), 1 - reflectivity));        /* :257 */
    throughput *= reflectivity;                                                           /* :260 */
    if (reflectivity > 0.0f && bounce + 1 < a->max_depth) {                               /* :263-268 */
      v3 R = normalize(sub(dir, mulvs(mulsv(2.0f, N), dot(N, dir))));
      orig = add(I, mulvs(R, 0.001f));
      dir = R;
      continue;
    }
    radiance = add(radiance, mulsv(throughput, bg));                                      /* :271 */
    break;
  }
  out3[0] = radiance.x; out3[1] = radiance.y; out3[2] = radiance.z;
  return 0;
}

/* kernel.cpp:9-33 (tracer.cpp:249-263 is the same loop): samples are identical rays, their colours are summed */
int rc_render(const rc_args_t* a, uint32_t y0, uint32_t y1, uint32_t* out_pixels, float* out_color) {
  for (uint32_t y = y0; y < y1; ++y) {
    for (uint32_t x = 0; x < a->dst_width; ++x) {
      v3 color = v3m(0, 0, 0);
      for (uint32_t s = 0; s < a->samples_per_pixel; ++s) {
        float ray[6], c[3];
        rc_generate_ray(a, x, y, ray);
        if (rc_radiance(a, ray, c) != 0) return -1;
        color = add(color, v3m(c[0], c[1], c[2]));
      }
      uint32_t idx = x + y * a->dst_width;
      int r = (int)(std_min(color.x, 1.f) * 255), g = (int)(std_min(color.y, 1.f) * 255), b = (int)(std_min(color.z, 1.f) * 255);   /* common.h:107-112 */
      out_pixels[idx] = (uint32_t)((r << 16) + (g << 8) + b);
      if (out_color) { out_color[3 * idx] = color.x; out_color[3 * idx + 1] = color.y; out_color[3 * idx + 2] = color.z; }
    }
  }
  return 0;
}
