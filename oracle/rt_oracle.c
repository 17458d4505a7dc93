/*
 * oracle/rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.  See rt_oracle.h.
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fPIC -shared (no -march=native, no -ffast-math).
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#include "rt_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#define EPSILON 1e-6f /* rt_traversal.cpp:7 */

/* std::min / std::max exactly as libstdc++ defines them (NaN behaviour matters):
 *   min(a,b) = (b < a) ? b : a ;  max(a,b) = (a < b) ? b : a            */
static inline float std_min(float a, float b) { return (b < a) ? b : a; }
static inline float std_max(float a, float b) { return (a < b) ? b : a; }

/* ---------------------------------------------------------------------------------------------
 * rt_traversal.cpp:318-339  BVHTraverser::ray_box_intersect
 * ------------------------------------------------------------------------------------------- */
float orc_ray_box(const float ray[6], float min_x, float min_y, float min_z,
                  float max_x, float max_y, float max_z) {
  float ro_x = ray[0], ro_y = ray[1], ro_z = ray[2];
  float rd_x = ray[3], rd_y = ray[4], rd_z = ray[5];
  float idir_x, idir_y, idir_z, tmin, tmax, tx1, tx2, ty1, ty2, tz1, tz2;
  idir_x = 1.0f / rd_x;
  idir_y = 1.0f / rd_y;
  idir_z = 1.0f / rd_z;
  tx1 = (min_x - ro_x) * idir_x;
  tx2 = (max_x - ro_x) * idir_x;
  tmin = std_min(tx1, tx2);
  tmax = std_max(tx1, tx2);
  ty1 = (min_y - ro_y) * idir_y;
  ty2 = (max_y - ro_y) * idir_y;
  tmin = std_max(tmin, std_min(ty1, ty2));
  tmax = std_min(tmax, std_max(ty1, ty2));
  tz1 = (min_z - ro_z) * idir_z;
  tz2 = (max_z - ro_z) * idir_z;
  tmin = std_max(tmin, std_min(tz1, tz2));
  tmax = std_min(tmax, std_max(tz1, tz2));
  return (tmax < tmin || tmax <= 0) ? ORC_LARGE_FLOAT : tmin;
}

/* ---------------------------------------------------------------------------------------------
 * rt_traversal.cpp:263-316  BVHTraverser::ray_tri_intersect (Moller-Trumbore)
 * ------------------------------------------------------------------------------------------- */
float orc_ray_tri(const float ray[6], const orc_tri_t* tri, float* bx, float* by, float* bz) {
  float v0_x = tri->v0[0], v0_y = tri->v0[1], v0_z = tri->v0[2];
  float v1_x = tri->v1[0], v1_y = tri->v1[1], v1_z = tri->v1[2];
  float v2_x = tri->v2[0], v2_y = tri->v2[1], v2_z = tri->v2[2];
  float ro_x = ray[0], ro_y = ray[1], ro_z = ray[2];
  float rd_x = ray[3], rd_y = ray[4], rd_z = ray[5];

  float edge1_x = v1_x - v0_x, edge1_y = v1_y - v0_y, edge1_z = v1_z - v0_z;
  float edge2_x = v2_x - v0_x, edge2_y = v2_y - v0_y, edge2_z = v2_z - v0_z;

  float h_x = rd_y * edge2_z - rd_z * edge2_y;
  float h_y = rd_z * edge2_x - rd_x * edge2_z;
  float h_z = rd_x * edge2_y - rd_y * edge2_x;

  float a = edge1_x * h_x + edge1_y * h_y + edge1_z * h_z;
  if (fabsf(a) < EPSILON) return ORC_LARGE_FLOAT;

  float f = 1 / a;
  float s_x = ro_x - v0_x, s_y = ro_y - v0_y, s_z = ro_z - v0_z;

  float w1 = f * (s_x * h_x + s_y * h_y + s_z * h_z);
  if (w1 < 0 || w1 > 1) return ORC_LARGE_FLOAT;

  float q_x = s_y * edge1_z - s_z * edge1_y;
  float q_y = s_z * edge1_x - s_x * edge1_z;
  float q_z = s_x * edge1_y - s_y * edge1_x;

  const float w2 = f * (rd_x * q_x + rd_y * q_y + rd_z * q_z);
  if (w2 < 0 || w1 + w2 > 1) return ORC_LARGE_FLOAT;

  const float tf = f * (edge2_x * q_x + edge2_y * q_y + edge2_z * q_z);
  if (tf <= EPSILON) return ORC_LARGE_FLOAT;

  *bx = w1;
  *by = w2;
  *bz = 1 - w1 - w2;
  return tf;
}

/* ---------------------------------------------------------------------------------------------
 * rt_traversal.cpp:231-261  BVHTraverser::ray_transform (rows 0-2 of invTransform)
 * ------------------------------------------------------------------------------------------- */
void orc_ray_transform(const float ray[6], const float m[12], float out[6]) {
  float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
  float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
  float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
  out[0] = m00 * ray[0] + m01 * ray[1] + m02 * ray[2] + m03;
  out[1] = m10 * ray[0] + m11 * ray[1] + m12 * ray[2] + m13;
  out[2] = m20 * ray[0] + m21 * ray[1] + m22 * ray[2] + m23;
  out[3] = m00 * ray[3] + m01 * ray[4] + m02 * ray[5];
  out[4] = m10 * ray[3] + m11 * ray[4] + m12 * ray[5];
  out[5] = m20 * ray[3] + m21 * ray[4] + m22 * ray[5];
}

/* rt_traversal.cpp:61-67  child box = origin + ldexp(float(q), e) */
void orc_child_box(const orc_node_t* n, int k, float box[6]) {
  const uint8_t* q = n->children[k].qaabb;
  box[0] = n->px + ldexpf((float)q[0], n->ex);
  box[1] = n->py + ldexpf((float)q[1], n->ey);
  box[2] = n->pz + ldexpf((float)q[2], n->ez);
  box[3] = n->px + ldexpf((float)q[3], n->ex);
  box[4] = n->py + ldexpf((float)q[4], n->ey);
  box[5] = n->pz + ldexpf((float)q[5], n->ez);
}

/* rt_traversal.cpp:219-225 */
static inline int is_top(const orc_node_t* n) { return (uint32_t)n->imask == 1; }
static inline int is_leaf(const orc_node_t* n) {
  return (is_top(n) && n->leafData != UINT32_MAX) || (!is_top(n) && n->leafData != 0);
}

typedef struct { float dist; uint32_t child; } child_isect_t;

/* std::sort(..., a.dist > b.dist) on <= 4 elements == libstdc++ __insertion_sort: stable,
 * farthest first (rt_traversal.cpp:76-78). */
static void sort_far_to_near(child_isect_t* a, int n) {
  for (int i = 1; i < n; ++i) {
    child_isect_t v = a[i];
    int j = i;
    while (j > 0 && v.dist > a[j - 1].dist) { a[j] = a[j - 1]; --j; }
    a[j] = v;
  }
}

/* =============================================================================================
 * Faithful traversal
 * =========================================================================================== */

/* sim/simx/types.h:1808-1840  ShortStack<TraversalStackEntry, 5> */
typedef struct { uint32_t node_ptr; uint8_t last; } sentry_t;
typedef struct { uint32_t head, bottom, count; sentry_t s[ORC_STACK_CAPACITY]; } sstack_t;

static void ss_push(sstack_t* st, sentry_t e) {
  if (st->count == ORC_STACK_CAPACITY) st->bottom = (st->bottom + 1) % ORC_STACK_CAPACITY;
  else st->count++;
  st->s[st->head] = e;
  st->head = (st->head + 1) % ORC_STACK_CAPACITY;
}
static sentry_t ss_pop(sstack_t* st) {
  sentry_t z = {0, 0};
  if (st->count == 0) return z;
  st->head = (st->head == 0) ? (ORC_STACK_CAPACITY - 1) : (st->head - 1);
  st->count--;
  return st->s[st->head];
}

typedef struct {
  const uint8_t* image; uint64_t image_size;
  uint32_t tlas_ptr, blas_ptr, qbvh_ptr, tri_ptr;
  orc_stats_t* st;
} fctx_t;

static void f_read(const fctx_t* c, void* dst, uint64_t addr, uint32_t size) {
  if (addr + size > c->image_size) { memset(dst, 0, size); c->st->oob++; return; }
  memcpy(dst, c->image + addr, size);
}

/* rt_traversal.cpp:171-213  findNextParentLevel + pop */
static int f_pop(const fctx_t* c, uint32_t* base_ptr, uint32_t* node_ptr, uint32_t* level,
                 uint32_t* trail, sstack_t* stack) {
  int32_t parent = -1;
  for (int i = (int)*level - 1; i >= 0; --i) {
    if (i < ORC_MAX_TRAIL_LEVEL && trail[i] != 4) { parent = i; break; }
  }
  if (parent < 0) return 1;
  trail[parent]++;
  for (int i = parent + 1; i < ORC_MAX_TRAIL_LEVEL; ++i) trail[i] = 0;
  if (stack->count == 0) {
    *base_ptr = c->tlas_ptr;
    *node_ptr = c->tlas_ptr;
    *level = 0;
    c->st->restarts++;
  } else {
    sentry_t e = ss_pop(stack);
    *node_ptr = e.node_ptr;
    if (e.last) trail[parent] = 4;
    *level = (uint32_t)parent + 1;
  }
  return 0;
}

/* rt_traversal.cpp:26-168  BVHTraverser::traverse; returns 1 when traversal completed, 0 when a
 * candidate is pending (any-hit shader must commit). */
static int f_traverse(const fctx_t* c, const float ray[6], orc_hit_t* hit, float* pending_dist,
                      uint32_t* trail, sstack_t* stack) {
  uint32_t level = 0;
  uint32_t base_ptr = c->tlas_ptr;
  uint32_t node_ptr = base_ptr;
  uint32_t blasIdx = 0;
  float cur_ray[6];
  memcpy(cur_ray, ray, sizeof cur_ray);
  orc_node_t node;
  int exit_ = 0;

  while (!exit_) {
    f_read(c, &node, node_ptr, sizeof(orc_node_t));
    c->st->node_reads++;

    if (!is_leaf(&node)) {
      child_isect_t isect[4];
      int cnt = 0;
      for (int i = 0; i < 4; ++i) {
        if (node.children[i].meta == 0) continue;
        float box[6];
        orc_child_box(&node, i, box);
        float d = orc_ray_box(is_top(&node) ? ray : cur_ray, box[0], box[1], box[2], box[3], box[4], box[5]);
        if (d < hit->dist) { isect[cnt].dist = d; isect[cnt].child = (uint32_t)i; cnt++; }
      }
      sort_far_to_near(isect, cnt);

      if (level >= ORC_MAX_TRAIL_LEVEL) { c->st->trail_overflow++; return 1; } /* reference: UB */
      uint32_t k = trail[level];
      uint32_t dropCount = (k == 4) ? (uint32_t)cnt - 1u : k; /* wraps when cnt==0 (:81) */
      /* :82-86 pops while size>0; equivalent closed form (avoids the 2^32-iteration spin) */
      if (dropCount >= (uint32_t)cnt) cnt = 0; else cnt -= (int)dropCount;

      if (is_top(&node) && base_ptr != c->tlas_ptr && cnt > 0) c->st->stale_base++;

      if (cnt == 0) {
        exit_ = f_pop(c, &base_ptr, &node_ptr, &level, trail, stack);
      } else {
        child_isect_t closest = isect[cnt - 1];
        cnt--;
        uint32_t nodeIdx = node.leftFirst + closest.child;
        node_ptr = base_ptr + nodeIdx * (uint32_t)sizeof(orc_node_t);
        if (cnt == 0) {
          trail[level] = 4;
        } else {
          for (int it = 0; it < cnt; ++it) {
            sentry_t e;
            e.node_ptr = base_ptr + (node.leftFirst + isect[it].child) * (uint32_t)sizeof(orc_node_t);
            e.last = (it == 0);
            ss_push(stack, e);
          }
        }
        level++;
      }
    } else if (is_top(&node)) {
      blasIdx = node.leafData;
      uint32_t blas_node_ptr = c->blas_ptr + blasIdx * 160u; /* :112 */
      struct { uint32_t bvh_offset; float inv[12]; } bn;
      f_read(c, &bn, blas_node_ptr, 52);
      c->st->inst_reads++;
      orc_ray_transform(ray, bn.inv, cur_ray);
      base_ptr = c->qbvh_ptr + bn.bvh_offset * (uint32_t)sizeof(orc_node_t);
      node_ptr = base_ptr;
    } else {
      uint32_t triCount = node.leafData;
      uint32_t leftFirst = node.leftFirst;
      for (uint32_t i = 0; i < triCount; ++i) {
        uint32_t triIdx = leftFirst + i;
        uint32_t tri_addr = c->tri_ptr + triIdx * (uint32_t)sizeof(orc_tri_t);
        orc_tri_t tri;
        f_read(c, &tri, tri_addr, sizeof tri);
        c->st->tri_reads++;
        float bx = 0, by = 0, bz = 0;
        float d = orc_ray_tri(cur_ray, &tri, &bx, &by, &bz);
        if (d < hit->dist) {
          *pending_dist = d;
          hit->bx = bx; hit->by = by; hit->bz = bz;
          hit->blasIdx = blasIdx;
          hit->triIdx = triIdx;
          while (stack->count) ss_pop(stack); /* :150-153 */
          return 0;
        }
      }
      exit_ = f_pop(c, &base_ptr, &node_ptr, &level, trail, stack);
    }
  }
  return 1;
}

int orc_trace_faithful(const uint8_t* image, uint64_t image_size,
                       uint32_t tlas_off, uint32_t blas_off, uint32_t bvh_off, uint32_t tri_off,
                       const float* rays, uint64_t n, const float* tmax, orc_hit_t* out,
                       orc_stats_t* stats, int any_hit_first) {
  orc_stats_t st;
  memset(&st, 0, sizeof st);
  fctx_t c = {image, image_size, tlas_off, blas_off, bvh_off, tri_off, &st};
  for (uint64_t i = 0; i < n; ++i) {
    /* rt_unit.cpp:50-60 init_ray: Hit() -> dist = LARGE_FLOAT, trail = {}, empty stack */
    orc_hit_t hit = {tmax ? tmax[i] : ORC_LARGE_FLOAT, 0, 0, 0, 0, 0}; /* tmax: extension, NULL = reference */
    float pending = 0;
    int accepted = 0;
    uint32_t trail[ORC_MAX_TRAIL_LEVEL];
    memset(trail, 0, sizeof trail);
    sstack_t stack;
    memset(&stack, 0, sizeof stack);
    for (;;) {
      int done = f_traverse(&c, rays + 6 * i, &hit, &pending, trail, &stack);
      if (done) break;
      hit.dist = pending; /* rt_unit.cpp:199-202 COMMIT_ACCEPT */
      st.accepts++;
      accepted = 1;
      if (any_hit_first) break;
    }
    if (!accepted) hit.dist = ORC_LARGE_FLOAT;
    out[i] = hit;
  }
  if (stats) *stats = st;
  return 0;
}

/* =============================================================================================
 * Canonical traversal (what the HIP kernels implement; equivalence argument in DESIGN.md s3)
 *   - one pass, full LIFO stack of (node, m) with m = max(entry distance of the node, entry
 *     distances of all its ancestors on the path);
 *   - a popped entry is visited only if m < hit.dist  (== the reference re-filtering every level of
 *     the re-descent with the shrunken hit.dist);
 *   - after an accept inside a leaf, the rest of the leaf is skipped unless path_m < hit.dist;
 *   - children: stable sort far->near, nearest visited first, ties -> higher child index first;
 *   - triangles in a leaf in index order, strict '<'.
 * =========================================================================================== */
typedef struct { uint32_t node; float m; } centry_t; /* node: bit31 = TLAS node */

int orc_trace_canonical(const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                        const orc_tri_t* tri, const float* rays, uint64_t n, const float* tmax,
                        orc_hit_t* out, orc_stats_t* stats, int any_hit_first) {
  orc_stats_t st;
  memset(&st, 0, sizeof st);
  size_t cap = 256;
  centry_t* stack = (centry_t*)malloc(cap * sizeof(centry_t));
  for (uint64_t r = 0; r < n; ++r) {
    const float* ray = rays + 6 * r;
    orc_hit_t hit = {tmax ? tmax[r] : ORC_LARGE_FLOAT, 0, 0, 0, 0, 0};
    int found = 0;
    float cur_ray[6];
    memcpy(cur_ray, ray, sizeof cur_ray);
    uint32_t blasIdx = 0;
    const orc_node_t* bbase = bvh; /* BLAS node base of the instance being traversed */
    size_t sp = 0;
    uint32_t cur = 0x80000000u; /* TLAS root */
    float path_m = -INFINITY;   /* root is never box-tested */
    int have = 1;
    while (have) {
      const orc_node_t* node = (cur & 0x80000000u) ? &tlas[cur & 0x7fffffffu] : &bbase[cur];
      st.node_reads++;
      int descend = 0;
      if (!is_leaf(node)) {
        child_isect_t isect[4];
        int cnt = 0;
        for (int i = 0; i < 4; ++i) {
          if (node->children[i].meta == 0) continue;
          float box[6];
          orc_child_box(node, i, box);
          float d = orc_ray_box(is_top(node) ? ray : cur_ray, box[0], box[1], box[2], box[3], box[4], box[5]);
          if (d < hit.dist) { isect[cnt].dist = d; isect[cnt].child = (uint32_t)i; cnt++; }
        }
        sort_far_to_near(isect, cnt);
        if (cnt > 0) {
          uint32_t tl = cur & 0x80000000u;
          if (sp + 4 > cap) { cap *= 2; stack = (centry_t*)realloc(stack, cap * sizeof(centry_t)); }
          for (int it = 0; it < cnt - 1; ++it) { /* far first, so the nearest pending is on top */
            stack[sp].node = tl | (node->leftFirst + isect[it].child);
            stack[sp].m = std_max(path_m, isect[it].dist);
            sp++;
          }
          if (sp > st.max_stack) st.max_stack = sp;
          cur = tl | (node->leftFirst + isect[cnt - 1].child);
          path_m = std_max(path_m, isect[cnt - 1].dist);
          descend = 1;
        }
      } else if (is_top(node)) {
        blasIdx = node->leafData;
        const orc_blas_t* b = &blas[blasIdx];
        st.inst_reads++;
        orc_ray_transform(ray, b->invTransform, cur_ray);
        bbase = bvh + b->bvh_offset;
        cur = 0; /* BLAS root, relative to bbase; path_m unchanged (same level) */
        descend = 1;
      } else {
        uint32_t triCount = node->leafData, leftFirst = node->leftFirst;
        for (uint32_t i = 0; i < triCount; ++i) {
          uint32_t triIdx = leftFirst + i;
          st.tri_reads++;
          float bx = 0, by = 0, bz = 0;
          float d = orc_ray_tri(cur_ray, &tri[triIdx], &bx, &by, &bz);
          if (d < hit.dist) {
            hit.dist = d; hit.bx = bx; hit.by = by; hit.bz = bz;
            hit.blasIdx = blasIdx; hit.triIdx = triIdx;
            found = 1;
            st.accepts++;
            if (any_hit_first) { sp = 0; break; }
            if (!(path_m < hit.dist)) { st.abandon++; break; }
          }
        }
        if (any_hit_first && found) { have = 0; break; }
      }
      if (!descend) {
        have = 0;
        while (sp > 0) {
          centry_t e = stack[--sp];
          if (e.m < hit.dist) { cur = e.node; path_m = e.m; have = 1; break; }
        }
        /* A popped BLAS entry always belongs to the instance being traversed (DFS order); a popped
         * TLAS entry re-derives cur_ray/bbase at its leaf. */
      }
    }
    if (!found) { hit.dist = ORC_LARGE_FLOAT; }
    out[r] = hit;
  }
  free(stack);
  if (stats) *stats = st;
  return 0;
}

/* =============================================================================================
 * Ray generation / shading / packing
 * =========================================================================================== */
typedef struct { float x, y, z; } f3;
static inline f3 f3_make(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }   /* geometry.h:346 */
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :542 */
static inline f3 f3_mul(f3 a, f3 b) { return f3_make(a.x * b.x, a.y * b.y, a.z * b.z); }   /* :715 */
static inline f3 f3_scale(f3 a, float b) { return f3_make(a.x * b, a.y * b, a.z * b); }    /* :721-722 */
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* :888 */
static inline f3 f3_cross(f3 a, f3 b) {                                                     /* :952 */
  return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 f3_normalize(f3 v) {                                                       /* :180,913-916 */
  float invLen = 1.0f / sqrtf(f3_dot(v, v));
  return f3_scale(v, invLen);
}

/* kernel.cpp:28-39  GenerateRay: u,v are evaluated in double then rounded to f32 */
void orc_generate_ray(uint32_t x, uint32_t y, uint32_t w, uint32_t h, float out6[6]) {
  f3 pos = f3_make(0.0f, 100.0f, 0.0f);
  f3 front = f3_make(1.0f, 0.0f, 0.0f);
  float FOV = 1.0f;
  float u = (float)((x * 2.0 - w) / h);
  float v = (float)((y * 2.0 - h) / h);
  f3 right = f3_cross(front, f3_make(0.0f, 1.0f, 0.0f));
  f3 up = f3_cross(right, front);
  f3 dir = f3_normalize(f3_add(f3_add(f3_scale(right, u), f3_scale(up, v)), f3_scale(front, FOV)));
  out6[0] = pos.x; out6[1] = pos.y; out6[2] = pos.z;
  out6[3] = dir.x; out6[4] = dir.y; out6[5] = dir.z;
}

/* the camera rays of rows [y0, y1), ray (x, y) at 6 * (x + (y - y0) * w): orc_generate_ray per pixel (checker convenience: whole
 * frames of rays without one foreign call per pixel) */
void orc_camera_rays(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, float* out) {
  for (uint32_t y = y0; y < y1; ++y)
    for (uint32_t x = 0; x < w; ++x) orc_generate_ray(x, y, w, h, out + 6 * ((uint64_t)x + (uint64_t)(y - y0) * w));
}

/* common.h:156-162 */
static f3 rgb8_to_f3(uint32_t c) {
  float s = 1 / 256.0f;
  int r = (c >> 16) & 255, g = (c >> 8) & 255, b = c & 255;
  return f3_make(r * s, g * s, b * s);
}

/* x86-64 g++ lowers uint32_t(float) to cvttss2si r64 + truncation; this is that behaviour made
 * explicit (negative uv wraps instead of being UB: SURVEY.md a16). */
static inline uint32_t f2u_x86(float f) { return (uint32_t)(int64_t)f; }

/* rtx_shading.h:5-18 texSample */
static f3 tex_sample(float u, float v, const uint32_t* pixels, uint32_t width, uint32_t height) {
  uint32_t iu = f2u_x86(u * width);
  uint32_t iv = f2u_x86(v * height);
  iu %= width;
  iv %= height;
  return rgb8_to_f3(pixels[iu + iv * width]);
}

/* rtx_shading.h:55-67 diffuseLighting */
static f3 diffuse_lighting(f3 pixel, f3 normal, f3 diffuse_color, f3 ambient, f3 light_color, f3 light_pos) {
  f3 L = f3_sub(light_pos, pixel);
  float dist = sqrtf(f3_dot(L, L));
  L = f3_scale(L, 1.0f / dist);
  float att = 1.0f / (1.0f + dist * 0.1f);
  float NdotL = std_max(0.0f, f3_dot(normal, L));
  return f3_mul(diffuse_color, f3_add(ambient, f3_scale(f3_scale(light_color, att), NdotL)));
}

/* closest.cpp:57-90 for one hit: the non-reflected diffuse contribution `throughput * diffuse *
 * (1 - reflectivity)` with throughput = 1 (:87), the reflectivity (:84), the hit point I and the shading
 * normal N.  occluded = result of the shadow extension (0 = the reference: no occlusion query). */
static void shade_terms(const float ray6[6], const orc_hit_t* hit,
                        const orc_blas_t* blas_ptr, const orc_triex_t* triEx_ptr, const orc_material_t* mat_ptr,
                        const uint8_t* tex_ptr, const orc_shade_params_t* p, int occluded,
                        f3* out_term, float* out_refl, f3* out_I, f3* out_N, f3* out_albedo) {
  float throughput = 1.0f;
  f3 orig = f3_make(ray6[0], ray6[1], ray6[2]);
  f3 dir = f3_make(ray6[3], ray6[4], ray6[5]);
  const orc_blas_t* blas = &blas_ptr[hit->blasIdx];
  const orc_triex_t* te = &triEx_ptr[hit->triIdx];
  const orc_material_t* mat = &mat_ptr[te->texId];
  f3 I = f3_add(orig, f3_scale(dir, hit->dist));                                      /* :61 */
  f3 N0 = f3_make(te->N0[0], te->N0[1], te->N0[2]);
  f3 N1 = f3_make(te->N1[0], te->N1[1], te->N1[2]);
  f3 N2 = f3_make(te->N2[0], te->N2[1], te->N2[2]);
  f3 N = f3_add(f3_add(f3_scale(N1, hit->bx), f3_scale(N2, hit->by)), f3_scale(N0, hit->bz)); /* :64 */
  /* :65-66  invTransform.transposed() copies the 3x3 block into an identity (geometry.h:1141-1147);
   * TransformVector = float4(N,0) * M (geometry.h:1280-1293): ((c0*x + c1*y) + c2*z) + c3*0 */
  const float* m = blas->invTransform;
  f3 Nt;
  Nt.x = m[0] * N.x + m[4] * N.y + m[8] * N.z + 0.0f * 0.0f;
  Nt.y = m[1] * N.x + m[5] * N.y + m[9] * N.z + 0.0f * 0.0f;
  Nt.z = m[2] * N.x + m[6] * N.y + m[10] * N.z + 0.0f * 0.0f;
  N = f3_normalize(Nt);
  float uvx = te->uv1[0] * hit->bx + te->uv2[0] * hit->by + te->uv0[0] * hit->bz;     /* :69 */
  float uvy = te->uv1[1] * hit->bx + te->uv2[1] * hit->by + te->uv0[1] * hit->bz;
  f3 texColor;
  if (mat->diffuse_tex_id >= 0) {                                                      /* :72-77 */
    const uint32_t* px = (const uint32_t*)(tex_ptr + mat->tex_offset);
    texColor = tex_sample(uvx, uvy, px, mat->tex_width, mat->tex_height);
  } else {
    texColor = f3_make(mat->diffuse[0], mat->diffuse[1], mat->diffuse[2]);
  }
  f3 ambient = f3_make(p->ambient[0], p->ambient[1], p->ambient[2]);
  f3 light_color = f3_make(p->light_color[0], p->light_color[1], p->light_color[2]);
  f3 light_pos = f3_make(p->light_pos[0], p->light_pos[1], p->light_pos[2]);
  f3 diffuse;
  if (!occluded) {
    diffuse = diffuse_lighting(I, N, texColor, ambient, light_color, light_pos);
  } else { /* shadow extension: the same expression with NdotL forced to 0 */
    f3 L = f3_sub(light_pos, I);
    float dist = sqrtf(f3_dot(L, L));
    float att = 1.0f / (1.0f + dist * 0.1f);
    diffuse = f3_mul(texColor, f3_add(ambient, f3_scale(f3_scale(light_color, att), 0.0f)));
  }
  float reflectivity = blas->reflectivity;                                             /* :84 */
  *out_term = f3_add(f3_make(0, 0, 0), f3_scale(f3_scale(diffuse, throughput), 1 - reflectivity)); /* :87 */
  *out_refl = reflectivity;
  *out_I = I; *out_N = N;
  if (out_albedo) *out_albedo = texColor;
}

void orc_shade(const float ray6[6], const orc_hit_t* hit,
               const orc_blas_t* blas_ptr, const orc_triex_t* triEx_ptr, const orc_material_t* mat_ptr,
               const uint8_t* tex_ptr, const orc_shade_params_t* p, float out_color[3]) {
  f3 background = f3_make(p->background[0], p->background[1], p->background[2]);
  if (hit->dist == ORC_LARGE_FLOAT) { /* rt_unit.cpp:107-109 -> miss.cpp:9-14 */
    out_color[0] = background.x; out_color[1] = background.y; out_color[2] = background.z;
    return;
  }
  f3 radiance, I, N;
  float throughput = 1.0f, reflectivity;
  shade_terms(ray6, hit, blas_ptr, triEx_ptr, mat_ptr, tex_ptr, p, 0, &radiance, &reflectivity, &I, &N, NULL);
  throughput *= reflectivity;                                                          /* :90 */
  /* :95-121: secondary ray only if reflectivity > 0 && bounce+1 < max_depth (orc_render_ex follows it);
   * this entry point is the else arm, which is all the shipped scene builder reaches (scene.cpp:96) */
  radiance = f3_add(radiance, f3_scale(background, throughput));                       /* :123 */
  out_color[0] = radiance.x; out_color[1] = radiance.y; out_color[2] = radiance.z;
}

/* Occlusion ray of the shadow extension (no reference counterpart; BASELINE "primary+shadow"): from the
 * hit point toward the light, origin pushed 1e-3 along L as the mirror bounce does (closest.cpp:104),
 * tmax = |L|; any accepted candidate occludes. */
static int occluded_toward_light(const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh, const orc_tri_t* tri,
                                 const float ray6[6], float hit_dist, const orc_shade_params_t* p, uint64_t* n_rays) {
  f3 orig = f3_make(ray6[0], ray6[1], ray6[2]), dir = f3_make(ray6[3], ray6[4], ray6[5]);
  f3 I = f3_add(orig, f3_scale(dir, hit_dist));
  f3 L = f3_sub(f3_make(p->light_pos[0], p->light_pos[1], p->light_pos[2]), I);
  float dist = sqrtf(f3_dot(L, L));
  L = f3_scale(L, 1.0f / dist);
  f3 so = f3_add(I, f3_scale(L, 0.001f));
  float sray[6] = {so.x, so.y, so.z, L.x, L.y, L.z};
  orc_hit_t sh;
  orc_trace_canonical(tlas, blas, bvh, tri, sray, 1, &dist, &sh, NULL, 1);
  if (n_rays) ++*n_rays;
  return sh.dist != ORC_LARGE_FLOAT;
}

/* Radiance carried by one ray: rt_unit.cpp:98-116 (closest hit or miss) -> closest.cpp:11-127 with the
 * mirror bounce of :95-121 followed recursively (secPayload.bounce = bounce + 1), miss.cpp:9-14. */
static f3 radiance_of(const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh, const orc_tri_t* tri,
                      const orc_triex_t* triEx, const orc_material_t* mat, const uint8_t* tex,
                      const orc_shade_params_t* p, int shadow, const float ray6[6], uint32_t bounce,
                      orc_hit_t* out_hit, uint64_t* n_rays) {
  f3 background = f3_make(p->background[0], p->background[1], p->background[2]);
  orc_hit_t hit;
  orc_trace_canonical(tlas, blas, bvh, tri, ray6, 1, NULL, &hit, NULL, 0);
  if (n_rays) ++*n_rays;
  if (out_hit) *out_hit = hit;
  if (hit.dist == ORC_LARGE_FLOAT) return background;
  int occ = shadow ? occluded_toward_light(tlas, blas, bvh, tri, ray6, hit.dist, p, n_rays) : 0;
  f3 radiance, I, N;
  float throughput = 1.0f, reflectivity;
  shade_terms(ray6, &hit, blas, triEx, mat, tex, p, occ, &radiance, &reflectivity, &I, &N, NULL);
  throughput *= reflectivity;                                                          /* :90 */
  if (reflectivity > 0.0f && bounce + 1 < p->max_depth) {                              /* :95 */
    f3 dir = f3_make(ray6[3], ray6[4], ray6[5]);
    /* :96  R = normalize(ray.dir - 2.0f * N * dot(N, ray.dir)):  (2*N) * dot, geometry.h:722,721 */
    f3 twoN = f3_make(2.0f * N.x, 2.0f * N.y, 2.0f * N.z);
    f3 R = f3_normalize(f3_sub(dir, f3_scale(twoN, f3_dot(N, dir))));
    f3 so = f3_add(I, f3_scale(R, 0.001f));                                            /* :99 */
    float sec[6] = {so.x, so.y, so.z, R.x, R.y, R.z};
    f3 sc = radiance_of(tlas, blas, bvh, tri, triEx, mat, tex, p, shadow, sec, bounce + 1, NULL, n_rays);
    radiance = f3_add(radiance, f3_scale(sc, throughput));                             /* :117 */
  } else {
    radiance = f3_add(radiance, f3_scale(background, throughput));                     /* :123 */
  }
  return radiance;
}

/* common.h:149-154 RGB32FtoRGB8 */
uint32_t orc_pack_rgb8(const float c[3]) {
  int r = (int)(std_min(c[0], 1.f) * 255);
  int g = (int)(std_min(c[1], 1.f) * 255);
  int b = (int)(std_min(c[2], 1.f) * 255);
  return (uint32_t)((r << 16) + (g << 8) + b);
}

int orc_render(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
               const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
               const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
               const uint8_t* tex, const orc_shade_params_t* p,
               uint32_t* out_pixels, orc_hit_t* out_hits, float* out_color) {
  for (uint32_t y = y0; y < y1; ++y) {
    for (uint32_t x = 0; x < w; ++x) {
      float ray[6];
      orc_generate_ray(x, y, w, h, ray);
      orc_hit_t hit;
      orc_trace_canonical(tlas, blas, bvh, tri, ray, 1, NULL, &hit, NULL, 0);
      float col[3];
      orc_shade(ray, &hit, blas, triEx, mat, tex, p, col);
      uint64_t idx = (uint64_t)x + (uint64_t)y * w;
      out_pixels[idx] = orc_pack_rgb8(col);
      if (out_hits) out_hits[idx] = hit;
      if (out_color) { out_color[3 * idx] = col[0]; out_color[3 * idx + 1] = col[1]; out_color[3 * idx + 2] = col[2]; }
    }
  }
  return 0;
}

/* orc_render with the mirror bounce followed (max_depth from the parameters) and, optionally, the
 * shadow extension at every shaded hit.  n_rays: rays traced (camera + bounce + occlusion). */
int orc_render_ex(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                  const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                  const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
                  const uint8_t* tex, const orc_shade_params_t* p, int shadow,
                  uint32_t* out_pixels, orc_hit_t* out_hits, float* out_color, uint64_t* n_rays) {
  if (n_rays) *n_rays = 0;
  for (uint32_t y = y0; y < y1; ++y) {
    for (uint32_t x = 0; x < w; ++x) {
      float ray[6];
      orc_generate_ray(x, y, w, h, ray);
      orc_hit_t hit;
      f3 c = radiance_of(tlas, blas, bvh, tri, triEx, mat, tex, p, shadow, ray, 0, &hit, n_rays);
      float col[3] = {c.x, c.y, c.z};
      uint64_t idx = (uint64_t)x + (uint64_t)y * w;
      out_pixels[idx] = orc_pack_rgb8(col);
      if (out_hits) out_hits[idx] = hit;
      if (out_color) { out_color[3 * idx] = col[0]; out_color[3 * idx + 1] = col[1]; out_color[3 * idx + 2] = col[2]; }
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Ambient occlusion (extension; BASELINE config 5 asks for "16 spp Monte-Carlo AO", the reference has
 * no such pass).  Only the RNG is the reference's (common.h:129-147: WangHash, Marsaglia xorshift32,
 * RandomFloat).  Everything else is defined here and mirrored operation by operation by the HIP kernel
 * (rt_ao_rays_kernel), using only IEEE add/mul/div/sqrt so that both sides produce the same rays:
 *   seed  = WangHash((x + y*W) * spp + s + 1 + user_seed * 0x9E3779B9), 0 -> 1
 *   (u,v) = first of up to 8 draws of (2*RandomFloat-1, 2*RandomFloat-1) with u*u + v*v < 1, else (0,0)
 *   z     = sqrt(1 - (u*u + v*v))                      cosine-weighted about the facing normal
 *   N'    = shading normal flipped to face the viewer; basis of Duff et al. 2017 (no trigonometry)
 *   ray   = (I + N'*1e-3, T*u + B*v + N'*z), tmax = radius; any accepted candidate occludes
 *   pixel = Lambert colour of the primary hit (closest.cpp else arm) * (unoccluded / spp)
 * ------------------------------------------------------------------------------------------- */
static uint32_t wang_hash(uint32_t s) {            /* common.h:129-135 */
  s = (s ^ 61) ^ (s >> 16);
  s *= 9, s = s ^ (s >> 4);
  s *= 0x27d4eb2d;
  s = s ^ (s >> 15);
  return s;
}
static uint32_t random_int(uint32_t* s) {          /* common.h:137-143 */
  *s ^= *s << 13;
  *s ^= *s >> 17;
  *s ^= *s << 5;
  return *s;
}
static float random_float(uint32_t* s) { return random_int(s) * 2.3283064365387e-10f; }   /* common.h:145-147 */

/* the three RNG helpers as the AO / bounce passes use them, exposed so that they can be pinned to the reference's own
 * (oracle/_ref: vxref_rng) and to tests/golden/rng.npz: same layout as vxref_rng */
void orc_rng(uint32_t seed, uint32_t n, uint32_t* hash, uint32_t* ints, float* floats) {
  for (uint32_t i = 0; i < n; ++i) hash[i] = wang_hash(seed + i);
  uint32_t s = wang_hash(seed);
  if (s == 0) s = 1;
  uint32_t t = s;
  for (uint32_t i = 0; i < n; ++i) ints[i] = random_int(&s);
  for (uint32_t i = 0; i < n; ++i) floats[i] = random_float(&t);
}

void orc_ao_ray(uint32_t x, uint32_t y, uint32_t w, uint32_t spp, uint32_t s, uint32_t user_seed,
                const float I[3], const float N[3], const float view_dir[3], float out6[6]) {
  uint32_t seed = wang_hash((x + y * w) * spp + s + 1u + user_seed * 0x9E3779B9u);
  if (seed == 0) seed = 1;
  float u = 0.0f, v = 0.0f, r2 = 0.0f;
  int ok = 0;
  for (int i = 0; i < 8 && !ok; ++i) {
    float a = 2.0f * random_float(&seed) - 1.0f;
    float b = 2.0f * random_float(&seed) - 1.0f;
    float q = a * a + b * b;
    if (q < 1.0f) { u = a; v = b; r2 = q; ok = 1; }
  }
  float z = sqrtf(1.0f - r2);
  float nx = N[0], ny = N[1], nz = N[2];
  if (nx * view_dir[0] + ny * view_dir[1] + nz * view_dir[2] > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
  float sign = nz >= 0.0f ? 1.0f : -1.0f;
  float a = -1.0f / (sign + nz);
  float b = nx * ny * a;
  float tx = 1.0f + sign * nx * nx * a, ty = sign * b, tz = -sign * nx;
  float bx = b, by = sign + ny * ny * a, bz = -ny;
  out6[0] = I[0] + nx * 0.001f; out6[1] = I[1] + ny * 0.001f; out6[2] = I[2] + nz * 0.001f;
  out6[3] = tx * u + bx * v + nx * z;
  out6[4] = ty * u + by * v + ny * z;
  out6[5] = tz * u + bz * v + nz * z;
}

int orc_render_ao(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                  const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                  const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
                  const uint8_t* tex, const orc_shade_params_t* p, uint32_t spp, float radius, uint32_t user_seed,
                  uint32_t* out_pixels, float* out_color, uint32_t* out_unoccluded, uint64_t* n_rays) {
  if (n_rays) *n_rays = 0;
  for (uint32_t y = y0; y < y1; ++y) {
    for (uint32_t x = 0; x < w; ++x) {
      float ray[6];
      orc_generate_ray(x, y, w, h, ray);
      orc_hit_t hit;
      orc_trace_canonical(tlas, blas, bvh, tri, ray, 1, NULL, &hit, NULL, 0);
      if (n_rays) ++*n_rays;
      uint64_t idx = (uint64_t)x + (uint64_t)y * w;
      float col[3];
      uint32_t open = 0;
      if (hit.dist == ORC_LARGE_FLOAT) {
        orc_shade(ray, &hit, blas, triEx, mat, tex, p, col);   /* background, no AO */
      } else {
        f3 term, I, N;
        float refl;
        shade_terms(ray, &hit, blas, triEx, mat, tex, p, 0, &term, &refl, &I, &N, NULL);
        orc_shade(ray, &hit, blas, triEx, mat, tex, p, col);
        float If[3] = {I.x, I.y, I.z}, Nf[3] = {N.x, N.y, N.z};
        for (uint32_t s = 0; s < spp; ++s) {
          float ao[6];
          orc_ao_ray(x, y, w, spp, s, user_seed, If, Nf, ray + 3, ao);
          orc_hit_t oh;
          orc_trace_canonical(tlas, blas, bvh, tri, ao, 1, &radius, &oh, NULL, 1);
          if (n_rays) ++*n_rays;
          if (oh.dist == ORC_LARGE_FLOAT) ++open;
        }
        float f = (float)open / (float)spp;
        col[0] *= f; col[1] *= f; col[2] *= f;
      }
      out_pixels[idx] = orc_pack_rgb8(col);
      if (out_color) { out_color[3 * idx] = col[0]; out_color[3 * idx + 1] = col[1]; out_color[3 * idx + 2] = col[2]; }
      if (out_unoccluded) out_unoccluded[idx] = open;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * One diffuse bounce (extension; BASELINE config 3 says "1 bounce diffuse", the reference has no such pass).
 * Per pixel with a primary hit: one cosine-weighted ray about the viewer-facing shading normal -- the AO
 * recipe above with spp = 1, sample 0 and no tmax -- traced for its closest hit;
 *   pixel = Lambert colour of the primary hit + albedo(primary) * (Lambert colour of the bounce hit | background)
 * where "Lambert colour" is closest.cpp's else arm (orc_shade) and albedo its texColor (:72-77).
 * ------------------------------------------------------------------------------------------- */
int orc_render_gi(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                  const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                  const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
                  const uint8_t* tex, const orc_shade_params_t* p, uint32_t user_seed,
                  uint32_t* out_pixels, float* out_color, uint64_t* n_rays) {
  if (n_rays) *n_rays = 0;
  for (uint32_t y = y0; y < y1; ++y) {
    for (uint32_t x = 0; x < w; ++x) {
      float ray[6], col[3];
      orc_generate_ray(x, y, w, h, ray);
      orc_hit_t hit;
      orc_trace_canonical(tlas, blas, bvh, tri, ray, 1, NULL, &hit, NULL, 0);
      if (n_rays) ++*n_rays;
      orc_shade(ray, &hit, blas, triEx, mat, tex, p, col);
      if (hit.dist != ORC_LARGE_FLOAT) {
        f3 term, I, N, albedo;
        float refl;
        shade_terms(ray, &hit, blas, triEx, mat, tex, p, 0, &term, &refl, &I, &N, &albedo);
        float If[3] = {I.x, I.y, I.z}, Nf[3] = {N.x, N.y, N.z}, b[6], c1[3];
        orc_ao_ray(x, y, w, 1, 0, user_seed, If, Nf, ray + 3, b);
        orc_hit_t bh;
        orc_trace_canonical(tlas, blas, bvh, tri, b, 1, NULL, &bh, NULL, 0);
        if (n_rays) ++*n_rays;
        orc_shade(b, &bh, blas, triEx, mat, tex, p, c1);
        col[0] = col[0] + albedo.x * c1[0]; col[1] = col[1] + albedo.y * c1[1]; col[2] = col[2] + albedo.z * c1[2];
      }
      uint64_t idx = (uint64_t)x + (uint64_t)y * w;
      out_pixels[idx] = orc_pack_rgb8(col);
      if (out_color) { out_color[3 * idx] = col[0]; out_color[3 * idx + 1] = col[1]; out_color[3 * idx + 2] = col[2]; }
    }
  }
  return 0;
}
