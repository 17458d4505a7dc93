/*
 * oracle/rt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C99, scalar f32, compile with -O2 -ffp-contract=off, no
 * -march=native) of the reference's ray-tracing hot path:
 *   sim/simx/rt_traversal.cpp:26-339, sim/simx/rt_unit.cpp:98-116,190-213,
 *   sim/simx/types.h:1808-1840, tests/regression/raytracing/kernel.cpp:28-39,95-106,
 *   shaders/closest.cpp:57-129, shaders/miss.cpp:9-14, rtx_shading.h:5-18,55-67,
 *   common.h:149-162, geometry.h:887-915,952,1141-1147,1273-1293.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (vortex-raytracing_amd/) never links, imports or calls it.
 *
 * Parity pin: tests/test_oracle_vs_ref.py checks every function below bit-for-bit
 * against oracle/_ref/libvxref.so (the reference's own rt_traversal.cpp / scene builder /
 * shading helpers compiled from /root/reference) and tests/test_oracle_golden.py checks it
 * against the committed fixtures in tests/golden/ that were produced by that library.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LARGE_FLOAT 1e30f
#define ORC_MAX_TRAIL_LEVEL 32
#define ORC_STACK_CAPACITY 5

#pragma pack(push, 1)
typedef struct { uint8_t meta; uint8_t qaabb[6]; } orc_child_t;
#pragma pack(pop)

/* bvh_quantized_node_t (raytracing/common.h:52-67) == sim BVHNode (rt_traversal.h:14-33): 52 B */
typedef struct {
  float px, py, pz;
  int8_t ex, ey, ez;
  uint8_t imask;
  uint32_t leftFirst;
  uint32_t leafData;
  orc_child_t children[4];
} orc_node_t;

/* blas_node_t (common.h:86-99): 160 B stride; the traverser reads the first 52 B */
typedef struct {
  uint32_t bvh_offset;
  float invTransform[16];
  float transform[16];
  uint64_t mat_offset;
  uint32_t tex_width, tex_height;
  float reflectivity;
  uint32_t _pad;
} orc_blas_t;

typedef struct { float v0[3], v1[3], v2[3]; } orc_tri_t;                 /* 36 B */
typedef struct { float N0[3], N1[3], N2[3]; float uv0[2], uv1[2], uv2[2]; uint32_t texId; } orc_triex_t; /* 64 B */
typedef struct {
  float ambient[3], diffuse[3], specular[3], emissive[3];
  float shininess, ior, dissolve, reflectivity;
  int32_t diffuse_tex_id, illum;
  uint32_t tex_width, tex_height;
  uint64_t tex_offset;
} orc_material_t;                                                         /* 88 B */

typedef struct { float dist, bx, by, bz; uint32_t blasIdx, triIdx; } orc_hit_t; /* 24 B */

typedef struct {
  uint64_t node_reads, inst_reads, tri_reads; /* N_node, N_inst, N_tri */
  uint64_t accepts;
  uint64_t restarts;        /* short-stack restarts (faithful mode only) */
  uint64_t stale_base;      /* TLAS internal node expanded with a non-TLAS base_ptr (reference quirk) */
  uint64_t trail_overflow;  /* level >= 32: undefined behaviour in the reference */
  uint64_t abandon;         /* re-descent after accept dropped the current path */
  uint64_t oob;
  uint64_t max_stack;       /* canonical mode: deepest full-stack use */
} orc_stats_t;

/* --- arithmetic leaves (each restates one reference function) --- */
float orc_ray_box(const float ray[6], float min_x, float min_y, float min_z,
                  float max_x, float max_y, float max_z);                 /* rt_traversal.cpp:318-339 */
float orc_ray_tri(const float ray[6], const orc_tri_t* tri, float* bx, float* by, float* bz); /* :263-316 */
void  orc_ray_transform(const float ray[6], const float m[12], float out[6]); /* :231-261 */
void  orc_child_box(const orc_node_t* n, int k, float box[6]);            /* :61-67 */

/* --- traversal --- */
/* Faithful restatement: trail[32] + 5-entry ShortStack + restart, re-entered from the root after
 * every accepted candidate (rt_traversal.cpp:26-213 + rt_unit.cpp:98-116,199-202). `image` is the
 * flat device memory; offsets are the 32-bit DCR values 0x6..0x9 (tracer.cpp:252-256).
 * any_hit_first: stop after the first ACCEPT (occlusion query; extension). */
int orc_trace_faithful(const uint8_t* image, uint64_t image_size,
                       uint32_t tlas_off, uint32_t blas_off, uint32_t bvh_off, uint32_t tri_off,
                       const float* rays, uint64_t n, const float* tmax, orc_hit_t* out,
                       orc_stats_t* stats, int any_hit_first);

/* Canonical restatement = the algorithm the HIP kernels implement: one pass, full stack, entries
 * carry m = max(entry distances on the path); proven equal to the faithful one on every fixture.
 * Counts N_node/N_inst/N_tri with an unbounded stack (SURVEY.md s8d algorithmic bytes). tmax: only
 * candidates with d < tmax are considered (1e30 = reference behaviour). */
int orc_trace_canonical(const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                        const orc_tri_t* tri, const float* rays, uint64_t n, const float* tmax,
                        orc_hit_t* out, orc_stats_t* stats, int any_hit_first);

/* --- ray generation, shading, pixel packing --- */
void orc_generate_ray(uint32_t x, uint32_t y, uint32_t w, uint32_t h, float out6[6]); /* kernel.cpp:28-39 */
void orc_camera_rays(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, float* out);       /* the same, rows [y0, y1) */

typedef struct {
  float ambient[3], light_color[3], light_pos[3], background[3];
  uint32_t max_depth;
} orc_shade_params_t;

/* closest.cpp:57-127 (no secondary ray: reflectivity<=0 or bounce+1>=max_depth) / miss.cpp:9-14 */
void orc_shade(const float ray6[6], const orc_hit_t* hit,
               const orc_blas_t* blas, const orc_triex_t* triEx, const orc_material_t* mat,
               const uint8_t* tex, const orc_shade_params_t* p, float out_color[3]);
uint32_t orc_pack_rgb8(const float c[3]);                                  /* common.h:149-154 */

/* Whole frame of the RTU test (kernel.cpp:41-126 net effect: one closest-hit query + one shade per
 * pixel; samples_per_pixel re-traces the same ray and overwrites).  rows [y0,y1). */
int orc_render(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
               const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
               const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
               const uint8_t* tex, const orc_shade_params_t* p,
               uint32_t* out_pixels, orc_hit_t* out_hits /* may be NULL */, float* out_color /* may be NULL, 3/pixel */);

/* orc_render following the mirror bounce of closest.cpp:95-121 up to p->max_depth, optionally with the
 * shadow extension (one occlusion ray toward the light per shaded hit, at every depth).  The bounce arm
 * cannot be pinned against the reference (its shaders only compile for RISC-V and its scene builder
 * never sets a reflectivity): parity unpinned for that arm, restated from the source text. */
int orc_render_ex(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                  const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                  const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
                  const uint8_t* tex, const orc_shade_params_t* p, int shadow,
                  uint32_t* out_pixels, orc_hit_t* out_hits /* may be NULL */, float* out_color /* may be NULL */,
                  uint64_t* n_rays /* may be NULL */);

/* Ambient-occlusion pass (extension; definition in rt_oracle.c).  Only the RNG is the reference's
 * (common.h:129-147); the checker and the HIP kernel share the sampling recipe operation by operation. */
/* WangHash / RandomInt / RandomFloat of common.h:129-147 as the passes below use them: hash[i] = WangHash(seed + i), ints / floats =
 * two xorshift streams from WangHash(seed) (0 -> 1) */
void orc_rng(uint32_t seed, uint32_t n, uint32_t* hash, uint32_t* ints, float* floats);
void orc_ao_ray(uint32_t x, uint32_t y, uint32_t w, uint32_t spp, uint32_t s, uint32_t user_seed,
                const float I[3], const float N[3], const float view_dir[3], float out6[6]);
int orc_render_ao(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                  const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                  const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
                  const uint8_t* tex, const orc_shade_params_t* p, uint32_t spp, float radius, uint32_t user_seed,
                  uint32_t* out_pixels, float* out_color /* may be NULL */, uint32_t* out_unoccluded /* may be NULL */,
                  uint64_t* n_rays /* may be NULL */);

/* One cosine-weighted diffuse bounce per primary hit (extension; definition in rt_oracle.c). */
int orc_render_gi(uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                  const orc_node_t* tlas, const orc_blas_t* blas, const orc_node_t* bvh,
                  const orc_tri_t* tri, const orc_triex_t* triEx, const orc_material_t* mat,
                  const uint8_t* tex, const orc_shade_params_t* p, uint32_t user_seed,
                  uint32_t* out_pixels, float* out_color /* may be NULL */, uint64_t* n_rays /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
