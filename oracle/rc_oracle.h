/* TEST INFRASTRUCTURE.  CPU restatement of the reference's software ray caster
 * (tests/regression/raycast: render.h, kernel.cpp, geometry.h:1416-1468), the "software twin" of the RTU
 * test.  Plain C99, scalar f32, -O2 -ffp-contract=off.  Pinned against the reference's own
 * object code (oracle/_ref/libvxref_rc.so, its `-c` CPU path) and the fixtures tests/golden/rc_*.npz.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it. */
#ifndef RC_ORACLE_H
#define RC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#pragma pack(push, 1)
typedef struct { float aabbMin[3]; uint32_t leftFirst; float aabbMax[3]; uint32_t triCount; } rc_bvh_node_t;    /* common.h:32-45, 32 B */
typedef struct { float aabbMin[3]; uint32_t leftRight; float aabbMax[3]; uint32_t blasIdx; } rc_tlas_node_t;   /* common.h:63-83, 32 B */
typedef struct {                                                                                                  /* common.h:48-60, 160 B */
  float transform[16]; float invTransform[16];
  uint32_t bvh_offset; uint32_t _pad0; uint64_t tex_offset; uint32_t tex_width, tex_height; float reflectivity; uint32_t _pad1;
} rc_blas_t;
typedef struct { float v0[3], v1[3], v2[3]; } rc_tri_t;                                                          /* geometry.h:1401-1405 */
typedef struct { float N0[3], N1[3], N2[3]; float uv0[2], uv1[2], uv2[2]; } rc_triex_t;                          /* common.h:19-22, 60 B */
typedef struct { float dist, bx, by, bz; uint32_t blasIdx, triIdx; } rc_hit_t;                                   /* common.h:24-29 */
#pragma pack(pop)

typedef struct {   /* the launch parameters of common.h:126-150 that the kernel reads, pointers resolved */
  uint32_t dst_width, dst_height;
  const rc_tri_t* tri; const rc_triex_t* triEx; const uint32_t* triIdx; const uint8_t* tex;
  const rc_bvh_node_t* bvh; const rc_blas_t* blas; const rc_tlas_node_t* tlas; uint32_t tlas_root;
  float camera_pos[3], camera_forward[3], camera_right[3], camera_up[3], viewplane[2];
  uint32_t samples_per_pixel, max_depth;
  float light_pos[3], light_color[3], ambient_color[3], background_color[3];
} rc_args_t;

/* render.h:143-190 (TLASIntersect and below); returns 0, or -1 if a traversal stack would exceed BVH_STACK_SIZE (UB in the reference) */
int rc_trace(const rc_args_t* a, const float ray6[6], rc_hit_t* hit);
void rc_generate_ray(const rc_args_t* a, uint32_t x, uint32_t y, float out6[6]);   /* render.h:192-211 */
int rc_radiance(const rc_args_t* a, const float ray6[6], float out3[3]);            /* render.h:213-275 Trace */
/* kernel.cpp:9-33 / tracer.cpp:249-263: rows [y0,y1) */
int rc_render(const rc_args_t* a, uint32_t y0, uint32_t y1, uint32_t* out_pixels, float* out_color /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
