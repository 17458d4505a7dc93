// Shared host/device plain-data layouts of the reference's scene buffers.
// Sizes are part of the boundary: the host app uploads these bytes as produced by the
// reference's scene builder (tests/regression/raytracing/common.h, geometry.h).
#pragma once
#include <stdint.h>

#define RT_LARGE_FLOAT 1e30f       // rt_traversal.h:7, geometry.h:15
#define RT_EPSILON 1e-6f           // rt_traversal.cpp:7
#define RT_BVH_WIDTH 4             // hw/VX_config.toml:246, raytracing/common.h:18
#define RT_NODE_BYTES 52           // sizeof(bvh_quantized_node_t)
#define RT_NODE_DWORDS 13
#define RT_BLAS_STRIDE 160         // rt_traversal.cpp:112 (hard-coded `* 160`)
#define RT_TRI_BYTES 36
#define RT_MAX_TRAIL 32            // TraversalTrail levels (rt_traversal.h: trail[32]); deeper = undefined behaviour in the reference
#define RT_TRIEX_BYTES 64
#define RT_MAT_BYTES 88
#define RT_MAX_LEVELS 32           // MAX_TRAIL_LEVEL (rt_traversal.h:8): deeper trees are UB in the reference
#define RT_STACK_ENTRIES (3 * RT_MAX_LEVELS)  // <=3 pending siblings per level

#pragma pack(push, 1)
struct rt_child_t { uint8_t meta; uint8_t qaabb[6]; };
#pragma pack(pop)

struct rt_qnode_t {            // common.h:52-67 / sim rt_traversal.h:14-33
  float origin[3];
  int8_t ex, ey, ez;
  uint8_t imask;               // 1 = TLAS node, 0 = BLAS node
  uint32_t leftFirst;          // first child index | first triangle index (already mesh-offset)
  uint32_t leafData;           // BLAS: triCount (0 = internal); TLAS: blasIdx (UINT32_MAX = internal)
  rt_child_t children[RT_BVH_WIDTH];
};
static_assert(sizeof(rt_qnode_t) == RT_NODE_BYTES, "node layout");

struct rt_blas_t {             // common.h:86-99
  uint32_t bvh_offset;
  float invTransform[16];
  float transform[16];
  uint64_t mat_offset;
  uint32_t tex_width, tex_height;
  float reflectivity;
  uint32_t _pad;
};
static_assert(sizeof(rt_blas_t) == RT_BLAS_STRIDE, "blas layout");

struct rt_tri_t { float v0[3], v1[3], v2[3]; };                    // geometry.h:1401-1405
static_assert(sizeof(rt_tri_t) == RT_TRI_BYTES, "tri layout");

struct rt_triex_t {            // common.h:39-43
  float N0[3], N1[3], N2[3];
  float uv0[2], uv1[2], uv2[2];
  uint32_t texId;
};
static_assert(sizeof(rt_triex_t) == RT_TRIEX_BYTES, "triEx layout");

struct rt_material_t {         // common.h:20-36
  float ambient[3], diffuse[3], specular[3], emissive[3];
  float shininess, ior, dissolve, reflectivity;
  int32_t diffuse_tex_id, illum;
  uint32_t tex_width, tex_height;
  uint64_t tex_offset;
};
static_assert(sizeof(rt_material_t) == RT_MAT_BYTES, "material layout");
