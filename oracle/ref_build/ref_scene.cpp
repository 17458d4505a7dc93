// TEST INFRASTRUCTURE ONLY (oracle/_ref): C entry points around the reference's own scene
// builder (tests/regression/raytracing/{mesh,surface,bvh,scene,treelet}.cpp), compiled
// where those sources lie.  Produces the byte images of the buffers the RTU path reads
// (tracer.cpp:124-161) and exposes the reference's shading helpers for golden pixels.
#include "scene.h"
#include "rtx_shading.h"
#include <vector>
#include <cstring>

extern "C" {

struct vxref_scene_t {
  Scene* scene;
};

void* vxref_scene_create(const char* const* obj_paths, int n_meshes) {
  std::vector<Mesh*> meshes(n_meshes);
  for (int i = 0; i < n_meshes; ++i) meshes[i] = new Mesh(obj_paths[i]);
  auto s = new vxref_scene_t;
  s->scene = new Scene(meshes);
  if (s->scene->init() != 0) { delete s->scene; delete s; return nullptr; }
  s->scene->build();  // Tracer::setup -> scene_->build() (tracer.cpp:181)
  return s;
}

void vxref_scene_destroy(void* h) {
  auto s = (vxref_scene_t*)h;
  delete s->scene;
  delete s;
}

// which: 0 tlas_qnodes 1 blas_nodes 2 bvh_quantized_nodes 3 tri 4 triEx 5 mat 6 tex 7 bvh_nodes(BVH2-ish wide nodes)
uint64_t vxref_scene_buffer(void* h, int which, const void** ptr) {
  auto sc = ((vxref_scene_t*)h)->scene;
  switch (which) {
  case 0: *ptr = sc->tlas_qnodes().data(); return sc->tlas_qnodes().size() * sizeof(bvh_quantized_node_t);
  case 1: *ptr = sc->blas_nodes().data(); return sc->blas_nodes().size() * sizeof(blas_node_t);
  case 2: *ptr = sc->bvh_quantized_nodes().data(); return sc->bvh_quantized_nodes().size() * sizeof(bvh_quantized_node_t);
  case 3: *ptr = sc->tri_buf().data(); return sc->tri_buf().size() * sizeof(tri_t);
  case 4: *ptr = sc->triEx_buf().data(); return sc->triEx_buf().size() * sizeof(tri_ex_t);
  case 5: *ptr = sc->mat_buf().data(); return sc->mat_buf().size() * sizeof(material_info_t);
  case 6: *ptr = sc->tex_buf().data(); return sc->tex_buf().size();
  case 7: *ptr = sc->bvh_nodes().data(); return sc->bvh_nodes().size() * sizeof(bvh_node_t);
  }
  *ptr = nullptr;
  return 0;
}

uint32_t vxref_sizeof(int which) {
  switch (which) {
  case 0: return sizeof(bvh_quantized_node_t);
  case 1: return sizeof(blas_node_t);
  case 2: return sizeof(tri_t);
  case 3: return sizeof(tri_ex_t);
  case 4: return sizeof(material_info_t);
  case 5: return sizeof(kernel_arg_t);
  case 6: return sizeof(bvh_node_t);
  }
  return 0;
}

// Camera ray of the RTU kernel (kernel.cpp:28-39), host-compiled.
void vxref_generate_ray(uint32_t x, uint32_t y, uint32_t w, uint32_t h, float* out6) {
  auto pos = float3_t(0.0, 100.0, 0.0);
  auto front = float3_t(1.0, 0.0, 0.0);
  float FOV = 1.0;
  float u = (x * 2.0 - w) / h;
  float v = (y * 2.0 - h) / h;
  auto right = cross(front, float3_t(0.0, 1.0, 0.0));
  auto up = cross(right, front);
  auto dir = normalize(u * right + v * up + FOV * front);
  out6[0] = pos.x; out6[1] = pos.y; out6[2] = pos.z;
  out6[3] = dir.x; out6[4] = dir.y; out6[5] = dir.z;
}

// Closest-hit shade restated with the reference's own helpers (closest.cpp:57-127 for
// reflectivity==0 / max_depth==1, which is all the shipped RTU test exercises), and the
// miss shader (miss.cpp:9-14).  Returns f32 colour and the packed RGB8 (common.h:149-154).
uint32_t vxref_shade(const float* ray6, float dist, float bx, float by, float bz,
                     uint32_t blasIdx, uint32_t triIdx, int is_hit,
                     const blas_node_t* blas_ptr, const tri_ex_t* triEx_ptr,
                     const material_info_t* mat_ptr, const uint8_t* tex_ptr,
                     const float* ambient3, const float* light_color3, const float* light_pos3,
                     const float* background3, float* out_color3) {
  float3_t radiance = {0, 0, 0};
  float3_t background(background3[0], background3[1], background3[2]);
  if (!is_hit) {
    radiance = background;
  } else {
    float throughput = 1.0f;
    ray_t ray;
    ray.orig = float3_t(ray6[0], ray6[1], ray6[2]);
    ray.dir = float3_t(ray6[3], ray6[4], ray6[5]);
    float3_t bcoords(bx, by, bz);
    auto& blas = blas_ptr[blasIdx];
    const tri_ex_t& triEx = triEx_ptr[triIdx];
    const material_info_t& mat = mat_ptr[triEx.texId];
    float3_t I = ray.orig + ray.dir * dist;
    float3_t N = triEx.N1 * bcoords.x + triEx.N2 * bcoords.y + triEx.N0 * bcoords.z;
    mat4_t invTranspose = blas.invTransform.transposed();
    N = normalize(TransformVector(N, invTranspose));
    float2_t uv = triEx.uv1 * bcoords.x + triEx.uv2 * bcoords.y + triEx.uv0 * bcoords.z;
    float3_t texColor;
    if (mat.diffuse_tex_id >= 0) {
      auto tex_pixels = reinterpret_cast<const uint32_t*>(tex_ptr + mat.tex_offset);
      texColor = texSample(uv, tex_pixels, mat.tex_width, mat.tex_height);
    } else {
      texColor = mat.diffuse;
    }
    float3_t diffuse = diffuseLighting(I, N, texColor,
                                       float3_t(ambient3[0], ambient3[1], ambient3[2]),
                                       float3_t(light_color3[0], light_color3[1], light_color3[2]),
                                       float3_t(light_pos3[0], light_pos3[1], light_pos3[2]));
    auto reflectivity = blas.reflectivity;
    radiance += throughput * diffuse * (1 - reflectivity);
    throughput *= reflectivity;
    radiance += background * throughput;
  }
  out_color3[0] = radiance.x; out_color3[1] = radiance.y; out_color3[2] = radiance.z;
  return RGB32FtoRGB8(radiance);
}

// The reference's RNG helpers (common.h:129-147), host-compiled: hash[i] = WangHash(seed + i); then one xorshift stream seeded
// with WangHash(seed) (0 -> 1, as every user must do: xorshift has the fixed point 0): ints[i] = RandomInt, and a second stream
// from the same seed: floats[i] = RandomFloat.
void vxref_rng(uint32_t seed, uint32_t n, uint32_t* hash, uint32_t* ints, float* floats) {
  for (uint32_t i = 0; i < n; ++i) hash[i] = WangHash(seed + i);
  uint32_t s = WangHash(seed);
  if (s == 0) s = 1;
  uint32_t t = s;
  for (uint32_t i = 0; i < n; ++i) ints[i] = RandomInt(&s);
  for (uint32_t i = 0; i < n; ++i) floats[i] = RandomFloat(&t);
}

}  // extern "C"
