// TEST INFRASTRUCTURE ONLY (oracle/_ref): C entry points around the reference's software ray caster
// (tests/regression/raycast/{mesh,surface,bvh,scene}.cpp and render.h), compiled where those sources lie.
// Builds scenes with the reference's own BVH2/TLAS builder, exposes the byte images of the buffers the kernel
// reads (tracer.cpp:107-118) and renders with the reference's own GenerateRay/Trace (its `-c` CPU path,
// tracer.cpp:249-263) -- the pin for oracle/rt_oracle.c's restatement of render.h.
#include "scene.h"
#define __UNIFORM__
#include "render.h"
#include <cstring>
#include <vector>

extern "C" {

struct rcref_scene_t { Scene* scene; };

// rotate != 0 applies Tracer::setup's scene rotation (tracer.cpp:172-176) before the build
void* rcref_scene_create(const char* const* objs, const char* const* texs, const float* refl, int n, int rotate) {
  std::vector<Mesh*> meshes(n);
  for (int i = 0; i < n; ++i) meshes[i] = new Mesh(objs[i], texs[i], refl[i]);
  auto s = new rcref_scene_t;
  s->scene = new Scene(meshes);
  if (s->scene->init() != 0) { delete s->scene; delete s; return nullptr; }
  if (rotate) {
    auto T = mat4_t::RotateX(-PI / 4) * mat4_t::RotateY(PI / 4);
    s->scene->applyTransform(T);
  }
  s->scene->build();
  return s;
}

void rcref_scene_destroy(void* h) {
  auto s = (rcref_scene_t*)h;
  delete s->scene;
  delete s;
}

// which: 0 tlas 1 blas 2 bvh 3 tri 4 triEx 5 triIdx 6 tex
uint64_t rcref_scene_buffer(void* h, int which, const void** ptr) {
  auto sc = ((rcref_scene_t*)h)->scene;
  switch (which) {
  case 0: *ptr = sc->tlas_nodes().data(); return sc->tlas_nodes().size() * sizeof(tlas_node_t);
  case 1: *ptr = sc->blas_nodes().data(); return sc->blas_nodes().size() * sizeof(blas_node_t);
  case 2: *ptr = sc->bvh_nodes().data(); return sc->bvh_nodes().size() * sizeof(bvh_node_t);
  case 3: *ptr = sc->tri_buf().data(); return sc->tri_buf().size() * sizeof(tri_t);
  case 4: *ptr = sc->triEx_buf().data(); return sc->triEx_buf().size() * sizeof(tri_ex_t);
  case 5: *ptr = sc->triIdx_buf().data(); return sc->triIdx_buf().size() * sizeof(uint32_t);
  case 6: *ptr = sc->tex_buf().data(); return sc->tex_buf().size();
  }
  *ptr = nullptr;
  return 0;
}

uint32_t rcref_tlas_root(void* h) { return ((rcref_scene_t*)h)->scene->tlas_root(); }

uint32_t rcref_sizeof(int which) {
  switch (which) {
  case 0: return sizeof(tlas_node_t);
  case 1: return sizeof(blas_node_t);
  case 2: return sizeof(bvh_node_t);
  case 3: return sizeof(tri_t);
  case 4: return sizeof(tri_ex_t);
  case 5: return sizeof(kernel_arg_t);
  }
  return 0;
}

// Tracer::setup camera + viewplane (tracer.cpp:183-203): out = pos(3) forward(3) right(3) up(3) viewplane(2)
void rcref_camera(void* h, float vfov_deg, float zoom, uint32_t w, uint32_t hgt, float* out14) {
  auto sc = ((rcref_scene_t*)h)->scene;
  float3_t camera_pos, camera_target, camera_up;
  sc->computeFramingCamera(vfov_deg * DEG2RAD, zoom, &camera_pos, &camera_target, &camera_up);
  float3_t forward = normalize(camera_target - camera_pos);
  float3_t right = normalize(cross(forward, camera_up));
  float3_t up = cross(right, forward);
  float aspect_ratio = float(w) / hgt;
  float viewport_height = 2.0f * tan(vfov_deg * 0.5f);   // as written there: the argument is in degrees
  float viewport_width = viewport_height * aspect_ratio;
  const float v[14] = {camera_pos.x, camera_pos.y, camera_pos.z, forward.x, forward.y, forward.z, right.x, right.y, right.z,
                       up.x, up.y, up.z, viewport_width, viewport_height};
  std::memcpy(out14, v, sizeof v);
}

// The reference's CPU path (Tracer::render, tracer.cpp:249-263) on this scene: cam14 as from rcref_camera,
// light12 = light_pos, light_color, ambient_color, background_color.
int rcref_render(void* h, uint32_t w, uint32_t hgt, uint32_t spp, uint32_t max_depth, const float* cam14, const float* light12, uint32_t* out) {
  auto sc = ((rcref_scene_t*)h)->scene;
  kernel_arg_t a{};
  a.dst_width = w; a.dst_height = hgt; a.dst_addr = (uint64_t)out;
  a.tri_addr = (uint64_t)sc->tri_buf().data();
  a.triEx_addr = (uint64_t)sc->triEx_buf().data();
  a.triIdx_addr = (uint64_t)sc->triIdx_buf().data();
  a.bvh_addr = (uint64_t)sc->bvh_nodes().data();
  a.tlas_addr = (uint64_t)sc->tlas_nodes().data();
  a.blas_addr = (uint64_t)sc->blas_nodes().data();
  a.tex_addr = (uint64_t)sc->tex_buf().data();
  a.tlas_root = sc->tlas_root();
  a.camera_pos = float3_t(cam14[0], cam14[1], cam14[2]);
  a.camera_forward = float3_t(cam14[3], cam14[4], cam14[5]);
  a.camera_right = float3_t(cam14[6], cam14[7], cam14[8]);
  a.camera_up = float3_t(cam14[9], cam14[10], cam14[11]);
  a.viewplane = {cam14[12], cam14[13]};
  a.samples_per_pixel = spp; a.max_depth = max_depth;
  a.light_pos = float3_t(light12[0], light12[1], light12[2]);
  a.light_color = float3_t(light12[3], light12[4], light12[5]);
  a.ambient_color = float3_t(light12[6], light12[7], light12[8]);
  a.background_color = float3_t(light12[9], light12[10], light12[11]);
  auto arg = &a;
  for (uint32_t y = 0; y < arg->dst_height; ++y) {
    for (uint32_t x = 0; x < arg->dst_width; ++x) {
      uint32_t out_idx = y * arg->dst_width + x;
      float3_t color = float3_t(0, 0, 0);
      for (uint32_t s = 0; s < arg->samples_per_pixel; ++s) {
        auto ray = GenerateRay(x, y, arg);
        color += Trace(ray, arg);
      }
      out[out_idx] = RGB32FtoRGB8(color);
    }
  }
  return 0;
}

// The same loop (Tracer::render, tracer.cpp:249-263: GenerateRay + Trace of render.h) over caller-provided buffers in the
// raycast formats, rows [y0,y1) only -- BASELINE.md s2's baseline B1 (the reference's software ray caster, single thread, render
// loop only) on a scene too large for the reference's recursive builder to be part of a bench run.  ptrs: tri, triEx, triIdx,
// bvh, tlas, blas, tex.  Returns the number of primary rays traced.
uint64_t rcref_render_buffers(const void* const* ptrs, uint32_t tlas_root, uint32_t w, uint32_t hgt, uint32_t y0, uint32_t y1, uint32_t x_step,
                              uint32_t spp, uint32_t max_depth, const float* cam14, const float* light12, uint32_t* out) {
  kernel_arg_t a{};
  a.dst_width = w; a.dst_height = hgt; a.dst_addr = (uint64_t)out;
  a.tri_addr = (uint64_t)ptrs[0]; a.triEx_addr = (uint64_t)ptrs[1]; a.triIdx_addr = (uint64_t)ptrs[2];
  a.bvh_addr = (uint64_t)ptrs[3]; a.tlas_addr = (uint64_t)ptrs[4]; a.blas_addr = (uint64_t)ptrs[5]; a.tex_addr = (uint64_t)ptrs[6];
  a.tlas_root = tlas_root;
  a.camera_pos = float3_t(cam14[0], cam14[1], cam14[2]);
  a.camera_forward = float3_t(cam14[3], cam14[4], cam14[5]);
  a.camera_right = float3_t(cam14[6], cam14[7], cam14[8]);
  a.camera_up = float3_t(cam14[9], cam14[10], cam14[11]);
  a.viewplane = {cam14[12], cam14[13]};
  a.samples_per_pixel = spp; a.max_depth = max_depth;
  a.light_pos = float3_t(light12[0], light12[1], light12[2]);
  a.light_color = float3_t(light12[3], light12[4], light12[5]);
  a.ambient_color = float3_t(light12[6], light12[7], light12[8]);
  a.background_color = float3_t(light12[9], light12[10], light12[11]);
  auto arg = &a;
  uint64_t n = 0;
  if (x_step == 0) x_step = 1;
  for (uint32_t y = y0; y < y1 && y < arg->dst_height; ++y) {
    for (uint32_t x = 0; x < arg->dst_width; x += x_step) {
      uint32_t out_idx = y * arg->dst_width + x;
      float3_t color = float3_t(0, 0, 0);
      for (uint32_t s = 0; s < arg->samples_per_pixel; ++s) {
        auto ray = GenerateRay(x, y, arg);
        color += Trace(ray, arg);
        ++n;
      }
      out[out_idx] = RGB32FtoRGB8(color);
    }
  }
  return n;
}

// Radiance of ARBITRARY rays through the reference's Trace (render.h:210-277: the iterative mirror bounce with reflective instances,
// reflectivity from the instance record, texSample of the instance's texture) -- e.g. for the RTU test's own camera rays.  This is what
// pins the RTU path's mirror arm (shaders/closest.cpp:95-121, same formulas: R = normalize(d - 2 N (N.d)), origin I + R * 0.001,
// colour = diffuse * (1 - r) + r * next) to reference object code: the RTU shaders themselves only build for RISC-V.
// light12 = light_pos, light_color, ambient_color, background_color.  out_rgb: 3 floats per ray; out_rgb8 (optional): RGB32FtoRGB8 of it.
int rcref_radiance(void* h, const float* rays6, uint64_t n, uint32_t max_depth, const float* light12, float* out_rgb, uint32_t* out_rgb8) {
  auto sc = ((rcref_scene_t*)h)->scene;
  kernel_arg_t a{};
  a.tri_addr = (uint64_t)sc->tri_buf().data();
  a.triEx_addr = (uint64_t)sc->triEx_buf().data();
  a.triIdx_addr = (uint64_t)sc->triIdx_buf().data();
  a.bvh_addr = (uint64_t)sc->bvh_nodes().data();
  a.tlas_addr = (uint64_t)sc->tlas_nodes().data();
  a.blas_addr = (uint64_t)sc->blas_nodes().data();
  a.tex_addr = (uint64_t)sc->tex_buf().data();
  a.tlas_root = sc->tlas_root();
  a.samples_per_pixel = 1; a.max_depth = max_depth;
  a.light_pos = float3_t(light12[0], light12[1], light12[2]);
  a.light_color = float3_t(light12[3], light12[4], light12[5]);
  a.ambient_color = float3_t(light12[6], light12[7], light12[8]);
  a.background_color = float3_t(light12[9], light12[10], light12[11]);
  for (uint64_t i = 0; i < n; ++i) {
    ray_t r{float3_t(rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2]), float3_t(rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5])};
    float3_t c = Trace(r, &a);
    out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    if (out_rgb8) out_rgb8[i] = RGB32FtoRGB8(c);
  }
  return 0;
}

// one ray through the reference's TLASIntersect (hit record) -- for traversal-only fixtures
void rcref_trace(void* h, const float* ray6, float* out_dist, float* out_bc3, uint32_t* out_blas, uint32_t* out_tri) {
  auto sc = ((rcref_scene_t*)h)->scene;
  ray_t r{float3_t(ray6[0], ray6[1], ray6[2]), float3_t(ray6[3], ray6[4], ray6[5])};
  ray_hit_t hit;
  TLASIntersect(r, sc->tlas_root(), sc->tlas_nodes().data(), sc->blas_nodes().data(), sc->bvh_nodes().data(),
                sc->triIdx_buf().data(), sc->tri_buf().data(), &hit);
  *out_dist = hit.dist; out_bc3[0] = hit.bcoords.x; out_bc3[1] = hit.bcoords.y; out_bc3[2] = hit.bcoords.z;
  *out_blas = hit.blasIdx; *out_tri = hit.triIdx;
}

}  // extern "C"
