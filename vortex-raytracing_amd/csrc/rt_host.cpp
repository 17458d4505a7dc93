// rt_host -- C++ host program of the ray-tracing test on the Vortex host API, same flow and CLI as
// the reference's tests/regression/raytracing/{main,tracer}.cpp: build scene + BVH4 on the host,
// vx_dev_open, upload the 4 kernel images, 11 x vx_mem_alloc/vx_mem_address into kernel_arg_t,
// 9 x vx_copy_to_dev, shader binding table, 4 DCR writes, vx_upload_bytes(kernel_arg), vx_start,
// vx_ready_wait, vx_copy_from_dev, ASCII PPM.  It only calls vx_* (libvortex.so) and the host-side
// scene builder; which device runs it is decided by VORTEX_DRIVER (default here: hip).
//
//   rt_host [-k kernel.vxbin] [-n meshes] [-m model] [-w width] [-h height] [-s samples] [-d depth] [-o out.ppm]
//   model: an .obj file, or proc:cornell | proc:blob:<subdiv> | proc:atrium:<level> | proc:hairball:<strands>:<segs>
//   extensions: -S (one shadow ray per hit), -r y0:y1 (row window), -q (no perf dump), -L x,y,z (light position),
//               -N frames (repeat Tracer::run's call sequence -- vx_upload_bytes, vx_start, vx_ready_wait, vx_mem_free -- that many times and
//               print the time per frame, then the same with vx_copy_from_dev in every frame: the drop-in path's own rate from a C++ host)
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../../include/vortex_hip.h"

extern "C" {
int vx_dev_open(vx_device_h*);
int vx_dev_close(vx_device_h);
int vx_mem_alloc(vx_device_h, uint64_t, int, vx_buffer_h*);
int vx_mem_free(vx_buffer_h);
int vx_mem_address(vx_buffer_h, uint64_t*);
int vx_copy_to_dev(vx_buffer_h, const void*, uint64_t, uint64_t);
int vx_copy_from_dev(void*, vx_buffer_h, uint64_t, uint64_t);
int vx_start(vx_device_h, vx_buffer_h, vx_buffer_h);
int vx_ready_wait(vx_device_h, uint64_t);
int vx_dcr_write(vx_device_h, uint32_t, uint32_t);
int vx_upload_kernel_file(vx_device_h, const char*, vx_buffer_h*);
int vx_upload_bytes(vx_device_h, const void*, uint64_t, vx_buffer_h*);
void* vxs_scene_create_procedural(const char*, uint32_t, uint32_t, uint32_t);
void* vxs_scene_load_obj(const char*, uint32_t);
void vxs_scene_destroy(void*);
uint64_t vxs_scene_buffer(void*, int, const void**);
void vxs_scene_info(void*, uint32_t*, float*);
}

#define RT_CHECK(_expr)                                        \
  do {                                                         \
    int _ret = _expr;                                          \
    if (0 == _ret) break;                                      \
    printf("Error: '%s' returned %d!\n", #_expr, (int)_ret);   \
    return _ret;                                               \
  } while (false)

static const char* kernel_file = "kernel.vxbin";
static const char* output_file = "output.ppm";
static std::string model = "proc:cornell";
static uint32_t mesh_count = 1, dst_width = 640, dst_height = 480, spp = 1, max_depth = 1;
static bool shadow = false, quiet = false;
static uint32_t row0 = 0, row1 = 0, n_frames = 0;
static float light_pos[3] = {0, 10, -10};

static void write_ppm(const std::vector<uint8_t>& out, uint32_t w, uint32_t h, const char* file) {
  // P3, vertical flip, bytes 2,1,0 of each little-endian pixel (tracer.cpp:15-33: same bytes, but the
  // reference streams three float conversions per pixel through operator<<; here a 256-entry table of
  // decimal strings fills one buffer -- a 4K frame takes tens of milliseconds instead of seconds)
  char dec[256][4]; uint8_t len[256];
  for (int v = 0; v < 256; ++v) len[v] = (uint8_t)std::snprintf(dec[v], sizeof dec[v], "%d", v);
  std::string s = "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
  const size_t head = s.size();
  s.resize(head + (size_t)w * h * 12);
  char* q = &s[head];
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* row = out.data() + (size_t)(h - 1 - y) * w * 4;
    for (uint32_t x = 0; x < w; ++x) {
      const uint8_t* px = row + (size_t)x * 4;
      for (int k = 2; k >= 0; --k) {
        const uint8_t v = px[k];
        std::memcpy(q, dec[v], len[v]); q += len[v];
        *q++ = k ? ' ' : '\n';
      }
    }
  }
  s.resize((size_t)(q - s.data()));
  std::ofstream ofs(file, std::ios::binary);
  ofs.write(s.data(), (std::streamsize)s.size());
  std::printf("Image saved to: %s\n", file);
}

static void* make_scene() {
  if (model.rfind("proc:", 0) == 0) {
    std::vector<std::string> f;
    size_t p = 5;
    while (p <= model.size()) { size_t q = model.find(':', p); if (q == std::string::npos) q = model.size(); f.push_back(model.substr(p, q - p)); p = q + 1; }
    uint32_t a = f.size() > 1 ? (uint32_t)std::atoi(f[1].c_str()) : 0, b = f.size() > 2 ? (uint32_t)std::atoi(f[2].c_str()) : 0;
    if (f[0] == "blob" && a == 0) a = 6;
    if (f[0] == "atrium" && a == 0) a = 8;
    return vxs_scene_create_procedural(f[0].c_str(), a, b, 1);
  }
  return vxs_scene_load_obj(model.c_str(), mesh_count);
}

int main(int argc, char** argv) {
  int opt;
  while ((opt = getopt(argc, argv, "k:n:m:w:h:s:f:z:d:o:cSqr:N:L:")) != -1) {
    switch (opt) {
    case 'k': kernel_file = optarg; break;
    case 'n': mesh_count = std::atoi(optarg); break;
    case 'm': model = optarg; break;
    case 'w': dst_width = std::atoi(optarg); break;
    case 'h': dst_height = std::atoi(optarg); break;
    case 's': spp = std::atoi(optarg); break;
    case 'd': max_depth = std::atoi(optarg); break;
    case 'o': output_file = optarg; break;
    case 'f': case 'z': break;   // vfov / zoom feed camera fields the RTU kernel ignores (kernel.cpp:28-39)
    case 'c': std::printf("-c: the reference's CPU render() of this test is an empty function (tracer.cpp:290-304)\n"); return -1;
    case 'S': shadow = true; break;
    case 'q': quiet = true; break;
    case 'r': std::sscanf(optarg, "%u:%u", &row0, &row1); break;
    case 'N': n_frames = std::atoi(optarg); break;
    case 'L': std::sscanf(optarg, "%f,%f,%f", &light_pos[0], &light_pos[1], &light_pos[2]); break;
    default: std::printf("Usage: [-k kernel] [-n meshes] [-w width] [-h height] [-m model] [-s samples] [-d depth] [-o output]\n"); return -1;
    }
  }
  if (quiet) setenv("VORTEX_HIP_QUIET", "1", 1);
  auto t0 = std::chrono::steady_clock::now();
  void* scene = make_scene();
  if (!scene) { std::printf("Error: cannot build scene '%s'\n", model.c_str()); return -1; }
  uint32_t info[6];
  vxs_scene_info(scene, info, nullptr);
  auto t1 = std::chrono::steady_clock::now();
  std::printf("scene '%s': %u triangles, %u BVH4 nodes, depth %u, built in %.2f s\n", model.c_str(), info[5], info[3], info[0],
              std::chrono::duration<double>(t1 - t0).count());

  vx_device_h dev = nullptr;
  RT_CHECK(vx_dev_open(&dev));
  std::string dir(kernel_file);
  auto slash = dir.find_last_of('/');
  dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
  vx_buffer_h krnl, miss, closest, anyhit;
  RT_CHECK(vx_upload_kernel_file(dev, kernel_file, &krnl));
  RT_CHECK(vx_upload_kernel_file(dev, (dir + "miss.vxbin").c_str(), &miss));
  RT_CHECK(vx_upload_kernel_file(dev, (dir + "closest.vxbin").c_str(), &closest));
  RT_CHECK(vx_upload_kernel_file(dev, (dir + "anyhit.vxbin").c_str(), &anyhit));

  vx_rt_kernel_arg_t ka;
  std::memset(&ka, 0, sizeof ka);
  ka.dst_width = dst_width; ka.dst_height = dst_height; ka.samples_per_pixel = spp; ka.max_depth = max_depth;
  // which: 0 tlas 1 blas 2 bvh 3 tri 4 triEx 5 mat 6 tex 7 triIdx
  vx_buffer_h buf[8];
  uint64_t* addr[8] = {&ka.tlas_addr, &ka.blas_addr, &ka.qBvh_addr, &ka.tri_addr, &ka.triEx_addr, &ka.mat_addr, &ka.tex_addr, &ka.triIdx_addr};
  for (int i = 0; i < 8; ++i) {
    const void* p; uint64_t n = vxs_scene_buffer(scene, i, &p);
    RT_CHECK(vx_mem_alloc(dev, n, VX_MEM_READ, &buf[i]));
    RT_CHECK(vx_mem_address(buf[i], addr[i]));
  }
  vx_buffer_h bvh2, out, sbt;
  RT_CHECK(vx_mem_alloc(dev, 64, VX_MEM_READ, &bvh2));            // un-quantised nodes: never read by the RTU path
  RT_CHECK(vx_mem_address(bvh2, &ka.bvh_addr));
  RT_CHECK(vx_mem_alloc(dev, (uint64_t)dst_width * dst_height * 4, VX_MEM_WRITE, &out));
  RT_CHECK(vx_mem_address(out, &ka.dst_addr));
  RT_CHECK(vx_mem_alloc(dev, 32, VX_MEM_READ, &sbt));
  RT_CHECK(vx_mem_address(sbt, &ka.sbt_addr));

  // Tracer::setup: lights (main.cpp:34-41), uploads, SBT, DCRs
  const float* lp = light_pos; const float lc[3] = {1, 1, 1}, am[3] = {0.4f, 0.4f, 0.4f}, bg[3] = {0.4f, 0.35f, 0.25f};
  std::memcpy(ka.light_pos, lp, 12); std::memcpy(ka.light_color, lc, 12);
  std::memcpy(ka.ambient_color, am, 12); std::memcpy(ka.background_color, bg, 12);
  for (int i = 0; i < 8; ++i) {
    const void* p; uint64_t n = vxs_scene_buffer(scene, i, &p);
    RT_CHECK(vx_copy_to_dev(buf[i], p, 0, n));
  }
  uint64_t tmp_sbt[4] = {0, 0, 0, 0};
  RT_CHECK(vx_mem_address(miss, &tmp_sbt[0]));
  RT_CHECK(vx_mem_address(closest, &tmp_sbt[1]));
  RT_CHECK(vx_mem_address(anyhit, &tmp_sbt[3]));
  RT_CHECK(vx_copy_to_dev(sbt, tmp_sbt, 0, sizeof tmp_sbt));
  RT_CHECK(vx_dcr_write(dev, VX_DCR_BASE_RTX_TLAS_PTR, (uint32_t)ka.tlas_addr));
  RT_CHECK(vx_dcr_write(dev, VX_DCR_BASE_RTX_BLAS_PTR, (uint32_t)ka.blas_addr));
  RT_CHECK(vx_dcr_write(dev, VX_DCR_BASE_RTX_BVH_PTR, (uint32_t)ka.qBvh_addr));
  RT_CHECK(vx_dcr_write(dev, VX_DCR_BASE_RTX_TRI_PTR, (uint32_t)ka.tri_addr));
  if (shadow) RT_CHECK(vx_dcr_write(dev, VX_DCR_HIP_SHADOW_RAYS, 1));
  if (row1) { RT_CHECK(vx_dcr_write(dev, VX_DCR_HIP_ROW_BEGIN, row0)); RT_CHECK(vx_dcr_write(dev, VX_DCR_HIP_ROW_END, row1)); }

  // Tracer::run
  std::printf("Begin rendering to %ux%u framebuffer.\n", dst_width, dst_height);
  std::vector<uint8_t> h_output((size_t)dst_width * dst_height * 4);
  vx_buffer_h args;
  RT_CHECK(vx_upload_bytes(dev, &ka, sizeof ka, &args));
  auto t2 = std::chrono::steady_clock::now();
  RT_CHECK(vx_start(dev, krnl, args));
  RT_CHECK(vx_ready_wait(dev, VX_MAX_TIMEOUT));
  auto t3 = std::chrono::steady_clock::now();
  RT_CHECK(vx_copy_from_dev(h_output.data(), out, 0, h_output.size()));
  std::printf("kernel wall time (start..ready_wait): %.3f ms\n", std::chrono::duration<double, std::milli>(t3 - t2).count());
  write_ppm(h_output, dst_width, dst_height, output_file);
  if (n_frames) {
    // the frame loop of a host that renders a sequence: per frame what Tracer::run does (tracer.cpp:262-288)
    for (int with_copy = 0; with_copy < 2; ++with_copy) {
      for (uint32_t f = 0; f < 5 + n_frames; ++f) {
        if (f == 5) t2 = std::chrono::steady_clock::now();
        vx_buffer_h a2;
        RT_CHECK(vx_upload_bytes(dev, &ka, sizeof ka, &a2));
        RT_CHECK(vx_start(dev, krnl, a2));
        RT_CHECK(vx_ready_wait(dev, VX_MAX_TIMEOUT));
        if (with_copy) RT_CHECK(vx_copy_from_dev(h_output.data(), out, 0, h_output.size()));
        vx_mem_free(a2);
      }
      t3 = std::chrono::steady_clock::now();
      std::printf("frame loop (%u frames, %s): %.4f ms per frame\n", n_frames, with_copy ? "upload_bytes + start + ready_wait + copy_from_dev" : "upload_bytes + start + ready_wait",
                  std::chrono::duration<double, std::milli>(t3 - t2).count() / n_frames);
    }
  }

  for (int i = 0; i < 8; ++i) vx_mem_free(buf[i]);
  vx_mem_free(bvh2); vx_mem_free(out); vx_mem_free(sbt); vx_mem_free(args);
  vx_mem_free(krnl); vx_mem_free(miss); vx_mem_free(closest); vx_mem_free(anyhit);
  vx_dev_close(dev);
  vxs_scene_destroy(scene);
  return 0;
}
