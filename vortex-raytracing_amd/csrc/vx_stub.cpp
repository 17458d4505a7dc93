// libvortex.so -- stand-alone build of the Vortex host API (reference runtime/include/vortex.h:80-145)
// so that host programs link and run where the reference tree is absent (the GPU box).  Same
// contract as the reference dispatcher (runtime/stub/vortex.cpp:56-97): the backend is
// "libvortex-$VORTEX_DRIVER.so", found through the dynamic loader path, resolved once per
// process, and must export `vx_dev_init(callbacks_t*)`.  The default driver here is "hip".
// The reference's own libvortex.so can be used instead of this file unchanged.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../../include/vortex_hip.h"

namespace {
callbacks_t g_cb;
void* g_lib = nullptr;
typedef int (*dev_init_fn)(callbacks_t*);

int load_backend() {
  if (g_lib) return 0;
  const char* drv = std::getenv("VORTEX_DRIVER");
  std::string name = std::string("libvortex-") + (drv ? drv : "hip") + ".so";
  void* lib = dlopen(name.c_str(), RTLD_LAZY);
  if (!lib) {
    // also look next to this library (in-tree builds)
    Dl_info info;
    if (dladdr((void*)&load_backend, &info) && info.dli_fname) {
      std::string here(info.dli_fname);
      auto pos = here.find_last_of('/');
      if (pos != std::string::npos) lib = dlopen((here.substr(0, pos + 1) + name).c_str(), RTLD_LAZY);
    }
  }
  if (!lib) { std::fprintf(stderr, "Cannot open library: %s\n", dlerror()); return 1; }
  auto init = (dev_init_fn)dlsym(lib, "vx_dev_init");
  if (!init) { std::fprintf(stderr, "Cannot load symbol 'vx_dev_init': %s\n", dlerror()); dlclose(lib); return 1; }
  if (init(&g_cb) != 0) { dlclose(lib); return 1; }
  g_lib = lib;
  return 0;
}

int read_file(const char* filename, std::vector<char>& out) {
  std::ifstream ifs(filename, std::ios::binary);
  if (!ifs) { std::fprintf(stderr, "Error: %s not found\n", filename); return -1; }
  ifs.seekg(0, ifs.end);
  auto size = ifs.tellg();
  out.resize((size_t)size);
  ifs.seekg(0, ifs.beg);
  ifs.read(out.data(), size);
  return 0;
}
}  // namespace

extern "C" {

int vx_dcr_write(vx_device_h hdevice, uint32_t addr, uint32_t value);
int vx_dump_perf(vx_device_h hdevice, FILE* stream);
int vx_mem_free(vx_buffer_h hbuffer);

int vx_dev_open(vx_device_h* hdevice) {
  if (load_backend() != 0) return 1;
  vx_device_h h;
  int err = g_cb.dev_open(&h);
  if (err) return err;
  // the five start-up DCRs the reference writes at open (stub/vortex.cpp:25-49)
  const uint64_t startup = 0x80000000ull;
  if ((err = vx_dcr_write(h, VX_DCR_BASE_STARTUP_ADDR0, (uint32_t)startup))) return err;
  if ((err = vx_dcr_write(h, VX_DCR_BASE_STARTUP_ADDR1, (uint32_t)(startup >> 32)))) return err;
  if ((err = vx_dcr_write(h, VX_DCR_BASE_STARTUP_ARG0, 0))) return err;
  if ((err = vx_dcr_write(h, VX_DCR_BASE_STARTUP_ARG1, 0))) return err;
  if ((err = vx_dcr_write(h, VX_DCR_BASE_MPM_CLASS, 0))) return err;
  *hdevice = h;
  return 0;
}

int vx_dev_close(vx_device_h hdevice) {
  if (std::getenv("VORTEX_HIP_QUIET") == nullptr) vx_dump_perf(hdevice, stdout);  // stub/vortex.cpp:99-104
  return g_cb.dev_close(hdevice);  // the backend stays loaded: HIP runtimes do not survive dlclose
}

int vx_dev_caps(vx_device_h h, uint32_t id, uint64_t* v) { return g_cb.dev_caps(h, id, v); }
int vx_mem_alloc(vx_device_h h, uint64_t size, int flags, vx_buffer_h* b) { return g_cb.mem_alloc(h, size, flags, b); }
int vx_mem_reserve(vx_device_h h, uint64_t addr, uint64_t size, int flags, vx_buffer_h* b) { return g_cb.mem_reserve(h, addr, size, flags, b); }
int vx_mem_free(vx_buffer_h b) { return g_lib ? g_cb.mem_free(b) : (b ? -1 : 0); }
int vx_mem_access(vx_buffer_h b, uint64_t off, uint64_t size, int flags) { return g_cb.mem_access(b, off, size, flags); }
int vx_mem_address(vx_buffer_h b, uint64_t* addr) { return g_cb.mem_address(b, addr); }
int vx_mem_info(vx_device_h h, uint64_t* f, uint64_t* u) { return g_cb.mem_info(h, f, u); }
int vx_copy_to_dev(vx_buffer_h b, const void* p, uint64_t off, uint64_t size) { return g_cb.copy_to_dev(b, p, off, size); }
int vx_copy_from_dev(void* p, vx_buffer_h b, uint64_t off, uint64_t size) { return g_cb.copy_from_dev(p, b, off, size); }
int vx_start(vx_device_h h, vx_buffer_h k, vx_buffer_h a) { return g_cb.start(h, k, a); }
int vx_ready_wait(vx_device_h h, uint64_t timeout) { return g_cb.ready_wait(h, timeout); }
int vx_dcr_read(vx_device_h h, uint32_t addr, uint32_t* v) { return g_cb.dcr_read(h, addr, v); }
int vx_dcr_write(vx_device_h h, uint32_t addr, uint32_t v) { return g_cb.dcr_write(h, addr, v); }

int vx_mpm_query(vx_device_h h, uint32_t addr, uint32_t core_id, uint64_t* value) {
  if (core_id != 0xffffffffu) return g_cb.mpm_query(h, addr, core_id, value);
  uint64_t cores = 0, sum = 0, cur = 0;   // all-cores sum (stub/vortex.cpp:160-176)
  int err = g_cb.dev_caps(h, VX_CAPS_NUM_CORES, &cores);
  if (err) return err;
  for (uint32_t i = 0; i < cores; ++i) {
    if ((err = g_cb.mpm_query(h, addr, i, &cur))) return err;
    sum += cur;
  }
  *value = sum;
  return 0;
}

// .vxbin = u64 min_vma, u64 max_vma, image bytes (kernel/scripts/vxbin.py:53-74; stub/utils.cpp:25-61)
int vx_upload_kernel_bytes(vx_device_h h, const void* content, uint64_t size, vx_buffer_h* out) {
  if (!h || !content || size <= 8 || !out) return -1;
  uint64_t hdr[2];
  if (size < sizeof hdr) return -1;
  std::memcpy(hdr, content, sizeof hdr);
  const uint64_t bin_size = size - sizeof hdr, runtime_size = hdr[1] - hdr[0];
  vx_buffer_h b;
  int err = vx_mem_reserve(h, hdr[0], runtime_size, 0, &b);
  if (err) return err;
  if ((err = vx_mem_access(b, 0, bin_size, VX_MEM_READ)) ||
      (err = vx_mem_access(b, bin_size, runtime_size - bin_size, VX_MEM_READ_WRITE)) ||
      (err = vx_copy_to_dev(b, (const char*)content + sizeof hdr, 0, bin_size))) {
    vx_mem_free(b);
    return err;
  }
  *out = b;
  return 0;
}

int vx_upload_bytes(vx_device_h h, const void* content, uint64_t size, vx_buffer_h* out) {
  if (!h || !content || size == 0 || !out) return -1;
  vx_buffer_h b;
  int err = vx_mem_alloc(h, size, VX_MEM_READ, &b);
  if (err) return err;
  if ((err = vx_copy_to_dev(b, content, 0, size))) { vx_mem_free(b); return err; }
  *out = b;
  return 0;
}

int vx_upload_kernel_file(vx_device_h h, const char* filename, vx_buffer_h* out) {
  if (!h || !filename || !out) return -1;
  std::vector<char> data;
  if (read_file(filename, data) != 0) return -1;
  return vx_upload_kernel_bytes(h, data.data(), data.size(), out);
}

int vx_upload_file(vx_device_h h, const char* filename, vx_buffer_h* out) {
  if (!h || !filename || !out) return -1;
  std::vector<char> data;
  if (read_file(filename, data) != 0) return -1;
  return vx_upload_bytes(h, data.data(), data.size(), out);
}

int vx_check_occupancy(vx_device_h h, uint32_t group_size, uint32_t* max_localmem) {
  uint64_t warps = 0, threads = 0, lmem = 0;
  int err;
  if ((err = vx_dev_caps(h, VX_CAPS_NUM_WARPS, &warps)) || (err = vx_dev_caps(h, VX_CAPS_NUM_THREADS, &threads))) return err;
  if (group_size > warps * threads) {
    std::printf("Error: cannot schedule kernel with group_size > threads_per_core (%u,%llu)\n", group_size, (unsigned long long)(warps * threads));
    return -1;
  }
  if (max_localmem) {
    if ((err = vx_dev_caps(h, VX_CAPS_LOCAL_MEM_SIZE, &lmem))) return err;
    const uint64_t warps_per_group = (group_size + threads - 1) / threads;
    const uint64_t groups = warps_per_group ? warps / warps_per_group : 1;
    *max_localmem = (uint32_t)(lmem / (groups ? groups : 1));
  }
  return 0;
}

// Base-class perf line of the reference (stub/perf.cpp:181-227): instrs, cycles, IPC.  For this
// backend "instrs" are rays traced by the last run and cycles are its shader-clock duration.
int vx_dump_perf(vx_device_h h, FILE* stream) {
  if (!stream) stream = stdout;
  uint64_t cores = 0, cycles = 0, cur = 0, instrs = 0, hz = 0;
  int err;
  if ((err = vx_dev_caps(h, VX_CAPS_NUM_CORES, &cores))) return err;
  if ((err = vx_dev_caps(h, VX_CAPS_CLOCK_RATE, &hz))) return err;
  for (uint32_t c = 0; c < cores; ++c) {
    if ((err = g_cb.mpm_query(h, VX_CSR_MCYCLE, c, &cur))) return err;
    if (cur > cycles) cycles = cur;
    if ((err = g_cb.mpm_query(h, VX_CSR_MINSTRET, c, &cur))) return err;
    instrs += cur;
  }
  const double ipc = cycles ? (double)instrs / (double)cycles : 0.0;
  std::fprintf(stream, "PERF: rays=%llu, cycles=%llu, rays/cycle=%f", (unsigned long long)instrs, (unsigned long long)cycles, ipc);
  if (cycles && hz) std::fprintf(stream, ", Mrays/s=%.1f", (double)instrs / ((double)cycles / (double)hz) / 1e6);
  std::fputc('\n', stream);
  return 0;
}

}  // extern "C"
