// libvortex-hip.so -- MI355X driver backend behind the reference's runtime plug-in boundary.
//
// Exports `vx_dev_init(callbacks_t*)` (reference runtime/common/callbacks.inc:20) and fills the 16
// callbacks with a HIP implementation of what runtime/simx/vortex.cpp does on the simulator:
// a device address space (64-byte blocks from USER_BASE_ADDR, like sim/common/mem_alloc.h), copies,
// DCRs, `start` = decode the uploaded kernel tag + kernel_arg_t and launch the HIP render kernel on
// a stream, `ready_wait` = join that stream.  It cannot execute RISC-V: the uploaded ".vxbin" blobs
// are interpreted as kernel selectors (16-byte vxbin header + "VXHIP1:<name>" tag; see
// INTEGRATION.md).  Anything else makes `start` fail loudly with -1.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/vortex_hip.h"
#include "rt_types.h"
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl.so.1 is loaded with dlopen when VORTEX_HIP_GATHER=rccl asks for it

extern "C" uint32_t* vxrt_status_word_device(void);   // rt_kernels.hip

// rays counter and status word of a run into host memory the device can write (see vx_device::enqueue_readback)
// ... and clears the counter for the next run: a run is its own launches + this one.  (Rounds 2-3 opened a run with a fill of the counter and
// an event record and closed it with a second event record -- three more packets on the stream, ~10 us each between a frame's launches.)
// The run's clock (MCYCLE) is taken on the device without packets on the run's stream: vx_start launches vx_stamp_kernel on a stream of its
// own (clk_stream) -- it starts with the run's first launch -- and this kernel, the run's last, reads the clock again: host[2] = the stamp,
// host[3] = now, and *clock goes back to ~0 for the next run.  (Round 5's first form stamped inside the traversal kernels: three more
// spilled registers and 0.6 % of the headline for a diagnostic; profiles/r05_h_run_clock_ab.txt.)  The two kernels are not ordered by an
// event -- that would be the packet this avoids; a stamp that has not landed reads as ~0 and the run reports the host's clock.
__global__ void vx_stamp_kernel(unsigned long long* __restrict__ clock) { *clock = (unsigned long long)wall_clock64(); }
__global__ void vx_readback_kernel(unsigned long long* __restrict__ rays, const uint32_t* __restrict__ status, unsigned long long* __restrict__ clock,
                                   unsigned long long* __restrict__ host) {
  host[0] = *rays;
  host[1] = (unsigned long long)*status;
  if (clock) { host[2] = *clock; host[3] = (unsigned long long)wall_clock64(); *clock = ~0ull; }
  *rays = 0ull;
  __threadfence_system();
}

namespace {

constexpr uint64_t kUserBase = 0x10000;      // USER_BASE_ADDR (hw/VX_config.toml), runtime/simx/vortex.cpp:52
constexpr uint64_t kBlockAlign = 64;         // CACHE_BLOCK_SIZE (runtime/common/common.h:29)
constexpr uint64_t kSlotBytes = 4096;         // small-buffer slab slots
constexpr uint32_t kSlotsPerSlab = 256;
constexpr uint64_t kShadowMax = 64 * 1024;   // buffers up to this size keep a host shadow (args, sbt, kernel tags)
constexpr const char* kTagPrefix = "VXHIP1:";

#define VXLOG(...) do { std::fprintf(stderr, "[vortex-hip] " __VA_ARGS__); std::fputc('\n', stderr); } while (0)

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) & ~(a - 1); }

struct vx_device;

struct Alloc {
  uint64_t va = 0;
  uint64_t size = 0;        // requested size
  uint64_t span = 0;        // block-aligned span in the address space
  void* dptr = nullptr;     // hipMalloc'ed backing store (span bytes)
  bool reserved = false;    // created by mem_reserve (kernel images)
  bool pooled = false;      // dptr is a slot of the small-buffer slab (no hipMalloc/hipFree of its own)
  uint64_t version = 0;     // bumped by every copy_to_dev into this allocation
  bool dev_stale = false;   // the host shadow is newer than the device copy (small uploads are sent to the device only when something there reads them)
  std::vector<uint8_t> shadow;  // host copy for small buffers
};

struct vx_buffer {          // same role as callbacks.inc:14-18
  vx_device* device;
  uint64_t addr;
  uint64_t size;
};

// One more GPU behind the same vx_device (VORTEX_HIP_DEVICES=a,b,...): it keeps its own copy of the scene's buffers and its own
// acceleration layout, traces tile rows k, k+n, ... of every whole frame into a framebuffer of its own and copies them into the first
// device's output buffer, where vx_copy_from_dev finds the frame.  The unmodified reference host sees one device.
struct Helper {
  int hip_dev = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;          // behind the share's copy into the first device's framebuffer
  struct Mirror { const void* src = nullptr; uint64_t version = 0, bytes = 0; void* dptr = nullptr; } m[7];
  vxrt_accel_t* accel = nullptr;
  uint64_t key[14] = {0};
  uint32_t* fb = nullptr; uint64_t fb_bytes = 0;
  unsigned long long* d_rays = nullptr;
  unsigned long long* h_back = nullptr;   // pinned, portable: [0] rays of the share, [1] that device's status word
  // VORTEX_HIP_GATHER=rccl: the share packed into contiguous bytes on its own device, sent to the first device through RCCL
  uint8_t* wire = nullptr; uint64_t wire_bytes = 0;
  hipEvent_t packed = nullptr;        // behind the packing copies on `stream`
  int rank = 0;                       // the communicator rank of this helper's GPU (0 = it shares the first device)
};

// RCCL from the C host (north_star: "host code stays in C ... RCCL gather over xGMI only for final image assembly"): the handful of entry
// points the gather needs, bound at run time so that a backend that never sets VORTEX_HIP_GATHER=rccl does not load the 1 GB library.
struct RcclApi {
  void* so = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load() {
    if (so) return true;
    so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
    Send = (decltype(Send))dlsym(so, "ncclSend");
    Recv = (decltype(Recv))dlsym(so, "ncclRecv");
    GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};

struct vx_device {
  int hip_dev = 0;
  hipStream_t stream = nullptr;
  std::vector<Helper> helpers;
  hipEvent_t ev_scene = nullptr;      // on `stream`, behind the uploads a helper's mirror copies read
  uint32_t fanned = 0;                // devices the pending run was split over (1 = this one alone)
  uint64_t n_fanned_runs = 0;         // vx_hip_device_stat 2
  // VORTEX_HIP_GATHER=rccl: one communicator per DISTINCT listed GPU (ncclCommInitAll; rank 0 = the first device); the shares reach the
  // first device's output buffer as one group of ncclSend / ncclRecv pairs per run instead of peer copies
  RcclApi rccl;
  std::vector<int> comm_devs;         // rank -> HIP device
  std::vector<ncclComm_t> comms;      // rank -> communicator
  uint8_t* wire0 = nullptr; uint64_t wire0_bytes = 0;   // the first device's landing buffer: the shares side by side
  uint64_t n_rccl_runs = 0;           // vx_hip_device_stat 7
  bool run_pending = false;
  bool have_timing = false;
  std::map<uint64_t, Alloc> allocs;   // keyed by va
  uint64_t used = 0;
  uint64_t total_mem = 0;
  std::unordered_map<uint32_t, uint32_t> dcrs;
  unsigned long long* d_rays = nullptr;
  unsigned long long last_rays = 0;
  unsigned long long* h_back = nullptr;   // pinned: [0] rays, [1] status word of the run -- written by the stream's last kernel; [4] scratch
  static constexpr uint64_t kStageSlot = 4096; static constexpr uint32_t kStageSlots = 64;
  char* stage = nullptr; uint32_t stage_next = 0;   // pinned staging ring of the small uploads
  // small buffers (kernel_arg_t, SBT, kernel selector images: re-allocated per run by the reference host, tracer.cpp:
  // 272-281) come from slabs of kSlotBytes slots instead of one hipMalloc/hipFree each
  std::vector<void*> slabs;
  std::vector<void*> free_slots;
  uint64_t n_accel_builds = 0, n_hip_mallocs = 0;   // vx_hip_device_stat
  float last_ms = 0.f;                // the last run's duration: the device's clock when the run has one, else the host's
  float last_host_ms = 0.f;           // vx_start -> the moment ready_wait saw the stream drained (vx_hip_device_stat 6, in us)
  uint64_t n_dev_clock_runs = 0, n_host_clock_runs = 0;   // vx_hip_device_stat 4 / 5
  unsigned long long* d_clock = nullptr;   // the run's clock on the device (vx_stamp_kernel / vx_readback_kernel)
  hipStream_t clk_stream = nullptr;        // the stamp's own stream
  std::chrono::steady_clock::time_point t_begin{};   // vx_start of the pending run
  hipDeviceProp_t prop{};
  // acceleration layout of the scene last started, rebuilt only when one of the four traversal
  // buffers was re-uploaded or re-pointed (key = device pointers + upload versions)
  vxrt_accel_t* accel = nullptr;
  vxrc_accel_t* rc_accel = nullptr;   // layout of the raycast twin's scene, rebuilt when one of its buffers is re-uploaded
  uint64_t rc_key[16] = {0};
  // reference-quirks mode (DCR 0x7F4): flat image of the address space, camera rays and hit records of the frame
  void* q_image = nullptr; uint64_t q_image_size = 0; struct QKey { uint64_t va, size, version; bool operator<(const QKey& o) const { return va < o.va; } }; std::vector<QKey> q_image_key; void* q_rays = nullptr; void* q_hits = nullptr; uint64_t q_rays_cap = 0;
  uint64_t accel_key[14] = {0};
  uint64_t upload_seq = 0;
  int init() {
    const char* e = std::getenv("VORTEX_HIP_DEVICE");
    if (!e) e = std::getenv("LOCAL_RANK");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { VXLOG("no HIP device"); return -1; }
    hip_dev = e ? std::atoi(e) % n : 0;
    // VORTEX_HIP_DEVICES=0,1,2,3: the first entry is the device of the address space, the others trace a share of every whole frame
    // (an index may repeat -- two shares on one GPU -- which is how the one-GPU tests drive this path)
    std::vector<int> list;
    if (const char* l = std::getenv("VORTEX_HIP_DEVICES")) {
      for (const char* p = l; *p;) {
        char* end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) { VXLOG("VORTEX_HIP_DEVICES: expected a comma-separated list of device indices, got '%s'", l); return -1; }
        if (v < 0 || v >= n) { VXLOG("VORTEX_HIP_DEVICES: device %ld of %d", v, n); return -1; }
        list.push_back((int)v);
        p = (*end == ',') ? end + 1 : end;
        if (*end && *end != ',') { VXLOG("VORTEX_HIP_DEVICES: expected a comma-separated list of device indices, got '%s'", l); return -1; }
      }
      if (list.size() > 8) { VXLOG("VORTEX_HIP_DEVICES: at most 8 devices"); return -1; }
      if (!list.empty()) hip_dev = list[0];
    }
    if (hipSetDevice(hip_dev) != hipSuccess) return -1;
    if (hipGetDeviceProperties(&prop, hip_dev) != hipSuccess) return -1;
    total_mem = prop.totalGlobalMem;
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (hipMalloc((void**)&d_rays, sizeof(unsigned long long)) != hipSuccess || hipMemset(d_rays, 0, sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMalloc((void**)&d_clock, sizeof(unsigned long long)) != hipSuccess || hipMemset(d_clock, 0xFF, sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipStreamCreateWithFlags(&clk_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&h_back, 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&stage, kStageSlot * kStageSlots, hipHostMallocDefault) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) h_back[i] = 0;
    if (list.size() > 1) {
      if (hipEventCreateWithFlags(&ev_scene, hipEventDisableTiming) != hipSuccess) return -1;
      for (size_t k = 1; k < list.size(); ++k) {
        Helper h;
        h.hip_dev = list[k];
        if (hipSetDevice(h.hip_dev) != hipSuccess) return -1;
        if (h.hip_dev != hip_dev) {
          // both directions: the mirrors are read from the first device, the share of the frame is written to it
          hipError_t pe = hipDeviceEnablePeerAccess(hip_dev, 0);
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) VXLOG("device %d cannot address device %d directly (%s): copies between them are staged", h.hip_dev, hip_dev, hipGetErrorString(pe));
          (void)hipGetLastError();
        }
        if (hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking) != hipSuccess) return -1;
        if (hipEventCreateWithFlags(&h.done, hipEventDisableTiming) != hipSuccess) return -1;
        if (hipMalloc((void**)&h.d_rays, sizeof(unsigned long long)) != hipSuccess || hipMemset(h.d_rays, 0, sizeof(unsigned long long)) != hipSuccess) return -1;
        if (hipHostMalloc((void**)&h.h_back, 8 * sizeof(unsigned long long), hipHostMallocPortable) != hipSuccess) return -1;
        for (int i = 0; i < 8; ++i) h.h_back[i] = 0;
        helpers.push_back(h);
      }
      if (hipSetDevice(hip_dev) != hipSuccess) return -1;
      for (auto& h : helpers) if (h.hip_dev != hip_dev) {
        hipError_t pe = hipDeviceEnablePeerAccess(h.hip_dev, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
      const char* g = std::getenv("VORTEX_HIP_GATHER");
      if (g && std::strcmp(g, "rccl") == 0) {
        if (!rccl.load()) { VXLOG("VORTEX_HIP_GATHER=rccl: librccl.so.1 could not be loaded (%s)", dlerror()); return -1; }
        comm_devs.push_back(hip_dev);
        for (auto& h : helpers) {
          size_t r = 0;
          while (r < comm_devs.size() && comm_devs[r] != h.hip_dev) ++r;
          if (r == comm_devs.size()) comm_devs.push_back(h.hip_dev);
          h.rank = (int)r;
          if (hipSetDevice(h.hip_dev) != hipSuccess || hipEventCreateWithFlags(&h.packed, hipEventDisableTiming) != hipSuccess) return -1;
        }
        comms.assign(comm_devs.size(), nullptr);
        const ncclResult_t nr = rccl.CommInitAll(comms.data(), (int)comm_devs.size(), comm_devs.data());
        if (nr != ncclSuccess) { VXLOG("VORTEX_HIP_GATHER=rccl: ncclCommInitAll over %zu devices: %s", comm_devs.size(), rccl.GetErrorString(nr)); comms.clear(); return -1; }
        if (hipSetDevice(hip_dev) != hipSuccess) return -1;
      } else if (g && *g && std::strcmp(g, "copy") != 0) { VXLOG("VORTEX_HIP_GATHER: 'rccl' or 'copy', got '%s'", g); return -1; }
    }
    return 0;
  }

  void free_helper(Helper& h) {
    (void)hipSetDevice(h.hip_dev);
    if (h.stream) (void)hipStreamSynchronize(h.stream);
    if (h.accel) (void)vxrt_accel_destroy(h.accel);
    for (auto& m : h.m) if (m.dptr) (void)hipFree(m.dptr);
    if (h.fb) (void)hipFree(h.fb);
    if (h.d_rays) (void)hipFree(h.d_rays);
    if (h.h_back) (void)hipHostFree(h.h_back);
    if (h.done) (void)hipEventDestroy(h.done);
    if (h.packed) (void)hipEventDestroy(h.packed);
    if (h.wire) (void)hipFree(h.wire);
    if (h.stream) (void)hipStreamDestroy(h.stream);
    h = Helper{};
  }

  // a helper's copy of the scene (the seven buffers of `sc`, as the first device holds them now) and its layout for them;
  // `ver` = upload versions of those buffers, `bytes` = their sizes.  Runs on the helper's stream, behind ev_scene.
  int prepare_helper(Helper& h, const vxrt_scene_t& sc, const uint64_t ver[7], const uint64_t bytes[7], uint64_t fb_bytes) {
    if (hipSetDevice(h.hip_dev) != hipSuccess) return -1;
    if (hipStreamWaitEvent(h.stream, ev_scene, 0) != hipSuccess) return -1;
    const void* src[7] = {sc.tlas, sc.blas, sc.bvh, sc.tri, sc.triEx, sc.mat, sc.tex};
    bool changed = false;
    for (int i = 0; i < 7; ++i) {
      Helper::Mirror& m = h.m[i];
      if (m.src == src[i] && m.bytes == bytes[i] && m.version == ver[i]) continue;
      changed = true;
      if (m.bytes != bytes[i] || !m.dptr) {
        if (hipStreamSynchronize(h.stream) != hipSuccess) return -1;
        if (h.accel) { (void)vxrt_accel_destroy(h.accel); h.accel = nullptr; }
        if (m.dptr) (void)hipFree(m.dptr);
        m = Helper::Mirror{};
        if (src[i] && bytes[i]) {
          if (hipMalloc(&m.dptr, bytes[i]) != hipSuccess) { VXLOG("hipMalloc(%llu) on device %d failed", (unsigned long long)bytes[i], h.hip_dev); return -1; }
          ++n_hip_mallocs;
        }
      }
      if (src[i] && bytes[i]) {
        const hipError_t ce = h.hip_dev == hip_dev ? hipMemcpyAsync(m.dptr, src[i], bytes[i], hipMemcpyDeviceToDevice, h.stream)
                                                   : hipMemcpyPeerAsync(m.dptr, h.hip_dev, src[i], hip_dev, bytes[i], h.stream);
        if (ce != hipSuccess) { VXLOG("scene copy to device %d: %s", h.hip_dev, hipGetErrorString(ce)); return -1; }
      }
      m.src = src[i]; m.bytes = bytes[i]; m.version = ver[i];
    }
    if (h.fb_bytes < fb_bytes) {
      if (hipStreamSynchronize(h.stream) != hipSuccess) return -1;
      if (h.fb) (void)hipFree(h.fb);
      h.fb = nullptr; h.fb_bytes = 0;
      if (hipMalloc((void**)&h.fb, fb_bytes) != hipSuccess) return -1;
      ++n_hip_mallocs;
      h.fb_bytes = fb_bytes;
    }
    if (changed || !h.accel) {
      if (h.accel) { (void)vxrt_accel_destroy(h.accel); h.accel = nullptr; }
      vxrt_scene_t hs = sc;
      hs.tlas = h.m[0].dptr; hs.blas = h.m[1].dptr; hs.bvh = h.m[2].dptr; hs.tri = h.m[3].dptr;
      hs.triEx = h.m[4].dptr; hs.mat = h.m[5].dptr; hs.tex = h.m[6].dptr;
      ++n_accel_builds;
      if (vxrt_accel_build(&hs, h.stream, &h.accel) != 0) { VXLOG("start: device %d rejected the scene its first device accepted", h.hip_dev); return -1; }
    }
    return 0;
  }

  // share k of n: `full` whole tile rows (8 rows each) and, when the frame's last tile row is the share's and is cut short, `tail` more bytes
  struct Share { uint32_t full; size_t tail, last_off, bytes; };
  static Share share_of(uint32_t k, uint32_t n, uint32_t width, uint32_t height) {
    const size_t row = (size_t)width * 4, band = row * 8;
    const uint32_t tile_rows = (height + 7) / 8;
    const uint32_t mine = tile_rows > k ? (tile_rows - k + n - 1) / n : 0;
    Share s{mine, 0, 0, 0};
    const uint32_t last = k + (mine ? (mine - 1) * n : 0);
    if (mine && (last + 1) * 8 > height) { --s.full; s.tail = row * (height - last * 8); s.last_off = band * last; }
    s.bytes = band * s.full + s.tail;
    return s;
  }

  // VORTEX_HIP_GATHER=rccl: every helper packs its share (tile rows k, k+n, ... of its framebuffer) into contiguous bytes on its own GPU; ONE
  // group of ncclSend (helper's GPU) / ncclRecv (first GPU) pairs moves the shares; the first device's stream copies them from its landing
  // buffer into the rows of the output buffer.  Helpers on the first device's own GPU (the one-GPU tests list a device twice) send to rank 0
  // from rank 0: RCCL's self send/recv, same calls.  Sends of one communicator go to one stream (the first device's, or the first helper's on
  // that GPU), behind the packing's event.
  int gather_rccl(uint32_t n, uint32_t width, uint32_t height, uint32_t* dst) {
    const size_t row = (size_t)width * 4, band = row * 8, pitch = band * n;
    uint64_t total = 0;
    for (uint32_t k = 1; k < n; ++k) total += share_of(k, n, width, height).bytes;
    if (wire0_bytes < total) {
      if (hipSetDevice(hip_dev) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return -1;
      if (wire0) (void)hipFree(wire0);
      wire0 = nullptr; wire0_bytes = 0;
      if (hipMalloc((void**)&wire0, total) != hipSuccess) return -1;
      ++n_hip_mallocs;
      wire0_bytes = total;
    }
    std::vector<hipStream_t> send_stream(comm_devs.size(), nullptr);
    send_stream[0] = stream;
    for (uint32_t k = 1; k < n; ++k) {
      Helper& h = helpers[k - 1];
      const Share sh = share_of(k, n, width, height);
      if (hipSetDevice(h.hip_dev) != hipSuccess) return -1;
      if (h.wire_bytes < sh.bytes) {
        if (hipStreamSynchronize(h.stream) != hipSuccess) return -1;
        if (h.wire) (void)hipFree(h.wire);
        h.wire = nullptr; h.wire_bytes = 0;
        if (hipMalloc((void**)&h.wire, sh.bytes ? sh.bytes : 16) != hipSuccess) return -1;
        ++n_hip_mallocs;
        h.wire_bytes = sh.bytes ? sh.bytes : 16;
      }
      hipError_t ce = hipSuccess;
      if (sh.full) ce = hipMemcpy2DAsync(h.wire, band, (const char*)h.fb + band * k, pitch, band, sh.full, hipMemcpyDeviceToDevice, h.stream);
      if (ce == hipSuccess && sh.tail) ce = hipMemcpyAsync(h.wire + band * sh.full, (const char*)h.fb + sh.last_off, sh.tail, hipMemcpyDeviceToDevice, h.stream);
      if (ce != hipSuccess || hipEventRecord(h.packed, h.stream) != hipSuccess) { VXLOG("packing device %d's share: %s", h.hip_dev, hipGetErrorString(ce)); return -1; }
      if (!send_stream[(size_t)h.rank]) send_stream[(size_t)h.rank] = h.stream;
      if (send_stream[(size_t)h.rank] != h.stream) {      // (the sends of one communicator share a stream)
        if (hipSetDevice(comm_devs[(size_t)h.rank]) != hipSuccess || hipStreamWaitEvent(send_stream[(size_t)h.rank], h.packed, 0) != hipSuccess) return -1;
      }
    }
    ncclResult_t nr = rccl.GroupStart();
    uint64_t off = 0;
    for (uint32_t k = 1; k < n && nr == ncclSuccess; ++k) {
      Helper& h = helpers[k - 1];
      const Share sh = share_of(k, n, width, height);
      if (!sh.bytes) continue;
      (void)hipSetDevice(h.hip_dev);
      nr = rccl.Send(h.wire, sh.bytes, ncclUint8, 0, comms[(size_t)h.rank], send_stream[(size_t)h.rank]);
      (void)hipSetDevice(hip_dev);
      if (nr == ncclSuccess) nr = rccl.Recv(wire0 + off, sh.bytes, ncclUint8, h.rank, comms[0], stream);
      off += sh.bytes;
    }
    const ncclResult_t ne = rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) { VXLOG("RCCL gather of the frame's shares: %s", rccl.GetErrorString(nr)); return -1; }
    if (hipSetDevice(hip_dev) != hipSuccess) return -1;
    off = 0;
    for (uint32_t k = 1; k < n; ++k) {
      const Share sh = share_of(k, n, width, height);
      hipError_t ce = hipSuccess;
      if (sh.full) ce = hipMemcpy2DAsync((char*)dst + band * k, pitch, wire0 + off, band, band, sh.full, hipMemcpyDeviceToDevice, stream);
      if (ce == hipSuccess && sh.tail) ce = hipMemcpyAsync((char*)dst + sh.last_off, wire0 + off + band * sh.full, sh.tail, hipMemcpyDeviceToDevice, stream);
      if (ce != hipSuccess) { VXLOG("placing device %d's share: %s", helpers[k - 1].hip_dev, hipGetErrorString(ce)); return -1; }
      off += sh.bytes;
    }
    // the shares' ray counts and status words, and the events the first device's stream waits for (as with peer copies)
    uint32_t* st = vxrt_status_word_device();
    if (!st) return -1;
    for (uint32_t k = 1; k < n; ++k) {
      Helper& h = helpers[k - 1];
      if (hipSetDevice(h.hip_dev) != hipSuccess) return -1;
      h.h_back[1] = 0;
      hipLaunchKernelGGL(vx_readback_kernel, dim3(1), dim3(1), 0, h.stream, h.d_rays, (const uint32_t*)st, (unsigned long long*)nullptr, h.h_back);
      if (hipGetLastError() != hipSuccess || hipEventRecord(h.done, h.stream) != hipSuccess) return -1;
    }
    ++n_rccl_runs;
    return 0;
  }

  // tile rows k, k+n, ... of the helper's framebuffer into the same rows of the first device's output buffer, then the share's ray
  // count and that device's status word into pinned memory, then the event the first device's stream waits for
  int finish_helper(Helper& h, uint32_t k, uint32_t n, uint32_t width, uint32_t height, uint32_t* dst) {
    if (hipSetDevice(h.hip_dev) != hipSuccess) return -1;
    const size_t row = (size_t)width * 4, band = row * 8, pitch = band * n;
    const uint32_t tile_rows = (height + 7) / 8;
    const uint32_t mine = tile_rows > k ? (tile_rows - k + n - 1) / n : 0;       // tile rows k, k+n, ...
    uint32_t full = mine;
    const uint32_t last = k + (mine ? (mine - 1) * n : 0);
    const bool ragged = mine && (last + 1) * 8 > height;                         // the frame's last tile row, cut short by the frame
    if (ragged) --full;
    hipError_t ce = hipSuccess;
    if (full) ce = hipMemcpy2DAsync((char*)dst + band * k, pitch, (const char*)h.fb + band * k, pitch, band, full, hipMemcpyDeviceToDevice, h.stream);
    if (ce != hipSuccess) {
      // (no rectangle copy between these devices: one copy per tile row)
      (void)hipGetLastError();
      ce = hipSuccess;
      for (uint32_t i = 0; i < full && ce == hipSuccess; ++i) {
        const size_t off = band * (k + (size_t)i * n);
        ce = h.hip_dev == hip_dev ? hipMemcpyAsync((char*)dst + off, (const char*)h.fb + off, band, hipMemcpyDeviceToDevice, h.stream)
                                  : hipMemcpyPeerAsync((char*)dst + off, hip_dev, (const char*)h.fb + off, h.hip_dev, band, h.stream);
      }
    }
    if (ce == hipSuccess && ragged) {
      const size_t off = band * last, len = row * (height - last * 8);
      ce = h.hip_dev == hip_dev ? hipMemcpyAsync((char*)dst + off, (const char*)h.fb + off, len, hipMemcpyDeviceToDevice, h.stream)
                                : hipMemcpyPeerAsync((char*)dst + off, hip_dev, (const char*)h.fb + off, h.hip_dev, len, h.stream);
    }
    if (ce != hipSuccess) { VXLOG("copy of device %d's share of the frame: %s", h.hip_dev, hipGetErrorString(ce)); return -1; }
    uint32_t* st = vxrt_status_word_device();
    if (!st) return -1;
    h.h_back[1] = 0;
    hipLaunchKernelGGL(vx_readback_kernel, dim3(1), dim3(1), 0, h.stream, h.d_rays, (const uint32_t*)st, (unsigned long long*)nullptr, h.h_back);
    if (hipGetLastError() != hipSuccess) return -1;
    return hipEventRecord(h.done, h.stream) == hipSuccess ? 0 : -1;
  }

  ~vx_device() {
    (void)hipSetDevice(hip_dev);
    if (stream) (void)hipStreamSynchronize(stream);   // simx dtor waits for the run (vortex.cpp:69-71)
    for (auto& h : helpers) { (void)hipSetDevice(h.hip_dev); if (h.stream) (void)hipStreamSynchronize(h.stream); }
    for (ncclComm_t c : comms) if (c) (void)rccl.CommDestroy(c);
    comms.clear();
    for (auto& h : helpers) free_helper(h);
    (void)hipSetDevice(hip_dev);
    if (wire0) (void)hipFree(wire0);
    if (ev_scene) (void)hipEventDestroy(ev_scene);
    if (accel) (void)vxrt_accel_destroy(accel);
    if (rc_accel) (void)vxrc_accel_destroy(rc_accel);
    for (auto& kv : allocs) if (kv.second.dptr && !kv.second.pooled) (void)hipFree(kv.second.dptr);
    for (void* sl : slabs) (void)hipFree(sl);
    if (d_rays) (void)hipFree(d_rays);
    if (clk_stream) { (void)hipStreamSynchronize(clk_stream); (void)hipStreamDestroy(clk_stream); }
    if (d_clock) (void)hipFree(d_clock);
    if (q_image) (void)hipFree(q_image);
    if (q_rays) (void)hipFree(q_rays);
    if (q_hits) (void)hipFree(q_hits);
    if (h_back) (void)hipHostFree(h_back);
    if (stage) (void)hipHostFree(stage);
    if (stream) (void)hipStreamDestroy(stream);
  }

  // first-fit in [kUserBase, 2^48), 64-byte blocks
  bool range_free(uint64_t va, uint64_t span) const {
    auto it = allocs.upper_bound(va);
    if (it != allocs.end() && it->first < va + span) return false;
    if (it != allocs.begin()) {
      --it;
      if (it->second.va + it->second.span > va) return false;
    }
    return true;
  }

  int back(Alloc& a) {
    (void)hipSetDevice(hip_dev);
    if (a.span <= kSlotBytes) {
      if (free_slots.empty()) {
        void* slab = nullptr;
        if (hipMalloc(&slab, (size_t)kSlotBytes * kSlotsPerSlab) != hipSuccess) { VXLOG("hipMalloc(slab) failed"); return -1; }
        ++n_hip_mallocs;
        slabs.push_back(slab);
        for (uint32_t i = 0; i < kSlotsPerSlab; ++i) free_slots.push_back((char*)slab + (size_t)(kSlotsPerSlab - 1 - i) * kSlotBytes);
      }
      a.dptr = free_slots.back();
      free_slots.pop_back();
      a.pooled = true;
    } else {
      if (hipMalloc(&a.dptr, a.span) != hipSuccess) { VXLOG("hipMalloc(%llu) failed", (unsigned long long)a.span); return -1; }
      ++n_hip_mallocs;
    }
    if (a.size <= kShadowMax) a.shadow.assign(a.size, 0);
    return 0;
  }

  // does the acceleration layout of the last run point into this allocation?
  bool accel_uses(const Alloc& a) const {
    if (!accel) return false;
    const uint64_t lo = (uint64_t)a.dptr, hi = lo + a.span;
    for (int i : {0, 2, 4, 6, 8, 9, 10}) if (accel_key[i] >= lo && accel_key[i] < hi && accel_key[i] != 0) return true;
    return false;
  }

  int mem_alloc(uint64_t size, uint64_t* va_out) {
    const uint64_t span = align_up(size, kBlockAlign);
    uint64_t va = kUserBase;
    for (auto& kv : allocs) {
      if (kv.second.va >= va + span) break;
      va = std::max(va, align_up(kv.second.va + kv.second.span, kBlockAlign));
    }
    Alloc a; a.va = va; a.size = size; a.span = span;
    if (back(a) != 0) return -1;
    used += span;
    allocs.emplace(va, std::move(a));
    *va_out = va;
    return 0;
  }

  int mem_reserve(uint64_t va, uint64_t size) {
    const uint64_t span = align_up(size, kBlockAlign);
    if (!range_free(va, span)) return -1;
    Alloc a; a.va = va; a.size = size; a.span = span; a.reserved = true;
    if (back(a) != 0) return -1;
    used += span;
    allocs.emplace(va, std::move(a));
    return 0;
  }

  int mem_free(uint64_t va) {
    auto it = allocs.find(va);
    if (it == allocs.end()) return -1;
    wait_idle();
    (void)hipSetDevice(hip_dev);
    if (accel_uses(it->second)) { (void)vxrt_accel_destroy(accel); accel = nullptr; }   // it references this buffer
    if (rc_accel) {
      const uint64_t lo = (uint64_t)it->second.dptr, hi = lo + it->second.span;
      for (int i : {0, 2, 4, 6, 8, 10, 12}) if (rc_key[i] >= lo && rc_key[i] < hi && rc_key[i] != 0) { (void)vxrc_accel_destroy(rc_accel); rc_accel = nullptr; break; }
    }
    if (it->second.pooled) free_slots.push_back(it->second.dptr);
    else if (it->second.dptr) (void)hipFree(it->second.dptr);
    used -= it->second.span;
    allocs.erase(it);
    return 0;
  }

  // allocation containing [va, va+len)
  Alloc* find(uint64_t va, uint64_t len = 1) {
    auto it = allocs.upper_bound(va);
    if (it == allocs.begin()) return nullptr;
    --it;
    Alloc& a = it->second;
    if (va < a.va || va + len > a.va + a.span) return nullptr;
    return &a;
  }

  // The RTU reads its base pointers from 32-bit DCRs (tracer.cpp:252-256 truncates them).  Resolve
  // a truncated value: the 64-bit kernel_arg address when its low half agrees, else the value
  // itself, else the unique allocation whose base has these low 32 bits.
  Alloc* find_low32(uint32_t low, uint64_t hint64, uint64_t* va_out) {
    uint64_t va = ((uint32_t)hint64 == low) ? hint64 : (uint64_t)low;
    if (Alloc* a = find(va)) { *va_out = va; return a; }
    Alloc* match = nullptr;
    for (auto& kv : allocs) {
      if ((uint32_t)kv.first == low) { if (match) return nullptr; match = &kv.second; }
    }
    if (match) *va_out = match->va;
    return match;
  }

  void wait_idle() {
    if (run_pending) {
      (void)hipSetDevice(hip_dev);
      (void)hipStreamSynchronize(stream);
      finish_run();
    }
  }

  // rays counter and status word travel back in the stream, so joining a run costs no extra synchronous copy
  // (a one-thread kernel that stores both words into the pinned, device-visible h_back: two 8-byte copies through the copy engine cost
  // the frame more than the launch)
  int enqueue_readback() {
    uint32_t* st = vxrt_status_word_device();
    if (!st) return -1;
    h_back[1] = 0;
    h_back[2] = ~0ull;
    hipLaunchKernelGGL(vx_readback_kernel, dim3(1), dim3(1), 0, stream, d_rays, (const uint32_t*)st, d_clock, h_back);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }

  // the start of a run on both clocks: the host's, and the device's through a one-thread kernel on a stream of its own
  void begin_run() {
    t_begin = std::chrono::steady_clock::now();
    (void)hipSetDevice(hip_dev);
    hipLaunchKernelGGL(vx_stamp_kernel, dim3(1), dim3(1), 0, clk_stream, d_clock);
    (void)hipGetLastError();
  }

  void finish_run() {
    if (!run_pending) return;
    run_pending = false;
    last_host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); have_timing = true;
    // the device's own clock (100 MHz ticks between the stamp vx_start launched and the run's last kernel, both on the first device)
    const bool dev_clock = h_back[2] != ~0ull && h_back[3] > h_back[2];
    last_ms = dev_clock ? (float)((double)(h_back[3] - h_back[2]) * 1e-5) : last_host_ms;
    ++(dev_clock ? n_dev_clock_runs : n_host_clock_runs);
    last_rays = h_back[0];   // copied back by the stream at the end of the run (enqueue_readback)
    if (fanned > 1) for (auto& h : helpers) last_rays += h.h_back[0];   // (their streams' events were joined by `stream` before its own readback)
  }

  int upload(uint64_t va, const void* src, uint64_t size) {
    Alloc* a = find(va, size ? size : 1);
    if (!a) return -1;
    wait_idle();
    (void)hipSetDevice(hip_dev);
    const uint64_t off = va - a->va;
    const bool shadowed = !a->shadow.empty() && off + size <= a->shadow.size();
    if (size && shadowed && a->size <= kStageSlot) {
      // A small buffer with a host shadow -- the kernel arguments of every frame (tracer.cpp:262-288), the SBT, the selector images -- is
      // decoded by start() from the shadow; nothing on the device reads it unless it is one of the scene's own buffers (a 12-triangle
      // scene) or the reference-quirks image wants the address space as it is.  The device copy is therefore made when one of those
      // needs it (flush_stale), not per upload: one copy packet less between two frames of the drop-in sequence.
      a->dev_stale = true;
    } else
    if (size && size <= kStageSlot && stage) {
      // small uploads (the kernel arguments of every frame, tracer.cpp:262-288): copied into a pinned slot -- the caller's buffer is
      // free again on return, as vx_copy_to_dev promises -- and sent by the stream the runs use, ahead of the next run, without a
      // synchronous copy's round trip.  A slot is reused after kStageSlots further uploads; the stream is drained then.
      if (stage_next == kStageSlots) { if (hipStreamSynchronize(stream) != hipSuccess) return -1; stage_next = 0; }
      char* slot = stage + (size_t)stage_next++ * kStageSlot;
      std::memcpy(slot, src, size);
      if (hipMemcpyAsync((char*)a->dptr + off, slot, size, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
    } else if (size) {
      if (hipStreamSynchronize(stream) != hipSuccess) return -1;   // (behind any staged upload still in the stream)
      if (hipMemcpy((char*)a->dptr + off, src, size, hipMemcpyHostToDevice) != hipSuccess) return -1;
    }
    if (shadowed) std::memcpy(a->shadow.data() + off, src, size);
    a->version = ++upload_seq;   // unique per device: a buffer freed and allocated again at the same address never repeats a version
    return 0;
  }

  // bring the device copy of a lazily uploaded buffer up to date (through the pinned staging ring, on the run's stream)
  int flush_stale(Alloc* a) {
    if (!a || !a->dev_stale) return 0;
    const uint64_t n = std::min<uint64_t>(a->size, a->shadow.size());
    if (n > kStageSlot || !stage) {
      if (hipStreamSynchronize(stream) != hipSuccess) return -1;
      if (hipMemcpy(a->dptr, a->shadow.data(), n, hipMemcpyHostToDevice) != hipSuccess) return -1;
    } else {
      if (stage_next == kStageSlots) { if (hipStreamSynchronize(stream) != hipSuccess) return -1; stage_next = 0; }
      char* slot = stage + (size_t)stage_next++ * kStageSlot;
      std::memcpy(slot, a->shadow.data(), n);
      if (hipMemcpyAsync(a->dptr, slot, n, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
    }
    a->dev_stale = false;
    return 0;
  }
  int flush_all_stale() {
    for (auto& kv : allocs) if (kv.second.dev_stale && flush_stale(&kv.second) != 0) return -1;
    return 0;
  }

  int download(void* dst, uint64_t va, uint64_t size) {
    Alloc* a = find(va, size ? size : 1);
    if (!a) return -1;
    wait_idle();
    (void)hipSetDevice(hip_dev);
    if (flush_stale(a) != 0) return -1;
    // (The destination stays ordinary pageable memory.  Registering it with the runtime once per pointer -- so that the copy is one DMA instead of
    // the runtime's staged copy -- was built and measured in round 5: 0.1735 against 0.168 ms for the 8.3 MB frame, no gain, and a registered
    // range the host frees behind the backend's back is a hazard: removed again, profiles/r05_b_host_register_ab.txt.)
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;   // (behind any staged upload still in the stream)
    if (size && hipMemcpy(dst, (char*)a->dptr + (va - a->va), size, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
  }

  std::string tag_of(uint64_t va) {
    Alloc* a = find(va);
    if (!a || a->shadow.empty()) return {};
    const size_t off = va - a->va;
    const size_t plen = std::strlen(kTagPrefix);
    if (a->shadow.size() < off + plen || std::memcmp(a->shadow.data() + off, kTagPrefix, plen) != 0) return {};
    std::string s;
    for (size_t i = off + plen; i < a->shadow.size() && a->shadow[i]; ++i) s.push_back((char)a->shadow[i]);
    return s;
  }

  int start(uint64_t krnl_va, uint64_t args_va);
  int start_raycast(uint64_t args_va);
  int ready_wait(uint64_t timeout_ms);
};

int vx_device::start(uint64_t krnl_va, uint64_t args_va) {
  wait_idle();   // ensure prior run completed (vortex.cpp:331-333)
  (void)hipSetDevice(hip_dev);
  fanned = 1;
  dcrs[VX_DCR_BASE_STARTUP_ADDR0] = (uint32_t)krnl_va;
  dcrs[VX_DCR_BASE_STARTUP_ADDR1] = (uint32_t)(krnl_va >> 32);
  dcrs[VX_DCR_BASE_STARTUP_ARG0] = (uint32_t)args_va;
  dcrs[VX_DCR_BASE_STARTUP_ARG1] = (uint32_t)(args_va >> 32);

  const std::string tag = tag_of(krnl_va);
  if (tag == "raycast.kernel") return start_raycast(args_va);   // the software twin (tests/regression/raycast)
  if (tag != "raytracing.kernel") {
    VXLOG("start: kernel image at 0x%llx is not a HIP kernel selector (tag '%s'); this backend cannot run RISC-V binaries",
          (unsigned long long)krnl_va, tag.c_str());
    return -1;
  }
  Alloc* aa = find(args_va, sizeof(vx_rt_kernel_arg_t));
  if (!aa || aa->shadow.size() < (args_va - aa->va) + sizeof(vx_rt_kernel_arg_t)) { VXLOG("start: bad kernel_arg buffer"); return -1; }
  vx_rt_kernel_arg_t ka;
  std::memcpy(&ka, aa->shadow.data() + (args_va - aa->va), sizeof ka);

  // shader binding table (tracer.cpp:244-250): [0]=miss [1]=closest [3]=anyhit
  {
    Alloc* sb = find(ka.sbt_addr, 32);
    if (!sb || sb->shadow.size() < (ka.sbt_addr - sb->va) + 32) { VXLOG("start: sbt_addr does not name a buffer of 4 entries"); return -1; }
    uint64_t sbt[4];
    std::memcpy(sbt, sb->shadow.data() + (ka.sbt_addr - sb->va), sizeof sbt);
    if (tag_of(sbt[0]) != "raytracing.miss" || tag_of(sbt[1]) != "raytracing.closest" || tag_of(sbt[3]) != "raytracing.anyhit") {
      VXLOG("start: shader binding table does not reference the HIP miss/closest/anyhit selectors");
      return -1;
    }
  }
  if (ka.samples_per_pixel == 0) { VXLOG("start: samples_per_pixel == 0 leaves pixels undefined in the reference (kernel.cpp:67-80)"); return -1; }

  // traversal buffers come from the RTU DCRs (rt_traversal.cpp:33-36), shading buffers from kernel_arg_t
  auto dcr = [&](uint32_t id, uint32_t* v) { auto it = dcrs.find(id); if (it == dcrs.end()) return false; *v = it->second; return true; };
  uint32_t d_tlas, d_blas, d_bvh, d_tri;
  if (!dcr(VX_DCR_BASE_RTX_TLAS_PTR, &d_tlas) || !dcr(VX_DCR_BASE_RTX_BLAS_PTR, &d_blas) ||
      !dcr(VX_DCR_BASE_RTX_BVH_PTR, &d_bvh) || !dcr(VX_DCR_BASE_RTX_TRI_PTR, &d_tri)) {
    VXLOG("start: RTX DCRs 0x6..0x9 not written");
    return -1;
  }
  struct Res { Alloc* a; uint64_t off; };
  auto res32 = [&](uint32_t low, uint64_t hint) -> Res {
    uint64_t va = 0;
    Alloc* a = find_low32(low, hint, &va);
    return {a, a ? va - a->va : 0};
  };
  auto res64 = [&](uint64_t va) -> Res { Alloc* a = find(va); return {a, a ? va - a->va : 0}; };
  Res r_tlas = res32(d_tlas, ka.tlas_addr), r_blas = res32(d_blas, ka.blas_addr);
  Res r_bvh = res32(d_bvh, ka.qBvh_addr), r_tri = res32(d_tri, ka.tri_addr);
  Res r_triex = res64(ka.triEx_addr), r_mat = res64(ka.mat_addr), r_tex = res64(ka.tex_addr), r_dst = res64(ka.dst_addr);
  if (!r_tlas.a || !r_blas.a || !r_bvh.a || !r_tri.a || !r_triex.a || !r_mat.a || !r_dst.a) {
    VXLOG("start: a scene/output address does not name device memory");
    return -1;
  }
  auto ptr = [](Res r) { return (const void*)((const char*)r.a->dptr + r.off); };
  auto count = [](Res r, uint64_t stride) { return (uint64_t)((r.a->size - r.off) / stride); };
  // the scene's own buffers are read on the device: small ones uploaded lazily get their device copy now
  for (Alloc* al : {r_tlas.a, r_blas.a, r_bvh.a, r_tri.a, r_triex.a, r_mat.a, r_tex.a})
    if (flush_stale(al) != 0) return -1;
  // ... and so does the output buffer: a host that cleared a small framebuffer (<= 4 KB: 32x32 pixels) before the run left a
  // lazy upload behind, which must reach the device BEFORE the kernels write pixels -- flushed later (by vx_copy_from_dev) it
  // would put the host's old bytes over the frame
  if (flush_stale(r_dst.a) != 0) return -1;
  vxrt_scene_t sc{};
  sc.tlas = ptr(r_tlas); sc.blas = ptr(r_blas); sc.bvh = ptr(r_bvh); sc.tri = ptr(r_tri);
  sc.triEx = ptr(r_triex); sc.mat = ptr(r_mat); sc.tex = r_tex.a ? ptr(r_tex) : nullptr;
  sc.n_tlas_nodes = (uint32_t)std::min<uint64_t>(count(r_tlas, RT_NODE_BYTES), 0x7fffffff);
  sc.n_blas = (uint32_t)std::min<uint64_t>(count(r_blas, RT_BLAS_STRIDE), 0x7fffffff);
  sc.n_bvh_nodes = (uint32_t)std::min<uint64_t>(count(r_bvh, RT_NODE_BYTES), 0x7fffffff);
  sc.n_tris = (uint32_t)std::min<uint64_t>(count(r_tri, RT_TRI_BYTES), 0x7fffffff);
  sc.n_mats = (uint32_t)std::min<uint64_t>(count(r_mat, RT_MAT_BYTES), 0x7fffffff);
  sc.tex_bytes = r_tex.a ? r_tex.a->size - r_tex.off : 0;
  if ((uint64_t)ka.dst_width * ka.dst_height * 4 > r_dst.a->size - r_dst.off) { VXLOG("start: output buffer too small"); return -1; }
  if (count(r_triex, RT_TRIEX_BYTES) < sc.n_tris) { VXLOG("start: triEx buffer smaller than tri buffer"); return -1; }

  vxrt_shade_params_t sp{};
  for (int i = 0; i < 3; ++i) {
    sp.ambient[i] = ka.ambient_color[i]; sp.light_color[i] = ka.light_color[i];
    sp.light_pos[i] = ka.light_pos[i]; sp.background[i] = ka.background_color[i];
  }
  sp.max_depth = ka.max_depth;
  uint32_t y0 = 0, y1 = 0, shadow = 0, row_stride = 0;
  dcr(VX_DCR_HIP_ROW_BEGIN, &y0);
  dcr(VX_DCR_HIP_ROW_END, &y1);
  dcr(VX_DCR_HIP_SHADOW_RAYS, &shadow);
  dcr(VX_DCR_HIP_ROW_STRIDE, &row_stride);
  if (y1 == 0 || y1 > ka.dst_height) y1 = ka.dst_height;
  if (y0 > y1) y0 = y1;
  if (row_stride > 1 && ((y0 & 7u) != 0 || y0 / 8u >= row_stride)) { VXLOG("start: DCR 0x7F3 (tile-row stride) needs ROW_BEGIN = 8 * phase with phase < stride"); return -1; }

  // (triEx / mat versions: the build also validates the indices shading follows, so a re-upload of those rebuilds too)
  const uint64_t key[14] = {(uint64_t)sc.tlas, r_tlas.a->version, (uint64_t)sc.blas, r_blas.a->version, (uint64_t)sc.bvh, r_bvh.a->version,
                            (uint64_t)sc.tri, r_tri.a->version, (uint64_t)sc.triEx, (uint64_t)sc.mat, (uint64_t)sc.tex,
                            ((uint64_t)sc.n_bvh_nodes << 32) | sc.n_tris, r_triex.a->version, r_mat.a->version ^ (sc.tex_bytes << 20)};
  if (!accel || std::memcmp(key, accel_key, sizeof key) != 0) {
    if (accel) { (void)vxrt_accel_destroy(accel); accel = nullptr; }
    ++n_accel_builds;
    if (vxrt_accel_build(&sc, stream, &accel) != 0) { VXLOG("start: scene rejected (malformed BVH: index out of range, wrong node kind or child not after parent)"); return -1; }
    std::memcpy(accel_key, key, sizeof key);
  }

  uint32_t* dstp = (uint32_t*)((char*)r_dst.a->dptr + r_dst.off);
  uint32_t quirks = 0;
  dcr(VX_DCR_HIP_REFERENCE_QUIRKS, &quirks);
  if (quirks) {
    // reference-quirks mode: the frame's camera rays through the literal restatement of the RTU on a flat image of the address space
    if (shadow || row_stride > 1 || ka.max_depth > 1) { VXLOG("start: reference-quirks mode renders closest-hit frames only (no shadow extension, row stride or mirror bounce)"); return -1; }
    if (flush_all_stale() != 0) return -1;   // (the image is the address space as the device holds it)
    // the image mirrors the address space: every allocation at the address vx_mem_address reported.  It is kept between runs and only
    // what changed is copied again -- the list of (address, size, upload version) entries, in address order, is compared entry by entry (no
    // rolling hash that could alias): the same set of allocations -> only the re-uploaded ones are copied (every frame re-uploads its
    // kernel arguments: 216 bytes, not the 4 GiB the whole image may span); another set -> cleared and rebuilt.
    uint64_t hi = 0;
    std::vector<QKey> ver;
    for (auto& kv : allocs) if (!kv.second.reserved && kv.first < 0x80000000ull) {
      hi = std::max(hi, kv.second.va + kv.second.span);
      ver.push_back(QKey{kv.first, kv.second.size, kv.second.version});
    }
    std::sort(ver.begin(), ver.end());
    if (hi == 0 || hi > 0xFFFFFFFFull) { VXLOG("start: reference-quirks mode needs the scene below 4 GiB of device address space"); return -1; }
    if (q_image_size < hi) {
      if (q_image) (void)hipFree(q_image);
      q_image = nullptr; q_image_size = 0; q_image_key.clear();
      if (hipMalloc(&q_image, hi) != hipSuccess) return -1;
      q_image_size = hi;
    }
    bool same_set = q_image_key.size() == ver.size();
    for (size_t i = 0; same_set && i < ver.size(); ++i) same_set = q_image_key[i].va == ver[i].va && q_image_key[i].size == ver[i].size;
    if (!same_set) {
      q_image_key.clear();
      if (hipMemsetAsync(q_image, 0, q_image_size, stream) != hipSuccess) return -1;
    }
    for (size_t i = 0; i < ver.size(); ++i) {
      if (same_set && q_image_key[i].version == ver[i].version) continue;
      const Alloc& al = allocs[ver[i].va];
      if (!al.dptr) continue;
      if (hipMemcpyAsync((char*)q_image + al.va, al.dptr, al.size, hipMemcpyDeviceToDevice, stream) != hipSuccess) { q_image_key.clear(); return -1; }
    }
    q_image_key = ver;
    const uint64_t nr = (uint64_t)ka.dst_width * (y1 - y0);
    if (q_rays_cap < nr) {
      if (q_rays) (void)hipFree(q_rays);
      if (q_hits) (void)hipFree(q_hits);
      q_rays = q_hits = nullptr; q_rays_cap = 0;
      if (hipMalloc(&q_rays, nr * 24) != hipSuccess || hipMalloc(&q_hits, nr * 24) != hipSuccess) return -1;
      q_rays_cap = nr;
    }
    begin_run();
    int rc = vxrt_camera_rays(ka.dst_width, ka.dst_height, y0, y1, (float*)q_rays, stream);
    // (the DCRs hold 32-bit device addresses: offsets into the image, as they are addresses into the simulator's RAM)
    if (rc == 0) rc = vxrt_trace_reference_quirks(q_image, q_image_size, d_tlas, d_blas, d_bvh, d_tri, (const float*)q_rays, nr, nullptr, (vxrt_hit_t*)q_hits, VXRT_MODE_CLOSEST, stream);
    if (rc == 0) rc = vxrt_shade_rays(accel, (const float*)q_rays, (const vxrt_hit_t*)q_hits, nr, &sp, nullptr, dstp + (size_t)y0 * ka.dst_width, stream);
    h_back[4] = nr;   // (pinned; the stream reads it after this call returns)
    if (rc == 0 && hipMemcpyAsync(d_rays, &h_back[4], sizeof(unsigned long long), hipMemcpyHostToDevice, stream) != hipSuccess) rc = -1;
    if (rc != 0) { VXLOG("start: reference-quirks launch rejected"); (void)hipMemsetAsync(d_rays, 0, sizeof(unsigned long long), stream); (void)hipMemsetAsync(d_clock, 0xFF, sizeof(unsigned long long), stream); return -1; }
    if (enqueue_readback() != 0) return -1;
    run_pending = true;
    return 0;
  }
  begin_run();
  // samples_per_pixel: the reference's kernel traces the SAME camera ray that many times into the same payload (kernel.cpp:67-80: GenerateRay
  // takes no sample index, the colour accumulation is commented out), so the pixel is the one sample's -- and the run costs spp times the rays
  // and the time.  Honoured as written: the frame is traced spp times (same pixels; MINSTRET and MCYCLE are what a `-s 4` run expects).
  // The samples of ONE vx_start are one workload, so they go out as one set of launches (the machinery of vxrt_render_batch on identical
  // frames that all land on the same pixels -- frame stride 0; every sample is traced AND shaded, the last writer of a pixel stores the value
  // the first one did): the launch's ramp and tail are paid once per run instead of once per sample.  Sets of at most VXRT_MAX_BATCH samples;
  // the mirror arm (max_depth > 1) keeps one frame per set of launches (its bounce levels are per frame).
  int rc = 0;
  std::vector<vxrt_shade_params_t> spv;
  auto samples = [&](auto&& one, auto&& many) {   // one(): a single sample's launches; many(params, n): n samples in one set
    uint32_t left = ka.samples_per_pixel;
    if (ka.max_depth > 1 || left == 1) { for (; left && rc == 0; --left) rc = one(); return; }
    spv.assign(std::min<uint32_t>(left, VXRT_MAX_BATCH), sp);
    while (left && rc == 0) {
      const uint32_t n = std::min<uint32_t>(left, VXRT_MAX_BATCH);
      rc = n == 1 ? one() : many(spv.data(), n);
      left -= n;
    }
  };
  const uint32_t n_dev = (uint32_t)helpers.size() + 1;
  if (n_dev > 1 && row_stride <= 1 && y0 == 0 && y1 == ka.dst_height && (ka.dst_height + 7) / 8 >= n_dev) {
    // VORTEX_HIP_DEVICES: a whole frame is split by interleaved 8-row tile rows over the listed devices (the split bench.py's ranks use);
    // a run the host already restricted (DCR 0x7F0-0x7F3) stays on the first device
    const uint64_t ver[7] = {r_tlas.a->version, r_blas.a->version, r_bvh.a->version, r_tri.a->version, r_triex.a->version, r_mat.a->version, r_tex.a ? r_tex.a->version : 0};
    const uint64_t bytes[7] = {r_tlas.a->size - r_tlas.off, r_blas.a->size - r_blas.off, r_bvh.a->size - r_bvh.off, r_tri.a->size - r_tri.off,
                               r_triex.a->size - r_triex.off, r_mat.a->size - r_mat.off, r_tex.a ? r_tex.a->size - r_tex.off : 0};
    if (hipEventRecord(ev_scene, stream) != hipSuccess) return -1;
    for (auto& h : helpers)
      if (prepare_helper(h, sc, ver, bytes, (uint64_t)ka.dst_width * ka.dst_height * 4) != 0) { (void)hipSetDevice(hip_dev); return -1; }
    begin_run();
    for (uint32_t k = 0; k < n_dev && rc == 0; ++k) {
      vxrt_accel_t* ak = k ? helpers[k - 1].accel : accel;
      uint32_t* fb = k ? helpers[k - 1].fb : dstp;
      unsigned long long* cnt = k ? helpers[k - 1].d_rays : d_rays;
      hipStream_t sk = k ? helpers[k - 1].stream : stream;
      (void)hipSetDevice(k ? helpers[k - 1].hip_dev : hip_dev);
      samples([&] { return vxrt_render_interleaved(ak, ka.dst_width, ka.dst_height, k, n_dev, &sp, (int)shadow, fb, nullptr, nullptr, cnt, sk); },
              [&](const vxrt_shade_params_t* pv, uint32_t n) { return vxrt_render_interleaved_batch(ak, ka.dst_width, ka.dst_height, k, n_dev, n, pv, (int)shadow, fb, 0, cnt, sk); });
    }
    if (!comms.empty()) { if (rc == 0) rc = gather_rccl(n_dev, ka.dst_width, ka.dst_height, dstp); }
    else for (uint32_t k = 1; k < n_dev && rc == 0; ++k) rc = finish_helper(helpers[k - 1], k, n_dev, ka.dst_width, ka.dst_height, dstp);
    (void)hipSetDevice(hip_dev);
    for (auto& h : helpers) if (rc == 0 && hipStreamWaitEvent(stream, h.done, 0) != hipSuccess) rc = -1;
    if (rc != 0) {
      VXLOG("start: launch on %u devices rejected", n_dev);
      for (auto& h : helpers) { (void)hipSetDevice(h.hip_dev); (void)hipStreamSynchronize(h.stream); (void)hipMemsetAsync(h.d_rays, 0, sizeof(unsigned long long), h.stream); }
      (void)hipSetDevice(hip_dev);
      (void)hipMemsetAsync(d_rays, 0, sizeof(unsigned long long), stream); (void)hipMemsetAsync(d_clock, 0xFF, sizeof(unsigned long long), stream);
      return -1;
    }
    if (enqueue_readback() != 0) return -1;
    fanned = n_dev;
    ++n_fanned_runs;
    run_pending = true;
    return 0;
  }
  if (row_stride > 1)
    samples([&] { return vxrt_render_interleaved(accel, ka.dst_width, ka.dst_height, y0 / 8u, row_stride, &sp, (int)shadow, dstp, nullptr, nullptr, d_rays, stream); },
            [&](const vxrt_shade_params_t* pv, uint32_t n) { return vxrt_render_interleaved_batch(accel, ka.dst_width, ka.dst_height, y0 / 8u, row_stride, n, pv, (int)shadow, dstp, 0, d_rays, stream); });
  else
    samples([&] { return vxrt_render(accel, ka.dst_width, ka.dst_height, y0, y1, &sp, (int)shadow, dstp, nullptr, nullptr, d_rays, stream); },
            [&](const vxrt_shade_params_t* pv, uint32_t n) { return vxrt_render_rows_batch(accel, ka.dst_width, ka.dst_height, y0, y1, n, pv, (int)shadow, dstp, 0, d_rays, stream); });
  if (rc != 0) { VXLOG("start: launch rejected (shape check)"); (void)hipMemsetAsync(d_rays, 0, sizeof(unsigned long long), stream); (void)hipMemsetAsync(d_clock, 0xFF, sizeof(unsigned long long), stream); return -1; }
  if (enqueue_readback() != 0) return -1;
  run_pending = true;
  return 0;
}

// tests/regression/raycast: kernel_arg_t of raycast/common.h:126-150 (192 bytes; offsets checked against the
// reference header with offsetof), all buffers addressed through it (tracer.cpp:107-150), no DCRs, no SBT.
#pragma pack(push, 1)
struct rc_kernel_arg_t {
  uint32_t dst_width, dst_height; uint64_t dst_addr;
  uint64_t tri_addr, triEx_addr, triIdx_addr, tex_addr, bvh_addr, blas_addr, tlas_addr;
  uint32_t tlas_root;
  float camera_pos[3], camera_forward[3], camera_right[3], camera_up[3], viewplane[2];
  uint32_t samples_per_pixel, max_depth;
  float light_pos[3], light_color[3], ambient_color[3], background_color[3];
  uint32_t _pad;
};
#pragma pack(pop)
static_assert(sizeof(rc_kernel_arg_t) == 192, "raycast kernel_arg_t layout");

int vx_device::start_raycast(uint64_t args_va) {
  Alloc* aa = find(args_va, sizeof(rc_kernel_arg_t));
  if (!aa || aa->shadow.size() < (args_va - aa->va) + sizeof(rc_kernel_arg_t)) { VXLOG("start: bad raycast kernel_arg buffer"); return -1; }
  rc_kernel_arg_t ka;
  std::memcpy(&ka, aa->shadow.data() + (args_va - aa->va), sizeof ka);
  if (ka.samples_per_pixel == 0) { VXLOG("start: samples_per_pixel == 0"); return -1; }
  struct Res { Alloc* a; uint64_t off; };
  auto res64 = [&](uint64_t va) -> Res { Alloc* a = find(va); return {a, a ? va - a->va : 0}; };
  Res r_tri = res64(ka.tri_addr), r_triex = res64(ka.triEx_addr), r_idx = res64(ka.triIdx_addr), r_tex = res64(ka.tex_addr);
  Res r_bvh = res64(ka.bvh_addr), r_blas = res64(ka.blas_addr), r_tlas = res64(ka.tlas_addr), r_dst = res64(ka.dst_addr);
  if (!r_tri.a || !r_triex.a || !r_idx.a || !r_tex.a || !r_bvh.a || !r_blas.a || !r_tlas.a || !r_dst.a) {
    VXLOG("start: a raycast scene/output address does not name device memory");
    return -1;
  }
  auto ptr = [](Res r) { return (const void*)((const char*)r.a->dptr + r.off); };
  auto count = [](Res r, uint64_t stride) { return (uint32_t)std::min<uint64_t>((r.a->size - r.off) / stride, 0x7fffffff); };
  for (Alloc* al : {r_tlas.a, r_blas.a, r_bvh.a, r_tri.a, r_triex.a, r_idx.a, r_tex.a, r_dst.a})      // (small scene buffers uploaded lazily: see upload(); the output buffer: see start())
    if (flush_stale(al) != 0) return -1;
  vxrc_scene_t sc{};
  sc.tlas = ptr(r_tlas); sc.blas = ptr(r_blas); sc.bvh = ptr(r_bvh); sc.tri = ptr(r_tri); sc.triEx = ptr(r_triex);
  sc.triIdx = ptr(r_idx); sc.tex = ptr(r_tex);
  sc.n_tlas_nodes = count(r_tlas, 32); sc.n_blas = count(r_blas, 160); sc.n_bvh_nodes = count(r_bvh, 32);
  sc.n_tris = std::min(count(r_tri, 36), count(r_triex, 60)); sc.n_tri_idx = count(r_idx, 4);
  sc.tex_bytes = r_tex.a->size - r_tex.off;
  sc.tlas_root = ka.tlas_root;
  if ((uint64_t)ka.dst_width * ka.dst_height * 4 > r_dst.a->size - r_dst.off) { VXLOG("start: output buffer too small"); return -1; }
  vxrc_params_t pr{};
  for (int i = 0; i < 3; ++i) {
    pr.camera_pos[i] = ka.camera_pos[i]; pr.camera_forward[i] = ka.camera_forward[i]; pr.camera_right[i] = ka.camera_right[i];
    pr.camera_up[i] = ka.camera_up[i]; pr.light_pos[i] = ka.light_pos[i]; pr.light_color[i] = ka.light_color[i];
    pr.ambient_color[i] = ka.ambient_color[i]; pr.background_color[i] = ka.background_color[i];
  }
  pr.viewplane[0] = ka.viewplane[0]; pr.viewplane[1] = ka.viewplane[1];
  pr.samples_per_pixel = ka.samples_per_pixel; pr.max_depth = ka.max_depth;
  uint32_t y0 = 0, y1 = 0;
  auto dcr = [&](uint32_t id, uint32_t* v) { auto it = dcrs.find(id); if (it == dcrs.end()) return false; *v = it->second; return true; };
  dcr(VX_DCR_HIP_ROW_BEGIN, &y0);
  dcr(VX_DCR_HIP_ROW_END, &y1);
  if (y1 == 0 || y1 > ka.dst_height) y1 = ka.dst_height;
  if (y0 > y1) y0 = y1;
  // (the layout build -- a stream synchronisation, allocations, three kernels, a copy back -- comes BEFORE the run's clock starts,
  // as in start(): the cycles mpm_query reports are the frame's)
  const uint64_t key[16] = {(uint64_t)sc.tlas, r_tlas.a->version, (uint64_t)sc.blas, r_blas.a->version, (uint64_t)sc.bvh, r_bvh.a->version,
                            (uint64_t)sc.tri, r_tri.a->version, (uint64_t)sc.triEx, r_triex.a->version, (uint64_t)sc.triIdx, r_idx.a->version,
                            (uint64_t)sc.tex, r_tex.a->version, ((uint64_t)sc.n_bvh_nodes << 32) | sc.n_tri_idx, ((uint64_t)sc.tlas_root << 32) | sc.n_tris};
  if (!rc_accel || std::memcmp(key, rc_key, sizeof key) != 0) {
    if (rc_accel) { (void)vxrc_accel_destroy(rc_accel); rc_accel = nullptr; }
    ++n_accel_builds;
    if (vxrc_accel_build(&sc, stream, &rc_accel) != 0) { VXLOG("start: raycast scene rejected (malformed BVH2: child / triangle index out of range or child not after parent)"); return -1; }
    std::memcpy(rc_key, key, sizeof key);
  }
  begin_run();
  const int rc = vxrc_render_accel(rc_accel, ka.dst_width, ka.dst_height, y0, y1, &pr, (uint32_t*)((char*)r_dst.a->dptr + r_dst.off), nullptr, stream);
  if (rc != 0) { VXLOG("start: raycast launch rejected (shape check)"); return -1; }
  if (enqueue_readback() != 0) return -1;
  run_pending = true;
  return 0;
}

int vx_device::ready_wait(uint64_t timeout_ms) {
  if (!run_pending) return 0;   // vortex.cpp:351-352
  (void)hipSetDevice(hip_dev);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { VXLOG("ready_wait: %s", hipGetErrorString(q)); return -1; }
    const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    if ((uint64_t)us >= timeout_ms * 1000ull) return -1;
    if (us > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));   // frames take well under 2 ms: poll those without sleeping
  }
  const uint32_t was_fanned = fanned;
  finish_run();
  uint32_t st = (uint32_t)h_back[1];
  if (was_fanned > 1) for (auto& h : helpers) if (h.h_back[1]) {
    st |= (uint32_t)h.h_back[1];
    uint32_t cleared = 0;
    (void)hipSetDevice(h.hip_dev);
    (void)vxrt_status(h.stream, &cleared);
    (void)hipSetDevice(hip_dev);
  }
  if (st != 0) {
    uint32_t cleared = 0;
    (void)vxrt_status(stream, &cleared);   // read-and-clear, so that the next run starts clean
    VXLOG("kernel reported status 0x%x (bit 0: traversal stack overflow, BVH deeper than %d levels; bit 2: index outside the scene buffers)", st, RT_MAX_LEVELS);
    return -1;
  }
  return 0;
}

}  // namespace

extern "C" int vx_hip_buffer_device_ptr(vx_buffer_h hbuffer, void** dev_ptr) {
  if (!hbuffer || !dev_ptr) return -1;
  auto b = (vx_buffer*)hbuffer;
  Alloc* a = b->device->find(b->addr);
  if (!a) return -1;
  *dev_ptr = (char*)a->dptr + (b->addr - a->va);
  return 0;
}

extern "C" int vx_hip_device_stat(vx_device_h hdevice, uint32_t which, uint64_t* value) {
  if (!hdevice || !value) return -1;
  auto d = (vx_device*)hdevice;
  switch (which) {
  case 0: *value = d->n_accel_builds; return 0;
  case 1: *value = d->n_hip_mallocs; return 0;
  case 2: *value = d->n_fanned_runs; return 0;          // runs split over the devices of VORTEX_HIP_DEVICES
  case 3: *value = d->helpers.size() + 1; return 0;     // devices behind this vx_device
  case 4: *value = d->n_dev_clock_runs; return 0;       // joined runs whose MCYCLE came from the device's clock
  case 5: *value = d->n_host_clock_runs; return 0;      // ... from the host's (the stamp had not landed: never observed)
  case 6: *value = (uint64_t)((double)d->last_host_ms * 1e3); return 0;   // the last joined run on the HOST's clock, microseconds
  case 7: *value = d->n_rccl_runs; return 0;            // runs whose shares were gathered through RCCL (VORTEX_HIP_GATHER=rccl)
  }
  return -1;
}

extern "C" int vx_dev_init(callbacks_t* cb) {
  if (!cb) return -1;

  cb->dev_open = [](vx_device_h* hdevice) -> int {
    if (!hdevice) return -1;
    auto d = new (std::nothrow) vx_device();
    if (!d) return -1;
    if (d->init() != 0) { delete d; return -1; }
    *hdevice = d;
    return 0;
  };

  cb->dev_close = [](vx_device_h hdevice) -> int {
    if (!hdevice) return -1;
    delete (vx_device*)hdevice;
    return 0;
  };

  cb->dev_caps = [](vx_device_h hdevice, uint32_t caps_id, uint64_t* value) -> int {
    if (!hdevice || !value) return -1;
    auto d = (vx_device*)hdevice;
    switch (caps_id) {
    case VX_CAPS_VERSION: *value = 0x950; break;                                   // gfx950
    case VX_CAPS_NUM_THREADS: *value = 64; break;                                  // lanes per wavefront
    case VX_CAPS_NUM_WARPS: *value = (uint64_t)d->prop.maxThreadsPerMultiProcessor / 64; break;
    case VX_CAPS_NUM_CORES: *value = (uint64_t)d->prop.multiProcessorCount; break; // CUs
    case VX_CAPS_CACHE_LINE_SIZE: *value = 128; break;
    case VX_CAPS_GLOBAL_MEM_SIZE: *value = d->total_mem; break;
    case VX_CAPS_LOCAL_MEM_SIZE: *value = (uint64_t)d->prop.sharedMemPerBlock; break;
    case VX_CAPS_ISA_FLAGS: *value = 0; break;                                     // no RISC-V ISA: every VX_ISA_* bit clear
    case VX_CAPS_NUM_MEM_BANKS: *value = 8; break;                                 // HBM3E stacks
    case VX_CAPS_MEM_BANK_SIZE: *value = d->total_mem / 8; break;
    case VX_CAPS_NUM_CLUSTERS: *value = 8; break;                                  // XCDs
    case VX_CAPS_SOCKET_SIZE: *value = 1; break;
    case VX_CAPS_ISSUE_WIDTH: *value = 4; break;                                   // SIMDs per CU
    case VX_CAPS_CLOCK_RATE: *value = (uint64_t)d->prop.clockRate * 1000ull; break; // kHz -> Hz
    case VX_CAPS_PEAK_MEM_BW: *value = 8000000; break;                             // MB/s, HBM3E spec
    default: VXLOG("invalid caps id: %u", caps_id); return -1;
    }
    return 0;
  };

  cb->mem_alloc = [](vx_device_h hdevice, uint64_t size, int, vx_buffer_h* hbuffer) -> int {
    if (!hdevice || !hbuffer || size == 0) return -1;        // callbacks.inc:61-65
    auto d = (vx_device*)hdevice;
    uint64_t va;
    if (d->mem_alloc(size, &va) != 0) return -1;
    *hbuffer = new vx_buffer{d, va, size};
    return 0;
  };

  cb->mem_reserve = [](vx_device_h hdevice, uint64_t address, uint64_t size, int, vx_buffer_h* hbuffer) -> int {
    if (!hdevice || !hbuffer || size == 0) return -1;
    auto d = (vx_device*)hdevice;
    if (d->mem_reserve(address, size) != 0) return -1;
    *hbuffer = new vx_buffer{d, address, size};
    return 0;
  };

  cb->mem_free = [](vx_buffer_h hbuffer) -> int {
    if (!hbuffer) return 0;                                   // callbacks.inc:100-102
    auto b = (vx_buffer*)hbuffer;
    int err = b->device->mem_free(b->addr);
    delete b;
    return err;
  };

  cb->mem_access = [](vx_buffer_h hbuffer, uint64_t offset, uint64_t size, int) -> int {
    if (!hbuffer) return -1;
    auto b = (vx_buffer*)hbuffer;
    if (offset + size > b->size) return -1;
    return 0;   // HBM has no per-range ACLs; bounds are still enforced here
  };

  cb->mem_address = [](vx_buffer_h hbuffer, uint64_t* address) -> int {
    if (!hbuffer || !address) return -1;
    *address = ((vx_buffer*)hbuffer)->addr;
    return 0;
  };

  cb->mem_info = [](vx_device_h hdevice, uint64_t* mem_free, uint64_t* mem_used) -> int {
    if (!hdevice) return -1;
    auto d = (vx_device*)hdevice;
    if (mem_free) *mem_free = d->total_mem > d->used ? d->total_mem - d->used : 0;
    if (mem_used) *mem_used = d->used;
    return 0;
  };

  cb->copy_to_dev = [](vx_buffer_h hbuffer, const void* host_ptr, uint64_t dst_offset, uint64_t size) -> int {
    if (!hbuffer || !host_ptr) return -1;
    auto b = (vx_buffer*)hbuffer;
    if (dst_offset + size > b->size) return -1;               // callbacks.inc:153-154
    return b->device->upload(b->addr + dst_offset, host_ptr, size);
  };

  cb->copy_from_dev = [](void* host_ptr, vx_buffer_h hbuffer, uint64_t src_offset, uint64_t size) -> int {
    if (!hbuffer || !host_ptr) return -1;
    auto b = (vx_buffer*)hbuffer;
    if (src_offset + size > b->size) return -1;
    return b->device->download(host_ptr, b->addr + src_offset, size);
  };

  cb->start = [](vx_device_h hdevice, vx_buffer_h hkernel, vx_buffer_h harguments) -> int {
    if (!hdevice || !hkernel || !harguments) return -1;
    return ((vx_device*)hdevice)->start(((vx_buffer*)hkernel)->addr, ((vx_buffer*)harguments)->addr);
  };

  cb->ready_wait = [](vx_device_h hdevice, uint64_t timeout) -> int {
    if (!hdevice) return -1;
    return ((vx_device*)hdevice)->ready_wait(timeout);
  };

  cb->dcr_read = [](vx_device_h hdevice, uint32_t addr, uint32_t* value) -> int {
    if (!hdevice || !value) return -1;
    auto d = (vx_device*)hdevice;
    auto it = d->dcrs.find(addr);
    if (it == d->dcrs.end()) return -1;                       // DeviceConfig::read, common.h:60-66
    *value = it->second;
    return 0;
  };

  cb->dcr_write = [](vx_device_h hdevice, uint32_t addr, uint32_t value) -> int {
    if (!hdevice) return -1;
    auto d = (vx_device*)hdevice;
    d->wait_idle();                                           // vortex.cpp:367-369
    d->dcrs[addr] = value;
    return 0;
  };

  cb->mpm_query = [](vx_device_h hdevice, uint32_t addr, uint32_t core_id, uint64_t* value) -> int {
    if (!hdevice || !value) return -1;
    auto d = (vx_device*)hdevice;
    const uint32_t offset = addr - VX_CSR_MPM_BASE;
    if (offset > 31) return -1;                               // vortex.cpp:380-382
    d->wait_idle();
    if (addr == VX_CSR_MCYCLE) *value = (uint64_t)((double)d->last_ms * (double)d->prop.clockRate);  // ms * kHz = cycles
    else if (addr == VX_CSR_MINSTRET) *value = core_id == 0 ? d->last_rays : 0;                      // rays traced by the last run
    else *value = 0;
    return 0;
  };

  return 0;
}
