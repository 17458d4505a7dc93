// Probe (measurement only): what does moving an 8.3 MB frame from the device into ordinary (pageable) host memory cost, by method?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 1920ull * 1080 * 4;
  void* d = nullptr; CK(hipMalloc(&d, n)); CK(hipMemset(d, 0x5a, n));
  char* h = (char*)aligned_alloc(4096, (n + 4095) & ~4095ull); memset(h, 1, n);
  char* pin = nullptr; CK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); for (int i = 0; i < 20; ++i) CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); double t1 = now();
    printf("pageable hipMemcpy D2H:              %.3f ms (%.1f GB/s)\n", (t1 - t0) / 20, n / ((t1 - t0) / 20) / 1e6);
    t0 = now(); for (int i = 0; i < 20; ++i) { CK(hipHostRegister(h, n, hipHostRegisterDefault)); CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); CK(hipHostUnregister(h)); } t1 = now();
    printf("register + copy + unregister:        %.3f ms\n", (t1 - t0) / 20);
    CK(hipHostRegister(h, n, hipHostRegisterDefault));
    t0 = now(); for (int i = 0; i < 20; ++i) CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); t1 = now();
    printf("registered destination:              %.3f ms (%.1f GB/s)\n", (t1 - t0) / 20, n / ((t1 - t0) / 20) / 1e6);
    CK(hipHostUnregister(h));
    t0 = now(); for (int i = 0; i < 20; ++i) { CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); } t1 = now();
    printf("into the runtime's pinned memory:    %.3f ms (%.1f GB/s)\n", (t1 - t0) / 20, n / ((t1 - t0) / 20) / 1e6);
    t0 = now(); for (int i = 0; i < 20; ++i) memcpy(h, pin, n); t1 = now();
    printf("memcpy pinned -> pageable, 1 thread: %.3f ms (%.1f GB/s)\n", (t1 - t0) / 20, n / ((t1 - t0) / 20) / 1e6);
    for (int nt : {2, 4, 8}) {
      t0 = now();
      for (int i = 0; i < 20; ++i) { std::vector<std::thread> th; for (int k = 0; k < nt; ++k) th.emplace_back([=] { memcpy(h + n / nt * k, pin + n / nt * k, n / nt); }); for (auto& x : th) x.join(); }
      t1 = now();
      printf("memcpy pinned -> pageable, %d threads (spawned per copy): %.3f ms\n", nt, (t1 - t0) / 20);
    }
    // pipelined: chunks of 1 MB staged through pinned memory on the stream, each copied out by this thread as soon as its event fires
    {
      const size_t ch = 1 << 20; const int nch = (int)((n + ch - 1) / ch);
      std::vector<hipEvent_t> ev(nch); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      t0 = now();
      for (int i = 0; i < 20; ++i) {
        for (int c = 0; c < nch; ++c) { const size_t o = c * ch, l = std::min(ch, n - o); CK(hipMemcpyAsync(pin + o, (char*)d + o, l, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[c], s)); }
        for (int c = 0; c < nch; ++c) { const size_t o = c * ch, l = std::min(ch, n - o); CK(hipEventSynchronize(ev[c])); memcpy(h + o, pin + o, l); }
      }
      t1 = now();
      printf("pipelined 1 MB chunks (DMA into pinned + memcpy by the caller's thread): %.3f ms\n", (t1 - t0) / 20);
    }
  }
  return 0;
}
