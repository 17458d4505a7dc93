// Probe (measurement only, not part of the product): does this stack support stream memory operations, and what do they cost?
//   stream A: a kernel that spins ~200 us, then stores 1 to a signal word (system-scope release)
//   stream B: hipStreamWaitValue32(word >= 1), then a kernel that stamps the 100 MHz clock
// Prints the attribute, the API return codes and the delay between the store and the dependent kernel's start.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void producer(uint32_t* flag, unsigned long long* stamp, uint32_t spin_us) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
  stamp[0] = wall_clock64();
  __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void consumer(unsigned long long* stamp) { stamp[1] = wall_clock64(); }
int main() {
  int dev = 0, can = -1;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  uint32_t* flag = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  unsigned long long* stamp = nullptr;
  CK(hipMalloc((void**)&stamp, 16));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(flag, 0, 8));
    CK(hipMemset(stamp, 0, 16));
    CK(hipDeviceSynchronize());
    // consumer side first: the wait is queued before the producer even starts
    e = hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
    if (rep == 0) printf("hipStreamWaitValue32 -> %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamp);
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, flag, stamp, 200u);
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    unsigned long long h[2];
    CK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
    printf("rep %d: store -> dependent kernel start: %.1f us\n", rep, (double)((long long)h[1] - (long long)h[0]) / 100.0);
  }
  // plain device memory as the wait target (the kernel's control block would be plain hipMalloc memory)
  uint32_t* plain = nullptr;
  CK(hipMalloc((void**)&plain, 8));
  CK(hipMemset(plain, 0, 8));
  CK(hipDeviceSynchronize());
  e = hipStreamWaitValue32(b, plain, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
  printf("hipStreamWaitValue32 on plain hipMalloc memory -> %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    CK(hipMemset(stamp, 0, 16));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamp);
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, plain, stamp, 200u);
    CK(hipStreamSynchronize(a));
    CK(hipStreamSynchronize(b));
    unsigned long long h[2];
    CK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
    printf("plain memory: store -> dependent kernel start: %.1f us\n", (double)((long long)h[1] - (long long)h[0]) / 100.0);
  }
  printf("done\n");
  return 0;
}
