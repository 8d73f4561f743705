// Does a pageable H2D hipMemcpyAsync stay ordered behind a hipMemsetAsync / kernel on the same (non-blocking) stream?
// Checked for hipMalloc memory and for a hipMemMap'ed range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void slow_fill(unsigned char* p, size_t n, unsigned char v, int spin) {
  for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(127);
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static int run(const char* what, unsigned char* dev, size_t cap, hipStream_t s) {
  int bad_total = 0;
  for (size_t bytes : {8ul, 125ul, 1000ul, 8000ul, 70000ul, 1000000ul}) {
    if (bytes > cap) continue;
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<unsigned char> h(bytes), back(bytes);
      for (size_t i = 0; i < bytes; ++i) h[i] = (unsigned char)(i * 7 + 1) | 1;
      if (mode == 0) CK(hipMemsetAsync(dev, 0xCD, cap, s));
      else slow_fill<<<64, 256, 0, s>>>(dev, cap, 0xCD, 2000);
      CK(hipMemcpyAsync(dev, h.data(), bytes, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(back.data(), dev, bytes, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < bytes; ++i) bad += back[i] != h[i];
      printf("%s bytes=%zu before=%s -> %zu bad bytes\n", what, bytes, mode ? "slow kernel" : "hipMemsetAsync", bad);
      bad_total += bad != 0;
    }
  }
  return bad_total;
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned char* d = nullptr;
  const size_t cap = 2 << 20;
  CK(hipMalloc((void**)&d, cap));
  int bad = run("hipMalloc", d, cap, s);
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  printf("granularity %zu\n", gran);
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, cap + gran, gran, nullptr, 0));
  hipMemGenericAllocationHandle_t h;
  CK(hipMemCreate(&h, cap, &prop, 0));
  CK(hipMemMap(va, cap, 0, h, 0));
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, cap, &acc, 1));
  bad += run("hipMemMap", (unsigned char*)va, cap, s);
  // interior pointer of the mapping, like the guard allocator hands out
  bad += run("hipMemMap+off", (unsigned char*)va + 4096 + 16, cap - 4096 - 16, s);
  printf("TOTAL failing cases: %d\n", bad);
  return 0;
}
