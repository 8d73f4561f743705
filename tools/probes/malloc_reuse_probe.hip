// hipMalloc / hipFree cycles with VA reuse: poison (kernel-side memset), pageable H2D copy, read back, compare; a few blocks stay live
// and are re-checked after every cycle.  The plain-allocator counterpart of vmm_probe (which shows stale data after VA reuse).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
struct Block { unsigned char* p; size_t bytes; std::vector<unsigned char> host; };
static Block make(size_t bytes, hipStream_t s, int salt) {
  Block b{};
  CK(hipMalloc((void**)&b.p, bytes));
  b.bytes = bytes;
  CK(hipMemsetAsync(b.p, 0xCD, bytes, s));
  b.host.resize(bytes);
  for (size_t i = 0; i < bytes; ++i) b.host[i] = (unsigned char)((i * 13 + bytes + salt) | 1);
  CK(hipMemcpyAsync(b.p, b.host.data(), bytes, hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));
  return b;
}
static size_t check(const Block& b) {
  std::vector<unsigned char> back(b.bytes);
  CK(hipMemcpy(back.data(), b.p, b.bytes, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < b.bytes; ++i) bad += back[i] != b.host[i];
  return bad;
}
int main(int argc, char** argv) {
  const int cycles = argc > 1 ? atoi(argv[1]) : 300;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<Block> live;
  int failing = 0;
  const size_t sizes[] = {256, 4096, 1000, 65536, 1 << 20, 3 << 20, 8 << 20, 100000, 512, 2 << 20};
  for (int i = 0; i < cycles; ++i) {
    Block b = make(sizes[i % 10] + (size_t)(i % 7) * 16, s, i);
    if (check(b)) { printf("cycle %d: fresh block (%zu B at %p) bad\n", i, b.bytes, b.p); ++failing; }
    if (i % 25 == 0 && live.size() < 8) live.push_back(b);
    else { CK(hipDeviceSynchronize()); CK(hipFree(b.p)); }
    for (size_t k = 0; k < live.size(); ++k)
      if (check(live[k])) { printf("cycle %d: live block %zu (%zu B at %p) bad\n", i, k, live[k].bytes, live[k].p); ++failing; }
  }
  printf("hipMalloc/hipFree reuse: failing checks %d of %d cycles\n", failing, cycles);
  return 0;
}
