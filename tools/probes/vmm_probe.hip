// Several hipMemMap'ed blocks, each poisoned with hipMemsetAsync and then filled by a pageable H2D copy on the same stream
// (what AH_DEBUG_GUARD=1 does per buffer): does every block still hold its data afterwards?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
struct Block { void* va; size_t va_bytes, map_bytes; hipMemGenericAllocationHandle_t h; unsigned char* p; size_t bytes; std::vector<unsigned char> host; };
static size_t gran;
static hipMemAllocationProp prop;
static char* hint = nullptr;  // != nullptr: every reservation asks for the next never-used address (argv[4] = 1)
static int hint_missed = 0;
static Block make(size_t bytes, hipStream_t s, bool memset_whole, int fill) {
  Block b{};
  const size_t padded = (bytes + 15) & ~(size_t)15;
  b.map_bytes = (padded + gran - 1) / gran * gran;
  b.va_bytes = b.map_bytes + gran;
  CK(hipMemAddressReserve(&b.va, b.va_bytes, gran, hint, 0));
  if (hint) {
    if (b.va != hint) ++hint_missed;
    if ((char*)b.va + b.va_bytes > hint) hint = (char*)b.va + b.va_bytes;
  }
  CK(hipMemCreate(&b.h, b.map_bytes, &prop, 0));
  CK(hipMemMap(b.va, b.map_bytes, 0, b.h, 0));
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(b.va, b.map_bytes, &acc, 1));
  b.p = (unsigned char*)b.va + b.map_bytes - padded;
  b.bytes = bytes;
  if (memset_whole) CK(hipMemsetAsync(b.va, fill, b.map_bytes, s));
  b.host.resize(bytes);
  for (size_t i = 0; i < bytes; ++i) b.host[i] = (unsigned char)((i * 13 + bytes) | 1);
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
  const int fill = argc > 1 ? (int)strtol(argv[1], nullptr, 16) : 0xCD;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  std::vector<Block> live;
  int failing = 0;
  if (argc > 4 && atoi(argv[4])) hint = (char*)0x200000000000ull;
  const int cycles = argc > 2 ? atoi(argv[2]) : 50;
  for (int i = 0; i < cycles; ++i) {  // the guard allocator's release: drain, unmap, release, free the reservation (VAs get reused)
    Block b = make(1000 + i, s, true, fill);
    if (check(b)) { printf("warm-up block %d bad\n", i); ++failing; }
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(b.va, b.map_bytes));
    CK(hipMemRelease(b.h));
    if (!(argc > 3 && atoi(argv[3]))) CK(hipMemAddressFree(b.va, b.va_bytes));  // argv[3] = 1: the reservation is kept (no VA reuse)
  }
  for (size_t bytes : {8ul, 1ul, 40ul, 512ul, 8ul, 8000ul, 125ul, 8000ul, 128ul, 70000ul, 100ul}) {
    live.push_back(make(bytes, s, true, fill));
    for (size_t k = 0; k < live.size(); ++k) {
      size_t bad = check(live[k]);
      if (bad) { printf("after block %zu (%zu B at %p, va %p): block %zu (%zu B at %p) has %zu bad bytes\n", live.size() - 1, bytes, live.back().p, live.back().va, k, live[k].bytes, live[k].p, bad); ++failing; }
    }
  }
  printf("fill %02X: failing checks %d (hints missed %d, last hint %p)\n", fill, failing, hint_missed, (void*)hint);
  return 0;
}
