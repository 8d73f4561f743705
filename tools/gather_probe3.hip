// gather_probe3.hip — does the ALLOCATION's memory type change what a random gather costs?  (r02 follow-up to
// gather_probe2: every 8-byte gather from ordinary hipMalloc memory is a 128-byte L2 line fill, whatever the load's
// cache policy bits say.)  Same shape as configs[1]'s take — 1e8 u32 indices into an 8 GiB Int64 table + 128 MiB
// bitmap — with the table in (a) hipMalloc memory, (b) hipDeviceMallocUncached, (c) hipDeviceMallocFinegrained; plus a
// streaming read and a streaming copy of the table, because a filter has to stream the same buffers.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe3.hip -o tools/gather_probe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void gen_idx(uint32_t* idx, size_t n, uint64_t range) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) idx[i] = (uint32_t)(mix(i) % range);
}
__global__ void fill(uint64_t* v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[i] = mix(i);
}
template <bool V, bool B>
__global__ void __launch_bounds__(256) gather(const uint64_t* vals, const uint8_t* bits, const uint32_t* idx, size_t n, uint64_t* out) {
  constexpr int KU = 4;
  for (size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * KU; base < n; base += (size_t)gridDim.x * 256 * KU) {
    uint32_t ix[KU];
    uint64_t v[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) ix[k] = base + k < n ? idx[base + k] : 0;
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      v[k] = 0;
      if (V) v[k] = vals[ix[k]];
      if (B) v[k] += (bits[ix[k] >> 3] >> (ix[k] & 7)) & 1;
    }
#pragma unroll
    for (int k = 0; k < KU; ++k) if (base + k < n) out[base + k] = v[k];
  }
}
__global__ void __launch_bounds__(256) stream_sum(const uint4* p, size_t n16, uint64_t* out) {
  uint64_t s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 x = p[i]; s += x.x + x.y + x.z + x.w; }
  if (s == 0x1234567) out[0] = s;
}
__global__ void __launch_bounds__(256) stream_copy(const uint4* p, uint4* d, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = p[i];
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 5;
  const size_t n = 100000000, rows = (size_t)1 << 30;
  uint32_t* idx; uint64_t* out;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, n * 8));
  gen_idx<<<4096, 256>>>(idx, n, rows);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"hipMalloc", "uncached", "finegrained"};
  for (int mode = 0; mode < 3; ++mode) {
    uint64_t* vals = nullptr; uint8_t* bits = nullptr; uint4* dst = nullptr;
    hipError_t e;
    if (mode == 0) { e = hipMalloc(&vals, rows * 8); if (e == hipSuccess) e = hipMalloc(&bits, rows / 8); }
    else {
      const unsigned fl = mode == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
      e = hipExtMallocWithFlags((void**)&vals, rows * 8, fl); if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&bits, rows / 8, fl);
    }
    if (e != hipSuccess) { printf("%s: allocation failed: %s\n", names[mode], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CK(hipMalloc(&dst, (size_t)1 << 30));
    fill<<<4096, 256>>>(vals, rows); fill<<<4096, 256>>>((uint64_t*)bits, rows / 64);
    CK(hipDeviceSynchronize());
#define RUN(NAME, BYTES, ...) do { __VA_ARGS__; CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int it = 0; it < iters; ++it) { __VA_ARGS__; } \
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-12s %-14s %8.3f ms  %7.1f GB/s\n", names[mode], NAME, ms / iters, (BYTES) / (ms / iters) / 1e6); fflush(stdout); } while (0)
    RUN("value", 0.0, (gather<true, false><<<4096, 256>>>(vals, bits, idx, n, out)));
    RUN("bit", 0.0, (gather<false, true><<<4096, 256>>>(vals, bits, idx, n, out)));
    RUN("value+bit", 0.0, (gather<true, true><<<4096, 256>>>(vals, bits, idx, n, out)));
    RUN("stream_sum", (double)rows * 8, (stream_sum<<<8192, 256>>>((const uint4*)vals, rows / 2, out)));
    RUN("stream_copy1G", 2.0 * (1 << 30), (stream_copy<<<8192, 256>>>((const uint4*)vals, dst, ((size_t)1 << 30) / 16)));
    RUN("write_into1G", 2.0 * (1 << 30), (stream_copy<<<8192, 256>>>((const uint4*)dst, (uint4*)vals, ((size_t)1 << 30) / 16)));
    CK(hipFree(vals)); CK(hipFree(bits)); CK(hipFree(dst));
  }
  return 0;
}
