// gather_probe.hip — what bounds a random 8-byte gather on this box?  1e8 uniform u32 indices
// into a table of 2^R bytes (R = 27 … 33), with plain / non-temporal loads,
// 4 … 16 gathers in flight per lane and 1× … 16× the grid.  If time falls with the table size below the MALL (256 MB) the bound is HBM line traffic;
// if it falls only under the TLB reach it is translation.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
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
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    idx[i] = (uint32_t)(mix(i) % range);
}
__global__ void fill(uint64_t* v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[i] = i * 3 + 1;
}

template <int MODE> __device__ __forceinline__ uint64_t ld(const uint64_t* p) {
  if constexpr (MODE == 0) return *p;
  else return __builtin_nontemporal_load(p);
}

template <int MODE, int KU>
__global__ void __launch_bounds__(256) gather(const uint64_t* vals, const uint32_t* idx, size_t n, uint64_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t base = (size_t)blockIdx.x * (256 * KU); base < n; base += (size_t)gridDim.x * (256 * KU)) {
    size_t wb = base + wave * (64 * KU);
    uint32_t ix[KU];
    uint64_t v[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      size_t i = wb + k * 64 + lane;
      ix[k] = i < n ? idx[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < KU; ++k) v[k] = ld<MODE>(vals + ix[k]);
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      size_t i = wb + k * 64 + lane;
      if (i < n) out[i] = v[k];
    }
  }
}

int main(int argc, char** argv) {
  const size_t n = 100000000;
  const size_t max_rows = (size_t)1 << 30;  // 8 GiB table
  uint64_t *vals, *out;
  uint32_t* idx;
  CK(hipMalloc(&vals, max_rows * 8));
  CK(hipMalloc(&out, n * 8));
  CK(hipMalloc(&idx, n * 4));
  fill<<<4096, 256>>>(vals, max_rows);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  for (int r = 30; r >= 21; r -= 3) {
    uint64_t rows = (uint64_t)1 << r;
    gen_idx<<<4096, 256>>>(idx, n, rows);
    CK(hipDeviceSynchronize());
#define RUN(NAME, ...)                                                                              \
  do {                                                                                              \
    __VA_ARGS__;                                                                                    \
    CK(hipDeviceSynchronize());                                                                     \
    CK(hipEventRecord(e0));                                                                         \
    for (int it = 0; it < 5; ++it) { __VA_ARGS__; }                                                 \
    CK(hipEventRecord(e1));                                                                         \
    CK(hipEventSynchronize(e1));                                                                    \
    float ms;                                                                                       \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                           \
    printf("table %6.0f MiB  %-22s %.3f ms  %.1f G gathers/s\n", rows * 8.0 / 1048576, NAME, ms / 5, \
           n / (ms / 5) / 1e6);                                                                     \
  } while (0)
    RUN("plain ku4", (gather<0, 4><<<grid, 256>>>(vals, idx, n, out)));
    RUN("plain ku8", (gather<0, 8><<<grid, 256>>>(vals, idx, n, out)));
    RUN("plain ku16", (gather<0, 16><<<grid, 256>>>(vals, idx, n, out)));
    RUN("nontemporal ku4", (gather<1, 4><<<grid, 256>>>(vals, idx, n, out)));
    RUN("nontemporal ku8", (gather<1, 8><<<grid, 256>>>(vals, idx, n, out)));
    RUN("plain ku4 grid x4", (gather<0, 4><<<grid * 4, 256>>>(vals, idx, n, out)));
    RUN("plain ku4 grid x16", (gather<0, 4><<<grid * 16, 256>>>(vals, idx, n, out)));
  }
  return 0;
}
