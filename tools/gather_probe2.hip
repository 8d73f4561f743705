// gather_probe2.hip — round-2 calibration of the random gather behind `take`.
//
// 1e8 u32 indices into an 8 GiB Int64 table (+ a 128 MiB validity bitmap), the configs[1] take shape.
// Every variant is its own kernel name so that a rocprofv3 --pmc pass attributes counters per variant:
//   value_plain / value_sc0 / value_sc1 / value_sc0sc1 / value_nt   8-byte value gather, load cache policies
//   bit_only                                                        1-bit validity gather
//   value_bit                                                       both (= take_kernel's memory behaviour)
//   *_sorted                                                        ascending indices (filter positions)
//   *_win32m                                                        indices bucketed into 256 x 32 MiB windows
//                                                                   (what a radix partition by value region would feed)
//   scatter_random / scatter_win                                    the un-permute a partitioned take would need:
//                                                                   8-byte stores to random / windowed positions
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe2.hip -o tools/gather_probe2
// Run:   tools/gather_probe2 [iters]      (iters = 1 under --pmc)
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

// mode 0: uniform in [0, range); mode 1: 256 windows in order; mode 2: ascending (i * range / n + jitter)
__global__ void gen_idx(uint32_t* idx, size_t n, uint64_t range, int mode) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint64_t r = mix(i);
    if (mode == 0) idx[i] = (uint32_t)(r % range);
    else if (mode == 1) {
      const uint64_t win = range / 256, per = (n + 255) / 256;
      idx[i] = (uint32_t)((i / per) * win + r % win);
    } else {
      const double step = (double)range / (double)n;
      uint64_t lo = (uint64_t)(i * step);
      uint64_t span = (uint64_t)step;
      idx[i] = (uint32_t)(lo + (span ? r % span : 0));
    }
  }
}
__global__ void fill(uint64_t* v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[i] = mix(i);
}

enum { P_PLAIN = 0, P_SC0, P_SC1, P_SC0SC1, P_NT };

template <int POL> __device__ __forceinline__ uint64_t ld8(const uint64_t* p) {
  uint64_t v;
  if constexpr (POL == P_PLAIN) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == P_SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == P_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POL == P_SC0SC1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

#define GATHER_BODY(LOADV, LOADB)                                                                                   \
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                       \
  constexpr int KU = 4;                                                                                             \
  for (size_t base = (size_t)blockIdx.x * (256 * KU); base < n; base += (size_t)gridDim.x * (256 * KU)) {          \
    size_t wb = base + wave * (64 * KU);                                                                            \
    uint32_t ix[KU];                                                                                                \
    uint64_t v[KU];                                                                                                 \
    uint32_t b[KU];                                                                                                 \
    _Pragma("unroll") for (int k = 0; k < KU; ++k) {                                                                \
      size_t i = wb + k * 64 + lane;                                                                                \
      ix[k] = i < n ? idx[i] : 0;                                                                                   \
    }                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < KU; ++k) { v[k] = 0; b[k] = 1; LOADV; LOADB; }                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
    _Pragma("unroll") for (int k = 0; k < KU; ++k) {                                                                \
      size_t i = wb + k * 64 + lane;                                                                                \
      unsigned long long w = __ballot(b[k] & 1);                                                                    \
      if (i < n) out[i] = v[k];                                                                                     \
      if (lane == 0 && (wb + k * 64) < n) outbits[(wb + k * 64) >> 6] = w;                                          \
    }                                                                                                               \
  }

#define DEF_VALUE(NAME, POL)                                                                                         \
  __global__ void __launch_bounds__(256) NAME(const uint64_t* vals, const uint8_t* bits, const uint32_t* idx,       \
                                              size_t n, uint64_t* out, unsigned long long* outbits) {               \
    GATHER_BODY(v[k] = ld8<POL>(vals + ix[k]), (void)0)                                                              \
  }
DEF_VALUE(value_plain, P_PLAIN)
DEF_VALUE(value_sc0, P_SC0)
DEF_VALUE(value_sc1, P_SC1)
DEF_VALUE(value_sc0sc1, P_SC0SC1)
DEF_VALUE(value_nt, P_NT)
DEF_VALUE(value_plain_sorted, P_PLAIN)
DEF_VALUE(value_plain_win32m, P_PLAIN)

#define DEF_BIT(NAME)                                                                                                \
  __global__ void __launch_bounds__(256) NAME(const uint64_t* vals, const uint8_t* bits, const uint32_t* idx,       \
                                              size_t n, uint64_t* out, unsigned long long* outbits) {               \
    GATHER_BODY((void)0, b[k] = bits[ix[k] >> 3] >> (ix[k] & 7))                                                     \
  }
DEF_BIT(bit_only)
DEF_BIT(bit_only_win32m)

#define DEF_BOTH(NAME)                                                                                               \
  __global__ void __launch_bounds__(256) NAME(const uint64_t* vals, const uint8_t* bits, const uint32_t* idx,       \
                                              size_t n, uint64_t* out, unsigned long long* outbits) {               \
    GATHER_BODY(v[k] = ld8<P_PLAIN>(vals + ix[k]), b[k] = bits[ix[k] >> 3] >> (ix[k] & 7))                           \
  }
DEF_BOTH(value_bit)
DEF_BOTH(value_bit_sorted)
DEF_BOTH(value_bit_win32m)

// un-permute: out[pos[i]] = src[i]
#define DEF_SCATTER(NAME)                                                                                            \
  __global__ void __launch_bounds__(256) NAME(const uint64_t* src, const uint32_t* pos, size_t n, uint64_t* out) {   \
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[pos[i]] = src[i]; \
  }
DEF_SCATTER(scatter_random)
DEF_SCATTER(scatter_win)

// a streaming partition pass stand-in: read idx, write (idx, i) pairs coalesced (the traffic a radix scatter moves)
__global__ void __launch_bounds__(256) pair_stream(const uint32_t* idx, size_t n, uint2* pairs) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    pairs[i] = make_uint2(idx[i], (uint32_t)i);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 5;
  const size_t n = 100000000;
  const size_t rows = (size_t)1 << 30;  // 8 GiB table
  uint64_t *vals, *out, *src;
  uint8_t* bits;
  unsigned long long* outbits;
  uint32_t *idx_r, *idx_w, *idx_s, *pos_r, *pos_w;
  uint2* pairs;
  CK(hipMalloc(&vals, rows * 8));
  CK(hipMalloc(&bits, rows / 8));
  CK(hipMalloc(&out, n * 8));
  CK(hipMalloc(&src, n * 8));
  CK(hipMalloc(&outbits, (n + 63) / 64 * 8));
  CK(hipMalloc(&idx_r, n * 4));
  CK(hipMalloc(&idx_w, n * 4));
  CK(hipMalloc(&idx_s, n * 4));
  CK(hipMalloc(&pos_r, n * 4));
  CK(hipMalloc(&pos_w, n * 4));
  CK(hipMalloc(&pairs, n * 8));
  fill<<<4096, 256>>>(vals, rows);
  fill<<<4096, 256>>>((uint64_t*)bits, rows / 64);
  fill<<<4096, 256>>>(src, n);
  gen_idx<<<4096, 256>>>(idx_r, n, rows, 0);
  gen_idx<<<4096, 256>>>(idx_w, n, rows, 1);
  gen_idx<<<4096, 256>>>(idx_s, n, rows, 2);
  gen_idx<<<4096, 256>>>(pos_r, n, n, 0);
  gen_idx<<<4096, 256>>>(pos_w, n, n, 1);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 4096;
#define RUN(NAME, REQS, ...)                                                                        \
  do {                                                                                              \
    if (iters > 1) { __VA_ARGS__; }                                                                 \
    CK(hipDeviceSynchronize());                                                                     \
    CK(hipEventRecord(e0));                                                                         \
    for (int it = 0; it < iters; ++it) { __VA_ARGS__; }                                             \
    CK(hipEventRecord(e1));                                                                         \
    CK(hipEventSynchronize(e1));                                                                    \
    float ms;                                                                                       \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                           \
    printf("%-22s %8.3f ms  %6.1f G requests/s\n", NAME, ms / iters, (REQS) * (double)n / (ms / iters) / 1e6); \
    fflush(stdout);                                                                                 \
  } while (0)
  RUN("value_plain", 1, (value_plain<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("value_sc0", 1, (value_sc0<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("value_sc1", 1, (value_sc1<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("value_sc0sc1", 1, (value_sc0sc1<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("value_nt", 1, (value_nt<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("bit_only", 1, (bit_only<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("value_bit", 2, (value_bit<<<grid, 256>>>(vals, bits, idx_r, n, out, outbits)));
  RUN("value_plain_sorted", 1, (value_plain_sorted<<<grid, 256>>>(vals, bits, idx_s, n, out, outbits)));
  RUN("value_bit_sorted", 2, (value_bit_sorted<<<grid, 256>>>(vals, bits, idx_s, n, out, outbits)));
  RUN("value_plain_win32m", 1, (value_plain_win32m<<<grid, 256>>>(vals, bits, idx_w, n, out, outbits)));
  RUN("bit_only_win32m", 1, (bit_only_win32m<<<grid, 256>>>(vals, bits, idx_w, n, out, outbits)));
  RUN("value_bit_win32m", 2, (value_bit_win32m<<<grid, 256>>>(vals, bits, idx_w, n, out, outbits)));
  RUN("scatter_random", 1, (scatter_random<<<grid, 256>>>(src, pos_r, n, out)));
  RUN("scatter_win", 1, (scatter_win<<<grid, 256>>>(src, pos_w, n, out)));
  RUN("pair_stream", 1, (pair_stream<<<grid, 256>>>(idx_r, n, pairs)));
  return 0;
}
