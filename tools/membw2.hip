// membw2.hip — store-policy probe: triad (2 reads + 1 write, like add_wrapping) and copy with
// plain / nontemporal stores and loads, grid sweep.  hipcc --offload-arch=gfx950 -O3 tools/membw2.hip -o tools/membw2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int U, bool NTS, bool NTL>
__global__ void __launch_bounds__(256) triad(const d2* a, const d2* b, d2* o, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
    d2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) {
      x[u] = NTL ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
      y[u] = NTL ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) {
      d2 r = x[u] + y[u];
      if (NTS) __builtin_nontemporal_store(r, o + i + u * stride); else o[i + u * stride] = r;
    }
  }
}
// block-contiguous variant: each block handles a contiguous chunk of U*256 vectors per iteration
template <int U, bool NTS>
__global__ void __launch_bounds__(256) triad_tile(const d2* a, const d2* b, d2* o, size_t n) {
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
    d2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + u * 256 + threadIdx.x; if (i < n) { x[u] = a[i]; y[u] = b[i]; } }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + u * 256 + threadIdx.x; if (i < n) { d2 r = x[u] + y[u];
      if (NTS) __builtin_nontemporal_store(r, o + i); else o[i] = r; } }
  }
}
template <typename F> double time_ms(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  size_t bytes = 8ull << 30; size_t n = bytes / 16;
  d2 *a, *b, *o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  for (int grid : {2048, 4096, 8192, 16384, 65536}) {
    double t0 = time_ms([&] { triad<4, false, false><<<grid, 256>>>(a, b, o, n); }, 5);
    double t1 = time_ms([&] { triad<4, true, false><<<grid, 256>>>(a, b, o, n); }, 5);
    double t2 = time_ms([&] { triad<4, true, true><<<grid, 256>>>(a, b, o, n); }, 5);
    double t3 = time_ms([&] { triad<2, true, false><<<grid, 256>>>(a, b, o, n); }, 5);
    double t4 = time_ms([&] { triad_tile<4, false><<<grid, 256>>>(a, b, o, n); }, 5);
    double t5 = time_ms([&] { triad_tile<4, true><<<grid, 256>>>(a, b, o, n); }, 5);
    printf("grid %6d triad GB/s: plain %.0f  nt-store %.0f  nt-store+nt-load %.0f  U2 nt-store %.0f  tile plain %.0f  tile nt %.0f\n", grid,
           3.0 * bytes / t0 / 1e6, 3.0 * bytes / t1 / 1e6, 3.0 * bytes / t2 / 1e6, 3.0 * bytes / t3 / 1e6, 3.0 * bytes / t4 / 1e6, 3.0 * bytes / t5 / 1e6);
  }
  return 0;
}
