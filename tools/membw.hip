// membw.hip — HBM roofline probe for this box (MI355X): read-only, copy and write streams
// with 16-byte accesses, so bench.py's roofline fractions can be read against the measured
// ceiling as well as the 8 TB/s datasheet peak.  Build: hipcc --offload-arch=gfx950 -O3
// tools/membw.hip -o tools/membw ; run: tools/membw [GiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(256) read_kernel(const u32x4* in, size_t n, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  for (; i < n; i += stride) acc ^= in[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// tile-contiguous variant: each block reads a contiguous 32 KiB tile (like filter_scatter)
__global__ void __launch_bounds__(256) read_tile_kernel(const u32x4* in, size_t n, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t tile = blockIdx.x; tile * 2048 < n; tile += gridDim.x) {
    const u32x4* p = in + tile * 2048;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int U>
__global__ void __launch_bounds__(256) copy_kernel(const u32x4* in, u32x4* out, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) out[i + u * stride] = v[u];
  }
  for (; i < n; i += stride) out[i] = in[i];
}

__global__ void __launch_bounds__(256) write_kernel(u32x4* out, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  u32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = v;
}

template <typename F>
double time_ms(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  double gib = argc > 1 ? atof(argv[1]) : 8.0;
  size_t bytes = (size_t)(gib * (1ull << 30));
  size_t n = bytes / 16;
  u32x4 *in, *out;
  unsigned* flag;
  CK(hipMalloc(&in, bytes));
  CK(hipMalloc(&out, bytes));
  CK(hipMalloc(&flag, 4));
  CK(hipMemset(in, 1, bytes));
  CK(hipMemset(out, 0, bytes));
  printf("buffer %.1f GiB\n", gib);
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    double r4 = time_ms([&] { read_kernel<4><<<grid, 256>>>(in, n, flag); }, 5);
    double r8 = time_ms([&] { read_kernel<8><<<grid, 256>>>(in, n, flag); }, 5);
    double rt = time_ms([&] { read_tile_kernel<<<grid, 256>>>(in, n, flag); }, 5);
    double c4 = time_ms([&] { copy_kernel<4><<<grid, 256>>>(in, out, n); }, 5);
    double w = time_ms([&] { write_kernel<<<grid, 256>>>(out, n); }, 5);
    printf("grid %6d: read U4 %.0f GB/s  read U8 %.0f GB/s  read tile %.0f GB/s  copy(r+w) %.0f GB/s  write %.0f GB/s\n",
           grid, bytes / r4 / 1e6, bytes / r8 / 1e6, bytes / rt / 1e6, 2.0 * bytes / c4 / 1e6, bytes / w / 1e6);
  }
  // one tile per block, like the non-persistent scatter kernel
  {
    unsigned g = (unsigned)((n + 2047) / 2048);
    double rt = time_ms([&] { read_tile_kernel<<<g, 256>>>(in, n, flag); }, 5);
    printf("grid %u (1 tile/block): read tile %.0f GB/s\n", g, bytes / rt / 1e6);
  }
  return 0;
}
