// mall_probe.hip — round 4, VERDICT r03 "Next round" #4: does pairing the two passes of a "read twice" kernel chunk by
// chunk through the 256 MiB Infinity Cache (MALL) buy anything on MI355X?
//
// Two product kernels read an operand stream twice by construction:
//   X2  Float64 -> Utf8: string_len reads the 4.3 GB value stream, string_write reads it AGAIN and writes ~8.3 GB
//       (offsets + text)                                               -> traffic mix "x2": A reads S, B reads S, writes 2 S
//   lazy predicate: filter_expr_count reads a and b, filter_scatter re-reads a and writes the 10 % selected rows
//                                                                      -> traffic mix "expr": A reads 2 S, B reads S, writes 0.1 S
// Before any product code: the same byte traffic with NOTHING else in the kernels (16-byte accesses, tile-contiguous
// blocks like the product's), run (i) un-paired = A over everything, then B over everything (today's plan), (ii) paired on
// one stream = for each chunk c: A(c); B(c), (iii) paired on two streams = A(c + 1) overlapped with B(c).  If pass B of
// chunk c finds c's lines in the MALL its reads cost no HBM time; the price is 2 launches per chunk.  The ratio
// (paired / un-paired) is the CEILING of what the product could gain — the kill criterion is < 8 %.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/mall_probe ; run: tools/mall_probe [GiB of S, default 4]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE16 = 2048;  // 32 KiB of 16-byte words per block iteration (8 loads in flight per lane)

// pass A: read NS streams tile by tile, keep a checksum (one 4-byte store per block)
template <int NS>
__global__ void __launch_bounds__(256) pass_a(const u32x4* s0, const u32x4* s1, size_t n16, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t tile = blockIdx.x; tile * TILE16 < n16; tile += gridDim.x) {
    const size_t base = tile * TILE16;
    u32x4 v[8 * NS];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      v[u] = i < n16 ? s0[i] : u32x4{0, 0, 0, 0};
      if (NS == 2) v[8 + u] = i < n16 ? s1[i] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 8 * NS; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}

// pass B: re-read s0, write WNUM / WDEN times as many bytes to dst (2/1 = offsets + text; 1/10 = the selected rows)
template <int WNUM, int WDEN>
__global__ void __launch_bounds__(256) pass_b(const u32x4* s0, u32x4* dst, size_t n16) {
  for (size_t tile = blockIdx.x; tile * TILE16 < n16; tile += gridDim.x) {
    const size_t base = tile * TILE16;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      v[u] = i < n16 ? s0[i] : u32x4{0, 0, 0, 0};
    }
    if (WNUM >= WDEN) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t i = base + u * 256 + threadIdx.x;
        if (i < n16)
          for (int k = 0; k < WNUM / WDEN; ++k) dst[(size_t)k * n16 + i] = v[u] + (unsigned)k;
      }
    } else {  // WNUM / WDEN of the tile's words, dense and contiguous like a compaction's output
      constexpr int OUT = TILE16 * WNUM / WDEN;  // 204 words per 2048-word tile
      u32x4 acc = v[0];
#pragma unroll
      for (int u = 1; u < 8; ++u) acc ^= v[u];
      if ((int)threadIdx.x < OUT) dst[tile * OUT + threadIdx.x] = acc;
    }
  }
}

struct Mix {
  const char* name;
  int a_streams;   // streams pass A reads
  int wnum, wden;  // bytes pass B writes per byte of S
};

static float run_mix(const Mix& m, const u32x4* s0, const u32x4* s1, u32x4* dst, unsigned* sink, size_t n16, size_t chunk16, int mode,
                     hipStream_t st0, hipStream_t st1, hipEvent_t* evA, hipEvent_t* evB) {
  // mode 0: un-paired; 1: paired, one stream; 2: paired, A(c + 1) on a second stream while B(c) runs
  auto launch_a = [&](size_t off, size_t n, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<size_t>(256 * 8, (n + TILE16 - 1) / TILE16);
    if (m.a_streams == 2) pass_a<2><<<grid, 256, 0, st>>>(s0 + off, s1 + off, n, sink);
    else pass_a<1><<<grid, 256, 0, st>>>(s0 + off, nullptr, n, sink);
  };
  auto launch_b = [&](size_t off, size_t n, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<size_t>(256 * 8, (n + TILE16 - 1) / TILE16);
    if (m.wnum == 2) pass_b<2, 1><<<grid, 256, 0, st>>>(s0 + off, dst + 2 * off, n);  // (chunk-local 2 x n region)
    else pass_b<1, 10><<<grid, 256, 0, st>>>(s0 + off, dst + (off / TILE16) * (TILE16 / 10), n);
  };
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(t0, st0));
  if (mode == 0) {
    launch_a(0, n16, st0);
    launch_b(0, n16, st0);
  } else if (mode == 1) {
    for (size_t off = 0; off < n16; off += chunk16) {
      const size_t n = std::min(chunk16, n16 - off);
      launch_a(off, n, st0);
      launch_b(off, n, st0);
    }
  } else {
    // A on st1 runs one chunk ahead; B(c) on st0 waits for A(c); A(c + 2) waits for B(c) so that A never runs more
    // than one chunk ahead of the writes (the MALL holds A's chunk until B has used it)
    size_t c = 0;
    const size_t nchunks = (n16 + chunk16 - 1) / chunk16;
    CK(hipStreamWaitEvent(st1, t0, 0));
    for (c = 0; c < nchunks; ++c) {
      const size_t off = c * chunk16, n = std::min(chunk16, n16 - off);
      if (c >= 2) CK(hipStreamWaitEvent(st1, evB[(c - 2) & 3], 0));
      launch_a(off, n, st1);
      CK(hipEventRecord(evA[c & 3], st1));
      CK(hipStreamWaitEvent(st0, evA[c & 3], 0));
      launch_b(off, n, st0);
      CK(hipEventRecord(evB[c & 3], st0));
    }
  }
  CK(hipEventRecord(t1, st0));
  CK(hipEventSynchronize(t1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, t0, t1));
  CK(hipEventDestroy(t0));
  CK(hipEventDestroy(t1));
  return ms;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)((TILE16 * 16) - 1);
  const size_t n16 = bytes / 16;
  u32x4 *s0, *s1, *dst;
  unsigned* sink;
  CK(hipMalloc(&s0, bytes));
  CK(hipMalloc(&s1, bytes));
  CK(hipMalloc(&dst, 2 * bytes + (1 << 20)));
  CK(hipMalloc(&sink, 1 << 16));
  CK(hipMemset(s0, 1, bytes));
  CK(hipMemset(s1, 2, bytes));
  CK(hipMemset(dst, 0, 2 * bytes));
  hipStream_t st0, st1;
  CK(hipStreamCreateWithFlags(&st0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&st1, hipStreamNonBlocking));
  hipEvent_t evA[4], evB[4];
  for (int i = 0; i < 4; ++i) {
    CK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evB[i], hipEventDisableTiming));
  }
  const Mix mixes[] = {{"x2   (A reads S | B reads S, writes 2 S)", 1, 2, 1}, {"expr (A reads 2 S | B reads S, writes 0.1 S)", 2, 1, 10}};
  const size_t chunk_mib[] = {16, 32, 64, 128, 256, 512};
  printf("mall_probe: S = %.2f GiB per stream; times are the best of 5 runs, ms\n", bytes / double(1ull << 30));
  for (const Mix& m : mixes) {
    const double moved = bytes * (double)(m.a_streams + 1) + bytes * (double)m.wnum / m.wden;
    float base = 1e30f;
    for (int r = 0; r < 5; ++r) base = std::min(base, run_mix(m, s0, s1, dst, sink, n16, n16, 0, st0, st1, evA, evB));
    printf("\n%s\n  un-paired: %.3f ms  (%.2f TB/s over the %.1f GB both passes move)\n", m.name, base, moved / base / 1e9, moved / 1e9);
    for (size_t cm : chunk_mib) {
      const size_t chunk16 = cm * (1u << 20) / 16;
      if (chunk16 >= n16) continue;
      float one = 1e30f, two = 1e30f;
      for (int r = 0; r < 5; ++r) {
        one = std::min(one, run_mix(m, s0, s1, dst, sink, n16, chunk16, 1, st0, st1, evA, evB));
        two = std::min(two, run_mix(m, s0, s1, dst, sink, n16, chunk16, 2, st0, st1, evA, evB));
      }
      printf("  chunk %4zu MiB (%4zu chunks): one stream %.3f ms (%+.1f %%)   two streams %.3f ms (%+.1f %%)\n", cm, (n16 + chunk16 - 1) / chunk16,
             one, (one / base - 1) * 100, two, (two / base - 1) * 100);
    }
  }
  return 0;
}
