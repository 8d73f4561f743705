// gen.hip — counter-based synthetic columns, generated in HBM
// because a 1e9-row table cannot be pushed over PCIe inside a bench run.  Every
// value is splitmix64(seed, global_row), so a CPU checker can reproduce any
// window bit-for-bit on the CPU.  Distributions follow the reference's bench
// generators (arrow/src/util/bench_util.rs:45-60,156-200): uniform values,
// Bernoulli validity / predicate bits, null slots zeroed.
#include "common.hpp"

namespace {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void gen_i64_kernel(int64_t* dst, int64_t n, uint64_t seed, int64_t lo, uint64_t range,
                               int64_t row0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = splitmix64(seed, (uint64_t)(row0 + i));
    dst[i] = range ? (int64_t)((uint64_t)lo + __umul64hi(r, range)) : (int64_t)r;
  }
}
__global__ void gen_i32_kernel(int32_t* dst, int64_t n, uint64_t seed, int64_t row0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (int32_t)(uint32_t)splitmix64(seed, (uint64_t)(row0 + i));
}
__global__ void gen_u32_kernel(uint32_t* dst, int64_t n, uint64_t seed, uint32_t bound,
                               int64_t row0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r = (uint32_t)(splitmix64(seed, (uint64_t)(row0 + i)) >> 32);
    dst[i] = bound ? (uint32_t)(((uint64_t)r * bound) >> 32) : r;
  }
}
__global__ void gen_f64_kernel(double* dst, int64_t n, uint64_t seed, double lo, double span,
                               int64_t row0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = splitmix64(seed, (uint64_t)(row0 + i));
    double u = (double)(r >> 11) * 0x1.0p-53;
    dst[i] = fma(u, span, lo);  // explicit fma: identical on host and device
  }
}
__global__ void gen_f32_kernel(float* dst, int64_t n, uint64_t seed, float lo, float span, int64_t row0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = splitmix64(seed, (uint64_t)(row0 + i));
    float u = (float)(r >> 40) * 0x1.0p-24f;
    dst[i] = fmaf(u, span, lo);
  }
}
// 1- and 2-byte columns: the low bytes of splitmix64(seed, row) (full-range Int8 / Int16 bit patterns)
template <typename T>
__global__ void gen_small_kernel(T* dst, int64_t n, uint64_t seed, int64_t row0) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (T)splitmix64(seed, (uint64_t)(row0 + i));
}
__global__ void gen_bits_kernel(unsigned long long* dst, int64_t n, uint64_t seed,
                                uint64_t threshold, int64_t row0) {
  int64_t nwords = (n + 63) >> 6;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
       w += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long word = 0;
    int64_t base = w << 6;
    int lim = (int)((n - base) < 64 ? (n - base) : 64);
    for (int b = 0; b < lim; ++b) {
      uint64_t r = splitmix64(seed, (uint64_t)(row0 + base + b));
      word |= (unsigned long long)((r >> 11) < threshold) << b;
    }
    dst[w] = word;
  }
}
template <typename T>
__global__ void zero_nulls_kernel(T* v, const unsigned long long* valid, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    if (!((valid[i >> 6] >> (i & 63)) & 1)) v[i] = T{};
}

__global__ void gen_iota_u32_kernel(uint32_t* dst, int64_t n, uint32_t start) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = start + (uint32_t)i;
}

int gen_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(n, 256), 256 * 32)); }

}  // namespace

extern "C" ah_status ah_gen_uniform_i64(ah_context* ctx, int64_t* dst, int64_t n, uint64_t seed,
                                        int64_t lo, int64_t hi, int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  uint64_t range = (uint64_t)hi - (uint64_t)lo + 1;  // 0 = full 64-bit range
  gen_i64_kernel<<<gen_grid(n), 256, 0, ctx->stream>>>(dst, n, seed, lo, range, row0);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_gen_uniform_i32(ah_context* ctx, int32_t* dst, int64_t n, uint64_t seed,
                                        int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  gen_i32_kernel<<<gen_grid(n), 256, 0, ctx->stream>>>(dst, n, seed, row0);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_gen_uniform_u32(ah_context* ctx, uint32_t* dst, int64_t n, uint64_t seed,
                                        uint32_t bound, int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  gen_u32_kernel<<<gen_grid(n), 256, 0, ctx->stream>>>(dst, n, seed, bound, row0);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_gen_uniform_f64(ah_context* ctx, double* dst, int64_t n, uint64_t seed,
                                        double lo, double hi, int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  gen_f64_kernel<<<gen_grid(n), 256, 0, ctx->stream>>>(dst, n, seed, lo, hi - lo, row0);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_gen_uniform_f32(ah_context* ctx, float* dst, int64_t n, uint64_t seed,
                                        float lo, float hi, int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  gen_f32_kernel<<<gen_grid(n), 256, 0, ctx->stream>>>(dst, n, seed, lo, hi - lo, row0);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_gen_uniform_small(ah_context* ctx, void* dst, int32_t byte_width, int64_t n,
                                          uint64_t seed, int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  if (byte_width == 1) gen_small_kernel<uint8_t><<<gen_grid(n), 256, 0, ctx->stream>>>((uint8_t*)dst, n, seed, row0);
  else if (byte_width == 2) gen_small_kernel<uint16_t><<<gen_grid(n), 256, 0, ctx->stream>>>((uint16_t*)dst, n, seed, row0);
  else return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported byte width %d", byte_width);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_gen_bernoulli_bits(ah_context* ctx, uint8_t* dst, int64_t n, uint64_t seed,
                                           double p_true, int64_t row0) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  double p = p_true < 0 ? 0 : (p_true > 1 ? 1 : p_true);
  uint64_t threshold = (uint64_t)(p * 9007199254740992.0);  // p * 2^53
  gen_bits_kernel<<<gen_grid((n + 63) >> 6), 256, 0, ctx->stream>>>((unsigned long long*)dst, n, seed,
                                                                    threshold, row0);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
extern "C" ah_status ah_zero_null_slots(ah_context* ctx, void* values, int32_t byte_width,
                                        const uint8_t* validity, int64_t n) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0 || !validity) return AH_OK;
  hipSetDevice(ctx->device);
  const unsigned long long* v = (const unsigned long long*)validity;
  int g = gen_grid(n);
  switch (byte_width) {
    case 1: zero_nulls_kernel<uint8_t><<<g, 256, 0, ctx->stream>>>((uint8_t*)values, v, n); break;
    case 2: zero_nulls_kernel<uint16_t><<<g, 256, 0, ctx->stream>>>((uint16_t*)values, v, n); break;
    case 4: zero_nulls_kernel<uint32_t><<<g, 256, 0, ctx->stream>>>((uint32_t*)values, v, n); break;
    case 8: zero_nulls_kernel<uint64_t><<<g, 256, 0, ctx->stream>>>((uint64_t*)values, v, n); break;
    default: return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported byte width %d", byte_width);
  }
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}

extern "C" ah_status ah_gen_iota_u32(ah_context* ctx, uint32_t* dst, int64_t n, uint32_t start) {
  ah_ctx_guard _guard(ctx);
  if (n <= 0) return AH_OK;
  hipSetDevice(ctx->device);
  gen_iota_u32_kernel<<<gen_grid(n), 256, 0, ctx->stream>>>(dst, n, start);
  AH_HIP(ctx, hipGetLastError());
  return AH_OK;
}
