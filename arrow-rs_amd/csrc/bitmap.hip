// bitmap.hip — word-parallel bitmap algebra with arbitrary bit offsets.
//
// Reference analogues: NullBuffer::union (arrow-buffer/src/buffer/null.rs:79-88,
// buffer_bin_and arrow-buffer/src/buffer/ops.rs:149), the distinct/not_distinct
// chunk formulas of arrow-ord/src/cmp.rs:325-374, bit_mask::set_bits
// (arrow-buffer/src/util/bit_mask.rs:33) and BooleanBuffer::count_set_bits.
// One thread per 64-bit OUTPUT word; inputs are funnel-shifted to the output
// alignment, so no read-modify-write and no atomics are needed.
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256) bitmap_op_kernel(int op, BitView a, BitView b, BitView c, BitView d,
                                                        int64_t len, unsigned long long* out,
                                                        unsigned long long* total) {
  int64_t nwords = (len + 63) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
    int64_t s = w << 6;
    uint64_t x = bv_fetch64(a, s, len), r;
    switch (op) {
      case BM_COPY: r = x; break;
      case BM_NOT: r = ~x; break;
      case BM_AND: r = x & bv_fetch64(b, s, len); break;
      case BM_DISTINCT_BOTH: {
        uint64_t y = bv_fetch64(b, s, len), n = bv_fetch64(c, s, len);
        r = (x ^ y) | (x & y & n);
        break;
      }
      case BM_NOT_DISTINCT_BOTH: {
        uint64_t y = bv_fetch64(b, s, len), e = bv_fetch64(c, s, len);
        r = ~(x | y) | (x & y & e);
        break;
      }
      case BM_OR: r = x | bv_fetch64(b, s, len); break;
      case BM_ANDNOT: r = x & ~bv_fetch64(b, s, len); break;
      case BM_OR_NOTB: r = x | ~bv_fetch64(b, s, len); break;
      case BM_NULLIF: r = x & ~(bv_fetch64(b, s, len) & bv_fetch64(c, s, len)); break;
      case BM_KLEENE_AND_NULLS: {
        uint64_t y = bv_fetch64(b, s, len), z = bv_fetch64(c, s, len), w4 = bv_fetch64(d, s, len);
        r = (x | (z & ~w4)) & (z | (x & ~y));
        break;
      }
      case BM_KLEENE_OR_NULLS: {
        uint64_t y = bv_fetch64(b, s, len), z = bv_fetch64(c, s, len), w4 = bv_fetch64(d, s, len);
        r = (x | (z & w4)) & (z | (x & y));
        break;
      }
      default: r = ~x | bv_fetch64(b, s, len); break;  // BM_ORNOT
    }
    int64_t rem = len - s;
    if (rem < 64) r &= (1ull << rem) - 1;
    out[w] = r;
    acc += __popcll(r);
  }
  if (total) {  // counted: the popcount accumulates in a scratch word (see ah_count_add)
    acc = wave_reduce_add64(acc);
    __shared__ unsigned long long sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ah_count_add(total, sm[0] + sm[1] + sm[2] + sm[3]);
  }
}

__global__ void __launch_bounds__(1024) bm_sum_kernel(const unsigned long long* in, int64_t n,
                                                      unsigned long long* out) {
  unsigned long long acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += in[i];
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s[16];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; i++) t += s[i];
    *out = t;
  }
}

// dst |= src bits placed at dst_off (dst range pre-zeroed); one thread per dst word.  partials: per-block counts for
// the no-wait accumulate form; total: the counted form (popcount accumulated in a scratch word)
__global__ void __launch_bounds__(256) set_bits_kernel(unsigned long long* dst, int64_t dst_off,
                                                       BitView src, int64_t len,
                                                       unsigned long long* partials, unsigned long long* total = nullptr) {
  int64_t first = dst_off >> 6, last = (dst_off + len - 1) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = first + (int64_t)blockIdx.x * 256 + threadIdx.x; w <= last;
       w += (int64_t)gridDim.x * 256) {
    int64_t s = (w << 6) - dst_off;  // source index of this word's bit 0
    uint64_t v = s >= 0 ? bv_fetch64(src, s, len) : (bv_fetch64(src, 0, len) << (-s));
    if (v) dst[w] |= v;
    acc += __popcll(v);
  }
  if (partials || total) {
    acc = wave_reduce_add64(acc);
    __shared__ unsigned long long sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long tot = sm[0] + sm[1] + sm[2] + sm[3];
      if (partials) partials[blockIdx.x] = tot;
      else ah_count_add(total, tot);
    }
  }
}

}  // namespace

ah_status ah_bitmap_op(ah_context* ctx, int op, BitView a, BitView b, BitView c, int64_t len,
                       unsigned long long* out_words, int64_t* set_bits, BitView d) {
  if (len <= 0) {
    if (set_bits) *set_bits = 0;
    return AH_OK;
  }
  int64_t nwords = (len + 63) >> 6;
  int grid = (int)std::min<int64_t>(4096, ah_ceil_div(nwords, 256));
  if (!set_bits) {
    bitmap_op_kernel<<<grid, 256, 0, ctx->stream>>>(op, a, b, c, d, len, out_words, nullptr);
    return AH_OK;
  }
  // counted: the kernel accumulates the popcount in a scratch word; one small kernel copies it to the host, zeroes it, posts
  bitmap_op_kernel<<<grid, 256, 0, ctx->stream>>>(op, a, b, c, d, len, out_words, ctx->scratch + AH_TICKET_COUNT);
  AH_HIP(ctx, ah_count_read(ctx, set_bits));
  return AH_OK;
}

// sum of `n` partial counts -> *acc += base - sum (atomic): the no-wait tail of the *_acc entry points
__global__ void __launch_bounds__(1024) bm_acc_kernel(const unsigned long long* in, int64_t n, unsigned long long base,
                                                      unsigned long long* acc) {
  unsigned long long a = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) a += in[i];
  a = wave_reduce_add64(a);
  __shared__ unsigned long long s[16];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; i++) t += s[i];
    if (base != t) atomicAdd(acc, base - t);
  }
}

// ah_bitmap_set_bits without the read-back: *nulls_acc (device) += len - popcount(copied bits); nothing waits
ah_status ah_bitmap_set_bits_acc(ah_context* ctx, uint8_t* dst, int64_t dst_bit_offset, const uint8_t* src,
                                 int64_t src_bit_offset, int64_t len, unsigned long long* nulls_acc) {
  if (len <= 0) return AH_OK;
  if (((uintptr_t)dst & 7) != 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "bitmap destination must be 8-byte aligned");
  int64_t first = dst_bit_offset >> 6, last = (dst_bit_offset + len - 1) >> 6;
  int grid = (int)std::min<int64_t>(4096, ah_ceil_div(last - first + 1, 256));
  unsigned long long* part = nullptr;
  const bool count = src != nullptr && nulls_acc != nullptr;  // a source without a null buffer appends no nulls
  if (count) AH_TRY(ah_pool_alloc(ctx, (size_t)(grid + 1) * 8, (void**)&part));
  set_bits_kernel<<<grid, 256, 0, ctx->stream>>>((unsigned long long*)dst, dst_bit_offset, make_bitview(src, src_bit_offset),
                                                 len, part);
  if (count) {
    bm_acc_kernel<<<1, 1024, 0, ctx->stream>>>(part, grid, (unsigned long long)len, nulls_acc);
    ah_pool_free(ctx, part);  // stream-ordered reuse
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "set_bits failed: %s", hipGetErrorString(e));
  return AH_OK;
}

extern "C" ah_status ah_bitmap_set_bits(ah_context* ctx, uint8_t* dst, int64_t dst_bit_offset,
                                        const uint8_t* src, int64_t src_bit_offset, int64_t len,
                                        int64_t* set_bits) {
  ah_ctx_guard _guard(ctx);
  if (set_bits) *set_bits = 0;
  if (len <= 0) return AH_OK;
  if (((uintptr_t)dst & 7) != 0)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "bitmap destination must be 8-byte aligned");
  hipSetDevice(ctx->device);
  int64_t first = dst_bit_offset >> 6, last = (dst_bit_offset + len - 1) >> 6;
  int grid = (int)std::min<int64_t>(4096, ah_ceil_div(last - first + 1, 256));
  if (!set_bits) {
    set_bits_kernel<<<grid, 256, 0, ctx->stream>>>((unsigned long long*)dst, dst_bit_offset, make_bitview(src, src_bit_offset), len, nullptr);
    return AH_OK;
  }
  set_bits_kernel<<<grid, 256, 0, ctx->stream>>>((unsigned long long*)dst, dst_bit_offset, make_bitview(src, src_bit_offset), len, nullptr,
                                                 ctx->scratch + AH_TICKET_COUNT);
  AH_HIP(ctx, ah_count_read(ctx, set_bits));
  return AH_OK;
}
