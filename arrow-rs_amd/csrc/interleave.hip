// arrow_select::interleave::interleave on MI355X: out[i] = arrays[a[i]][r[i]] — `take` across several arrays.
//
// Reference: arrow-select/src/interleave.rs — argument checks :78-97, `interleave_primitive` :151-175 (values by
// (array, row) pairs; a null buffer iff some input has nulls, `Interleave::new` :118-148), out-of-range pairs panic
// through slice indexing.  Lane per output row; the per-array table (value pointer, validity view, length) sits in a
// small device buffer; a wave's validity word is a ballot; Boolean values are gathered bit by bit into a ballot word.
#include <hip/hip_runtime.h>

#include <vector>

#include "common.hpp"

namespace {

struct Src {
  const void* values;
  BitView bits;   // Boolean values
  BitView valid;  // words == nullptr: no nulls
  int64_t length;
};

template <int W> struct Elem { uint8_t b[W]; };
template <> struct Elem<1> { uint8_t v; };
template <> struct Elem<2> { uint16_t v; };
template <> struct Elem<4> { uint32_t v; };
template <> struct Elem<8> { uint64_t v; };
template <> struct alignas(16) Elem<16> { uint4 v; };
template <> struct alignas(16) Elem<32> { uint4 v[2]; };

// W == 0: Boolean
template <int W>
__global__ __launch_bounds__(256) void interleave_kernel(const Src* __restrict__ srcs, int n_src, const uint32_t* __restrict__ aidx,
                                                        const uint32_t* __restrict__ ridx, int64_t n, void* out,
                                                        unsigned long long* out_valid, unsigned long long* valid_slots,
                                                        unsigned long long* first_err) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (n + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = wave0; w < nwords; w += nwaves) {
    const int64_t i = w * 64 + lane;
    bool valid = false, bit = false;
    if (i < n) {
      const uint32_t a = aidx[i], r = ridx[i];
      if (a >= (uint32_t)n_src || (int64_t)r >= srcs[a < (uint32_t)n_src ? a : 0].length) {
        atomicMin(first_err, (unsigned long long)i);
      } else {
        const Src s = srcs[a];
        if constexpr (W == 0) bit = bv_get(s.bits, r) != 0;
        else ((Elem<(W ? W : 1)>*)out)[i] = ((const Elem<(W ? W : 1)>*)s.values)[r];
        valid = bv_get(s.valid, r) != 0;
      }
    }
    if constexpr (W == 0) {
      const unsigned long long bw = __ballot(bit);
      if (lane == 0) ((unsigned long long*)out)[w] = bw;
    }
    const unsigned long long vw = __ballot(valid);
    if (lane == 0) {
      if (out_valid) out_valid[w] = vw;
      acc += __popcll(vw);
    }
  }
  if (lane == 0 && acc) atomicAdd(&valid_slots[wave0 & 63], acc);
}

}  // namespace

extern "C" ah_status ah_interleave(ah_context* ctx, int32_t n_arrays, const ah_array_view* arrays,
                                   const ah_array_view* array_index, const ah_array_view* row_index, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out || !array_index || !row_index || (n_arrays > 0 && !arrays)) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (n_arrays <= 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "interleave requires input of at least one array");  // :78
  const ah_type t = arrays[0].type;
  for (int i = 1; i < n_arrays; ++i)
    if (arrays[i].type != t)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "It is not possible to interleave arrays of different data types (%s and %s)",
                     ah_type_name(t), ah_type_name(arrays[i].type));  // :86-92
  if (array_index->type != AH_UINT32 || row_index->type != AH_UINT32 || array_index->length != row_index->length)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "interleave indices are two UInt32 arrays of equal length (array, row)");
  const int w = ah_type_width(t);
  if (w < 0 || t == AH_UTF8_VIEW || t == AH_BINARY_VIEW) return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "interleave of %s", ah_type_name(t));
  const int64_t n = array_index->length;
  out->type = t;
  out->length = n;
  if (n == 0) return AH_OK;  // :95-97
  std::vector<Src> host((size_t)n_arrays);
  bool any_nulls = false;
  for (int i = 0; i < n_arrays; ++i) {
    int64_t nulls = 0;
    AH_TRY(ah_resolve_null_count(ctx, &arrays[i], &nulls));
    const bool has = arrays[i].validity && nulls > 0;
    any_nulls |= has;
    host[i].values = arrays[i].values;
    host[i].bits = w == 0 ? make_bitview(arrays[i].values, arrays[i].values_bit_offset) : BitView{nullptr, 0};
    host[i].valid = has ? make_bitview(arrays[i].validity, arrays[i].validity_bit_offset) : BitView{nullptr, 0};
    host[i].length = arrays[i].length;
  }
  const size_t vbytes = w ? (size_t)n * w : ah_bitmap_bytes(n), bbytes = any_nulls ? ah_bitmap_bytes(n) : 0;
  void *ov = nullptr, *ob = nullptr, *scratch = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  ah_status st = any_nulls ? ah_out_alloc(ctx, bbytes, &ob) : AH_OK;
  const size_t tbytes = sizeof(Src) * (size_t)n_arrays;
  if (st == AH_OK) st = ah_pool_alloc(ctx, tbytes + 65 * 8, &scratch);
  auto cleanup = [&](ah_status s) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    if (scratch) ah_pool_free(ctx, scratch);
    return s;
  };
  if (st != AH_OK) return cleanup(st);
  unsigned long long* slots = (unsigned long long*)scratch;  // 64 valid counters + first error position
  Src* table = (Src*)((char*)scratch + 65 * 8);
  hipError_t e = hipMemsetAsync(slots, 0, 64 * 8, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(slots + 64, 0xFF, 8, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(table, host.data(), tbytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    ah_prof_scope ps(ctx, "interleave");
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(((n + 63) / 64 + 3) / 4, 256 * 16)));
    const uint32_t* ai = (const uint32_t*)array_index->values;
    const uint32_t* ri = (const uint32_t*)row_index->values;
#define AH_IL(W) hipLaunchKernelGGL(interleave_kernel<W>, grid, dim3(256), 0, ctx->stream, table, n_arrays, ai, ri, n, ov, (unsigned long long*)ob, slots, slots + 64)
    switch (w) {
      case 0: AH_IL(0); break;
      case 1: AH_IL(1); break;
      case 2: AH_IL(2); break;
      case 4: AH_IL(4); break;
      case 8: AH_IL(8); break;
      case 16: AH_IL(16); break;
      default: AH_IL(32); break;
    }
#undef AH_IL
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, slots, 65 * 8);
  if (e != hipSuccess) return cleanup(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in interleave", hipGetErrorString(e)));
  if (ctx->pinned[64] != ~0ull) {  // slice indexing panics in the reference (`arrays[a].value(r)`)
    const int64_t pos = (int64_t)ctx->pinned[64];
    uint32_t pair[2] = {0, 0};
    hipMemcpy(&pair[0], (const uint32_t*)array_index->values + pos, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&pair[1], (const uint32_t*)row_index->values + pos, 4, hipMemcpyDeviceToHost);
    cleanup(AH_OK);
    if (pair[0] >= (uint32_t)n_arrays)
      return ah_fail(ctx, AH_PANIC, "index out of bounds: the len is %d but the index is %u", n_arrays, pair[0]);
    return ah_fail(ctx, AH_PANIC, "index out of bounds: the len is %lld but the index is %u", (long long)arrays[pair[0]].length, pair[1]);
  }
  int64_t valid = 0;
  for (int i = 0; i < 64; ++i) valid += (int64_t)ctx->pinned[i];
  ah_pool_free(ctx, scratch);
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (any_nulls) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = n - valid;
  }
  return AH_OK;
}
