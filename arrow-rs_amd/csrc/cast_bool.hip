// Boolean <-> numeric casts (arrow-cast/src/cast/mod.rs:1243-1290).
//   numeric -> Boolean: `numeric_to_bool_cast` :2661-2676 — `value != T::default()` (so NaN is true and -0.0 is
//     false), null rows false; the BooleanBuilder yields a null buffer only if a null was appended.
//   Boolean -> numeric: `bool_to_numeric_cast` :2704-2721 — true -> 1, false -> 0, null rows default;
//     `from_trusted_len_iter` always yields a null buffer.
// Lane per row: a wave's 64 results / inputs are one bitmap word (ballot / funnel-shift fetch).
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void num_to_bool_kernel(const T* in, BitView valid, int64_t len,
                                                         unsigned long long* out_vals, unsigned long long* out_valid,
                                                         unsigned long long* valid_slots) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (len + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = wave0; w < nwords; w += nwaves) {
    const int64_t row = w * 64 + lane;
    const unsigned long long vw = bv_fetch64(valid, w * 64, len);
    const bool nz = row < len && in[row] != T{};
    const unsigned long long bits = __ballot(nz) & vw;
    if (lane == 0) {
      out_vals[w] = bits;
      if (out_valid) out_valid[w] = vw;
      acc += __popcll(vw);
    }
  }
  if (lane == 0 && acc) atomicAdd(&valid_slots[wave0 & 63], acc);
}

template <typename T>
__global__ __launch_bounds__(256) void bool_to_num_kernel(BitView bits, BitView valid, int64_t len, T* out,
                                                         unsigned long long* out_valid, unsigned long long* valid_slots) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (len + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = wave0; w < nwords; w += nwaves) {
    const int64_t row = w * 64 + lane;
    const unsigned long long vw = bv_fetch64(valid, w * 64, len);
    const unsigned long long bw = bv_fetch64(bits, w * 64, len) & vw;
    if (row < len) out[row] = ((bw >> lane) & 1ull) ? (T)1 : T{};
    if (lane == 0) {
      out_valid[w] = vw;
      acc += __popcll(vw);
    }
  }
  if (lane == 0 && acc) atomicAdd(&valid_slots[wave0 & 63], acc);
}

template <typename F>
ah_status with_slots(ah_context* ctx, F&& launch, int64_t* valid) {
  void* slots = nullptr;
  AH_TRY(ah_pool_alloc(ctx, 64 * 8, &slots));
  hipError_t e = hipMemsetAsync(slots, 0, 64 * 8, ctx->stream);
  if (e == hipSuccess) {
    launch((unsigned long long*)slots);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, slots, 64 * 8);
  ah_pool_free(ctx, slots);
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in Boolean cast", hipGetErrorString(e));
  *valid = 0;
  for (int i = 0; i < 64; ++i) *valid += (int64_t)ctx->pinned[i];
  return AH_OK;
}

}  // namespace

ah_status ah_cast_bool(ah_context* ctx, const ah_array_view* v, ah_type to, ah_array_out* out) {
  const ah_type from = v->type;
  const int64_t n = v->length;
  out->type = to;
  out->length = n;
  if (n == 0) return AH_OK;
  int64_t nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, v, &nulls));
  const bool has_nulls = v->validity && nulls > 0;
  const BitView valid = has_nulls ? make_bitview(v->validity, v->validity_bit_offset) : BitView{nullptr, 0};
  const size_t bbytes = ah_bitmap_bytes(n);
  const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(((n + 63) / 64 + 3) / 4, 256 * 16)));
  int64_t nvalid = 0;
  ah_prof_scope ps(ctx, "cast_bool");
  if (to == AH_BOOL) {
    void *ov = nullptr, *ob = nullptr;
    AH_TRY(ah_out_alloc(ctx, bbytes, &ov));
    ah_status st = has_nulls ? ah_out_alloc(ctx, bbytes, &ob) : AH_OK;
    if (st == AH_OK)
      st = with_slots(ctx, [&](unsigned long long* slots) {
        switch (from) {
#define AH_N2B(TAG, T) case TAG: hipLaunchKernelGGL(num_to_bool_kernel<T>, grid, dim3(256), 0, ctx->stream, (const T*)v->values, valid, n, (unsigned long long*)ov, (unsigned long long*)ob, slots); break;
          AH_N2B(AH_INT8, int8_t) AH_N2B(AH_INT16, int16_t) AH_N2B(AH_INT32, int32_t) AH_N2B(AH_INT64, int64_t)
          AH_N2B(AH_UINT8, uint8_t) AH_N2B(AH_UINT16, uint16_t) AH_N2B(AH_UINT32, uint32_t) AH_N2B(AH_UINT64, uint64_t)
          AH_N2B(AH_FLOAT32, float) AH_N2B(AH_FLOAT64, double)
#undef AH_N2B
          default: break;
        }
      }, &nvalid);
    if (st != AH_OK) {
      ah_out_free(ctx, ov, bbytes);
      ah_out_free(ctx, ob, bbytes);
      return st;
    }
    out->values = ov;
    out->values_bytes = (int64_t)bbytes;
    if (has_nulls) {
      out->validity = (uint8_t*)ob;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = n - nvalid;
    }
    return AH_OK;
  }
  const int w = ah_type_width(to);
  const size_t vbytes = (size_t)n * w;
  void *ov = nullptr, *ob = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  ah_status st = ah_out_alloc(ctx, bbytes, &ob);
  const BitView bits = make_bitview(v->values, v->values_bit_offset);
  if (st == AH_OK)
    st = with_slots(ctx, [&](unsigned long long* slots) {
      switch (to) {
#define AH_B2N(TAG, T) case TAG: hipLaunchKernelGGL(bool_to_num_kernel<T>, grid, dim3(256), 0, ctx->stream, bits, valid, n, (T*)ov, (unsigned long long*)ob, slots); break;
        AH_B2N(AH_INT8, int8_t) AH_B2N(AH_INT16, int16_t) AH_B2N(AH_INT32, int32_t) AH_B2N(AH_INT64, int64_t)
        AH_B2N(AH_UINT8, uint8_t) AH_B2N(AH_UINT16, uint16_t) AH_B2N(AH_UINT32, uint32_t) AH_B2N(AH_UINT64, uint64_t)
        AH_B2N(AH_FLOAT32, float) AH_B2N(AH_FLOAT64, double)
#undef AH_B2N
        default: break;
      }
    }, &nvalid);
  if (st != AH_OK) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    return st;
  }
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  out->validity = (uint8_t*)ob;  // always present (from_trusted_len_iter)
  out->validity_bytes = (int64_t)bbytes;
  out->null_count = n - nvalid;
  return AH_OK;
}
