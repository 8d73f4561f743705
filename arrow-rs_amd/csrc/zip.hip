// arrow_select::zip::zip on MI355X: out[i] = mask[i] ? truthy[i] : falsy[i], either side optionally a scalar.
//
// Reference: arrow-select/src/zip.rs — argument checks :104-140, `zip_impl` :148-200 (mask nulls select `falsy`:
// `maybe_prep_null_mask_filter`; `MutableArrayData` gives the result a null buffer iff an input has nulls),
// scalar/scalar paths :388-470 (same logical result).
//
// One streaming pass, lane per row: the mask word of a wave's 64 rows is a wave-uniform 64-bit value, each lane loads
// ONLY the selected side (so a selective mask reads one input, not two) and the output validity word is
// `__ballot(valid)`.  Algorithmic bytes per row for 8-byte values: 8 (selected side) + 8 (out) + 3 bits in + 1 bit out.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {

template <int W> struct Vec { uint8_t b[W]; };
template <> struct Vec<1> { uint8_t v; };
template <> struct Vec<2> { uint16_t v; };
template <> struct Vec<4> { uint32_t v; };
template <> struct Vec<8> { uint64_t v; };
template <> struct alignas(16) Vec<16> { uint4 v; };
template <> struct alignas(16) Vec<32> { uint4 v[2]; };

struct ZipArgs {
  BitView mask, mask_valid;  // mask_valid.words == nullptr: no nulls
  const void *t, *f;
  BitView tbits, fbits;  // Boolean values only
  BitView tv, fv;  // validity (nullptr words = all valid)
  int t_scalar, f_scalar;
  void* out;
  unsigned long long* out_valid;  // nullptr: the result carries no null buffer
  unsigned long long* valid_slots;  // 64 counters
  int64_t len;
};

template <int W>
__global__ __launch_bounds__(256) void zip_kernel(ZipArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (a.len + 63) / 64;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  unsigned long long valid_acc = 0;
  for (int64_t w = wave0; w < nwords; w += nwaves) {
    const int64_t row = w * 64 + lane;
    unsigned long long m = bv_fetch64(a.mask, w * 64, a.len);
    if (a.mask_valid.words) m &= bv_fetch64(a.mask_valid, w * 64, a.len);
    const bool in = row < a.len;
    const bool sel = (m >> lane) & 1ull;
    const int64_t j = (sel ? a.t_scalar : a.f_scalar) ? 0 : row;
    bool valid = false;
    if (in) {
      const Vec<W>* src = (const Vec<W>*)(sel ? a.t : a.f);
      ((Vec<W>*)a.out)[row] = src[j];
      valid = bv_get(sel ? a.tv : a.fv, j) != 0;
    }
    const unsigned long long vw = __ballot(valid);
    if (a.out_valid && lane == 0) a.out_valid[w] = vw;
    if (lane == 0) valid_acc += __popcll(vw);
  }
  if (lane == 0 && valid_acc) atomicAdd(&a.valid_slots[(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6) & 63], valid_acc);
}

// Boolean values: whole words at a time
__global__ __launch_bounds__(256) void zip_bool_kernel(ZipArgs a) {
  const int64_t nwords = (a.len + 63) / 64;
  unsigned long long valid_acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * 256) {
    unsigned long long m = bv_fetch64(a.mask, w * 64, a.len);
    if (a.mask_valid.words) m &= bv_fetch64(a.mask_valid, w * 64, a.len);
    auto side = [&](BitView bits, int scalar, int64_t n) -> unsigned long long {
      if (!bits.words) return ~0ull;
      if (scalar) return bv_get(bits, 0) ? ~0ull : 0ull;
      return bv_fetch64(bits, w * 64, n);
    };
    const unsigned long long tvals = side(a.tbits, a.t_scalar, a.len), fvals = side(a.fbits, a.f_scalar, a.len);
    const unsigned long long tvld = side(a.tv, a.t_scalar, a.len), fvld = side(a.fv, a.f_scalar, a.len);
    const int64_t rem = a.len - w * 64;
    const unsigned long long inmask = rem >= 64 ? ~0ull : ((1ull << rem) - 1);
    ((unsigned long long*)a.out)[w] = ((m & tvals) | (~m & fvals)) & inmask;
    const unsigned long long vw = ((m & tvld) | (~m & fvld)) & inmask;
    if (a.out_valid) a.out_valid[w] = vw;
    valid_acc += __popcll(vw);
  }
  valid_acc = wave_reduce_add64(valid_acc);
  if ((threadIdx.x & 63) == 0 && valid_acc) atomicAdd(&a.valid_slots[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 63], valid_acc);
}

}  // namespace

extern "C" ah_status ah_zip(ah_context* ctx, const ah_array_view* mask, const ah_array_view* truthy, int32_t t_scalar,
                            const ah_array_view* falsy, int32_t f_scalar, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !mask || !truthy || !falsy || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (mask->type != AH_BOOL) return ah_fail(ctx, AH_INVALID_ARGUMENT, "zip mask must be Boolean, got %s", ah_type_name(mask->type));
  if (truthy->type != falsy->type) return ah_fail(ctx, AH_INVALID_ARGUMENT, "arguments need to have the same data type");  // :115
  if (t_scalar && truthy->length != 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar arrays must have 1 element");
  if (!t_scalar && truthy->length != mask->length) return ah_fail(ctx, AH_INVALID_ARGUMENT, "all arrays should have the same length");
  if (f_scalar && falsy->length != 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "scalar arrays must have 1 element");
  if (!f_scalar && falsy->length != mask->length) return ah_fail(ctx, AH_INVALID_ARGUMENT, "all arrays should have the same length");
  const ah_type t = truthy->type;
  const int w = ah_type_width(t);
  if (w < 0) return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "zip of %s", ah_type_name(t));
  const int64_t n = mask->length;
  out->type = t;
  out->length = n;
  if (n == 0) return AH_OK;
  int64_t tn = 0, fn = 0, mn = 0;
  AH_TRY(ah_resolve_null_count(ctx, truthy, &tn));
  AH_TRY(ah_resolve_null_count(ctx, falsy, &fn));
  AH_TRY(ah_resolve_null_count(ctx, mask, &mn));
  const bool any_nulls = tn > 0 || fn > 0;
  const size_t vbytes = w ? (size_t)n * w : ah_bitmap_bytes(n), bbytes = any_nulls ? ah_bitmap_bytes(n) : 0;
  void *ov = nullptr, *ob = nullptr, *slots = nullptr;
  AH_TRY(ah_out_alloc(ctx, vbytes, &ov));
  ah_status st = any_nulls ? ah_out_alloc(ctx, bbytes, &ob) : AH_OK;
  if (st == AH_OK) st = ah_pool_alloc(ctx, 64 * 8, &slots);
  auto cleanup = [&](ah_status s) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    if (slots) ah_pool_free(ctx, slots);
    return s;
  };
  if (st != AH_OK) return cleanup(st);
  ZipArgs a;
  a.mask = make_bitview(mask->values, mask->values_bit_offset);
  a.mask_valid = (mask->validity && mn > 0) ? make_bitview(mask->validity, mask->validity_bit_offset) : BitView{nullptr, 0};
  a.t = truthy->values;
  a.f = falsy->values;
  a.tv = tn > 0 ? make_bitview(truthy->validity, truthy->validity_bit_offset) : BitView{nullptr, 0};
  a.fv = fn > 0 ? make_bitview(falsy->validity, falsy->validity_bit_offset) : BitView{nullptr, 0};
  a.tbits = a.fbits = BitView{nullptr, 0};
  a.t_scalar = t_scalar;
  a.f_scalar = f_scalar;
  a.out = ov;
  a.out_valid = (unsigned long long*)ob;
  a.valid_slots = (unsigned long long*)slots;
  a.len = n;
  hipError_t e = hipMemsetAsync(slots, 0, 64 * 8, ctx->stream);
  if (e == hipSuccess) {
    ah_prof_scope ps(ctx, "zip");
    const int64_t nwords = (n + 63) / 64;
    if (w == 0) {
      a.tbits = make_bitview(truthy->values, truthy->values_bit_offset);  // Boolean: values are bit streams
      a.fbits = make_bitview(falsy->values, falsy->values_bit_offset);
      hipLaunchKernelGGL(zip_bool_kernel, dim3((unsigned)std::min<int64_t>((nwords + 255) / 256, 4096)), dim3(256), 0,
                         ctx->stream, a);
    } else {
      const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((nwords + 3) / 4, 256 * 16)));
      switch (w) {
        case 1: hipLaunchKernelGGL(zip_kernel<1>, grid, dim3(256), 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(zip_kernel<2>, grid, dim3(256), 0, ctx->stream, a); break;
        case 4: hipLaunchKernelGGL(zip_kernel<4>, grid, dim3(256), 0, ctx->stream, a); break;
        case 8: hipLaunchKernelGGL(zip_kernel<8>, grid, dim3(256), 0, ctx->stream, a); break;
        case 16: hipLaunchKernelGGL(zip_kernel<16>, grid, dim3(256), 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(zip_kernel<32>, grid, dim3(256), 0, ctx->stream, a); break;
      }
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, slots, 64 * 8);
  if (e != hipSuccess) return cleanup(ah_fail(ctx, AH_HIP_ERROR, "HIP error %s in zip", hipGetErrorString(e)));
  int64_t valid = 0;
  for (int i = 0; i < 64; ++i) valid += (int64_t)ctx->pinned[i];
  ah_pool_free(ctx, slots);
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (any_nulls) {
    out->validity = (uint8_t*)ob;
    out->validity_bytes = (int64_t)bbytes;
    out->null_count = n - valid;
  }
  return AH_OK;
}
