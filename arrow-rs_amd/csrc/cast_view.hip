// cast_view.hip — the `-> Utf8View` arms of arrow_cast::cast.
//
// Reference: arrow-cast/src/cast/mod.rs:1546-1548 `(from_type, Utf8View) if from_type.is_primitive() =>
// value_to_string_view` (cast/string.rs:41-63: the SAME ArrayFormatter as value_to_string, appended to a
// StringViewBuilder), and :1302 / :1432 `(Utf8 | LargeUtf8, Utf8View)`.  Goldens: test_cast_float_to_utf8view
// (mod.rs:4857-4876) and test_cast_int_to_utf8view (:4827-4853).
//
// A view is 16 bytes (arrow-array/src/array/byte_view_array.rs:42-72, arrow-data/src/byte_view.rs): u32 length, then
// either <= 12 bytes of inline text (zero padded: equality of inline views is a 16-byte compare) or {4-byte prefix,
// u32 buffer index, u32 offset} into a variadic data buffer.  The reference's builder cuts its data into growing
// blocks; block structure is not part of the logical value.  Here: ONE data buffer (index 0) that holds exactly the
// strings longer than 12 bytes, back to back in row order — for a Float64 column of the cast chain that is the ~1 % of
// rows printed in exponent form; most casts produce no data buffer at all.
//
// Numeric sources go through the X2 passes first (ah_cast_to_string -> LargeUtf8: offsets + text, csrc/cast_string.hip),
// then — like a Utf8 / LargeUtf8 source directly — through two launches: per-block byte totals of the long strings
// (+ one scan launch), then the views and the long strings' bytes.  The result's data buffer travels in the
// ah_array_out's `offsets` / `offsets_bytes` fields (a view array has no offsets; see include/arrow_hip.h).
#include "common.hpp"
#include "scan_chain.hpp"

ah_status ah_cast_to_string(ah_context* ctx, const ah_array_view* values, ah_type to_type, ah_array_out* out);

namespace {

constexpr int VT = 256;          // threads per block
constexpr int VR = 4;            // consecutive rows per thread
constexpr int VB = VT * VR;      // rows per block
constexpr int INLINE_MAX = 12;   // MAX_INLINE_VIEW_LEN (arrow-data/src/byte_view.rs:26)

__device__ __forceinline__ int block_sum(int v, int* s_wave) {
  v = wave_scan_incl(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) s_wave[wave] = v;
  __syncthreads();
  int tot = 0;
  for (int w = 0; w < VT / 64; ++w) tot += s_wave[w];
  return tot;
}

// pass 1: bytes of the long (> 12) strings of valid rows, per block of VB rows
template <typename OFF>
__global__ void __launch_bounds__(VT) view_long_bytes_kernel(const OFF* offs, BitView valid, int64_t len, uint32_t* block_bytes) {
  __shared__ unsigned long long s_tot[VT / 64];
  const int64_t r0 = (int64_t)blockIdx.x * VB + (int64_t)threadIdx.x * VR;
  // 64-bit sums, SATURATED into the u32 block total: a LargeUtf8 source with more than 2-4 GiB of long strings inside one
  // block's 1024 rows wrapped here BEFORE the caller's `long_bytes > INT32_MAX` test, which could then pass on a wrapped
  // total; a saturated block keeps the grand total above the limit
  unsigned long long mine = 0;
  OFF o[VR + 1];
#pragma unroll
  for (int k = 0; k <= VR; ++k) o[k] = r0 + k <= len ? offs[r0 + k] : OFF(0);
#pragma unroll
  for (int k = 0; k < VR; ++k)
    if (r0 + k < len) {
      const int64_t l = (int64_t)o[k + 1] - (int64_t)o[k];
      if (l > INLINE_MAX && bv_get(valid, r0 + k)) mine += (unsigned long long)l;
    }
  mine = wave_reduce_add64(mine);
  if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    for (int w = 0; w < VT / 64; ++w) tot += s_tot[w];
    block_bytes[blockIdx.x] = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
  }
}

// the first 12 bytes at `p` (which may have any alignment) as three dwords; bytes at index >= l are zero
__device__ __forceinline__ void load12(const uint8_t* data, int64_t base, int l, int64_t data_bytes, uint32_t d[3]) {
  // four aligned dwords cover bytes [a, a + 12] whatever a & 3 is — the first of them starts up to 3 bytes BEFORE a, so the
  // fast path needs a 4-aligned buffer start or base >= 3 (an unaligned / sliced values buffer read before its start: ADVICE r04)
  if (base + 16 <= data_bytes && ((((uintptr_t)data) & 3) == 0 || base >= 3)) {
    const uintptr_t a = (uintptr_t)(data + base);
    const uint32_t* p = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
    const int sh = (int)(a & 3) * 8;
    d[0] = __funnelshift_r(w0, w1, sh);
    d[1] = __funnelshift_r(w1, w2, sh);
    d[2] = __funnelshift_r(w2, w3, sh);
  } else {  // the tail of the buffer: byte loads, never past its end
    d[0] = d[1] = d[2] = 0;
    for (int k = 0; k < 12 && k < l; ++k) d[k >> 2] |= (uint32_t)data[base + k] << ((k & 3) * 8);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int keep = l - 4 * k;
    if (keep <= 0) d[k] = 0;
    else if (keep < 4) d[k] &= (1u << (8 * keep)) - 1u;
  }
}

// pass 2: the views, and the long strings' bytes at block_prefix[block] + their rank inside the block
template <typename OFF>
__global__ void __launch_bounds__(VT) view_write_kernel(const OFF* offs, const uint8_t* data, int64_t data_bytes, BitView valid, int64_t len,
                                                        const unsigned long long* block_prefix, uint4* views, uint8_t* out_data) {
  __shared__ int s_wave[VT / 64];
  const int64_t r0 = (int64_t)blockIdx.x * VB + (int64_t)threadIdx.x * VR;
  OFF o[VR + 1];
#pragma unroll
  for (int k = 0; k <= VR; ++k) o[k] = r0 + k <= len ? offs[r0 + k] : OFF(0);
  int l[VR];
  bool ok[VR];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < VR; ++k) {
    ok[k] = r0 + k < len && bv_get(valid, r0 + k);
    l[k] = r0 + k < len ? (int)((int64_t)o[k + 1] - (int64_t)o[k]) : 0;
    if (ok[k] && l[k] > INLINE_MAX) mine += l[k];
  }
  // exclusive position of this thread's long bytes inside the block
  int incl = wave_scan_incl(mine);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  unsigned long long pos = block_prefix[blockIdx.x] + (unsigned long long)(base + incl - mine);
#pragma unroll
  for (int k = 0; k < VR; ++k) {
    if (r0 + k >= len) break;
    uint4 v = {0u, 0u, 0u, 0u};  // append_null: an all-zero view (GenericByteViewBuilder::append_null)
    if (ok[k]) {
      const int64_t b = (int64_t)o[k];
      uint32_t d[3];
      load12(data, b, l[k], data_bytes, d);
      v.x = (uint32_t)l[k];
      if (l[k] <= INLINE_MAX) {
        v.y = d[0], v.z = d[1], v.w = d[2];
      } else {
        v.y = d[0];              // the 4-byte prefix
        v.z = 0u;                // buffer index
        v.w = (uint32_t)pos;     // offset inside it
        uint8_t* dst = out_data + pos;
        const uint8_t* src = data + b;
        for (int j = 0; j < l[k]; ++j) dst[j] = src[j];
        pos += (unsigned long long)l[k];
      }
    }
    views[r0 + k] = v;
  }
}

template <typename OFF>
ah_status build_views(ah_context* ctx, const OFF* offs, const uint8_t* data, BitView valid, int64_t len, ah_array_out* out) {
  const int64_t nblocks = ah_ceil_div(len, (int64_t)VB);
  const size_t b_bytes = ((size_t)nblocks * 4 + 255) & ~(size_t)255, b_pre = ((size_t)nblocks * 8 + 255) & ~(size_t)255;
  char* scratch = nullptr;
  AH_TRY(ah_pool_alloc(ctx, b_bytes + b_pre + 256 + (size_t)AH_SCAN_CHAIN_MAX_BLOCKS * 8, (void**)&scratch));
  uint32_t* block_bytes = (uint32_t*)scratch;
  unsigned long long* prefix = (unsigned long long*)(scratch + b_bytes);
  unsigned long long* total = (unsigned long long*)(scratch + b_bytes + b_pre);  // two words
  unsigned long long* scan_slots = (unsigned long long*)(scratch + b_bytes + b_pre + 256);
  {
    ah_prof_scope ps(ctx, "cast_view_len");
    view_long_bytes_kernel<OFF><<<(unsigned)nblocks, VT, 0, ctx->stream>>>(offs, valid, len, block_bytes);
    // block totals (u32) -> block bases, total[0] = bytes of all long strings, total[1] = the last offset (the extent of the
    // source text the views may read): one launch of chained workgroups (scan_chain.hpp)
    hipMemsetAsync(scan_slots, 0, (size_t)AH_SCAN_CHAIN_MAX_BLOCKS * 8, ctx->stream);
    ScanChainExtra extra;
    extra.last_off = offs + len;
    extra.last_wide = sizeof(OFF) == 8 ? 1 : 0;
    ah_launch_chained_scan<uint32_t>(ctx, block_bytes, nblocks, prefix, total, extra, scan_slots);
  }
  hipError_t e = ah_d2h_wait(ctx, ctx->pinned, total, 16);  // both words in the call's one wait before the allocation
  if (e != hipSuccess) {
    ah_pool_free(ctx, scratch);
    return ah_fail(ctx, AH_HIP_ERROR, "string view length pass failed: %s", hipGetErrorString(e));
  }
  const uint64_t long_bytes = ctx->pinned[0];
  const int64_t data_bytes = (int64_t)ctx->pinned[1];
  if (long_bytes > (uint64_t)INT32_MAX) {  // a view's offset is a u32 the reference reads as i32-sized blocks (<= 2 GiB each)
    ah_pool_free(ctx, scratch);
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "Utf8View result with %llu bytes of out-of-line text (more than one 2 GiB data buffer)",
                   (unsigned long long)long_bytes);
  }
  void* views = nullptr;
  void* od = nullptr;
  const size_t vbytes = (size_t)len * 16;
  ah_status st = ah_out_alloc(ctx, vbytes, &views);
  if (st == AH_OK && long_bytes > 0) st = ah_out_alloc(ctx, (size_t)long_bytes, &od);
  if (st == AH_OK) {
    ah_prof_scope ps(ctx, "cast_view_write");
    view_write_kernel<OFF><<<(unsigned)nblocks, VT, 0, ctx->stream>>>(offs, data, data_bytes, valid, len, prefix, (uint4*)views, (uint8_t*)od);
    e = hipGetLastError();
    if (e == hipSuccess) e = ah_stream_wait(ctx);
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "string view write pass failed: %s", hipGetErrorString(e));
  }
  ah_pool_free(ctx, scratch);
  if (st != AH_OK) {
    ah_out_free(ctx, views, vbytes);
    ah_out_free(ctx, od, (size_t)long_bytes);
    return st;
  }
  out->values = views;
  out->values_bytes = (int64_t)vbytes;
  out->offsets = od;  // the ONE variadic data buffer (nullptr: every string is inline)
  out->offsets_bytes = (int64_t)long_bytes;
  return AH_OK;
}

}  // namespace

// numeric | Utf8 | LargeUtf8  ->  Utf8View
ah_status ah_cast_to_string_view(ah_context* ctx, const ah_array_view* values, ah_array_out* out) {
  const int64_t len = values->length;
  out->type = AH_UTF8_VIEW;
  out->length = len;
  if (len == 0) return AH_OK;
  const bool from_text = values->type == AH_UTF8 || values->type == AH_LARGE_UTF8;
  int64_t nulls = 0;
  AH_TRY(ah_resolve_null_count(ctx, values, &nulls));
  ah_array_out tmp;
  ah_out_init(&tmp);
  const void* offs = values->offsets;
  const uint8_t* data = (const uint8_t*)values->values;
  bool large = values->type == AH_LARGE_UTF8;
  if (!from_text) {  // value_to_string_view: the formatter of value_to_string (cast/string.rs:21 vs :41)
    AH_TRY(ah_cast_to_string(ctx, values, AH_LARGE_UTF8, &tmp));
    offs = tmp.offsets;
    data = (const uint8_t*)tmp.values;
    large = true;
  }
  const BitView valid = (values->validity && nulls > 0) ? make_bitview(values->validity, values->validity_bit_offset) : BitView{nullptr, 0};
  ah_status st = large ? build_views<int64_t>(ctx, (const int64_t*)offs, data, valid, len, out)
                       : build_views<int32_t>(ctx, (const int32_t*)offs, data, valid, len, out);
  if (st == AH_OK && nulls > 0) {  // the null buffer is the source's (cloned)
    void* nb = nullptr;
    const size_t bbytes = ah_bitmap_bytes(len);
    st = ah_out_alloc(ctx, bbytes, &nb);
    if (st == AH_OK) st = ah_bitmap_op(ctx, BM_COPY, valid, BitView{nullptr, 0}, BitView{nullptr, 0}, len, (unsigned long long*)nb, nullptr);
    if (st == AH_OK) {
      hipError_t e = ah_stream_wait(ctx);
      if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "string view validity copy failed: %s", hipGetErrorString(e));
    }
    if (st == AH_OK) {
      out->validity = (uint8_t*)nb;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = nulls;
    } else {
      ah_out_free(ctx, nb, bbytes);
    }
  }
  ah_array_release(ctx, &tmp);
  if (st != AH_OK) {
    const ah_type t = out->type;
    ah_array_release(ctx, out);
    out->type = t;
  }
  return st;
}
