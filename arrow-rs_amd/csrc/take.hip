// take.hip — arrow_select::take on MI355X.
//
// Reference path (arrow-select/src/take.rs): take :89 -> check_bounds :167
// (optional) -> ToIndices :1030-1084 -> take_impl :212 -> take_primitive :405
// { take_native :432, take_nulls :418 { take_bits :459 } } / take_boolean :489.
//
// MI355X design: one gather kernel.  A wave owns 256 consecutive output rows as
// 4 x 64; index loads and value stores are fully coalesced, the 4 gathers per
// lane are independent.  Calibrated on MI355X (profiles/r02_take_ablation.md): every
// random 8-byte gather is ONE 128-byte L2 line fill (TCC_EA0_RDREQ_128B = 1.03e8 per
// 1e8 indices) and the validity-bit gather a second one (1.00e8, partly served by the
// Infinity Cache): 26.6 GB moved per launch = 6.8 TB/s, 0.85 of the HBM peak.  The
// kernel is bound by line-fill bandwidth: 8 gathers in flight per lane, sc0 / sc1 /
// nt load policies, more resident waves, splitting value / bit gathers, and a
// radix-partitioned gather + un-permute were all measured and none helps.
// The output validity word for each group of 64 rows is one __ballot of
// (index-valid & values-valid[idx]) — the reference's collect_bool
// (arrow-buffer/src/buffer/mutable.rs:761-791) for free.  Out-of-bounds rows are
// reported through a device atomicMin on the first offending position so the
// host can reproduce the reference's panic / error text exactly.
#include "common.hpp"

#include <type_traits>

namespace {

// 16- and 32-byte natives as clang vector types: first-class values the optimizer keeps in registers (as structs of
// four dwords the per-thread arrays of them stayed in scratch: 144 bytes per thread in the Decimal128 / i256 variants)
typedef uint32_t E16 __attribute__((ext_vector_type(4)));
typedef uint32_t E32 __attribute__((ext_vector_type(8), aligned(16)));  // i256 buffers are only 16-byte aligned
template <int W> struct Elem;
template <> struct Elem<1> { using type = uint8_t; };
template <> struct Elem<2> { using type = uint16_t; };
template <> struct Elem<4> { using type = uint32_t; };
template <> struct Elem<8> { using type = uint64_t; };
template <> struct Elem<16> { using type = E16; };
template <> struct Elem<32> { using type = E32; };

// ToIndices (take.rs:1030-1084): i32/i64 reinterpret, 8/16-bit `as u32`.
__device__ __forceinline__ uint64_t to_index(uint8_t v) { return v; }
__device__ __forceinline__ uint64_t to_index(uint16_t v) { return v; }
__device__ __forceinline__ uint64_t to_index(uint32_t v) { return v; }
__device__ __forceinline__ uint64_t to_index(uint64_t v) { return v; }
__device__ __forceinline__ uint64_t to_index(int8_t v) { return (uint32_t)(int32_t)v; }
__device__ __forceinline__ uint64_t to_index(int16_t v) { return (uint32_t)(int32_t)v; }
__device__ __forceinline__ uint64_t to_index(int32_t v) { return (uint32_t)v; }
__device__ __forceinline__ uint64_t to_index(int64_t v) { return (uint64_t)v; }

struct TakeArgs {
  const void* values;   // W != 0: elements; W == 0: unused
  BitView vbits;        // W == 0: Boolean value bits
  BitView vvalid;       // values validity (words == nullptr: none)
  int64_t values_len;
  const void* indices;
  BitView ivalid;       // index validity (words == nullptr: none)
  int64_t n;
  void* out_values;                 // W != 0: elements; W == 0: u64 words of value bits
  unsigned long long* out_valid;    // nullptr when no validity is produced
  unsigned long long* block_valid;  // per-block count of valid output rows
  unsigned long long* first_oob;    // atomicMin of first out-of-bounds position
};

// 105 SGPRs = 6 workgroups per CU; capping them at 80 (8 per CU) changes nothing (3.95 vs 3.98 ms, r02): the
// gather is bound by 128-byte line fills (header), not by the number of waves in flight.
template <int W, typename IDX, bool OUT_VALID, int KU>
__global__ void __launch_bounds__(256) take_kernel(TakeArgs a) {
  using ET = typename Elem<W == 0 ? 1 : W>::type;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const IDX* idx = (const IDX*)a.indices;
  const ET* vals = (const ET*)a.values;
  ET* out = (ET*)a.out_values;
  unsigned long long* out_bits = (unsigned long long*)a.out_values;
  unsigned long long nvalid = 0;
  unsigned long long oob = ~0ull;

  for (int64_t base = (int64_t)blockIdx.x * (256 * KU); base < a.n; base += (int64_t)gridDim.x * (256 * KU)) {
    const int64_t wbase = base + wave * (64 * KU);
    uint64_t ix[KU];
    bool live[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      int64_t i = wbase + k * 64 + lane;
      live[k] = i < a.n;
      ix[k] = live[k] ? to_index(idx[i]) : 0;
    }
    uint64_t iv[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) iv[k] = bv_fetch64(a.ivalid, wbase + k * 64, a.n);

    ET v[KU];
    int vb[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      bool ivalid = (iv[k] >> lane) & 1;
      bool inb = ix[k] < (uint64_t)a.values_len;
      // take_native :432-457: in-bounds always gathers (even under a null
      // index); OOB under a null index -> T::default(); OOB under a valid
      // index -> panic (reported through first_oob).
      if constexpr (W != 0) {
        ET z{};
        v[k] = z;
        if (live[k] && inb) v[k] = vals[ix[k]];
      }
      if (live[k] && !inb && ivalid) {
        unsigned long long pos = (unsigned long long)(wbase + k * 64 + lane);
        oob = pos < oob ? pos : oob;
      }
      int bit = 1;
      if constexpr (W == 0) bit = (live[k] && inb && ivalid) ? bv_get(a.vbits, (int64_t)ix[k]) : 0;
      int valid = ivalid ? 1 : 0;
      if (a.vvalid.words) valid = (live[k] && inb && ivalid) ? bv_get(a.vvalid, (int64_t)ix[k]) : 0;
      if constexpr (W == 0) v[k] = (ET)bit;
      vb[k] = live[k] ? valid : 0;
    }
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      int64_t i = wbase + k * 64 + lane;
      if constexpr (W != 0) {
        if (live[k]) out[i] = v[k];
      } else {
        // take_bits :459-486: bit set only for valid indices whose source bit is set
        unsigned long long w = __ballot(live[k] && v[k]);
        if (lane == 0 && (wbase + k * 64) < a.n) out_bits[(wbase + k * 64) >> 6] = w;
      }
      if constexpr (OUT_VALID) {
        unsigned long long w = __ballot(vb[k]);
        if (lane == 0 && (wbase + k * 64) < a.n) {
          a.out_valid[(wbase + k * 64) >> 6] = w;
          nvalid += __popcll(w);
        }
      }
    }
  }
  // first OOB position: wave min, then one atomic per wave that saw one
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(oob, o, 64);
    oob = other < oob ? other : oob;
  }
  if (lane == 0 && oob != ~0ull) atomicMin(a.first_oob, oob);
  if constexpr (OUT_VALID) {
    __shared__ unsigned long long s[4];
    if (lane == 0) s[wave] = nvalid;
    __syncthreads();
    if (t == 0) a.block_valid[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
  }
}

// check_bounds (take.rs:167-209): first position whose VALID index is >= len
// (or negative).  Signed indices compare in their own type, not reinterpreted.
template <typename IDX>
__global__ void __launch_bounds__(256) check_bounds_kernel(const IDX* idx, BitView ivalid, int64_t n,
                                                           int64_t len, int check_negative,
                                                           unsigned long long* first_bad) {
  unsigned long long bad = ~0ull;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    IDX r = idx[i];
    bool valid = bv_get(ivalid, i);
    bool isbad;
    // with index nulls the reference only tests `index >= len` (take.rs:183-191); negatives are
    // caught there only by the no-null fold (:193-195)
    if constexpr (std::is_signed<IDX>::value) isbad = (check_negative && r < 0) || (int64_t)r >= len;
    else isbad = (uint64_t)r >= (uint64_t)len;
    if (valid && isbad && (unsigned long long)i < bad) bad = (unsigned long long)i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(bad, o, 64);
    bad = other < bad ? other : bad;
  }
  if ((threadIdx.x & 63) == 0 && bad != ~0ull) atomicMin(first_bad, bad);
}

// After the gather: mail[0] = first out-of-bounds position (~0 = none), mail[1] = valid output rows; the
// position word goes back to ~0 for the next call (ctx->scratch is self-cleaning) and the mailbox is posted —
// the host's ONE wait of the call.  `in` == nullptr: no validity was produced.
__global__ void __launch_bounds__(1024) take_finish_kernel(const unsigned long long* in, int64_t n,
                                                           unsigned long long* first_oob, uint64_t* mail,
                                                           uint64_t seq) {
  unsigned long long acc = 0;
  if (in)
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += in[i];
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s[16];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    for (int i = 0; i < 16; i++) tot += s[i];
    const unsigned long long oob = *first_oob;
    *first_oob = ~0ull;
    __hip_atomic_store(mail, (uint64_t)oob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mail + 1, (uint64_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ah_mail_post(mail, seq);
  }
}

// Deferred mode (and hipGraph capture): nothing is read back.  If the gather met an out-of-bounds index, the FIRST such fault
// in stream order is parked in the context's fault slot {position, index value as the reference prints it, values length,
// kind} — ah_synchronize / ah_array_resolve raise it with the synchronous call's text — and the position word is re-armed
// for the next call (also on every replay of a captured graph).
__global__ void __launch_bounds__(64) take_finish_deferred_kernel(unsigned long long* first_oob, const void* indices, int iw, int is_signed,
                                                                  unsigned long long values_len, unsigned long long kind,
                                                                  unsigned long long* fault) {
  if (threadIdx.x != 0) return;
  const unsigned long long oob = *first_oob;
  if (oob == ~0ull) return;
  *first_oob = ~0ull;
  if (fault[0] != ~0ull) return;  // an earlier call's fault is still waiting for the host
  unsigned long long raw = 0;
  const unsigned char* p = (const unsigned char*)indices + oob * (unsigned long long)iw;
  for (int b = 0; b < iw; ++b) raw |= (unsigned long long)p[b] << (8 * b);
  unsigned long long uv = raw;  // ToIndices (take.rs:1030-1084): i8 / i16 widen with their sign, then everything is taken as unsigned
  if (is_signed && iw == 1) uv = (uint32_t)(int32_t)(int8_t)raw;
  else if (is_signed && iw == 2) uv = (uint32_t)(int32_t)(int16_t)raw;
  else if (iw == 4) uv = (uint32_t)raw;
  fault[1] = uv;
  fault[2] = values_len;
  fault[3] = kind;
  __threadfence();
  fault[0] = oob;
}

template <int W, typename IDX>
void launch_take_wi(ah_context* ctx, const TakeArgs& a, bool out_valid, int grid) {
  if (out_valid) take_kernel<W, IDX, true, 4><<<grid, 256, 0, ctx->stream>>>(a);
  else take_kernel<W, IDX, false, 4><<<grid, 256, 0, ctx->stream>>>(a);
}

template <int W>
ah_status launch_take_w(ah_context* ctx, ah_type it, const TakeArgs& a, bool ov, int grid) {
  switch (it) {
    case AH_INT8: launch_take_wi<W, int8_t>(ctx, a, ov, grid); break;
    case AH_UINT8: launch_take_wi<W, uint8_t>(ctx, a, ov, grid); break;
    case AH_INT16: launch_take_wi<W, int16_t>(ctx, a, ov, grid); break;
    case AH_UINT16: launch_take_wi<W, uint16_t>(ctx, a, ov, grid); break;
    case AH_INT32: launch_take_wi<W, int32_t>(ctx, a, ov, grid); break;
    case AH_UINT32: launch_take_wi<W, uint32_t>(ctx, a, ov, grid); break;
    case AH_INT64: launch_take_wi<W, int64_t>(ctx, a, ov, grid); break;
    case AH_UINT64: launch_take_wi<W, uint64_t>(ctx, a, ov, grid); break;
    default:
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "Take only supported for integers, got %s",
                     ah_type_name(it));
  }
  return AH_OK;
}

ah_status launch_take(ah_context* ctx, int width, ah_type it, const TakeArgs& a, bool ov, int grid) {
  switch (width) {
    case 0: return launch_take_w<0>(ctx, it, a, ov, grid);
    case 1: return launch_take_w<1>(ctx, it, a, ov, grid);
    case 2: return launch_take_w<2>(ctx, it, a, ov, grid);
    case 4: return launch_take_w<4>(ctx, it, a, ov, grid);
    case 8: return launch_take_w<8>(ctx, it, a, ov, grid);
    case 16: return launch_take_w<16>(ctx, it, a, ov, grid);
    case 32: return launch_take_w<32>(ctx, it, a, ov, grid);
  }
  return ah_fail(ctx, AH_INVALID_ARGUMENT, "unsupported value width %d", width);
}

template <typename IDX>
void launch_cb(ah_context* ctx, const ah_array_view* ind, BitView iv, int64_t len,
               unsigned long long* first, int grid, int check_negative) {
  check_bounds_kernel<IDX><<<grid, 256, 0, ctx->stream>>>((const IDX*)ind->values, iv, ind->length,
                                                          len, check_negative, first);
}

// value of an index slot, as the reference would print it
ah_status read_index(ah_context* ctx, const ah_array_view* ind, int64_t pos, int64_t* sval,
                     uint64_t* uval) {
  int w = ah_type_width(ind->type);
  uint64_t raw = 0;
  AH_HIP(ctx, hipMemcpyAsync(&raw, (const char*)ind->values + pos * w, w, hipMemcpyDeviceToHost,
                             ctx->stream));
  AH_HIP(ctx, ah_stream_wait(ctx));
  switch (ind->type) {
    case AH_INT8: *sval = (int8_t)raw; *uval = (uint32_t)(int32_t)(int8_t)raw; break;
    case AH_INT16: *sval = (int16_t)raw; *uval = (uint32_t)(int32_t)(int16_t)raw; break;
    case AH_INT32: *sval = (int32_t)raw; *uval = (uint32_t)raw; break;
    case AH_INT64: *sval = (int64_t)raw; *uval = raw; break;
    default: *sval = (int64_t)raw; *uval = raw; break;
  }
  return AH_OK;
}

}  // namespace

extern "C" ah_status ah_take(ah_context* ctx, const ah_array_view* values,
                             const ah_array_view* indices, int32_t check_bounds,
                             ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !values || !indices || !out) return AH_INVALID_ARGUMENT;
  ah_out_init(out);
  hipSetDevice(ctx->device);
  if (!ah_type_is_integer(indices->type))
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Take only supported for integers, got %s",
                   ah_type_name(indices->type));
  const bool is_string = values->type == AH_UTF8 || values->type == AH_LARGE_UTF8;
  const int width = ah_type_width(values->type);
  if (width < 0 && !is_string)
    return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "take not supported for type %s",
                   ah_type_name(values->type));
  const int64_t n = indices->length;
  out->type = values->type;

  // Deferred mode: a take of fixed-width values without check_bounds only ENQUEUES — the reference's
  // out-of-bounds panic is raised by the next ah_synchronize / ah_array_resolve (same text) instead of at return.  It needs
  // the index null count to be known (it picks the panic's wording) or no index validity at all.
  const bool defer = ctx->deferred && !check_bounds && !is_string && n > 0 && (!indices->validity || indices->null_count >= 0);
  int64_t idx_nulls = 0;
  if (defer) idx_nulls = indices->validity ? indices->null_count : 0;
  else AH_TRY(ah_resolve_null_count(ctx, indices, &idx_nulls));
  BitView ivalid = indices->validity ? make_bitview(indices->validity, indices->validity_bit_offset)
                                     : BitView{nullptr, 0};

  unsigned long long* flags = nullptr;  // per-block valid-row counts
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ah_ceil_div(n, 1024), 256 * 16));
  AH_TRY(ah_pool_alloc(ctx, (size_t)(grid + 4) * 8, (void**)&flags));
  unsigned long long* block_valid = flags;
  // first out-of-bounds / first bad position: a persistent word that is all-ones between calls (the kernels
  // that read it back restore it), so no per-call memset
  unsigned long long* first_pos = ctx->scratch + AH_SCRATCH_ONES;

  // TakeOptions::check_bounds (take.rs:97-99, 167-209)
  if (check_bounds && n > 0) {
    int iw = ah_type_width(indices->type);
    bool representable = true;  // T::Native::from_usize(len) (take.rs:174-181)
    if (iw < 8) {
      uint64_t maxv = ah_type_is_signed(indices->type) ? ((1ull << (iw * 8 - 1)) - 1)
                                                       : ((1ull << (iw * 8)) - 1);
      representable = (uint64_t)values->length <= maxv;
    }
    if (representable) {
      int g = (int)std::min<int64_t>(ah_ceil_div(n, 256), 4096);
      switch (indices->type) {
        case AH_INT8: launch_cb<int8_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        case AH_UINT8: launch_cb<uint8_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        case AH_INT16: launch_cb<int16_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        case AH_UINT16: launch_cb<uint16_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        case AH_INT32: launch_cb<int32_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        case AH_UINT32: launch_cb<uint32_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        case AH_INT64: launch_cb<int64_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
        default: launch_cb<uint64_t>(ctx, indices, ivalid, values->length, first_pos, g, idx_nulls == 0); break;
      }
      hipError_t e = ah_d2h_wait(ctx, ctx->pinned, first_pos, 8, true, ~0ull);
      if (e != hipSuccess) {
        ah_pool_free(ctx, flags);
        return ah_fail(ctx, AH_HIP_ERROR, "check_bounds failed: %s", hipGetErrorString(e));
      }
      if (ctx->pinned[0] != ~0ull) {
        int64_t pos = (int64_t)ctx->pinned[0], sv = 0;
        uint64_t uv = 0;
        ah_pool_free(ctx, flags);
        AH_TRY(read_index(ctx, indices, pos, &sv, &uv));
        if (ah_type_is_signed(indices->type))
          return ah_fail(ctx, AH_COMPUTE_ERROR,
                         "Array index out of bounds, cannot get item at index %lld from %lld entries",
                         (long long)sv, (long long)values->length);
        return ah_fail(ctx, AH_COMPUTE_ERROR,
                       "Array index out of bounds, cannot get item at index %llu from %lld entries",
                       (unsigned long long)uv, (long long)values->length);
      }
    }
  }

  if (is_string) {  // take_bytes (take.rs:499)
    ah_pool_free(ctx, flags);
    ah_status st = ah_take_bytes(ctx, values, indices, out);
    if (st != AH_OK) ah_out_init(out);
    return st;
  }
  if (n == 0) {  // take_impl :215-217
    ah_pool_free(ctx, flags);
    out->length = 0;
    return AH_OK;
  }

  int64_t val_nulls = 0;
  if (defer) {
    val_nulls = values->validity ? (values->null_count != 0 ? 1 : 0) : 0;  // unknown counts as "has nulls": the validity is gathered
  } else {
    ah_status st = ah_resolve_null_count(ctx, values, &val_nulls);
    if (st != AH_OK) {
      ah_pool_free(ctx, flags);
      return st;
    }
  }
  const bool val_has_nulls = values->validity && val_nulls > 0;  // take_nulls :422
  const bool out_valid = val_has_nulls || indices->validity != nullptr;

  size_t vbytes = width ? (size_t)n * width : ah_bitmap_bytes(n);
  size_t bbytes = out_valid ? ah_bitmap_bytes(n) : 0;
  void* ov = nullptr;
  void* ob = nullptr;
  ah_status st = ah_out_alloc(ctx, vbytes, &ov);
  if (st == AH_OK && out_valid) st = ah_out_alloc(ctx, bbytes, &ob);
  if (st != AH_OK) {
    ah_out_free(ctx, ov, vbytes);
    ah_pool_free(ctx, flags);
    return st;
  }
  TakeArgs a{};
  a.values = values->values;
  a.vbits = width == 0 ? make_bitview(values->values, values->values_bit_offset) : BitView{nullptr, 0};
  a.vvalid = val_has_nulls ? make_bitview(values->validity, values->validity_bit_offset)
                           : BitView{nullptr, 0};
  a.values_len = values->length;
  a.indices = indices->values;
  a.ivalid = ivalid;
  a.n = n;
  a.out_values = ov;
  a.out_valid = (unsigned long long*)ob;
  a.block_valid = block_valid;
  a.first_oob = first_pos;
  {
    ah_prof_scope ps(ctx, "take_gather");
    st = launch_take(ctx, width, indices->type, a, out_valid, grid);
  }
  hipError_t e = hipSuccess;
  if (st == AH_OK && defer) {
    const unsigned long long kind = width == 0 ? 0 : (idx_nulls > 0 ? 1 : 2);
    take_finish_deferred_kernel<<<1, 64, 0, ctx->stream>>>(first_pos, indices->values, ah_type_width(indices->type),
                                                           ah_type_is_signed(indices->type) ? 1 : 0, (unsigned long long)values->length, kind,
                                                           ctx->fault_dev);
    e = hipGetLastError();
    ah_pool_free(ctx, flags);  // (reuse is stream-ordered; while a graph is recorded the block stays parked with it)
    if (e != hipSuccess) {
      ah_out_free(ctx, ov, vbytes);
      ah_out_free(ctx, ob, bbytes);
      return ah_fail(ctx, AH_HIP_ERROR, "take failed: %s", hipGetErrorString(e));
    }
    ctx->fault_armed = true;
    ctx->inflight = true;
    out->length = n;
    out->values = ov;
    out->values_bytes = (int64_t)vbytes;
    if (out_valid) {  // kept even if it turns out all-valid; the count is unknown until ah_array_resolve
      out->validity = (uint8_t*)ob;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = -1;
    }
    return AH_OK;
  }
  if (st == AH_OK) {
    const uint64_t seq = ah_mail_next(ctx);
    take_finish_kernel<<<1, 1024, 0, ctx->stream>>>(out_valid ? block_valid : nullptr, grid, first_pos,
                                                    ctx->pinned_dev, seq);
    e = hipGetLastError();
    if (e == hipSuccess) e = ah_mail_wait(ctx, seq);
  }
  ah_pool_free(ctx, flags);
  if (st != AH_OK || e != hipSuccess) {
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    if (st != AH_OK) return st;
    return ah_fail(ctx, AH_HIP_ERROR, "take failed: %s", hipGetErrorString(e));
  }
  if (ctx->pinned[0] != ~0ull) {  // the reference panics (take.rs:447, :454)
    int64_t pos = (int64_t)ctx->pinned[0], sv = 0;
    uint64_t uv = 0;
    ah_out_free(ctx, ov, vbytes);
    ah_out_free(ctx, ob, bbytes);
    AH_TRY(read_index(ctx, indices, pos, &sv, &uv));
    if (width == 0)  // BooleanBuffer::value (arrow-buffer/src/buffer/boolean.rs:495)
      return ah_fail(ctx, AH_PANIC, "assertion failed: idx < self.bit_len");
    if (idx_nulls > 0)
      return ah_fail(ctx, AH_PANIC, "Out-of-bounds index %llu", (unsigned long long)uv);
    return ah_fail(ctx, AH_PANIC, "index out of bounds: the len is %lld but the index is %llu",
                   (long long)values->length, (unsigned long long)uv);
  }
  out->length = n;
  out->values = ov;
  out->values_bytes = (int64_t)vbytes;
  if (out_valid) {
    int64_t nulls = n - (int64_t)ctx->pinned[1];
    if (val_has_nulls && nulls == 0) {  // from_unsliced_buffer (null.rs:266-270) -> None
      ah_out_free(ctx, ob, bbytes);
    } else {  // values without nulls: clone of the index nulls (take.rs:428)
      out->validity = (uint8_t*)ob;
      out->validity_bytes = (int64_t)bbytes;
      out->null_count = nulls;
    }
  }
  return AH_OK;
}
