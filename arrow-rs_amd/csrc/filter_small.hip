// filter_small.hip — `filter` / `filter_record_batch` for query-engine-sized batches: ONE launch, ONE host wait.
//
// Reference shapes: arrow/benches/filter_kernels.rs:39-45 (512 .. 65 536 rows), DataFusion's 8 192-row batches.  The
// general path (filter.hip) costs a count pass + its host wait (K sizes the output) + a scatter + its wait: ~36 us
// per call however small the input, against ~30 us for the CPU kernels on 8 192 rows x 8 columns.
// Below AH_FILTER_SMALL_MAX rows the whole call is latency, so:
//
//   * outputs are allocated at their worst-case size (K <= predicate length; <= 8 MiB per column here) BEFORE anything
//     runs, which removes the count round trip;
//   * one kernel (blockIdx.y = column, blockIdx.x = tile of 4096 rows) does count, prefix and scatter: a tile's first
//     output position is the popcount of the predicate words before it, recomputed by the tile itself from L2 (at
//     most 128 KiB of mask per tile) — no inter-workgroup hand-off, no dispatch-order assumption, no pre-zeroed flags;
//   * output validity words are written exactly once, by the tile that owns the word's FIRST bit: the owner of a
//     tile's last (partial) word gathers the missing bits forward from the rows that follow the tile — no atomics, so
//     no bitmap memset launch either;
//   * completion: one 64-bit atomicAdd per tile on its column's counter word carries {valid rows, selected rows}; ONE
//     small kernel behind the launch(es) copies the counters into the host's pinned mailbox, zeroes them and posts the
//     sequence word the host spins on.  (Posting from inside the filter kernel — last tile to arrive — was 2 us faster
//     and was dropped: the host would return while the kernel that wrote the output is still retiring; a separate
//     kernel starts only after it has ended, end-of-kernel release included.)  So: two launches, one wait.
//
// Results are identical to the general path (same IterationStrategy special cases: K == 0 -> empty, K == len -> the
// zero-copy slice, filter.rs:545-546; null buffer dropped when the result has no nulls, :523-525).
#include "common.hpp"
#include "filter_internal.hpp"

namespace {

constexpr int SMALL_THREADS = 256;
constexpr int SMALL_MAX_COLS = 24;  // ticket words available in ctx->scratch

struct SmallCol {
  const void* values;
  BitView vvalid;
  void* out_values;
  unsigned long long* out_valid;
};
struct SmallArgs {
  BitView mask, mask_valid;
  int64_t len;      // predicate length
  int ntiles;
  int col0;         // index of this launch's first column among all columns of the call (ticket / mail slot)
  unsigned long long* tickets;  // one zero-between-calls counter word per column: {valid rows : 32 | selected rows : 32}
  SmallCol c[8];
};

__device__ __forceinline__ uint64_t sel_word(const SmallArgs& a, int64_t s) {  // mask AND mask validity, bits >= len zero
  uint64_t m = bv_fetch64(a.mask, s, a.len);
  if (a.mask_valid.words) m &= bv_fetch64(a.mask_valid, s, a.len);
  return m;
}

template <int W, int V, bool HAS_VALID>
__global__ void __launch_bounds__(SMALL_THREADS) filter_small_kernel(SmallArgs a) {
  constexpr int T = tile_rows(W);
  constexpr int CAP = stage_cap(W);
  constexpr int RPT = T / SMALL_THREADS;
  constexpr int L = RPT / V;
  constexpr int NW = T / 64;
  constexpr uint32_t VMASK = (V >= 32) ? 0xFFFFFFFFu : ((1u << V) - 1u);
  constexpr int EPV = W >= 16 ? 1 : 16 / W;
  using ET = typename Elem<W>::type;

  __shared__ uint64_t s_m[NW];
  __shared__ uint64_t s_v[HAS_VALID ? NW : 1];
  __shared__ uint32_t s_base[NW];
  __shared__ uint32_t s_total;
  __shared__ unsigned long long s_pre[4];
  __shared__ uint64_t s_tail;
  __shared__ uint32_t s_vc[4];
  __shared__ __attribute__((aligned(16))) ET s_vals[CAP + EPV];
  __shared__ uint8_t s_flag[HAS_VALID ? CAP : 1];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const SmallCol c = a.c[blockIdx.y];
  const int tile = blockIdx.x;
  const int64_t row0 = (int64_t)tile * T;

  // 1. this tile's values: 16-byte loads issued first (rows past the predicate are never selected)
  Vec<W, V> regs[L];
  {
    const ET* vp = (const ET*)c.values;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int64_t r = row0 + (int64_t)(l * SMALL_THREADS + t) * V;
      if (r + V <= a.len) {
        regs[l] = *(const Vec<W, V>*)(vp + r);
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (r + e < a.len) regs[l].u.e[e] = vp[r + e];
      }
    }
  }

  // 2. the tile's first output position: popcount of every predicate word before it (L2-resident: every tile of every
  //    column reads the same <= 128 KiB)
  unsigned long long pre = 0;
  {
    const int64_t nprev = (int64_t)tile * NW;  // words before the tile: all whole (their rows are < row0 <= len)
    if (a.mask.off == 0 && !a.mask_valid.words) {  // the usual predicate: word-aligned bits, no nulls — plain independent loads
      const uint64_t* mw = a.mask.words;
#pragma unroll 8
      for (int64_t w = t; w < nprev; w += SMALL_THREADS) pre += __popcll(mw[w]);
    } else {  // bit offsets / predicate nulls: four words' loads in flight at a time (bv_fetch64 would wait for each)
      const bool has_mv = a.mask_valid.words != nullptr;
      for (int64_t w0 = t; w0 < nprev; w0 += 4 * SMALL_THREADS) {
        BvRaw rm[4], rv[4] = {};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t w = w0 + j * SMALL_THREADS, wc = w < nprev ? w : 0;
          rm[j] = bv_issue(a.mask, wc << 6, a.len);
          if (has_mv) rv[j] = bv_issue(a.mask_valid, wc << 6, a.len);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t w = w0 + j * SMALL_THREADS, wc = w < nprev ? w : 0;
          uint64_t m = bv_finish(rm[j], wc << 6, a.len);
          if (has_mv) m &= bv_finish(rv[j], wc << 6, a.len);
          if (w < nprev) pre += __popcll(m);
        }
      }
    }
  }
  pre = wave_reduce_add64(pre);
  if (lane == 0) s_pre[wave] = pre;

  // 3. wave 0: the tile's word table (mask, validity, exclusive popcount prefix)
  if (wave == 0) {
    uint64_t m = 0, v = 0;
    const int64_t s = row0 + ((int64_t)lane << 6);
    if (lane < NW && s < a.len) {
      m = sel_word(a, s);
      if constexpr (HAS_VALID) v = bv_fetch64(c.vvalid, s, a.len);
    }
    const int cnt = __popcll(m);
    const int incl = wave_scan_incl(cnt);
    if (lane < NW) {
      s_m[lane] = m;
      if constexpr (HAS_VALID) s_v[lane] = v;
      s_base[lane] = (uint32_t)(incl - cnt);
    }
    if (lane == 63) s_total = (uint32_t)incl;
  }
  __syncthreads();
  const int total = (int)s_total;
  const int64_t P = (int64_t)(s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3]);  // rows selected before this tile

  // 4. validity of the tile's last output word: when that word starts inside this tile's range [P, P + total) but ends
  //    beyond it, its remaining bits belong to the selected rows that FOLLOW the tile — gather them forward
  int need = 0;
  if constexpr (HAS_VALID) {
    const int tailpos = (int)((P + total) & 63);
    const int64_t last_word_first = (P + total - 1) & ~63ll;
    if (total > 0 && tailpos != 0 && last_word_first >= P) need = 64 - tailpos;
    if (wave == 0) {
      uint64_t tail = 0;
      int got = 0;
      for (int64_t base = row0 + T; need > 0 && base < a.len && got < need; base += 4096) {  // wave-uniform loop
        const int64_t s = base + ((int64_t)lane << 6);
        uint64_t m = 0, vv = 0;
        if (s < a.len) {
          m = sel_word(a, s);
          vv = bv_fetch64(c.vvalid, s, a.len);
        }
        const int cnt = __popcll(m);
        const int incl = wave_scan_incl(cnt);
        int r = got + incl - cnt;  // rank, among the rows after the tile, of this lane's first selected row
        uint64_t local = 0;
        while (m != 0 && r < need) {
          const int b = __ffsll((long long)m) - 1;
          local |= ((vv >> b) & 1ull) << (tailpos + r);
          m &= m - 1;
          ++r;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local |= __shfl_xor(local, o, 64);
        tail |= local;
        got += __shfl(incl, 63, 64);
      }
      if (lane == 0) s_tail = tail;
    }
  }

  int vc = 0;
  if (total > 0) {
    // chunks of the tile's selected rows: the first one ends on a 64-row boundary of the OUTPUT, so every later chunk
    // starts word-aligned and only the tile's first and last words are partial
    int p0 = 0;
    while (p0 < total) {
      const int64_t g = P + p0;
      const int lead = (int)(g & 63);
      int cnt = total - p0;
      if (cnt > CAP - lead) cnt = CAP - lead;
      const int phase = (int)(g & (EPV - 1));
      __syncthreads();  // s_tail / the previous chunk's LDS reads
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int r0 = (l * SMALL_THREADS + t) * V;
        const int w = r0 >> 6, sh = r0 & 63;
        const uint64_t word = s_m[w];
        const uint32_t bits = (uint32_t)(word >> sh) & VMASK;
        if (bits) {
          const uint32_t base = s_base[w] + (uint32_t)__popcll(word & ((1ull << sh) - 1ull)) - (uint32_t)p0;
          uint32_t vb = 0;
          if constexpr (HAS_VALID) vb = (uint32_t)(s_v[w] >> sh);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            if ((bits >> e) & 1u) {
              const uint32_t pos = base + (uint32_t)__popc(bits & ((1u << e) - 1u));
              if (pos < (uint32_t)cnt) {  // unsigned: also rejects positions before p0
                s_vals[pos + phase] = regs[l].u.e[e];
                if constexpr (HAS_VALID) s_flag[pos] = (uint8_t)((vb >> e) & 1u);
              }
            }
          }
        }
      }
      __syncthreads();
      // coalesced write-out of the chunk's values (16-byte stores where the output alignment allows)
      ET* op = (ET*)c.out_values + g;
      if constexpr (EPV > 1) {
        const int first = phase, last = phase + cnt;
        const int vfirst = (first + EPV - 1) / EPV, vlast = last / EPV;
        ET* gbase = op - phase;
        if (vfirst < vlast) {
          for (int j = vfirst + t; j < vlast; j += SMALL_THREADS)
            *(Vec<W, EPV>*)(gbase + j * EPV) = *(const Vec<W, EPV>*)(s_vals + j * EPV);
          if (t < vfirst * EPV - first) gbase[first + t] = s_vals[first + t];
          if (t < last - vlast * EPV) gbase[vlast * EPV + t] = s_vals[vlast * EPV + t];
        } else {
          for (int j = first + t; j < last; j += SMALL_THREADS) gbase[j] = s_vals[j];
        }
      } else {
        for (int j = t; j < cnt; j += SMALL_THREADS) op[j] = s_vals[j];
      }
      if constexpr (HAS_VALID) {
        const int64_t g0 = g & ~63ll;
        const int span64 = (lead + cnt + 63) & ~63;
        const bool last_chunk = p0 + cnt >= total;
        for (int q = t; q < span64; q += SMALL_THREADS) {
          const int j = q - lead;
          const int f = (j >= 0 && j < cnt) ? (int)s_flag[j] : 0;
          uint64_t word = __ballot(f);
          if (lane == 0) {
            vc += __popcll(word);
            const int64_t wfirst = g0 + (q & ~63);  // output position of this word's bit 0
            const bool owned = wfirst >= P;         // else: the word's first bit belongs to an earlier tile, which writes it
            if (last_chunk && (q & ~63) + 64 >= lead + cnt) word |= s_tail;  // the tile's last word: bits of the rows that follow
            if (owned) c.out_valid[wfirst >> 6] = word;
          }
        }
      }
      p0 += cnt;
    }
  }

  // 5. this tile's {valid rows, selected rows} into the column's counter word
  if constexpr (HAS_VALID) {
    if (lane == 0) s_vc[wave] = (uint32_t)vc;
  }
  __syncthreads();
  if (t == 0 && total > 0) {
    const unsigned long long valid = HAS_VALID ? (unsigned long long)(s_vc[0] + s_vc[1] + s_vc[2] + s_vc[3]) : (unsigned long long)total;
    atomicAdd(&a.tickets[a.col0 + (int)blockIdx.y], (valid << 32) | (unsigned long long)total);
  }
}

template <int W, bool HV>
void launch_small(ah_context* ctx, const SmallArgs& a, int ncols, bool aligned16) {
  const dim3 grid((unsigned)a.ntiles, (unsigned)ncols), block(SMALL_THREADS);
  constexpr int VV = W >= 16 ? 1 : 16 / W;
  if (aligned16 || VV == 1) filter_small_kernel<W, VV, HV><<<grid, block, 0, ctx->stream>>>(a);
  else filter_small_kernel<W, 1, HV><<<grid, block, 0, ctx->stream>>>(a);
}
template <bool HV>
void launch_small_w(ah_context* ctx, int width, const SmallArgs& a, int ncols, bool aligned16) {
  switch (width) {
    case 1: launch_small<1, HV>(ctx, a, ncols, aligned16); break;
    case 2: launch_small<2, HV>(ctx, a, ncols, aligned16); break;
    case 4: launch_small<4, HV>(ctx, a, ncols, aligned16); break;
    case 8: launch_small<8, HV>(ctx, a, ncols, aligned16); break;
    case 16: launch_small<16, HV>(ctx, a, ncols, aligned16); break;
    default: launch_small<32, HV>(ctx, a, ncols, aligned16); break;
  }
}

}  // namespace

ah_status ah_filter_small(ah_context* ctx, int ncols, const ah_array_view* columns, const ah_array_view* predicate,
                          ah_array_out* outs, int64_t* out_rows) {
  if (ncols < 1 || ncols > SMALL_MAX_COLS || predicate->type != AH_BOOL) return AH_NOT_YET_IMPLEMENTED;
  const int64_t len = predicate->length;
  if (len <= 0 || len > AH_FILTER_SMALL_MAX || ctx->deferred) return AH_NOT_YET_IMPLEMENTED;
  std::vector<int> width((size_t)ncols);
  std::vector<char> hv((size_t)ncols);
  for (int c = 0; c < ncols; ++c) {
    const ah_array_view& v = columns[c];
    const int w = ah_type_width(v.type);
    if (w <= 0 || v.type == AH_BOOL || len > v.length || !v.values) return AH_NOT_YET_IMPLEMENTED;  // (errors: the general path's texts)
    width[c] = w;
    hv[c] = v.validity != nullptr && v.null_count != 0;  // unknown (-1) counts as nullable: the kernel counts
  }
  hipSetDevice(ctx->device);
  // worst-case outputs
  std::vector<void*> ov((size_t)ncols, nullptr), ob((size_t)ncols, nullptr);
  std::vector<size_t> vbytes((size_t)ncols), bbytes((size_t)ncols, 0);
  ah_status st = AH_OK;
  for (int c = 0; c < ncols && st == AH_OK; ++c) {
    vbytes[c] = (size_t)len * width[c];
    st = ah_out_alloc(ctx, vbytes[c], &ov[c]);
    if (st == AH_OK && hv[c]) {
      bbytes[c] = ah_bitmap_bytes(len);
      st = ah_out_alloc(ctx, bbytes[c], &ob[c]);
    }
  }
  auto free_all = [&]() {
    for (int c = 0; c < ncols; ++c) {
      ah_out_free(ctx, ov[c], vbytes[c]);
      ah_out_free(ctx, ob[c], bbytes[c]);
    }
  };
  if (st != AH_OK) {
    free_all();
    return st;
  }
  SmallArgs base{};
  base.mask = make_bitview(predicate->values, predicate->values_bit_offset);
  base.mask_valid = (predicate->validity && predicate->null_count != 0) ? make_bitview(predicate->validity, predicate->validity_bit_offset)
                                                                         : BitView{nullptr, 0};
  base.len = len;
  base.tickets = ctx->scratch + AH_SCRATCH_TICKETS;
  // one launch per (width, nullable) shape, up to 8 columns each; tickets and mail slots are indexed by the column's
  // position in launch order, so remember where each input column went
  std::vector<int> slot((size_t)ncols, -1);
  std::vector<char> done((size_t)ncols, 0);
  int next_slot = 0;
  {
    ah_prof_scope ps(ctx, "filter_small");
    for (int c = 0; c < ncols; ++c) {
      if (done[c]) continue;
      SmallArgs a = base;
      a.col0 = next_slot;
      a.ntiles = (int)ah_ceil_div(len, tile_rows(width[c]));
      int g = 0;
      bool aligned16 = true;
      for (int d = c; d < ncols && g < 8; ++d) {
        if (done[d] || width[d] != width[c] || hv[d] != hv[c]) continue;
        const ah_array_view& v = columns[d];
        a.c[g].values = v.values;
        a.c[g].vvalid = hv[d] ? make_bitview(v.validity, v.validity_bit_offset) : BitView{nullptr, 0};
        a.c[g].out_values = ov[d];
        a.c[g].out_valid = (unsigned long long*)ob[d];
        aligned16 = aligned16 && (((uintptr_t)v.values) & 15) == 0;
        slot[d] = next_slot++;
        done[d] = 1;
        ++g;
      }
      if (hv[c]) launch_small_w<true>(ctx, width[c], a, g, aligned16);
      else launch_small_w<false>(ctx, width[c], a, g, aligned16);
    }
  }
  hipError_t e = hipGetLastError();
  // the call's one wait: the counter words -> pinned slots [0, ncols), counters back to zero, mailbox posted
  if (e == hipSuccess) e = ah_d2h_wait(ctx, ctx->pinned, ctx->scratch + AH_SCRATCH_TICKETS, (size_t)ncols * 8, /*reset=*/true);
  if (e != hipSuccess) {
    // a launch failed part-way: the counters may be half counted
    hipStreamSynchronize(ctx->stream);
    hipMemsetAsync(ctx->scratch + AH_SCRATCH_TICKETS, 0, (SMALL_MAX_COLS + 1) * 8, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    free_all();
    return ah_fail(ctx, AH_HIP_ERROR, "filter failed: %s", hipGetErrorString(e));
  }
  const int64_t K = (int64_t)(ctx->pinned[slot[0]] & 0xFFFFFFFFull);  // the same for every column
  if (out_rows) *out_rows = K;
  for (int c = 0; c < ncols; ++c) {
    ah_array_out* out = &outs[c];
    ah_out_init(out);
    out->type = columns[c].type;
    const ah_array_view& v = columns[c];
    const int64_t valid = (int64_t)(ctx->pinned[slot[c]] >> 32);
    // IterationStrategy::default_strategy (filter.rs:346-364) special cases, as the general path
    if (K == 0) {  // None -> new_empty_array(data_type) :545
      ah_out_free(ctx, ov[c], vbytes[c]);
      ah_out_free(ctx, ob[c], bbytes[c]);
      continue;
    }
    if (K == len) {  // All -> values.slice(0, count) :546 (zero-copy)
      ah_out_free(ctx, ov[c], vbytes[c]);
      ah_out_free(ctx, ob[c], bbytes[c]);
      out->length = K;
      out->values = const_cast<void*>(v.values);
      out->values_bytes = K * width[c];
      out->flags = AH_OUT_BORROWED;
      if (v.validity) {
        out->validity = const_cast<uint8_t*>(v.validity);
        out->validity_bit_offset = v.validity_bit_offset;
        out->null_count = hv[c] ? K - valid : 0;
      }
      continue;
    }
    out->length = K;
    out->values = ov[c];
    out->values_bytes = (int64_t)vbytes[c];  // the allocation (worst case); K * width of it are the values
    if (hv[c]) {
      const int64_t nulls = K - valid;
      if (nulls == 0) {  // filter_nulls :523-525 -> None
        ah_out_free(ctx, ob[c], bbytes[c]);
      } else {
        out->validity = (uint8_t*)ob[c];
        out->validity_bytes = (int64_t)bbytes[c];
        out->null_count = nulls;
      }
    }
  }
  return AH_OK;
}
