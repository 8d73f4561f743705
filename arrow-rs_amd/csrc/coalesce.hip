// coalesce.hip — arrow_select::coalesce::BatchCoalescer as a native object behind the C ABI.
//
// Reference: arrow-select/src/coalesce.rs:148-700 (state machine: exact-size output batches in input order, the
// large-batch bypass cases 1-3 of push_batch :296-420, push_batch_with_filter :229, finish_buffered_batch :536) and
// coalesce/primitive.rs:28-207 (InProgressPrimitiveArray: copy_rows, copy_rows_by_filter_from, NullBufferBuilder).
//
// The state machine is host logic and lives here in C++ (round 1 had it in the Python mirror only: ~40 us of
// interpreter time per pushed batch, 2.5 ms per 1e9 rows at 2^24-row batches).  Data movement is the device
// primitives of filter.hip / concat.hip: the filter scatters STRAIGHT into the in-progress output batch at row
// `buffered_rows` (no intermediate filtered array), and no per-push call waits for the GPU — appended-null counts
// accumulate in one device word per column and are read in ONE wait when an output batch is finished; the only
// wait of a filtered push is the predicate's count, which the "fits / does not fit" decision needs.
// Fixed-width columns are InProgressPrimitiveArray (coalesce/primitive.rs): values scattered / copied straight into the
// in-progress buffers.  Boolean, Utf8 and LargeUtf8 columns are GenericInProgressArray (coalesce/generic.rs:32-108): the
// in-progress batch keeps a list of pieces — filtered results are ADOPTED (and shared between two output batches when
// they straddle a boundary, like the reference's Arc'ed slices), unfiltered rows are copied out of the caller's
// borrowed buffers — and `concat`s them when the batch is finished.
// Utf8View / BinaryView columns (round 4; InProgressByteViewArray, coalesce/byte_view.rs:39-330): the 16-byte views are a
// fixed-width column like any other; what the reference's builder does on top — append the source's data buffers to the
// in-progress array and shift every appended view's buffer index by the number of buffers already there
// (byte_view.rs `append_views_and_update_buffer_index`) — is split between the two sides of the C ABI: the variadic data
// buffers never enter it (filter / take of views work the same way), so the HOST declares how many buffers each pushed
// batch carries (ah_coalescer_declare_view_buffers), the library shifts the buffer indices of the rows it appends
// (rebase_views_kernel) and reports, per completed batch, WHICH inputs contributed rows and in which order
// (ah_coalescer_completed_batch_sources: the push sequence numbers) — the host concatenates those inputs' buffer lists.
// The reference additionally copies strings out of sparsely used buffers (its `ideal_buffer_size` GC heuristic); that
// changes memory footprint, not the logical value, and is not done here.
#include "common.hpp"
#include "filter_internal.hpp"

#include <algorithm>
#include <ctime>
#include <deque>
#include <memory>
#include <vector>

extern "C" ah_status ah_filter_predicate_apply_into_acc(ah_context*, const ah_filter_predicate*, const ah_array_view*, void*,
                                                        uint8_t*, int64_t, uint64_t*);
extern "C" ah_status ah_filter_predicates_build(ah_context*, int32_t, const ah_array_view*, ah_filter_predicate**);
// filter.hip: the columns of one batch through one scatter launch (AH_NOT_YET_IMPLEMENTED: shapes differ, nothing done)
ah_status ah_filter_apply_into_acc_cols(ah_context*, const ah_filter_predicate*, int, const ah_array_view*, void* const*,
                                        uint8_t* const*, int64_t, unsigned long long*, int64_t, int64_t, int, double);
// filter.hip: up to 8 (batch, window) segments through one scatter launch
ah_status ah_filter_apply_multi(ah_context*, int, const ah_filter_predicate* const*, const ah_array_view* const*, const int64_t*,
                                const int64_t*, const int64_t*, int, void* const*, uint8_t* const*, unsigned long long*);
// filter.hip: ncols x 64 NULL-row counters -> pinned words (summed per column), counters back to zero; one launch, no
// wait: *seq is posted to the context mailbox behind it
ah_status ah_coalesce_post_nulls(ah_context*, unsigned long long*, int, uint64_t*, uint64_t*);
ah_status ah_coalesce_wait(ah_context*, uint64_t);
// filter.hip: ah_filter_predicate_build in two halves (enqueue the count pass / wait for K)
ah_status ah_filter_predicate_begin(ah_context*, const ah_array_view*, ah_filter_predicate**, uint64_t*, bool*);
ah_status ah_filter_predicate_end(ah_context*, ah_filter_predicate*, uint64_t, bool);
extern "C" void ah_filter_predicate_free(ah_context*, ah_filter_predicate*);
// filter.hip: the counts of up to 64 predicates into the CALLER's pinned words, posted / waited separately
ah_status ah_filter_predicates_begin(ah_context*, int32_t, const ah_array_view*, ah_filter_predicate**, uint64_t*, uint64_t*, uint64_t*);
ah_status ah_filter_predicates_end(ah_context*, int32_t, ah_filter_predicate**, const uint64_t*, uint64_t, const uint64_t*);

extern "C" ah_status ah_take(ah_context*, const ah_array_view*, const ah_array_view*, int32_t, ah_array_out*);

namespace {

// AH_COALESCE_TIMING=1: host time of the slab push's phases, printed when a coalescer is destroyed (profiles/r05_coalesce_sweep.md)
struct SlabTiming {
  bool on = false;
  double build = 0, upload_count = 0, wait = 0, append = 0;
  int64_t pushes = 0, batches = 0;
};
SlabTiming& slab_timing() {
  static SlabTiming t = [] {
    SlabTiming x;
    const char* e = getenv("AH_COALESCE_TIMING");
    x.on = e && e[0] == '1';
    return x;
  }();
  return t;
}
double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// an owned array shared by the pieces cut out of it
struct GenOwner {
  ah_context* ctx;
  ah_array_out out;
  GenOwner(ah_context* c, const ah_array_out& o) : ctx(c), out(o) {}
  GenOwner(const GenOwner&) = delete;
  ~GenOwner() { ah_array_release(ctx, &out); }
};
struct GenPiece {
  std::shared_ptr<GenOwner> owner;
  int64_t offset, len;
};

struct CoColumn {
  ah_type type = AH_INT64;
  int width = 8;
  bool generic = false;  // Boolean / Utf8 / LargeUtf8: pieces + concat
  bool is_view = false;  // Utf8View / BinaryView: 16-byte views as a fixed-width column + buffer-index bookkeeping
  uint32_t view_base = 0;  // data buffers the in-progress batch already references (from earlier contributing inputs)
  uint32_t cur_base = 0;   // ... as it was when the CURRENT input started contributing: what its views are shifted by
  int32_t cur_nbuf = 0;    // data buffers of the current input
  void* values = nullptr;
  uint8_t* validity = nullptr;
  size_t vbytes = 0, bbytes = 0;
  std::vector<GenPiece> pieces;
};

// rows [offset, offset + len) of an array as a view (Array::slice for the three generic layouts and the primitives)
ah_array_view slice_view(ah_type t, const void* values, int64_t values_bit_offset, const uint8_t* validity,
                         int64_t validity_bit_offset, const void* offsets, int64_t whole_len, int64_t whole_nulls, int64_t offset,
                         int64_t len) {
  ah_array_view v{};
  v.type = t;
  v.length = len;
  const int w = ah_type_width(t);
  if (t == AH_UTF8 || t == AH_LARGE_UTF8) {
    v.values = values;
    v.offsets = offsets ? (const char*)offsets + (size_t)offset * (t == AH_UTF8 ? 4 : 8) : nullptr;
  } else if (t == AH_BOOL) {
    v.values = values;
    v.values_bit_offset = values_bit_offset + offset;
  } else {
    v.values = values ? (const char*)values + (size_t)offset * w : nullptr;
  }
  if (validity) {
    v.validity = validity;
    v.validity_bit_offset = validity_bit_offset + offset;
    v.null_count = (offset == 0 && len == whole_len) ? whole_nulls : -1;
  }
  return v;
}
ah_array_view piece_view(const GenPiece& p) {
  const ah_array_out& o = p.owner->out;
  return slice_view(o.type, o.values, o.values_bit_offset, o.validity, o.validity_bit_offset, o.offsets, o.length,
                    o.validity ? o.null_count : 0, p.offset, p.len);
}

struct CoBatch {
  std::vector<ah_array_out> cols;
  std::vector<uint64_t> sources;  // push sequence numbers of the inputs that contributed rows, in order (view schemas)
  int64_t rows = 0;
  uint64_t tag = 0;  // != 0: a bypassed input batch (borrowed buffers), the caller's tag for it
  // a finished batch whose null counts are still on their way: they land in pinned words [ring, ring + ncols) once the
  // mailbox has passed `seq`; resolved (validity kept or dropped) when the batch is fetched
  bool pending = false;
  uint64_t seq = 0;
  int ring = 0;
  // a batch carved out of a slab push: its null counts are words [slab_index * ncols, ...) of the push's pinned block
  std::shared_ptr<struct SlabNulls> slab_nulls;
  int64_t slab_index = 0;
  // run_n > 0: this entry stands for run_n consecutive target-row batches of one slab (batch j = rows [j * target, ...) of the
  // base pointers); they are materialised one by one when fetched — a push at the reference's batch sizes completes thousands
  int64_t run_n = 0, run_next = 0;
  ah_slab* run_slab = nullptr;  // one reference held until the run is used up
  std::vector<void*> run_values;
  std::vector<uint8_t*> run_valid;
};

// null counts of the full output batches of one slab push, written by ONE kernel into pinned words the batches share
struct SlabNulls {
  ah_context* ctx = nullptr;
  uint64_t* pin = nullptr;
  size_t bytes = 0;
  uint64_t seq = 0;
  bool arrived = false;
  ~SlabNulls() {
    if (pin) ah_pinned_free(ctx, pin, bytes);
  }
};

// pinned block of a slab push in flight: [wave prefixes (nwaves + 1 words)][table staging]; grown on demand, two per coalescer
struct SlabPin {
  void* host = nullptr;
  void* dev = nullptr;
  size_t bytes = 0;
  bool busy = false;
};

// device copy of one push's tables.  Uploaded on the CONTEXT's copy stream, so the DMA (2.5 MB per 2^27 rows of 8192-row
// batches: ~50 us) runs beside the previous push's scatter instead of in front of this push's count.  Three, rotating: a buffer
// is rewritten only behind `idle`, recorded on the context's stream after the last kernel of the push that read it.
struct SlabTables {
  void* dev = nullptr;
  size_t bytes = 0;
  hipEvent_t idle = nullptr, uploaded = nullptr;
  bool owner_live = false;  // a push between _begin and _end / _abort reads it
  bool used = false;        // `idle` has been recorded at least once
};

}  // namespace

struct ah_coalescer {
  int ncols = 0;
  int64_t target = 0;
  int64_t limit = -1;  // biggest_coalesce_batch_size (coalesce.rs:196-216); < 0 = None
  std::vector<CoColumn> cols;
  int64_t buffered = 0;
  std::deque<CoBatch> completed;
  int64_t completed_batches = 0;  // batches the queue stands for (a slab run counts each of its batches)
  uint64_t* acc = nullptr;  // device: 64 appended-null counters per column of the in-progress batch (scatter tiles spread
                            // their atomics over them; copies add to the first); summed once per finished batch
  uint64_t* pin = nullptr;      // own pinned words for the finished batches' null counts (host view / device view)
  uint64_t* pin_dev = nullptr;
  size_t pin_bytes = 0;
  int ring_next = 0, ring_slots = 0;  // ring of ncols-word groups in `pin`
  // two groups of 64 pinned count words for pushes whose counts are in flight (push_batches_with_filters_begin / _end)
  uint64_t* cnt_pin = nullptr;
  uint64_t* cnt_pin_dev = nullptr;
  uint64_t* quant_pin = nullptr;
  uint64_t* quant_pin_dev = nullptr;
  bool cnt_busy[2] = {false, false};
  SlabPin slab_pin[2];
  SlabTables slab_tbl[3];
  int slab_tbl_next = 0;
  // view schemas: every push is an "input" with a sequence number (0, 1, 2, ... in push order; the host counts the same
  // way); `declared` holds the per-column data-buffer counts of the inputs about to be pushed
  bool has_views = false;
  uint64_t input_seq = 0, cur_seq = 0;
  std::deque<std::vector<int32_t>> declared;
  std::vector<uint64_t> sources;  // inputs that have contributed rows to the in-progress batch
  double selectivity = 0.1;  // of the last filtered push: picks the speculative scatter's load-predication mode
  bool failed = false;       // a device error hit after rows had been enqueued into the in-progress batch
};

namespace {

// views [0, n): buffer_index += base for the views that point into a data buffer (length > 12; byte_view.rs
// `append_views_and_update_buffer_index`)
__global__ void __launch_bounds__(256) rebase_views_kernel(uint4* views, int64_t n, uint32_t base) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint4 v = views[i];
    if (v.x > 12u) {
      v.z += base;
      views[i] = v;
    }
  }
}

// a push starts: its sequence number, and (view schemas) the data-buffer counts the host declared for it
ah_status begin_input(ah_context* ctx, ah_coalescer* co) {
  if (co->has_views && co->declared.empty())  // (checked before the sequence number moves: host and library count together)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "BatchCoalescer with view columns: call ah_coalescer_declare_view_buffers before each push");
  co->cur_seq = co->input_seq++;
  if (!co->has_views) return AH_OK;
  const std::vector<int32_t> counts = co->declared.front();
  co->declared.pop_front();
  for (int i = 0; i < co->ncols; ++i) co->cols[i].cur_nbuf = co->cols[i].is_view ? counts[(size_t)i] : 0;
  return AH_OK;
}

// every push consumes exactly `n` sequence numbers and (view schemas) `n` declared buffer-count entries, whether or not it
// succeeds — the host counts its pushes the same way, so a rejected push cannot leave the two sides out of step
struct InputGuard {
  ah_coalescer* co;
  uint64_t seq0;
  size_t decl0;
  int n;
  InputGuard(ah_coalescer* c, int n_) : co(c), seq0(c->input_seq), decl0(c->declared.size()), n(n_) {}
  ~InputGuard() {
    co->input_seq = seq0 + (uint64_t)n;
    const size_t want = decl0 > (size_t)n ? decl0 - (size_t)n : 0;
    while (co->declared.size() > want) co->declared.pop_front();
  }
};

// the current input is about to append rows to the in-progress batch: the first time it does, it becomes a source of
// that batch and its views are shifted by the buffers the batch already references
void note_contribution(ah_coalescer* co) {
  if (!co->has_views) return;
  if (!co->sources.empty() && co->sources.back() == co->cur_seq) return;
  co->sources.push_back(co->cur_seq);
  for (auto& c : co->cols)
    if (c.is_view) {
      c.cur_base = c.view_base;
      c.view_base += (uint32_t)c.cur_nbuf;
    }
}

// up to 8 fresh bitmaps zeroed by one launch (blockIdx.y = bitmap): four 1 MiB hipMemsetAsync calls per output batch were
// 30 launches of ~4.6 us per 10^9 rows
struct ZeroArgs {
  unsigned long long* p[8];
  size_t words[8];
};
__global__ void __launch_bounds__(256) zero_bitmaps_kernel(ZeroArgs z) {
  unsigned long long* p = z.p[blockIdx.y];
  const size_t n = z.words[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0ull;
}
ah_status zero_bitmaps(ah_context* ctx, int n, uint8_t* const* bitmaps, const size_t* bytes) {
  for (int base = 0; base < n; base += 8) {
    const int m = std::min(8, n - base);
    ZeroArgs z{};
    size_t most = 0;
    for (int i = 0; i < m; ++i) {
      z.p[i] = (unsigned long long*)bitmaps[base + i];
      z.words[i] = bytes[base + i] / 8;  // (ah_bitmap_bytes: whole words)
      most = std::max(most, z.words[i]);
    }
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(256, (most + 1023) / 1024));
    zero_bitmaps_kernel<<<dim3(gx, (unsigned)m), 256, 0, ctx->stream>>>(z);
    if (hipGetLastError() != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "coalescer bitmap reset failed");
  }
  return AH_OK;
}

// value + validity buffers of one output batch for every fixed-width column i with values[i] == nullptr (generic
// columns keep piece lists); all or nothing
ah_status alloc_window(ah_context* ctx, ah_coalescer* co, void** values, uint8_t** validity) {
  std::vector<int> fresh;
  ah_status st = AH_OK;
  for (int i = 0; i < co->ncols && st == AH_OK; ++i) {
    CoColumn& c = co->cols[i];
    if (values[i] || c.generic) continue;
    c.vbytes = std::max<size_t>((size_t)co->target * c.width, 8);
    c.bbytes = ah_bitmap_bytes(co->target);
    void* b = nullptr;
    st = ah_out_alloc(ctx, c.vbytes, &values[i]);
    if (st == AH_OK) st = ah_out_alloc(ctx, c.bbytes, &b);
    if (st != AH_OK) {  // never leave a column with values but no validity: the next call would skip it
      ah_out_free(ctx, values[i], c.vbytes);
      values[i] = nullptr;
      break;
    }
    validity[i] = (uint8_t*)b;
    fresh.push_back(i);
  }
  if (st == AH_OK && !fresh.empty()) {
    std::vector<uint8_t*> bm;
    std::vector<size_t> by;
    for (int i : fresh) bm.push_back(validity[i]), by.push_back(co->cols[i].bbytes);
    st = zero_bitmaps(ctx, (int)bm.size(), bm.data(), by.data());
  }
  if (st != AH_OK)
    for (int i : fresh) {
      ah_out_free(ctx, values[i], co->cols[i].vbytes);
      ah_out_free(ctx, validity[i], co->cols[i].bbytes);
      values[i] = nullptr, validity[i] = nullptr;
    }
  return st;
}

ah_status ensure_capacity(ah_context* ctx, ah_coalescer* co) {  // allocate on first write (primitive.rs:57-61)
  bool need = false;
  for (auto& c : co->cols) need = need || (!c.values && !c.generic);
  if (!need) return AH_OK;
  std::vector<void*> values((size_t)co->ncols);
  std::vector<uint8_t*> validity((size_t)co->ncols);
  for (int i = 0; i < co->ncols; ++i) values[(size_t)i] = co->cols[i].values, validity[(size_t)i] = co->cols[i].validity;
  AH_TRY(alloc_window(ctx, co, values.data(), validity.data()));
  for (int i = 0; i < co->ncols; ++i) co->cols[i].values = values[(size_t)i], co->cols[i].validity = validity[(size_t)i];
  return AH_OK;
}

// fills in what the null counts decide: NullBufferBuilder::finish keeps a buffer only if a null was ever appended
void resolve_batch(ah_context* ctx, ah_coalescer* co, CoBatch& b, const uint64_t* nulls) {
  for (int i = 0; i < co->ncols; ++i) {
    if (co->cols[i].generic) continue;
    ah_array_out& o = b.cols[i];
    if (nulls[i] > 0) {
      o.null_count = (int64_t)nulls[i];
    } else {
      ah_out_free(ctx, o.validity, (size_t)o.validity_bytes);
      o.validity = nullptr;
      o.validity_bytes = 0;
    }
  }
  b.pending = false;
}
ah_status resolve_pending(ah_context* ctx, ah_coalescer* co, CoBatch& b) {
  if (!b.pending) return AH_OK;
  if (b.slab_nulls) {
    SlabNulls& sn = *b.slab_nulls;
    if (!sn.arrived) {
      AH_TRY(ah_coalesce_wait(ctx, sn.seq));
      sn.arrived = true;
    }
    uint64_t nulls[AH_TBL_MAX_COLS];
    for (int i = 0; i < co->ncols; ++i) nulls[i] = __atomic_load_n(&sn.pin[(size_t)b.slab_index * co->ncols + i], __ATOMIC_RELAXED);
    resolve_batch(ctx, co, b, nulls);
    b.slab_nulls.reset();
    return AH_OK;
  }
  AH_TRY(ah_coalesce_wait(ctx, b.seq));
  resolve_batch(ctx, co, b, co->pin + (size_t)b.ring * co->ncols);
  return AH_OK;
}

ah_status finish_buffered(ah_context* ctx, ah_coalescer* co) {  // coalesce.rs:536
  if (co->buffered == 0) return AH_OK;
  // NO wait: one kernel folds the null counters into this batch's pinned ring slot and posts the mailbox; the host
  // looks at them when the batch is fetched.  (A wait here kept the host from enqueueing the next pushes: ~15 us of
  // idle GPU per output batch.)  The ring slot must be free: the batch that used it last has been resolved.
  const int ring = co->ring_next;
  for (auto& old : co->completed)
    if (old.pending && !old.slab_nulls && old.ring == ring) AH_TRY(resolve_pending(ctx, co, old));
  co->ring_next = (co->ring_next + 1) % co->ring_slots;
  uint64_t seq = 0;
  AH_TRY(ah_coalesce_post_nulls(ctx, (unsigned long long*)co->acc, co->ncols, co->pin_dev + (size_t)ring * co->ncols, &seq));
  CoBatch b;
  b.rows = co->buffered;
  b.pending = true;
  b.seq = seq;
  b.ring = ring;
  b.sources.swap(co->sources);  // (view schemas) the next batch starts with no buffers attached
  for (auto& c : co->cols) c.view_base = c.cur_base = 0;
  b.cols.resize((size_t)co->ncols);
  for (int i = 0; i < co->ncols; ++i) {
    CoColumn& c = co->cols[i];
    ah_array_out& o = b.cols[i];
    ah_out_init(&o);
    if (c.generic) {  // GenericInProgressArray::finish (generic.rs:90-108): concat of the buffered pieces
      ah_status gs = AH_OK;
      if (c.pieces.size() == 1 && c.pieces[0].owner.use_count() == 1 && c.pieces[0].offset == 0 &&
          c.pieces[0].len == c.pieces[0].owner->out.length) {
        o = c.pieces[0].owner->out;  // the one piece IS the batch: hand its buffers over
        ah_out_init(&c.pieces[0].owner->out);
      } else {
        std::vector<ah_array_view> views;
        for (auto& p : c.pieces) views.push_back(piece_view(p));
        gs = ah_concat(ctx, (int32_t)views.size(), views.data(), &o);
      }
      c.pieces.clear();
      if (gs != AH_OK) {
        for (int k = 0; k < i; ++k) ah_array_release(ctx, &b.cols[k]);
        co->failed = true;
        return gs;
      }
      continue;
    }
    o.type = c.type;
    o.length = co->buffered;
    o.values = c.values;
    o.values_bytes = (int64_t)c.vbytes;
    o.validity = c.validity;  // kept or dropped when the null count is known (resolve_batch)
    o.validity_bytes = (int64_t)c.bbytes;
    o.null_count = -1;
    c.values = nullptr;
    c.validity = nullptr;
  }
  co->completed.push_back(std::move(b)), co->completed_batches += 1;
  co->buffered = 0;
  return AH_OK;
}

// `owners` (optional, per column): the array behind columns[i] is owned by the coalescer (a filtered or taken
// result): generic columns then reference it instead of copying the rows out
ah_status copy_rows_all(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t offset, int64_t len,
                        const std::shared_ptr<GenOwner>* owners = nullptr) {
  AH_TRY(ensure_capacity(ctx, co));
  note_contribution(co);
  for (int i = 0; i < co->ncols; ++i) {
    if (co->cols[i].generic) {
      if (owners && owners[i]) {
        co->cols[i].pieces.push_back(GenPiece{owners[i], offset, len});
        continue;
      }
      // the caller's buffers are only borrowed for this call (the reference keeps an Arc'ed slice): copy the rows out
      const ah_array_view& s = columns[i];
      const ah_array_view sl = slice_view(s.type, s.values, s.values_bit_offset, s.validity, s.validity_bit_offset, s.offsets,
                                          s.length, s.null_count, offset, len);
      ah_array_out piece;
      const ah_status gs = ah_concat(ctx, 1, &sl, &piece);
      if (gs != AH_OK) {
        if (i > 0) co->failed = true;
        return gs;
      }
      co->cols[i].pieces.push_back(GenPiece{std::make_shared<GenOwner>(ctx, piece), 0, len});
      continue;
    }
    const ah_status st = ah_copy_rows_into_acc(ctx, &columns[i], offset, len, co->cols[i].values, co->cols[i].validity,
                                               co->buffered, co->acc + (size_t)i * 64);
    if (st != AH_OK) {
      // earlier columns already have these rows (and their null counters bumped) while `buffered` has not advanced: a
      // retried push would double-count — the in-progress batch cannot be trusted any more
      if (i > 0) co->failed = true;
      return st;
    }
    if (co->cols[i].is_view && co->cols[i].cur_base > 0 && len > 0) {  // stream-ordered behind the copy
      const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, ah_ceil_div(len, 256)));
      rebase_views_kernel<<<g, 256, 0, ctx->stream>>>((uint4*)co->cols[i].values + co->buffered, len, co->cols[i].cur_base);
    }
  }
  return AH_OK;
}

ah_status check_columns(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows) {
  if (co->failed) return ah_fail(ctx, AH_INVALID_ARGUMENT, "BatchCoalescer: unusable after an earlier device error");
  for (int i = 0; i < co->ncols; ++i) {
    if (columns[i].type != co->cols[i].type)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "column %d has type %s, the coalescer expects %s", i,
                     ah_type_name(columns[i].type), ah_type_name(co->cols[i].type));
    if (columns[i].length != num_rows)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "column %d has %lld rows, the batch %lld", i, (long long)columns[i].length,
                     (long long)num_rows);
  }
  return AH_OK;
}

// the large batch goes out as it is: borrowed views of the caller's buffers (coalesce.rs:330-360, cases 1 and 2)
ah_status bypass(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows, uint64_t tag) {
  CoBatch b;
  b.rows = num_rows;
  b.tag = tag ? tag : 1;
  b.cols.resize((size_t)co->ncols);
  for (int i = 0; i < co->ncols; ++i) {
    ah_array_out& o = b.cols[i];
    ah_out_init(&o);
    o.type = columns[i].type;
    o.length = num_rows;
    o.values = const_cast<void*>(columns[i].values);
    o.values_bytes = co->cols[i].width > 0 ? num_rows * co->cols[i].width : 0;
    o.values_bit_offset = columns[i].values_bit_offset;
    o.offsets = const_cast<void*>(columns[i].offsets);
    o.flags = AH_OUT_BORROWED;
    if (columns[i].validity) {
      int64_t nulls = 0;
      AH_TRY(ah_resolve_null_count(ctx, &columns[i], &nulls));
      o.validity = const_cast<uint8_t*>(columns[i].validity);
      o.validity_bit_offset = columns[i].validity_bit_offset;
      o.null_count = nulls;
    }
  }
  if (co->has_views) b.sources.push_back(co->cur_seq);  // the batch leaves with its own buffers, indices untouched
  co->completed.push_back(std::move(b)), co->completed_batches += 1;
  return AH_OK;
}

// `offset0`: rows [0, offset0) of `columns` are already in the coalescer (a speculative scatter put them there)
ah_status push_batch_impl(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows, uint64_t tag,
                          int32_t* bypassed, int64_t offset0 = 0, const std::shared_ptr<GenOwner>* owners = nullptr) {
  if (num_rows - offset0 <= 0) return AH_OK;
  if (offset0 == 0 && co->limit >= 0 && num_rows > co->limit) {
    if (co->buffered == 0) {  // case 1
      if (bypassed) *bypassed = 1;
      return bypass(ctx, co, columns, num_rows, tag);
    }
    if (co->buffered > co->limit) {  // case 2: flush, then bypass
      AH_TRY(finish_buffered(ctx, co));
      if (bypassed) *bypassed = 1;
      return bypass(ctx, co, columns, num_rows, tag);
    }
  }
  int64_t remaining_rows = num_rows - offset0, offset = offset0;
  while (remaining_rows > co->target - co->buffered) {
    const int64_t room = co->target - co->buffered;
    AH_TRY(copy_rows_all(ctx, co, columns, offset, room, owners));
    co->buffered += room;
    offset += room;
    remaining_rows -= room;
    AH_TRY(finish_buffered(ctx, co));
  }
  if (remaining_rows > 0) AH_TRY(copy_rows_all(ctx, co, columns, offset, remaining_rows, owners));
  co->buffered += remaining_rows;
  if (co->buffered >= co->target) AH_TRY(finish_buffered(ctx, co));
  return AH_OK;
}

void release_batch(ah_context* ctx, CoBatch& b) {
  for (auto& o : b.cols) ah_array_release(ctx, &o);
  if (b.run_slab) ah_slab_unref(ctx, b.run_slab), b.run_slab = nullptr;
}

// the front entry of the completed queue as ONE batch: outs[0 .. ncols), rows, tag; pops the entry (a run: when used up)
ah_status pop_front_batch(ah_context* ctx, ah_coalescer* co, ah_array_out* outs, int64_t* num_rows, uint64_t* tag) {
  CoBatch& b = co->completed.front();
  if (b.run_n > 0) {
    SlabNulls& sn = *b.slab_nulls;
    if (!sn.arrived) {
      AH_TRY(ah_coalesce_wait(ctx, sn.seq));
      sn.arrived = true;
    }
    const int64_t j = b.run_next;
    for (int k = 0; k < co->ncols; ++k) {
      ah_array_out& o = outs[k];
      ah_out_init(&o);
      const int w = co->cols[k].width;
      o.type = co->cols[k].type;
      o.length = co->target;
      o.values = (char*)b.run_values[(size_t)k] + (size_t)j * co->target * w;
      o.values_bytes = co->target * w;
      ah_slab_slice(ctx, b.run_slab, o.values);
      const uint64_t nulls = __atomic_load_n(&sn.pin[(size_t)j * co->ncols + k], __ATOMIC_RELAXED);
      if (nulls > 0) {  // NullBufferBuilder::finish keeps a buffer only if a null was ever appended
        o.validity = b.run_valid[(size_t)k] + (size_t)j * co->target / 8;
        o.validity_bytes = co->target / 8;
        o.null_count = (int64_t)nulls;
        ah_slab_slice(ctx, b.run_slab, o.validity);
      }
    }
    *num_rows = co->target;
    if (tag) *tag = 0;
    if (++b.run_next == b.run_n) {
      ah_slab_unref(ctx, b.run_slab);
      b.run_slab = nullptr;
      co->completed.pop_front();
    }
    co->completed_batches -= 1;
    return AH_OK;
  }
  AH_TRY(resolve_pending(ctx, co, b));
  for (int i = 0; i < co->ncols; ++i) outs[i] = b.cols[i];
  *num_rows = b.rows;
  if (tag) *tag = b.tag;
  co->completed.pop_front();
  co->completed_batches -= 1;
  return AH_OK;
}

}  // namespace

extern "C" ah_status ah_coalescer_create(ah_context* ctx, int32_t n_columns, const ah_type* types, int64_t target_batch_size,
                                         ah_coalescer** out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out || n_columns < 1 || n_columns > 200 || !types || target_batch_size < 1) return AH_INVALID_ARGUMENT;
  *out = nullptr;
  hipSetDevice(ctx->device);
  auto* co = new ah_coalescer();
  co->ncols = n_columns;
  co->target = target_batch_size;
  co->cols.resize((size_t)n_columns);
  for (int i = 0; i < n_columns; ++i) {
    const int w = ah_type_width(types[i]);
    const bool generic = types[i] == AH_BOOL || types[i] == AH_UTF8 || types[i] == AH_LARGE_UTF8;
    if (w <= 0 && !generic) {
      delete co;
      return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "BatchCoalescer column type %s", ah_type_name(types[i]));
    }
    co->cols[i].type = types[i];
    co->cols[i].width = w;
    co->cols[i].generic = generic;
    co->cols[i].is_view = types[i] == AH_UTF8_VIEW || types[i] == AH_BINARY_VIEW;
    co->has_views = co->has_views || co->cols[i].is_view;
  }
  // acc: 64 appended-null counters per column
  const size_t acc_bytes = (size_t)n_columns * 8 * 64;
  ah_status st = ah_pool_alloc(ctx, acc_bytes, (void**)&co->acc);
  if (st == AH_OK && hipMemsetAsync(co->acc, 0, acc_bytes, ctx->stream) != hipSuccess)
    st = ah_fail(ctx, AH_HIP_ERROR, "coalescer counter reset failed");
  if (st == AH_OK) {
    co->ring_slots = std::max(4, 512 / n_columns);
    const size_t ring_words = (size_t)co->ring_slots * n_columns;
    void *hp = nullptr, *dp = nullptr;
    co->pin_bytes = (ring_words + 128 + 2 * 64 * 32) * 8;  // ... + 2 x 64 predicates x 32 quantile words (AH_FILTER_QUANTS)
    st = ah_pinned_alloc(ctx, co->pin_bytes, &hp, &dp);  // from the context's cache: hipHostMalloc is 0.1-0.3 ms
    if (st == AH_OK) {
      co->pin = (uint64_t*)hp;
      co->pin_dev = (uint64_t*)dp;
      co->cnt_pin = co->pin + ring_words;  // 2 x 64 count words behind the null-count ring
      co->cnt_pin_dev = co->pin_dev + ring_words;
      co->quant_pin = co->cnt_pin + 128;  // the predicates' quantile prefixes (filter_internal.hpp), 64 x 32 words per slot
      co->quant_pin_dev = co->cnt_pin_dev + 128;
    }
  }
  if (st != AH_OK) {
    ah_pool_free(ctx, co->acc);
    delete co;
    return st;
  }
  *out = co;
  return AH_OK;
}

extern "C" void ah_coalescer_destroy(ah_context* ctx, ah_coalescer* co) {
  ah_ctx_guard _guard(ctx);
  if (!co) return;
  if (slab_timing().on && slab_timing().pushes) {
    SlabTiming& t = slab_timing();
    fprintf(stderr, "arrow_hip: slab pushes %lld (%lld batches): host us — table build %.0f, upload + count enqueue %.0f, count wait %.0f, append %.0f\n",
            (long long)t.pushes, (long long)t.batches, t.build, t.upload_count, t.wait, t.append);
    t = SlabTiming{};
    t.on = true;
  }
  if (ctx) {
    (void)ah_stream_wait(ctx);  // scatters into the in-progress buffers may still be in flight
    for (auto& c : co->cols) {
      ah_out_free(ctx, c.values, c.vbytes);
      ah_out_free(ctx, c.validity, c.bbytes);
      c.pieces.clear();  // releases the owned arrays while the context is alive
    }
    for (auto& b : co->completed) release_batch(ctx, b);  // (the wait above passed every pending batch's kernel)
    ah_pool_free(ctx, co->acc);
    ah_pinned_free(ctx, co->pin, co->pin_bytes);
    for (auto& sp : co->slab_pin)
      if (sp.host) ah_pinned_free(ctx, sp.host, sp.bytes);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (auto& tb : co->slab_tbl) {
      ah_pool_free(ctx, tb.dev);
      if (tb.idle) (void)hipEventDestroy(tb.idle);
      if (tb.uploaded) (void)hipEventDestroy(tb.uploaded);
    }
  }
  delete co;
}

extern "C" void ah_coalescer_set_biggest_coalesce_batch_size(ah_coalescer* co, int64_t limit) {
  if (co) co->limit = limit;
}
extern "C" int64_t ah_coalescer_buffered_rows(const ah_coalescer* co) { return co ? co->buffered : 0; }
extern "C" int32_t ah_coalescer_completed_count(const ah_coalescer* co) {
  return co ? (int32_t)std::min<int64_t>(co->completed_batches, INT32_MAX) : 0;
}

// push_batch (coalesce.rs:296): `tag` identifies the caller's batch; *bypassed = 1 tells the caller that this very batch
// was queued untouched (large-batch bypass): it comes back from ah_coalescer_next_completed_batch with that tag and
// BORROWED buffers, so the caller keeps the input alive until then
extern "C" ah_status ah_coalescer_push_batch(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows,
                                             uint64_t tag, int32_t* bypassed) {
  ah_ctx_guard _guard(ctx);
  if (bypassed) *bypassed = 0;
  if (!ctx || !co || !columns || num_rows < 0) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  InputGuard input_guard(co, 1);
  AH_TRY(check_columns(ctx, co, columns, num_rows));
  AH_TRY(begin_input(ctx, co));
  return push_batch_impl(ctx, co, columns, num_rows, tag, bypassed);
}

namespace {

ah_status check_filter(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows, const ah_array_view* filter) {
  if (filter->type != AH_BOOL)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "filter predicate must be Boolean, got %s", ah_type_name(filter->type));
  if (filter->length > num_rows)
    return ah_fail(ctx, AH_INVALID_ARGUMENT, "Filter predicate of length %lld is larger than target array of length %lld",
                   (long long)filter->length, (long long)num_rows);
  return check_columns(ctx, co, columns, num_rows);
}

// push_batch_with_filter (coalesce.rs:229) once the predicate's count is known
// `done0`: positions [0, done0) of the filtered stream are already appended (speculative scatter; only without a
// bypass limit)
ah_status push_filtered_impl(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns, int64_t num_rows,
                             const ah_array_view* filter, ah_filter_predicate* p, uint64_t tag, int32_t* bypassed,
                             int64_t done0 = 0) {
  const int64_t selected = ah_filter_predicate_count(p);
  ah_status st = AH_OK;
  if (selected == 0 || done0 >= selected) return AH_OK;  // nothing (more) to append
  if (done0 == 0 && selected == num_rows && filter->length == num_rows) return push_batch_impl(ctx, co, columns, num_rows, tag, bypassed);
  const bool exceeds = co->limit >= 0 && selected > co->limit;
  int64_t done = done0;
  // (every window is a launch over the whole input batch: a push that would fill more than three output batches
  // goes through the materialised path below instead — one filter, then copies)
  // view schemas: always filter first, then copy (the appended views must pass through copy_rows_all's index shift)
  if (!exceeds && !co->has_views && co->ncols <= 8 && selected - done <= (co->target - co->buffered) + 2 * co->target) {
    // Same-shape nullable columns: ONE scatter launch per output batch the filtered rows land in — positions
    // [done, done + take) of the filtered stream go straight into the in-progress batch, also when the batch straddles
    // two (or more) output batches.  No intermediate filtered array, no host wait besides finish_buffered's.
    void* dv[8];
    uint8_t* db[8];
    while (done < selected && st == AH_OK) {
      st = ensure_capacity(ctx, co);
      if (st != AH_OK) break;
      for (int i = 0; i < co->ncols; ++i) dv[i] = co->cols[i].values, db[i] = co->cols[i].validity;
      const int64_t take = std::min(co->target - co->buffered, selected - done);
      const ah_status fs = ah_filter_apply_into_acc_cols(ctx, p, co->ncols, columns, dv, db, co->buffered,
                                                         (unsigned long long*)co->acc, done, done + take, 0, 0.0);
      if (fs == AH_NOT_YET_IMPLEMENTED && done == 0) break;  // shapes differ: the per-column paths below
      if (fs != AH_OK) return fs == AH_NOT_YET_IMPLEMENTED ? ah_fail(ctx, AH_INVALID_ARGUMENT, "coalescer: column shapes changed") : fs;
      co->buffered += take;
      done += take;
      if (co->buffered >= co->target) st = finish_buffered(ctx, co);
    }
    if (st != AH_OK || done == selected) return st;
  }
  const bool does_not_fit = selected - done > co->target - co->buffered;
  if (exceeds || does_not_fit || done > 0 || co->has_views) {  // materialise the filtered batch, then split it across output batches
    std::vector<ah_array_out> outs((size_t)co->ncols);
    std::vector<ah_array_view> views((size_t)co->ncols);
    for (auto& o : outs) ah_out_init(&o);
    // (the filtered batch leaves as an OWNED completed batch — the large-batch cases 1 / 2 below — rather than being copied)
    const bool leaves_as_is = co->limit >= 0 && selected > co->limit && (co->buffered == 0 || co->buffered > co->limit);
    for (int i = 0; i < co->ncols && st == AH_OK; ++i) {
      st = ah_filter_predicate_apply(ctx, p, &columns[i], &outs[i]);
      if (st == AH_OK && (co->cols[i].generic || leaves_as_is) && (outs[i].flags & AH_OUT_BORROWED)) {
        // the `All` strategy of a predicate shorter than its batch: a borrowed slice — a generic column would ADOPT it, and a
        // batch that leaves as it is would carry it into the completed queue: either way the caller's buffers would be used
        // past this call, and the caller is only told to keep BYPASSED batches alive.  Copy the rows out instead.
        // (round 6, found by AH_DEBUG_GUARD: the fixed-width columns of such a batch read released memory when fetched —
        // tests/test_gpu_parity.py::test_batch_coalescer_grouped_pushes_equal_single_pushes, limit 64, predicate 2 rows short)
        ah_array_release(ctx, &outs[i]);
        const ah_array_view& sv = columns[i];
        const ah_array_view sl = slice_view(sv.type, sv.values, sv.values_bit_offset, sv.validity, sv.validity_bit_offset, sv.offsets,
                                            sv.length, sv.null_count, 0, selected);
        st = ah_concat(ctx, 1, &sl, &outs[i]);
      }
    }
    if (st == AH_OK) {
      for (int i = 0; i < co->ncols; ++i) {
        ah_array_view& v = views[i];
        memset(&v, 0, sizeof v);
        v.type = outs[i].type;
        v.length = outs[i].length;
        v.null_count = outs[i].validity ? outs[i].null_count : 0;
        v.values = outs[i].values;
        v.values_bit_offset = outs[i].values_bit_offset;
        v.validity = outs[i].validity;
        v.validity_bit_offset = outs[i].validity_bit_offset;
      }
      // a filtered batch that is itself bypassed would hand out buffers this call owns: emit it as an OWNED batch
      if (leaves_as_is) {
        if (co->buffered > co->limit) st = finish_buffered(ctx, co);
        if (st == AH_OK) {
          CoBatch b;
          b.rows = selected;
          b.cols = outs;  // ownership moves to the completed queue
          for (auto& o : outs) ah_out_init(&o);
          if (co->has_views) b.sources.push_back(co->cur_seq);
          co->completed.push_back(std::move(b)), co->completed_batches += 1;
        }
      } else {
        // generic columns adopt their filtered array (shared between the output batches it straddles)
        std::vector<std::shared_ptr<GenOwner>> owners((size_t)co->ncols);
        for (int i = 0; i < co->ncols; ++i)
          if (co->cols[i].generic) {
            owners[i] = std::make_shared<GenOwner>(ctx, outs[i]);
            ah_out_init(&outs[i]);
          }
        st = push_batch_impl(ctx, co, views.data(), selected, 0, nullptr, done, owners.data());
        if (st == AH_OK) st = ah_stream_wait(ctx) == hipSuccess ? AH_OK : ah_fail(ctx, AH_HIP_ERROR, "coalescer copy failed");
      }
    }
    for (auto& o : outs) ah_array_release(ctx, &o);  // the copies out of them have finished (wait above)
    return st;
  }
  st = ensure_capacity(ctx, co);
  for (int i = 0; i < co->ncols && st == AH_OK; ++i) {
    if (co->cols[i].generic) {  // copy_rows_by_filter_from of GenericInProgressArray (generic.rs:81-88): keep the filtered array
      ah_array_out piece;
      st = ah_filter_predicate_apply(ctx, p, &columns[i], &piece);
      if (st == AH_OK && (piece.flags & AH_OUT_BORROWED)) {
        // a predicate SHORTER than its batch that selects every one of its rows takes filter's `All` strategy: the result
        // is a borrowed slice of the caller's buffers (filter.rs:546), which the coalescer may not keep past this call —
        // copy those rows out like an unfiltered push does
        ah_array_release(ctx, &piece);
        const ah_array_view& sv = columns[i];
        const ah_array_view sl = slice_view(sv.type, sv.values, sv.values_bit_offset, sv.validity, sv.validity_bit_offset, sv.offsets,
                                            sv.length, sv.null_count, 0, selected);
        st = ah_concat(ctx, 1, &sl, &piece);
      }
      if (st == AH_OK) co->cols[i].pieces.push_back(GenPiece{std::make_shared<GenOwner>(ctx, piece), 0, selected});
    } else {
      st = ah_filter_predicate_apply_into_acc(ctx, p, &columns[i], co->cols[i].values, co->cols[i].validity, co->buffered,
                                              co->acc + (size_t)i * 64);
    }
    if (st != AH_OK && i > 0) co->failed = true;  // some columns have the rows, `buffered` does not
  }
  if (st == AH_OK) {
    co->buffered += selected;
    if (co->buffered >= co->target) st = finish_buffered(ctx, co);
  }
  return st;
}

}  // namespace

// push_batch_with_filter (coalesce.rs:229)
extern "C" ah_status ah_coalescer_push_batch_with_filter(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns,
                                                         int64_t num_rows, const ah_array_view* filter, uint64_t tag,
                                                         int32_t* bypassed) {
  ah_ctx_guard _guard(ctx);
  if (bypassed) *bypassed = 0;
  if (!ctx || !co || !columns || !filter || num_rows < 0) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  InputGuard input_guard(co, 1);
  AH_TRY(check_filter(ctx, co, columns, num_rows, filter));
  AH_TRY(begin_input(ctx, co));
  ah_filter_predicate* p = nullptr;
  // One count pass for all columns; its K is the push's one host wait.  Without a bypass limit the rows that fit the
  // in-progress batch do not depend on K, so their scatter is enqueued BEFORE the wait (it clips itself to the rows
  // that exist): the GPU works through the count's round trip to the host instead of idling (~15 us per push).
  uint64_t seq = 0;
  bool enqueued = false;
  AH_TRY(ah_filter_predicate_begin(ctx, filter, &p, &seq, &enqueued));
  ah_status st = AH_OK;
  int64_t room = 0;
  bool speculated = false;
  if (enqueued && co->limit < 0 && co->ncols <= 8 && !co->has_views) {
    st = ensure_capacity(ctx, co);
    if (st == AH_OK) {
      void* dv[8];
      uint8_t* db[8];
      for (int i = 0; i < co->ncols; ++i) dv[i] = co->cols[i].values, db[i] = co->cols[i].validity;
      room = co->target - co->buffered;
      const ah_status fs = ah_filter_apply_into_acc_cols(ctx, p, co->ncols, columns, dv, db, co->buffered,
                                                         (unsigned long long*)co->acc, 0, room, 1, co->selectivity);
      if (fs == AH_OK) speculated = true;
      else if (fs != AH_NOT_YET_IMPLEMENTED) st = fs;
    }
  }
  if (st == AH_OK) st = ah_filter_predicate_end(ctx, p, seq, enqueued);
  if (st == AH_OK) {
    const int64_t selected = ah_filter_predicate_count(p);
    if (filter->length > 0) co->selectivity = (double)selected / (double)filter->length;
    int64_t done0 = 0;
    if (speculated) {
      done0 = std::min(selected, room);
      co->buffered += done0;
      if (co->buffered >= co->target) st = finish_buffered(ctx, co);
    }
    if (st == AH_OK) st = push_filtered_impl(ctx, co, columns, num_rows, filter, p, tag, bypassed, done0);
  }
  // the speculative scatter may have written rows (and validity bits) past `buffered_rows` that the bookkeeping
  // never accounted for: the in-progress batch cannot be trusted any more
  if (st != AH_OK && speculated) co->failed = true;
  ah_filter_predicate_free(ctx, p);
  return st;
}

namespace {
bool group_fusable(const ah_coalescer* co, int n, const ah_array_view* columns, const ah_array_view* filters);
ah_status append_group(ah_context* ctx, ah_coalescer* co, int m, const ah_array_view* columns, const int64_t* num_rows,
                       const ah_array_view* filters, const uint64_t* tags, int32_t* bypassed, ah_filter_predicate* const* preds,
                       bool fusable);
}  // namespace

// ------------------------------------------------------------------------------------------------------- slab push
// Round 5: the reference's operating point is 8192-row input batches (coalesce.rs:172-173 "Typical
// values are 4096 or 8192 rows"; arrow/benches/coalesce_kernels.rs:34) — 122 000 pushes and 12 000 output batches per 1e9
// rows at 10 % selected.  A grouped push of ANY number of batches is one table upload, one count launch, one scan, and one
// scatter launch per destination (per value width):
//   * the rows that top up the in-progress batch go into its buffers (positions [0, room) of the push's filtered stream);
//   * every further row goes into ONE slab — a single allocation holding ceil(rest / target) output batches back to back,
//     so output batch j simply IS rows [j * target, (j + 1) * target) of the slab (values and validity bits alike: the target
//     must be a multiple of 64), handed out as slices that keep the block alive (ah_slab, context.hip); the slab's last,
//     partial batch becomes the new in-progress batch;
//   * NULL rows are counted from the output bitmaps behind the scatter: per full slab batch into pinned words the batches
//     share (read when a batch is fetched), for the top-up and the tail into the in-progress counters.
// The host waits once per push: for the wave prefixes (positions of every 65 536-row group's first selected row), from
// which it knows K, how the rows split between top-up and slab, and which tiles each launch needs.
namespace {

struct SlabPush {
  int n = 0, slot = -1, tbl = -1;
  int64_t total_rows = 0;
  ah_tbl_push t{};
  void* dev_block = nullptr;
  uint64_t seq = 0;
  std::vector<int64_t> chunk0, tile0;  // per batch (n + 1 entries): first 1024-row chunk of the push's numbering / first 4096-row tile
  bool aligned16 = true;
};

bool slab_enabled() {
  const char* e = getenv("AH_COALESCE_SLAB");  // "0": the round-4 grouped path (A/B runs); read per call
  return !(e && e[0] == '0');
}

// the slab path serves: no bypass limit, fixed-width columns of 1 / 2 / 4 / 8 bytes (at most 8), a target that is a multiple
// of 64, the built-in allocator, a synchronous context
bool slab_eligible(ah_context* ctx, const ah_coalescer* co, int n, const ah_array_view* columns, const int64_t* num_rows,
                   const ah_array_view* filters) {
  if (!slab_enabled() || n < 2 || co->limit >= 0 || co->has_views || co->ncols > AH_TBL_MAX_COLS || (co->target & 63)) return false;
  if (ctx->alloc || ctx->deferred || ctx->capturing) return false;
  for (int k = 0; k < co->ncols; ++k) {
    const int w = co->cols[k].width;
    if (co->cols[k].generic || !(w == 1 || w == 2 || w == 4 || w == 8)) return false;
  }
  int64_t rows = 0;
  for (int i = 0; i < n; ++i) {  // (a predicate longer than its batch: check_filter reports it)
    if (filters[i].length > num_rows[i] || (filters[i].length >> 12) >= INT32_MAX) return false;
    rows += filters[i].length;
  }
  // A handful of very large batches keeps the round-4 grouped path: per-batch predicate objects with quantile prefixes and at
  // most 8 segments per launch — measured at 2^24-row batches, 8 per push: 3.07-3.24 ms per 1e9 rows against 3.27-3.55 here
  // (profiles/r05_coalesce_sweep.md).  Below ~4 Mi rows per batch, or with more batches than that path groups, the tables win.
  // (and only while the push fills a handful of output batches: that path launches once per output window — 12 000 launches and
  // 147 ms per 1e9 rows at a target of 8192)
  static const char* force = getenv("AH_COALESCE_SLAB");
  const double est_out_batches = (double)rows * co->selectivity / (double)co->target;
  if (!(force && force[0] == '1') && n <= 8 && rows >= (int64_t)n << 22 && est_out_batches <= 4.0) return false;
  return true;
}

ah_status slab_pin_reserve(ah_context* ctx, SlabPin& sp, size_t bytes) {
  if (sp.bytes >= bytes) return AH_OK;
  if (sp.host) ah_pinned_free(ctx, sp.host, sp.bytes);
  sp.host = sp.dev = nullptr;
  sp.bytes = 0;
  size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
  AH_TRY(ah_pinned_alloc(ctx, want, &sp.host, &sp.dev));
  sp.bytes = want;
  return AH_OK;
}

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// the most output one slab holds (a kept batch pins its slab)
size_t slab_cap_bytes() {  // (read per push, not cached: the tests shrink it to put many slabs into a small push)
  const char* e = getenv("AH_COALESCE_SLAB_BYTES");
  const long long v = e ? atoll(e) : 0;
  return v > 0 ? (size_t)v : (size_t)256 << 20;
}

// Table uploads run on the context's own stream, in front of the count.  AH_COALESCE_COPY_STREAM=1 moves them to a second
// stream beside the previous push's scatter (2-3 % at 8192-row batches, profiles/r05_coalesce_sweep.md): OFF by default since
// round 6 — it is the one place where a pool block changes stream ownership through hand-placed events, it landed minutes
// before the round-5 driver run that faulted, and although nothing has tied it to that fault (profiles/r06_crash.md) the gain
// does not pay for the doubt.
bool slab_copy_stream_enabled() {
  static const bool on = [] {
    const char* e = getenv("AH_COALESCE_COPY_STREAM");
    return e && e[0] == '1';
  }();
  return on;
}

ah_status slab_tables_reserve(ah_context* ctx, SlabTables& tb, size_t bytes, bool* fresh) {
  if (!ctx->copy_stream && slab_copy_stream_enabled() &&
      hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
    return ah_fail(ctx, AH_HIP_ERROR, "copy stream creation failed");
  if (!tb.idle && (hipEventCreateWithFlags(&tb.idle, hipEventDisableTiming) != hipSuccess ||
                   hipEventCreateWithFlags(&tb.uploaded, hipEventDisableTiming) != hipSuccess))
    return ah_fail(ctx, AH_HIP_ERROR, "coalescer event creation failed");
  if (tb.bytes >= bytes) return AH_OK;
  ah_pool_free(ctx, tb.dev);  // (its readers ran on the context's stream: pool reuse is ordered behind them)
  tb.dev = nullptr;
  tb.bytes = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
  AH_TRY(ah_pool_alloc(ctx, want, &tb.dev));
  tb.bytes = want;
  *fresh = true;
  return AH_OK;
}

// the push that read table buffer `tbl` has enqueued its last kernel (or none): the buffer may be rewritten behind this point
void slab_tables_release(ah_context* ctx, ah_coalescer* co, int tbl) {
  if (tbl < 0) return;
  SlabTables& tb = co->slab_tbl[tbl];
  if (ctx->copy_stream && hipEventRecord(tb.idle, ctx->stream) == hipSuccess) tb.used = true;
  tb.owner_live = false;
}

// tables built and uploaded, count + scan enqueued, mailbox posted: nothing waited for
ah_status slab_begin(ah_context* ctx, ah_coalescer* co, int n, const ah_array_view* columns, const int64_t* num_rows,
                     const ah_array_view* filters, SlabPush** out) {
  *out = nullptr;
  const int slot = !co->slab_pin[0].busy ? 0 : (!co->slab_pin[1].busy ? 1 : -1);
  if (slot < 0) return AH_NOT_YET_IMPLEMENTED;  // two pushes already in flight
  std::unique_ptr<SlabPush> sp(new SlabPush());
  sp->n = n;
  sp->slot = slot;
  sp->chunk0.resize((size_t)n + 1);
  sp->tile0.resize((size_t)n + 1);
  SlabTiming& tm = slab_timing();
  const double t_0 = tm.on ? now_us() : 0;
  int64_t nchunks = 0, ntiles = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t len = filters[i].length;
    sp->chunk0[(size_t)i] = nchunks;
    sp->tile0[(size_t)i] = ntiles;
    static_assert(AH_FILTER_CHUNK_ROWS == 1024, "chunk shift");
    nchunks += (len + AH_FILTER_CHUNK_ROWS - 1) >> 10;  // (shifts: this loop runs 122 000 times per 1e9 rows at 8192-row batches)
    ntiles += (len + 4095) >> 12;
    sp->total_rows += len;
  }
  sp->chunk0[(size_t)n] = nchunks;
  sp->tile0[(size_t)n] = ntiles;
  const int64_t nwaves = (nchunks + 63) >> 6;  // a count wave owns 64 consecutive chunks of the push, across batches
  if (nwaves == 0) {  // every predicate is empty: nothing to count, nothing to append
    *out = sp.release();
    (*out)->slot = -1;
    return AH_OK;
  }
  // pinned block: [nwaves + 1 prefix words][segs][cols][chunk_seg][tiles]; device: the same tables in one of the coalescer's
  // three table buffers, and a pool block with chunk_prefix, wave_total, wave_prefix
  const size_t b_pref = up256(((size_t)nwaves + 1) * 8), b_seg = up256((size_t)n * sizeof(ah_tbl_seg)),
               b_col = up256((size_t)n * co->ncols * sizeof(ah_tbl_col)),
               b_cs = up256((size_t)nchunks * 4), b_til = up256((size_t)ntiles * sizeof(ah_tbl_tile));
  const size_t b_tables = b_seg + b_col + b_cs + b_til;
  const size_t b_cp = up256((size_t)nchunks * 4), b_wt = up256((size_t)nwaves * 4), b_wp = up256(((size_t)nwaves + 1) * 8);
  SlabPin& pin = co->slab_pin[slot];
  AH_TRY(slab_pin_reserve(ctx, pin, b_pref + b_tables));
  int tbl = -1;
  for (int q = 0; q < 3 && tbl < 0; ++q)
    if (!co->slab_tbl[(co->slab_tbl_next + q) % 3].owner_live) tbl = (co->slab_tbl_next + q) % 3;
  if (tbl < 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "BatchCoalescer: no table buffer free");  // (two pushes in flight at most)
  SlabTables& tb = co->slab_tbl[tbl];
  bool fresh = false;
  AH_TRY(slab_tables_reserve(ctx, tb, b_tables, &fresh));
  AH_TRY(ah_pool_alloc(ctx, b_cp + b_wt + b_wp, &sp->dev_block));
  char* hs = (char*)pin.host + b_pref;
  auto* segs = (ah_tbl_seg*)hs;
  auto* tcols = (ah_tbl_col*)(hs + b_seg);
  auto* chunk_seg = (int32_t*)(hs + b_seg + b_col);
  auto* tiles = (ah_tbl_tile*)(hs + b_seg + b_col + b_cs);
  int64_t ti = 0;
  for (int i = 0; i < n; ++i) {
    const ah_array_view& f = filters[i];
    ah_tbl_seg& sg = segs[i];
    sg.mask = make_bitview(f.values, f.values_bit_offset);
    sg.mask_valid = (f.validity && f.null_count != 0) ? make_bitview(f.validity, f.validity_bit_offset) : BitView{nullptr, 0};
    sg.len = f.length;
    sg.chunk0 = sp->chunk0[(size_t)i];
    for (int k = 0; k < co->ncols; ++k) {
      const ah_array_view& v = columns[(size_t)i * co->ncols + k];
      ah_tbl_col& tc = tcols[(size_t)i * co->ncols + k];
      tc.values = v.values;
      tc.vvalid = (v.validity && v.null_count != 0) ? make_bitview(v.validity, v.validity_bit_offset) : BitView{nullptr, 0};
      if (((uintptr_t)v.values) & 15) sp->aligned16 = false;
    }
    for (int64_t c = sp->chunk0[(size_t)i]; c < sp->chunk0[(size_t)i + 1]; ++c) chunk_seg[c] = i;
    const int64_t nt = sp->tile0[(size_t)i + 1] - sp->tile0[(size_t)i];
    for (int64_t q = 0; q < nt; ++q) tiles[ti++] = ah_tbl_tile{i, (int32_t)q};
  }
  char *db = (char*)tb.dev, *ws = (char*)sp->dev_block;
  sp->t.segs = (const ah_tbl_seg*)db;
  sp->t.cols = (const ah_tbl_col*)(db + b_seg);
  sp->t.ncols = co->ncols;
  sp->t.chunk_seg = (const int32_t*)(db + b_seg + b_col);
  sp->t.tiles = (const ah_tbl_tile*)(db + b_seg + b_col + b_cs);
  sp->t.nsegs = n, sp->t.nchunks = nchunks, sp->t.nwaves = nwaves, sp->t.ntiles = ntiles;
  sp->t.chunk_prefix = (uint32_t*)ws;
  sp->t.wave_total = (uint32_t*)(ws + b_cp);
  sp->t.wave_prefix = (unsigned long long*)(ws + b_cp + b_wt);
  ah_status st = AH_OK;
  const double t_1 = tm.on ? now_us() : 0;
  // the upload: on the copy stream, behind the last kernel that read this buffer; the count waits for it
  const bool side = ctx->copy_stream != nullptr;
  hipStream_t cs = side ? ctx->copy_stream : ctx->stream;
  bool ok = true;
  if (side && fresh) ok = hipEventRecord(tb.idle, ctx->stream) == hipSuccess;  // the block's previous life ran on the context's stream
  if (ok && side && (fresh || tb.used)) ok = hipStreamWaitEvent(cs, tb.idle, 0) == hipSuccess;
  ok = ok && hipMemcpyAsync(db, hs, b_tables, hipMemcpyHostToDevice, cs) == hipSuccess;
  if (ok && side) ok = hipEventRecord(tb.uploaded, cs) == hipSuccess && hipStreamWaitEvent(ctx->stream, tb.uploaded, 0) == hipSuccess;
  if (!ok) st = ah_fail(ctx, AH_HIP_ERROR, "coalescer table upload failed");
  if (st == AH_OK) st = ah_filter_table_count(ctx, sp->t, (uint64_t*)pin.dev, &sp->seq);
  if (st != AH_OK) {
    if (side) (void)hipStreamSynchronize(cs);
    (void)ah_stream_wait(ctx);
    ah_pool_free(ctx, sp->dev_block);
    return st;
  }
  if (tm.on) tm.build += t_1 - t_0, tm.upload_count += now_us() - t_1, tm.pushes += 1, tm.batches += n;
  tb.owner_live = true;
  sp->tbl = tbl;
  co->slab_tbl_next = (tbl + 1) % 3;
  pin.busy = true;
  *out = sp.release();
  return AH_OK;
}

void slab_abort(ah_context* ctx, ah_coalescer* co, SlabPush* sp) {
  if (!sp) return;
  if (sp->slot >= 0) {
    (void)ah_stream_wait(ctx);  // the count kernels may still be writing the tables' block and the pinned words
    co->slab_pin[sp->slot].busy = false;
    slab_tables_release(ctx, co, sp->tbl);
  }
  ah_pool_free(ctx, sp->dev_block);
  delete sp;
}

// waits for the wave prefixes and appends the push's selected rows: the same output batches, in the same order, as n calls
// of push_batch_with_filter.  Consumes `sp`.
ah_status slab_end(ah_context* ctx, ah_coalescer* co, SlabPush* sp_raw) {
  std::unique_ptr<SlabPush> sp(sp_raw);
  if (sp->slot < 0) return AH_OK;  // nothing was enqueued (every predicate empty)
  SlabPin& pin = co->slab_pin[sp->slot];
  SlabTiming& tm = slab_timing();
  const double t_0 = tm.on ? now_us() : 0;
  struct AppendTimer {
    SlabTiming& t;
    double t1 = 0;
    ~AppendTimer() {
      if (t.on && t1 > 0) t.append += now_us() - t1;
    }
  } append_timer{tm};
  const bool later_work = sp->seq != ctx->mail_seq || ctx->inflight;
  const hipError_t we = ah_mail_wait(ctx, sp->seq);
  if (tm.on) append_timer.t1 = now_us(), tm.wait += append_timer.t1 - t_0;
  if (later_work && ctx->wait_mode != 1) ctx->inflight = true;
  pin.busy = false;
  struct FreeBlock {  // (pool reuse is stream-ordered behind the launches below; the table buffer's `idle` event likewise)
    ah_context* c;
    ah_coalescer* co;
    void* b;
    int tbl;
    ~FreeBlock() {
      ah_pool_free(c, b);
      slab_tables_release(c, co, tbl);
    }
  } fb{ctx, co, sp->dev_block, sp->tbl};
  if (we != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "coalescer count failed: %s", hipGetErrorString(we));
  if (co->failed) return ah_fail(ctx, AH_INVALID_ARGUMENT, "BatchCoalescer: unusable after an earlier device error");
  const uint64_t* P = (const uint64_t*)pin.host;  // [nwaves + 1]
  const int64_t nwaves = sp->t.nwaves, K = (int64_t)__atomic_load_n(&P[nwaves], __ATOMIC_RELAXED);
  if (sp->total_rows > 0) co->selectivity = (double)K / (double)sp->total_rows;
  if (K == 0) return AH_OK;
  const bool sparse = K * 32 <= sp->total_rows, skip = (double)K < 0.12 * (double)sp->total_rows;
  // the tile that holds global chunk gc (a batch's tiles are 4 chunks each from ITS first chunk: a tile can straddle two waves)
  auto tile_with_chunk = [&](int64_t gc) -> int64_t {
    const int64_t i = (int64_t)(std::upper_bound(sp->chunk0.begin(), sp->chunk0.end(), gc) - sp->chunk0.begin()) - 1;  // its batch
    return sp->tile0[(size_t)i] + ((gc - sp->chunk0[(size_t)i]) >> 2);
  };
  auto tiles_before_wave = [&](int64_t w) -> int64_t {  // one past the last tile with a chunk of waves [0, w)
    if (w <= 0) return 0;
    return tile_with_chunk(std::min(w * 64, sp->t.nchunks) - 1) + 1;
  };
  auto first_tile_of_wave = [&](int64_t w) -> int64_t {  // the first tile with a chunk of wave w (w == nwaves: one past the last)
    return w >= nwaves ? sp->t.ntiles : tile_with_chunk(w * 64);
  };
  auto first_wave_reaching = [&](int64_t pos) -> int64_t {  // smallest w with P[w] >= pos  (P is non-decreasing)
    int64_t lo = 0, hi = nwaves;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)P[mid] >= pos) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  };
  // launch columns grouped by value width
  int widths[4] = {1, 2, 4, 8};
  auto scatter_all = [&](void* const* dv, uint8_t* const* db, int64_t tile_lo, int64_t tile_hi, int64_t win_lo, int64_t win_hi,
                         int64_t out_base) -> ah_status {
    for (int w : widths) {
      int idx[AH_TBL_MAX_COLS], m = 0;
      ah_tbl_dst dst[AH_TBL_MAX_COLS];
      for (int k = 0; k < co->ncols; ++k)
        if (co->cols[k].width == w) {
          idx[m] = k;
          dst[m].out_values = dv[k];
          dst[m].out_valid = (unsigned long long*)db[k];
          ++m;
        }
      if (m == 0) continue;
      AH_TRY(ah_filter_table_scatter(ctx, sp->t, w, m, idx, dst, tile_lo, tile_hi, win_lo, win_hi, out_base, sp->aligned16, sparse, skip));
    }
    return AH_OK;
  };
  ah_status st = AH_OK;
  bool enq = false;
  // 1. top up the in-progress batch
  int64_t take1 = 0;
  const bool have_buf = co->cols[0].values != nullptr;
  if (have_buf && co->buffered < co->target) {
    take1 = std::min(K, co->target - co->buffered);
    void* dv[AH_TBL_MAX_COLS];
    uint8_t* db[AH_TBL_MAX_COLS];
    const unsigned long long* bits[AH_TBL_MAX_COLS];
    unsigned long long* slots[AH_TBL_MAX_COLS];
    for (int k = 0; k < co->ncols; ++k) {
      dv[k] = co->cols[k].values, db[k] = co->cols[k].validity;
      bits[k] = (const unsigned long long*)co->cols[k].validity;
      slots[k] = (unsigned long long*)co->acc + (size_t)k * 64;
    }
    const int64_t w_hi = first_wave_reaching(take1);  // waves [0, w_hi) start before position take1
    st = scatter_all(dv, db, 0, tiles_before_wave(w_hi), 0, take1, co->buffered);
    enq = st == AH_OK;
    if (st == AH_OK) st = ah_filter_count_nulls_range(ctx, co->ncols, bits, slots, co->buffered, take1);
    if (st == AH_OK) {
      co->buffered += take1;
      if (co->buffered >= co->target) st = finish_buffered(ctx, co);
    }
  }
  // 2. everything else: slabs of whole output batches.  A slab stays allocated until its LAST slice is released, so a consumer
  // that keeps one 8192-row batch of a push would otherwise pin the push's whole output (GBs for a 2^27-row push; the reference
  // allocates every batch on its own, coalesce.rs:560-600).  Slabs are therefore capped (AH_COALESCE_SLAB_BYTES, default
  // 256 MiB; at least one batch): a kept batch retains at most one cap's worth.  Every slab costs its own scatter launch,
  // bitmap reset and NULL-count launches (~20 us): at 64 MiB (75 us of scatter per slab) that was 0.3 ms per 1e9 rows of
  // 8192-row batches, at 256 MiB it is within 2 % of uncapped (profiles/r06_coalesce_slab_cap.md).
  const int64_t rest = K - take1;
  if (st == AH_OK && rest > 0) {
    size_t per_batch = 0;
    for (int k = 0; k < co->ncols; ++k) per_batch += (size_t)co->target * co->cols[k].width + (size_t)co->target / 8;
    const int64_t cap_batches = std::max<int64_t>(1, (int64_t)(slab_cap_bytes() / std::max<size_t>(per_batch, 1)));
    const int64_t nb_total = ah_ceil_div(rest, co->target);
    for (int64_t b0 = 0; b0 < nb_total && st == AH_OK; b0 += cap_batches) {
      const int64_t nb = std::min(cap_batches, nb_total - b0);
      const int64_t lo = take1 + b0 * co->target, hi = std::min(K, lo + nb * co->target);
      const int64_t nfull = (hi - lo) / co->target, tail = (hi - lo) % co->target;  // (a tail only in the push's last slab)
      size_t off = 0, voff[AH_TBL_MAX_COLS], boff[AH_TBL_MAX_COLS];
      for (int k = 0; k < co->ncols; ++k) {
        voff[k] = off;
        off += up256((size_t)nb * co->target * co->cols[k].width);
      }
      const size_t bits_begin = off;
      for (int k = 0; k < co->ncols; ++k) {
        boff[k] = off;
        off += up256((size_t)nb * co->target / 8);
      }
      ah_slab* slab = nullptr;
      st = ah_slab_create(ctx, off, &slab);
      if (st != AH_OK) break;
      char* base = (char*)slab->block;
      void* dv[AH_TBL_MAX_COLS];
      uint8_t* db[AH_TBL_MAX_COLS];
      const unsigned long long* bits[AH_TBL_MAX_COLS];
      unsigned long long* slots[AH_TBL_MAX_COLS];
      for (int k = 0; k < co->ncols; ++k) {
        dv[k] = base + voff[k], db[k] = (uint8_t*)(base + boff[k]);
        bits[k] = (const unsigned long long*)db[k];
        slots[k] = (unsigned long long*)co->acc + (size_t)k * 64;
      }
      if (hipMemsetAsync(base + bits_begin, 0, off - bits_begin, ctx->stream) != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "coalescer bitmap reset failed");
      // waves that can hold a position in [lo, hi): from the last wave that starts at or before `lo` ...
      int64_t w_lo = first_wave_reaching(lo + 1);  // first wave starting AFTER lo ...
      w_lo = w_lo > 0 ? w_lo - 1 : 0;               // ... so the one before it holds position lo
      // ... to the last wave that starts before `hi`
      const int64_t tile_hi = hi >= K ? sp->t.ntiles : tiles_before_wave(first_wave_reaching(hi));
      if (st == AH_OK) st = scatter_all(dv, db, first_tile_of_wave(w_lo), tile_hi, lo, hi, 0);
      enq = enq || st == AH_OK;
      std::shared_ptr<SlabNulls> sn;
      if (st == AH_OK && nfull > 0) {
        sn = std::make_shared<SlabNulls>();
        sn->ctx = ctx;
        sn->bytes = (size_t)nfull * co->ncols * 8;
        void *hp = nullptr, *dp = nullptr;
        st = ah_pinned_alloc(ctx, sn->bytes, &hp, &dp);
        if (st == AH_OK) {
          sn->pin = (uint64_t*)hp;
          st = ah_filter_count_nulls_batches(ctx, co->ncols, bits, co->target, nfull, co->target, (unsigned long long*)dp, &sn->seq);
        }
      }
      if (st == AH_OK && tail > 0) st = ah_filter_count_nulls_range(ctx, co->ncols, bits, slots, nfull * co->target, tail);
      if (st == AH_OK) {
        if (nfull > 0) {  // ONE queue entry for the nfull whole batches: materialised when fetched (pop_front_batch)
          CoBatch b;
          b.rows = co->target;
          b.pending = false;
          b.ring = -1;
          b.slab_nulls = sn;
          b.run_n = nfull;
          b.run_slab = slab;
          ah_slab_slice(ctx, slab, nullptr);  // the run's own reference
          for (int k = 0; k < co->ncols; ++k) b.run_values.push_back(dv[k]), b.run_valid.push_back(db[k]);
          co->completed.push_back(std::move(b));
          co->completed_batches += nfull;
        }
        if (tail > 0) {  // the slab's last, partial batch is the new in-progress batch (capacity: a whole target)
          for (int k = 0; k < co->ncols; ++k) {
            CoColumn& c = co->cols[k];
            c.values = (char*)dv[k] + (size_t)nfull * co->target * c.width;
            c.validity = db[k] + (size_t)nfull * co->target / 8;
            c.vbytes = (size_t)co->target * c.width;
            c.bbytes = (size_t)co->target / 8;
            ah_slab_slice(ctx, slab, c.values);
            ah_slab_slice(ctx, slab, c.validity);
          }
          co->buffered = tail;
        }
      }
      ah_slab_unref(ctx, slab);  // the slices keep the block alive (none were made on failure: it goes back to the pool)
    }
  }
  if (st != AH_OK && enq) co->failed = true;  // rows were enqueued that the bookkeeping may not cover
  return st;
}


}  // namespace

// `n` filtered pushes in one call, for a host that has several batches queued: the n count passes are enqueued
// back to back and read with ONE wait (ah_filter_predicates_build), then every batch is appended exactly as n calls
// of ah_coalescer_push_batch_with_filter would — the same output batches in the same order — without the GPU idling
// for a count round trip between batches.  columns: n x n_columns views, batch-major.
extern "C" ah_status ah_coalescer_push_batches_with_filters(ah_context* ctx, ah_coalescer* co, int32_t n,
                                                            const ah_array_view* columns, const int64_t* num_rows,
                                                            const ah_array_view* filters, const uint64_t* tags,
                                                            int32_t* bypassed) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || n < 0 || (n > 0 && (!columns || !num_rows || !filters))) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  InputGuard input_guard(co, n);
  for (int i = 0; i < n; ++i) {
    if (bypassed) bypassed[i] = 0;
    if (num_rows[i] < 0) return AH_INVALID_ARGUMENT;
    AH_TRY(check_filter(ctx, co, columns + (size_t)i * co->ncols, num_rows[i], &filters[i]));
  }
  if (slab_eligible(ctx, co, n, columns, num_rows, filters)) {  // any number of batches: one count, one scatter per destination
    SlabPush* sp = nullptr;
    const ah_status bs = slab_begin(ctx, co, n, columns, num_rows, filters, &sp);
    if (bs == AH_OK) return slab_end(ctx, co, sp);
    if (bs != AH_NOT_YET_IMPLEMENTED) return bs;
  }
  // Grouped scatter: when there is no bypass limit and every column of every batch has the same shape (one width,
  // all nullable), the batches that land in one output window leave through ONE launch (up to 8 per launch) — 2^24-row
  // batches pay the ramp and tail of a launch once per window instead of once per batch.  Same output batches, same order.
  const bool fusable = group_fusable(co, n, columns, filters);
  ah_status st = AH_OK;
  for (int base = 0; base < n && st == AH_OK; base += 64) {
    const int m = std::min(64, n - base);
    std::vector<ah_filter_predicate*> preds((size_t)m, nullptr);
    // the counts through the coalescer's own pinned words when a slot is free (no pipelined push in flight there): the same
    // one wait, and the predicates come back with their quantile prefixes, so cut batches launch only the tiles they need
    const int slot = fusable ? (!co->cnt_busy[0] ? 0 : (!co->cnt_busy[1] ? 1 : -1)) : -1;
    st = AH_NOT_YET_IMPLEMENTED;
    if (slot >= 0) {
      uint64_t seq = 0;
      st = ah_filter_predicates_begin(ctx, m, filters + base, preds.data(), co->cnt_pin_dev + 64 * slot, &seq,
                                      co->quant_pin_dev + (size_t)64 * 32 * slot);
      if (st == AH_OK) {
        st = ah_filter_predicates_end(ctx, m, preds.data(), co->cnt_pin + 64 * slot, seq, co->quant_pin + (size_t)64 * 32 * slot);
        if (st != AH_OK) {
          (void)ah_stream_wait(ctx);  // the count kernels may still be writing the blocks freed below
          for (auto*& p : preds) ah_filter_predicate_free(ctx, p), p = nullptr;
        }
      }
    }
    if (st == AH_NOT_YET_IMPLEMENTED) st = ah_filter_predicates_build(ctx, m, filters + base, preds.data());  // (any shape)
    if (st == AH_OK)
      st = append_group(ctx, co, m, columns + (size_t)base * co->ncols, num_rows + base, filters + base, tags ? tags + base : nullptr,
                        bypassed ? bypassed + base : nullptr, preds.data(), fusable);
    for (auto* p : preds) ah_filter_predicate_free(ctx, p);
  }
  return st;
}

namespace {

// the grouped scatter is possible when there is no bypass limit and every column of every batch has the same shape
bool group_fusable(const ah_coalescer* co, int n, const ah_array_view* columns, const ah_array_view* filters) {
  bool fusable = co->limit < 0 && co->ncols <= 8 && n > 1 && !co->has_views;
  // generic columns (Boolean: width 0, Utf8 / LargeUtf8: width -1) keep piece lists, not in-progress buffers: they take
  // the per-batch path (an all-Boolean or all-Utf8 schema has "one width" too)
  for (int k = 0; k < co->ncols && fusable; ++k)
    if (co->cols[k].generic || co->cols[k].width <= 0) fusable = false;
  for (int i = 0; i < n && fusable; ++i)
    for (int k = 0; k < co->ncols; ++k) {
      const ah_array_view& v = columns[(size_t)i * co->ncols + k];
      if (co->cols[k].width != co->cols[0].width || !v.validity || v.null_count == 0 || filters[i].length > v.length) fusable = false;
    }
  return fusable;
}

// `m` batches whose predicates (with their counts) are built: appended exactly as m calls of push_batch_with_filter would
ah_status append_group(ah_context* ctx, ah_coalescer* co, int m, const ah_array_view* columns, const int64_t* num_rows,
                       const ah_array_view* filters, const uint64_t* tags, int32_t* bypassed, ah_filter_predicate* const* preds,
                       bool fusable) {
  ah_status st = AH_OK;
  const int base = 0;
  {
    if (st == AH_OK && fusable) {
      const ah_filter_predicate* seg_p[8];
      const ah_array_view* seg_c[8];
      int64_t seg_lo[8], seg_hi[8], seg_base[8];
      int nseg = 0;
      int64_t virt = co->buffered;  // rows of the in-progress batch once the pending segments have run
      bool enq = false;
      auto flush = [&]() -> ah_status {
        if (nseg == 0) return AH_OK;
        void* dv[8];
        uint8_t* db[8];
        for (int k = 0; k < co->ncols; ++k) dv[k] = co->cols[k].values, db[k] = co->cols[k].validity;
        const ah_status fs = ah_filter_apply_multi(ctx, nseg, seg_p, seg_c, seg_lo, seg_hi, seg_base, co->ncols, dv, db,
                                                   (unsigned long long*)co->acc);
        nseg = 0;
        if (fs == AH_OK) {
          enq = true;
          co->buffered = virt;  // the bookkeeping only ever covers rows whose scatter was enqueued
        } else {
          co->failed = true;  // segments of this window were planned against `virt`; nothing may be appended after them
        }
        return fs;
      };
      for (int j = 0; j < m && st == AH_OK; ++j) {
        const int i = base + j;
        const ah_array_view* cols_i = columns + (size_t)i * co->ncols;
        const int64_t selected = ah_filter_predicate_count(preds[j]);
        (void)begin_input(ctx, co);  // (fused path = no view columns: only the sequence number advances)
        if (selected == 0) continue;
        if (selected == num_rows[i] && filters[i].length == num_rows[i]) {  // every row: plain copies (coalesce.rs:229 -> push_batch)
          st = flush();
          if (st == AH_OK) st = push_batch_impl(ctx, co, cols_i, num_rows[i], tags ? tags[i] : 0, bypassed ? &bypassed[i] : nullptr);
          virt = co->buffered;
          continue;
        }
        int64_t done = 0;
        while (done < selected && st == AH_OK) {
          if (nseg == 0) st = ensure_capacity(ctx, co);
          if (st != AH_OK) break;
          const int64_t take = std::min(co->target - virt, selected - done);
          seg_p[nseg] = preds[j], seg_c[nseg] = cols_i, seg_lo[nseg] = done, seg_hi[nseg] = done + take, seg_base[nseg] = virt;
          ++nseg;
          virt += take;
          done += take;
          if (virt >= co->target) {  // the window is full: run its segments, hand the batch over
            st = flush();
            if (st == AH_OK) st = finish_buffered(ctx, co);
            virt = co->buffered;
          } else if (nseg == 8) {
            st = flush();
          }
        }
      }
      if (st == AH_OK) st = flush();
      if (st != AH_OK && enq) co->failed = true;  // rows were enqueued that the bookkeeping may not cover
    } else {
      for (int j = 0; j < m && st == AH_OK; ++j) {
        const int i = base + j;
        st = begin_input(ctx, co);
        if (st == AH_OK)
          st = push_filtered_impl(ctx, co, columns + (size_t)i * co->ncols, num_rows[i], &filters[i], preds[j], tags ? tags[i] : 0,
                                  bypassed ? &bypassed[i] : nullptr);
      }
    }
  }
  return st;
}

}  // namespace

// The same push in two halves, for a host that has the NEXT group of batches in hand while this one is appended
// (every grouped push ended in a count wait with the GPU idle behind it — 0.6-0.7 ms per 1e9 rows).
//   _begin  checks the arguments, enqueues the count passes of the n (<= 64) predicates and returns at once; the counts
//           travel to pinned words the coalescer owns (two groups may be in flight)
//   _end    waits for those counts (long there if the host called _begin of the next group first: the GPU meanwhile
//           runs the previous group's scatters) and appends the batches exactly as the one-call form does.
// Between the two the caller keeps the batches' device buffers alive; the view structs themselves are copied.
// Calls on one coalescer must be ended in the order they were begun.  An engine loop:
//     h1 = begin(group 1); for g in 2..: { h2 = begin(group g); end(h1); h1 = h2; }  end(h1);
struct ah_coalescer_push {
  SlabPush* slab = nullptr;  // the push travels as device tables (slab push): nothing else below is filled
  int n = 0, slot = -1;
  bool counted = false;  // the predicates were built synchronously in _begin (shapes the multi-count pass does not take)
  uint64_t seq = 0;
  std::vector<ah_array_view> columns, filters;
  std::vector<int64_t> num_rows;
  std::vector<uint64_t> tags;
  std::vector<ah_filter_predicate*> preds;
};

extern "C" ah_status ah_coalescer_push_batches_with_filters_begin(ah_context* ctx, ah_coalescer* co, int32_t n,
                                                                  const ah_array_view* columns, const int64_t* num_rows,
                                                                  const ah_array_view* filters, const uint64_t* tags,
                                                                  ah_coalescer_push** handle) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !handle || n < 0 || (n > 0 && (!columns || !num_rows || !filters))) return AH_INVALID_ARGUMENT;
  *handle = nullptr;
  hipSetDevice(ctx->device);
  for (int i = 0; i < n; ++i) {
    if (num_rows[i] < 0) return AH_INVALID_ARGUMENT;
    AH_TRY(check_filter(ctx, co, columns + (size_t)i * co->ncols, num_rows[i], &filters[i]));
  }
  if (slab_eligible(ctx, co, n, columns, num_rows, filters)) {
    SlabPush* sp = nullptr;
    const ah_status bs = slab_begin(ctx, co, n, columns, num_rows, filters, &sp);
    if (bs == AH_OK) {
      auto* hs = new ah_coalescer_push();
      hs->n = n;
      hs->slab = sp;
      *handle = hs;
      return AH_OK;
    }
    if (bs != AH_NOT_YET_IMPLEMENTED) return bs;
  }
  auto* h = new ah_coalescer_push();
  h->n = n;
  h->columns.assign(columns, columns + (size_t)n * co->ncols);
  h->filters.assign(filters, filters + n);
  h->num_rows.assign(num_rows, num_rows + n);
  if (tags) h->tags.assign(tags, tags + n);
  h->preds.assign((size_t)n, nullptr);
  ah_status st = AH_OK;
  if (n > 64) {
    // More than 64 batches and the slab path declined — for a reason that can be transient or a property of the context
    // (deferred mode, an allocator hook, a capture, two pushes already in flight, AH_COALESCE_SLAB=0, a generic column;
    // ADVICE r05): counted here, 64 predicates per count pass and one wait each, and appended by _end exactly as the one-call
    // form appends them.  (Before round 6 this was an AH_INVALID_ARGUMENT.)
    for (int base = 0; base < n && st == AH_OK; base += 64)
      st = ah_filter_predicates_build(ctx, std::min(64, n - base), h->filters.data() + base, h->preds.data() + base);
    h->counted = st == AH_OK;
    if (st != AH_OK) {
      (void)ah_stream_wait(ctx);
      for (auto*& p : h->preds) ah_filter_predicate_free(ctx, p), p = nullptr;
    }
  } else if (n > 0) {
    const int slot = !co->cnt_busy[0] ? 0 : (!co->cnt_busy[1] ? 1 : -1);
    st = slot < 0 ? AH_NOT_YET_IMPLEMENTED
                  : ah_filter_predicates_begin(ctx, n, h->filters.data(), h->preds.data(), co->cnt_pin_dev + 64 * slot, &h->seq,
                                               co->quant_pin_dev + (size_t)64 * 32 * slot);
    if (st == AH_OK) {
      h->slot = slot;
      co->cnt_busy[slot] = true;
    } else if (st == AH_NOT_YET_IMPLEMENTED) {  // both slots in flight, or a predicate the multi-count pass does not take
      st = ah_filter_predicates_build(ctx, n, h->filters.data(), h->preds.data());
      h->counted = st == AH_OK;
    }
  }
  if (st != AH_OK) {
    delete h;
    return st;
  }
  *handle = h;
  return AH_OK;
}

extern "C" ah_status ah_coalescer_push_batches_with_filters_end(ah_context* ctx, ah_coalescer* co, ah_coalescer_push* h,
                                                                int32_t* bypassed) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !h) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  ah_status st = AH_OK;
  InputGuard input_guard(co, h->n);
  for (int i = 0; i < h->n && bypassed; ++i) bypassed[i] = 0;
  if (h->slab) {
    st = slab_end(ctx, co, h->slab);
    delete h;
    return st;
  }
  if (h->slot >= 0) {
    st = ah_filter_predicates_end(ctx, h->n, h->preds.data(), co->cnt_pin + 64 * h->slot, h->seq, co->quant_pin + (size_t)64 * 32 * h->slot);
    co->cnt_busy[h->slot] = false;
    if (st != AH_OK) (void)ah_stream_wait(ctx);  // the count kernels may still be writing the blocks freed below
  }
  if (st == AH_OK && co->failed) st = ah_fail(ctx, AH_INVALID_ARGUMENT, "BatchCoalescer: unusable after an earlier device error");
  if (st == AH_OK && h->n > 0) {
    const bool fusable = group_fusable(co, h->n, h->columns.data(), h->filters.data());
    for (int base = 0; base < h->n && st == AH_OK; base += 64)  // (groups of 64, like the one-call form)
      st = append_group(ctx, co, std::min(64, h->n - base), h->columns.data() + (size_t)base * co->ncols, h->num_rows.data() + base,
                        h->filters.data() + base, h->tags.empty() ? nullptr : h->tags.data() + base, bypassed ? bypassed + base : nullptr,
                        h->preds.data() + base, fusable);
  }
  for (auto* p : h->preds) ah_filter_predicate_free(ctx, p);
  delete h;
  return st;
}

// A begun push the host will not end (an error between begin and end): waits for its count kernels, gives its pinned slot
// and predicate blocks back.  The batches of the group are NOT appended; their sequence numbers are still consumed.
extern "C" void ah_coalescer_push_abort(ah_context* ctx, ah_coalescer* co, ah_coalescer_push* h) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !h) return;
  hipSetDevice(ctx->device);
  InputGuard input_guard(co, h->n);
  if (h->slab) {
    slab_abort(ctx, co, h->slab);
  } else {
    (void)ah_stream_wait(ctx);
    if (h->slot >= 0) co->cnt_busy[h->slot] = false;
    for (auto* p : h->preds) ah_filter_predicate_free(ctx, p);
  }
  delete h;
}

// push_batch_with_indices (coalesce.rs:289-297): `take_record_batch(&batch, indices)` then push_batch of the result.
// The taken columns are owned here: generic columns adopt them, fixed-width ones are copied into the in-progress
// buffers (the reference materialises and copies as well — its own "todo: optimize").
extern "C" ah_status ah_coalescer_push_batch_with_indices(ah_context* ctx, ah_coalescer* co, const ah_array_view* columns,
                                                          int64_t num_rows, const ah_array_view* indices) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !columns || !indices || num_rows < 0) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  InputGuard input_guard(co, 1);
  AH_TRY(check_columns(ctx, co, columns, num_rows));
  AH_TRY(begin_input(ctx, co));
  std::vector<ah_array_out> outs((size_t)co->ncols);
  for (auto& o : outs) ah_out_init(&o);
  ah_status st = AH_OK;
  for (int i = 0; i < co->ncols && st == AH_OK; ++i) st = ah_take(ctx, &columns[i], indices, 0, &outs[i]);
  const int64_t rows = indices->length;
  if (st == AH_OK && rows > 0) {
    if (co->limit >= 0 && rows > co->limit && (co->buffered == 0 || co->buffered > co->limit)) {
      // large-batch cases 1 / 2 on a batch this call owns: it leaves as an OWNED completed batch
      if (co->buffered > co->limit) st = finish_buffered(ctx, co);
      if (st == AH_OK) {
        CoBatch b;
        b.rows = rows;
        b.cols = outs;
        for (auto& o : outs) ah_out_init(&o);
        if (co->has_views) b.sources.push_back(co->cur_seq);
        co->completed.push_back(std::move(b)), co->completed_batches += 1;
      }
    } else {
      std::vector<ah_array_view> views((size_t)co->ncols);
      std::vector<std::shared_ptr<GenOwner>> owners((size_t)co->ncols);
      for (int i = 0; i < co->ncols; ++i) {
        const ah_array_out& o = outs[i];
        views[i] = slice_view(o.type, o.values, o.values_bit_offset, o.validity, o.validity_bit_offset, o.offsets, o.length,
                              o.validity ? o.null_count : 0, 0, o.length);
        if (co->cols[i].generic) {
          owners[i] = std::make_shared<GenOwner>(ctx, outs[i]);
          ah_out_init(&outs[i]);
        }
      }
      // (bypass cannot trigger again inside: limit < 0, or rows <= limit, or case 3 = coalesce normally)
      const int64_t saved_limit = co->limit;
      co->limit = -1;
      st = push_batch_impl(ctx, co, views.data(), rows, 0, nullptr, 0, owners.data());
      co->limit = saved_limit;
    }
  }
  for (auto& o : outs) ah_array_release(ctx, &o);  // fixed-width copies are stream-ordered behind the pool's reuse
  return st;
}

// View schemas: the data-buffer counts (n_columns entries per batch; ignored for non-view columns) of the next `n_batches`
// pushed batches, in push order.  The buffers themselves stay with the host.
extern "C" ah_status ah_coalescer_declare_view_buffers(ah_context* ctx, ah_coalescer* co, int32_t n_batches, const int32_t* counts) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || n_batches < 0 || (n_batches > 0 && !counts)) return AH_INVALID_ARGUMENT;
  for (int b = 0; b < n_batches; ++b) {
    std::vector<int32_t> c(counts + (size_t)b * co->ncols, counts + (size_t)(b + 1) * co->ncols);
    for (int i = 0; i < co->ncols; ++i)
      if (co->cols[i].is_view && c[(size_t)i] < 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "negative data buffer count for column %d", i);
    co->declared.push_back(std::move(c));
  }
  return AH_OK;
}

// The inputs (push sequence numbers: the first push of a coalescer's life is 0, every pushed batch counts — grouped
// pushes one per batch) whose rows make up the FRONT completed batch, in order: a view column's data buffers are the
// concatenation of those inputs' buffer lists.  *n = how many there are (may exceed `cap`: call again with more room).
extern "C" ah_status ah_coalescer_completed_batch_sources(ah_context* ctx, ah_coalescer* co, uint64_t* seqs, int32_t cap, int32_t* n) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !n || cap < 0 || (cap > 0 && !seqs)) return AH_INVALID_ARGUMENT;
  if (co->completed.empty()) {
    *n = 0;
    return AH_OK;
  }
  const std::vector<uint64_t>& s = co->completed.front().sources;
  *n = (int32_t)s.size();
  for (int32_t i = 0; i < cap && i < (int32_t)s.size(); ++i) seqs[i] = s[(size_t)i];
  return AH_OK;
}

extern "C" ah_status ah_coalescer_finish_buffered_batch(ah_context* ctx, ah_coalescer* co) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  return finish_buffered(ctx, co);
}

// next_completed_batch (coalesce.rs:566): *num_rows = -1 when there is none.  `outs` receives n_columns results the
// caller releases with ah_array_release; *tag != 0 marks a bypassed input batch (borrowed buffers).
extern "C" ah_status ah_coalescer_next_completed_batch(ah_context* ctx, ah_coalescer* co, ah_array_out* outs, int64_t* num_rows,
                                                       uint64_t* tag) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !outs || !num_rows) return AH_INVALID_ARGUMENT;
  if (tag) *tag = 0;
  if (co->completed.empty()) {
    *num_rows = -1;
    return AH_OK;
  }
  return pop_front_batch(ctx, co, outs, num_rows, tag);
}

// next_completed_batch for up to `max_batches` batches in one call (a slab push completes thousands of 8192-row batches at
// once): outs receives *n x n_columns results, batch-major; num_rows / tags (optional) one entry per batch.
extern "C" ah_status ah_coalescer_next_completed_batches(ah_context* ctx, ah_coalescer* co, int32_t max_batches, ah_array_out* outs,
                                                         int64_t* num_rows, uint64_t* tags, int32_t* n) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !co || !outs || !num_rows || !n || max_batches < 0) return AH_INVALID_ARGUMENT;
  *n = 0;
  while (*n < max_batches && !co->completed.empty()) {
    AH_TRY(pop_front_batch(ctx, co, outs + (size_t)*n * co->ncols, &num_rows[*n], tags ? &tags[*n] : nullptr));
    *n += 1;
  }
  return AH_OK;
}

// ah_array_release for `n` results in one call
extern "C" void ah_arrays_release(ah_context* ctx, ah_array_out* outs, int64_t n) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !outs) return;
  for (int64_t i = 0; i < n; ++i) ah_array_release(ctx, &outs[i]);
}
