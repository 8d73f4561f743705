// temporal_cast.hpp — the arithmetic and the planner of the temporal casts (cast_temporal.hip).
//
// Reference: arrow-cast/src/cast/mod.rs:1700-2260 (the "temporal casts" match arms), timestamp_to_date32 :633-659,
// as_time_res_with_timezone :615-631, adjust_timestamp_to_timezone :2629-2649, and the conversions they call in
// arrow-array/src/temporal_conversions.rs:141-216 (split_second = div_euclid / rem_euclid; DateTime::from_timestamp
// of chrono 0.4.45, Cargo.lock:854 — a third-party crate absent from /root/reference: days = secs.div_euclid(86400)
// must name a NaiveDate in MIN..=MAX = -262143-01-01..=+262142-12-31).
//
// Plain C++ on purpose (like parse_num.hpp): tests/cpp/temporal_cast_host_test.cpp compiles this header for the
// HOST and runs every (from, to) pair row by row against the oracle and an independent Python model before the
// same source reaches a GPU.  Nothing here touches memory: `tc_row` is the closure the reference hands to
// unary / unary_opt / try_unary, `tc_make_plan` is its `match (from_type, to_type)`.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/arrow_hip.h"

#ifdef __HIPCC__
#define TC_FN __host__ __device__ __forceinline__
#else
#define TC_FN static inline
#endif

namespace tc {

enum TOp : int {
  T_MUL_WRAP = 0,     // unary:     x as i64 * k, wrapping                      (e.g. mod.rs:1762-1766)
  T_MUL_CHECKED = 1,  // unary_opt / try_unary: checked_mul(k) in the OUTPUT width (mod.rs:1783-1793, :1899-1905)
  T_DIV = 2,          // unary:     (x / k) as O, truncating                    (mod.rs:1805-1851)
  T_DIV_TRY_I32 = 3,  // unary_opt / try_unary: i32::try_from(x / k)            (mod.rs:1767-1781)
  T_TS_DATE32 = 4,    // try_unary: as_datetime(x).date() in the source zone    (mod.rs:633-659)
  T_TS_TIME = 5,      // try_unary: as_datetime(x).time() in the source zone    (mod.rs:1974-2165)
  T_TZ_ADJUST = 6     // unary_opt / try_unary: wall clock kept, zone attached  (mod.rs:2629-2649)
};

// chrono NaiveDate::MIN / MAX as days since 1970-01-01 (MIN_YEAR = -262143, MAX_YEAR = 262142)
constexpr int64_t kMinDay = -96465292;
constexpr int64_t kMaxDay = 95026236;

struct TParams {
  int op;
  int64_t k;     // factor / divisor
  int64_t mult;  // source units per second (T_TS_*, T_TZ_ADJUST)
  int64_t off;   // zone offset in seconds
  int64_t tmul;  // T_TS_TIME: out = second_of_day * tmul + nanos / ndiv
  int64_t ndiv;
  int64_t sub_to_ns;  // nanoseconds per source tick = 1e9 / mult
  int64_t lo, hi;     // T_TS_*: the ticks of chrono's first and last representable second (unused for nanoseconds)
  int64_t ratio;      // T_TS_TIME: target ticks per source tick (tmul >= mult) or source ticks per target tick
};

// Every divisor the arms use is one of a handful of constants, so the divisions are written against COMPILE-TIME
// divisors (the compiler turns them into a multiply-high and shifts) behind a switch on the kernel argument — a
// uniform scalar branch.  A 64-bit division by a run-time value is a ~60-instruction software routine on gfx950 and
// made the first version of this kernel issue-bound at 27-42 % of the HBM peak.
template <int64_t D>
TC_FN int64_t tc_fdiv_c(int64_t a) {  // floor(a / D), D > 0  (i64::div_euclid)
  int64_t q = a / D;
  return (a - q * D < 0) ? q - 1 : q;
}
TC_FN int64_t tc_div_trunc(int64_t a, int64_t d) {  // Rust `/`
  switch (d) {
    case 1: return a;
    case 1000: return a / 1000;
    case 1000000: return a / 1000000;
    case 1000000000: return a / 1000000000;
    case 86400000: return a / 86400000;
    default: return a / d;
  }
}
// split_second (temporal_conversions.rs:213-216): whole seconds (div_euclid) and the remaining ticks (rem_euclid)
TC_FN void tc_split_second(int64_t x, int64_t mult, int64_t* sec, int64_t* sub) {
  switch (mult) {
    case 1: *sec = x; break;
    case 1000: *sec = tc_fdiv_c<1000>(x); break;
    case 1000000: *sec = tc_fdiv_c<1000000>(x); break;
    default: *sec = tc_fdiv_c<1000000000>(x); break;
  }
  *sub = x - *sec * mult;
}
// DateTime::from_timestamp accepts the second iff its day is a NaiveDate: one range test on the seconds
constexpr int64_t kMinSec = kMinDay * 86400, kMaxSec = kMaxDay * 86400 + 86399;

template <typename O>
TC_FN O tc_wrap_to(int64_t v) {
  return (O)(typename std::make_unsigned<O>::type)(uint64_t)v;  // Rust `as`: keep the low bits
}

// One arm's closure.  OP is a template parameter so that each launch carries the code of its own arm only.
template <typename I, typename O, int OP>
TC_FN bool tc_row_op(const TParams& a, I v, O* o) {
  const int64_t x = (int64_t)v;
  if constexpr (OP == T_MUL_WRAP) {
    *o = tc_wrap_to<O>((int64_t)((uint64_t)x * (uint64_t)a.k));
    return true;
  } else if constexpr (OP == T_MUL_CHECKED) {
    long long p;
    if (__builtin_mul_overflow((long long)x, (long long)a.k, &p)) return false;
    if (sizeof(O) == 4 && (p < INT32_MIN || p > INT32_MAX)) return false;
    *o = (O)p;
    return true;
  } else if constexpr (OP == T_DIV) {
    *o = tc_wrap_to<O>(tc_div_trunc(x, a.k));
    return true;
  } else if constexpr (OP == T_DIV_TRY_I32) {
    int64_t q = tc_div_trunc(x, a.k);
    if (q < INT32_MIN || q > INT32_MAX) return false;
    *o = (O)q;
    return true;
  } else if constexpr (OP == T_TS_DATE32 || OP == T_TS_TIME) {
    // as_datetime(x) in the source zone.  seconds = div_euclid(x, mult), day = div_euclid(seconds + off, 86400) folds
    // into ONE floor division of the shifted tick count by the ticks of a day; the calendar range test is a compare
    // on x (nanoseconds never leave the calendar: +-292 years).
    if (a.mult != 1000000000 && (x < a.lo || x > a.hi)) return false;
    long long xs;
    int64_t day, tod;
    if (__builtin_add_overflow((long long)x, (long long)(a.off * a.mult), &xs)) {
      // within a zone offset of the i64 ends (nanoseconds only): the two-step form cannot overflow
      int64_t sec, sub;
      tc_split_second(x, a.mult, &sec, &sub);
      day = tc_fdiv_c<86400>(sec + a.off);
      tod = (sec + a.off - day * 86400) * a.mult + sub;
    } else {
      switch (a.mult) {
        case 1: day = tc_fdiv_c<86400ll>(xs); break;
        case 1000: day = tc_fdiv_c<86400000ll>(xs); break;
        case 1000000: day = tc_fdiv_c<86400000000ll>(xs); break;
        default: day = tc_fdiv_c<86400000000000ll>(xs); break;
      }
      tod = xs - day * (86400 * a.mult);
    }
    if constexpr (OP == T_TS_DATE32) {
      *o = (O)day;
    } else {
      // time_to_time32s .. time_to_time64ns of (second of day, nanosecond) == the ticks since local midnight rescaled
      *o = tc_wrap_to<O>(a.tmul >= a.mult ? tod * a.ratio : tc_div_trunc(tod, a.ratio));
    }
    return true;
  } else {  // T_TZ_ADJUST
    int64_t sec, sub;
    tc_split_second(x, a.mult, &sec, &sub);
    if (sec < kMinSec || sec > kMaxSec) return false;
    int64_t shifted = sec - a.off;  // `local - offset` must stay a NaiveDateTime
    if (shifted < kMinSec || shifted > kMaxSec) return false;
    long long r;
    if (__builtin_sub_overflow((long long)x, (long long)(a.off * a.mult), &r)) return false;
    *o = (O)r;
    return true;
  }
}

// run-time form (the host harness; the kernels are instantiated per arm)
template <typename I, typename O>
TC_FN bool tc_row(const TParams& a, I v, O* o) {
  switch (a.op) {
    case T_MUL_WRAP: return tc_row_op<I, O, T_MUL_WRAP>(a, v, o);
    case T_MUL_CHECKED: return tc_row_op<I, O, T_MUL_CHECKED>(a, v, o);
    case T_DIV: return tc_row_op<I, O, T_DIV>(a, v, o);
    case T_DIV_TRY_I32: return tc_row_op<I, O, T_DIV_TRY_I32>(a, v, o);
    case T_TS_DATE32: return tc_row_op<I, O, T_TS_DATE32>(a, v, o);
    case T_TS_TIME: return tc_row_op<I, O, T_TS_TIME>(a, v, o);
    default: return tc_row_op<I, O, T_TZ_ADJUST>(a, v, o);
  }
}

// ------------------------------------------------------------------ planner (host)
static inline bool is_plain_numeric(int32_t id) { return (id >= AH_INT8 && id <= AH_UINT64) || id == AH_FLOAT32 || id == AH_FLOAT64; }
static inline bool is_temporal(int32_t id) { return id >= AH_DT_DATE32 && id <= AH_DT_DURATION; }

static inline ah_type physical_of(const ah_data_type& t) {
  switch (t.id) {
    case AH_DT_DATE32:
    case AH_DT_TIME32: return AH_INT32;
    case AH_DT_DATE64:
    case AH_DT_TIME64:
    case AH_DT_TIMESTAMP:
    case AH_DT_DURATION: return AH_INT64;
    default: return (ah_type)t.id;
  }
}

static inline bool unit_ok(const ah_data_type& t) {
  if (t.unit < AH_SECOND || t.unit > AH_NANOSECOND) return false;
  if (t.id == AH_DT_TIME32) return t.unit == AH_SECOND || t.unit == AH_MILLISECOND;
  if (t.id == AH_DT_TIME64) return t.unit == AH_MICROSECOND || t.unit == AH_NANOSECOND;
  return true;
}

static inline const char* unit_text(int u) {
  static const char* n[] = {"s", "ms", "\xC2\xB5s", "ns"};  // TimeUnit's Display (arrow-schema/src/datatype.rs:447)
  return (u >= 0 && u < 4) ? n[u] : "?";
}
static inline const char* unit_type_name(int u) {
  static const char* n[] = {"Second", "Millisecond", "Microsecond", "Nanosecond"};
  return (u >= 0 && u < 4) ? n[u] : "?";
}

static inline const char* plain_type_name(int32_t id) {
  switch (id) {
    case AH_BOOL: return "Boolean";
    case AH_INT8: return "Int8";
    case AH_INT16: return "Int16";
    case AH_INT32: return "Int32";
    case AH_INT64: return "Int64";
    case AH_UINT8: return "UInt8";
    case AH_UINT16: return "UInt16";
    case AH_UINT32: return "UInt32";
    case AH_UINT64: return "UInt64";
    case AH_FLOAT16: return "Float16";
    case AH_FLOAT32: return "Float32";
    case AH_FLOAT64: return "Float64";
    case AH_UTF8: return "Utf8";
    case AH_LARGE_UTF8: return "LargeUtf8";
    case AH_UTF8_VIEW: return "Utf8View";
    case AH_BINARY_VIEW: return "BinaryView";
    default: return "?";
  }
}

// DataType's Display (arrow-schema/src/datatype_display.rs:45-60); a zone is rendered from its offset
static inline std::string type_text(const ah_data_type& t) {
  char b[96];
  switch (t.id) {
    case AH_DT_DATE32: return "Date32";
    case AH_DT_DATE64: return "Date64";
    case AH_DT_TIME32: snprintf(b, sizeof b, "Time32(%s)", unit_text(t.unit)); return b;
    case AH_DT_TIME64: snprintf(b, sizeof b, "Time64(%s)", unit_text(t.unit)); return b;
    case AH_DT_DURATION: snprintf(b, sizeof b, "Duration(%s)", unit_text(t.unit)); return b;
    case AH_DT_TIMESTAMP:
      if (t.has_tz) {
        int o = t.tz_offset_seconds, ao = o < 0 ? -o : o;
        snprintf(b, sizeof b, "Timestamp(%s, \"%c%02d:%02d\")", unit_text(t.unit), o < 0 ? '-' : '+', ao / 3600, ao / 60 % 60);
      } else {
        snprintf(b, sizeof b, "Timestamp(%s)", unit_text(t.unit));
      }
      return b;
    default: return plain_type_name(t.id);
  }
}

constexpr int64_t kUnitsPerSecond[4] = {1, 1000, 1000000, 1000000000};

struct Step {
  enum Kind { NUMERIC, KERNEL } kind;
  ah_type to_phys;  // NUMERIC: ah_cast target; KERNEL: output width
  TParams a;        // KERNEL: the closure's constants
  enum Mode { UNARY, OPT_OR_TRY, TRY_ONLY } mode;
  ah_status err_status;
  std::string err_fmt;  // one %s = the failing source value
};

static inline Step numeric_step(ah_type to) {
  Step s{};
  s.kind = Step::NUMERIC;
  s.to_phys = to;
  return s;
}
static inline Step kernel_step(ah_type to, int op, Step::Mode mode, int64_t k) {
  Step s{};
  s.kind = Step::KERNEL;
  s.to_phys = to;
  s.a.op = op;
  s.a.k = k;
  s.a.mult = 1;
  s.a.ndiv = 1;
  s.a.sub_to_ns = 1000000000;
  s.mode = mode;
  s.err_status = AH_CAST_ERROR;
  return s;
}
static inline void set_calendar_bounds(TParams* p) {
  if (p->mult == 1000000000) return;  // i64 nanoseconds cannot leave the calendar
  p->lo = kMinSec * p->mult;
  p->hi = kMaxSec * p->mult + (p->mult - 1);
}
static inline Step mul_wrap(ah_type to, int64_t k) { return kernel_step(to, T_MUL_WRAP, Step::UNARY, k); }
static inline Step div_trunc(ah_type to, int64_t k) { return kernel_step(to, T_DIV, Step::UNARY, k); }
static inline Step mul_checked(ah_type to, int64_t k) {
  Step s = kernel_step(to, T_MUL_CHECKED, Step::OPT_OR_TRY, k);
  s.err_status = AH_ARITHMETIC_OVERFLOW;
  s.err_fmt = "Overflow happened on: %s * " + std::to_string(k);
  return s;
}
static inline Step tz_adjust(int unit, int off) {
  Step s = kernel_step(AH_INT64, T_TZ_ADJUST, Step::OPT_OR_TRY, 0);
  s.a.mult = kUnitsPerSecond[unit];
  s.a.off = off;
  s.err_fmt = "Cannot cast timezone to different timezone";
  return s;
}

// unit conversion shared by Timestamp -> Timestamp and Duration -> Duration (mod.rs:1880-1905, :2257-2279)
static inline void convert_units(int from_unit, int to_unit, std::vector<Step>* plan) {
  int64_t f = kUnitsPerSecond[from_unit], t = kUnitsPerSecond[to_unit];
  if (f > t) plan->push_back(div_trunc(AH_INT64, f / t));
  else if (f < t) plan->push_back(mul_checked(AH_INT64, t / f));
  else plan->push_back(numeric_step(AH_INT64));  // `time_array.clone()`
}

// The reference's match arms, in its order.  Returns false for "Casting from {from} to {to} not supported".
static inline bool make_plan(const ah_data_type& f, const ah_data_type& t, std::vector<Step>* plan) {
  const int32_t F = f.id, T = t.id;
  auto needs_unit = [](int32_t id) { return id == AH_DT_TIME32 || id == AH_DT_TIME64 || id == AH_DT_TIMESTAMP || id == AH_DT_DURATION; };
  if ((needs_unit(F) && !unit_ok(f)) || (needs_unit(T) && !unit_ok(t))) return false;  // e.g. Time32(µs): no such Rust type
  auto same = [&] {
    if (F != T) return false;
    if (F == AH_DT_DATE32 || F == AH_DT_DATE64) return true;
    if (f.unit != t.unit) return false;
    if (F == AH_DT_TIMESTAMP) return f.has_tz == t.has_tz && (!f.has_tz || f.tz_offset_seconds == t.tz_offset_seconds);
    return true;
  };
  if (same()) {  // mod.rs:797-799
    plan->push_back(numeric_step(physical_of(t)));
    return true;
  }
  const int64_t MS_DAY = 86400000ll;
  // ---- Int32 / Int64 <-> Date / Time (mod.rs:1701-1761)
  if (F == AH_INT32 && (T == AH_DT_DATE32 || T == AH_DT_TIME32)) return plan->push_back(numeric_step(AH_INT32)), true;
  if (F == AH_INT32 && T == AH_DT_DATE64) return plan->push_back(mul_wrap(AH_INT64, MS_DAY)), true;
  if ((F == AH_DT_DATE32 || F == AH_DT_TIME32) && T == AH_INT32) return plan->push_back(numeric_step(AH_INT32)), true;
  if ((F == AH_DT_DATE32 || F == AH_DT_TIME32) && T == AH_INT64) return plan->push_back(numeric_step(AH_INT64)), true;
  if (F == AH_INT64 && (T == AH_DT_DATE64 || T == AH_DT_TIME64)) return plan->push_back(numeric_step(AH_INT64)), true;
  if (F == AH_INT64 && T == AH_DT_DATE32) return plan->push_back(numeric_step(AH_INT32)), true;
  if ((F == AH_DT_DATE64 || F == AH_DT_TIME64) && T == AH_INT64) return plan->push_back(numeric_step(AH_INT64)), true;
  if (F == AH_DT_DATE64 && T == AH_INT32) return plan->push_back(numeric_step(AH_INT32)), true;
  // ---- Date <-> Date (mod.rs:1762-1781)
  if (F == AH_DT_DATE32 && T == AH_DT_DATE64) return plan->push_back(mul_wrap(AH_INT64, MS_DAY)), true;
  if (F == AH_DT_DATE64 && T == AH_DT_DATE32) {
    Step s = kernel_step(AH_INT32, T_DIV_TRY_I32, Step::OPT_OR_TRY, MS_DAY);
    s.err_fmt = "Cannot cast Date64 value %s to Date32 without overflow";
    plan->push_back(s);
    return true;
  }
  // ---- Time <-> Time (mod.rs:1783-1851)
  if ((F == AH_DT_TIME32 || F == AH_DT_TIME64) && (T == AH_DT_TIME32 || T == AH_DT_TIME64)) {
    const ah_type to = physical_of(t);
    int64_t fu = kUnitsPerSecond[f.unit], tu = kUnitsPerSecond[t.unit];
    if (fu > tu) plan->push_back(div_trunc(to, fu / tu));
    else if (F == AH_DT_TIME32 && T == AH_DT_TIME32) plan->push_back(mul_checked(AH_INT32, tu / fu));  // s -> ms
    else plan->push_back(mul_wrap(to, tu / fu));
    return true;
  }
  // ---- Timestamp / Duration <-> numbers (mod.rs:1854-1878, :2236-2255)
  if ((F == AH_DT_TIMESTAMP || F == AH_DT_DURATION) && is_plain_numeric(T)) return plan->push_back(numeric_step((ah_type)T)), true;
  if (is_plain_numeric(F) && (T == AH_DT_TIMESTAMP || T == AH_DT_DURATION)) return plan->push_back(numeric_step(AH_INT64)), true;
  // ---- Timestamp -> Timestamp (mod.rs:1880-1937)
  if (F == AH_DT_TIMESTAMP && T == AH_DT_TIMESTAMP) {
    const bool adjust = !f.has_tz && t.has_tz;
    if (!(adjust && f.unit == t.unit)) convert_units(f.unit, t.unit, plan);  // a clone before the adjust pass is dropped
    if (adjust) plan->push_back(tz_adjust(t.unit, t.tz_offset_seconds));
    return true;
  }
  if (F == AH_DT_TIMESTAMP && T == AH_DT_DATE32) {  // timestamp_to_date32, mod.rs:633-659
    Step s = kernel_step(AH_INT32, T_TS_DATE32, Step::TRY_ONLY, 0);
    s.a.mult = kUnitsPerSecond[f.unit];
    s.a.off = f.has_tz ? f.tz_offset_seconds : 0;
    set_calendar_bounds(&s.a);
    s.err_fmt = std::string("Cannot convert arrow_array::types::Timestamp") + unit_type_name(f.unit) + "Type %s to datetime";
    plan->push_back(s);
    return true;
  }
  if (F == AH_DT_TIMESTAMP && T == AH_DT_DATE64) {  // mod.rs:1950-1973 (the zone is not consulted)
    switch (f.unit) {
      case AH_SECOND: plan->push_back(mul_checked(AH_INT64, 1000)); break;
      case AH_MILLISECOND: plan->push_back(numeric_step(AH_INT64)); break;
      case AH_MICROSECOND: plan->push_back(div_trunc(AH_INT64, 1000)); break;
      default: plan->push_back(div_trunc(AH_INT64, 1000000)); break;
    }
    return true;
  }
  if (F == AH_DT_TIMESTAMP && (T == AH_DT_TIME32 || T == AH_DT_TIME64)) {  // mod.rs:1974-2165
    Step s = kernel_step(physical_of(t), T_TS_TIME, Step::TRY_ONLY, 0);
    s.a.mult = kUnitsPerSecond[f.unit];
    s.a.off = f.has_tz ? f.tz_offset_seconds : 0;
    s.a.tmul = kUnitsPerSecond[t.unit];
    s.a.ndiv = 1000000000ll / kUnitsPerSecond[t.unit];  // time_to_time32s..64ns (temporal_conversions.rs:113-139)
    s.a.sub_to_ns = 1000000000ll / kUnitsPerSecond[f.unit];
    s.a.ratio = s.a.tmul >= s.a.mult ? s.a.tmul / s.a.mult : s.a.mult / s.a.tmul;
    set_calendar_bounds(&s.a);
    s.err_fmt = std::string("Failed to create naive time with arrow_array::types::Timestamp") + unit_type_name(f.unit) + "Type %s";
    plan->push_back(s);
    return true;
  }
  // ---- Date -> Timestamp (mod.rs:2166-2234), then the Timestamp -> Timestamp arm attaches the zone
  if ((F == AH_DT_DATE64 || F == AH_DT_DATE32) && T == AH_DT_TIMESTAMP) {
    if (F == AH_DT_DATE64) {
      switch (t.unit) {
        case AH_SECOND: plan->push_back(div_trunc(AH_INT64, 1000)); break;
        case AH_MILLISECOND:
          if (!t.has_tz) plan->push_back(numeric_step(AH_INT64));  // reinterpret; with a zone the adjust pass is the copy
          break;
        case AH_MICROSECOND: plan->push_back(mul_wrap(AH_INT64, 1000)); break;
        default: plan->push_back(mul_wrap(AH_INT64, 1000000)); break;
      }
    } else {
      switch (t.unit) {
        case AH_SECOND: plan->push_back(mul_wrap(AH_INT64, 86400)); break;
        case AH_MILLISECOND: plan->push_back(mul_wrap(AH_INT64, MS_DAY)); break;
        case AH_MICROSECOND: plan->push_back(mul_checked(AH_INT64, MS_DAY * 1000)); break;
        default: plan->push_back(mul_checked(AH_INT64, MS_DAY * 1000000)); break;
      }
    }
    if (t.has_tz) plan->push_back(tz_adjust(t.unit, t.tz_offset_seconds));
    return true;
  }
  if (F == AH_DT_DURATION && T == AH_DT_DURATION) {  // mod.rs:2257-2279
    convert_units(f.unit, t.unit, plan);
    return true;
  }
  return false;
}

}  // namespace tc
