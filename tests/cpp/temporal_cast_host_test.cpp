// Host harness for arrow-rs_amd/csrc/temporal_cast.hpp — the SAME header cast_temporal.hip compiles for gfx950.
// It runs the product's planner (tc::make_plan) and row closure (tc::tc_row) on the CPU over cases read from
// stdin, with the array-level rules of unary / unary_opt / try_unary applied exactly as tcast_kernel and
// run_kernel_step apply them, so tests/test_temporal_cast_cpu.py can compare every (from, to) pair with the
// oracle and with an independent Python model without a GPU.  Test infrastructure only.
//
// stdin, one case per block:
//   case <f.id> <f.unit> <f.has_tz> <f.off> <t.id> <t.unit> <t.has_tz> <t.off> <safe> <has_validity> <n>
//   n lines: <value> <valid>
// stdout per case:
//   unsupported <message>            | error <status> <message>
//   ok <out_physical> <has_validity> <n>   then n lines: <value> <valid>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../arrow-rs_amd/csrc/temporal_cast.hpp"

using namespace tc;

struct Col {
  ah_type type = AH_INT64;
  std::vector<int64_t> v;
  std::vector<char> valid;
  bool has_validity = false;
};

struct Err {
  int status = 0;
  std::string msg;
};

static bool in_range(ah_type t, int64_t v) { return t != AH_INT32 || (v >= INT32_MIN && v <= INT32_MAX); }

// ah_cast between Int32 and Int64 (cast.hip: safe = unary_opt, unsafe = try_unary; same type = clone)
static bool numeric_step(const Col& in, ah_type to, bool safe, Col* out, Err* err) {
  out->type = to;
  const size_t n = in.v.size();
  out->v.assign(n, 0);
  if (in.type == to) {
    *out = in;
    return true;
  }
  if (to != AH_INT32 && to != AH_INT64) {
    err->status = -1;
    err->msg = "harness: non-integer target";
    return false;
  }
  out->has_validity = safe || in.has_validity;
  out->valid.assign(n, 1);
  for (size_t i = 0; i < n; ++i) {
    bool valid = !in.has_validity || in.valid[i];
    out->valid[i] = valid;
    if (!valid) continue;
    if (in_range(to, in.v[i])) {
      out->v[i] = in.v[i];
    } else if (safe) {
      out->valid[i] = 0;
    } else {
      err->status = AH_CAST_ERROR;
      err->msg = "Can't cast value " + std::to_string(in.v[i]) + " to type Int32";
      return false;
    }
  }
  return true;
}

template <typename I, typename O>
static bool row(const TParams& p, int64_t v, int64_t* o) {
  O r{};
  bool ok = tc_row<I, O>(p, (I)v, &r);
  *o = ok ? (int64_t)r : 0;
  return ok;
}

static bool kernel_step(const Col& in, const Step& s, bool safe, Col* out, Err* err) {
  const size_t n = in.v.size();
  out->type = s.to_phys;
  out->v.assign(n, 0);
  const bool fail_is_null = s.mode == Step::OPT_OR_TRY && safe;
  out->has_validity = fail_is_null || in.has_validity;
  out->valid.assign(n, 1);
  for (size_t i = 0; i < n; ++i) {
    bool valid = !in.has_validity || in.valid[i];
    out->valid[i] = valid;
    if (!valid && s.mode != Step::UNARY) continue;
    int64_t o = 0;
    bool ok;
    if (in.type == AH_INT32 && s.to_phys == AH_INT32) ok = row<int32_t, int32_t>(s.a, in.v[i], &o);
    else if (in.type == AH_INT32) ok = row<int32_t, int64_t>(s.a, in.v[i], &o);
    else if (s.to_phys == AH_INT32) ok = row<int64_t, int32_t>(s.a, in.v[i], &o);
    else ok = row<int64_t, int64_t>(s.a, in.v[i], &o);
    out->v[i] = o;
    if (ok) continue;
    if (s.mode == Step::UNARY) abort();  // unary closures cannot fail
    if (fail_is_null) {
      out->valid[i] = 0;
      continue;
    }
    char buf[256];
    snprintf(buf, sizeof buf, s.err_fmt.c_str(), std::to_string(in.v[i]).c_str());
    err->status = s.err_status;
    err->msg = buf;
    return false;
  }
  return true;
}

int main() {
  char word[16];
  while (scanf("%15s", word) == 1) {
    ah_data_type f{}, t{};
    int safe, has_validity;
    long n;
    if (scanf("%d %d %d %d %d %d %d %d %d %d %ld", &f.id, &f.unit, &f.has_tz, &f.tz_offset_seconds, &t.id, &t.unit,
              &t.has_tz, &t.tz_offset_seconds, &safe, &has_validity, &n) != 11)
      return 2;
    Col cur;
    cur.type = physical_of(f);
    cur.has_validity = has_validity != 0;
    cur.v.resize(n);
    cur.valid.resize(n);
    for (long i = 0; i < n; ++i) {
      long long v;
      int ok;
      if (scanf("%lld %d", &v, &ok) != 2) return 2;
      cur.v[i] = v;
      cur.valid[i] = (char)ok;
    }
    std::vector<Step> plan;
    if (!make_plan(f, t, &plan)) {
      printf("unsupported Casting from %s to %s not supported\n", type_text(f).c_str(), type_text(t).c_str());
      continue;
    }
    Err err;
    bool ok = true;
    for (const Step& s : plan) {
      Col next;
      ok = s.kind == Step::NUMERIC ? numeric_step(cur, s.to_phys, safe != 0, &next, &err) : kernel_step(cur, s, safe != 0, &next, &err);
      if (!ok) break;
      cur = next;
    }
    if (!ok) {
      printf("error %d %s\n", err.status, err.msg.c_str());
      continue;
    }
    printf("ok %d %d %ld\n", (int)cur.type, cur.has_validity ? 1 : 0, n);
    for (long i = 0; i < n; ++i) printf("%" PRId64 " %d\n", cur.v[i], cur.has_validity ? (int)cur.valid[i] : 1);
  }
  return 0;
}
