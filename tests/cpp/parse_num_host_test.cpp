// Host-side check of arrow-rs_amd/csrc/parse_num.hpp (the text -> number algorithms the cast_parse kernels run):
// the Eisel-Lemire conversion and the big-integer slow path against glibc strtod / strtof (both correctly
// rounded), the integer parser against strtoll-style arithmetic in __int128, and the grammar on the reference's
// own literals (arrow-cast/src/parse.rs:2882-2955).  Built and run by tests/test_oracle_golden.py with g++; prints
// "ok <cases>" or the first mismatch.  No GPU involved: this runs the product's algorithm header, compiled for
// the host, before the same code is trusted on the device.
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../arrow-rs_amd/csrc/parse_num.hpp"

struct Str {
  const uint8_t* p;
  uint8_t operator[](int64_t i) const { return p[i]; }
};

template <typename F>
static bool parse_float(const std::string& t, F* out, bool* used_slow) {
  Str s{(const uint8_t*)t.data()};
  PnDec d;
  int64_t b, e;
  *used_slow = false;
  if (!pn_scan_float(s, (int64_t)t.size(), &d, &b, &e)) return false;
  bool slow;
  auto bits = pn_convert<F>(d, &slow);
  if (slow) {
    *used_slow = true;
    static PnBig lhs, rhs;
    const PnBin lo = pn_compute_float<F>(d.q, d.w);
    if (pn_slow_round_up<F>(s, b, e, lo, &lhs, &rhs)) bits = pn_next_up_magnitude<F>(bits);
  }
  memcpy(out, &bits, sizeof(F));
  return true;
}

static long long g_cases = 0, g_slow = 0;

template <typename F>
static bool check_value(const std::string& t) {
  F got;
  bool slow;
  ++g_cases;
  if (!parse_float<F>(t, &got, &slow)) {
    printf("FAIL rejected a valid number: '%s'\n", t.c_str());
    return false;
  }
  g_slow += slow;
  F want = sizeof(F) == 8 ? (F)strtod(t.c_str(), nullptr) : (F)strtof(t.c_str(), nullptr);
  if (memcmp(&got, &want, sizeof(F)) != 0) {
    printf("FAIL %s '%s': got %a want %a (slow=%d)\n", sizeof(F) == 8 ? "f64" : "f32", t.c_str(), (double)got, (double)want, (int)slow);
    return false;
  }
  return true;
}

static bool check_both(const std::string& t) { return check_value<double>(t) && check_value<float>(t); }

template <typename T>
static bool check_int(const std::string& t) {
  // the reference's rule restated with wide arithmetic: trim, [+-]?digits+, value inside T
  ++g_cases;
  size_t b = 0, e = t.size();
  auto sp = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\f' || c == '\r'; };
  while (e > b && sp(t[e - 1])) --e;
  while (b < e && sp(t[b])) ++b;
  bool ok = e > b;
  bool neg = false;
  size_t i = b;
  if (ok && (t[i] == '-' || t[i] == '+')) {
    neg = t[i] == '-';
    ++i;
  }
  ok = ok && i < e;
  __int128 v = 0;
  for (; ok && i < e; ++i) {
    if (t[i] < '0' || t[i] > '9') ok = false;
    else if (v < ((__int128)1 << 100)) v = v * 10 + (t[i] - '0');
  }
  if (neg) v = -v;
  constexpr bool sg = (T)-1 < (T)0;
  const __int128 lo = sg ? -((__int128)1 << (sizeof(T) * 8 - 1)) : 0;
  const __int128 hi = sg ? ((__int128)1 << (sizeof(T) * 8 - 1)) - 1 : ((__int128)1 << (sizeof(T) * 8)) - 1;
  ok = ok && v >= lo && v <= hi;
  T got = 0;
  Str s{(const uint8_t*)t.data()};
  const bool gok = pn_parse_int<T>(s, (int64_t)t.size(), &got);
  if (gok != ok || (ok && (__int128)got != v)) {
    printf("FAIL int%zu%s '%s': got ok=%d %lld want ok=%d %lld\n", sizeof(T) * 8, sg ? "" : "u", t.c_str(), (int)gok, (long long)got, (int)ok,
           (long long)v);
    return false;
  }
  return true;
}

template <typename F>
static bool expect_none(const std::string& t) {
  F v;
  bool slow;
  ++g_cases;
  if (parse_float<F>(t, &v, &slow)) {
    printf("FAIL accepted '%s'\n", t.c_str());
    return false;
  }
  return true;
}

int main(int argc, char** argv) {
  const long long rounds = argc > 1 ? atoll(argv[1]) : 2000000;
  std::mt19937_64 rng(42);
  char buf[1200];

  // grammar: the reference's literals and the documented forms
  const char* valid[] = {"1.5", " 1.5", "\t\n 20.54", "\n2.5", "\n-942.5423", "\n\t\n\t\n40.5123", " -1.5", "1.5 ", "40.5123\n",
                         "40.5123\n\t\n\t\n", "-942.5423\t", " 1.5 ", "\t\n 20.54 \t", "\n-942.5423\n", "3", "4.56", "8.9", "5.", ".5",
                         "+5", "-0", "0", "-0.0", "1e5", "1E5", "1e+5", "1e-5", "1.e5", ".5e1", "00012.5000", "1e400", "1e-400", "-1e400",
                         "123456789012345678901234567890", "0.000000000000000000000000000001", "4.9e-324", "2.4703282292062327e-324",
                         "2.4703282292062328e-324", "1.7976931348623157e308", "1.7976931348623158e308", "1.7976931348623159e308",
                         "9007199254740993", "9007199254740992.5", "9007199254740993.0000000000000000000000001", "1e23", "8.5e-45",
                         "7.0064923216240853546186479164495e-46", "7.0064923216240853546186479164496e-46", "3.4028235e38", "3.4028236e38",
                         "3.40282356779733661637539395458142568448e38", "16777217", "16777217.0000000000000001", "1e22", "1e-22",
                         "0e999999999999999999999", "1e-999999999999999999999", "1e999999999999999999999",
                         "0.0000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000001e100"};
  for (const char* t : valid)
    if (!check_both(t)) return 1;
  const char* specials[] = {"nan", "NaN", "NAN", "inf", "Inf", "INF", "infinity", "Infinity", "INFINITY", "+inf", "-inf", "-nan", " inf ", "-Infinity"};
  for (const char* t : specials) {
    double v;
    bool slow;
    ++g_cases;
    if (!parse_float<double>(t, &v, &slow)) {
      printf("FAIL special '%s' rejected\n", t);
      return 1;
    }
    std::string tt(t);
    std::string low;
    for (char ch : tt) low += (char)tolower(ch);
    const bool want_nan = low.find("nan") != std::string::npos;
    const bool want_neg = tt.find('-') != std::string::npos;
    if (want_nan ? !(v != v) : !(std::isinf(v))) {
      printf("FAIL special '%s' -> %f\n", t, v);
      return 1;
    }
    if (std::signbit(v) != want_neg) {
      printf("FAIL special '%s' sign\n", t);
      return 1;
    }
  }
  const char* invalid[] = {"", "+", "-", ".", "+.", "e5", ".e5", "1e", "1e+", "1e-", "1.5abc", "40.5123x", "seven", "1 2", "1.2.3", "--1", "+-1",
                           "1e5.5", "0x10", "1_000", "in", "infinit", "infinityy", "na", "nanx", "1nan", " ", "\t\n", "1.5\x0b", "\x0b" "1.5", "1,5",
                           "1e 5", "1 e5", "+ 1", "- 1", "1f", "1d", "١"};
  for (const char* t : invalid)
    if (!expect_none<double>(t) || !expect_none<float>(t)) return 1;

  // integers: the reference's literals + edges for every width
  const char* ints[] = {"5", "6", "seven", "8", "9.1", "123", "-123", "86374", " 3", "          30", "\n \n 100", " \n25", "\t800", "\t  \n \t 851",
                        "\t\n\t\n\n\n\t1", " \n-25", "\t-800", "3 ", "30          ", "-25 \n", "800\t", " 3 ", "\n \n 100 \n", "\t-800\t\n", "30x",
                        "100px", "-25!", "3j", "", "+", "-", "+5", "-0", "-00", "+0", "00000000000000000000000000005", "-5", "127", "128", "-128", "-129",
                        "255", "256", "32767", "32768", "-32768", "-32769", "65535", "65536", "2147483647", "2147483648", "-2147483648",
                        "-2147483649", "4294967295", "4294967296", "9223372036854775807", "9223372036854775808", "-9223372036854775808",
                        "-9223372036854775809", "18446744073709551615", "18446744073709551616", "99999999999999999999999999", "1 1", "+-1",
                        "1+", "1-", " + 1", "\x0b" "1", "1\x0b", "1\f", "\f1", "1\r\n", "0x1", "1e3", "1.0", "١"};
  for (const char* t : ints) {
    if (!check_int<int8_t>(t) || !check_int<int16_t>(t) || !check_int<int32_t>(t) || !check_int<int64_t>(t)) return 1;
    if (!check_int<uint8_t>(t) || !check_int<uint16_t>(t) || !check_int<uint32_t>(t) || !check_int<uint64_t>(t)) return 1;
  }
  for (long long r = 0; r < rounds / 4; ++r) {  // random integer texts around every width's limits
    const int nd = 1 + (int)(rng() % 21);
    std::string t;
    if (rng() % 3 == 0) t += (rng() & 1) ? '-' : '+';
    for (int k = 0; k < nd; ++k) t += (char)('0' + rng() % 10);
    if (rng() % 16 == 0) t.insert(0, " ");
    if (rng() % 16 == 0) t += "\n";
    if (!check_int<int8_t>(t) || !check_int<int16_t>(t) || !check_int<int32_t>(t) || !check_int<int64_t>(t)) return 1;
    if (!check_int<uint8_t>(t) || !check_int<uint16_t>(t) || !check_int<uint32_t>(t) || !check_int<uint64_t>(t)) return 1;
  }

  // floats: (a) random bit patterns printed with 17 / 9 / shortest-ish digits, (b) random decimal texts,
  // (c) exact midpoints between adjacent floats (and their neighbours one digit away): ties and the slow path
  for (long long r = 0; r < rounds; ++r) {
    uint64_t bits = rng();
    double d;
    memcpy(&d, &bits, 8);
    if (d != d || std::isinf(d)) continue;
    snprintf(buf, sizeof buf, "%.*e", (int)(rng() % 20), d);
    if (!check_both(buf)) return 1;
    uint32_t b32 = (uint32_t)rng();
    float f;
    memcpy(&f, &b32, 4);
    if (f == f && !std::isinf(f)) {
      snprintf(buf, sizeof buf, "%.*e", (int)(rng() % 12), (double)f);
      if (!check_both(buf)) return 1;
    }
  }
  for (long long r = 0; r < rounds; ++r) {
    std::string t;
    if (rng() & 1) t += '-';
    const int ni = (int)(rng() % 25), nf = (int)(rng() % 25);
    for (int k = 0; k < ni; ++k) t += (char)('0' + rng() % 10);
    if (nf || rng() % 4 == 0) t += '.';
    for (int k = 0; k < nf; ++k) t += (char)('0' + rng() % 10);
    if (ni + nf == 0) t += '7';
    if (rng() % 2) {
      t += (rng() & 1) ? 'e' : 'E';
      const int ex = (int)(rng() % 700) - 350;
      t += std::to_string(ex);
    }
    if (!check_both(t)) return 1;
  }
  // exact midpoints: (2M + 1) * 2^(E - 1) printed in full with %.*f / big %e precision, then nudged
  for (long long r = 0; r < rounds / 20; ++r) {
    uint64_t bits = rng() & 0x7FFFFFFFFFFFFFFFull;
    if ((r & 3) == 0) bits &= 0x000FFFFFFFFFFFFFull;              // subnormal doubles
    if ((r & 3) == 1) bits = (bits & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1000 + rng() % 60) << 52);  // near 1
    double d;
    memcpy(&d, &bits, 8);
    if (d != d || std::isinf(d)) continue;
    const double up = std::nextafter(d, INFINITY);
    if (std::isinf(up)) continue;
    // the midpoint is exact in long double arithmetic only for normal ranges; build its decimal expansion with
    // exact integer arithmetic instead: print d and up with 800 significant digits (glibc prints exactly)
    char a[900], c[900];
    snprintf(a, sizeof a, "%.780e", d);
    snprintf(c, sizeof c, "%.780e", up);
    // midpoint digits = (a + c) / 2 computed on the digit strings when the exponents agree
    const char* ea = strchr(a, 'e');
    const char* ec = strchr(c, 'e');
    if (!ea || !ec || strcmp(ea, ec) != 0) continue;
    std::string da, dc;
    for (const char* p = a; p < ea; ++p) if (*p != '.') da += *p;
    for (const char* p = c; p < ec; ++p) if (*p != '.') dc += *p;
    if (da.size() != dc.size()) continue;
    std::string sum(da.size() + 1, '0');
    int carry = 0;
    for (int k = (int)da.size() - 1; k >= 0; --k) {
      const int v = (da[k] - '0') + (dc[k] - '0') + carry;
      sum[k + 1] = (char)('0' + v % 10);
      carry = v / 10;
    }
    sum[0] = (char)('0' + carry);
    std::string half;  // sum / 2, one more digit
    int rem = 0;
    for (size_t k = 0; k < sum.size(); ++k) {
      const int v = rem * 10 + (sum[k] - '0');
      half += (char)('0' + v / 2);
      rem = v % 2;
    }
    half += rem ? '5' : '0';
    // half has a leading extra digit (from sum[0]); value = 0.half[0] half[1]. … relative to a's layout "d.ddd"
    std::string mid = half.substr(0, 2) + "." + half.substr(2) + ea;  // half[0] is the carry digit: tens place
    // mid is "XY.ddd…e±NN" where a was "Y.ddd…": same scale, so it is the exact midpoint text
    if (!check_value<double>(mid)) return 1;
    std::string below = mid, above = mid;
    const size_t epos = mid.find('e');
    // one unit in a far digit below / above the tie
    above.insert(epos, "0000000001");
    size_t k = epos;
    std::string body = mid.substr(0, epos);
    // below: decrement the last nonzero digit and append 9s
    size_t q = body.find_last_not_of("0.");
    if (q != std::string::npos && body[q] != '.') {
      body[q] = (char)(body[q] - 1);
      for (size_t z = q + 1; z < body.size(); ++z)
        if (body[z] == '0') body[z] = '9';
      below = body + "9999" + mid.substr(epos);
      if (!check_value<double>(below)) return 1;
    }
    (void)k;
    if (!check_value<double>(above)) return 1;
  }
  // float midpoints (exact in double)
  for (long long r = 0; r < rounds / 4; ++r) {
    uint32_t b32 = (uint32_t)rng() & 0x7FFFFFFFu;
    if ((r & 3) == 0) b32 &= 0x007FFFFFu;
    float f;
    memcpy(&f, &b32, 4);
    if (f != f || std::isinf(f)) continue;
    const float up = std::nextafter(f, INFINITY);
    if (std::isinf(up)) continue;
    const double m = ((double)f + (double)up) * 0.5;  // exact
    snprintf(buf, sizeof buf, "%.200e", m);
    if (!check_value<float>(buf)) return 1;
    std::string t(buf);
    const size_t epos = t.find('e');
    std::string above = t;
    above.insert(epos, "1");
    if (!check_value<float>(above)) return 1;
    snprintf(buf, sizeof buf, "%.200e", std::nextafter(m, 0.0));
    if (!check_value<float>(buf)) return 1;
  }
  printf("ok %lld cases, %lld through the big-integer path\n", g_cases, g_slow);
  return 0;
}
