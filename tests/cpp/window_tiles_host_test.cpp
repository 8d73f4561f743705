// window_tiles_host_test.cpp — TEST INFRASTRUCTURE: csrc/window_tiles.hpp against brute force on the CPU.
// The coalescer appends positions [lo, hi) of a batch's filtered stream with a scatter launch restricted to the tiles the
// bound returns; a tile that holds a position of the window but lies outside the bound would lose rows silently.  Random
// selections (uniform, empty ends, narrow bands, bands on group boundaries), every tile size the kernels use, group counts
// that do and do not divide by the quantile step, windows at every interesting position.
//   usage: window_tiles_host_test <cases> <seed>      exit 0 + "WINDOW_TILES_OK" = every bound was a superset
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../arrow-rs_amd/csrc/window_tiles.hpp"

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  st ^= st >> 12;
  st ^= st << 25;
  st ^= st >> 27;
  return st * 2685821657736338717ull;
}
static int64_t below(int64_t n) { return n > 0 ? (int64_t)(rnd() % (uint64_t)n) : 0; }

int main(int argc, char** argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 2000;
  if (argc > 2) st ^= (uint64_t)atoll(argv[2]) * 0xD1B54A32D192ED03ull;
  const int64_t ROWS_PER_GROUP = 65536;  // 64 chunks x 1024 rows (group_shift 6)
  const int QUANTS = 32;
  long checked = 0, tight = 0;
  for (long c = 0; c < cases; ++c) {
    const int64_t T = (int64_t[]){4096, 2048, 1024}[below(3)];
    const int64_t len = 1 + below(c % 7 == 0 ? 40 * ROWS_PER_GROUP : 9 * ROWS_PER_GROUP);
    const int64_t ngroups = (len + ROWS_PER_GROUP - 1) / ROWS_PER_GROUP, ntiles = (len + T - 1) / T;
    // per-group selected counts (only the group totals matter to the bound; positions inside a group are irrelevant)
    std::vector<int64_t> g((size_t)ngroups);
    const int shape = (int)below(5);
    for (int64_t i = 0; i < ngroups; ++i) {
      const int64_t rows = std::min<int64_t>(ROWS_PER_GROUP, len - i * ROWS_PER_GROUP);
      int64_t k = 0;
      if (shape == 0) k = below(rows + 1);
      else if (shape == 1) k = i * 3 < ngroups ? 0 : below(rows + 1);            // nothing in the first third
      else if (shape == 2) k = i * 3 > 2 * ngroups ? 0 : below(rows + 1);        // nothing in the last third
      else if (shape == 3) k = (i == below(ngroups)) ? rows : 0;                 // one full group (maybe none)
      else k = below(3) == 0 ? below(rows + 1) : 0;                              // sparse groups
      g[(size_t)i] = k;
    }
    std::vector<int64_t> pre((size_t)ngroups + 1, 0);
    for (int64_t i = 0; i < ngroups; ++i) pre[(size_t)i + 1] = pre[(size_t)i] + g[(size_t)i];
    const int64_t count = pre[(size_t)ngroups];
    // what the group-scan kernel publishes: the exclusive prefix of every quant_step-th group
    const int64_t step = (ngroups + QUANTS - 1) / QUANTS;
    const int quant_n = (int)((ngroups + step - 1) / step);
    uint64_t quant[QUANTS + 1] = {};
    for (int k = 0; k < quant_n; ++k) quant[k] = (uint64_t)pre[(size_t)(k * step)];
    quant[quant_n] = (uint64_t)count;
    for (int w = 0; w < 24; ++w) {
      int64_t lo = below(count + 2), hi = below(count + 3);
      if (w == 0) lo = 0, hi = 0;                // the whole stream
      if (w == 1) lo = 0, hi = count;
      if (w == 2 && count) lo = count - 1, hi = count;
      if (w == 3 && count) lo = 0, hi = 1;
      if (w == 4 && quant_n > 1) lo = (int64_t)quant[1 + below(quant_n - 1)], hi = lo + 1;      // first position of a quantile
      if (w == 5 && quant_n > 1) hi = (int64_t)quant[1 + below(quant_n - 1)], lo = hi > 0 ? hi - 1 : 0;  // last position before one
      int64_t t_lo = -1, t_hi = -1;
      ah_window_tiles(quant_n, step, quant, count, ROWS_PER_GROUP, lo, hi, T, ntiles, &t_lo, &t_hi);
      if (t_lo < 0 || t_hi > ntiles || t_hi < t_lo) {
        fprintf(stderr, "case %ld: bound [%lld, %lld) outside [0, %lld]\n", c, (long long)t_lo, (long long)t_hi, (long long)ntiles);
        return 1;
      }
      int64_t eh = (hi == 0 || hi > count) ? count : hi;
      // brute force at GROUP granularity: a group holding any position of [lo, eh) must be covered by the tile bound
      for (int64_t i = 0; i < ngroups && eh > lo; ++i) {
        const int64_t first = pre[(size_t)i], last = pre[(size_t)i + 1];  // positions [first, last) live in group i
        if (last <= lo || first >= eh || last == first) continue;
        const int64_t tile0 = i * ROWS_PER_GROUP / T, tile1 = std::min(ntiles, (i + 1) * ROWS_PER_GROUP / T);
        if (tile0 < t_lo || tile1 > t_hi) {
          fprintf(stderr, "case %ld: window [%lld, %lld) has rows in group %lld (tiles [%lld, %lld)) outside the bound [%lld, %lld)\n", c,
                  (long long)lo, (long long)eh, (long long)i, (long long)tile0, (long long)tile1, (long long)t_lo, (long long)t_hi);
          return 1;
        }
      }
      ++checked;
      if (t_hi - t_lo < ntiles) ++tight;
    }
  }
  printf("window_tiles_host_test: %ld windows checked, %ld of them bounded below the full tile range\nWINDOW_TILES_OK\n", checked, tight);
  return 0;
}
