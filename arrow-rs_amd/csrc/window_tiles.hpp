// window_tiles.hpp — host-side bound on the scatter tiles that can hold a window of a predicate's filtered stream.
// Plain C++ (no HIP): the library uses it in filter.hip; tests/cpp/window_tiles_host_test.cpp checks it on the CPU
// against brute force over random selections (a bound one tile short loses rows silently).
#pragma once
#include <algorithm>
#include <cstdint>

// quant[k] = selected rows in count groups [0, k * quant_step), k = 0 .. quant_n; quant[quant_n] = count (all selected rows).
// One count group = rows_per_group rows (a multiple of the tile size T).  Positions [lo, hi) of the filtered stream
// (0-based among the selected rows; hi == 0 or hi > count: to the end) lie in tiles [*t_lo, *t_hi) of the ntiles tiles of
// T rows — a SUPERSET bound: tiles outside it hold no position of the window, tiles inside it may hold none.
inline void ah_window_tiles(int quant_n, int64_t quant_step, const uint64_t* quant, int64_t count, int64_t rows_per_group, int64_t lo,
                            int64_t hi, int64_t T, int64_t ntiles, int64_t* t_lo, int64_t* t_hi) {
  *t_lo = 0;
  *t_hi = ntiles;
  if (quant_n <= 0) return;
  if (hi == 0 || hi > count) hi = count;
  if (hi <= lo) {  // nothing to append
    *t_hi = 0;
    return;
  }
  int a = 0, b = quant_n - 1;
  while (a + 1 < quant_n && (int64_t)quant[a + 1] <= lo) ++a;  // position lo lies in quantile a
  while (b > a && (int64_t)quant[b] >= hi) --b;                // position hi - 1 lies in quantile b
  const int64_t rows_per_q = quant_step * rows_per_group;
  *t_lo = std::min<int64_t>(ntiles, (int64_t)a * rows_per_q / T);
  *t_hi = std::min<int64_t>(ntiles, ((int64_t)b + 1) * rows_per_q / T);
}
