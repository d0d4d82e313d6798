// topolow_amd/csrc/relax_fold.h -- host-side construction of one cross-validation fold's problem from
// the cell list of the full matrix (no device code).
//
// Mirrors what the reference does per fold in R (R/adaptive_sampling.R:2600-2640 masks the held-out
// cells, then euclidean_embedding's pre-processing R/core.R:269-436 runs on the masked n x n matrix):
// ordering by mean dissimilarity, degrees, the upper-triangle edge list in column-major scan order,
// and the out-of-sample cells -- but from the list of non-NA cells, O(E log E) instead of several
// n x n passes.  The Python twin is topolow_amd/cv.py: FoldBuilder.fold_numpy, to which this must be
// (and is tested to be) identical, including the last bit of the means that decide the ordering:
// NumPy sums a row of the zero-filled matrix pairwise, and zeros are exact identities of that
// summation tree, so the same tree over the non-zero cells alone gives the same sum.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/topolow_relax.h"

namespace topolow {

// Sum of a dense row of length (hi - lo) whose non-zero entries are (pos[q], val[q]), q in [b, e),
// positions ascending -- in the association order of NumPy's pairwise summation (blocks of 128,
// eight strided accumulators per block, remainder added one by one).
inline double fold_pairwise(const int32_t* pos, const double* val, int b, int e, int lo, int hi) {
  const int len = hi - lo;
  if (len < 8) {
    double res = 0.0;
    for (int q = b; q < e; ++q) res += val[q];
    return res;
  }
  if (len <= 128) {
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int body = len - len % 8;
    int q = b;
    for (; q < e && pos[q] - lo < body; ++q) r[(pos[q] - lo) & 7] += val[q];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; q < e; ++q) res += val[q];
    return res;
  }
  int n2 = len / 2;
  n2 -= n2 % 8;
  const int mid = lo + n2;
  const int m = (int)(std::lower_bound(pos + b, pos + e, mid) - pos);
  return fold_pairwise(pos, val, b, m, lo, mid) + fold_pairwise(pos, val, m, e, mid, hi);
}

inline int fold_problem(const topolow_cell_list* L, const int64_t* picks, int64_t n_picks,
                        int32_t preserve_order, int32_t named, int32_t* order, int32_t* degrees,
                        int32_t* edge_i, int32_t* edge_j, double* edge_dist, int32_t* edge_thresh,
                        int64_t* n_edges, int32_t* hold_i, int32_t* hold_j, double* hold_truth,
                        int64_t* n_hold, double* numeric_max) {
  const int n = L->n;
  const int64_t nc = L->n_cells;
  if (n < 1 || nc < 0) return TOPOLOW_ERR_BAD_ARGUMENT;
  // 1. the held-out cells and their mirrors, every cell once, ascending linear (column-major) index
  std::vector<int64_t> lin;
  lin.reserve((size_t)n_picks * 2);
  for (int64_t q = 0; q < n_picks; ++q) {
    const int64_t r = picks[q] % n, c = picks[q] / n;
    if (picks[q] < 0 || c >= n) return TOPOLOW_ERR_BAD_ARGUMENT;
    lin.push_back(r + c * n);
    lin.push_back(c + r * n);
  }
  std::sort(lin.begin(), lin.end());
  lin.erase(std::unique(lin.begin(), lin.end()), lin.end());
  std::vector<char> keep((size_t)nc, 1);
  std::vector<int64_t> dropped;
  for (int64_t x : lin) {
    const int64_t at = L->pos_of[x];
    if (at >= 0) { keep[(size_t)at] = 0; dropped.push_back(at); }
  }
  // 2. ordering by mean dissimilarity (R/core.R:269-319): mean of row mean and column mean over the
  //    non-NA off-diagonal cells, threshold prefixes stripped
  bool reordered = false;
  std::vector<int32_t> inv(n);
  for (int i = 0; i < n; ++i) inv[i] = i;
  if (n > 1 && !preserve_order) {
    std::vector<double> avg(n);
    std::vector<int32_t> pos;
    std::vector<double> val;
    std::vector<double> csum(n, 0.0);
    std::vector<int32_t> ccnt(n, 0);
    // column sums: NumPy adds the rows of the matrix one after another, i.e. per column in
    // ascending row order -- the order of the (column-major) cell list itself
    for (int64_t q = 0; q < nc; ++q) {
      if (!keep[(size_t)q] || L->row[q] == L->col[q]) continue;
      csum[L->col[q]] += L->value[q];
      ccnt[L->col[q]] += 1;
    }
    for (int i = 0; i < n; ++i) {
      pos.clear();
      val.clear();
      for (int64_t p = L->row_ptr[i]; p < L->row_ptr[i + 1]; ++p) {
        const int64_t q = L->by_row[p];
        if (!keep[(size_t)q] || L->col[q] == i) continue;
        pos.push_back(L->col[q]);
        val.push_back(L->value[q]);
      }
      const double rs = fold_pairwise(pos.data(), val.data(), 0, (int)pos.size(), 0, n);
      const double rm = pos.empty() ? NAN : rs / (double)pos.size();
      const double cm = ccnt[i] == 0 ? NAN : csum[i] / (double)ccnt[i];
      const double a = (rm + cm) / 2.0;
      avg[i] = std::isnan(a) ? 0.0 : a;
    }
    int positive = 0;
    for (int i = 0; i < n; ++i) positive += avg[i] > 0 ? 1 : 0;
    if (positive > 1) {
      std::vector<int32_t> ord(n);
      for (int i = 0; i < n; ++i) ord[i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return avg[x] < avg[y]; });
      for (int i = 0; i < n; ++i) { order[i] = ord[i]; inv[ord[i]] = i; }
      reordered = true;
    }
  }
  if (!reordered) order[0] = -1;
  // 3. degrees (non-NA cells of a row, diagonal included: R/core.R:341), edges, largest numeric value
  for (int i = 0; i < n; ++i) degrees[i] = 0;
  struct Edge { int32_t i, j; double d; int32_t t; };
  std::vector<Edge> edges;
  edges.reserve((size_t)nc / 2 + 1);
  double vmax = NAN;
  for (int64_t q = 0; q < nc; ++q) {
    if (!keep[(size_t)q]) continue;
    const int32_t r = inv[L->row[q]], c = inv[L->col[q]];
    degrees[r] += 1;
    if (L->code[q] == 0 && !(L->value[q] <= vmax)) vmax = L->value[q];   // NaN-safe running maximum
    if (r < c) edges.push_back(Edge{r, c, L->value[q], L->code[q]});
  }
  std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) {
    return a.j != b.j ? a.j < b.j : a.i < b.i;                           // column-major scan
  });
  for (size_t q = 0; q < edges.size(); ++q) {
    edge_i[q] = edges[q].i; edge_j[q] = edges[q].j; edge_dist[q] = edges[q].d; edge_thresh[q] = edges[q].t;
  }
  *n_edges = (int64_t)edges.size();
  *numeric_max = vmax;
  // 4. out-of-sample cells: held out AND numeric in the truth; lined up with the prediction by name
  //    (R/error_metrics.R:100-112) -- an unnamed matrix is compared in the returned numbering
  int64_t nh = 0;
  for (int64_t at : dropped) {
    if (L->code[at] != 0) continue;
    hold_i[nh] = named ? inv[L->row[at]] : L->row[at];
    hold_j[nh] = named ? inv[L->col[at]] : L->col[at];
    hold_truth[nh] = L->value[at];
    ++nh;
  }
  *n_hold = nh;
  return TOPOLOW_OK;
}

}  // namespace topolow
