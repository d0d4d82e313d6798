// tests/models/slab_model.cpp -- TEST INFRASTRUCTURE.
//
// CPU model of the *schedule* the HIP large-N path uses (topolow_amd/csrc/relax_slab.hip):
// the reference's per-pair update rules (src/optimization.cpp:203-281 of the reference) are
// applied "row-owner" style -- point i applies only its own half of pair (i,c) -- over S
// column slabs per iteration, with positions frozen inside a stage (all points move at the
// end of the stage).  This is NOT the reference's algorithm order (the oracle is); it exists
// so the HIP kernels can be checked stage-for-stage against an independent implementation,
// and so schedule studies run at C speed.
//
// Build: g++ -O2 -fopenmp -std=gnu++17 -shared -fPIC (tests/conftest.py does it).

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

template <typename real>
void stage(const real* pin, real* pout, int n, int dim, const double* D, const int32_t* T,
           const double* g, const int32_t* ranges, int n_ranges, double k, double c_rep) {
  // pin/pout: row-major n x dim.  D/T: column-major n x n as R passes them; the cell of the
  // unordered pair {i,c} is the upper-triangle one, D[min + max*n] (src/optimization.cpp:217).
  // at most 8 threads, and none for small problems: test machines report hundreds of hardware
  // threads behind a CPU quota, where a default-sized team turns a millisecond loop into seconds
#pragma omp parallel for schedule(dynamic, 16) num_threads(8) if (n >= 1024)
  for (int i = 0; i < n; ++i) {
    real acc[64];
    for (int d = 0; d < dim; ++d) acc[d] = 0;
    const real gi = (real)g[i];
    const real inv_spring = (real)1 / ((real)4 * gi + (real)k);
    const real inv_rep = (real)1 / gi;
    for (int rg = 0; rg < n_ranges; ++rg) {
      for (int c = ranges[2 * rg]; c < ranges[2 * rg + 1]; ++c) {
        if (c == i || c >= n) continue;  // ranges live in the padded column space [0, roundup4(n))
        real delta[64];
        real s = 0;
        for (int d = 0; d < dim; ++d) {
          delta[d] = pin[(size_t)c * dim + d] - pin[(size_t)i * dim + d];
          s += delta[d] * delta[d];
        }
        const real r = std::sqrt(s);
        const real rs = r + (real)0.01;
        const int lo = i < c ? i : c, hi = i < c ? c : i;
        const size_t cell = (size_t)lo + (size_t)hi * n;
        const double t = D[cell];
        const int code = T[cell];
        bool spring = false;
        if (std::isfinite(t)) {
          if (code == 0) spring = true;
          else if (code == 1) spring = r < (real)t;
          else spring = r > (real)t;
        }
        real coef;
        if (spring) coef = (real)2 * (real)k * ((real)t - r) / rs * inv_spring;
        else coef = (real)c_rep / ((real)2 * rs * rs * rs) * inv_rep;
        for (int d = 0; d < dim; ++d) acc[d] += delta[d] * coef;
      }
    }
    for (int d = 0; d < dim; ++d) pout[(size_t)i * dim + d] = pin[(size_t)i * dim + d] - acc[d];
  }
}

}  // namespace

extern "C" {

// One stage.  positions row-major n x dim (float64 in/out; arithmetic in `arith`: 0=f64,1=f32).
// ranges: n_ranges x 2 int32 [begin,end).
int slab_model_stage(const double* pos_in, double* pos_out, int n, int dim, const double* D,
                     const int32_t* T, const int32_t* degrees, const int32_t* ranges,
                     int n_ranges, double k, double c_rep, int arith) {
  std::vector<double> g(n);
  for (int i = 0; i < n; ++i) g[i] = degrees[i] + 1.0;
  if (dim > 64) return 1;
  if (arith == 1) {
    std::vector<float> a((size_t)n * dim), b((size_t)n * dim);
    for (size_t q = 0; q < a.size(); ++q) a[q] = (float)pos_in[q];
    stage<float>(a.data(), b.data(), n, dim, D, T, g.data(), ranges, n_ranges, k, c_rep);
    for (size_t q = 0; q < a.size(); ++q) pos_out[q] = b[q];
  } else {
    stage<double>(pos_in, pos_out, n, dim, D, T, g.data(), ranges, n_ranges, k, c_rep);
  }
  return 0;
}

// A run of iterations.  plan: for iteration `it`, stage `s`: 4 int32 (r0b, r0e, r1b, r1e) at
// plan[(it*max_stages + s)*4]; n_stages[it] stages are used.  k is cooled after each
// iteration (k *= 1-cooling) exactly as the reference does; no controller here (callers
// check MAE through the oracle).  Returns positions after the last iteration and final k.
int slab_model_run(const double* pos_in, double* pos_out, int n, int dim, const double* D,
                   const int32_t* T, const int32_t* degrees, const int32_t* plan,
                   const int32_t* n_stages, int max_stages, int n_iter, double k0,
                   double cooling, double c_rep, int arith, double* k_out) {
  if (dim > 64) return 1;
  std::vector<double> g(n);
  for (int i = 0; i < n; ++i) g[i] = degrees[i] + 1.0;
  const size_t nd = (size_t)n * dim;
  double k = k0;
  if (arith == 1) {
    std::vector<float> a(nd), b(nd);
    for (size_t q = 0; q < nd; ++q) a[q] = (float)pos_in[q];
    for (int it = 0; it < n_iter; ++it) {
      for (int s = 0; s < n_stages[it]; ++s) {
        const int32_t* r = plan + ((size_t)it * max_stages + s) * 4;
        const int nr = (r[3] > r[2]) ? 2 : 1;
        stage<float>(a.data(), b.data(), n, dim, D, T, g.data(), r, nr, k, c_rep);
        a.swap(b);
      }
      k *= (1.0 - cooling);
    }
    for (size_t q = 0; q < nd; ++q) pos_out[q] = a[q];
  } else {
    std::vector<double> a(pos_in, pos_in + nd), b(nd);
    for (int it = 0; it < n_iter; ++it) {
      for (int s = 0; s < n_stages[it]; ++s) {
        const int32_t* r = plan + ((size_t)it * max_stages + s) * 4;
        const int nr = (r[3] > r[2]) ? 2 : 1;
        stage<double>(a.data(), b.data(), n, dim, D, T, g.data(), r, nr, k, c_rep);
        a.swap(b);
      }
      k *= (1.0 - cooling);
    }
    std::memcpy(pos_out, a.data(), nd * sizeof(double));
  }
  if (k_out) *k_out = k;
  return 0;
}

}  // extern "C"
