/* topolow_amd/r/topolow_shim.c -- R `.Call` shim over libtopolow_relax.so.
 *
 * Drop-in for the reference's generated glue (src/RcppExports.cpp:16-49 of omid-arhami/topolow
 * v2.1.0) and, through it, for src/optimization.cpp: it exports the SAME native symbol
 *     _topolow_optimize_layout_exact_cpp   (16 SEXP arguments, registered with arity 16)
 * and the same R_init_topolow, so the reference's R code (R/RcppExports.R:4-6, R/core.R:439-456)
 * runs unchanged -- `euclidean_embedding()` keeps its signature and its returned object.
 * Additional entries (INTEGRATION.md shows the R side of each):
 *     _topolow_optimize_layout_exact_batch  many `.Call` payloads in one launch -- the per-fold loop
 *                                           of likelihood_function (R/adaptive_sampling.R:2604-2693)
 *     _topolow_cv_fold                      one fold's payload from the list of non-NA cells
 *                                           (R/adaptive_sampling.R:2608-2616 + R/core.R:269-436)
 *     _topolow_cv_sweep                     all folds of all parameter sets in one call: per-fold out-of-sample scores
 *     _topolow_est_distances                as.matrix(dist(positions))  (R/core.R:474)
 *     _topolow_est_distances_cols           a block of its columns, for n x n results too large to hold
 *
 * No logic lives here: unmarshal, call the library, marshal, and turn error codes into R errors
 * AFTER every native resource has been released (Rf_error longjmps).
 * R is absent from the build image, so this file is compiled only where R is installed:
 *     R CMD SHLIB topolow_shim.c -L<dir> -ltopolow_relax -I<repo>/include
 * Backend options travel through R options(), never through the function signatures:
 *     options(topolow.seed = 1L, topolow.schedule = "auto"|"slab"|"gs",
 *             topolow.precision = "auto"|"f32"|"f64", topolow.device = 0L,
 *             topolow.devices = c(0L, 1L, ...))      # ONE embedding row-sharded over these GPUs
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "topolow_relax.h"

/* R_CheckUserInterrupt() longjmps; probing it through R_ToplevelExec keeps control here so the
 * library can release its device memory before the interrupt is re-raised. */
static void probe_interrupt(void* dummy) { (void)dummy; R_CheckUserInterrupt(); }
static int32_t interrupt_pending(void* user) {
  (void)user;
  return R_ToplevelExec(probe_interrupt, NULL) == FALSE;
}

/* verbose lines: Rprintf obeys sink() and the console, as Rcpp::Rcout does in the reference */
static void print_line(const char* line, void* user) {
  (void)user;
  Rprintf("%s", line);
}

static int opt_int(const char* name, int dflt) {
  SEXP v = Rf_GetOption1(Rf_install(name));
  if (v == R_NilValue || Rf_length(v) < 1) return dflt;
  return Rf_asInteger(v);
}

static int opt_choice(const char* name, const char* a, int va, const char* b, int vb, int dflt) {
  SEXP v = Rf_GetOption1(Rf_install(name));
  if (v == R_NilValue || !Rf_isString(v) || Rf_length(v) < 1) return dflt;
  const char* s = CHAR(STRING_ELT(v, 0));
  if (strcmp(s, a) == 0) return va;
  if (strcmp(s, b) == 0) return vb;
  return dflt;
}

static uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

/* Seed of the pair-order stream.  The reference seeds its shuffle from std::random_device and draws
 * NOTHING from R's RNG (src/optimization.cpp:153-154; the generated RNGScope only saves and restores
 * the state, src/RcppExports.cpp:19), so neither may this shim: options(topolow.seed) if set;
 * otherwise a hash of `.Random.seed` READ in place (so set.seed() still makes a whole script
 * reproducible, and successive calls differ because euclidean_embedding draws its start positions
 * in between, R/core.R:412) mixed with a per-session call counter; without an initialised R RNG,
 * clock and pid. */
static uint64_t order_seed(void) {
  static uint64_t calls = 0;
  SEXP seed = Rf_GetOption1(Rf_install("topolow.seed"));
  if (seed != R_NilValue && Rf_length(seed) >= 1) return (uint64_t)Rf_asReal(seed);
  ++calls;
  SEXP rs = Rf_findVarInFrame(R_GlobalEnv, Rf_install(".Random.seed"));
  if (rs != R_UnboundValue && Rf_isInteger(rs) && XLENGTH(rs) > 0) {
    uint64_t h = 0x243f6a8885a308d3ull;
    const int* v = INTEGER(rs);
    for (R_xlen_t q = 0; q < XLENGTH(rs); ++q) h = mix64(h ^ (uint64_t)(uint32_t)v[q]);
    return mix64(h ^ (calls * 0xd1342543de82ef95ull));
  }
  return mix64((uint64_t)time(NULL) ^ ((uint64_t)getpid() << 32) ^ (uint64_t)clock() ^ calls);
}

/* options(topolow.*) -> topolow_options; `devices` needs room for the ordinals of topolow.devices */
static void fill_options(topolow_options* opt, int32_t* devices, int max_devices) {
  topolow_default_options(opt);
  opt->seed = order_seed();
  opt->schedule = opt_choice("topolow.schedule", "slab", TOPOLOW_SCHEDULE_SLAB, "gs",
                             TOPOLOW_SCHEDULE_GS, TOPOLOW_SCHEDULE_AUTO);
  opt->precision = opt_choice("topolow.precision", "f32", TOPOLOW_PRECISION_F32, "f64",
                              TOPOLOW_PRECISION_F64, TOPOLOW_PRECISION_AUTO);
  opt->device = opt_int("topolow.device", -1);
  opt->slab_stages = opt_int("topolow.slab_stages", 0);
  opt->gs_max_n = opt_int("topolow.gs_max_n", 0);
  opt->interrupt_cb = interrupt_pending;   /* polled every 50 iterations, reference :364 */
  opt->print_cb = print_line;
  SEXP devs = Rf_GetOption1(Rf_install("topolow.devices"));
  if (devs != R_NilValue && Rf_length(devs) >= 1 && (Rf_isInteger(devs) || Rf_isReal(devs))) {
    int nd = Rf_length(devs);
    if (nd > max_devices) nd = max_devices;
    for (int q = 0; q < nd; ++q)
      devices[q] = Rf_isInteger(devs) ? INTEGER(devs)[q] : (int32_t)REAL(devs)[q];
    opt->devices = devices;
    opt->n_devices = nd;
  }
}

static SEXP named_list(int n, const char* const* names) {
  SEXP out = PROTECT(Rf_allocVector(VECSXP, n));
  SEXP nm = PROTECT(Rf_allocVector(STRSXP, n));
  for (int q = 0; q < n; ++q) SET_STRING_ELT(nm, q, Rf_mkChar(names[q]));
  Rf_setAttrib(out, R_NamesSymbol, nm);
  UNPROTECT(2);
  return out;
}

SEXP _topolow_optimize_layout_exact_cpp(SEXP initial_positionsSEXP, SEXP dissimilarity_matrixSEXP,
                                        SEXP threshold_matrixSEXP, SEXP degreesSEXP,
                                        SEXP edge_iSEXP, SEXP edge_jSEXP, SEXP edge_distSEXP,
                                        SEXP edge_threshSEXP, SEXP n_iterSEXP, SEXP k0SEXP,
                                        SEXP cooling_rateSEXP, SEXP c_repulsionSEXP,
                                        SEXP relative_epsilonSEXP, SEXP convergence_windowSEXP,
                                        SEXP convergence_check_freqSEXP, SEXP verboseSEXP) {
  /* argument types exactly as R/core.R:439-456 passes them */
  if (!Rf_isReal(initial_positionsSEXP) || !Rf_isMatrix(initial_positionsSEXP))
    Rf_error("initial_positions must be a numeric matrix");
  if (!Rf_isReal(dissimilarity_matrixSEXP) || !Rf_isInteger(threshold_matrixSEXP))
    Rf_error("dissimilarity_matrix must be double and threshold_matrix integer");
  const int n = Rf_nrows(initial_positionsSEXP);
  const int ndim = Rf_ncols(initial_positionsSEXP);
  const R_xlen_t n_edges = XLENGTH(edge_iSEXP);

  topolow_options opt;
  int32_t devices[64];
  fill_options(&opt, devices, 64);

  SEXP positions = PROTECT(Rf_allocMatrix(REALSXP, n, ndim));
  int converged = 0, iterations = 0;
  double final_mae = 0.0, final_k = 0.0;
  char err[512];
  err[0] = '\0';

  /* inputs are R-owned and only read (the reference deep-copies positions, :134) */
  const int rc = topolow_optimize_layout_exact(
      REAL(initial_positionsSEXP), n, ndim, REAL(dissimilarity_matrixSEXP),
      INTEGER(threshold_matrixSEXP), INTEGER(degreesSEXP), INTEGER(edge_iSEXP),
      INTEGER(edge_jSEXP), REAL(edge_distSEXP), INTEGER(edge_threshSEXP), (int64_t)n_edges,
      Rf_asInteger(n_iterSEXP), Rf_asReal(k0SEXP), Rf_asReal(cooling_rateSEXP),
      Rf_asReal(c_repulsionSEXP), Rf_asReal(relative_epsilonSEXP),
      Rf_asInteger(convergence_windowSEXP), Rf_asInteger(convergence_check_freqSEXP),
      Rf_asLogical(verboseSEXP), &opt, REAL(positions), &converged, &iterations, &final_mae,
      &final_k, NULL, err, sizeof err);
  if (rc != TOPOLOW_OK) {
    UNPROTECT(1);
    /* every device buffer is already released inside the library */
    if (rc == TOPOLOW_ERR_INTERRUPTED) Rf_onintr();   /* re-raise the user's interrupt */
    Rf_error("%s", err[0] ? err : "libtopolow_relax failed");
  }

  /* list(positions, converged, iterations, final_mae, final_k) -- src/optimization.cpp:375-381 */
  static const char* const names[] = {"positions", "converged", "iterations", "final_mae", "final_k"};
  SEXP out = PROTECT(named_list(5, names));
  SET_VECTOR_ELT(out, 0, positions);
  SET_VECTOR_ELT(out, 1, Rf_ScalarLogical(converged));
  SET_VECTOR_ELT(out, 2, Rf_ScalarInteger(iterations));
  SET_VECTOR_ELT(out, 3, Rf_ScalarReal(final_mae));
  SET_VECTOR_ELT(out, 4, Rf_ScalarReal(final_k));
  UNPROTECT(2);
  return out;
}

/* Many embeddings in one launch.  `calls`: a list whose elements are lists holding the 16 arguments
 * of _topolow_optimize_layout_exact_cpp in its order (dissimilarity_matrix and threshold_matrix may
 * both be NULL: the edge list then IS the matrix), optionally followed by holdout_i, holdout_j
 * (0-based) and holdout_truth.  Returns one list per call:
 *   positions, converged, iterations, final_mae, final_k   -- as the single call
 *   iterations_run, error (NA or the message the single call would have raised),
 *   holdout_sum_abs, holdout_count                          -- sum |truth - distance| over the holdout
 * Seeds: options(topolow.seed) + index of the call, or the order_seed() stream. */
SEXP _topolow_optimize_layout_exact_batch(SEXP callsSEXP) {
  if (TYPEOF(callsSEXP) != VECSXP) Rf_error("calls must be a list of .Call argument lists");
  const int count = Rf_length(callsSEXP);
  topolow_problem* pb = (topolow_problem*)R_alloc(count > 0 ? count : 1, sizeof(topolow_problem));
  topolow_result* rs = (topolow_result*)R_alloc(count > 0 ? count : 1, sizeof(topolow_result));
  memset(pb, 0, sizeof(topolow_problem) * (size_t)(count > 0 ? count : 1));
  memset(rs, 0, sizeof(topolow_result) * (size_t)(count > 0 ? count : 1));
  const uint64_t seed0 = order_seed();
  SEXP out = PROTECT(Rf_allocVector(VECSXP, count));
  for (int b = 0; b < count; ++b) {
    SEXP a = VECTOR_ELT(callsSEXP, b);
    if (TYPEOF(a) != VECSXP || Rf_length(a) < 16) {
      UNPROTECT(1);
      Rf_error("call %d: expected the 16 arguments of optimize_layout_exact_cpp", b + 1);
    }
    SEXP pos0 = VECTOR_ELT(a, 0), D = VECTOR_ELT(a, 1), T = VECTOR_ELT(a, 2);
    if (!Rf_isReal(pos0) || !Rf_isMatrix(pos0) || (D == R_NilValue) != (T == R_NilValue) ||
        (D != R_NilValue && (!Rf_isReal(D) || !Rf_isInteger(T)))) {
      UNPROTECT(1);
      Rf_error("call %d: bad initial_positions / dissimilarity_matrix / threshold_matrix", b + 1);
    }
    topolow_problem* p = &pb[b];
    p->n = Rf_nrows(pos0);
    p->ndim = Rf_ncols(pos0);
    p->initial_positions = REAL(pos0);
    p->dissimilarity_matrix = D == R_NilValue ? NULL : REAL(D);
    p->threshold_matrix = T == R_NilValue ? NULL : INTEGER(T);
    p->degrees = INTEGER(VECTOR_ELT(a, 3));
    p->edge_i = INTEGER(VECTOR_ELT(a, 4));
    p->edge_j = INTEGER(VECTOR_ELT(a, 5));
    p->edge_dist = REAL(VECTOR_ELT(a, 6));
    p->edge_thresh = INTEGER(VECTOR_ELT(a, 7));
    p->n_edges = (int64_t)XLENGTH(VECTOR_ELT(a, 4));
    p->n_iter = Rf_asInteger(VECTOR_ELT(a, 8));
    p->k0 = Rf_asReal(VECTOR_ELT(a, 9));
    p->cooling_rate = Rf_asReal(VECTOR_ELT(a, 10));
    p->c_repulsion = Rf_asReal(VECTOR_ELT(a, 11));
    p->relative_epsilon = Rf_asReal(VECTOR_ELT(a, 12));
    p->convergence_window = Rf_asInteger(VECTOR_ELT(a, 13));
    p->convergence_check_freq = Rf_asInteger(VECTOR_ELT(a, 14));
    p->seed = mix64(seed0 + (uint64_t)b);
    if (Rf_length(a) >= 19 && VECTOR_ELT(a, 16) != R_NilValue) {
      p->holdout_i = INTEGER(VECTOR_ELT(a, 16));
      p->holdout_j = INTEGER(VECTOR_ELT(a, 17));
      p->holdout_truth = REAL(VECTOR_ELT(a, 18));
      p->n_holdout = (int64_t)XLENGTH(VECTOR_ELT(a, 16));
    }
    static const char* const names[] = {"positions", "converged", "iterations", "final_mae", "final_k",
                                        "iterations_run", "error", "holdout_sum_abs", "holdout_count"};
    SEXP r = PROTECT(named_list(9, names));
    SEXP positions = PROTECT(Rf_allocMatrix(REALSXP, p->n, p->ndim));
    SET_VECTOR_ELT(r, 0, positions);
    SET_VECTOR_ELT(out, b, r);
    UNPROTECT(2);
    rs[b].positions_out = REAL(positions);
  }
  char err[512];
  err[0] = '\0';
  const int precision = opt_choice("topolow.precision", "f32", TOPOLOW_PRECISION_F32, "f64",
                                   TOPOLOW_PRECISION_F64, TOPOLOW_PRECISION_F64);
  const int rc = topolow_optimize_layout_exact_batch(pb, rs, count, precision, opt_int("topolow.device", -1),
                                                     NULL, err, sizeof err);
  if (rc != TOPOLOW_OK) {
    UNPROTECT(1);
    Rf_error("%s", err[0] ? err : "libtopolow_relax failed");
  }
  for (int b = 0; b < count; ++b) {
    SEXP r = VECTOR_ELT(out, b);
    SET_VECTOR_ELT(r, 1, Rf_ScalarLogical(rs[b].converged));
    SET_VECTOR_ELT(r, 2, Rf_ScalarInteger(rs[b].iterations));
    SET_VECTOR_ELT(r, 3, Rf_ScalarReal(rs[b].final_mae));
    SET_VECTOR_ELT(r, 4, Rf_ScalarReal(rs[b].final_k));
    SET_VECTOR_ELT(r, 5, Rf_ScalarInteger(rs[b].iterations_run));
    if (rs[b].error_code == TOPOLOW_ERR_NONFINITE) {
      char msg[128];
      snprintf(msg, sizeof msg, "Numerical instability at iteration %d. Reduce k0 or c_repulsion.",
               rs[b].error_iteration);
      SET_VECTOR_ELT(r, 6, Rf_mkString(msg));
    } else {
      SET_VECTOR_ELT(r, 6, Rf_ScalarString(R_NaString));
    }
    SET_VECTOR_ELT(r, 7, Rf_ScalarReal(rs[b].holdout_sum_abs));
    SET_VECTOR_ELT(r, 8, Rf_ScalarReal((double)rs[b].holdout_count));
  }
  UNPROTECT(1);
  return out;
}

/* One fold's `.Call` payload from the non-NA cells of the full matrix, listed in column-major order
 * as which(!is.na(m), arr.ind = TRUE) lists them: row, col (0-based INTSXP), value (REALSXP,
 * threshold prefix stripped), code (INTSXP: 0 none, 1 ">", -1 "<"), n; picks = the held-out linear
 * indices (0-based, REALSXP or INTSXP; their mirrors are held out too).  Returns
 * list(order (0-based, or NULL when the input order is kept), degrees, edge_i, edge_j, edge_dist,
 *      edge_thresh, holdout_i, holdout_j, holdout_truth, numeric_max). */
SEXP _topolow_cv_fold(SEXP rowSEXP, SEXP colSEXP, SEXP valueSEXP, SEXP codeSEXP, SEXP nSEXP,
                      SEXP picksSEXP, SEXP preserve_orderSEXP, SEXP namedSEXP) {
  if (!Rf_isInteger(rowSEXP) || !Rf_isInteger(colSEXP) || !Rf_isReal(valueSEXP) || !Rf_isInteger(codeSEXP))
    Rf_error("row, col, code must be integer and value double");
  const int n = Rf_asInteger(nSEXP);
  const int64_t m = (int64_t)XLENGTH(rowSEXP);
  if (n < 1 || XLENGTH(colSEXP) != m || XLENGTH(valueSEXP) != m || XLENGTH(codeSEXP) != m)
    Rf_error("cell columns must have one length and n must be positive");
  const int64_t np = (int64_t)XLENGTH(picksSEXP);
  int64_t* pos_of = (int64_t*)R_alloc((size_t)n * n, sizeof(int64_t));
  int64_t* by_row = (int64_t*)R_alloc(m > 0 ? m : 1, sizeof(int64_t));
  int64_t* row_ptr = (int64_t*)R_alloc((size_t)n + 1, sizeof(int64_t));
  int64_t* picks = (int64_t*)R_alloc(np > 0 ? np : 1, sizeof(int64_t));
  for (int64_t q = 0; q < np; ++q)
    picks[q] = Rf_isInteger(picksSEXP) ? (int64_t)INTEGER(picksSEXP)[q] : (int64_t)REAL(picksSEXP)[q];
  if (topolow_cell_list_index(n, m, INTEGER(rowSEXP), INTEGER(colSEXP), pos_of, by_row, row_ptr) != TOPOLOW_OK)
    Rf_error("cell list: row / col out of range");
  topolow_cell_list cells;
  memset(&cells, 0, sizeof cells);
  cells.n = n;
  cells.n_cells = m;
  cells.row = INTEGER(rowSEXP);
  cells.col = INTEGER(colSEXP);
  cells.value = REAL(valueSEXP);
  cells.code = INTEGER(codeSEXP);
  cells.pos_of = pos_of;
  cells.by_row = by_row;
  cells.row_ptr = row_ptr;
  const size_t cap = (size_t)(m > 0 ? m : 1);
  int32_t* order = (int32_t*)R_alloc((size_t)n, sizeof(int32_t));
  int32_t* degrees = (int32_t*)R_alloc((size_t)n, sizeof(int32_t));
  int32_t* ei = (int32_t*)R_alloc(cap, sizeof(int32_t));
  int32_t* ej = (int32_t*)R_alloc(cap, sizeof(int32_t));
  double* ed = (double*)R_alloc(cap, sizeof(double));
  int32_t* et = (int32_t*)R_alloc(cap, sizeof(int32_t));
  int32_t* hi = (int32_t*)R_alloc(cap, sizeof(int32_t));
  int32_t* hj = (int32_t*)R_alloc(cap, sizeof(int32_t));
  double* ht = (double*)R_alloc(cap, sizeof(double));
  int64_t ne = 0, nh = 0;
  double vmax = 0.0;
  if (topolow_cv_fold(&cells, picks, np, Rf_asLogical(preserve_orderSEXP), Rf_asLogical(namedSEXP), order,
                      degrees, ei, ej, ed, et, &ne, hi, hj, ht, &nh, &vmax) != TOPOLOW_OK)
    Rf_error("topolow_cv_fold failed");
  static const char* const names[] = {"order", "degrees", "edge_i", "edge_j", "edge_dist", "edge_thresh",
                                      "holdout_i", "holdout_j", "holdout_truth", "numeric_max"};
  SEXP out = PROTECT(named_list(10, names));
  if (order[0] >= 0) {
    SEXP v = PROTECT(Rf_allocVector(INTSXP, n));
    memcpy(INTEGER(v), order, sizeof(int32_t) * (size_t)n);
    SET_VECTOR_ELT(out, 0, v);
    UNPROTECT(1);
  }
  {
    SEXP v = PROTECT(Rf_allocVector(INTSXP, n));
    memcpy(INTEGER(v), degrees, sizeof(int32_t) * (size_t)n);
    SET_VECTOR_ELT(out, 1, v);
    UNPROTECT(1);
  }
  const int32_t* isrc[] = {ei, ej, NULL, et, hi, hj, NULL};
  const double* dsrc[] = {NULL, NULL, ed, NULL, NULL, NULL, ht};
  for (int q = 0; q < 7; ++q) {
    const int64_t len = q < 4 ? ne : nh;
    SEXP v = PROTECT(Rf_allocVector(isrc[q] ? INTSXP : REALSXP, (R_xlen_t)len));
    if (isrc[q]) memcpy(INTEGER(v), isrc[q], sizeof(int32_t) * (size_t)len);
    else memcpy(REAL(v), dsrc[q], sizeof(double) * (size_t)len);
    SET_VECTOR_ELT(out, 2 + q, v);
    UNPROTECT(1);
  }
  SET_VECTOR_ELT(out, 9, Rf_ScalarReal(vmax));
  UNPROTECT(1);
  return out;
}

/* A whole cross-validation sweep (topolow_cv_sweep): ONE argument, a list of 20 -- NAMED (any order; the names are
 * those below, kCvSweepArgs) or unnamed in this order --
 *   row, col (integer, 0-based), value (double), code (integer): the non-NA cells of the matrix, column-major order;
 *   n; named (logical); preserve_order (logical);
 *   ndim (integer), k0, cooling_rate, c_repulsion (double): one entry per fold;
 *   picks (held-out cells, linear column-major indices, all folds one after another), picks_offset (F + 1);
 *   unit_draws (per fold ndim x (n - 1) runif(0, 1) numbers, row-major, one after another), draws_offset (F + 1);
 *   seeds (one per fold); n_iter; relative_epsilon; convergence_counter; convergence_check_freq.
 * Returns list(holdout_sum_abs, holdout_count, iterations, converged, error_code, device_seconds): what
 * likelihood_function pools per parameter set (R/adaptive_sampling.R:2660-2720). */
static int64_t* as_i64(SEXP v, int64_t* len) {
  const int64_t m = (int64_t)XLENGTH(v);
  int64_t* out = (int64_t*)R_alloc(m > 0 ? (size_t)m : 1, sizeof(int64_t));
  for (int64_t q = 0; q < m; ++q) out[q] = Rf_isInteger(v) ? (int64_t)INTEGER(v)[q] : (int64_t)REAL(v)[q];
  if (len) *len = m;
  return out;
}

static const char* const kCvSweepArgs[20] = {
    "row", "col", "value", "code", "n", "named", "preserve_order", "ndim", "k0", "cooling_rate", "c_repulsion", "picks",
    "picks_offset", "unit_draws", "draws_offset", "seeds", "n_iter", "relative_epsilon", "convergence_counter",
    "convergence_check_freq"};

/* The 20 arguments of a sweep in kCvSweepArgs order: by name when the list carries names (every one of the 20 must
 * be there, extra or misspelt names are errors), by position otherwise. */
static SEXP cv_sweep_args(SEXP a) {
  if (!Rf_isNewList(a) || Rf_length(a) != 20) Rf_error("cv_sweep: one list of 20 elements expected");
  SEXP names = Rf_getAttrib(a, R_NamesSymbol);
  if (names == R_NilValue) return a;
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 20));
  int found[20];
  memset(found, 0, sizeof found);
  for (int q = 0; q < 20; ++q) {
    const char* nm = CHAR(STRING_ELT(names, q));
    int at = -1;
    for (int j = 0; j < 20; ++j)
      if (strcmp(nm, kCvSweepArgs[j]) == 0) at = j;
    if (at < 0 || found[at]) {
      UNPROTECT(1);
      Rf_error("cv_sweep: unknown or repeated argument name '%s'", nm);
    }
    found[at] = 1;
    SET_VECTOR_ELT(out, at, VECTOR_ELT(a, q));
  }
  UNPROTECT(1);     /* (nothing below allocates R memory before the elements have been read) */
  return out;
}

SEXP _topolow_cv_sweep(SEXP arg) {
  SEXP a = PROTECT(cv_sweep_args(arg));
#define CV_SWEEP_FAIL(msg) do { UNPROTECT(1); Rf_error(msg); } while (0)
  SEXP row = VECTOR_ELT(a, 0), col = VECTOR_ELT(a, 1), value = VECTOR_ELT(a, 2), code = VECTOR_ELT(a, 3);
  if (!Rf_isInteger(row) || !Rf_isInteger(col) || !Rf_isReal(value) || !Rf_isInteger(code))
    CV_SWEEP_FAIL("row, col, code must be integer and value double");
  const int n = Rf_asInteger(VECTOR_ELT(a, 4));
  const int64_t m = (int64_t)XLENGTH(row);
  if (n < 1 || XLENGTH(col) != m || XLENGTH(value) != m || XLENGTH(code) != m)
    CV_SWEEP_FAIL("cell columns must have one length and n must be positive");
  SEXP ndim = VECTOR_ELT(a, 7), k0 = VECTOR_ELT(a, 8), cool = VECTOR_ELT(a, 9), crep = VECTOR_ELT(a, 10);
  const int F = Rf_length(ndim);
  if (!Rf_isInteger(ndim) || !Rf_isReal(k0) || !Rf_isReal(cool) || !Rf_isReal(crep) || Rf_length(k0) != F ||
      Rf_length(cool) != F || Rf_length(crep) != F || !Rf_isReal(VECTOR_ELT(a, 13)))
    CV_SWEEP_FAIL("per-fold parameters: integer ndim, double k0 / cooling_rate / c_repulsion / unit_draws of one length");
  int64_t np = 0, npo = 0, ndo = 0, ns = 0;
  int64_t* picks = as_i64(VECTOR_ELT(a, 11), &np);
  int64_t* p_off = as_i64(VECTOR_ELT(a, 12), &npo);
  int64_t* d_off = as_i64(VECTOR_ELT(a, 14), &ndo);
  int64_t* seeds_i = as_i64(VECTOR_ELT(a, 15), &ns);
  if (npo != F + 1 || ndo != F + 1 || ns != F || p_off[F] != np || d_off[F] != (int64_t)XLENGTH(VECTOR_ELT(a, 13)))
    CV_SWEEP_FAIL("offsets must have one entry per fold plus one and end at the lengths of picks / unit_draws");
  uint64_t* seeds = (uint64_t*)R_alloc(F > 0 ? (size_t)F : 1, sizeof(uint64_t));
  for (int f = 0; f < F; ++f) seeds[f] = (uint64_t)seeds_i[f];
  int64_t* pos_of = (int64_t*)R_alloc((size_t)n * n, sizeof(int64_t));
  int64_t* by_row = (int64_t*)R_alloc(m > 0 ? (size_t)m : 1, sizeof(int64_t));
  int64_t* row_ptr = (int64_t*)R_alloc((size_t)n + 1, sizeof(int64_t));
  if (topolow_cell_list_index(n, m, INTEGER(row), INTEGER(col), pos_of, by_row, row_ptr) != TOPOLOW_OK)
    CV_SWEEP_FAIL("cell list: row / col out of range");
  topolow_cell_list cells;
  memset(&cells, 0, sizeof cells);
  cells.n = n; cells.n_cells = m;
  cells.row = INTEGER(row); cells.col = INTEGER(col); cells.value = REAL(value); cells.code = INTEGER(code);
  cells.pos_of = pos_of; cells.by_row = by_row; cells.row_ptr = row_ptr;
  static const char* const names[] = {"holdout_sum_abs", "holdout_count", "iterations", "converged", "error_code",
                                      "device_seconds"};
  SEXP out = PROTECT(named_list(6, names));
  SEXP hs = PROTECT(Rf_allocVector(REALSXP, F)), hc = PROTECT(Rf_allocVector(REALSXP, F));
  SEXP it = PROTECT(Rf_allocVector(INTSXP, F)), cv = PROTECT(Rf_allocVector(INTSXP, F)), ec = PROTECT(Rf_allocVector(INTSXP, F));
  int64_t* hcount = (int64_t*)R_alloc(F > 0 ? (size_t)F : 1, sizeof(int64_t));
  double secs = 0.0;
  char err[512];
  err[0] = 0;
  const int precision = opt_choice("topolow.precision", "f32", TOPOLOW_PRECISION_F32, "f64", TOPOLOW_PRECISION_F64,
                                   TOPOLOW_PRECISION_F64);
  const int rc = topolow_cv_sweep(&cells, Rf_asLogical(VECTOR_ELT(a, 5)), Rf_asLogical(VECTOR_ELT(a, 6)), F, INTEGER(ndim),
                                  REAL(k0), REAL(cool), REAL(crep), picks, p_off, REAL(VECTOR_ELT(a, 13)), d_off, seeds,
                                  Rf_asInteger(VECTOR_ELT(a, 16)), Rf_asReal(VECTOR_ELT(a, 17)),
                                  Rf_asInteger(VECTOR_ELT(a, 18)), Rf_asInteger(VECTOR_ELT(a, 19)), precision,
                                  opt_int("topolow.device", -1), REAL(hs), hcount, INTEGER(it), INTEGER(cv), INTEGER(ec),
                                  &secs, err, sizeof err);
  if (rc != TOPOLOW_OK) {
    UNPROTECT(7);
    Rf_error("%s", err[0] ? err : "topolow_cv_sweep failed");
  }
  for (int f = 0; f < F; ++f) REAL(hc)[f] = (double)hcount[f];
  SET_VECTOR_ELT(out, 0, hs); SET_VECTOR_ELT(out, 1, hc); SET_VECTOR_ELT(out, 2, it); SET_VECTOR_ELT(out, 3, cv);
  SET_VECTOR_ELT(out, 4, ec); SET_VECTOR_ELT(out, 5, Rf_ScalarReal(secs));
  UNPROTECT(7);
#undef CV_SWEEP_FAIL
  return out;
}

/* Optional: as.matrix(dist(positions)) on the GPU (reference R/core.R:474). */
SEXP _topolow_est_distances(SEXP positionsSEXP) {
  if (!Rf_isReal(positionsSEXP) || !Rf_isMatrix(positionsSEXP))
    Rf_error("positions must be a numeric matrix");
  const int n = Rf_nrows(positionsSEXP), ndim = Rf_ncols(positionsSEXP);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, n));
  char err[512];
  err[0] = '\0';
  const int rc = topolow_est_distances(REAL(positionsSEXP), n, ndim, REAL(out),
                                       opt_int("topolow.device", -1), err, sizeof err);
  UNPROTECT(1);
  if (rc != TOPOLOW_OK) Rf_error("%s", err[0] ? err : "libtopolow_relax failed");
  return out;
}

/* Columns first..last (1-based, inclusive) of the same matrix: an n x (last - first + 1) block, for
 * problems whose full n x n result should be streamed rather than held (BASELINE config 4). */
SEXP _topolow_est_distances_cols(SEXP positionsSEXP, SEXP firstSEXP, SEXP lastSEXP) {
  if (!Rf_isReal(positionsSEXP) || !Rf_isMatrix(positionsSEXP))
    Rf_error("positions must be a numeric matrix");
  const int n = Rf_nrows(positionsSEXP), ndim = Rf_ncols(positionsSEXP);
  const int first = Rf_asInteger(firstSEXP), last = Rf_asInteger(lastSEXP);
  if (first < 1 || last > n || first > last) Rf_error("columns out of range");
  /* the library writes rows first..last of the symmetric matrix row-major = these columns column-major */
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, last - first + 1));
  char err[512];
  err[0] = '\0';
  const int rc = topolow_est_distances_rows(REAL(positionsSEXP), n, ndim, first - 1, last, REAL(out),
                                            opt_int("topolow.device", -1), err, sizeof err);
  UNPROTECT(1);
  if (rc != TOPOLOW_OK) Rf_error("%s", err[0] ? err : "libtopolow_relax failed");
  return out;
}

static const R_CallMethodDef CallEntries[] = {
    {"_topolow_optimize_layout_exact_cpp", (DL_FUNC)&_topolow_optimize_layout_exact_cpp, 16},
    {"_topolow_optimize_layout_exact_batch", (DL_FUNC)&_topolow_optimize_layout_exact_batch, 1},
    {"_topolow_cv_fold", (DL_FUNC)&_topolow_cv_fold, 8},
    {"_topolow_cv_sweep", (DL_FUNC)&_topolow_cv_sweep, 1},
    {"_topolow_est_distances", (DL_FUNC)&_topolow_est_distances, 1},
    {"_topolow_est_distances_cols", (DL_FUNC)&_topolow_est_distances_cols, 3},
    {NULL, NULL, 0}};

void R_init_topolow(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
