/* topolow_amd/r/topolow_shim.c -- R `.Call` shim over libtopolow_relax.so.
 *
 * Drop-in for the reference's generated glue (src/RcppExports.cpp:16-49 of omid-arhami/topolow
 * v2.1.0) and, through it, for src/optimization.cpp: it exports the SAME native symbol
 *     _topolow_optimize_layout_exact_cpp   (16 SEXP arguments, registered with arity 16)
 * and the same R_init_topolow, so the reference's R code (R/RcppExports.R:4-6, R/core.R:439-456)
 * runs unchanged -- `euclidean_embedding()` keeps its signature and its returned object.
 *
 * No logic lives here: unmarshal, call topolow_optimize_layout_exact(), marshal, and turn error
 * codes into R errors AFTER every native resource has been released (Rf_error longjmps).
 * R is absent from the build image, so this file is compiled only where R is installed:
 *     R CMD SHLIB topolow_shim.c -L<dir> -ltopolow_relax -I<repo>/include
 * Backend options travel through R options(), never through the function signatures:
 *     options(topolow.seed = 1L, topolow.schedule = "auto"|"slab"|"gs",
 *             topolow.precision = "auto"|"f32"|"f64", topolow.device = 0L)
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "topolow_relax.h"

/* R_CheckUserInterrupt() longjmps; probing it through R_ToplevelExec keeps control here so the
 * library can release its device memory before the interrupt is re-raised. */
static void probe_interrupt(void* dummy) { (void)dummy; R_CheckUserInterrupt(); }
static int32_t interrupt_pending(void* user) {
  (void)user;
  return R_ToplevelExec(probe_interrupt, NULL) == FALSE;
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
  topolow_default_options(&opt);
  {
    SEXP seed = Rf_GetOption1(Rf_install("topolow.seed"));
    if (seed != R_NilValue && Rf_length(seed) >= 1) {
      opt.seed = (uint64_t)Rf_asReal(seed);
    } else {
      /* the reference seeds its shuffle from std::random_device (src/optimization.cpp:153-154);
         draw from R's RNG instead so set.seed() makes the whole run reproducible */
      GetRNGstate();
      opt.seed = (uint64_t)(unif_rand() * 9007199254740992.0);
      PutRNGstate();
    }
  }
  opt.schedule = opt_choice("topolow.schedule", "slab", TOPOLOW_SCHEDULE_SLAB, "gs",
                            TOPOLOW_SCHEDULE_GS, TOPOLOW_SCHEDULE_AUTO);
  opt.precision = opt_choice("topolow.precision", "f32", TOPOLOW_PRECISION_F32, "f64",
                             TOPOLOW_PRECISION_F64, TOPOLOW_PRECISION_AUTO);
  opt.device = opt_int("topolow.device", -1);
  opt.slab_stages = opt_int("topolow.slab_stages", 0);
  opt.gs_max_n = opt_int("topolow.gs_max_n", 0);
  opt.interrupt_cb = interrupt_pending;   /* polled every 50 iterations, reference :364 */

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
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 5));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, 5));
  SET_VECTOR_ELT(out, 0, positions);
  SET_VECTOR_ELT(out, 1, Rf_ScalarLogical(converged));
  SET_VECTOR_ELT(out, 2, Rf_ScalarInteger(iterations));
  SET_VECTOR_ELT(out, 3, Rf_ScalarReal(final_mae));
  SET_VECTOR_ELT(out, 4, Rf_ScalarReal(final_k));
  SET_STRING_ELT(names, 0, Rf_mkChar("positions"));
  SET_STRING_ELT(names, 1, Rf_mkChar("converged"));
  SET_STRING_ELT(names, 2, Rf_mkChar("iterations"));
  SET_STRING_ELT(names, 3, Rf_mkChar("final_mae"));
  SET_STRING_ELT(names, 4, Rf_mkChar("final_k"));
  Rf_setAttrib(out, R_NamesSymbol, names);
  UNPROTECT(3);
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

static const R_CallMethodDef CallEntries[] = {
    {"_topolow_optimize_layout_exact_cpp", (DL_FUNC)&_topolow_optimize_layout_exact_cpp, 16},
    {"_topolow_est_distances", (DL_FUNC)&_topolow_est_distances, 1},
    {NULL, NULL, 0}};

void R_init_topolow(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
