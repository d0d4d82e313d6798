/* tests/fake_r/fake_r.c -- implementation of the R C API test double (Rinternals.h here) and the
 * harness that plays R's part of `.Call("_topolow_optimize_layout_exact_cpp", ...)`:
 *   harness <input file>
 * The input file is text: first the options lines "opt <name> int|real|str <value>", then
 *   n ndim n_edges n_iter convergence_window convergence_check_freq verbose
 *   k0 cooling_rate c_repulsion relative_epsilon
 * followed by the arrays in the order R/core.R:439-456 passes them (initial_positions, dissimilarity
 * matrix ("Inf" allowed), threshold matrix, degrees, edge_i, edge_j, edge_dist, edge_thresh), all
 * column-major.  The result list (or the R error) is printed as one JSON object.
 * Test infrastructure only (tests/test_r_shim.py). */
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "R_ext/Rdynload.h"
#include "Rinternals.h"

static struct fake_sexp nil_obj = {NILSXP, 0, 0, 0, NULL, NULL};
static struct fake_sexp names_sym = {CHARSXP, 5, 0, 0, (void*)"names", NULL};
static struct fake_sexp global_env = {NILSXP, 0, 0, 0, NULL, NULL};
static struct fake_sexp unbound = {NILSXP, 0, 0, 0, NULL, NULL};
static struct fake_sexp na_string = {CHARSXP, 2, 0, 0, (void*)"NA", NULL};
SEXP R_NilValue = &nil_obj;
SEXP R_NamesSymbol = &names_sym;
SEXP R_GlobalEnv = &global_env;
SEXP R_UnboundValue = &unbound;
SEXP R_NaString = &na_string;
static SEXP random_seed = NULL;
static int unif_calls = 0;
static char printed[1 << 16];
static size_t printed_len = 0;

static jmp_buf error_jmp;
static char error_msg[1024];
static int protect_depth = 0;
int fake_r_interrupt_after = 0;
static int interrupt_calls = 0;
static int interrupted = 0;

SEXP Rf_allocVector(int type, R_xlen_t n) {
  SEXP x = (SEXP)calloc(1, sizeof *x);
  x->type = type;
  x->length = n;
  size_t el = type == REALSXP ? sizeof(double) : (type == STRSXP || type == VECSXP) ? sizeof(SEXP) : sizeof(int);
  x->data = calloc(n ? (size_t)n : 1, el);
  if (type == STRSXP || type == VECSXP)
    for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)x->data)[i] = R_NilValue;
  return x;
}
SEXP Rf_allocMatrix(int type, int nrow, int ncol) {
  SEXP x = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
  x->nrow = nrow;
  x->ncol = ncol;
  return x;
}
SEXP Rf_mkChar(const char* s) {
  SEXP x = (SEXP)calloc(1, sizeof *x);
  x->type = CHARSXP;
  x->length = (R_xlen_t)strlen(s);
  x->data = strdup(s);
  return x;
}
SEXP Rf_mkString(const char* s) {
  SEXP x = Rf_allocVector(STRSXP, 1);
  SET_STRING_ELT(x, 0, Rf_mkChar(s));
  return x;
}
SEXP Rf_ScalarLogical(int v) { SEXP x = Rf_allocVector(LGLSXP, 1); LOGICAL(x)[0] = v; return x; }
SEXP Rf_ScalarInteger(int v) { SEXP x = Rf_allocVector(INTSXP, 1); INTEGER(x)[0] = v; return x; }
SEXP Rf_ScalarReal(double v) { SEXP x = Rf_allocVector(REALSXP, 1); REAL(x)[0] = v; return x; }
SEXP Rf_install(const char* name) { return Rf_mkChar(name); }
SEXP Rf_ScalarString(SEXP ch) { SEXP x = Rf_allocVector(STRSXP, 1); SET_STRING_ELT(x, 0, ch); return x; }
SEXP Rf_findVarInFrame(SEXP env, SEXP sym) {
  if (env == R_GlobalEnv && strcmp(CHAR(sym), ".Random.seed") == 0 && random_seed != NULL) return random_seed;
  return R_UnboundValue;
}
char* R_alloc(size_t n, int size) { return (char*)calloc(n ? n : 1, (size_t)size); }
void Rprintf(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  const int w = vsnprintf(printed + printed_len, sizeof printed - printed_len, fmt, ap);
  va_end(ap);
  if (w > 0) printed_len += (size_t)w < sizeof printed - printed_len ? (size_t)w : sizeof printed - printed_len - 1;
}
void fake_r_set_random_seed(int present) {
  if (!present) { random_seed = NULL; return; }
  random_seed = Rf_allocVector(INTSXP, 626);   /* Mersenne-Twister: kind, mti, 624 words */
  unsigned long long z = 0x1234abcdull * (unsigned)present;
  for (int i = 0; i < 626; ++i) { z = z * 6364136223846793005ull + 1442695040888963407ull; INTEGER(random_seed)[i] = (int)(z >> 33); }
}
int fake_r_unif_rand_calls(void) { return unif_calls; }
unsigned long long fake_r_random_seed_hash(void) {
  unsigned long long h = 1469598103934665603ull;
  if (random_seed) for (int i = 0; i < 626; ++i) h = (h ^ (unsigned)INTEGER(random_seed)[i]) * 1099511628211ull;
  return h;
}

static struct { char name[64]; SEXP value; } options[32];
static int n_options = 0;
static void set_option(const char* name, SEXP v) {
  strncpy(options[n_options].name, name, 63);
  options[n_options++].value = v;
}
void fake_r_set_option_int(const char* name, int v) { set_option(name, Rf_ScalarInteger(v)); }
void fake_r_set_option_real(const char* name, double v) { set_option(name, Rf_ScalarReal(v)); }
void fake_r_set_option_string(const char* name, const char* v) { set_option(name, Rf_mkString(v)); }
SEXP Rf_GetOption1(SEXP tag) {
  for (int i = 0; i < n_options; ++i)
    if (strcmp(options[i].name, CHAR(tag)) == 0) return options[i].value;
  return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP what, SEXP value) { if (what == R_NamesSymbol) x->names = value; return value; }
SEXP Rf_getAttrib(SEXP x, SEXP what) { return (what == R_NamesSymbol && x->names) ? x->names : R_NilValue; }

double* REAL(SEXP x) { return (double*)x->data; }
int* INTEGER(SEXP x) { return (int*)x->data; }
int* LOGICAL(SEXP x) { return (int*)x->data; }
const char* CHAR(SEXP x) { return (const char*)x->data; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)x->data)[i]; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)x->data)[i]; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP*)x->data)[i] = v; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP*)x->data)[i] = v; return v; }

int Rf_length(SEXP x) { return (int)x->length; }
R_xlen_t XLENGTH(SEXP x) { return x->length; }
int Rf_nrows(SEXP x) { return x->ncol > 0 ? x->nrow : (int)x->length; }
int Rf_ncols(SEXP x) { return x->ncol > 0 ? x->ncol : 1; }
Rboolean Rf_isReal(SEXP x) { return x->type == REALSXP; }
Rboolean Rf_isInteger(SEXP x) { return x->type == INTSXP; }
Rboolean Rf_isString(SEXP x) { return x->type == STRSXP; }
Rboolean Rf_isNewList(SEXP x) { return x->type == VECSXP; }
Rboolean Rf_isMatrix(SEXP x) { return x->ncol > 0; }
int Rf_asInteger(SEXP x) { return x->type == REALSXP ? (int)REAL(x)[0] : INTEGER(x)[0]; }
double Rf_asReal(SEXP x) { return x->type == REALSXP ? REAL(x)[0] : (double)INTEGER(x)[0]; }
int Rf_asLogical(SEXP x) { return x->type == REALSXP ? REAL(x)[0] != 0.0 : INTEGER(x)[0] != 0; }

SEXP Rf_protect(SEXP x) { ++protect_depth; return x; }
void Rf_unprotect(int n) { protect_depth -= n; }
int fake_r_protect_depth(void) { return protect_depth; }

void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_msg, sizeof error_msg, fmt, ap);
  va_end(ap);
  longjmp(error_jmp, 1);
}
void Rf_onintr(void) { interrupted = 1; }
/* an "interrupt" is a long jump out of R_CheckUserInterrupt, caught by R_ToplevelExec */
static jmp_buf toplevel_jmp;
static int in_toplevel = 0;
void R_CheckUserInterrupt(void) {
  ++interrupt_calls;
  if (fake_r_interrupt_after > 0 && interrupt_calls >= fake_r_interrupt_after) {
    if (in_toplevel) longjmp(toplevel_jmp, 1);
    Rf_error("interrupt outside R_ToplevelExec");
  }
}
Rboolean R_ToplevelExec(void (*fun)(void*), void* data) {
  in_toplevel = 1;
  if (setjmp(toplevel_jmp) != 0) { in_toplevel = 0; return FALSE; }
  fun(data);
  in_toplevel = 0;
  return TRUE;
}
static int rng_open = 0;
static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
void GetRNGstate(void) { ++rng_open; }
void PutRNGstate(void) { --rng_open; }
double unif_rand(void) {
  if (rng_open != 1) Rf_error("unif_rand outside GetRNGstate/PutRNGstate");
  ++unif_calls;
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}

static DllInfo dll;
int R_registerRoutines(DllInfo* info, const void* c, const R_CallMethodDef* call, const void* f, const void* e) {
  (void)c; (void)f; (void)e;
  info->call_routines = call;
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value) { info->use_dynamic_symbols = value; return TRUE; }

/* ---------------------------------------------------------------------------------------------
 * harness
 * ------------------------------------------------------------------------------------------- */
void R_init_topolow(DllInfo* dll);
typedef SEXP (*call16)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

static double read_num(FILE* f) {
  char tok[64];
  if (fscanf(f, "%63s", tok) != 1) { fprintf(stderr, "short input\n"); exit(2); }
  if (strcmp(tok, "Inf") == 0) return INFINITY;
  return strtod(tok, NULL);
}
static SEXP read_real(FILE* f, int nrow, int ncol) {
  SEXP x = ncol > 0 ? Rf_allocMatrix(REALSXP, nrow, ncol) : Rf_allocVector(REALSXP, nrow);
  for (R_xlen_t i = 0; i < x->length; ++i) REAL(x)[i] = read_num(f);
  return x;
}
static SEXP read_int(FILE* f, int nrow, int ncol) {
  SEXP x = ncol > 0 ? Rf_allocMatrix(INTSXP, nrow, ncol) : Rf_allocVector(INTSXP, nrow);
  for (R_xlen_t i = 0; i < x->length; ++i) INTEGER(x)[i] = (int)read_num(f);
  return x;
}

static void print_json_string(const char* t) {
  putchar('"');
  for (; *t; ++t) {
    if (*t == '"' || *t == '\\') { putchar('\\'); putchar(*t); }
    else if (*t == '\n') fputs("\\n", stdout);
    else putchar(*t);
  }
  putchar('"');
}
static void print_vec(SEXP v) {
  printf("[");
  if (v != R_NilValue)
    for (R_xlen_t i = 0; i < XLENGTH(v); ++i) {
      if (v->type == REALSXP) printf("%s%.17g", i ? ", " : "", REAL(v)[i]);
      else printf("%s%d", i ? ", " : "", INTEGER(v)[i]);
    }
  printf("]");
}
static void print_tail(void) {
  printf("\"protect_depth\": %d, \"interrupt_polls\": %d, \"unif_rand_calls\": %d, \"random_seed_hash\": \"%llu\", "
         "\"printed\": ", protect_depth, interrupt_calls, unif_calls, fake_r_random_seed_hash());
  print_json_string(printed);
  printf("}\n");
}
static DL_FUNC find_routine(const char* name, int arity) {
  for (const R_CallMethodDef* m = dll.call_routines; m && m->name; ++m)
    if (strcmp(m->name, name) == 0 && m->numArgs == arity) return m->fun;
  return NULL;
}

/* Input: option lines, then an optional "mode <single|batch K H|cvfold>" line (default single).
 *   single : the 16-argument call (header comment).
 *   batch  : the same input, sent K times through _topolow_optimize_layout_exact_batch as a list of
 *            argument lists; calls with an odd index pass NULL matrices (the edge list is the matrix);
 *            the first H edges are scored as hold-out pairs.
 *   cvfold : "n m n_picks preserve_order named" then row, col, value, code, picks.
 *   cvsweep: the 20-element list of _topolow_cv_sweep (see there), flat. */
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 2;
  char word[64];
  long at = ftell(f);
  char mode[16] = "single";
  int batch_k = 0, batch_h = 0;
  while (fscanf(f, "%63s", word) == 1 && (strcmp(word, "opt") == 0 || strcmp(word, "mode") == 0)) {
    if (strcmp(word, "mode") == 0) {
      if (fscanf(f, "%15s", mode) != 1) return 2;
      if (strcmp(mode, "batch") == 0 && fscanf(f, "%d %d", &batch_k, &batch_h) != 2) return 2;
      at = ftell(f);
      continue;
    }
    char name[64], kind[16], val[64];
    if (fscanf(f, "%63s %15s %63s", name, kind, val) != 3) return 2;
    if (strcmp(name, "fake.random_seed") == 0) fake_r_set_random_seed(atoi(val));
    else if (strcmp(name, "fake.interrupt_after") == 0) fake_r_interrupt_after = atoi(val);
    else if (strcmp(kind, "int") == 0) fake_r_set_option_int(name, atoi(val));
    else if (strcmp(kind, "real") == 0) fake_r_set_option_real(name, atof(val));
    else if (strcmp(kind, "ints") == 0) {   /* comma-separated integer vector */
      int vals[64], nv = 0;
      for (char* tok = strtok(val, ","); tok && nv < 64; tok = strtok(NULL, ",")) vals[nv++] = atoi(tok);
      SEXP v = Rf_allocVector(INTSXP, nv);
      memcpy(INTEGER(v), vals, sizeof(int) * (size_t)nv);
      set_option(name, v);
    }
    else fake_r_set_option_string(name, val);
    at = ftell(f);
  }
  fseek(f, at, SEEK_SET);
  /* what useDynLib(topolow, .registration = TRUE) does: init, then look the routines up by name */
  R_init_topolow(&dll);
  if (dll.use_dynamic_symbols != FALSE) { printf("{\"registration\": \"bad\"}\n"); return 1; }

  if (strcmp(mode, "cvfold") == 0) {
    typedef SEXP (*call8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
    call8 fn = (call8)find_routine("_topolow_cv_fold", 8);
    if (!fn) { printf("{\"registration\": \"bad\"}\n"); return 1; }
    const int n = (int)read_num(f), m = (int)read_num(f), np = (int)read_num(f);
    const int preserve = (int)read_num(f), named = (int)read_num(f);
    SEXP row = read_int(f, m, 0), col = read_int(f, m, 0), val = read_real(f, m, 0), code = read_int(f, m, 0);
    SEXP picks = read_real(f, np, 0);
    fclose(f);
    if (setjmp(error_jmp) != 0) {
      printf("{\"error\": \"%s\", ", error_msg);
      print_tail();
      return 0;
    }
    SEXP out = fn(row, col, val, code, Rf_ScalarInteger(n), picks, Rf_ScalarLogical(preserve), Rf_ScalarLogical(named));
    SEXP names = Rf_getAttrib(out, R_NamesSymbol);
    printf("{");
    for (int i = 0; i < Rf_length(out); ++i) {
      printf("\"%s\": ", CHAR(STRING_ELT(names, i)));
      print_vec(VECTOR_ELT(out, i));
      printf(", ");
    }
    print_tail();
    return 0;
  }

  if (strcmp(mode, "cvsweep") == 0 || strcmp(mode, "cvsweep_named") == 0) {
    /* "n m F preserve named n_iter window freq eps n_picks n_draws" then row, col, value, code (m each), ndim, k0,
     * cooling_rate, c_repulsion (F each), picks, picks_offset (F + 1), unit_draws, draws_offset (F + 1), seeds (F) */
    typedef SEXP (*call1)(SEXP);
    call1 fn = (call1)find_routine("_topolow_cv_sweep", 1);
    if (!fn) { printf("{\"registration\": \"bad\"}\n"); return 1; }
    const int n = (int)read_num(f), m = (int)read_num(f), F = (int)read_num(f);
    const int preserve = (int)read_num(f), named = (int)read_num(f), n_iter = (int)read_num(f);
    const int window = (int)read_num(f), freq = (int)read_num(f);
    const double eps = read_num(f);
    const int np = (int)read_num(f), nd = (int)read_num(f);
    SEXP a = Rf_allocVector(VECSXP, 20);
    SET_VECTOR_ELT(a, 0, read_int(f, m, 0)); SET_VECTOR_ELT(a, 1, read_int(f, m, 0));
    SET_VECTOR_ELT(a, 2, read_real(f, m, 0)); SET_VECTOR_ELT(a, 3, read_int(f, m, 0));
    SET_VECTOR_ELT(a, 4, Rf_ScalarInteger(n)); SET_VECTOR_ELT(a, 5, Rf_ScalarLogical(named));
    SET_VECTOR_ELT(a, 6, Rf_ScalarLogical(preserve));
    SET_VECTOR_ELT(a, 7, read_int(f, F, 0)); SET_VECTOR_ELT(a, 8, read_real(f, F, 0));
    SET_VECTOR_ELT(a, 9, read_real(f, F, 0)); SET_VECTOR_ELT(a, 10, read_real(f, F, 0));
    SET_VECTOR_ELT(a, 11, read_real(f, np, 0)); SET_VECTOR_ELT(a, 12, read_real(f, F + 1, 0));
    SET_VECTOR_ELT(a, 13, read_real(f, nd, 0)); SET_VECTOR_ELT(a, 14, read_real(f, F + 1, 0));
    SET_VECTOR_ELT(a, 15, read_real(f, F, 0));
    SET_VECTOR_ELT(a, 16, Rf_ScalarInteger(n_iter)); SET_VECTOR_ELT(a, 17, Rf_ScalarReal(eps));
    SET_VECTOR_ELT(a, 18, Rf_ScalarInteger(window)); SET_VECTOR_ELT(a, 19, Rf_ScalarInteger(freq));
    fclose(f);
    if (strcmp(mode, "cvsweep_named") == 0) {
      /* the same arguments as a NAMED list in reverse order: the entry must find them by name */
      static const char* const arg_names[20] = {
          "row", "col", "value", "code", "n", "named", "preserve_order", "ndim", "k0", "cooling_rate", "c_repulsion",
          "picks", "picks_offset", "unit_draws", "draws_offset", "seeds", "n_iter", "relative_epsilon",
          "convergence_counter", "convergence_check_freq"};
      SEXP b = Rf_allocVector(VECSXP, 20), nm = Rf_allocVector(STRSXP, 20);
      for (int q = 0; q < 20; ++q) {
        SET_VECTOR_ELT(b, 19 - q, VECTOR_ELT(a, q));
        SET_STRING_ELT(nm, 19 - q, Rf_mkChar(arg_names[q]));
      }
      Rf_setAttrib(b, R_NamesSymbol, nm);
      a = b;
    }
    if (setjmp(error_jmp) != 0) {
      printf("{\"error\": \"%s\", ", error_msg);
      print_tail();
      return 0;
    }
    SEXP out = fn(a);
    SEXP names = Rf_getAttrib(out, R_NamesSymbol);
    printf("{");
    for (int i = 0; i < Rf_length(out); ++i) {
      printf("\"%s\": ", CHAR(STRING_ELT(names, i)));
      print_vec(VECTOR_ELT(out, i));
      printf(", ");
    }
    print_tail();
    return 0;
  }

  const int n = (int)read_num(f), ndim = (int)read_num(f), n_edges = (int)read_num(f), n_iter = (int)read_num(f);
  const int window = (int)read_num(f), freq = (int)read_num(f), verbose = (int)read_num(f);
  const double k0 = read_num(f), cool = read_num(f), c_rep = read_num(f), eps = read_num(f);
  SEXP pos = read_real(f, n, ndim), D = read_real(f, n, n), T = read_int(f, n, n), deg = read_int(f, n, 0);
  SEXP ei = read_int(f, n_edges, 0), ej = read_int(f, n_edges, 0), ed = read_real(f, n_edges, 0);
  SEXP et = read_int(f, n_edges, 0);
  fclose(f);
  SEXP iter = Rf_ScalarInteger(n_iter);
  SEXP sk0 = Rf_ScalarReal(k0), scool = Rf_ScalarReal(cool), scr = Rf_ScalarReal(c_rep), seps = Rf_ScalarReal(eps);

  if (strcmp(mode, "batch") == 0) {
    typedef SEXP (*call1)(SEXP);
    call1 fn = (call1)find_routine("_topolow_optimize_layout_exact_batch", 1);
    if (!fn) { printf("{\"registration\": \"bad\"}\n"); return 1; }
    SEXP calls = Rf_allocVector(VECSXP, batch_k);
    SEXP hi = Rf_allocVector(INTSXP, batch_h), hj = Rf_allocVector(INTSXP, batch_h), ht = Rf_allocVector(REALSXP, batch_h);
    for (int q = 0; q < batch_h; ++q) { INTEGER(hi)[q] = INTEGER(ei)[q]; INTEGER(hj)[q] = INTEGER(ej)[q]; REAL(ht)[q] = REAL(ed)[q]; }
    for (int b = 0; b < batch_k; ++b) {
      SEXP a = Rf_allocVector(VECSXP, 19);
      SEXP parts[19] = {pos, (b & 1) ? R_NilValue : D, (b & 1) ? R_NilValue : T, deg, ei, ej, ed, et, iter, sk0, scool, scr,
                        seps, Rf_ScalarInteger(window), Rf_ScalarInteger(freq), Rf_ScalarLogical(verbose), hi, hj, ht};
      for (int q = 0; q < 19; ++q) SET_VECTOR_ELT(a, q, parts[q]);
      SET_VECTOR_ELT(calls, b, a);
    }
    if (setjmp(error_jmp) != 0) {
      printf("{\"error\": \"%s\", ", error_msg);
      print_tail();
      return 0;
    }
    SEXP out = fn(calls);
    printf("{\"results\": [");
    for (int b = 0; b < Rf_length(out); ++b) {
      SEXP r = VECTOR_ELT(out, b);
      printf("%s{\"positions\": ", b ? ", " : "");
      print_vec(VECTOR_ELT(r, 0));
      SEXP e = VECTOR_ELT(r, 6);
      printf(", \"converged\": %d, \"iterations\": %d, \"final_mae\": %.17g, \"final_k\": %.17g, \"iterations_run\": %d, "
             "\"error\": ", LOGICAL(VECTOR_ELT(r, 1))[0], INTEGER(VECTOR_ELT(r, 2))[0], REAL(VECTOR_ELT(r, 3))[0],
             REAL(VECTOR_ELT(r, 4))[0], INTEGER(VECTOR_ELT(r, 5))[0]);
      if (STRING_ELT(e, 0) == R_NaString) printf("null"); else print_json_string(CHAR(STRING_ELT(e, 0)));
      printf(", \"holdout_sum_abs\": %.17g, \"holdout_count\": %.17g, \"n_names\": %d}", REAL(VECTOR_ELT(r, 7))[0],
             REAL(VECTOR_ELT(r, 8))[0], Rf_length(Rf_getAttrib(r, R_NamesSymbol)));
    }
    printf("], ");
    print_tail();
    return 0;
  }

  call16 fn = (call16)find_routine("_topolow_optimize_layout_exact_cpp", 16);
  if (!fn) {
    printf("{\"registration\": \"bad\"}\n");
    return 1;
  }
  if (setjmp(error_jmp) != 0) {
    printf("{\"error\": \"%s\", \"interrupted\": %d, ", error_msg, interrupted);
    print_tail();
    return 0;
  }
  SEXP out = fn(pos, D, T, deg, ei, ej, ed, et, iter, sk0, scool, scr, seps, Rf_ScalarInteger(window),
                Rf_ScalarInteger(freq), Rf_ScalarLogical(verbose));
  printf("{\"names\": [");
  SEXP names = Rf_getAttrib(out, R_NamesSymbol);
  for (int i = 0; i < Rf_length(names); ++i) printf("%s\"%s\"", i ? ", " : "", CHAR(STRING_ELT(names, i)));
  SEXP p = VECTOR_ELT(out, 0);
  printf("], \"dim\": [%d, %d], \"positions\": ", Rf_nrows(p), Rf_ncols(p));
  print_vec(p);
  printf(", \"converged\": %d, \"iterations\": %d, \"final_mae\": %.17g, \"final_k\": %.17g, "
         "\"types\": [%d, %d, %d, %d, %d], ",
         LOGICAL(VECTOR_ELT(out, 1))[0], INTEGER(VECTOR_ELT(out, 2))[0], REAL(VECTOR_ELT(out, 3))[0],
         REAL(VECTOR_ELT(out, 4))[0], VECTOR_ELT(out, 0)->type, VECTOR_ELT(out, 1)->type,
         VECTOR_ELT(out, 2)->type, VECTOR_ELT(out, 3)->type, VECTOR_ELT(out, 4)->type);
  print_tail();
  return 0;
}
