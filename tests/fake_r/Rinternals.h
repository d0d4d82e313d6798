/* tests/fake_r/Rinternals.h -- a TEST DOUBLE of the slice of R's C API that
 * topolow_amd/r/topolow_shim.c touches, so the shim can be compiled and driven end to end in an image
 * without R (tests/test_r_shim.py).  It is not R and implements nothing beyond what the harness
 * needs: vectors of the five types the shim sees, a names attribute, options as a lookup table, and
 * Rf_error as a longjmp to the harness.  Test infrastructure only. */
#ifndef FAKE_RINTERNALS_H
#define FAKE_RINTERNALS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FALSE = 0, TRUE = 1 } Rboolean;
typedef ptrdiff_t R_xlen_t;

#define NILSXP 0
#define CHARSXP 9
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19

typedef struct fake_sexp {
  int type;
  R_xlen_t length;
  int nrow, ncol;          /* ncol > 0: a matrix */
  void* data;              /* double / int / struct fake_sexp* / char, by type */
  struct fake_sexp* names;
} * SEXP;

extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;
extern SEXP R_GlobalEnv;
extern SEXP R_UnboundValue;
extern SEXP R_NaString;
#define TYPEOF(x) ((x)->type)

SEXP Rf_allocVector(int type, R_xlen_t n);
SEXP Rf_allocMatrix(int type, int nrow, int ncol);
SEXP Rf_mkChar(const char* s);
SEXP Rf_mkString(const char* s);
SEXP Rf_ScalarLogical(int v);
SEXP Rf_ScalarInteger(int v);
SEXP Rf_ScalarReal(double v);
SEXP Rf_ScalarString(SEXP ch);
SEXP Rf_findVarInFrame(SEXP env, SEXP sym);   /* knows `.Random.seed` of the global environment only */
char* R_alloc(size_t n, int size);
void Rprintf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
SEXP Rf_install(const char* name);
SEXP Rf_GetOption1(SEXP tag);
SEXP Rf_setAttrib(SEXP x, SEXP what, SEXP value);
SEXP Rf_getAttrib(SEXP x, SEXP what);

double* REAL(SEXP x);
int* INTEGER(SEXP x);
int* LOGICAL(SEXP x);
const char* CHAR(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);

int Rf_length(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
Rboolean Rf_isReal(SEXP x);
Rboolean Rf_isInteger(SEXP x);
Rboolean Rf_isString(SEXP x);
Rboolean Rf_isNewList(SEXP x);
Rboolean Rf_isMatrix(SEXP x);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int Rf_asLogical(SEXP x);

SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)

void Rf_error(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_onintr(void);
void R_CheckUserInterrupt(void);
Rboolean R_ToplevelExec(void (*fun)(void*), void* data);

void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);

/* harness side */
void fake_r_set_option_int(const char* name, int v);
void fake_r_set_option_real(const char* name, double v);
void fake_r_set_option_string(const char* name, const char* v);
int fake_r_protect_depth(void);
void fake_r_set_random_seed(int present);   /* give the global environment a `.Random.seed` */
int fake_r_unif_rand_calls(void);
unsigned long long fake_r_random_seed_hash(void);
extern int fake_r_interrupt_after;   /* > 0: R_CheckUserInterrupt "interrupts" at that call */

#ifdef __cplusplus
}
#endif
#endif
