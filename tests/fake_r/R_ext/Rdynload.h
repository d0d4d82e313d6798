/* tests/fake_r/R_ext/Rdynload.h -- routine registration of the test double (see ../Rinternals.h). */
#ifndef FAKE_RDYNLOAD_H
#define FAKE_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct fake_dllinfo { const R_CallMethodDef* call_routines; int use_dynamic_symbols; } DllInfo;
int R_registerRoutines(DllInfo* info, const void* c, const R_CallMethodDef* call, const void* f, const void* e);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
