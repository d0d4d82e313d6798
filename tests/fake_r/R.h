/* tests/fake_r/R.h -- see Rinternals.h in this directory (test double, not R). */
#ifndef FAKE_R_H
#define FAKE_R_H
#include "Rinternals.h"
#endif
