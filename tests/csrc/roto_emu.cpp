// roto_emu.cpp - TEST INFRASTRUCTURE ONLY.  Builds qmps_amd/csrc/qmps_roto_rule.h (the scalar minimisers of the double-frequency
// rotosolve update, the very source the gfx950 kernels include) with g++, so that the CPU test-suite can compare it call by call
// with the answers scipy's minimize_scalar gave the reference (tests/golden/refshim_golden.npz).  Nothing in the product loads this.
#include "qmps_roto_rule.h"

extern "C" {
// f(x) = a sin 2x + b cos 2x + c sin x + d cos x;  rule 0: the reference's bounded Brent search, 1: global argmin
double roto_emu_step(double a, double b, double c, double d, int rule) { return qmps::double_sinusoid_step(a, b, c, d, rule); }
void roto_emu_steps(long n, const double* abcd, int rule, double* out) {
  for (long i = 0; i < n; ++i) out[i] = qmps::double_sinusoid_step(abcd[4 * i], abcd[4 * i + 1], abcd[4 * i + 2], abcd[4 * i + 3], rule);
}
}
