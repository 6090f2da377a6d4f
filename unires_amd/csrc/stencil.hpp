// stencil.hpp - flat streaming form of q = a0 p + c DtD p for regime A = I (stencil.hip).
#pragma once
#include "common.hpp"

namespace unires {

int dtd_flat_blocks(Dim3i dd);  // partial sums written per launch
// q = a0 p + c DtD p with per-axis weights cx, cy, cz = c / vx^2 (+ partials of sum p*q, or of the
// objective sum (q - 2 objb) p without storing q).  Non-zero return: volume outside the kernel's
// domain, nothing launched.
int launch_dtd_flat(const float *p, float *q, Dim3i dd, float a0, float cx, float cy, float cz,
                    double *partials, const float *objb, const int *done, hipStream_t st);

}  // namespace unires
