// aligned.hpp - one-kernel CG matvec for grid-aligned observations (aligned.hip).
#pragma once
#include "common.hpp"

namespace unires {

bool affine_is_integer_shift(const Affine &A, int off[3]);
int aligned_blocks(Dim3i dd);
// q = tau AtA p + a0 p + c DtD p (+ partials of sum p*q, or of the objective
// sum (q - 2 objb) p without storing q).  Non-zero return: the operator is not
// an integer shift + z-only slice profile; nothing launched.
int launch_ata_aligned(const float *p, float *q, Dim3i dd, Dim3i gd, Dim3i xd, const Taps &T,
                       const Scaling &S2, const Affine &A, float tau, float a0, float cx,
                       float cy, float cz, double *partials, const float *objb, const int *done,
                       hipStream_t st);

// Regime A = I through the same line kernel: q = a0 p + c DtD p (+ partials).
int launch_dtd_lines(const float *p, float *q, Dim3i dd, float a0, float cx, float cy, float cz,
                     double *partials, const float *objb, const int *done, hipStream_t st);

}  // namespace unires
