// shift.hpp - one-kernel CG matvec for observations translated (not rotated) against the grid (shift.hip).
#pragma once
#include "common.hpp"

namespace unires {

// per-operator tables: the x / y blends and the z-line operator of AtA (device, one allocation)
struct ShiftPlan {
  float *dev = nullptr;
  size_t cap = 0;  // floats allocated
  size_t o_e = 0, o_cx = 0, o_cy = 0, o_f = 0, o_e4 = 0, o_kmin = 0;
  bool lane_window = false;  // the lane-window tables (e4, kmin) describe the operator
  int nf = 0, s = 1, oz = 0, padl = 0, padr = 0, wave_floats = 0, xdz = 0;
  Dim3i dd{0, 0, 0};
  float key[12] = {0};  // the affine it was built for
  bool valid = false;
};

// Builds (or rebuilds) the tables; synchronous copies (plan time only).  Non-zero: the operator is not
// an identity linear part + non-integer translation with a z-only slice profile (plan left invalid).
// S2 = the even / odd scaling of AtA, S(2 scl).
int shift_build(ShiftPlan &S, Dim3i dd, Dim3i gd, Dim3i xd, const Taps &T, const Scaling &S2, const Affine &A,
                float tol);
void shift_free(ShiftPlan &S);
int shift_blocks(Dim3i dd);  // partial sums written per launch
bool shift_fast(const ShiftPlan &S, Dim3i dd);  // the x-marching kernel's fast form applies (it then also beats aligned.hip's kernel)
// q = tau AtA p + a0 p + c DtD p (+ partials of sum p*q, or of the objective sum (q - 2 objb) p without
// storing q).  Non-zero return: no valid plan for this operator / unaligned volumes; nothing launched.
int launch_ata_shift(const ShiftPlan &S, const float *p, float *q, Dim3i dd, const Affine &A, float tau, float a0,
                     float cx, float cy, float cz, double *partials, const float *objb, const int *done,
                     hipStream_t st);

}  // namespace unires
