// fftpre.hpp - FFT-diagonal preconditioner (build-side extension named by the north star; the
// reference runs unpreconditioned, unires/_update.py:136-137):
//     M^-1 r = IFFT( FFT(r) / (a + sum_d c_d (2 - 2 cos(2 pi k_d / n_d))) )
// i.e. the exact inverse of  a I + sum_d c_d L_d  with PERIODIC second differences L_d - the
// circulant neighbour of the operator  sum tau AtA + rho lam^2 DtD  (a = mean diagonal of the
// data term, c_d = rho lam^2 / vx_d^2).  Symmetric positive definite, so PCG applies.
#pragma once
#include <hipfft/hipfft.h>

#include "common.hpp"

namespace unires {

struct FftPre {
  hipfftHandle fwd = 0, inv = 0;
  bool have_plans = false;
  Dim3i d{0, 0, 0};
  float2 *freq = nullptr;  // (X, Y, Z/2+1) complex
  float *z = nullptr;      // (X, Y, Z) preconditioned residual
  float *lam[3] = {nullptr, nullptr, nullptr};  // per-axis eigenvalues 2 - 2 cos(2 pi k / n)
  float a = 0.f, c[3] = {0.f, 0.f, 0.f};
};

// Allocates plans and buffers for volumes of size d (idempotent).  Returns 0 / hipfft or hip error.
int fftpre_setup(FftPre &F, Dim3i d);
void fftpre_destroy(FftPre &F);
// out = M^-1 in (in is preserved; out may be F.z).  Returns 0 or an error code.
int fftpre_apply(FftPre &F, const float *in, float *out, hipStream_t st);

}  // namespace unires
