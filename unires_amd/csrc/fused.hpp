// fused.hpp - launchers of the fused matvec kernels (fused.hip).
#pragma once
#include "common.hpp"

namespace unires {

struct PushSrc {
  const float *data;  // grid-space volume (convup == 0) or x-space volume (convup == 1)
  int convup;
  Dim3i xd;  // x-space dims (convup only)
  Dim3i gd;  // grid dims
  Taps T;
  Scaling S;
};

struct PushEpilogue {
  const float *p = nullptr;  // + a0 p + c DtD p  (c split per axis as c / vx_d^2)
  float a0 = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  int accumulate = 0;          // dst += instead of dst =
  double *partials = nullptr;  // push_tile_blocks(dd) doubles: pieces of sum(p * dst)
  const float *objb = nullptr; // objective mode: partials = sum (dst - 2 objb) * p, dst not stored
  int grid_cap = 0;            // persistent kernels (k_splat2, k_ata1): at most this many workgroups; 0 = as many as the chip holds
};

// xs = S conv_down pull_A(src).  Returns non-zero if the tile does not fit LDS
// (nothing launched; caller uses the unfused kernels).
int launch_pull_conv(const float *src, Dim3i sd, const Affine &A, const Taps &T, const Scaling &S,
                     float *dst, Dim3i xd, Dim3i gd, float tol, const int *done, hipStream_t st);

// Host-side geometry check that makes the atomic-free splat of k_push_tile race-free
// for a given affine (compute once per operator, ~1 ms).
struct SplatSafety {
  int row_sep = 1 << 14;
  int use_atomics = 1;
};
void splat_safety(const Affine &A, int &row_sep, int &use_atomics);

int push_tile_blocks(Dim3i dd);
// Returns non-zero (nothing launched) if the conv_up fan-in exceeds the kernel tables.
int launch_push_tile(const PushSrc &src, const Affine &A, const Affine &Ainv,
                     const SplatSafety &safe, float alpha, float tol, const PushEpilogue &ep,
                     float *dst, Dim3i dd, const int *done, hipStream_t st);

// Lean specialisation of the tile push (splat.hip): grid-space source, or conv_up along z
// with fan-in <= 2.  Non-zero return: not applicable, nothing launched.
int splat_blocks(Dim3i dd, const Affine &A);  // the tile shape depends on the operator
int launch_splat(const PushSrc &src, const Affine &A, const Affine &Ainv, const SplatSafety &safe,
                 float alpha, float tol, const PushEpilogue &ep, float *dst, Dim3i dd,
                 const int *done, hipStream_t st);

}  // namespace unires
