// pull2.hpp - LDS-window pull (+ conv_down along z), see pull2.hip.
#pragma once
#include <string.h>

#include "fused.hpp"

namespace unires {

// per-operator table of the kernel's workgroup geometry (window origins), built with the plan
struct PullPlan {
  int *rec = nullptr;  // device
  size_t cap = 0;      // workgroup records allocated
  int *itab = nullptr; // device: per staging item {plane group * 4, byte offset inside the window, cxl | cyl << 16, 0}
  size_t itab_cap = 0; // items allocated
  int *wtab = nullptr; // device: dispatch index -> {record, bi, bj, bc} of the kernel's walk (equal cost per XCD), then a class byte per workgroup
  size_t wtab_cap = 0;
  bool use_wtab = false;
  bool valid = false;
  float tol = 0.f;               // in-FOV tolerance the all-outside flags were computed for
  unsigned char key[192] = {0};  // the geometry it was built for
};

// Builds (or rebuilds) the table; synchronises the device (plan time only).  Non-zero: the
// operator is outside the kernel's domain (plan left invalid; callers use the general kernels).
int pull2_build(PullPlan &Q, Dim3i sd, const Affine &A, const Taps &T, Dim3i xd, Dim3i gd, float tol);
void pull2_free(PullPlan &Q);

// xs = S conv_down_z pull_A(src)  (dst is the grid-space volume when T has no taps).
// Non-zero return: no valid plan for this operator, nothing launched.
int launch_pull_conv2(const PullPlan &Q, const float *src, Dim3i sd, const Affine &A, const Taps &T,
                      const Scaling &S, float *dst, Dim3i xd, Dim3i gd, float tol, const int *done,
                      hipStream_t st);

}  // namespace unires
