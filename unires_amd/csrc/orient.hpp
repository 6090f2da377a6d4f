// orient.hpp - orientation of an observation's voxel axes against the output lattice (orient.hip).
//
// The reference hands whatever affine a NIfTI file carries straight to the operator
// (unires/_core.py:145-168 resets CT affines only, _util.py:134-197 keeps `mat` as read,
// _project.py:147-159 builds a dense grid from it): sagittal / coronal storage (voxel axes
// permuted against the world axes) and LAS-vs-RAS reflections reach A as a signed axis permutation
// in the linear part of M = mat_y \ rigid mat_yx.  The kernels of the fused path are built for
// grids whose axis d runs mainly along +d of the output; the plan therefore relabels the x-space
// voxel axes ONCE per operator (fill_repeat, api.hip) so that they do, and the few entry points
// that take or return x-space volumes in the caller's layout re-order them with the kernels here.
#pragma once
#include "common.hpp"

namespace unires {

// canonical axis j of the plan = caller's axis perm[j], reversed where flip[j]
struct Orient {
  int perm[3] = {0, 1, 2};
  int flip[3] = {0, 0, 0};
  bool identity() const {
    return perm[0] == 0 && perm[1] == 1 && perm[2] == 2 && !flip[0] && !flip[1] && !flip[2];
  }
};

// Signed permutation that brings the linear part of the grid -> output affine A closest to a
// positive diagonal: maximises sum_j |A[j][perm[j]]| / |column perm[j]|; the identity wins ties.
Orient orient_of(const Affine &A);

// dst (canonical layout, dims dc) <- src (caller's layout, dims du);  dc[j] = du[perm[j]]
void launch_to_canonical(const Orient &O, const float *src, Dim3i du, float *dst, hipStream_t st);
// dst (caller's layout, dims du) <- src (canonical layout)
void launch_from_canonical(const Orient &O, const float *src, float *dst, Dim3i du, hipStream_t st);

}  // namespace unires
