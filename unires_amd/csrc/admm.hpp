// admm.hpp - launchers of the z/w-update and objective kernels (admm.hip).
#pragma once
#include "common.hpp"

namespace unires {

// scale = JTV shrinkage factor (norm_only == 0) or partials of sum(JTV norm) (norm_only == 1);
// any number of channels (more than 8: chained launches carrying the sum of squares in `scale`, which
// must then be given); returns the number of partials written (or -1 if nc > 8 without `scale`).
int launch_jtv_scale(const float *const *y, const float *lam, int nc, const float *w,
                     const float *z_old, Dim3i d, const float vx[3], float rho, float alpha,
                     float *scale, double *partials, int norm_only, hipStream_t st);
void launch_zw_update(const float *y, float lam, const float *scale, float *z, float *w, Dim3i d,
                      const float vx[3], float rho, float alpha, hipStream_t st);
int scaling_sums_blocks(Dim3i d);
// part: 5 * scaling_sums_blocks(d) doubles of scratch (per-workgroup sums, added in index order)
void launch_scaling_sums(const float *x, const float *y, Dim3i d, int dim_thick, double *part, double *out,
                         hipStream_t st);
void launch_rigid_sums(const float *gr3, const float *diff, const float *ctc, Dim3i dm,
                       const float D[6][12], double *out, hipStream_t st);
void launch_clean_fov(float *y, Dim3i d, const Affine &M, Dim3i dx, hipStream_t st);
int launch_masked_sse(const float *x, const float *ay, size_t n, double *partials, hipStream_t st);

}  // namespace unires
