// admm.hpp - launchers of the z/w-update and objective kernels (admm.hip).
#pragma once
#include "common.hpp"

namespace unires {

// scale = JTV shrinkage factor (norm_only == 0) or partials of sum(JTV norm) (norm_only == 1);
// any number of channels (more than 8: chained launches carrying the sum of squares in `scale`, which
// must then be given); returns the number of partials written (or -1 if nc > 8 without `scale`).
int launch_jtv_scale(const float *const *y, const float *lam, int nc, const float *w,
                     const float *z_old, Dim3i d, const float vx[3], float rho, float alpha,
                     float *scale, double *part, double *out, int norm_only, hipStream_t st);
int jtv_scale_blocks(Dim3i d);  // doubles of scratch (`part`) a launch with `out` needs
// out[k] = sum_b part[b * ncols + k] in index order (the second stage of every float64 reduction here: no atomics)
void launch_sum_cols(const double *part, int nb, int ncols, double *out, hipStream_t st);
void launch_zw_update(const float *y, float lam, const float *scale, float *z, float *w, Dim3i d,
                      const float vx[3], float rho, float alpha, hipStream_t st);
int scaling_sums_blocks(Dim3i d);
// part: 5 * scaling_sums_blocks(d) doubles of scratch (per-workgroup sums, added in index order)
void launch_scaling_sums(const float *x, const float *y, Dim3i d, int dim_thick, double *part, double *out,
                         hipStream_t st);
int rigid_sums_blocks(Dim3i dm);
// part: 27 * rigid_sums_blocks(dm) doubles of scratch
void launch_rigid_sums(const float *gr3, const float *diff, const float *ctc, Dim3i dm,
                       const float D[6][12], double *part, double *out, hipStream_t st);
void launch_clean_fov(float *y, Dim3i d, const Affine &M, Dim3i dx, hipStream_t st);
int masked_sse_blocks(size_t n);
// part: masked_sse_blocks(n) doubles of scratch
int launch_masked_sse(const float *x, const float *ay, size_t n, double *part, double *out, hipStream_t st);

}  // namespace unires
