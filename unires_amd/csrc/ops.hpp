// ops.hpp - host launchers of the op-level kernels (ops.hip).
#pragma once
#include "common.hpp"

namespace unires {

void launch_pull(const float *src, Dim3i sd, const Affine &A, float *dst, Dim3i gd, float tol,
                 const int *done, hipStream_t st);
void launch_conv_down(const float *src, Dim3i gd, const Taps &T, const Scaling &S, float *dst,
                      Dim3i xd, const int *done, hipStream_t st);
void launch_conv_up(const float *xs, Dim3i xd, const Taps &T, const Scaling &S, float *dst,
                    Dim3i gd, hipStream_t st);
void launch_grad(const float *src, Dim3i d, const float vx[3], float *dst3, hipStream_t st);
// dst = [add +] scale * Dt(ca*ua + cb*ub)   (ub, add may be NULL)
void launch_div(const float *ua, const float *ub, float ca, float cb, Dim3i d, const float vx[3],
                float scale, const float *add, float *dst, hipStream_t st);
void launch_pull_grad(const float *src, Dim3i sd, const Affine &A, float *dst, Dim3i gd, float tol,
                      hipStream_t st);
// separable (one 1-D pass per axis) forms of conv_down / conv_up for profiles with many taps;
// a, b: scratch volumes of at least numel(gd) floats each
void launch_conv_down_sep(const float *g, Dim3i gd, const Taps &T, const Scaling &S, float *dst,
                          Dim3i xd, float *a, float *b, const int *done, hipStream_t st);
float *launch_conv_up_sep(const float *xs, Dim3i xd, const Taps &T, const Scaling &S, Dim3i gd,
                          float *a, float *b, hipStream_t st);
// dst = conv_up_ax(S conv_down_ax(src)) for a stride-2 axis ax (0 or 1) in one marching pass: src and dst are sd
// volumes, the n_mid-long intermediate stays in registers.  Non-zero: not available, nothing launched.
int launch_conv_downup2(const float *src, Dim3i sd, const Taps &T, const Scaling &S, int ax, int n_mid, float *dst,
                        const int *done, hipStream_t st);
// dst (sd.x, ny_mid, sd.z) = conv_up_x(S conv_down_x(S conv_down_y(src))), stride-2 profiles along x and y, one
// kernel (S applies on its own axis, 0 or 1).  gy > 0: conv_up_y as well, dst is (sd.x, gy, sd.z).  Non-zero: not
// available, nothing launched.
int launch_conv_ydown_xdownup2(const float *src, Dim3i sd, const Taps &T, const Scaling &S, int nx_mid, int ny_mid,
                               int gy, float *dst, const int *done, hipStream_t st);
int dtd_num_blocks(Dim3i d);
// dst = a*src + c*DtD(src); partials (nullable, dtd_num_blocks doubles) gets sum(src*dst) pieces;
// with objb (needs partials): partials = sum (dst - 2 objb) * src and dst is not stored.
void launch_dtd(const float *src, Dim3i d, const float vx[3], float a, float c, float *dst,
                double *partials, const float *objb, const int *done, hipStream_t st);

}  // namespace unires
