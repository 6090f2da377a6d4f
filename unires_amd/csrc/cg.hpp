// cg.hpp - device-resident CG state + launchers of the vector / scalar kernels (cg.hip).
#pragma once
#include "common.hpp"

namespace unires {

constexpr int kMaxCgIter = 4096;

struct CgState {  // lives in device memory, owned by the plan
  double rz, pAp, alpha, beta, obj_max, obj_min;
  int done, iters;
  unsigned gen, pad_;  // solves started on this plan (k_sc_init counts); tags the published progress word
  double rzpp[2];  // r.z of iterations k (slot k & 1) and k - 1: the folded kernels read one slot while
                   // workgroup 0 of the same launch writes the other
  double obj[kMaxCgIter + 1];  // objective trace; iteration k at slot k % (kMaxCgIter + 1)
  // UNIRES_STOP_MAXGAIN_GUARDED (k_sc_beta_guarded / k_sc_obj_guarded)
  int skip_fresh;        // 1: this iteration's fresh objective (a second A(x)) is not needed - its kernels return at entry
  int fresh_prev_ok;     // fresh_prev holds the FRESH objective of the previous iteration
  double rec_prev, fresh_prev, gain_rec;  // recurred objective of the previous iteration; this iteration's recurred gain
};

// Progress word of a solve, published by its scalar kernels to host-mapped memory (chunked solves,
// api.hip): generation << 32 | done << 31 | iterations completed.
__host__ __device__ inline unsigned long long cg_progress_word(unsigned gen, int done, int iters) {
  return ((unsigned long long)gen << 32) | (done ? 0x80000000ull : 0ull) | (unsigned)(iters & 0x7fffffff);
}

int vec_num_blocks(size_t n);  // grid (= number of partials) of the vector kernels
void launch_residual_init(const float *b, const float *ax, const float *x, float *r, float *p,
                          size_t n, double *part_rr, double *part_obj, const float *M,
                          hipStream_t st);
void launch_dot(const float *a, const float *b, size_t n, double *part, const int *done,
                hipStream_t st);
void launch_update_xr(const CgState *s, const float *p, const float *ap, float *x, float *r,
                      const float *b, size_t n, double *part_rr, double *part_obj, const float *M,
                      hipStream_t st);
// x == nullptr in launch_update_xr defers "x += alpha p" to launch_update_p(..., x): one volume
// pass less per iteration (only valid when nothing stops the solve between the two launches)
void launch_update_p(const CgState *s, const float *r, float *p, size_t n, const float *M, float *x,
                     hipStream_t st);
// M = nullptr: identity preconditioner; else z = r / M (Jacobi)
void launch_scale_shift(float a, float c, float *y, size_t n, hipStream_t st);
void launch_fill(float v, float *y, size_t n, hipStream_t st);
void launch_div(const float *a, const float *m, float *y, size_t n, hipStream_t st);  // y = a / m
void launch_axpy(float a, const float *x, float *y, size_t n, hipStream_t st);
// hostw (nullable): host-mapped progress word, written (system scope) when an iteration's last scalar
// kernel has run.  k < 0 in sc_beta / sc_obj: the iteration index is the state's own counter (iters + 1 /
// iters), so that a captured chunk of iterations can be replayed for any part of a solve.
void launch_sc_init(CgState *s, const double *part_rr, const double *part_obj, int g, int mode,
                    int check, unsigned long long *hostw, hipStream_t st);
void launch_sc_alpha(CgState *s, const double *part, int g, hipStream_t st);
void launch_sc_beta(CgState *s, const double *part_rr, const double *part_obj, int g, int k,
                    int obj_kind, double tol, unsigned long long *hostw, hipStream_t st);
void launch_sc_obj(CgState *s, const double *part, int g, int k, double tol, unsigned long long *hostw,
                   hipStream_t st);
// guarded 'max_gain' (api.hip: cg_enqueue_iters): sc_beta with the recurred objective decides whether the fresh one is
// needed; sc_obj, if it runs, decides with it
void launch_sc_beta_guarded(CgState *s, const double *part_rr, const double *part_obj, int g, int k, double tol,
                            unsigned long long *hostw, hipStream_t st);
void launch_sc_obj_guarded(CgState *s, const double *part, int g, int k, double tol, unsigned long long *hostw,
                           hipStream_t st);
void launch_sum_to(const double *part, int g, double *out, hipStream_t st);
// Folded forms (no scalar kernels between the matvec and the vector updates): EVERY workgroup
// re-reduces the producer's partial sums in a fixed order in its prologue - visibility comes from
// the kernel boundary, the order is the same everywhere, so all workgroups hold the same alpha /
// beta bit for bit; workgroup 0 records them in the state.
int vec_num_blocks_fold(size_t n);
// r -= alpha Ap with alpha = rz / sum(part_pap[0..g)); part_rr (vec_num_blocks_fold(n) doubles, NOT the
// buffer part_pap lives in) gets sum r*z
void launch_update_r_fold(CgState *s, const double *part_pap, int g, int k, const float *ap, float *r,
                          size_t n, double *part_rr, const float *M, hipStream_t st);
// x += alpha p; p = z + beta p with beta = sum(part_rr[0..g)) / rz; records rz, beta, iters = k
void launch_update_px_fold(CgState *s, const double *part_rr, int g, int k, const float *r, float *p,
                           float *x, size_t n, const float *M, hipStream_t st);

}  // namespace unires
