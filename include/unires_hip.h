/*
 * unires_hip.h - C ABI of libunires_hip.so: the MI355X (gfx950) implementation of
 * the UniRes ADMM y-update hot path.
 *
 * All volumes are float32, C-contiguous (X,Y,Z) with Z fastest, resident in
 * device memory; all `dim`/`M`/`taps`/descriptor arguments are HOST pointers
 * read during the call (never retained).  Every entry point returns an int
 * status (0 = OK) and never throws; unires_last_error() gives the message of
 * the calling thread's last failure.  Kernels are launched on the caller's
 * stream (`stream` is a hipStream_t passed as void*; NULL = default stream).
 * No entry point synchronises a stream, except unires_cg_solve when the caller asks for
 * the realised iteration count / objective trace on the host.  (A solve that can stop
 * early - tol > 0 - keeps the calling thread until its last chunk of iterations is
 * enqueued; it watches a host-mapped progress word, not the stream.)
 *
 * The reference (brudfors/UniRes) has no FFI layer: its seam is Python calls
 * into nitorch + torch.  Each entry point below cites the
 * reference call it replaces (paths relative to the reference tree).
 */
#ifndef UNIRES_HIP_H
#define UNIRES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNIRES_HIP_ABI_VERSION 1
#define UNIRES_MAX_TAPS 32 /* per axis */

enum unires_status {
  UNIRES_OK = 0,
  UNIRES_ERR_NULL = 1,        /* null pointer argument                         */
  UNIRES_ERR_DIM = 2,         /* inconsistent / non-positive dimensions        */
  UNIRES_ERR_ARG = 3,         /* bad enum / value ("Undefined operator" ...)   */
  UNIRES_ERR_HIP = 4,         /* HIP runtime error (message has hipGetErrorString) */
  UNIRES_ERR_ALLOC = 5,       /* device allocation failed                      */
  UNIRES_ERR_UNSUPPORTED = 6  /* valid request this build cannot serve         */
};

/* unires/_project.py:122-126: 'A' | 'At' | 'AtA' ('none' is regime 0) */
enum unires_operator { UNIRES_OP_A = 0, UNIRES_OP_AT = 1, UNIRES_OP_ATA = 2 };

/* unires/_core.py:209-260 selects one of three operator regimes */
enum unires_regime {
  UNIRES_REGIME_IDENTITY = 0, /* sett.do_proj == False: A = I   (_project.py:76-77,91-92) */
  UNIRES_REGIME_DENOISE = 1,  /* method 'denoising': A = pull   (_project.py:180-188)     */
  UNIRES_REGIME_SUPERRES = 2  /* method 'super-resolution': A = S conv_down pull (:161-179) */
};

/* nitorch cg(stop=...) objective branch */
enum unires_cg_stop {
  UNIRES_STOP_RESIDUAL = 0,       /* 'e': obj = sqrt(r.z)                                  */
  UNIRES_STOP_MAXGAIN = 1,        /* anything else, e.g. 'max_gain' as UniRes passes
                                     (_update.py:145): obj = 0.5*sum(x*(A(x)-2b)),
                                     one extra A(x) per iteration (reference-faithful)     */
  UNIRES_STOP_MAXGAIN_RECURRED = 2, /* same objective from the recurred residual,
                                     obj = -0.5*sum(x*(b+r)): no extra A(x) (build-side)   */
  UNIRES_STOP_MAXGAIN_GUARDED = 3  /* the recurred objective while its gain is >= 4 tol, the fresh one
                                     (mode 1's) from there on: mode 1's decisions - both values of a close
                                     gain are fresh ones - at about mode 2's cost                          */
};

enum unires_precond {
  UNIRES_PRECOND_IDENTITY = 0, /* the reference's only live mode (_update.py:136-137) */
  UNIRES_PRECOND_JACOBI = 1,   /* _precond (_update.py:80-102), commented out at :136:
                                  z = r / (tau AtA(1) + 2 rho lam^2 sum_d 1/vx_d^2)         */
  UNIRES_PRECOND_FFT = 2       /* build-side extension (not in the reference): exact inverse of
                                  the circulant operator a I + rho lam^2 sum_d L_d / vx_d^2 with
                                  periodic second differences L_d, a = mean diagonal of
                                  sum_n tau_n AtA_n; two rocFFT real 3-D transforms per call  */
};

const char *unires_last_error(void);
int unires_abi_version(void);

/* ------------------------------------------------------------------------
 * Op level - one call per nitorch / torch function on the path
 * ---------------------------------------------------------------------- */

/* nitorch grid_pull(src, affine_grid(M, gdim), 'linear', bound='zero',
 * extrapolate=False)  (_project.py:159,164,174,183,187).  M is the row-major
 * 3x4 float32 affine mapping a grid voxel (i,j,k) to src voxel coordinates.
 * dst[gdim] is overwritten. fov_tol = nitorch's in-FOV tolerance (5e-2). */
int unires_pull3d_affine(const float *src, const int32_t sdim[3], const float M[12], float *dst,
                         const int32_t gdim[3], float fov_tol, void *stream);

/* nitorch grid_grad(src, affine_grid(M, gdim), 'linear', bound='zero', extrapolate=False)
 * (_update.py:508, the rigid Gauss-Newton's spatial derivatives): gradient of the trilinear
 * sample w.r.t. the voxel coordinate; dst3 is (gdim, 3), component fastest. */
int unires_pull_grad3d_affine(const float *src, const int32_t sdim[3], const float M[12],
                              float *dst3, const int32_t gdim[3], float fov_tol, void *stream);

/* nitorch grid_push(src, affine_grid(M, gdim), shape=ddim, ...)
 * (_project.py:172,179,185,188): dst (+)= alpha * push(src).
 * accumulate == 0 overwrites dst. */
int unires_push3d_affine(const float *src, const int32_t gdim[3], const float M[12], float *dst,
                         const int32_t ddim[3], float alpha, float fov_tol, int accumulate,
                         void *stream);

/* F.conv3d(src, smo_ker, stride)  (_project.py:153) with a separable kernel
 * smo_ker = taps[0] (x) taps[1] (x) taps[2] (cross-correlation, no padding), followed
 * by the optional even/odd slice scaling of _apply_scaling (_project.py:9-24):
 * slices with even index along scl_dim are multiplied by exp(scl), odd by
 * exp(-scl); scl == 0 skips it.  Requires sdim = (ddim-1)*stride + ntaps. */
int unires_conv_down3d(const float *src, const int32_t sdim[3], const float *const taps[3],
                       const int32_t ntaps[3], const int32_t stride[3], float *dst,
                       const int32_t ddim[3], float scl, int32_t scl_dim, void *stream);

/* F.conv_transpose3d(S(scl) src, smo_ker, stride)  (_project.py:154,168-171): exact
 * adjoint of unires_conv_down3d, scaling applied to the INPUT.  src has sdim
 * (low-res), dst has ddim = (sdim-1)*stride + ntaps and is overwritten. */
int unires_conv_up3d(const float *src, const int32_t sdim[3], const float *const taps[3],
                     const int32_t ntaps[3], const int32_t stride[3], float *dst,
                     const int32_t ddim[3], float scl, int32_t scl_dim, void *stream);

/* nitorch im_gradient(src, vx, bound='zero', which='forward')
 * (_project.py:314; _update.py:168,176,188,419): dst is (3,X,Y,Z). */
int unires_grad_fwd_zero(const float *src, const int32_t dim[3], const float vx[3], float *dst3,
                         void *stream);

/* nitorch im_divergence(src3, vx, bound='zero', which='forward') - the POSITIVE
 * adjoint of the gradient (_project.py:315; _update.py:132). */
int unires_div_fwd_zero(const float *src3, const int32_t dim[3], const float vx[3], float *dst,
                        void *stream);

/* dst = a*src + c*DtD(src): _DtD (_project.py:300-317) fused with the scalar
 * multiply-add of _proj (_project.py:87). */
int unires_dtd(const float *src, const int32_t dim[3], const float vx[3], float a, float c,
               float *dst, void *stream);

/* ------------------------------------------------------------------------
 * Fused level - plan + matvec + RHS + CG  (unires/_update.py:118-152)
 * ---------------------------------------------------------------------- */

/* One observation ("repeat") of a channel: the fields of _proj_op
 * (struct.py:36-54) that the operators read, flattened. */
typedef struct unires_repeat {
  int32_t dim_x[3];  /* observed image dims                                             */
  int32_t dim_g[3];  /* grid dims: dim_yx (super-resolution) or dim_x (denoising)       */
  float M[12];       /* float32(mat_y \ (rigid*mat_yx|mat_x)), row-major 3x4 (:147,150,159) */
  int32_t ratio[3];  /* conv stride (:153)                                              */
  int32_t ntaps[3];  /* separable factors of smo_ker (:277)                             */
  const float *taps[3]; /* HOST pointers, ntaps[d] floats each                          */
  float scl;         /* even/odd slice scaling parameter po.scl (:287-290); 0 = off     */
  int32_t dim_thick; /* axis the scaling runs along (:241)                              */
  float tau;         /* noise precision x[c][n].tau (_core.py:134-136)                  */
} unires_repeat_t;

typedef struct unires_plan unires_plan_t;

/* Builds the per-channel operator  sum_n tau_n A_n^T A_n + rho lam^2 D^T D  and
 * allocates every workspace it will ever need (r, p, Ap, one grid-space and one
 * x-space intermediate, reduction partials).  Nothing is allocated afterwards. */
int unires_plan_create(unires_plan_t **plan, const int32_t dim_y[3], const float vx_y[3],
                       int32_t regime, int32_t n_repeats, const unires_repeat_t *repeats,
                       float fov_tol);
int unires_plan_destroy(unires_plan_t *plan);
/* Replace repeat n's descriptor (after a rigid / scaling update; _update.py:576). */
int unires_plan_set_repeat(unires_plan_t *plan, int32_t n, const unires_repeat_t *repeat);
/* Hint: the caller keeps `n_concurrent` solves (this plan's and other plans') in flight on the device at once - the
 * channels of one y-update, which do not couple (unires/_update.py:122-150), each on a stream of its own.  The
 * plan then sizes the persistent kernels of its matvec so that other channels' kernels find room on the CUs (see
 * api.hip; results are unchanged, reductions are summed in another - still fixed - order).  1 = the default. */
int unires_plan_set_concurrency(unires_plan_t *plan, int32_t n_concurrent);
/* Bytes of device workspace the plan owns. */
int64_t unires_plan_workspace_bytes(const unires_plan_t *plan);
/* Which kernels repeat n's operator runs on (no counterpart in the reference, whose _proj_apply
 * takes any mat_x through a dense grid, _project.py:147-159; here the plan relabels the observation's
 * voxel axes so that sagittal / coronal / reflected storage takes the same kernels as axial).
 * info[0..2] = caller's voxel axis behind canonical axis 0..2, info[3] = bit j set where canonical axis
 * j is reversed, info[4] = 1 if the LDS-window pull serves it, info[5] = 0 if the schedule-driven
 * splat does not serve it, else 2 + its conv_up axis (1: grid-space source, 2..4: along x / y / z, 5: all), info[6] bit 0 = the translation-only one-kernel
 * matvec serves it, bit 1 = the single-pass AtA kernel of the denoising regime does (ata1.hip), info[7] = 1 if the convolutions run as separable passes. */
int unires_plan_repeat_info(const unires_plan_t *plan, int32_t n, int32_t info[8]);
/* The relabelling a plan applies to an operator with grid -> output affine M (host arithmetic only, no device
 * call): perm[j] = caller's voxel axis behind canonical axis j, flip[j] = 1 where it is reversed, chosen so that
 * the permuted, sign-flipped linear part of M is as close to a positive diagonal as a signed permutation gets
 * (identity for every axis-aligned, right-handed geometry: the descriptor is then used as it is). */
int unires_orient_of(const float M[12], int32_t perm[3], int32_t flip[3]);
/* Measurement aid (no counterpart in the reference; bench.py's roofline leg).  While on,
 * unires_cg_solve launches its kernels one by one (no hipGraph replay) and brackets every operator
 * application A(p) of the solve with HIP events on the caller's stream.  Meaningful with tol == 0
 * only: launches enqueued after a solve has converged return at entry and are still counted.
 * on == 2: no events; instead every A(p) of a solve is enqueued TWICE (it is idempotent: same result), the solve
 * stays what it is in production - one hipGraph replayed.  The difference between a solve timed this way and a
 * plain one, divided by its iterations, is what an operator application costs INSIDE the replayed graph
 * (kernel + its dependent boundary, no host launch latency): bench.py's `us_per_launch_in_graph`. */
int unires_plan_time_matvecs(unires_plan_t *plan, int32_t on);
/* Waits for the recorded events; returns how many applications were recorded since the last call and
 * the sum of their durations (microseconds), and forgets them. */
int unires_plan_matvec_time(unires_plan_t *plan, int32_t *launches, double *total_us);

/* _proj_apply(operator, ., po_n)  (_project.py:99-190) for repeat n, WITHOUT tau.
 * in/out sizes follow the operator (A: dim_y -> dim_x; At: dim_x -> dim_y; AtA: dim_y -> dim_y). */
int unires_proj_apply(unires_plan_t *plan, int32_t n, int32_t op, const float *in, float *out,
                      void *stream);

/* q = sum_n tau_n AtA_n p + rho*lam^2 DtD p   (_proj('AtA'), _project.py:73-87).
 * If dot_dev != NULL the float64 sum(p*q) is written there (device pointer). */
int unires_ata_matvec(unires_plan_t *plan, float rho, float lam, const float *p, float *q,
                      double *dot_dev, void *stream);

/* b = sum_n tau_n At_n x_n - lam * Dt(w_c - rho z_c)   (_update.py:124-133).
 * x_ptrs: HOST array of n_repeats device pointers; w_c,z_c: (3,X,Y,Z) device. */
int unires_rhs_assemble(unires_plan_t *plan, const float *const *x_ptrs, const float *w_c,
                        const float *z_c, float rho, float lam, float *b, void *stream);

/* The data term of the RHS alone, atx = sum_n tau_n At_n x_n  (_update.py:125-128).  It only
 * changes when an observation, its rigid or its scaling changes, so a caller may keep it
 * across ADMM iterations (the reference recomputes it every time). */
int unires_atx_assemble(unires_plan_t *plan, const float *const *x_ptrs, float *atx,
                        void *stream);

/* b = atx - lam * Dt(w_c - rho z_c)   (_update.py:131-133) from a kept atx. */
int unires_rhs_from_atx(unires_plan_t *plan, const float *atx, const float *w_c,
                        const float *z_c, float rho, float lam, float *b, void *stream);

/* Builds the diagonal of _precond (_update.py:80-102) for (rho, lam) into plan-owned memory
 * (allocated on the first call): M = tau AtA(1) + 2 rho lam^2 sum_d 1/vx_d^2.  Like the
 * reference it supports one repeat per channel only.  If m_out != NULL the diagonal is also
 * copied there (device, dim_y floats).  Needed before unires_cg_solve(precond_mode = JACOBI)
 * with the same rho and lam. */
int unires_precond_build(unires_plan_t *plan, int32_t precond_mode, float rho, float lam,
                         float *m_out, void *stream);

/* out = precond(in) for the preconditioner last built on this plan (identity: copy). */
int unires_precond_apply(unires_plan_t *plan, const float *in, float *out, void *stream);

/* nitorch cg(A=lhs, b, x, precond, max_iter, tolerance, stop,
 * inplace=True, sum_dtype=float64)  (_update.py:142-148): x is updated in place.
 * tol == 0 runs exactly max_iter iterations with no objective evaluation.
 * If iters_out != NULL the call synchronises the stream and returns the realised
 * iteration count; obj_trace (HOST, max_iter+1 doubles, may be NULL) then
 * receives the objective values obj[0..iters].
 * tol > 0 (the reference's default, struct.py:65-67): the solve is enqueued in chunks of
 * UNIRES_CG_CHUNK iterations (default 2), one chunk ahead of the device; the kernel that ends an
 * iteration publishes (done, iterations) to host-mapped memory and the next chunk is enqueued only
 * while the stopping test has not fired - same kernels, same order, same iterate and trace as
 * enqueuing all max_iter iterations, without their no-op tail.  max_iter may then exceed 4096
 * (nitorch's default is 10 numel); obj_trace receives at most 4097 values (a ring beyond that). */
int unires_cg_solve(unires_plan_t *plan, float rho, float lam, const float *b, float *x,
                    int32_t max_iter, double tol, int32_t stop_mode, int32_t precond_mode,
                    int32_t *iters_out, double *obj_trace, void *stream);

/* The channels' solves of one y-update together (_update.py:122-150 loops over the channels; they do
 * not couple inside the y-update): n plans, each with its own b, x, rho, lam and stream (all HOST
 * arrays of n entries).  One host loop feeds the chunks of all of them, so channels on separate
 * streams keep overlapping on the device.  iters_out: n ints or NULL (non-NULL synchronises every
 * stream); obj_trace: n x (min(max_iter, 4096) + 1) doubles or NULL. */
int unires_cg_solve_many(int32_t n, unires_plan_t *const *plans, const float *rho, const float *lam,
                         const float *const *b, float *const *x, int32_t max_iter, double tol,
                         int32_t stop_mode, int32_t precond_mode, int32_t *iters_out,
                         double *obj_trace, void *const *streams);

/* ------------------------------------------------------------------------
 * Next rows of the path: the updates and sums that follow the
 * y-update inside every ADMM iteration  (unires/_update.py:154-195, 396-427)
 * ---------------------------------------------------------------------- */

/* UPDATE z and w  (_update.py:160-193), all channels:
 *   u_c = w_c/rho + lam_c D y_c ;  s = max(|u| - 1/rho, 0) / (|u| + 1e-7), |u| joint over c,d
 *   z_c = s u_c ;  w_c += rho (lam_c D y_c - z_c)
 * y_ptrs / lam: HOST arrays of n_channels device pointers / scalars (any n_channels >= 1; more than 8 run as chained launches);
 * z, w: (C,3,X,Y,Z) device, updated in place; jtv: (X,Y,Z) device, receives s (the
 * reference returns it as `tmp`, _update.py:195).  alpha is the over-relaxation
 * parameter (1 = none). */
int unires_zw_update(const float *const *y_ptrs, const float *lam, int32_t n_channels,
                     const int32_t dim[3], const float vx[3], float rho, float alpha, float *z,
                     float *w, float *jtv, void *stream);

/* -ln p(y) = sum_v sqrt(sum_c |lam_c D y_c|^2)  (_update.py:419-425) -> *out_dev (float64). */
int unires_nll_prior(const float *const *y_ptrs, const float *lam, int32_t n_channels,
                     const int32_t dim[3], const float vx[3], double *out_dev, void *stream);

/* sum_{x != 0} (x - ay)^2 in float64  (_update.py:414-417; the caller multiplies by tau/2). */
int unires_masked_sse(const float *x, const float *ay, int64_t n, double *out_dev, void *stream);

/* Gradient and Hessian sums of one rigid Gauss-Newton step (_update_rigid_channel,
 * _update.py:622-650) in one pass over the grid: gr3 (dim,3) = grid_grad of the pulled image,
 * diff (dim) = residual (conv_transposed, _update.py:524), ctc (dim) = conv_transpose(conv(1))
 * or NULL, d_rigid = the six 3x4 matrices mat_y^-1 dR/dq_i mat (row-major, float32).
 * out_dev (27 float64, device): [0..5] gradient, [6..26] upper triangle of the 6x6 Hessian,
 * row by row. */
int unires_rigid_sums(const float *gr3, const float *diff, const float *ctc, const int32_t dim[3],
                      const float d_rigid[72], double *out_dev, void *stream);

/* fit()'s clean_fov post-processing (run.py:150-164): y[v] = 0 where the voxel M v of the
 * low-resolution image lies outside [0, dim_x) on any axis; M (12 floats, row-major 3x4) =
 * float32 of inv(mat_y^-1 rigid mat_x).  Call once per observation. */
int unires_clean_fov(float *y, const int32_t dim_y[3], const float M[12], const int32_t dim_x[3],
                     void *stream);

/* The masked sums of one Gauss-Newton step on the even/odd slice scaling
 * (_update_scaling, _update.py:310-336); x, ay: x-space volumes `dim`, ay = A y with the
 * current scaling; slices alternate along dim_thick ('odd' = [::2], 'even' = [1::2], :430-445).
 *   out_dev[0] = sum_{x!=0} (x-ay)^2          (ll = 0.5 tau out[0])
 *   out_dev[1] = sum_even ay (x-ay)   out_dev[2] = sum_odd ay (x-ay)   (gr  = tau (out[1]-out[2]))
 *   out_dev[3] = sum_even ay^2        out_dev[4] = sum_odd ay^2        (Hes = tau (out[3]+out[4]))
 * float32 terms, float64 sums; out_dev: 5 doubles on the device. */
int unires_scaling_sums(const float *x, const float *ay, const int32_t dim[3], int32_t dim_thick,
                        double *out_dev, void *stream);

/* ------------------------------------------------------------------------
 * Stream marks: following a stream from the host by reading memory.
 * Not part of the reference (single process, unires/run.py); it serves the one-process-per-GPU batch
 * mode, where eight ranks share one host: every hipEventQuery / hipStreamQuery on running
 * work costs the runtime's signal thread CPU time (profiles/r05_host_profile.txt).  A mark is a 64-bit
 * word in mapped host memory; unires_mark_signal enqueues a one-thread kernel that stores `value` into it
 * once everything enqueued on `stream` before it has finished; unires_mark_read is a plain load (never
 * blocks, no runtime call).  Seeing the value orders NOTHING else for the host: results are still read
 * through a stream synchronisation / a blocking copy - the mark only tells when that will not wait.
 * ---------------------------------------------------------------------- */
typedef struct unires_mark unires_mark_t;
int unires_mark_create(unires_mark_t **out);
int unires_mark_destroy(unires_mark_t *mark);
int unires_mark_signal(unires_mark_t *mark, uint64_t value, void *stream);
int unires_mark_read(const unires_mark_t *mark, uint64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* UNIRES_HIP_H */
