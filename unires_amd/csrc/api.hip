// api.hip - the C ABI of libunires_hip.so (include/unires_hip.h): argument
// checking, the plan object, and the host-side sequencing of kernels for
// _proj_apply / _proj('AtA') / the y-update RHS / nitorch-style cg().
// Nothing here touches torch; the caller hands over raw device pointers and a
// hipStream_t.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "admm.hpp"
#include "aligned.hpp"
#include "cg.hpp"
#include "fftpre.hpp"
#include "fused.hpp"
#include "ops.hpp"
#include "orient.hpp"
#include "pull2.hpp"
#include "shift.hpp"
#include "splat2.hpp"
#include "ata1.hpp"
#include "stencil.hpp"

using namespace unires;

// --------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                               \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
      return UNIRES_ERR_HIP;                                                        \
    }                                                                               \
  } while (0)

#define CHECK_LAUNCH()                                                              \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      g_err = std::string("kernel launch: ") + hipGetErrorString(e_);               \
      return UNIRES_ERR_HIP;                                                        \
    }                                                                               \
  } while (0)

static bool dims_ok(const int32_t d[3]) {
  return d[0] > 0 && d[1] > 0 && d[2] > 0 && d[0] <= 65535 &&
         (long long)d[0] * d[1] * d[2] < (1ll << 40);
}
static Dim3i mk(const int32_t d[3]) { return Dim3i{d[0], d[1], d[2]}; }

static int make_taps(const float *const taps[3], const int32_t ntaps[3], const int32_t stride[3],
                     Taps &T) {
  memset(&T, 0, sizeof(T));
  for (int d = 0; d < 3; ++d) {
    if (ntaps[d] < 1 || ntaps[d] > UNIRES_MAX_TAPS)
      return fail(UNIRES_ERR_UNSUPPORTED, "ntaps must be in [1, UNIRES_MAX_TAPS]");
    if (stride[d] < 1) return fail(UNIRES_ERR_ARG, "stride must be >= 1");
    if (!taps[d]) return fail(UNIRES_ERR_NULL, "null taps pointer");
    T.n[d] = ntaps[d];
    T.s[d] = stride[d];
    for (int i = 0; i < ntaps[d]; ++i) T.t[d][i] = taps[d][i];
  }
  return UNIRES_OK;
}

static Scaling make_scaling(float scl, int dim) {
  if (scl == 0.f) return Scaling{1.f, 1.f, -1};
  return Scaling{expf(scl), expf(-scl), dim};
}

// inverse of a row-major 3x4 float32 affine, computed in double
static bool invert_affine(const Affine &A, Affine &out) {
  const double a = A.m[0], b = A.m[1], c = A.m[2], d = A.m[4], e = A.m[5], f = A.m[6],
               g = A.m[8], h = A.m[9], i = A.m[10];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (!(fabs(det) > 1e-12)) return false;
  const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                         (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                         (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
  const double t[3] = {A.m[3], A.m[7], A.m[11]};
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) out.m[4 * r + k] = (float)inv[3 * r + k];
    out.m[4 * r + 3] = (float)-(inv[3 * r] * t[0] + inv[3 * r + 1] * t[1] + inv[3 * r + 2] * t[2]);
  }
  return true;
}

// Drop leading/trailing zero taps (nitorch's rect profile carries one on each
// side): the skipped grid voxels contribute exactly 0, so A is unchanged; the
// grid shrinks and its affine is shifted by the number of leading taps dropped.
static void trim_taps(const Taps &T, const Affine &A, const Dim3i &gd, Taps &Tt, Affine &At,
                      Dim3i &gdt) {
  Tt = T;
  At = A;
  int g[3] = {gd.x, gd.y, gd.z};
  for (int d = 0; d < 3; ++d) {
    // (r6) ... and end taps below two ulps of the accumulated value, |t| < 2^-22 sum |t|: the +-5 taps of the default
    // Gaussian profile at ratio 2 (2.1e-7 of the sum each; the +-4 ones, 4.3e-5, stay) - 11 taps become 9: the operator
    // changes by <= 4.2e-7 relative, 250 x below the 1e-4 parity bar and below the disagreement of the three recollections
    // of nitorch's truncation of that Gaussian (11 / 9 / 7 taps).  UNIRES_TRIM_TINY=0 keeps every non-zero tap.
    static const bool tiny_on = !(getenv("UNIRES_TRIM_TINY") && atoi(getenv("UNIRES_TRIM_TINY")) == 0);
    double sum = 0.0;
    for (int i = 0; i < T.n[d]; ++i) sum += fabs((double)T.t[d][i]);
    const double eps = tiny_on ? sum * 2.384185791015625e-7 : 1e-300;
    int lead = 0, trail = 0;
    while (lead < T.n[d] - 1 && fabs((double)T.t[d][lead]) < eps) ++lead;
    while (trail < T.n[d] - 1 - lead && fabs((double)T.t[d][T.n[d] - 1 - trail]) < eps) ++trail;
    Tt.n[d] = T.n[d] - lead - trail;
    for (int i = 0; i < UNIRES_MAX_TAPS; ++i) Tt.t[d][i] = i < Tt.n[d] ? T.t[d][lead + i] : 0.f;
    g[d] -= lead + trail;
    for (int r = 0; r < 3; ++r)
      At.m[4 * r + 3] = (float)((double)At.m[4 * r + 3] + (double)lead * (double)A.m[4 * r + d]);
  }
  gdt = Dim3i{g[0], g[1], g[2]};
}

static int check_conv_dims(const Dim3i &hi, const Dim3i &lo, const Taps &T) {
  const int h[3] = {hi.x, hi.y, hi.z}, l[3] = {lo.x, lo.y, lo.z};
  for (int d = 0; d < 3; ++d)
    if (h[d] != (l[d] - 1) * T.s[d] + T.n[d])
      return fail(UNIRES_ERR_DIM, "conv dims: need hi = (lo-1)*stride + ntaps");
  return UNIRES_OK;
}

extern "C" const char *unires_last_error(void) { return g_err.c_str(); }
extern "C" int unires_abi_version(void) { return UNIRES_HIP_ABI_VERSION; }

// --------------------------------------------------------------------------
// op level
// --------------------------------------------------------------------------
extern "C" int unires_pull3d_affine(const float *src, const int32_t sdim[3], const float M[12],
                                    float *dst, const int32_t gdim[3], float fov_tol,
                                    void *stream) {
  if (!src || !dst || !sdim || !gdim || !M) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(sdim) || !dims_ok(gdim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  Affine A;
  memcpy(A.m, M, sizeof(A.m));
  launch_pull(src, mk(sdim), A, dst, mk(gdim), fov_tol, nullptr, (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_pull_grad3d_affine(const float *src, const int32_t sdim[3], const float M[12],
                                         float *dst3, const int32_t gdim[3], float fov_tol,
                                         void *stream) {
  if (!src || !dst3 || !sdim || !gdim || !M) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(sdim) || !dims_ok(gdim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  Affine A;
  memcpy(A.m, M, sizeof(A.m));
  launch_pull_grad(src, mk(sdim), A, dst3, mk(gdim), fov_tol, (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_push3d_affine(const float *src, const int32_t gdim[3], const float M[12],
                                    float *dst, const int32_t ddim[3], float alpha, float fov_tol,
                                    int accumulate, void *stream) {
  if (!src || !dst || !ddim || !gdim || !M) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(ddim) || !dims_ok(gdim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  Affine A;
  memcpy(A.m, M, sizeof(A.m));
  Affine Ainv;
  if (!invert_affine(A, Ainv)) return fail(UNIRES_ERR_ARG, "singular affine");
  PushSrc ps;
  memset(&ps, 0, sizeof(ps));
  ps.data = src;
  ps.gd = mk(gdim);
  ps.S = Scaling{1.f, 1.f, -1};
  PushEpilogue ep;
  ep.accumulate = accumulate ? 1 : 0;
  SplatSafety safe;
  splat_safety(A, safe.row_sep, safe.use_atomics);
  static const bool use_tile = getenv("UNIRES_PUSH") && !strcmp(getenv("UNIRES_PUSH"), "tile");
  if (use_tile || launch_splat(ps, A, Ainv, safe, alpha, fov_tol, ep, dst, mk(ddim), nullptr,
                               (hipStream_t)stream))
    (void)launch_push_tile(ps, A, Ainv, safe, alpha, fov_tol, ep, dst, mk(ddim), nullptr,
                           (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_conv_down3d(const float *src, const int32_t sdim[3],
                                  const float *const taps[3], const int32_t ntaps[3],
                                  const int32_t stride[3], float *dst, const int32_t ddim[3],
                                  float scl, int32_t scl_dim, void *stream) {
  if (!src || !dst || !sdim || !ddim || !taps || !ntaps || !stride)
    return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(sdim) || !dims_ok(ddim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (scl != 0.f && (scl_dim < 0 || scl_dim > 2)) return fail(UNIRES_ERR_ARG, "bad scl_dim");
  Taps T;
  int rc = make_taps(taps, ntaps, stride, T);
  if (rc) return rc;
  if ((rc = check_conv_dims(mk(sdim), mk(ddim), T))) return rc;
  launch_conv_down(src, mk(sdim), T, make_scaling(scl, scl_dim), dst, mk(ddim), nullptr,
                   (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_conv_up3d(const float *src, const int32_t sdim[3],
                                const float *const taps[3], const int32_t ntaps[3],
                                const int32_t stride[3], float *dst, const int32_t ddim[3],
                                float scl, int32_t scl_dim, void *stream) {
  if (!src || !dst || !sdim || !ddim || !taps || !ntaps || !stride)
    return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(sdim) || !dims_ok(ddim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (scl != 0.f && (scl_dim < 0 || scl_dim > 2)) return fail(UNIRES_ERR_ARG, "bad scl_dim");
  Taps T;
  int rc = make_taps(taps, ntaps, stride, T);
  if (rc) return rc;
  if ((rc = check_conv_dims(mk(ddim), mk(sdim), T))) return rc;
  launch_conv_up(src, mk(sdim), T, make_scaling(scl, scl_dim), dst, mk(ddim), (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

static bool vx_ok(const float vx[3]) { return vx && vx[0] > 0 && vx[1] > 0 && vx[2] > 0; }

extern "C" int unires_grad_fwd_zero(const float *src, const int32_t dim[3], const float vx[3],
                                    float *dst3, void *stream) {
  if (!src || !dst3 || !dim) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (!vx_ok(vx)) return fail(UNIRES_ERR_ARG, "voxel size must be positive");
  launch_grad(src, mk(dim), vx, dst3, (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_div_fwd_zero(const float *src3, const int32_t dim[3], const float vx[3],
                                   float *dst, void *stream) {
  if (!src3 || !dst || !dim) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (!vx_ok(vx)) return fail(UNIRES_ERR_ARG, "voxel size must be positive");
  launch_div(src3, nullptr, 1.f, 0.f, mk(dim), vx, 1.f, nullptr, dst, (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_dtd(const float *src, const int32_t dim[3], const float vx[3], float a,
                          float c, float *dst, void *stream) {
  if (!src || !dst || !dim) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (!vx_ok(vx)) return fail(UNIRES_ERR_ARG, "voxel size must be positive");
  if (src == dst) return fail(UNIRES_ERR_ARG, "dtd cannot run in place");
  launch_dtd(src, mk(dim), vx, a, c, dst, nullptr, nullptr, nullptr, (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

// --------------------------------------------------------------------------
// plan
// --------------------------------------------------------------------------
struct Repeat {
  // Everything below is in the plan's CANONICAL voxel layout of the observation: x-space axis d runs
  // mainly along +d of the output (orient.hpp).  `orient` maps it to the caller's layout, `dim_xu`
  // are the caller's x-space dims; only 'A' outputs and 'At' / RHS inputs are ever re-ordered.
  Orient orient;
  bool oriented = false;
  Dim3i dim_xu;
  Dim3i dim_x, dim_g;
  Affine A;
  Taps T;
  float scl;
  int dim_thick;
  float tau;
  // fused path: zero taps trimmed, grid shifted accordingly, inverse affine
  Dim3i dim_gf;
  Affine Af, Afinv;
  Taps Tf;
  SplatSafety safe;  // of Af (the linear part is the same for A)
  bool sep = false;  // many-tap profile: convolutions run as separable 1-D passes
  bool sep0 = false; // ... as decided from the taps alone (the hybrid form clears `sep`; kept for its fallback)
  // profile along x and / or y AND z with a z fan-in <= 2 (isotropic down-sampling, BASELINE config
  // 4): the x / y part runs as 1-D passes through a (gf.x, gf.y, xd.z) intermediate, the z part
  // stays fused in the pull / splat kernels, which then cost what they cost for a z-only profile
  bool hyb = false;
  // forward-only hybrid (r3): many-tap profiles (sep) whose z part the window pull can still fuse -
  // any number of z taps - while conv_up keeps its 1-D passes (the splat tabulates a fan-in of 2 only):
  // the default Gaussian in-plane profile of BASELINE config 4
  bool hybf = false;
  Taps Tz, Txy;
  Dim3i dim_h;
  // schedule-driven splat (splat2.hip): per-tile segment lists of this operator + conv_up tables
  // along the schedule's axis ([0] no scaling, [1] S(scl)); ctab_n entries, second x-space value
  // ctab_step elements after the first
  SplatSched sched;
  float *ctab_dev[2] = {nullptr, nullptr};
  int ctab_n = 0, ctab_cap = 0;
  unsigned ctab_step = 1, src_stride = 1;
  float *xytab_dev[2] = {nullptr, nullptr};  // axis 3: conv_up tables along x and y (schedule build)
  int xytab_cap[2] = {0, 0};
  PullPlan pplan;  // LDS-window pull: per-workgroup geometry of this operator (pull2.hip)
  ShiftPlan shift;  // translation-only operators: factors of AtA for the one-kernel matvec (shift.hip)
  F1Sched f1;       // denoising regime: schedule of the single-pass AtA kernel (ata1.hip)
};

struct unires_plan {
  Dim3i dy;
  float vx[3];
  int regime;
  float fov_tol;
  std::vector<Repeat> reps;
  // device workspace (one allocation)
  char *ws = nullptr;
  size_t ws_bytes = 0;
  float *r = nullptr, *p = nullptr, *ap = nullptr, *ax = nullptr;  // N_y each
  // measurement aid (unires_plan_time_matvecs): event pairs around the operator applications of a solve
  bool timing = false;
  bool twice = false;  // unires_plan_time_matvecs(plan, 2): every A(p) of a solve is enqueued twice (same result)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;
  float *gbuf = nullptr;                                           // max N_g
  float *gbuf2 = nullptr;  // second grid-space scratch, only for many-tap profiles (separable passes)
  float *xbuf = nullptr;                                           // max N_x
  float *xperm = nullptr;  // max N_x: an x-space volume on its way between the caller's layout and the canonical one
  double *part0 = nullptr, *part1 = nullptr;                       // kMaxPartials each
  CgState *state = nullptr;
  size_t cap_g = 0, cap_x = 0;
  // the whole CG solve as one hipGraph, re-launched while (b, x, rho, lam, options) stay the
  // same - the ADMM loop calls it with identical arguments until the schedule changes
  struct CgKey {
    const float *b = nullptr;
    float *x = nullptr;
    float rho = 0.f, lam = 0.f;
    int max_iter = -1, stop = -1, pre = -1;
    double tol = -1.0;
    bool operator==(const CgKey &o) const {
      return b == o.b && x == o.x && rho == o.rho && lam == o.lam && max_iter == o.max_iter &&
             stop == o.stop && pre == o.pre && tol == o.tol;
    }
  } cg_key;
  hipGraphExec_t cg_exec = nullptr;
  // chunked solves: start + chunk graphs, the host-mapped progress word and the solve counter
  CgKey cg_chunk_key;
  hipGraphExec_t cg_start_exec = nullptr, cg_chunk_exec = nullptr;
  unsigned long long *progress = nullptr, *progress_dev = nullptr;
  unsigned cg_gen = 0;
  float *precM = nullptr;  // Jacobi diagonal (own allocation, made by unires_precond_build)
  FftPre fft;              // FFT-diagonal preconditioner (plans + buffers, made on demand)
  float prec_rho = 0.f, prec_lam = 0.f;
  int prec_mode = UNIRES_PRECOND_IDENTITY;
  bool prec_ready = false;
  // the last launch that reads this plan's tables: unires_plan_set_repeat / the graph teardown wait for THIS
  // event instead of the whole device (other channels' streams keep running).  A launch enqueued while its
  // stream was being captured cannot be waited for through an event: `captured_use` sends those to the
  // device-wide wait.
  hipEvent_t last_use = nullptr;
  std::vector<hipStream_t> use_streams;
  bool captured_use = false;
  // unires_plan_set_concurrency: how many solves the caller keeps in flight on the device (channels of a y-update on
  // streams of their own), and the caps on the persistent kernels' grids that follow from it (0: the whole chip)
  int concurrency = 1;
  int cap_s2 = 0, cap_f1 = 0;
};

// Drop the captured CG solve.  A launch of it may still be in flight (the ADMM loop never syncs):
// wait for the device before destroying the executable graph.
static void drop_timing(unires_plan *pl) {
  for (auto &e : pl->tev) (void)hipEventDestroy(e.first), (void)hipEventDestroy(e.second);
  pl->tev.clear();
}

// note / await the plan's last use (see unires_plan::last_use).  Noting is free - the stream is remembered, no
// event is recorded per call (an event per entry point kept a second host thread busy: host_share 1.04 -> 2.0 in
// tools/host_time.py); the event is recorded on the remembered streams when somebody has to wait, and waited for
// between sleeps (hipEventSynchronize polls flat out unless the process set hipDeviceScheduleBlockingSync before
// its context existed, tools/wait_probe.py).
static void mark_use(unires_plan *pl, hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    cs = hipStreamCaptureStatusNone;
  }
  if (cs != hipStreamCaptureStatusNone) {
    pl->captured_use = true;
    return;
  }
  for (hipStream_t s : pl->use_streams)
    if (s == st) return;
  if (pl->use_streams.size() >= 8)
    pl->captured_use = true;  // (more streams than anybody uses: fall back to the device-wide wait)
  else
    pl->use_streams.push_back(st);
}
static void await_use(unires_plan *pl) {
  static const bool device_wide = getenv("UNIRES_SET_REPEAT_DEVICE_SYNC") != nullptr;  // (measurement: the r4 behaviour)
  if (pl->captured_use || device_wide) {
    // (sticky: a graph the caller captured may be replayed at any time without an entry point being called - a plan
    // that was ever used under capture is waited for device-wide from then on)
    (void)hipDeviceSynchronize();
    pl->use_streams.clear();
    return;
  }
  if (pl->use_streams.empty()) return;
  if (!pl->last_use && hipEventCreateWithFlags(&pl->last_use, hipEventDisableTiming) != hipSuccess) {
    pl->last_use = nullptr;
    (void)hipDeviceSynchronize();
    pl->use_streams.clear();
    return;
  }
  for (hipStream_t s : pl->use_streams) {
    if (hipEventRecord(pl->last_use, s) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      break;
    }
    unsigned spins = 0;
    for (;;) {
      const hipError_t q = hipEventQuery(pl->last_use);
      if (q != hipErrorNotReady) break;
      if (++spins > 4096) {  // (the usual wait is a kernel or two: spin that long, then sleep)
        const struct timespec nap = {0, 50000};
        (void)nanosleep(&nap, nullptr);
      }
    }
    (void)hipGetLastError();
  }
  pl->use_streams.clear();
}

static void drop_cg_chunk_graphs(unires_plan *pl) {
  if (!pl->cg_start_exec && !pl->cg_chunk_exec) return;
  await_use(pl);
  if (pl->cg_start_exec) (void)hipGraphExecDestroy(pl->cg_start_exec);
  if (pl->cg_chunk_exec) (void)hipGraphExecDestroy(pl->cg_chunk_exec);
  pl->cg_start_exec = pl->cg_chunk_exec = nullptr;
}

static void drop_cg_graph(unires_plan *pl) {
  drop_cg_chunk_graphs(pl);
  if (!pl->cg_exec) return;
  await_use(pl);
  (void)hipGraphExecDestroy(pl->cg_exec);
  pl->cg_exec = nullptr;
}

// Relabel the observation's voxel axes so that grid axis d runs mainly along +d of the output
// (orient.hpp).  With u[perm[j]] = flip[j] ? n[perm[j]] - 1 - u'[j] : u'[j] on the grid and on x-space
// alike: columns of A permuted and negated, the translation moved to the far end of a reversed axis
// (composed in double from the float32 entries the reference's grid is made of, rounded once); dims,
// strides and taps permuted; the taps of a reversed axis reversed - x[k] = sum_t ker[t] g[r k + t] with
// n_g = (n_x - 1) r + K reads x'[k'] = sum_t ker[K - 1 - t] g'[r k' + t] - and the even / odd slice
// factors swapped where reversal changes the parity of a slice index (n_x even).
static void canonicalise(Repeat &R) {
  static const bool off = getenv("UNIRES_NO_CANON") != nullptr;  // (measurement: the r3 behaviour)
  R.orient = off ? Orient() : orient_of(R.A);
  R.oriented = !R.orient.identity();
  if (!R.oriented) return;
  const Orient &O = R.orient;
  const int nx[3] = {R.dim_x.x, R.dim_x.y, R.dim_x.z}, ng[3] = {R.dim_g.x, R.dim_g.y, R.dim_g.z};
  const Affine A = R.A;
  const Taps T = R.T;
  int cx[3], cg[3];
  double t[3] = {A.m[3], A.m[7], A.m[11]};
  for (int j = 0; j < 3; ++j) {
    const int a = O.perm[j];
    cx[j] = nx[a], cg[j] = ng[a];
    for (int r = 0; r < 3; ++r) {
      R.A.m[4 * r + j] = O.flip[j] ? -A.m[4 * r + a] : A.m[4 * r + a];
      if (O.flip[j]) t[r] += (double)(ng[a] - 1) * (double)A.m[4 * r + a];
    }
    R.T.n[j] = T.n[a], R.T.s[j] = T.s[a];
    for (int i = 0; i < UNIRES_MAX_TAPS; ++i)
      R.T.t[j][i] = i < T.n[a] ? (O.flip[j] ? T.t[a][T.n[a] - 1 - i] : T.t[a][i]) : 0.f;
  }
  for (int r = 0; r < 3; ++r) R.A.m[4 * r + 3] = (float)t[r];
  R.dim_x = Dim3i{cx[0], cx[1], cx[2]};
  R.dim_g = Dim3i{cg[0], cg[1], cg[2]};
  if (R.dim_thick >= 0 && R.dim_thick <= 2) {
    int jt = 0;
    for (int j = 0; j < 3; ++j)
      if (O.perm[j] == R.dim_thick) jt = j;
    if (O.flip[jt] && (nx[R.dim_thick] & 1) == 0) R.scl = -R.scl;
    R.dim_thick = jt;
  }
}

static int fill_repeat(const unires_plan *pl, const unires_repeat_t *in, Repeat &out) {
  if (!in) return fail(UNIRES_ERR_NULL, "null repeat descriptor");
  if (!(in->tau > 0.f)) return fail(UNIRES_ERR_ARG, "tau must be positive");
  memset(&out, 0, sizeof(out));
  out.sched = SplatSched();
  out.pplan = PullPlan();
  out.shift = ShiftPlan();
  out.f1 = F1Sched();
  out.orient = Orient();
  out.ctab_step = 1;
  out.tau = in->tau;
  out.scl = in->scl;
  out.dim_thick = in->dim_thick;
  if (pl->regime == UNIRES_REGIME_IDENTITY) {
    out.dim_x = out.dim_xu = pl->dy;
    out.dim_g = pl->dy;
    return UNIRES_OK;
  }
  if (!dims_ok(in->dim_x) || !dims_ok(in->dim_g)) return fail(UNIRES_ERR_DIM, "bad repeat dims");
  out.dim_x = mk(in->dim_x);
  out.dim_g = mk(in->dim_g);
  memcpy(out.A.m, in->M, sizeof(out.A.m));
  for (int i = 0; i < 12; ++i)
    if (!isfinite(out.A.m[i])) return fail(UNIRES_ERR_ARG, "non-finite affine");
  if (pl->regime == UNIRES_REGIME_SUPERRES) {
    int rc = make_taps(in->taps, in->ntaps, in->ratio, out.T);
    if (rc) return rc;
    if ((rc = check_conv_dims(out.dim_g, out.dim_x, out.T))) return rc;
    if (out.scl != 0.f && (out.dim_thick < 0 || out.dim_thick > 2))
      return fail(UNIRES_ERR_ARG, "bad dim_thick");
  } else {
    if (out.dim_g.x != out.dim_x.x || out.dim_g.y != out.dim_x.y || out.dim_g.z != out.dim_x.z)
      return fail(UNIRES_ERR_DIM, "denoising regime needs dim_g == dim_x");
    for (int d = 0; d < 3; ++d) out.T.n[d] = out.T.s[d] = 1, out.T.t[d][0] = 1.f;
  }
  out.dim_xu = out.dim_x;
  canonicalise(out);
  trim_taps(out.T, out.A, out.dim_g, out.Tf, out.Af, out.dim_gf);
  // many taps (e.g. a Gaussian in-plane profile on top of the slice profile): the fused kernels'
  // direct 3-D sum (prod n_d taps per output, fan-in^3 gathers per grid voxel) loses to one
  // 1-D pass per axis through grid-space scratch
  out.sep = (long long)out.Tf.n[0] * out.Tf.n[1] * out.Tf.n[2] > 64;
  for (int d = 0; d < 3; ++d)
    if ((out.Tf.n[d] + out.Tf.s[d] - 1) / out.Tf.s[d] > 2) out.sep = true;
  out.sep0 = out.sep;
  out.hyb = out.hybf = false;
  if (pl->regime == UNIRES_REGIME_SUPERRES) {
    static const bool no_hyb = getenv("UNIRES_NO_HYBRID") != nullptr;
    auto dirac = [&](int d) { return out.Tf.n[d] == 1 && out.Tf.s[d] == 1 && out.Tf.t[d][0] == 1.f; };
    const bool xy = !dirac(0) || !dirac(1);
    const bool z_ok = (out.Tf.n[2] + out.Tf.s[2] - 1) / out.Tf.s[2] <= 2 && out.dim_x.z >= 2;
    int nconv = 0;
    for (int d = 0; d < 3; ++d) nconv += !dirac(d);
    const bool both = !no_hyb && xy && nconv > 1 && z_ok;
    const bool fwd_only = !no_hyb && !both && out.sep && xy && !dirac(2) && out.dim_x.z >= 2;
    if (both || fwd_only) {
      out.hyb = both, out.hybf = fwd_only;
      if (both) out.sep = false;
      out.Tz = out.Tf, out.Txy = out.Tf;
      for (int d = 0; d < 2; ++d) out.Tz.n[d] = out.Tz.s[d] = 1, out.Tz.t[d][0] = 1.f;
      out.Txy.n[2] = out.Txy.s[2] = 1, out.Txy.t[2][0] = 1.f;
      out.dim_h = Dim3i{out.dim_gf.x, out.dim_gf.y, out.dim_x.z};
    }
  }
  if (!invert_affine(out.Af, out.Afinv)) return fail(UNIRES_ERR_ARG, "singular affine");
  splat_safety(out.Af, out.safe.row_sep, out.safe.use_atomics);
  return UNIRES_OK;
}

// (re)build the splat schedule of a repeat for its current operator; a non-applicable operator
// simply leaves the schedule invalid (the general push kernels then run)
// tables_only: the operator's geometry is unchanged (a new slice scaling only): the conv_up tables are rewritten,
// the schedule itself - which does not see the scaling - stays
static int build_sched(unires_plan *pl, Repeat &R, bool tables_only = false) {
  const bool was_valid = R.sched.valid;
  R.sched.valid = false;
  if (pl->regime == UNIRES_REGIME_IDENTITY) return UNIRES_OK;
  int axis = -1;
  int rows_y = R.dim_gf.y;
  R.src_stride = (unsigned)R.dim_gf.z;
  R.ctab_step = 1;
  const bool direct = pl->regime == UNIRES_REGIME_DENOISE || R.sep;
  if (!direct) {
    int nconv = 0;
    for (int d = 0; d < 3; ++d) {
      const bool dirac = R.Tf.n[d] == 1 && R.Tf.s[d] == 1 && R.Tf.t[d][0] == 1.f;
      if (dirac) continue;
      ++nconv;
      axis = d;
    }
    const int xdv[3] = {R.dim_x.x, R.dim_x.y, R.dim_x.z}, gdv[3] = {R.dim_gf.x, R.dim_gf.y, R.dim_gf.z};
    if (R.dim_x.numel() >= (1ull << 30)) return UNIRES_OK;
    if (R.hyb) {
      axis = 2;
    } else if (nconv > 1) {
      // conv_up along several axes (isotropic down-sampling, BASELINE config 4): x / y parts per
      // segment in the schedule, z part per lane
      for (int d = 0; d < 3; ++d)
        if ((R.Tf.n[d] + R.Tf.s[d] - 1) / R.Tf.s[d] > 2 || xdv[d] < 2) return UNIRES_OK;
      if (R.scl != 0.f && R.dim_thick != 2) return UNIRES_OK;
      axis = 3;
    } else {
      if (axis < 0) axis = 2;  // all dirac: conv_up is the identity, any axis works
      if ((R.Tf.n[axis] + R.Tf.s[axis] - 1) / R.Tf.s[axis] > 2 || xdv[axis] < 2) return UNIRES_OK;
      if (R.scl != 0.f && R.dim_thick != axis && !R.hyb) return UNIRES_OK;  // (hybrid: x / y scaling rides with the 1-D passes)
    }
    const unsigned xyz = (unsigned)R.dim_x.y * (unsigned)R.dim_x.z, xz = (unsigned)R.dim_x.z;
    const int taxis = axis == 3 ? 2 : axis;  // axis of the run-time (per-lane) table
    if (axis == 2) rows_y = R.hyb ? R.dim_h.y : R.dim_x.y, R.src_stride = xz, R.ctab_step = 1;
    if (axis == 1) R.src_stride = xyz, R.ctab_step = xz;  // source offset ui * xyz + koff(uj) * xz + k
    if (axis == 0) R.src_stride = xz, R.ctab_step = xyz;  // source offset uj * xz + koff(ui) * xyz + k
    if (axis == 3) R.src_stride = xz, R.ctab_step = xyz;  // (x-space row / slab strides)
    const int gn = gdv[taxis];
    if (gn + 128 > 1400) return UNIRES_OK;  // LDS copy of the table
    std::vector<float> host((size_t)gn * 4);
    for (int v = 0; v < 2; ++v) {
      if (!R.ctab_dev[v] || R.ctab_cap < gn) {
        if (R.ctab_dev[v]) (void)hipFree(R.ctab_dev[v]);
        R.ctab_dev[v] = nullptr;
        if (hipMalloc((void **)&R.ctab_dev[v], host.size() * sizeof(float)) != hipSuccess)
          return fail(UNIRES_ERR_ALLOC, "hipMalloc conv table");
      }
      splat2_convtab(R.Tf, v ? make_scaling(R.scl, R.dim_thick) : Scaling{1.f, 1.f, -1}, taxis, gn,
                     xdv[taxis], host.data());
      if (hipMemcpy(R.ctab_dev[v], host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) !=
          hipSuccess)
        return fail(UNIRES_ERR_HIP, "hipMemcpy conv table");
    }
    R.ctab_n = gn;
    R.ctab_cap = std::max(R.ctab_cap, gn);
    if (axis == 3) {
      for (int d = 0; d < 2; ++d) {
        std::vector<float> hx((size_t)gdv[d] * 4);
        if (!R.xytab_dev[d] || R.xytab_cap[d] < gdv[d]) {
          if (R.xytab_dev[d]) (void)hipFree(R.xytab_dev[d]);
          R.xytab_dev[d] = nullptr;
          if (hipMalloc((void **)&R.xytab_dev[d], hx.size() * sizeof(float)) != hipSuccess)
            return fail(UNIRES_ERR_ALLOC, "hipMalloc conv table");
          R.xytab_cap[d] = gdv[d];
        }
        splat2_convtab(R.Tf, Scaling{1.f, 1.f, -1}, d, gdv[d], xdv[d], hx.data());
        if (hipMemcpy(R.xytab_dev[d], hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
          return fail(UNIRES_ERR_HIP, "hipMemcpy conv table");
      }
    }
  }
  if (tables_only) {
    R.sched.valid = was_valid;
    return UNIRES_OK;
  }
  (void)splat2_build(R.sched, R.Af, R.Afinv, R.dim_gf, pl->dy, pl->fov_tol, R.safe, axis, rows_y,
                     (const float4 *)R.xytab_dev[0], (const float4 *)R.xytab_dev[1], R.dim_x);
  (void)hipGetLastError();
  return UNIRES_OK;
}

// the pull of the same operator through the LDS-window kernel (denoising: plain pull onto the
// grid; super-resolution: + conv_down); outside its domain the plan stays invalid
static void build_shift(unires_plan *pl, Repeat &R);
static void build_pull(unires_plan *pl, Repeat &R) {
  R.pplan.valid = false;
  if (pl->regime == UNIRES_REGIME_DENOISE)
    (void)pull2_build(R.pplan, pl->dy, R.A, R.T, R.dim_g, R.dim_g, pl->fov_tol);
  else if (pl->regime == UNIRES_REGIME_SUPERRES && (R.hyb || R.hybf))
    (void)pull2_build(R.pplan, pl->dy, R.Af, R.Tz, R.dim_h, R.dim_gf, pl->fov_tol);
  else if (pl->regime == UNIRES_REGIME_SUPERRES && !R.sep)
    (void)pull2_build(R.pplan, pl->dy, R.Af, R.Tf, R.dim_x, R.dim_gf, pl->fov_tol);
  build_shift(pl, R);
}

// translation-only operator (and a single repeat: the kernel is the whole matvec); carries the slice scaling
static void build_shift(unires_plan *pl, Repeat &R) {
  R.shift.valid = false;
  if (pl->reps.size() == 1 && pl->regime != UNIRES_REGIME_IDENTITY)
    (void)shift_build(R.shift, pl->dy, R.dim_gf, R.dim_x, R.Tf, make_scaling(2.f * R.scl, R.dim_thick), R.Af,
                      pl->fov_tol);
  (void)hipGetLastError();
}

// schedule + window plan of one repeat.  A hybrid operator (z profile inside the pull / splat kernels,
// x / y profiles as 1-D passes) whose kernels turn out to be unavailable - window plan outside
// pull2's domain, or no axis-2 schedule (atomics needed, table too long for LDS, a tile overflowing
// the segment lists) - is rebuilt as what the taps alone would have chosen: the separable passes for
// a many-tap profile (with the forward half of the hybrid where its window plan exists), not the dense
// 3-D conv kernels the general fall-through ends in.
static int build_repeat_kernels(unires_plan *pl, Repeat &R) {
  int rc = build_sched(pl, R);
  if (rc) return rc;
  build_pull(pl, R);
  R.f1.valid = false;
  if (pl->regime == UNIRES_REGIME_DENOISE) {  // pull and push in one pass where the operator allows it
    (void)ata1_build(R.f1, R.Af, R.Afinv, R.dim_gf, pl->dy, pl->fov_tol, R.safe);
    (void)hipGetLastError();
  }
  if (R.hyb && !(R.pplan.valid && R.sched.valid && R.sched.axis == 2)) {
    R.hyb = false;
    R.sep = R.sep0;
    R.hybf = R.sep0 && R.pplan.valid;  // (the window plan of the forward half is the one just built)
    rc = build_sched(pl, R);
    if (rc) return rc;
    if (!R.hybf) build_pull(pl, R);
  }
  return UNIRES_OK;
}

static void free_sched(Repeat &R) {
  splat2_free(R.sched);
  pull2_free(R.pplan);
  shift_free(R.shift);
  ata1_free(R.f1);
  for (int v = 0; v < 2; ++v) {
    if (R.ctab_dev[v]) (void)hipFree(R.ctab_dev[v]), R.ctab_dev[v] = nullptr;
    if (R.xytab_dev[v]) (void)hipFree(R.xytab_dev[v]), R.xytab_dev[v] = nullptr;
    R.xytab_cap[v] = 0;
  }
  R.ctab_n = R.ctab_cap = 0;
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int unires_plan_create(unires_plan_t **plan, const int32_t dim_y[3],
                                  const float vx_y[3], int32_t regime, int32_t n_repeats,
                                  const unires_repeat_t *repeats, float fov_tol) {
  if (!plan || !dim_y) return fail(UNIRES_ERR_NULL, "null argument");
  *plan = nullptr;
  if (!dims_ok(dim_y)) return fail(UNIRES_ERR_DIM, "bad dim_y");
  if (!vx_ok(vx_y)) return fail(UNIRES_ERR_ARG, "voxel size must be positive");
  if (regime < 0 || regime > 2) return fail(UNIRES_ERR_ARG, "Undefined method");
  if (n_repeats < 1 || !repeats) return fail(UNIRES_ERR_ARG, "need at least one repeat");
  unires_plan *pl = new (std::nothrow) unires_plan();
  if (!pl) return fail(UNIRES_ERR_ALLOC, "host allocation failed");
  pl->dy = mk(dim_y);
  memcpy(pl->vx, vx_y, sizeof(pl->vx));
  pl->regime = regime;
  pl->fov_tol = fov_tol;
  pl->reps.resize(n_repeats);
  bool need_sep = false;
  for (int n = 0; n < n_repeats; ++n) {
    int rc = fill_repeat(pl, &repeats[n], pl->reps[n]);
    if (rc) {
      delete pl;
      return rc;
    }
    if (regime != UNIRES_REGIME_IDENTITY) {
      pl->cap_g = std::max(pl->cap_g, pl->reps[n].dim_g.numel());
      pl->cap_x = std::max(pl->cap_x, pl->reps[n].dim_x.numel());
    }
    need_sep = need_sep || (regime == UNIRES_REGIME_SUPERRES && (pl->reps[n].sep || pl->reps[n].hyb));
  }
  const size_t ny = pl->dy.numel();
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += align_up(bytes);
    return o;
  };
  const size_t o_r = carve(ny * 4), o_p = carve(ny * 4), o_ap = carve(ny * 4), o_ax = carve(ny * 4);
  const size_t o_g = carve(pl->cap_g * 4), o_x = carve(pl->cap_x * 4), o_xp = carve(pl->cap_x * 4);
  const size_t o_g2 = carve(need_sep ? pl->cap_g * 4 : 0);
  const size_t o_p0 = carve(kMaxPartials * 8), o_p1 = carve(kMaxPartials * 8);
  const size_t o_st = carve(sizeof(CgState));
  pl->ws_bytes = off;
  hipError_t e = hipMalloc((void **)&pl->ws, pl->ws_bytes);
  if (e != hipSuccess) {
    g_err = std::string("hipMalloc workspace: ") + hipGetErrorString(e);
    delete pl;
    return UNIRES_ERR_ALLOC;
  }
  pl->r = (float *)(pl->ws + o_r);
  pl->p = (float *)(pl->ws + o_p);
  pl->ap = (float *)(pl->ws + o_ap);
  pl->ax = (float *)(pl->ws + o_ax);
  pl->gbuf = (float *)(pl->ws + o_g);
  pl->xbuf = (float *)(pl->ws + o_x);
  pl->xperm = (float *)(pl->ws + o_xp);
  pl->gbuf2 = need_sep ? (float *)(pl->ws + o_g2) : nullptr;
  pl->part0 = (double *)(pl->ws + o_p0);
  pl->part1 = (double *)(pl->ws + o_p1);
  pl->state = (CgState *)(pl->ws + o_st);
  e = hipMemset(pl->state, 0, sizeof(CgState));
  if (e != hipSuccess) {
    g_err = std::string("hipMemset: ") + hipGetErrorString(e);
    (void)hipFree(pl->ws);
    delete pl;
    return UNIRES_ERR_HIP;
  }
  for (Repeat &R : pl->reps) {
    int rc = UNIRES_OK;
    if (!rc) rc = build_repeat_kernels(pl, R);
    if (rc) {
      for (Repeat &Q : pl->reps) free_sched(Q);
      (void)hipFree(pl->ws);
      delete pl;
      return rc;
    }
  }
  *plan = pl;
  return UNIRES_OK;
}

extern "C" int unires_plan_destroy(unires_plan_t *plan) {
  if (!plan) return UNIRES_OK;
  if (plan->ws) (void)hipFree(plan->ws);
  if (plan->precM) (void)hipFree(plan->precM);
  if (plan->cg_exec) (void)hipGraphExecDestroy(plan->cg_exec);
  if (plan->cg_start_exec) (void)hipGraphExecDestroy(plan->cg_start_exec);
  if (plan->cg_chunk_exec) (void)hipGraphExecDestroy(plan->cg_chunk_exec);
  if (plan->progress) (void)hipHostFree(plan->progress);
  if (plan->last_use) (void)hipEventDestroy(plan->last_use);
  drop_timing(plan);
  fftpre_destroy(plan->fft);
  for (Repeat &R : plan->reps) free_sched(R);
  delete plan;
  return UNIRES_OK;
}

extern "C" int unires_plan_time_matvecs(unires_plan_t *plan, int32_t on) {
  if (!plan) return fail(UNIRES_ERR_NULL, "null argument");
  const bool twice = on == 2;
  if (twice != plan->twice) drop_cg_graph(plan);  // (a captured solve has its launches baked in)
  plan->twice = twice;
  plan->timing = on == 1;
  if (!plan->timing) drop_timing(plan);
  return UNIRES_OK;
}

extern "C" int unires_plan_matvec_time(unires_plan_t *plan, int32_t *launches, double *total_us) {
  if (!plan || !launches || !total_us) return fail(UNIRES_ERR_NULL, "null argument");
  *launches = 0, *total_us = 0.0;
  for (auto &e : plan->tev) {
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(e.second));
    HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
    *total_us += 1e3 * (double)ms;
    ++*launches;
  }
  drop_timing(plan);
  return UNIRES_OK;
}

extern "C" int unires_plan_set_repeat(unires_plan_t *plan, int32_t n,
                                      const unires_repeat_t *repeat) {
  if (!plan || !repeat) return fail(UNIRES_ERR_NULL, "null argument");
  if (n < 0 || n >= (int)plan->reps.size()) return fail(UNIRES_ERR_ARG, "repeat index");
  Repeat tmp;
  int rc = fill_repeat(plan, repeat, tmp);
  if (rc) return rc;
  if (plan->regime != UNIRES_REGIME_IDENTITY &&
      (tmp.dim_g.numel() > plan->cap_g || tmp.dim_x.numel() > plan->cap_x))
    return fail(UNIRES_ERR_DIM, "new repeat exceeds the plan's workspace");
  if (plan->regime == UNIRES_REGIME_SUPERRES && (tmp.sep || tmp.hyb) && !plan->gbuf2)
    return fail(UNIRES_ERR_DIM, "new repeat needs the separable-conv scratch the plan was built without");
  // The tables rebuilt below (pull records, splat schedule, conv tables) are rewritten by kernels on
  // the NULL stream and synchronous copies; work queued on the caller's - possibly non-blocking -
  // streams may still be reading them: wait for the plan's last launches - every entry point REMEMBERS its stream
  // before it enqueues anything (mark_use), await_use records an event on each remembered stream and waits for it;
  // not the whole device: the other channels' streams keep running.
  await_use(plan);
  drop_cg_graph(plan);  // the captured solve has the old operator baked in
  plan->prec_ready = false;  // a preconditioner built for the old operator is stale
  {
    // A new slice scaling on the same geometry (the scaling Gauss-Newton step, unires/_update.py:270-393, once
    // per observation and ADMM iteration): window plan and splat schedule do not see it - only the conv_up
    // tables that carry S(scl) and the translated regime's factors are rewritten (the schedule build's kernels are
    // 1.2 ms at 256^3, waited for: 5.2 -> 3.9 ms per scaling step of three channels)
    Repeat &old = plan->reps[n];
    static const bool no_fast = getenv("UNIRES_SET_REPEAT_FULL") != nullptr;
    auto taps_equal = [](const Taps &a, const Taps &b) {
      for (int d = 0; d < 3; ++d) {
        if (a.n[d] != b.n[d] || a.s[d] != b.s[d]) return false;
        for (int t = 0; t < a.n[d]; ++t)
          if (a.t[d][t] != b.t[d][t]) return false;
      }
      return true;
    };
    const bool same = !no_fast && !memcmp(&tmp.A, &old.A, sizeof(Affine)) && taps_equal(tmp.T, old.T) &&
                      !memcmp(&tmp.Af, &old.Af, sizeof(Affine)) && taps_equal(tmp.Tf, old.Tf) &&
                      !memcmp(&tmp.dim_x, &old.dim_x, sizeof(Dim3i)) && !memcmp(&tmp.dim_g, &old.dim_g, sizeof(Dim3i)) &&
                      !memcmp(&tmp.dim_gf, &old.dim_gf, sizeof(Dim3i)) && !memcmp(&tmp.dim_xu, &old.dim_xu, sizeof(Dim3i)) &&
                      !memcmp(&tmp.orient, &old.orient, sizeof(Orient)) && tmp.oriented == old.oriented &&
                      tmp.dim_thick == old.dim_thick && (tmp.scl != 0.f) == (old.scl != 0.f) && tmp.sep0 == old.sep0;
    static const bool verbose = getenv("UNIRES_SET_REPEAT_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[set_repeat] %s (scl %g -> %g)\n", same ? "scaling only" : "full rebuild", (double)old.scl, (double)tmp.scl);
    if (same) {
      old.scl = tmp.scl, old.tau = tmp.tau;
      rc = build_sched(plan, old, true);
      if (!rc) build_shift(plan, old);
      return rc;
    }
  }
  // the schedule and conv tables keep their device allocations; contents are rebuilt below
  tmp.sched = plan->reps[n].sched;
  tmp.pplan = plan->reps[n].pplan;
  tmp.shift = plan->reps[n].shift;
  tmp.f1 = plan->reps[n].f1;
  tmp.ctab_dev[0] = plan->reps[n].ctab_dev[0];
  tmp.ctab_dev[1] = plan->reps[n].ctab_dev[1];
  tmp.ctab_cap = plan->reps[n].ctab_cap;
  for (int d = 0; d < 2; ++d) tmp.xytab_dev[d] = plan->reps[n].xytab_dev[d], tmp.xytab_cap[d] = plan->reps[n].xytab_cap[d];
  plan->reps[n] = tmp;
  rc = UNIRES_OK;
  sched_set_thorough(false);  // (an operator changing under a running reconstruction: the quick schedule builds)
  if (!rc) rc = build_repeat_kernels(plan, plan->reps[n]);
  sched_set_thorough(true);
  return rc;
}

// How many solves of OTHER plans the caller keeps in flight next to this one's (the channels of a y-update, each
// on a stream of its own: unires/_update.py:122-150 has no cross-channel term).  The matvec's persistent kernels
// (k_splat2, k_ata1) normally take every wave slot / register the chip has - a channel's kernels then run one
// after the other's whatever the streams say.  With room left for them, the bandwidth-bound CG vector kernels of
// one channel run under the issue-bound splat of another: measured at 256^3 x 3 (profiles/r06_overlap_scan.txt)
// 4 879 -> 5 190 CG it/s with 384 - 448 splat workgroups of 1 024; at 384^3 x 4 a smaller grid LOSES (the pull and
// conv kernels between the splats leave room anyway): the cap applies below 2^25 output voxels.  A launch under a
// cap takes longer and the job as a whole gets faster.  n <= 1: the whole chip (the default).
extern "C" int unires_plan_set_concurrency(unires_plan_t *plan, int32_t n_concurrent) {
  if (!plan) return fail(UNIRES_ERR_NULL, "null argument");
  if (n_concurrent < 1 || n_concurrent > 64) return fail(UNIRES_ERR_ARG, "concurrency out of range (1 .. 64)");
  if (n_concurrent == plan->concurrency) return UNIRES_OK;
  int cap_s2 = 0, cap_f1 = 0;
  if (n_concurrent > 1 && plan->dy.numel() <= (1ull << 25)) {
    int dev = 0, ncu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    static const int s2_16 = getenv("UNIRES_SHARE_S2") ? atoi(getenv("UNIRES_SHARE_S2")) : 28;  // workgroups per 16 CUs
    static const int f1_16 = getenv("UNIRES_SHARE_F1") ? atoi(getenv("UNIRES_SHARE_F1")) : 48;
    cap_s2 = std::max(8, ncu * s2_16 / 16 / 8 * 8);  // 448 of 1 024 on 256 CUs
    cap_f1 = std::max(8, ncu * f1_16 / 16 / 8 * 8);  // 768 of 1 024
  }
  if (cap_s2 != plan->cap_s2 || cap_f1 != plan->cap_f1) drop_cg_graph(plan);  // (a captured solve has its grids baked in)
  plan->concurrency = n_concurrent, plan->cap_s2 = cap_s2, plan->cap_f1 = cap_f1;
  return UNIRES_OK;
}

extern "C" int64_t unires_plan_workspace_bytes(const unires_plan_t *plan) {
  return plan ? (int64_t)plan->ws_bytes : 0;
}

extern "C" int unires_orient_of(const float M[12], int32_t perm[3], int32_t flip[3]) {
  if (!M || !perm || !flip) return fail(UNIRES_ERR_NULL, "null argument");
  Affine A;
  memcpy(A.m, M, sizeof(A.m));
  for (int i = 0; i < 12; ++i)
    if (!isfinite(A.m[i])) return fail(UNIRES_ERR_ARG, "non-finite affine");
  const Orient O = orient_of(A);
  for (int j = 0; j < 3; ++j) perm[j] = O.perm[j], flip[j] = O.flip[j];
  return UNIRES_OK;
}

extern "C" int unires_plan_repeat_info(const unires_plan_t *plan, int32_t n, int32_t info[8]) {
  if (!plan || !info) return fail(UNIRES_ERR_NULL, "null argument");
  if (n < 0 || n >= (int)plan->reps.size()) return fail(UNIRES_ERR_ARG, "repeat index");
  const Repeat &R = plan->reps[n];
  const bool id = plan->regime == UNIRES_REGIME_IDENTITY;
  for (int j = 0; j < 3; ++j) info[j] = id ? j : R.orient.perm[j];
  info[3] = id ? 0 : (R.orient.flip[0] | (R.orient.flip[1] << 1) | (R.orient.flip[2] << 2));
  info[4] = R.pplan.valid ? 1 : 0;
  info[5] = R.sched.valid ? 2 + R.sched.axis : 0;
  info[6] = (R.shift.valid ? 1 : 0) | (R.f1.valid ? 2 : 0);
  info[7] = R.sep ? 1 : 0;
  return UNIRES_OK;
}

// --------------------------------------------------------------------------
// operators
// --------------------------------------------------------------------------
static PushSrc push_src(const Repeat &R, const float *data, bool convup, float scl) {
  PushSrc src;
  src.data = data;
  src.convup = convup ? 1 : 0;
  src.xd = R.dim_x;
  src.gd = convup ? R.dim_gf : R.dim_g;
  src.T = R.Tf;
  src.S = make_scaling(scl, R.dim_thick);
  return src;
}

// scaling split for the hybrid path: the part along z goes with the fused kernels, the rest with
// the 1-D passes
static Scaling scaling_z(const Scaling &S) { return S.dim == 2 ? S : Scaling{1.f, 1.f, -1}; }
static Scaling scaling_xy(const Scaling &S) { return S.dim == 2 ? Scaling{1.f, 1.f, -1} : S; }

// hybrid forward: out = S conv_down_xy (conv_down_z pull(in));  false: not available for this repeat
static bool hybrid_forward(unires_plan *pl, const Repeat &R, const float *in, const Scaling &S, float *out,
                           const int *done, hipStream_t st) {
  if (!(R.hyb || R.hybf) || !pl->gbuf2 || !R.pplan.valid) return false;
  if (launch_pull_conv2(R.pplan, in, pl->dy, R.Af, R.Tz, scaling_z(S), pl->gbuf, R.dim_h, R.dim_gf,
                        pl->fov_tol, done, st))
    return false;
  launch_conv_down_sep(pl->gbuf, R.dim_h, R.Txy, scaling_xy(S), out, R.dim_x, pl->gbuf, pl->gbuf2, done, st);
  return true;
}

// x-space intermediate of AtA: xbuf = S(2 scl) conv_down pull(in)  (regime 2) or
// gbuf = pull(in) (regime 1); returns the push source that finishes the operator.
static PushSrc ata_forward(unires_plan *pl, const Repeat &R, const float *in, const int *done,
                           hipStream_t st) {
  if (pl->regime == UNIRES_REGIME_DENOISE) {
    if (launch_pull_conv2(R.pplan, in, pl->dy, R.A, R.T, Scaling{1.f, 1.f, -1}, pl->gbuf, R.dim_g, R.dim_g,
                          pl->fov_tol, done, st))
      launch_pull(in, pl->dy, R.A, pl->gbuf, R.dim_g, pl->fov_tol, done, st);
    return push_src(R, pl->gbuf, false, 0.f);
  }
  // S(2 scl) once between conv and conv^T  (unires/_project.py:175-177)
  const Scaling S2 = make_scaling(2.f * R.scl, R.dim_thick);
  // A^T A with stride-2 profiles along x and y: the x-space volume is only a way station, so the passes on either
  // side of it run as one kernel (ops.hip: k_conv_ydown_xdownup2, k_conv1d_downup2_m) and the push gets a crafted
  // source - a volume that is x-complete (forward-only hybrid: conv_up_y and z follow as one kernel, then the
  // grid-source splat) or x- and y-complete (hybrid: the z-profile splat takes it as it is)
  static const bool push_default = getenv("UNIRES_PUSH") == nullptr;
  const bool fwd_only = R.hybf && !R.hyb && R.sep && !(R.sched.valid && R.sched.axis >= 0);
  const bool both = R.hyb && R.sched.valid && R.sched.axis == 2 && R.Tf.s[1] == 2;
  if (push_default && (fwd_only || both) && pl->gbuf2 && R.pplan.valid && R.Tf.s[0] == 2 && !(R.Tf.n[0] == 1)) {
    // (x taps = Dirac for what follows the x pair; y taps too where conv_up_y went in)
    Taps Ty = R.Txy;
    Ty.n[0] = Ty.s[0] = 1, Ty.t[0][0] = 1.f;
    const Dim3i dxy = Dim3i{R.dim_h.x, R.dim_x.y, R.dim_x.z};
    const Scaling Sx = S2.dim == 0 ? S2 : Scaling{1.f, 1.f, -1}, Srest = S2.dim == 0 ? Scaling{1.f, 1.f, -1} : S2;
    const bool y_active = !(Ty.n[1] == 1 && Ty.s[1] == 1 && Ty.t[1][0] == 1.f) || Srest.dim == 1;
    if (y_active && !launch_pull_conv2(R.pplan, in, pl->dy, R.Af, R.Tz, scaling_z(Srest), pl->gbuf, R.dim_h, R.dim_gf,
                                       pl->fov_tol, done, st)) {
      // ... and conv_down_y in front of it in the same kernel where its taps are compiled in; where the z part
      // lives in the splat (`both`) conv_up_y goes in as well: the push source is then x- and y-complete
      const int gy = both ? R.dim_h.y : 0;
      if (!launch_conv_ydown_xdownup2(pl->gbuf, R.dim_h, R.Txy, scaling_xy(S2), R.dim_x.x, R.dim_x.y, gy, pl->gbuf2,
                                      done, st)) {
        PushSrc src = push_src(R, pl->gbuf2, true, 0.f);
        src.xd = both ? Dim3i{R.dim_h.x, R.dim_h.y, R.dim_x.z} : dxy;
        src.T.n[0] = src.T.s[0] = 1, src.T.t[0][0] = 1.f;
        if (both) src.T.n[1] = src.T.s[1] = 1, src.T.t[1][0] = 1.f;
        return src;
      }
      if (both && R.Tf.n[0] * R.Tf.n[1] <= 16) {  // (the fused 2-D kernels of ops.hip serve these taps)
        launch_conv_down_sep(pl->gbuf, R.dim_h, R.Txy, scaling_xy(S2), pl->xbuf, R.dim_x, pl->gbuf, pl->gbuf2, done, st);
        return push_src(R, pl->xbuf, true, 0.f);
      }
      launch_conv_down_sep(pl->gbuf, R.dim_h, Ty, scaling_xy(Srest), pl->gbuf2, dxy, pl->gbuf, pl->gbuf2, done, st);
      if (!launch_conv_downup2(pl->gbuf2, dxy, R.Tf, Sx, 0, R.dim_x.x, pl->gbuf, done, st)) {
        PushSrc src = push_src(R, pl->gbuf, true, 0.f);
        src.xd = dxy;
        src.T.n[0] = src.T.s[0] = 1, src.T.t[0][0] = 1.f;
        return src;
      }
      // not available for these taps: finish the x pass the usual way
      Taps Tx = R.Txy;
      Tx.n[1] = Tx.s[1] = 1, Tx.t[1][0] = 1.f;
      launch_conv_down_sep(pl->gbuf2, dxy, Tx, Sx, pl->xbuf, R.dim_x, pl->gbuf, pl->gbuf, done, st);
      return push_src(R, pl->xbuf, true, 0.f);
    }
  }
  if (hybrid_forward(pl, R, in, S2, pl->xbuf, done, st)) return push_src(R, pl->xbuf, true, 0.f);
  if (R.sep && pl->gbuf2) {
    launch_pull(in, pl->dy, R.Af, pl->gbuf, R.dim_gf, pl->fov_tol, done, st);
    launch_conv_down_sep(pl->gbuf, R.dim_gf, R.Tf, S2, pl->xbuf, R.dim_x, pl->gbuf, pl->gbuf2, done, st);
    return push_src(R, pl->xbuf, true, 0.f);
  }
  if (launch_pull_conv2(R.pplan, in, pl->dy, R.Af, R.Tf, S2, pl->xbuf, R.dim_x, R.dim_gf, pl->fov_tol, done, st) &&
      launch_pull_conv(in, pl->dy, R.Af, R.Tf, S2, pl->xbuf, R.dim_x, R.dim_gf, pl->fov_tol, done,
                       st)) {
    launch_pull(in, pl->dy, R.A, pl->gbuf, R.dim_g, pl->fov_tol, done, st);
    launch_conv_down(pl->gbuf, R.dim_g, R.T, S2, pl->xbuf, R.dim_x, done, st);
  }
  return push_src(R, pl->xbuf, true, 0.f);
}

// out = [out +] alpha * push(src) [+ epilogue]; falls back to a materialised conv_up when
// the conv_up fan-in is beyond what the fused kernel tabulates.
static int push_any(unires_plan *pl, const PushSrc &src, const Repeat &R, float alpha,
                     const PushEpilogue &ep, float *out, const int *done, hipStream_t st) {
  const Affine &A = src.convup ? R.Af : R.A;
  // default: the schedule-driven splat (k_splat2), then the r1 tile kernels where an operator is outside its
  // domain; UNIRES_PUSH=tile forces the general tile kernel (tests' cross-check)
  static const char *mode = getenv("UNIRES_PUSH");
  static const bool use_tile = mode && !strcmp(mode, "tile");
  if (!use_tile && mode == nullptr && R.hyb && src.convup && R.sched.valid && R.sched.axis == 2 && pl->gbuf2) {
    // conv_up along x / y as 1-D passes, then the z-profile splat with the intermediate as its source
    Taps Txy = src.T;  // (= R.Txy, or with the x part done already: ata_forward)
    Txy.n[2] = Txy.s[2] = 1, Txy.t[2][0] = 1.f;
    const Scaling Sxy = scaling_xy(src.S);
    const bool xy_done = Sxy.dim < 0 && Txy.n[0] == 1 && Txy.s[0] == 1 && Txy.t[0][0] == 1.f && Txy.n[1] == 1 &&
                         Txy.s[1] == 1 && Txy.t[1][0] == 1.f;  // (ata_forward's one-kernel x / y part: nothing left)
    const float *h = xy_done ? src.data : launch_conv_up_sep(src.data, src.xd, Txy, Sxy, R.dim_h, pl->gbuf, pl->gbuf2, st);
    const float4 *tab = (const float4 *)R.ctab_dev[src.S.dim == 2 ? 1 : 0];
    if (!launch_splat2(R.sched, h, R.dim_h.numel(), tab, R.ctab_n, R.src_stride, R.ctab_step, R.src_stride,
                       R.ctab_step, A, alpha, ep, out, pl->dy, done, st))
      return ep.partials ? splat2_blocks(pl->dy, ep.grid_cap) : 0;
  }
  if (!use_tile && mode == nullptr && !R.hyb && R.sched.valid && (src.convup != 0) == (R.sched.axis >= 0)) {
    const float4 *tab = src.convup ? (const float4 *)R.ctab_dev[src.S.dim >= 0 ? 1 : 0] : nullptr;
    const size_t numel = src.convup ? src.xd.numel() : src.gd.numel();
    if (!launch_splat2(R.sched, src.data, numel, tab, R.ctab_n, R.src_stride, R.ctab_step, R.src_stride,
                       R.ctab_step, A, alpha, ep, out, pl->dy, done, st))
      return ep.partials ? splat2_blocks(pl->dy, ep.grid_cap) : 0;
  }
  if (!use_tile &&
      !launch_splat(src, A, R.Afinv, R.safe, alpha, pl->fov_tol, ep, out, pl->dy, done, st))
    return ep.partials ? splat_blocks(pl->dy, A) : 0;
  if (!use_tile && src.convup && R.sep && pl->gbuf2) {
    // many-tap profile: conv_up as 1-D passes into grid space, then the grid-source splat
    PushSrc d = src;
    d.data = launch_conv_up_sep(src.data, src.xd, src.T, src.S, src.gd, pl->gbuf, pl->gbuf2, st);
    d.convup = 0;
    if (mode == nullptr && R.sched.valid && R.sched.axis < 0 &&
        !launch_splat2(R.sched, d.data, d.gd.numel(), nullptr, 0, R.src_stride, 1, 0, 0, A, alpha, ep, out,
                       pl->dy, done, st))
      return ep.partials ? splat2_blocks(pl->dy, ep.grid_cap) : 0;
    if (!launch_splat(d, A, R.Afinv, R.safe, alpha, pl->fov_tol, ep, out, pl->dy, done, st))
      return ep.partials ? splat_blocks(pl->dy, A) : 0;
    (void)launch_push_tile(d, A, R.Afinv, R.safe, alpha, pl->fov_tol, ep, out, pl->dy, done, st);
    return ep.partials ? push_tile_blocks(pl->dy) : 0;
  }
  if (launch_push_tile(src, A, R.Afinv, R.safe, alpha, pl->fov_tol, ep, out, pl->dy, done, st)) {
    launch_conv_up(src.data, src.xd, src.T, src.S, pl->gbuf, src.gd, st);
    PushSrc d = src;
    d.data = pl->gbuf;
    d.convup = 0;
    (void)launch_push_tile(d, A, R.Afinv, R.safe, alpha, pl->fov_tol, ep, out, pl->dy, done, st);
  }
  return ep.partials ? push_tile_blocks(pl->dy) : 0;
}

// out (+)= alpha * At_n(x)
static void at_accumulate(unires_plan *pl, const Repeat &R, const float *x, float *out, float alpha,
                          bool accumulate, hipStream_t st) {
  if (pl->regime == UNIRES_REGIME_IDENTITY) {
    launch_axpy(alpha, x, out, pl->dy.numel(), st);  // caller initialised out
    return;
  }
  if (R.oriented) {  // the caller's voxel layout -> the plan's
    launch_to_canonical(R.orient, x, R.dim_xu, pl->xperm, st);
    x = pl->xperm;
  }
  PushEpilogue ep;
  ep.accumulate = accumulate ? 1 : 0;
  ep.grid_cap = pl->cap_s2;
  const bool sr = pl->regime == UNIRES_REGIME_SUPERRES;
  push_any(pl, push_src(R, x, sr, sr ? R.scl : 0.f), R, alpha, ep, out, nullptr, st);
}

extern "C" int unires_proj_apply(unires_plan_t *plan, int32_t n, int32_t op, const float *in,
                                 float *out, void *stream) {
  if (!plan || !in || !out) return fail(UNIRES_ERR_NULL, "null argument");
  if (n < 0 || n >= (int)plan->reps.size()) return fail(UNIRES_ERR_ARG, "repeat index");
  if (op != UNIRES_OP_A && op != UNIRES_OP_AT && op != UNIRES_OP_ATA)
    return fail(UNIRES_ERR_ARG, "Undefined operator");
  if (in == out) return fail(UNIRES_ERR_ARG, "proj_apply cannot run in place");
  hipStream_t st = (hipStream_t)stream;
  mark_use(plan, st);  // (before anything is enqueued: an error return below is remembered too)
  const Repeat &R = plan->reps[n];
  const size_t ny = plan->dy.numel();
  if (plan->regime == UNIRES_REGIME_IDENTITY) {  // operator 'none': return dat
    HIP_TRY(hipMemcpyAsync(out, in, ny * sizeof(float), hipMemcpyDeviceToDevice, st));
    return UNIRES_OK;
  }
  if (op == UNIRES_OP_A) {
    float *const out_user = out;
    if (R.oriented) out = plan->xperm;  // canonical layout first, re-ordered into the caller's below
    if (plan->regime == UNIRES_REGIME_DENOISE) {
      if (launch_pull_conv2(R.pplan, in, plan->dy, R.A, R.T, Scaling{1.f, 1.f, -1}, out, R.dim_g, R.dim_g,
                            plan->fov_tol, nullptr, st))
        launch_pull(in, plan->dy, R.A, out, R.dim_g, plan->fov_tol, nullptr, st);
    } else {
      if (hybrid_forward(plan, R, in, make_scaling(R.scl, R.dim_thick), out, nullptr, st)) {
      } else if (R.sep && plan->gbuf2) {
        launch_pull(in, plan->dy, R.Af, plan->gbuf, R.dim_gf, plan->fov_tol, nullptr, st);
        launch_conv_down_sep(plan->gbuf, R.dim_gf, R.Tf, make_scaling(R.scl, R.dim_thick), out,
                             R.dim_x, plan->gbuf, plan->gbuf2, nullptr, st);
      } else if (launch_pull_conv2(R.pplan, in, plan->dy, R.Af, R.Tf, make_scaling(R.scl, R.dim_thick), out,
                                   R.dim_x, R.dim_gf, plan->fov_tol, nullptr, st) &&
                 launch_pull_conv(in, plan->dy, R.Af, R.Tf, make_scaling(R.scl, R.dim_thick), out,
                                  R.dim_x, R.dim_gf, plan->fov_tol, nullptr, st)) {
        launch_pull(in, plan->dy, R.A, plan->gbuf, R.dim_g, plan->fov_tol, nullptr, st);
        launch_conv_down(plan->gbuf, R.dim_g, R.T, make_scaling(R.scl, R.dim_thick), out, R.dim_x,
                         nullptr, st);
      }
    }
    if (R.oriented) launch_from_canonical(R.orient, plan->xperm, out_user, R.dim_xu, st);
  } else if (op == UNIRES_OP_AT) {
    at_accumulate(plan, R, in, out, 1.f, false, st);
  } else {
    if (!(plan->regime == UNIRES_REGIME_DENOISE && R.f1.valid &&
          !launch_ata1(R.f1, in, R.Af, 1.f, PushEpilogue(), out, plan->dy, nullptr, st))) {
      const PushSrc src = ata_forward(plan, R, in, nullptr, st);
      push_any(plan, src, R, 1.f, PushEpilogue(), out, nullptr, st);
    }
  }
  CHECK_LAUNCH();
  return UNIRES_OK;
}

// q = sum_n tau_n AtA_n p + rho lam^2 DtD p ; optional dot partials of sum(p*q).
// With objb (and part): the partials hold sum (q - 2 objb) * p instead and the final q is not
// stored (q is still scratch for the partial sums of a multi-repeat operator).
// Returns the number of partials written (0 if none requested).
static int matvec(unires_plan *pl, float rho, float lam, const float *p, float *q, double *part,
                  const int *done, hipStream_t st, const float *objb = nullptr) {
  const float c = rho * (lam * lam);
  if (pl->regime == UNIRES_REGIME_IDENTITY) {
    float a0 = 0.f;
    for (const Repeat &R : pl->reps) a0 += R.tau;
    static const bool no_lines = getenv("UNIRES_NO_ALIGNED") != nullptr;
    static const bool no_flat = getenv("UNIRES_NO_FLAT") != nullptr;
    const float ivx = 1.f / (pl->vx[0] * pl->vx[0]), ivy = 1.f / (pl->vx[1] * pl->vx[1]),
                ivz = 1.f / (pl->vx[2] * pl->vx[2]);
    // one flat streaming pass (stencil.hip); the line kernel and the generic one are its fallbacks
    if (!no_flat && !launch_dtd_flat(p, q, pl->dy, a0, c * ivx, c * ivy, c * ivz, part, objb, done, st))
      return part ? dtd_flat_blocks(pl->dy) : 0;
    if (!no_lines &&
        !launch_dtd_lines(p, q, pl->dy, a0, c / (pl->vx[0] * pl->vx[0]), c / (pl->vx[1] * pl->vx[1]),
                          c / (pl->vx[2] * pl->vx[2]), part, objb, done, st))
      return part ? aligned_blocks(pl->dy) : 0;
    launch_dtd(p, pl->dy, pl->vx, a0, c, q, part, objb, done, st);
    return part ? dtd_num_blocks(pl->dy) : 0;
  }
  const size_t nrep = pl->reps.size();
  static const bool no_aligned = getenv("UNIRES_NO_ALIGNED") != nullptr;
  if (nrep == 1 && !no_aligned) {
    // grid-aligned observation (identity + integer shift, z slice profile): one streaming kernel
    const Repeat &R = pl->reps[0];
    const float ivx = 1.f / (pl->vx[0] * pl->vx[0]), ivy = 1.f / (pl->vx[1] * pl->vx[1]),
                ivz = 1.f / (pl->vx[2] * pl->vx[2]);
    // where the x-marching kernel's fast form applies it serves integer shifts too (31.5 us against
    // k_ata_aligned4x2's 36 - 37 at 256^3)
    if (shift_fast(R.shift, pl->dy) &&
        !launch_ata_shift(R.shift, p, q, pl->dy, R.Af, R.tau, 0.f, c * ivx, c * ivy, c * ivz, part, objb, done, st))
      return part ? shift_blocks(pl->dy) : 0;
    if (!launch_ata_aligned(p, q, pl->dy, R.dim_gf, R.dim_x, R.Tf,
                            make_scaling(2.f * R.scl, R.dim_thick), R.Af, R.tau, 0.f, c * ivx,
                            c * ivy, c * ivz, part, objb, done, st))
      return part ? aligned_blocks(pl->dy) : 0;
    // ... or translated by a fraction of a voxel (no rotation): the factorised one-kernel matvec
    if (!launch_ata_shift(R.shift, p, q, pl->dy, R.Af, R.tau, 0.f, c * ivx, c * ivy, c * ivz, part, objb, done, st))
      return part ? shift_blocks(pl->dy) : 0;
  }
  // regimes 1/2: two kernels per repeat; the last one also adds c DtD p and the dot
  int npart = 0;
  for (size_t n = 0; n < nrep; ++n) {
    const Repeat &R = pl->reps[n];
    PushEpilogue ep;
    ep.p = p;
    ep.accumulate = n > 0;
    if (n == 0) {  // the stencil term goes in once
      ep.cx = c / (pl->vx[0] * pl->vx[0]);
      ep.cy = c / (pl->vx[1] * pl->vx[1]);
      ep.cz = c / (pl->vx[2] * pl->vx[2]);
    }
    if (n + 1 == nrep) ep.partials = part, ep.objb = objb;
    // denoising regime: pull, push, stencil and dot in ONE pass over p (ata1.hip)
    ep.grid_cap = pl->cap_f1;
    if (pl->regime == UNIRES_REGIME_DENOISE && R.f1.valid &&
        !launch_ata1(R.f1, p, R.Af, R.tau, ep, q, pl->dy, done, st)) {
      npart = ep.partials ? ata1_blocks(pl->dy, ep.grid_cap) : 0;
      continue;
    }
    ep.grid_cap = pl->cap_s2;
    const PushSrc src = ata_forward(pl, R, p, done, st);
    npart = push_any(pl, src, R, R.tau, ep, q, done, st);
  }
  return npart;
}

extern "C" int unires_ata_matvec(unires_plan_t *plan, float rho, float lam, const float *p,
                                 float *q, double *dot_dev, void *stream) {
  if (!plan || !p || !q) return fail(UNIRES_ERR_NULL, "null argument");
  if (p == q) return fail(UNIRES_ERR_ARG, "matvec cannot run in place");
  hipStream_t st = (hipStream_t)stream;
  mark_use(plan, st);  // (before anything is enqueued: an error return below is remembered too)
  const int g = matvec(plan, rho, lam, p, q, dot_dev ? plan->part0 : nullptr, nullptr, st);
  if (dot_dev) launch_sum_to(plan->part0, g, dot_dev, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_precond_build(unires_plan_t *plan, int32_t precond_mode, float rho,
                                    float lam, float *m_out, void *stream) {
  if (!plan) return fail(UNIRES_ERR_NULL, "null plan");
  // The ADMM loop asks for the preconditioner every iteration: nothing to do while the mode,
  // rho, lam and the operator (set_repeat clears prec_ready) are what it was built for.  The
  // captured CG solve survives a rebuild too: it reads the diagonal at run time and its key
  // holds the mode, rho and lam.
  if (precond_mode == UNIRES_PRECOND_IDENTITY) {
    plan->prec_ready = false;
    return UNIRES_OK;
  }
  if (plan->prec_ready && plan->prec_mode == precond_mode && plan->prec_rho == rho && plan->prec_lam == lam &&
      !m_out)
    return UNIRES_OK;
  if (precond_mode != UNIRES_PRECOND_JACOBI && precond_mode != UNIRES_PRECOND_FFT)
    return fail(UNIRES_ERR_UNSUPPORTED, "preconditioner modes: identity (0), Jacobi (1), FFT (2)");
  hipStream_t st = (hipStream_t)stream;
  mark_use(plan, st);  // (before anything is enqueued: an error return below is remembered too)
  const size_t ny = plan->dy.numel();
  if (precond_mode == UNIRES_PRECOND_FFT) {
    if (int rc = fftpre_setup(plan->fft, plan->dy))
      return fail(rc == 2 ? UNIRES_ERR_ALLOC : UNIRES_ERR_HIP, "hipFFT plan / buffer creation failed");
    FftPre &F = plan->fft;
    // a = mean diagonal of the data term: mean_v sum_n tau_n (AtA_n 1)(v)
    double a = 0.0;
    if (plan->regime == UNIRES_REGIME_IDENTITY) {
      for (const Repeat &R : plan->reps) a += R.tau;
    } else {
      launch_fill(1.f, plan->ax, ny, st);
      for (size_t n = 0; n < plan->reps.size(); ++n) {
        const Repeat &R = plan->reps[n];
        const PushSrc src = ata_forward(plan, R, plan->ax, nullptr, st);
        PushEpilogue ep;
        ep.accumulate = n > 0;
        push_any(plan, src, R, R.tau, ep, F.z, nullptr, st);
      }
      launch_dot(F.z, plan->ax, ny, plan->part0, nullptr, st);
      launch_sum_to(plan->part0, vec_num_blocks(ny), &plan->state->rz, st);
      HIP_TRY(hipMemcpyAsync(&a, &plan->state->rz, sizeof(double), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      a /= (double)ny;
    }
    if (!(a > 0.0)) return fail(UNIRES_ERR_ARG, "data term has an empty diagonal");
    F.a = (float)a;
    for (int d = 0; d < 3; ++d) F.c[d] = rho * (lam * lam) / (plan->vx[d] * plan->vx[d]);
    if (m_out) return fail(UNIRES_ERR_ARG, "m_out is only defined for the Jacobi diagonal");
    CHECK_LAUNCH();
    plan->prec_rho = rho, plan->prec_lam = lam, plan->prec_mode = precond_mode, plan->prec_ready = true;
    return UNIRES_OK;
  }
  if (plan->reps.size() != 1)  // the reference raises ValueError here (_update.py:84-85)
    return fail(UNIRES_ERR_ARG, "CG pre-conditioning only supports one repeat per contrast.");
  if (!plan->precM) HIP_TRY(hipMalloc((void **)&plan->precM, ny * sizeof(float)));
  const Repeat &R = plan->reps[0];
  float c = 0.f;  // 2 rho lam^2 sum_d 1/vx_d^2, float32 like the reference's 0-d tensors
  for (int d = 0; d < 3; ++d) c += 1.f / (plan->vx[d] * plan->vx[d]);
  c = 2.f * rho * (lam * lam) * c;
  if (plan->regime == UNIRES_REGIME_IDENTITY) {
    launch_fill(R.tau + c, plan->precM, ny, st);
  } else {
    launch_fill(1.f, plan->ax, ny, st);
    const PushSrc src = ata_forward(plan, R, plan->ax, nullptr, st);
    push_any(plan, src, R, 1.f, PushEpilogue(), plan->precM, nullptr, st);
    launch_scale_shift(R.tau, c, plan->precM, ny, st);
  }
  if (m_out)
    HIP_TRY(hipMemcpyAsync(m_out, plan->precM, ny * sizeof(float), hipMemcpyDeviceToDevice, st));
  CHECK_LAUNCH();
  plan->prec_rho = rho, plan->prec_lam = lam, plan->prec_mode = precond_mode, plan->prec_ready = true;
  return UNIRES_OK;
}

extern "C" int unires_precond_apply(unires_plan_t *plan, const float *in, float *out, void *stream) {
  if (!plan || !in || !out) return fail(UNIRES_ERR_NULL, "null argument");
  if (in == out) return fail(UNIRES_ERR_ARG, "precond_apply cannot run in place");
  hipStream_t st = (hipStream_t)stream;
  mark_use(plan, st);  // (before anything is enqueued: an error return below is remembered too)
  const size_t ny = plan->dy.numel();
  if (!plan->prec_ready || plan->prec_mode == UNIRES_PRECOND_IDENTITY) {
    HIP_TRY(hipMemcpyAsync(out, in, ny * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else if (plan->prec_mode == UNIRES_PRECOND_FFT) {
    if (fftpre_apply(plan->fft, in, out, st)) return fail(UNIRES_ERR_HIP, "hipFFT execution failed");
  } else {
    launch_div(in, plan->precM, out, ny, st);
  }
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_rhs_assemble(unires_plan_t *plan, const float *const *x_ptrs,
                                   const float *w_c, const float *z_c, float rho, float lam,
                                   float *b, void *stream) {
  if (!plan || !x_ptrs || !w_c || !z_c || !b) return fail(UNIRES_ERR_NULL, "null argument");
  for (size_t n = 0; n < plan->reps.size(); ++n)
    if (!x_ptrs[n]) return fail(UNIRES_ERR_NULL, "null observation pointer");
  hipStream_t st = (hipStream_t)stream;
  mark_use(plan, st);  // (before anything is enqueued: an error return below is remembered too)
  // b = -lam * Dt(w - rho z)   (unires/_update.py:131-133)
  launch_div(w_c, z_c, 1.f, -rho, plan->dy, plan->vx, -lam, nullptr, b, st);
  // b += tau_n At_n x_n         (unires/_update.py:125-128)
  for (size_t n = 0; n < plan->reps.size(); ++n)
    at_accumulate(plan, plan->reps[n], x_ptrs[n], b, plan->reps[n].tau, true, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_atx_assemble(unires_plan_t *plan, const float *const *x_ptrs, float *atx,
                                   void *stream) {
  if (!plan || !x_ptrs || !atx) return fail(UNIRES_ERR_NULL, "null argument");
  for (size_t n = 0; n < plan->reps.size(); ++n)
    if (!x_ptrs[n]) return fail(UNIRES_ERR_NULL, "null observation pointer");
  hipStream_t st = (hipStream_t)stream;
  mark_use(plan, st);  // (before anything is enqueued: an error return below is remembered too)
  if (plan->regime == UNIRES_REGIME_IDENTITY)
    HIP_TRY(hipMemsetAsync(atx, 0, plan->dy.numel() * sizeof(float), st));
  for (size_t n = 0; n < plan->reps.size(); ++n)
    at_accumulate(plan, plan->reps[n], x_ptrs[n], atx, plan->reps[n].tau,
                  n > 0 || plan->regime == UNIRES_REGIME_IDENTITY, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_rhs_from_atx(unires_plan_t *plan, const float *atx, const float *w_c,
                                   const float *z_c, float rho, float lam, float *b,
                                   void *stream) {
  if (!plan || !atx || !w_c || !z_c || !b) return fail(UNIRES_ERR_NULL, "null argument");
  launch_div(w_c, z_c, 1.f, -rho, plan->dy, plan->vx, -lam, atx, b, (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

// --------------------------------------------------------------------------
// CG  (nitorch.core.optim.cg as UniRes calls it)
// --------------------------------------------------------------------------
// The solve is enqueued in two parts: the start (r = b - A x, p, r.z, obj[0]) and runs of iterations.
// `dev_k`: the iteration index is the device state's own counter (a captured chunk of iterations then
// serves any part of a solve); `hostw`: the scalar kernel that ends an iteration publishes the state's
// (generation, done, iterations) to this host-mapped word.
static int cg_enqueue_start(unires_plan *pl, float rho, float lam, const float *b, float *x, double tol,
                            int stop_mode, const float *M, bool fft, unsigned long long *hostw,
                            hipStream_t st) {
  const size_t ny = pl->dy.numel();
  const bool check = tol != 0.0;
  CgState *S = pl->state;
  const int gv = vec_num_blocks(ny);
  // r = b - A(x); p = r; rz = r.r; obj[0]
  HIP_TRY(hipMemsetAsync(&S->done, 0, sizeof(int), st));
  matvec(pl, rho, lam, x, pl->ap, nullptr, nullptr, st);
  const bool want_obj0 = check && stop_mode != UNIRES_STOP_RESIDUAL;
  launch_residual_init(b, pl->ap, x, pl->r, pl->p, ny, pl->part0, want_obj0 ? pl->part1 : nullptr,
                       M, st);
  if (fft) {  // z = M^-1 r ; p = z ; rz = r.z
    if (fftpre_apply(pl->fft, pl->r, pl->fft.z, st)) return fail(UNIRES_ERR_HIP, "hipFFT execution failed");
    HIP_TRY(hipMemcpyAsync(pl->p, pl->fft.z, ny * sizeof(float), hipMemcpyDeviceToDevice, st));
    launch_dot(pl->r, pl->fft.z, ny, pl->part0, nullptr, st);
  }
  launch_sc_init(S, pl->part0, pl->part1, gv, stop_mode, check ? 1 : 0, hostw, st);
  return UNIRES_OK;
}

static int cg_enqueue_iters(unires_plan *pl, float rho, float lam, const float *b, float *x, int k_first,
                            int count, double tol, int stop_mode, const float *M, bool fft, bool dev_k,
                            unsigned long long *hostw, hipStream_t st) {
  const size_t ny = pl->dy.numel();
  const bool check = tol != 0.0;
  CgState *S = pl->state;
  const int *done = &S->done;
  const int gv = vec_num_blocks(ny);
  // UNIRES_CG_FOLD=1: alpha / beta in the prologues of the vector kernels instead of one-block
  // kernels of their own
  static const bool fold_on = getenv("UNIRES_CG_FOLD") && getenv("UNIRES_CG_FOLD")[0] == '1';
  const int gf = vec_num_blocks_fold(ny);
  for (int k = k_first; k < k_first + count; ++k) {
    const int kk = dev_k ? -1 : k;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (pl->timing && hipEventCreate(&ev0) == hipSuccess) {
      if (hipEventCreate(&ev1) == hipSuccess) {
        (void)hipEventRecord(ev0, st);
      } else {
        (void)hipEventDestroy(ev0);
        ev0 = nullptr;
      }
    }
    if (pl->twice) (void)matvec(pl, rho, lam, pl->p, pl->ap, pl->part0, done, st);  // (measurement: see the header)
    const int g = matvec(pl, rho, lam, pl->p, pl->ap, pl->part0, done, st);
    if (ev0 && ev1) {
      (void)hipEventRecord(ev1, st);
      if (pl->tev.size() >= 65536) {  // a caller that never collects: forget the oldest pair
        (void)hipEventDestroy(pl->tev.front().first), (void)hipEventDestroy(pl->tev.front().second);
        pl->tev.erase(pl->tev.begin());
      }
      pl->tev.emplace_back(ev0, ev1);
    }
    const bool guarded = check && stop_mode == UNIRES_STOP_MAXGAIN_GUARDED;
    const bool recur = check && (stop_mode == UNIRES_STOP_MAXGAIN_RECURRED || guarded);
    int obj_kind = 0;
    if (check && stop_mode == UNIRES_STOP_RESIDUAL) obj_kind = 1;
    if (recur) obj_kind = 2;
    // "x += alpha p" rides with the p update unless sc_beta can stop the solve in between
    const bool lazy_x = obj_kind == 0;
    if (fold_on && lazy_x && !fft && !dev_k) {
      // no scalar kernels: alpha in the prologue of the r update (its r.z partials go to part1 -
      // part0 is being read by every workgroup), beta in the prologue of the x / p update
      launch_update_r_fold(S, pl->part0, g, k, pl->ap, pl->r, ny, pl->part1, M, st);
      launch_update_px_fold(S, pl->part1, gf, k, pl->r, pl->p, x, ny, M, st);
    } else {
      launch_sc_alpha(S, pl->part0, g, st);
      launch_update_xr(S, pl->p, pl->ap, lazy_x ? nullptr : x, pl->r, b, ny, pl->part0,
                       recur ? pl->part1 : nullptr, M, st);
      if (fft) {  // (the transforms also run after convergence: hipFFT has no device-side skip)
        if (fftpre_apply(pl->fft, pl->r, pl->fft.z, st)) return fail(UNIRES_ERR_HIP, "hipFFT execution failed");
        launch_dot(pl->r, pl->fft.z, ny, pl->part0, done, st);
      }
      if (guarded)
        launch_sc_beta_guarded(S, pl->part0, pl->part1, gv, kk, tol, hostw, st);
      else
        launch_sc_beta(S, pl->part0, pl->part1, gv, kk, obj_kind, tol, obj_kind ? hostw : nullptr, st);
      launch_update_p(S, fft ? pl->fft.z : pl->r, pl->p, ny, M, lazy_x ? x : nullptr, st);
    }
    if (check && stop_mode == UNIRES_STOP_MAXGAIN) {
      // objective sum x (Ax - 2b) folded into the matvec epilogue: A(x) is never stored
      const int go = matvec(pl, rho, lam, x, pl->ax, pl->part1, done, st, b);
      launch_sc_obj(S, pl->part1, go, kk, tol, hostw, st);
    }
    if (guarded) {
      // ... the same, but its kernels return at entry unless k_sc_beta_guarded asked for it (cg.hip)
      const int go = matvec(pl, rho, lam, x, pl->ax, pl->part1, &S->skip_fresh, st, b);
      launch_sc_obj_guarded(S, pl->part1, go, kk, tol, hostw, st);
    }
  }
  return UNIRES_OK;
}

// Enqueues the whole solve (every kernel of nitorch's cg()) on `st`.
static int cg_enqueue(unires_plan *pl, float rho, float lam, const float *b, float *x, int max_iter,
                      double tol, int stop_mode, const float *M, bool fft, hipStream_t st) {
  const int rc = cg_enqueue_start(pl, rho, lam, b, x, tol, stop_mode, M, fft, nullptr, st);
  if (rc) return rc;
  return cg_enqueue_iters(pl, rho, lam, b, x, 1, max_iter, tol, stop_mode, M, fft, false, nullptr, st);
}

// --------------------------------------------------------------------------
// Chunked solves (round 4): a solve that can stop early (tolerance > 0: the reference's default,
// struct.py:65-67 cgs_tol = 1e-3, 'max_gain') is enqueued as the start + chunks of `chunk` iterations,
// one chunk of look-ahead.  The kernel that ends an iteration publishes (generation, done, iterations)
// to a host-mapped word; the host enqueues the next chunk when the older of the two in flight has
// finished and the flag is not up - no stream synchronisation, the device never idles, and at most
// 2 chunk - 1 iterations run as no-op kernels after convergence (enqueuing all max_iter iterations, as
// rounds 1-3 did, ran 45 of config 3's 60 iterations as ~10 no-op kernels each).  The realised
// iteration count, iterate and objective trace are those of the full enqueue: the same kernels in
// the same order, the stopping test on the device.  Start and chunk are hipGraphs, captured once per
// (b, x, rho, lam, options) and replayed.
// --------------------------------------------------------------------------
struct CgRun {
  unires_plan *pl = nullptr;
  float rho = 0.f, lam = 0.f;
  const float *b = nullptr;
  float *x = nullptr;
  int max_iter = 0, stop = 0, chunk = 2;
  double tol = 0.0;
  const float *M = nullptr;
  bool fft = false;
  hipStream_t st = nullptr;
  int enqueued = 0;     // iterations enqueued so far
  unsigned gen = 0;     // generation of this solve
  bool finished = false;
  bool use_graph = false;
};

// The host counts the solves it starts (cg_gen), k_sc_init counts the ones that run (state->gen); the chunked
// driver matches the two in the progress word.  After an enqueue / capture / launch that FAILED somewhere
// between the two increments they may be out of step for the life of the plan: read the device's back.
static void cg_resync_gen(unires_plan *pl) {
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  unsigned g = pl->cg_gen;
  if (pl->state && hipMemcpy(&g, &pl->state->gen, sizeof(g), hipMemcpyDeviceToHost) == hipSuccess) pl->cg_gen = g;
  (void)hipGetLastError();
}

static int cg_chunk_size() {
  static const int k = [] {
    const char *e = getenv("UNIRES_CG_CHUNK");
    const int v = e ? atoi(e) : 2;
    return v < 1 ? 1 : (v > 64 ? 64 : v);
  }();
  return k;
}

static int capture_graph(hipStream_t st, hipGraphExec_t *exec, const std::function<int()> &body) {
  *exec = nullptr;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return -1;  // capture refused: the caller launches plainly
  }
  const int rc = body();
  hipGraph_t graph = nullptr;
  const hipError_t ce = hipStreamEndCapture(st, &graph);
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (ce != hipSuccess || !graph) return fail(UNIRES_ERR_HIP, "hipStreamEndCapture failed");
  const hipError_t ge = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ge != hipSuccess) {
    *exec = nullptr;
    return fail(UNIRES_ERR_HIP, "hipGraphInstantiate failed");
  }
  return UNIRES_OK;
}

static int cg_run_enqueue_chunk(CgRun &R) {
  unires_plan *pl = R.pl;
  const int n = std::min(R.chunk, R.max_iter - R.enqueued);
  if (n <= 0) return UNIRES_OK;
  if (R.use_graph && n == R.chunk && pl->cg_chunk_exec) {
    HIP_TRY(hipGraphLaunch(pl->cg_chunk_exec, R.st));
  } else {
    const int rc = cg_enqueue_iters(pl, R.rho, R.lam, R.b, R.x, R.enqueued + 1, n, R.tol, R.stop, R.M, R.fft,
                                    true, pl->progress_dev, R.st);
    if (rc) return rc;
  }
  R.enqueued += n;
  return UNIRES_OK;
}

static int cg_run_start_impl(CgRun &R) {
  unires_plan *pl = R.pl;
  if (!pl->progress) {
    HIP_TRY(hipHostMalloc((void **)&pl->progress, 64, hipHostMallocMapped));
    *pl->progress = 0ull;
    HIP_TRY(hipHostGetDevicePointer((void **)&pl->progress_dev, (void *)pl->progress, 0));
  }
  R.gen = ++pl->cg_gen;
  R.enqueued = 0;
  R.finished = false;
  unires_plan::CgKey key;
  key.b = R.b, key.x = R.x, key.rho = R.rho, key.lam = R.lam, key.max_iter = -2 - R.chunk, key.stop = R.stop;
  key.pre = R.M ? UNIRES_PRECOND_JACOBI : UNIRES_PRECOND_IDENTITY, key.tol = R.tol;
  if (R.use_graph && !(pl->cg_start_exec && pl->cg_chunk_exec && pl->cg_chunk_key == key)) {
    drop_cg_chunk_graphs(pl);
    int rc = capture_graph(R.st, &pl->cg_start_exec, [&] {
      return cg_enqueue_start(pl, R.rho, R.lam, R.b, R.x, R.tol, R.stop, R.M, R.fft, pl->progress_dev, R.st);
    });
    if (rc > 0) return rc;
    if (rc == 0)
      rc = capture_graph(R.st, &pl->cg_chunk_exec, [&] {
        return cg_enqueue_iters(pl, R.rho, R.lam, R.b, R.x, 1, R.chunk, R.tol, R.stop, R.M, R.fft, true,
                                pl->progress_dev, R.st);
      });
    if (rc > 0) return rc;
    if (rc < 0 || !pl->cg_start_exec || !pl->cg_chunk_exec) {
      drop_cg_chunk_graphs(pl);
      R.use_graph = false;
    } else {
      pl->cg_chunk_key = key;
    }
  }
  if (R.use_graph) {
    HIP_TRY(hipGraphLaunch(pl->cg_start_exec, R.st));
  } else {
    const int rc = cg_enqueue_start(pl, R.rho, R.lam, R.b, R.x, R.tol, R.stop, R.M, R.fft, pl->progress_dev, R.st);
    if (rc) return rc;
  }
  // two chunks in flight
  int rc = cg_run_enqueue_chunk(R);
  if (!rc) rc = cg_run_enqueue_chunk(R);
  if (rc) return rc;
  if (R.enqueued >= R.max_iter) R.finished = true;
  return UNIRES_OK;
}

static int cg_run_start(CgRun &R) {
  const int rc = cg_run_start_impl(R);
  if (rc) cg_resync_gen(R.pl);
  return rc;
}

// One look at the progress word (never blocks): enqueues the next chunk when the older chunk in flight is
// through and the solve has not converged.
static int cg_run_poll(CgRun &R) {
  if (R.finished) return UNIRES_OK;
  const unsigned long long w = __atomic_load_n(R.pl->progress, __ATOMIC_ACQUIRE);
  if ((unsigned)(w >> 32) != R.gen) return UNIRES_OK;  // this solve's first kernels have not run yet
  if (w & 0x80000000ull) {
    R.finished = true;
    return UNIRES_OK;
  }
  const int iters = (int)(w & 0x7fffffffull);
  while (!R.finished && iters >= R.enqueued - R.chunk) {
    const int rc = cg_run_enqueue_chunk(R);
    if (rc) return rc;
    if (R.enqueued >= R.max_iter) R.finished = true;
  }
  return UNIRES_OK;
}

// Drive a set of runs (each on its own plan and stream) until every one has everything it needs
// enqueued.  The host spins on the progress words; every so often it asks the streams for errors.
static int cg_runs_drive(std::vector<CgRun> &runs) {
  unsigned spins = 0;
  for (;;) {
    bool all = true;
    for (CgRun &R : runs) {
      if (R.finished) continue;
      const int rc = cg_run_poll(R);
      if (rc) return rc;
      all = all && R.finished;
    }
    if (all) return UNIRES_OK;
    // a chunk is hundreds of microseconds of device work and one more is queued behind it: after a short spin
    // the thread sleeps between looks (eight ranks on one host must not each burn a core on the wait)
    if (++spins > 256) {
      const struct timespec nap = {0, 50000};
      (void)nanosleep(&nap, nullptr);
    }
    // (a watchdog, not the progress signal - the word is: every 512th nap, ~25 ms, catches a faulted stream as
    // well as every 16th did, and a stream query on running work is not free: the runtime submits a marker
    // packet for it and its signal thread handles the completion)
    if ((spins <= 256 && (spins & 0xff) == 0) || (spins > 256 && (spins & 0x1ff) == 0)) {
      for (CgRun &R : runs) {
        if (R.finished) continue;
        const hipError_t q = hipStreamQuery(R.st);
        if (q == hipSuccess) {
          // the stream drained: whatever was enqueued has run and published; a look at the word must
          // either end the run or enqueue more
          const int before = R.enqueued;
          const int rc = cg_run_poll(R);
          if (rc) return rc;
          if (!R.finished && R.enqueued == before)
            return fail(UNIRES_ERR_HIP, "chunked CG: the stream drained without the expected progress");
        } else if (q != hipErrorNotReady) {
          g_err = std::string("chunked CG: ") + hipGetErrorString(q);
          return UNIRES_ERR_HIP;
        }
      }
    }
  }
}

static int cg_check_args(unires_plan *plan, float rho, float lam, const float *b, float *x, int32_t max_iter,
                         double tol, int32_t stop_mode, int32_t precond_mode) {
  if (!plan || !b || !x) return fail(UNIRES_ERR_NULL, "null argument");
  if (b == x) return fail(UNIRES_ERR_ARG, "b and x must not alias");
  if (max_iter < 0) return fail(UNIRES_ERR_ARG, "max_iter out of range");
  if (max_iter > kMaxCgIter && !(tol > 0.0))
    return fail(UNIRES_ERR_ARG, "max_iter beyond 4096 needs a tolerance (the solve is then enqueued in chunks)");
  if (stop_mode < 0 || stop_mode > 3) return fail(UNIRES_ERR_ARG, "bad stop mode");
  if (precond_mode < UNIRES_PRECOND_IDENTITY || precond_mode > UNIRES_PRECOND_FFT)
    return fail(UNIRES_ERR_UNSUPPORTED, "preconditioner modes: identity (0), Jacobi (1), FFT (2)");
  if (precond_mode != UNIRES_PRECOND_IDENTITY &&
      (!plan->prec_ready || plan->prec_mode != precond_mode || plan->prec_rho != rho ||
       plan->prec_lam != lam))
    return fail(UNIRES_ERR_ARG, "call unires_precond_build with this mode, rho and lam first");
  if (!(tol >= 0.0)) return fail(UNIRES_ERR_ARG, "tolerance must be >= 0");
  return UNIRES_OK;
}

static bool cg_graphs_on() {
  static const bool on = !(getenv("UNIRES_CG_GRAPH") && getenv("UNIRES_CG_GRAPH")[0] == '0');
  return on;
}

// chunked enqueue: solves that can stop early, unless switched off (UNIRES_CG_CHUNK=0: the full enqueue)
static bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs != hipStreamCaptureStatusNone;
}

static bool cg_chunked(double tol, int max_iter) {
  static const bool off = getenv("UNIRES_CG_CHUNK") && atoi(getenv("UNIRES_CG_CHUNK")) == 0;
  return tol != 0.0 && max_iter > 0 && (!off || max_iter > kMaxCgIter);
}

static CgRun cg_make_run(unires_plan *pl, float rho, float lam, const float *b, float *x, int max_iter,
                         double tol, int stop_mode, int precond_mode, hipStream_t st) {
  CgRun R;
  R.pl = pl, R.rho = rho, R.lam = lam, R.b = b, R.x = x, R.max_iter = max_iter, R.tol = tol, R.stop = stop_mode;
  R.M = precond_mode == UNIRES_PRECOND_JACOBI ? pl->precM : nullptr;
  R.fft = precond_mode == UNIRES_PRECOND_FFT;
  R.st = st;
  R.chunk = cg_chunk_size();
  R.use_graph = cg_graphs_on() && !R.fft && !pl->timing;
  return R;
}

static int cg_read_back(unires_plan *pl, int max_iter, double tol, int32_t *iters_out, double *obj_trace,
                        hipStream_t st) {
  CgState *S = pl->state;
  int it = 0;
  HIP_TRY(hipMemcpyAsync(&it, &S->iters, sizeof(int), hipMemcpyDeviceToHost, st));
  if (obj_trace && tol != 0.0)
    HIP_TRY(hipMemcpyAsync(obj_trace, S->obj, sizeof(double) * (size_t)(std::min(max_iter, kMaxCgIter) + 1),
                           hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *iters_out = it;
  return UNIRES_OK;
}

extern "C" int unires_cg_solve(unires_plan_t *plan, float rho, float lam, const float *b, float *x,
                               int32_t max_iter, double tol, int32_t stop_mode,
                               int32_t precond_mode, int32_t *iters_out, double *obj_trace,
                               void *stream) {
  int rc = cg_check_args(plan, rho, lam, b, x, max_iter, tol, stop_mode, precond_mode);
  if (rc) return rc;
  const float *M = precond_mode == UNIRES_PRECOND_JACOBI ? plan->precM : nullptr;
  const bool fft = precond_mode == UNIRES_PRECOND_FFT;
  hipStream_t st = (hipStream_t)stream;
  unires_plan *pl = plan;
  mark_use(pl, st);  // (before anything is enqueued: a failed start or drive below is remembered too)

  // (a stream under capture - e.g. the caller's torch.cuda.graph - runs nothing until the graph is launched: the
  // chunk feeder would wait for progress that never comes.  The whole solve then joins the capture, as in r3;
  // kernels after convergence return at entry.)
  if (stream_capturing(st) && tol != 0.0 && max_iter > kMaxCgIter)
    return fail(UNIRES_ERR_ARG, "a solve with a tolerance and more than 4096 iterations cannot join a stream capture "
                                "(it is fed to the device chunk by chunk, following its progress)");
  if (cg_chunked(tol, max_iter) && !(stream_capturing(st) && max_iter <= kMaxCgIter)) {
    std::vector<CgRun> runs(1, cg_make_run(pl, rho, lam, b, x, max_iter, tol, stop_mode, precond_mode, st));
    if ((rc = cg_run_start(runs[0]))) return rc;
    if ((rc = cg_runs_drive(runs))) return rc;
    mark_use(pl, st);
    CHECK_LAUNCH();
    return iters_out ? cg_read_back(pl, max_iter, tol, iters_out, obj_trace, st) : UNIRES_OK;
  }
  ++pl->cg_gen;  // (k_sc_init counts every solve)

  // hipGraph replay (UNIRES_CG_GRAPH=0 disables): the ~8 launches per iteration of a solve are
  // captured once and re-launched as one graph while the arguments stay the same
  unires_plan::CgKey key;
  key.b = b, key.x = x, key.rho = rho, key.lam = lam, key.max_iter = max_iter, key.stop = stop_mode;
  key.pre = precond_mode, key.tol = tol;
  const bool graphable = cg_graphs_on() && !fft && max_iter > 0 && !pl->timing;  // (events go with plain launches)
  if (graphable && pl->cg_exec && pl->cg_key == key) {
    HIP_TRY(hipGraphLaunch(pl->cg_exec, st));
  } else {
    if (graphable && pl->cg_exec) drop_cg_graph(pl);
    rc = graphable ? capture_graph(st, &pl->cg_exec, [&] {
      return cg_enqueue(pl, rho, lam, b, x, max_iter, tol, stop_mode, M, fft, st);
    }) : -1;
    if (rc > 0) {
      cg_resync_gen(pl);
      return rc;
    }
    if (rc == 0) {
      pl->cg_key = key;
      if (hipGraphLaunch(pl->cg_exec, st) != hipSuccess) {
        cg_resync_gen(pl);
        return fail(UNIRES_ERR_HIP, "hipGraphLaunch failed");
      }
    } else if ((rc = cg_enqueue(pl, rho, lam, b, x, max_iter, tol, stop_mode, M, fft, st))) {
      cg_resync_gen(pl);
      return rc;
    }
  }
  mark_use(pl, st);
  CHECK_LAUNCH();
  return iters_out ? cg_read_back(pl, max_iter, tol, iters_out, obj_trace, st) : UNIRES_OK;
}

// Several channels' solves at once, each on its own plan and stream (unires/_update.py:122-150 loops over
// the channels; they do not couple inside the y-update): the chunks of all of them are fed from one host
// loop, so that the channels still overlap on the device where they run on separate streams.
extern "C" int unires_cg_solve_many(int32_t n, unires_plan_t *const *plans, const float *rho, const float *lam,
                                    const float *const *b, float *const *x, int32_t max_iter, double tol,
                                    int32_t stop_mode, int32_t precond_mode, int32_t *iters_out,
                                    double *obj_trace, void *const *streams) {
  if (n < 1 || !plans || !rho || !lam || !b || !x || !streams) return fail(UNIRES_ERR_NULL, "null argument");
  for (int c = 0; c < n; ++c) {
    const int rc = cg_check_args(plans[c], rho[c], lam[c], b[c], x[c], max_iter, tol, stop_mode, precond_mode);
    if (rc) return rc;
    for (int d = 0; d < c; ++d)
      if (plans[d] == plans[c]) return fail(UNIRES_ERR_ARG, "one plan per solve");
  }
  bool capturing = false;
  for (int c = 0; c < n; ++c) capturing = capturing || stream_capturing((hipStream_t)streams[c]);
  if (capturing && tol != 0.0 && max_iter > kMaxCgIter)
    return fail(UNIRES_ERR_ARG, "a solve with a tolerance and more than 4096 iterations cannot join a stream capture "
                                "(it is fed to the device chunk by chunk, following its progress)");
  for (int c = 0; c < n; ++c) mark_use(plans[c], (hipStream_t)streams[c]);  // (before anything is enqueued)
  if (!cg_chunked(tol, max_iter) || (capturing && max_iter <= kMaxCgIter)) {  // nothing to steer: each solve is enqueued whole
    for (int c = 0; c < n; ++c) {
      const int rc = unires_cg_solve(plans[c], rho[c], lam[c], b[c], x[c], max_iter, tol, stop_mode, precond_mode,
                                     nullptr, nullptr, streams[c]);
      if (rc) return rc;
    }
  } else {
    std::vector<CgRun> runs;
    for (int c = 0; c < n; ++c)
      runs.push_back(cg_make_run(plans[c], rho[c], lam[c], b[c], x[c], max_iter, tol, stop_mode, precond_mode,
                                 (hipStream_t)streams[c]));
    for (CgRun &R : runs) {
      const int rc = cg_run_start(R);
      if (rc) return rc;
    }
    const int rc = cg_runs_drive(runs);
    if (rc) return rc;
    for (int c = 0; c < n; ++c) mark_use(plans[c], (hipStream_t)streams[c]);
    CHECK_LAUNCH();
  }
  if (iters_out)
    for (int c = 0; c < n; ++c) {
      const int rc = cg_read_back(plans[c], max_iter, tol, iters_out + c,
                                  obj_trace ? obj_trace + (size_t)c * (std::min(max_iter, kMaxCgIter) + 1) : nullptr,
                                  (hipStream_t)streams[c]);
      if (rc) return rc;
    }
  return UNIRES_OK;
}

// --------------------------------------------------------------------------
// z / w updates and objective sums  (unires/_update.py:154-195, 396-427)
// --------------------------------------------------------------------------
static int check_channels(const float *const *y_ptrs, const float *lam, int32_t n) {
  if (!y_ptrs || !lam) return fail(UNIRES_ERR_NULL, "null argument");
  if (n < 1 || n > 4096) return fail(UNIRES_ERR_ARG, "channel count out of range");
  for (int c = 0; c < n; ++c)
    if (!y_ptrs[c]) return fail(UNIRES_ERR_NULL, "null channel pointer");
  return UNIRES_OK;
}

// scratch of the float64 reductions (per-workgroup sums, added in index order by a second launch): one buffer per
// (device, stream), grown on demand, used in stream order by the launches that share it
static int reduce_scratch(hipStream_t st, size_t ndoubles, double **out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, std::pair<double *, size_t>> scratch;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto &slot = scratch[std::make_pair(dev, st)];
  if (slot.second < ndoubles) {
    if (slot.first) {
      HIP_TRY(hipDeviceSynchronize());
      (void)hipFree(slot.first);
      slot = {nullptr, 0};
    }
    const size_t n = std::max<size_t>(ndoubles, 16384);
    HIP_TRY(hipMalloc((void **)&slot.first, n * sizeof(double)));
    slot.second = n;
  }
  *out = slot.first;
  return UNIRES_OK;
}

extern "C" int unires_zw_update(const float *const *y_ptrs, const float *lam, int32_t n_channels,
                                const int32_t dim[3], const float vx[3], float rho, float alpha,
                                float *z, float *w, float *jtv, void *stream) {
  int rc = check_channels(y_ptrs, lam, n_channels);
  if (rc) return rc;
  if (!z || !w || !jtv || !dim) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (!vx_ok(vx)) return fail(UNIRES_ERR_ARG, "voxel size must be positive");
  if (!(rho > 0.f)) return fail(UNIRES_ERR_ARG, "rho must be positive");
  hipStream_t st = (hipStream_t)stream;
  const Dim3i d = mk(dim);
  launch_jtv_scale(y_ptrs, lam, n_channels, w, z, d, vx, rho, alpha, jtv, nullptr, nullptr, 0, st);
  const size_t n = d.numel();
  for (int c = 0; c < n_channels; ++c)
    launch_zw_update(y_ptrs[c], lam[c], jtv, z + (size_t)c * 3 * n, w + (size_t)c * 3 * n, d, vx,
                     rho, alpha, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_nll_prior(const float *const *y_ptrs, const float *lam, int32_t n_channels,
                                const int32_t dim[3], const float vx[3], double *out_dev,
                                void *stream) {
  int rc = check_channels(y_ptrs, lam, n_channels);
  if (rc) return rc;
  if (!out_dev || !dim) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dimensions");
  if (!vx_ok(vx)) return fail(UNIRES_ERR_ARG, "voxel size must be positive");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(double), st));
  // more than 8 channels: the running sum of squares needs a volume of scratch.  It is kept (one per
  // device, grown on demand, used in stream order by the chained launches) instead of allocated and freed
  // per call: no allocation inside a stream capture, nothing to leak on an error path
  float *acc = nullptr;
  if (n_channels > 8) {
    static std::mutex mu;
    static std::map<int, std::pair<float *, size_t>> scratch;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const size_t need = mk(dim).numel() * sizeof(float);
    std::lock_guard<std::mutex> lock(mu);
    auto &slot = scratch[dev];
    if (slot.second < need) {
      if (slot.first) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(slot.first);
        slot = {nullptr, 0};
      }
      HIP_TRY(hipMalloc((void **)&slot.first, need));
      slot.second = need;
    }
    acc = slot.first;
  }
  double *part = nullptr;
  if ((rc = reduce_scratch(st, (size_t)jtv_scale_blocks(mk(dim)), &part))) return rc;
  launch_jtv_scale(y_ptrs, lam, n_channels, nullptr, nullptr, mk(dim), vx, 1.f, 1.f, acc, part, out_dev, 1, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_scaling_sums(const float *x, const float *ay, const int32_t dim[3],
                                   int32_t dim_thick, double *out_dev, void *stream) {
  if (!x || !ay || !dim || !out_dev) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dims");
  if (dim_thick < 0 || dim_thick > 2) return fail(UNIRES_ERR_ARG, "bad dim_thick");
  hipStream_t st = (hipStream_t)stream;
  double *part = nullptr;
  if (int rc = reduce_scratch(st, 5 * (size_t)scaling_sums_blocks(mk(dim)), &part)) return rc;
  launch_scaling_sums(x, ay, mk(dim), dim_thick, part, out_dev, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_rigid_sums(const float *gr3, const float *diff, const float *ctc,
                                 const int32_t dim[3], const float d_rigid[72], double *out_dev,
                                 void *stream) {
  if (!gr3 || !diff || !dim || !d_rigid || !out_dev) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim)) return fail(UNIRES_ERR_DIM, "bad dims");
  hipStream_t st = (hipStream_t)stream;
  float D[6][12];
  memcpy(D, d_rigid, sizeof(D));
  double *part = nullptr;
  if (int rc = reduce_scratch(st, 27 * (size_t)rigid_sums_blocks(mk(dim)), &part)) return rc;
  launch_rigid_sums(gr3, diff, ctc, mk(dim), D, part, out_dev, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_clean_fov(float *y, const int32_t dim_y[3], const float M[12],
                                const int32_t dim_x[3], void *stream) {
  if (!y || !dim_y || !M || !dim_x) return fail(UNIRES_ERR_NULL, "null argument");
  if (!dims_ok(dim_y) || !dims_ok(dim_x)) return fail(UNIRES_ERR_DIM, "bad dims");
  Affine A;
  memcpy(A.m, M, sizeof(A.m));
  launch_clean_fov(y, mk(dim_y), A, mk(dim_x), (hipStream_t)stream);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_masked_sse(const float *x, const float *ay, int64_t n, double *out_dev,
                                 void *stream) {
  if (!x || !ay || !out_dev) return fail(UNIRES_ERR_NULL, "null argument");
  if (n < 1) return fail(UNIRES_ERR_DIM, "bad length");
  hipStream_t st = (hipStream_t)stream;
  double *part = nullptr;
  if (int rc = reduce_scratch(st, (size_t)masked_sse_blocks((size_t)n), &part)) return rc;
  launch_masked_sse(x, ay, (size_t)n, part, out_dev, st);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

// --------------------------------------------------------------------------
// Stream marks (round 5): how a host thread follows its GPU WITHOUT runtime calls.
// Every hipEventQuery / hipStreamQuery on work that is still running makes the runtime submit a marker
// packet whose completion wakes its signal thread, which then polls for a while before it sleeps again:
// measured on the MI355X box (tools/host_profile.py, profiles/r05_host_profile.txt) that helper thread
// cost 4.3 ms of CPU per 13.6 ms ADMM iteration under a 0.5 ms event poll.  A mark is a word of mapped host
// memory that a one-thread kernel sets when the stream gets there: the host reads memory, nothing else.
// --------------------------------------------------------------------------
struct unires_mark {
  unsigned long long *host = nullptr;
  unsigned long long *dev = nullptr;
};

__global__ void k_mark(unsigned long long *w, unsigned long long v) {
  __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int unires_mark_create(unires_mark_t **out) {
  if (!out) return fail(UNIRES_ERR_NULL, "null argument");
  *out = nullptr;
  unires_mark *m = new (std::nothrow) unires_mark;
  if (!m) return fail(UNIRES_ERR_ALLOC, "out of host memory");
  // (portable: a process that drives several devices may signal the mark from any of them)
  if (hipHostMalloc((void **)&m->host, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    delete m;
    return fail(UNIRES_ERR_ALLOC, "hipHostMalloc failed");
  }
  *m->host = 0ull;
  if (hipHostGetDevicePointer((void **)&m->dev, (void *)m->host, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(m->host);
    delete m;
    return fail(UNIRES_ERR_HIP, "hipHostGetDevicePointer failed");
  }
  *out = m;
  return UNIRES_OK;
}

extern "C" int unires_mark_destroy(unires_mark_t *m) {
  if (!m) return UNIRES_OK;
  if (m->host) (void)hipHostFree(m->host);  // (waits for the device: no kernel writes it afterwards)
  delete m;
  return UNIRES_OK;
}

extern "C" int unires_mark_signal(unires_mark_t *m, uint64_t value, void *stream) {
  if (!m) return fail(UNIRES_ERR_NULL, "null argument");
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, (hipStream_t)stream, m->dev, (unsigned long long)value);
  CHECK_LAUNCH();
  return UNIRES_OK;
}

extern "C" int unires_mark_read(const unires_mark_t *m, uint64_t *value) {
  if (!m || !value) return fail(UNIRES_ERR_NULL, "null argument");
  *value = (uint64_t)__atomic_load_n(m->host, __ATOMIC_ACQUIRE);
  return UNIRES_OK;
}
