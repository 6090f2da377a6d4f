// cg.hip - CG vector kernels (fused updates + float64 dot partials) and the
// one-block scalar kernels that turn partials into alpha / beta / objective /
// convergence flag ON THE DEVICE, so a whole nitorch-style cg() call
// (unires/_update.py:142-148) is enqueued without a single host sync.
//
// Numerics follow the reference: products are rounded to float32 and summed
// in float64 (torch.sum(p * Ap, dtype=float64)); alpha/beta are float64 scalars
// rounded to float32 at use (0-d tensor promotion); x += alpha*p is a rounded
// multiply followed by a rounded add (no FMA contraction).
#include "cg.hpp"

namespace unires {

static inline int vec_blocks(size_t n) {
  size_t b = (n / 4 + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  const size_t cap = 4096;  // enough workgroups to saturate HBM
  return (int)(b < cap ? b : cap);
}

#define GRID_STRIDE_VEC4(n)                                                   \
  const size_t n4 = (n) / 4;                                                  \
  const size_t stride = (size_t)gridDim.x * blockDim.x;                       \
  const size_t tid0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x

__device__ __forceinline__ float4 ld4(const float *p, size_t i) {
  return reinterpret_cast<const float4 *>(p)[i];
}
__device__ __forceinline__ void st4(float *p, size_t i, float4 v) {
  reinterpret_cast<float4 *>(p)[i] = v;
}
// Streaming (non-temporal) forms for x.  The iterate is touched once per CG iteration (read + write in
// k_update_p) and by nothing else, while p, A p and r are each re-read right after they are written (p by
// the matvec, A p and r by the next update).  At 256^3 the four vectors are 268 MB against 256 MB of
// Infinity Cache: with x allocating like the others the start of the freshly written p was gone before
// the pull read it.  Measured (config 3): 4 430 -> 4 620 CG iterations/s; k_splat2 80.6 -> 76.6 us inside
// the solve.  (The same hint on the last reads of A p and r: no further gain / slightly worse.)
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float *p, size_t i) {
#ifdef UNIRES_CG_X_PLAIN
  return ld4(p, i);
#else
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p) + i);
  return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st4_stream(float *p, size_t i, float4 v) {
#ifdef UNIRES_CG_X_PLAIN
  st4(p, i, v);
#else
  __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p) + i);
#endif
}

// obj term of nitorch cg for stop != 'e': (A(x) - 2b) * x, rounded like

// z = precond(r): identity (M == nullptr) or Jacobi r / M (unires/_update.py:80-102: x / M)
__device__ __forceinline__ float4 zval4(float4 r, const float *__restrict__ M, size_t i) {
  if (!M) return r;
  const float4 m = ld4(M, i);
  return make_float4(__fdiv_rn(r.x, m.x), __fdiv_rn(r.y, m.y), __fdiv_rn(r.z, m.z),
                     __fdiv_rn(r.w, m.w));
}
__device__ __forceinline__ float zval1(float r, const float *__restrict__ M, size_t i) {
  return M ? __fdiv_rn(r, M[i]) : r;
}

// r = b - A(x); z = precond(r); p = z; partial[0..G) = sum r*z; partial2 = sum (Ax-2b)*x (optional)
__global__ void __launch_bounds__(kBlock)
    k_residual_init(const float *__restrict__ b, const float *__restrict__ ax,
                    const float *__restrict__ x, float *__restrict__ r, float *__restrict__ p,
                    size_t n, double *__restrict__ part_rr, double *__restrict__ part_obj,
                    const float *__restrict__ M) {
  GRID_STRIDE_VEC4(n);
  double rr = 0.0, ob = 0.0;
  for (size_t i = tid0; i < n4; i += stride) {
    const float4 vb = ld4(b, i), va = ld4(ax, i);
    float4 vr;
    vr.x = __fsub_rn(vb.x, va.x);
    vr.y = __fsub_rn(vb.y, va.y);
    vr.z = __fsub_rn(vb.z, va.z);
    vr.w = __fsub_rn(vb.w, va.w);
    st4(r, i, vr);
    const float4 vz = zval4(vr, M, i);
    st4(p, i, vz);
    rr += (double)__fmul_rn(vr.x, vz.x) + (double)__fmul_rn(vr.y, vz.y) +
          (double)__fmul_rn(vr.z, vz.z) + (double)__fmul_rn(vr.w, vz.w);
    if (part_obj) {
      const float4 vx = ld4(x, i);
      ob += (double)obj_term(va.x, vb.x, vx.x) + (double)obj_term(va.y, vb.y, vx.y) +
            (double)obj_term(va.z, vb.z, vx.z) + (double)obj_term(va.w, vb.w, vx.w);
    }
  }
  for (size_t i = n4 * 4 + tid0; i < n; i += stride) {  // tail
    const float vr = __fsub_rn(b[i], ax[i]);
    r[i] = vr;
    const float vz = zval1(vr, M, i);
    p[i] = vz;
    rr += (double)__fmul_rn(vr, vz);
    if (part_obj) ob += (double)obj_term(ax[i], b[i], x[i]);
  }
  const double t = block_sum(rr);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = t;
  if (part_obj) {
    const double t2 = block_sum(ob);
    if (threadIdx.x == 0) part_obj[blockIdx.x] = t2;
  }
}

// partial = sum a*b
__global__ void __launch_bounds__(kBlock)
    k_dot(const float *__restrict__ a, const float *__restrict__ b, size_t n,
          double *__restrict__ part, const int *__restrict__ done) {
  if (done && *done) return;
  GRID_STRIDE_VEC4(n);
  double acc = 0.0;
  for (size_t i = tid0; i < n4; i += stride) {
    const float4 va = ld4(a, i), vb = ld4(b, i);
    acc += (double)__fmul_rn(va.x, vb.x) + (double)__fmul_rn(va.y, vb.y) +
           (double)__fmul_rn(va.z, vb.z) + (double)__fmul_rn(va.w, vb.w);
  }
  for (size_t i = n4 * 4 + tid0; i < n; i += stride) acc += (double)__fmul_rn(a[i], b[i]);
  const double t = block_sum(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// x += alpha p ; r -= alpha Ap ; part_rr = sum r*r ;
// part_obj (optional) = sum x*(b+r)  (recurred objective -0.5*sum x (b+r))
// XUPD = false: only r is updated here and "x += alpha p" rides along with the p update
// (k_update_p<true>), which reads the OLD p anyway: 3 + 5 instead of 6 + 3 volume passes per
// iteration, same operations and roundings.
template <bool XUPD>
__global__ void __launch_bounds__(kBlock)
    k_update_xr(const CgState *__restrict__ st, const float *__restrict__ p,
                const float *__restrict__ ap, float *__restrict__ x, float *__restrict__ r,
                const float *__restrict__ b, size_t n, double *__restrict__ part_rr,
                double *__restrict__ part_obj, const float *__restrict__ M) {
  if (st->done) return;
  const float alpha = (float)st->alpha;
  GRID_STRIDE_VEC4(n);
  double rr = 0.0, ob = 0.0;
  for (size_t i = tid0; i < n4; i += stride) {
    const float4 va = ld4(ap, i);
    float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vr = ld4(r, i);
    if (XUPD) {
      const float4 vp = ld4(p, i);
      vx = ld4(x, i);
      vx.x = __fadd_rn(vx.x, __fmul_rn(alpha, vp.x));
      vx.y = __fadd_rn(vx.y, __fmul_rn(alpha, vp.y));
      vx.z = __fadd_rn(vx.z, __fmul_rn(alpha, vp.z));
      vx.w = __fadd_rn(vx.w, __fmul_rn(alpha, vp.w));
      st4(x, i, vx);
    }
    vr.x = __fsub_rn(vr.x, __fmul_rn(alpha, va.x));
    vr.y = __fsub_rn(vr.y, __fmul_rn(alpha, va.y));
    vr.z = __fsub_rn(vr.z, __fmul_rn(alpha, va.z));
    vr.w = __fsub_rn(vr.w, __fmul_rn(alpha, va.w));
    st4(r, i, vr);
    const float4 vz = zval4(vr, M, i);
    rr += (double)__fmul_rn(vr.x, vz.x) + (double)__fmul_rn(vr.y, vz.y) +
          (double)__fmul_rn(vr.z, vz.z) + (double)__fmul_rn(vr.w, vz.w);
    if (XUPD && part_obj) {
      const float4 vb = ld4(b, i);
      ob += (double)__fmul_rn(vx.x, __fadd_rn(vb.x, vr.x)) +
            (double)__fmul_rn(vx.y, __fadd_rn(vb.y, vr.y)) +
            (double)__fmul_rn(vx.z, __fadd_rn(vb.z, vr.z)) +
            (double)__fmul_rn(vx.w, __fadd_rn(vb.w, vr.w));
    }
  }
  for (size_t i = n4 * 4 + tid0; i < n; i += stride) {
    float vx = 0.f;
    if (XUPD) {
      vx = __fadd_rn(x[i], __fmul_rn(alpha, p[i]));
      x[i] = vx;
    }
    const float vr = __fsub_rn(r[i], __fmul_rn(alpha, ap[i]));
    r[i] = vr;
    rr += (double)__fmul_rn(vr, zval1(vr, M, i));
    if (XUPD && part_obj) ob += (double)__fmul_rn(vx, __fadd_rn(b[i], vr));
  }
  const double t = block_sum(rr);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = t;
  if (XUPD && part_obj) {
    const double t2 = block_sum(ob);
    if (threadIdx.x == 0) part_obj[blockIdx.x] = t2;
  }
}

// p = beta*p + z   (p *= beta; p += z);  XUPD: first x += alpha * p with the old p
template <bool XUPD>
__global__ void __launch_bounds__(kBlock)
    k_update_p(const CgState *__restrict__ st, const float *__restrict__ r, float *__restrict__ p,
               size_t n, const float *__restrict__ M, float *__restrict__ x) {
  if (st->done) return;
  const float beta = (float)st->beta, alpha = (float)st->alpha;
  GRID_STRIDE_VEC4(n);
  for (size_t i = tid0; i < n4; i += stride) {
    const float4 vr = zval4(ld4(r, i), M, i);
    float4 vp = ld4(p, i);
    if (XUPD) {
      float4 vx = ld4_stream(x, i);
      vx.x = __fadd_rn(vx.x, __fmul_rn(alpha, vp.x));
      vx.y = __fadd_rn(vx.y, __fmul_rn(alpha, vp.y));
      vx.z = __fadd_rn(vx.z, __fmul_rn(alpha, vp.z));
      vx.w = __fadd_rn(vx.w, __fmul_rn(alpha, vp.w));
      st4_stream(x, i, vx);
    }
    vp.x = __fadd_rn(__fmul_rn(beta, vp.x), vr.x);
    vp.y = __fadd_rn(__fmul_rn(beta, vp.y), vr.y);
    vp.z = __fadd_rn(__fmul_rn(beta, vp.z), vr.z);
    vp.w = __fadd_rn(__fmul_rn(beta, vp.w), vr.w);
    st4(p, i, vp);
  }
  for (size_t i = n4 * 4 + tid0; i < n; i += stride) {
    const float vp = p[i];
    if (XUPD) x[i] = __fadd_rn(x[i], __fmul_rn(alpha, vp));
    p[i] = __fadd_rn(__fmul_rn(beta, vp), zval1(r[i], M, i));
  }
}

// y = a*x + y (generic axpy; used by the identity regime's RHS)
__global__ void __launch_bounds__(kBlock)
    k_axpy(float a, const float *__restrict__ x, float *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = fmaf(a, x[i], y[i]);
}

// ---- folded forms: the scalar steps ride in the prologues of the vector kernels ------------
// (r2 tried the opposite fold - the LAST workgroup of the producer sums the partials - which needs
// an agent-scope release per workgroup, an L2 write-back each: 4 200 -> 1 550 iterations/s.  Here
// nothing crosses workgroups inside a launch: every consumer workgroup reads the producer's <= 8192
// partials (L2 hits, 64 KB at most) after the kernel boundary and sums them in the same fixed
// order.)
__device__ __forceinline__ double reduce_all(const double *__restrict__ part, int g) {
  __shared__ double s_tot[kBlock / kWave];
  double v = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * kBlock < g; i += 8 * kBlock) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[i + u * kBlock];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; i < g; i += kBlock) v += part[i];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) s_tot[threadIdx.x / kWave] = v;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) tot += s_tot[w];
  return tot;  // the same value, bit for bit, in every thread of every workgroup
}

// Chunked form of the fused updates: a workgroup takes chunks of kFoldIt * kBlock float4, ALL loads of
// a chunk are issued before anything else, and the partial-sum reduction of the prologue runs behind
// them (first chunk only): its round trip (L2 reads of the partials, one LDS exchange) overlaps the
// latency of the chunk's own loads instead of preceding it.  4096 workgroups, as the unfolded kernels
// (1024 persistent ones lost 12 % of the streaming rate: 57 instead of 50 us for the x / p update).
constexpr int kFoldIt = 4;
static inline int vec_blocks_fold(size_t n) {
  size_t b = (n / 4 + (size_t)kFoldIt * kBlock - 1) / ((size_t)kFoldIt * kBlock);
  if (b < 1) b = 1;
  return (int)(b < 4096 ? b : 4096);
}

// r update: 1024 persistent workgroups (3 volume passes keep the streaming rate at this size; its
// <= 1024 partial sums are what the 4096 workgroups of the x / p update re-reduce - 4 loads per thread)
static inline int vec_blocks_rfold(size_t n) {
  size_t b = (n / 4 + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  return (int)(b < 1024 ? b : 1024);
}
__global__ void __launch_bounds__(kBlock)
    k_update_r_fold(CgState *__restrict__ st, const double *__restrict__ part_pap, int g, int k,
                    const float *__restrict__ ap, float *__restrict__ r, size_t n,
                    double *__restrict__ part_rr, const float *__restrict__ M) {
  if (st->done) return;
  const double pap = reduce_all(part_pap, g);
  const double alpha_d = st->rzpp[(k - 1) & 1] / pap;
  if (blockIdx.x == 0 && threadIdx.x == 0) st->pAp = pap, st->alpha = alpha_d;
  const float alpha = (float)alpha_d;
  GRID_STRIDE_VEC4(n);
  double rr = 0.0;
  for (size_t i = tid0; i < n4; i += stride) {
    const float4 va = ld4(ap, i);
    float4 vr = ld4(r, i);
    vr.x = __fsub_rn(vr.x, __fmul_rn(alpha, va.x));
    vr.y = __fsub_rn(vr.y, __fmul_rn(alpha, va.y));
    vr.z = __fsub_rn(vr.z, __fmul_rn(alpha, va.z));
    vr.w = __fsub_rn(vr.w, __fmul_rn(alpha, va.w));
    st4(r, i, vr);
    const float4 vz = zval4(vr, M, i);
    rr += (double)__fmul_rn(vr.x, vz.x) + (double)__fmul_rn(vr.y, vz.y) +
          (double)__fmul_rn(vr.z, vz.z) + (double)__fmul_rn(vr.w, vz.w);
  }
  for (size_t i = n4 * 4 + tid0; i < n; i += stride) {
    const float vr = __fsub_rn(r[i], __fmul_rn(alpha, ap[i]));
    r[i] = vr;
    rr += (double)__fmul_rn(vr, zval1(vr, M, i));
  }
  const double t = block_sum(rr);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = t;
}

__global__ void __launch_bounds__(kBlock)
    k_update_px_fold(CgState *__restrict__ st, const double *__restrict__ part_rr, int g, int k,
                     const float *__restrict__ r, float *__restrict__ p, float *__restrict__ x, size_t n,
                     const float *__restrict__ M) {
  if (st->done) return;
  const size_t n4 = n / 4, chunk = (size_t)kFoldIt * kBlock;
  float alpha = 0.f, beta = 0.f;
  bool have = false;
  for (size_t c = blockIdx.x; c * chunk < n4 || !have; c += gridDim.x) {
    float4 vz[kFoldIt], vp[kFoldIt], vx[kFoldIt];
    size_t idx[kFoldIt];
#pragma unroll
    for (int u = 0; u < kFoldIt; ++u) {
      idx[u] = c * chunk + (size_t)u * kBlock + threadIdx.x;
      if (idx[u] < n4) vz[u] = zval4(ld4(r, idx[u]), M, idx[u]), vp[u] = ld4(p, idx[u]), vx[u] = ld4(x, idx[u]);
    }
    if (!have) {
      const double rrs = reduce_all(part_rr, g);
      const double rz0 = st->rzpp[(k - 1) & 1];
      const double beta_d = rrs / rz0;
      beta = (float)beta_d, alpha = (float)st->alpha;
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rzpp[k & 1] = rrs;
        st->rz = rrs;
        st->beta = beta_d;
        st->iters = k;
      }
      have = true;
    }
#pragma unroll
    for (int u = 0; u < kFoldIt; ++u) {
      if (idx[u] < n4) {
        float4 a = vx[u], b = vp[u];
        a.x = __fadd_rn(a.x, __fmul_rn(alpha, b.x));
        a.y = __fadd_rn(a.y, __fmul_rn(alpha, b.y));
        a.z = __fadd_rn(a.z, __fmul_rn(alpha, b.z));
        a.w = __fadd_rn(a.w, __fmul_rn(alpha, b.w));
        st4(x, idx[u], a);
        b.x = __fadd_rn(__fmul_rn(beta, b.x), vz[u].x);
        b.y = __fadd_rn(__fmul_rn(beta, b.y), vz[u].y);
        b.z = __fadd_rn(__fmul_rn(beta, b.z), vz[u].z);
        b.w = __fadd_rn(__fmul_rn(beta, b.w), vz[u].w);
        st4(p, idx[u], b);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
    const size_t i = n4 * 4 + threadIdx.x;
    const float vp = p[i];
    x[i] = __fadd_rn(x[i], __fmul_rn(alpha, vp));
    p[i] = __fadd_rn(__fmul_rn(beta, vp), zval1(r[i], M, i));
  }
}

// ---- scalar kernels: <<<1, kBlock>>> -------------------------------------
// (Tried and dropped, r2: folding k_sc_alpha / k_sc_beta into the kernels that write their
// partials - every workgroup takes a ticket, the last one sums all partials in a fixed order.  The
// ticket needs an agent-scope release so that the partials of workgroups on other XCDs are visible,
// and on gfx950 that is an L2 write-back per workgroup: config 3 fell from 4 200 to 1 550 CG
// iterations/s, the aligned variant from 6 400 to 1 800.  Two 5 us launches per iteration it is.)
__device__ __forceinline__ double sum_partials(const double *part, int g) {
  // loads batched eight deep (the additions keep their order): this one-block kernel is pure
  // latency, and sixteen dependent load->add steps per thread were most of its 6 us
  double v = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * kBlock < g; i += 8 * kBlock) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[i + u * kBlock];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; i < g; i += kBlock) v += part[i];
  return block_sum(v);
}

// (r6) ... and the loads in front of everything else.  A scalar kernel is a chain of trips to memory - the stop flag,
// two batches of partials, the previous scalar - behind a ~1.5 us launch: 4.7 us for a kernel that adds 4 096 numbers,
// twice per CG iteration.  part_issue() requests a thread's first sixteen partials (all of them for up to 4 096) BEFORE
// the kernel looks at its stop flag; part_finish() adds them in the order sum_partials() does (i = tid, tid + 256, ...:
// the same sums, bit for bit) and goes on from memory for longer lists.  The state's scalars are read in the same trip.
// A kernel that finds its solve done does NOT return early (the compiler sinks the requests below such a branch, and the
// chain is back): it sums what it loaded and writes nothing.
constexpr int kPartDeep = 16;
struct PartLoad {
  double t[kPartDeep];
};
__device__ __forceinline__ void part_issue(const double *part, int g, PartLoad &L) {
#pragma unroll
  for (int u = 0; u < kPartDeep; ++u) {
    // (every thread loads - past the list's end the last entry again, never added: no branch between the requests)
    const int i = min((int)threadIdx.x + u * kBlock, max(g - 1, 0));
    L.t[u] = part[i];
  }
}
__device__ __forceinline__ double part_finish(const double *part, int g, const PartLoad &L) {
  double v = 0.0;
#pragma unroll
  for (int u = 0; u < kPartDeep; ++u)
    if ((int)threadIdx.x + u * kBlock < g) v += L.t[u];
  for (int i = (int)threadIdx.x + kPartDeep * kBlock; i < g; i += kBlock) v += part[i];
  return block_sum(v);
}

// nitorch get_gain(obj[:k+1], 'decreasing') and the |gain| < tol test
__device__ __forceinline__ void publish(const CgState *st, unsigned long long *hostw) {
  if (hostw)
    __hip_atomic_store(hostw, cg_progress_word(st->gen, st->done, st->iters), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void record_obj(CgState *st, int k, double obj, double tol) {
  constexpr int kRing = kMaxCgIter + 1;
  const double prev = k > 0 ? st->obj[(k - 1) % kRing] : 0.0;
  st->obj[k % kRing] = obj;
  if (k == 0) {
    st->obj_max = obj;
    st->obj_min = obj;
    return;
  }
  st->obj_max = fmax(st->obj_max, obj);
  st->obj_min = fmin(st->obj_min, obj);
  const double gain = (prev - obj) / (st->obj_max - st->obj_min);
  if (fabs(gain) < tol) st->done = 1;  // NaN compares false, like torch
  // (a non-finite objective never passes that test: nitorch would grind through its whole budget - 10 numel
  // iterations by default - on NaNs; the device-resident solve stops instead and reports where)
  if (!isfinite(obj)) st->done = 1;
}

__global__ void __launch_bounds__(kBlock)
    k_sc_init(CgState *st, const double *part_rr, const double *part_obj, int g, int mode,
              int check, unsigned long long *hostw) {
  const double rr = sum_partials(part_rr, g);
  double ob = 0.0;
  if (check && mode != UNIRES_STOP_RESIDUAL) ob = sum_partials(part_obj, g);
  if (threadIdx.x == 0) {
    st->rz = rr;
    st->rzpp[0] = rr;
    st->rzpp[1] = rr;
    st->done = 0;
    st->iters = 0;
    st->alpha = 0.0;
    st->beta = 0.0;
    st->gen += 1u;
    if (check) record_obj(st, 0, mode == UNIRES_STOP_RESIDUAL ? sqrt(rr) : 0.5 * ob, 0.0);
    // (the objective of the start is a fresh one: r = b - A(x) was just computed)
    st->skip_fresh = 1, st->fresh_prev_ok = 1, st->rec_prev = 0.5 * ob, st->fresh_prev = 0.5 * ob, st->gain_rec = 0.0;
    publish(st, hostw);
  }
}

__global__ void __launch_bounds__(kBlock) k_sc_alpha(CgState *st, const double *part_pap, int g) {
  PartLoad L;
  part_issue(part_pap, g, L);
  const int done = st->done;
  const double rz = st->rz;
  const double pap = part_finish(part_pap, g, L);
  if (threadIdx.x == 0 && !done) {
    st->pAp = pap;
    st->alpha = rz / pap;
  }
}

// obj_kind: 0 none, 1 sqrt(rz) ('e'), 2 recurred (-0.5 * sum x(b+r))
__global__ void __launch_bounds__(kBlock)
    k_sc_beta(CgState *st, const double *part_rr, const double *part_obj, int g, int k,
              int obj_kind, double tol, unsigned long long *hostw) {
  PartLoad Lr, Lo;
  part_issue(part_rr, g, Lr);
  if (obj_kind == 2) part_issue(part_obj, g, Lo);
  const int done = st->done, iters = st->iters;
  const double rz0 = st->rz;
  if (k < 0) k = iters + 1;
  const double rr = part_finish(part_rr, g, Lr);
  double ob = 0.0;
  if (obj_kind == 2) ob = part_finish(part_obj, g, Lo);
  if (threadIdx.x == 0 && !done) {
    st->rz = rr;
    st->rzpp[k & 1] = rr;
    st->beta = rr / rz0;
    st->iters = k;
    if (obj_kind == 1) record_obj(st, k, sqrt(rr), tol);
    if (obj_kind == 2) record_obj(st, k, -0.5 * ob, tol);
    publish(st, hostw);
  }
}

__global__ void __launch_bounds__(kBlock)
    k_sc_obj(CgState *st, const double *part_obj, int g, int k, double tol, unsigned long long *hostw) {
  PartLoad L;
  part_issue(part_obj, g, L);
  const int done = st->done, iters = st->iters;
  if (k < 0) k = iters;
  const double ob = part_finish(part_obj, g, L);
  if (threadIdx.x == 0 && !done) {
    record_obj(st, k, 0.5 * ob, tol);
    publish(st, hostw);
  }
}

// Guarded 'max_gain' (UNIRES_STOP_MAXGAIN_GUARDED).  nitorch decides |gain| < tol on the objective 0.5 sum x (A(x) - 2b):
// a second A(x) per iteration.  The same objective follows from the recurred residual, -0.5 sum x (b + r), for free;
// the two differ by 0.5 x.(r_recurred - r_true), rounding drift.  Far from the threshold the drift cannot change the
// decision: while the recurred gain is >= kGuardHi x tol the solve goes on without the second A(x).  Below that the
// fresh objective is computed every iteration, so that by the time a decision is close BOTH values in the gain are
// fresh ones - nitorch's own arithmetic; where the previous fresh value is missing (the gain fell through the band in
// one step, short solves) the consistent recurred pair decides.  And below the band: a recurred gain under
// kGuardLo x tol stops the solve there and then - the fresh gain is below tol as well, by a margin (tol / 2) that
// is ~500 x the drift (tests/test_gpu_guard.py bounds it at 1e-5 of the range, measured ~1e-7) - so a solve whose
// gain falls through the band in one step pays no second A(x) at all (config 3, first ADMM iteration: gains / tol
// of the three channels 1000 .. 3.03 1.56 0.827 | 1000 0.338 | 1000 1.14 0.129 - six fresh objectives become four).
// The band's upper edge follows the solve (r6): what it is for is that the iteration BEFORE a close decision has a
// fresh objective, i.e. that a gain which may be followed by one near tol is inside the band.  Gains contract by a
// factor rho from one iteration to the next - 0.1 .. 0.5 in the first y-updates of a reconstruction, 0.8 .. 0.95 in
// the later, long solves (profiles/r06_gains.txt) - so the edge sits where the PREDICTED next gain, rho x gain, is
// still kGuardNear x tol away from the threshold: hi = kGuardNear / rho, between kGuardNear and kGuardHi.  A long
// solve creeping towards tol (gains / tol ... 3.2 3.0 2.6 2.2 1.85 1.47 1.30 1.26 1.16 1.07) pays 5 fresh objectives
// where the fixed edge at 4 paid 11; a solve that falls through in two steps keeps the edge at 4.
constexpr double kGuardHi = 4.0, kGuardLo = 0.5, kGuardNear = 1.5;
__global__ void __launch_bounds__(kBlock)
    k_sc_beta_guarded(CgState *st, const double *part_rr, const double *part_obj, int g, int k, double tol,
                      unsigned long long *hostw) {
  PartLoad Lr, Lo;
  part_issue(part_rr, g, Lr);
  part_issue(part_obj, g, Lo);
  const int done = st->done, iters = st->iters;
  const double rz0 = st->rz;
  if (k < 0) k = iters + 1;
  const double rr = part_finish(part_rr, g, Lr);
  const double ob = part_finish(part_obj, g, Lo);
  if (threadIdx.x == 0 && !done) {
    constexpr int kRing = kMaxCgIter + 1;
    st->rz = rr;
    st->rzpp[k & 1] = rr;
    st->beta = rr / rz0;
    st->iters = k;
    const double rec = -0.5 * ob;
    st->obj[k % kRing] = rec;  // (replaced by the fresh value if that is computed)
    st->obj_max = fmax(st->obj_max, rec);
    st->obj_min = fmin(st->obj_min, rec);
    const double gain = (st->rec_prev - rec) / (st->obj_max - st->obj_min);
    const double gain_before = fabs(st->gain_rec);  // (0 in front of the first iteration)
    st->gain_rec = gain;
    st->rec_prev = rec;
    double hi = kGuardHi;
    if (gain_before > 0.0 && isfinite(gain_before) && isfinite(gain)) {
      const double rho = fabs(gain) / gain_before;  // (> 1 where the gains are not monotone: the edge stays at kGuardNear)
      hi = fmin(kGuardHi, fmax(kGuardNear, kGuardNear / fmax(rho, 1e-6)));
    }
    const bool far = fabs(gain) >= hi * tol;  // (NaN: not far - the fresh objective decides, as in nitorch)
    const bool under = fabs(gain) < kGuardLo * tol;  // (NaN: not under)
    st->skip_fresh = (far || under) ? 1 : 0;
    if (far) st->fresh_prev_ok = 0;
    if (under) st->done = 1;  // converged by a margin the drift cannot bridge
    if (!isfinite(rec)) st->done = 1, st->skip_fresh = 1;
    publish(st, hostw);
  }
}

__global__ void __launch_bounds__(kBlock)
    k_sc_obj_guarded(CgState *st, const double *part_obj, int g, int k, double tol, unsigned long long *hostw) {
  PartLoad L;
  part_issue(part_obj, g, L);
  const int done = st->done, skip = st->skip_fresh, iters = st->iters;
  if (k < 0) k = iters;
  const double ob = part_finish(part_obj, g, L);
  if (threadIdx.x == 0 && !done && !skip) {
    constexpr int kRing = kMaxCgIter + 1;
    const double fresh = 0.5 * ob;
    st->obj[k % kRing] = fresh;
    st->obj_max = fmax(st->obj_max, fresh);
    st->obj_min = fmin(st->obj_min, fresh);
    const double gain = st->fresh_prev_ok ? (st->fresh_prev - fresh) / (st->obj_max - st->obj_min) : st->gain_rec;
    st->fresh_prev = fresh, st->fresh_prev_ok = 1;
    if (fabs(gain) < tol || !isfinite(fresh)) st->done = 1, st->skip_fresh = 1;
    publish(st, hostw);
  }
}

__global__ void __launch_bounds__(kBlock) k_sum_to(const double *part, int g, double *out) {
  const double v = sum_partials(part, g);
  if (threadIdx.x == 0) *out = v;
}

// ---- launchers -------------------------------------------------------------
int vec_num_blocks(size_t n) { return vec_blocks(n); }
int vec_num_blocks_fold(size_t n) { return vec_blocks_rfold(n); }  // partials the r update writes
void launch_update_r_fold(CgState *s, const double *part_pap, int g, int k, const float *ap, float *r,
                          size_t n, double *part_rr, const float *M, hipStream_t st) {
  hipLaunchKernelGGL(k_update_r_fold, dim3(vec_blocks_rfold(n)), dim3(kBlock), 0, st, s, part_pap, g, k, ap, r,
                     n, part_rr, M);
}
void launch_update_px_fold(CgState *s, const double *part_rr, int g, int k, const float *r, float *p,
                           float *x, size_t n, const float *M, hipStream_t st) {
  hipLaunchKernelGGL(k_update_px_fold, dim3(vec_blocks_fold(n)), dim3(kBlock), 0, st, s, part_rr, g, k, r, p,
                     x, n, M);
}

void launch_residual_init(const float *b, const float *ax, const float *x, float *r, float *p,
                          size_t n, double *part_rr, double *part_obj, const float *M,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_residual_init, dim3(vec_blocks(n)), dim3(kBlock), 0, st, b, ax, x, r, p, n,
                     part_rr, part_obj, M);
}
void launch_dot(const float *a, const float *b, size_t n, double *part, const int *done,
                hipStream_t st) {
  hipLaunchKernelGGL(k_dot, dim3(vec_blocks(n)), dim3(kBlock), 0, st, a, b, n, part, done);
}
void launch_update_xr(const CgState *s, const float *p, const float *ap, float *x, float *r,
                      const float *b, size_t n, double *part_rr, double *part_obj,
                      const float *M, hipStream_t st) {
  if (x)
    hipLaunchKernelGGL(k_update_xr<true>, dim3(vec_blocks(n)), dim3(kBlock), 0, st, s, p, ap, x, r, b,
                       n, part_rr, part_obj, M);
  else
    hipLaunchKernelGGL(k_update_xr<false>, dim3(vec_blocks(n)), dim3(kBlock), 0, st, s, p, ap, x, r,
                       b, n, part_rr, part_obj, M);
}
void launch_update_p(const CgState *s, const float *r, float *p, size_t n, const float *M, float *x,
                     hipStream_t st) {
  if (x)
    hipLaunchKernelGGL(k_update_p<true>, dim3(vec_blocks(n)), dim3(kBlock), 0, st, s, r, p, n, M, x);
  else
    hipLaunchKernelGGL(k_update_p<false>, dim3(vec_blocks(n)), dim3(kBlock), 0, st, s, r, p, n, M, x);
}
// y = a*y + c  (preconditioner diagonal: tau * AtA(1) + const)
__global__ void __launch_bounds__(kBlock) k_scale_shift(float a, float c, float *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = __fadd_rn(__fmul_rn(a, y[i]), c);
}
void launch_scale_shift(float a, float c, float *y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_scale_shift, dim3(vec_blocks(n) * 4 > 4096 ? 4096 : vec_blocks(n) * 4),
                     dim3(kBlock), 0, st, a, c, y, n);
}
__global__ void __launch_bounds__(kBlock) k_fill(float v, float *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = v;
}
void launch_fill(float v, float *y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_fill, dim3(vec_blocks(n) * 4 > 4096 ? 4096 : vec_blocks(n) * 4), dim3(kBlock), 0,
                     st, v, y, n);
}
__global__ void __launch_bounds__(kBlock)
    k_divide(const float *__restrict__ a, const float *__restrict__ m, float *__restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = __fdiv_rn(a[i], m[i]);
}
void launch_div(const float *a, const float *m, float *y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_divide, dim3(vec_blocks(n) * 4 > 4096 ? 4096 : vec_blocks(n) * 4), dim3(kBlock), 0,
                     st, a, m, y, n);
}
void launch_axpy(float a, const float *x, float *y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_axpy, dim3(vec_blocks(n) * 4 > 4096 ? 4096 : vec_blocks(n) * 4),
                     dim3(kBlock), 0, st, a, x, y, n);
}
void launch_sc_init(CgState *s, const double *part_rr, const double *part_obj, int g, int mode,
                    int check, unsigned long long *hostw, hipStream_t st) {
  hipLaunchKernelGGL(k_sc_init, dim3(1), dim3(kBlock), 0, st, s, part_rr, part_obj, g, mode,
                     check, hostw);
}
void launch_sc_alpha(CgState *s, const double *part, int g, hipStream_t st) {
  hipLaunchKernelGGL(k_sc_alpha, dim3(1), dim3(kBlock), 0, st, s, part, g);
}
void launch_sc_beta(CgState *s, const double *part_rr, const double *part_obj, int g, int k,
                    int obj_kind, double tol, unsigned long long *hostw, hipStream_t st) {
  hipLaunchKernelGGL(k_sc_beta, dim3(1), dim3(kBlock), 0, st, s, part_rr, part_obj, g, k,
                     obj_kind, tol, hostw);
}
void launch_sc_obj(CgState *s, const double *part, int g, int k, double tol, unsigned long long *hostw,
                   hipStream_t st) {
  hipLaunchKernelGGL(k_sc_obj, dim3(1), dim3(kBlock), 0, st, s, part, g, k, tol, hostw);
}
void launch_sc_beta_guarded(CgState *s, const double *part_rr, const double *part_obj, int g, int k, double tol,
                            unsigned long long *hostw, hipStream_t st) {
  hipLaunchKernelGGL(k_sc_beta_guarded, dim3(1), dim3(kBlock), 0, st, s, part_rr, part_obj, g, k, tol, hostw);
}
void launch_sc_obj_guarded(CgState *s, const double *part, int g, int k, double tol, unsigned long long *hostw,
                           hipStream_t st) {
  hipLaunchKernelGGL(k_sc_obj_guarded, dim3(1), dim3(kBlock), 0, st, s, part, g, k, tol, hostw);
}
void launch_sum_to(const double *part, int g, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(kBlock), 0, st, part, g, out);
}

}  // namespace unires
