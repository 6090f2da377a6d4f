// admm.hip - the updates that follow the y-update in every ADMM iteration
// (unires/_update.py:154-195): joint-total-variation
// shrinkage of z, dual ascent on w, and the objective's prior / likelihood sums.
//
// The reference evaluates im_gradient three times per channel and materialises a
// dozen volume-sized temporaries; here Dy is recomputed in registers:
//   k_jtv_scale : s(v) = max(n - 1/rho, 0) / (n + 1e-7),  n = sqrt(sum_c |w_c/rho + lam_c D y_c|^2)
//   k_zw_update : z_c = s * (w_c/rho + lam_c D y_c);  w_c += rho * (lam_c D y_c - z_c)
// (alpha == 1 path; over-relaxation is a per-voxel blend with z_old and is built too).
#include <string.h>

#include <algorithm>

#include "admm.hpp"

namespace unires {

struct ChanPtrs {  // up to 8 channels per launch; more channels run as chained launches
  const float *y[8];
  float lam[8];
  int n;
  int c0;     // index of y[0] among the subject's channels (offset into w / z_old)
  int first;  // 0: the squared magnitude of the earlier channels is read from `scale`
  int last;   // 0: the running squared magnitude is stored to `scale` instead of the final result
};

// forward differences of y at (i,j,k), zero bound, times lam / vx
__device__ __forceinline__ void grad_at(const float *__restrict__ y, size_t idx, int i, int j,
                                        int k, const Dim3i &d, float sx, float sy, float sz,
                                        float &gx, float &gy, float &gz) {
  const size_t px = (size_t)d.y * d.z, py = d.z;
  const bool hx = i + 1 < d.x, hy = j + 1 < d.y, hz = k + 1 < d.z;
  const float c = y[idx];
  const float vx = y[hx ? idx + px : idx], vy = y[hy ? idx + py : idx], vz = y[hz ? idx + 1 : idx];
  gx = ((hx ? vx : 0.f) - c) * sx;
  gy = ((hy ? vy : 0.f) - c) * sy;
  gz = ((hz ? vz : 0.f) - c) * sz;
}

// s = shrinkage factor of the joint TV norm (written to `scale`); optional partial of
// sum(n) (the -ln p(y) term of the objective when called with w = 0, rho = 1).
// A workgroup owns one 64 x 4 patch of (z, y) and walks along x in steps of gridDim.z planes: no index
// divisions (r3's flat tile loop paid two 64-bit divisions per tile: 1.6 - 2.9 TB/s); NC channels compiled
// in (0: any number up to 8), so that the 7 loads per channel and voxel of all channels are in flight
// together; the sum over channels keeps its order and float32 arithmetic.
template <int NC>
__global__ void __launch_bounds__(kBlock)
    k_jtv_scale(ChanPtrs C, const float *__restrict__ w, const float *__restrict__ z_old, Dim3i d,
                float ivx, float ivy, float ivz, float rho, float alpha, float *__restrict__ scale,
                double *__restrict__ partials /* one per workgroup */, int norm_only) {
  const size_t n = d.numel();
  const float irho = 1.f / rho;
  double tot = 0.0;
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const bool inside = k < d.z && j < d.y;
  const int nc = NC ? NC : C.n;
  if (inside)
    for (int i = blockIdx.z; i < d.x; i += gridDim.z) {
      const size_t idx = ((size_t)i * d.y + j) * d.z + k;
      // (more than 8 channels: the sum over channels continues, in the same order and the same float32
      // arithmetic, from the value the previous launch left in `scale`)
      float acc = C.first ? 0.f : scale[idx];
#pragma unroll
      for (int c = 0; c < (NC ? NC : 8); ++c) {
        if (c >= nc) break;
        float gx, gy, gz;
        grad_at(C.y[c], idx, i, j, k, d, C.lam[c] * ivx, C.lam[c] * ivy, C.lam[c] * ivz, gx, gy, gz);
        const size_t o = (size_t)(C.c0 + c) * 3 * n + idx;
        if (alpha != 1.f) {  // Dy = alpha*Dy + (1-alpha)*z_old   (_update.py:169-170)
          gx = alpha * gx + (1.f - alpha) * z_old[o];
          gy = alpha * gy + (1.f - alpha) * z_old[o + n];
          gz = alpha * gz + (1.f - alpha) * z_old[o + 2 * n];
        }
        if (!norm_only) {
          gx += w[o] * irho, gy += w[o + n] * irho, gz += w[o + 2 * n] * irho;
        }
        acc += gx * gx + gy * gy + gz * gz;
      }
      if (!C.last) {
        scale[idx] = acc;
        continue;
      }
      const float nrm = sqrtf(acc);
      if (norm_only) {
        tot += (double)nrm;
      } else {
        scale[idx] = fmaxf(nrm - irho, 0.f) / (nrm + 1e-7f);
      }
    }
  if (partials && C.last) {  // one sum per workgroup, added in index order by k_sum_cols (no atomics: the same result every run)
    const double s = block_sum(tot);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
  }
}

// z_c = s * (w_c/rho + Dy_c);  w_c += rho * (Dy_c - z_c)     for ONE channel
__global__ void __launch_bounds__(kBlock)
    k_zw_update(const float *__restrict__ y, float lam, const float *__restrict__ scale,
                float *__restrict__ z, float *__restrict__ w, Dim3i d, float ivx, float ivy,
                float ivz, float rho, float alpha) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z;
  if (k >= d.z || j >= d.y) return;
  const size_t n = d.numel();
  const size_t idx = ((size_t)i * d.y + j) * d.z + k;
  float g[3];
  grad_at(y, idx, i, j, k, d, lam * ivx, lam * ivy, lam * ivz, g[0], g[1], g[2]);
  const float s = scale[idx], irho = 1.f / rho;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const size_t o = (size_t)a * n + idx;
    float dy = g[a];
    if (alpha != 1.f) dy = alpha * dy + (1.f - alpha) * z[o];
    const float wo = w[o];
    const float zn = s * (wo * irho + dy);
    z[o] = zn;
    w[o] = wo + rho * (dy - zn);
  }
}

// partial of sum_{x != 0} (x - Ay)^2   (masked likelihood term, _update.py:414-417)
__global__ void __launch_bounds__(kBlock)
    k_masked_sse(const float *__restrict__ x, const float *__restrict__ ay, size_t n,
                 double *__restrict__ partials) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float xv = x[i];
    if (xv != 0.f) {
      const float r = xv - ay[i];
      acc += (double)__fmul_rn(r, r);
    }
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// out[k] = sum_b part[b * ncols + k], b in index order (thread t takes b = t, t + 256, ...; block_sum's tree is fixed)
__global__ void __launch_bounds__(kBlock) k_sum_cols(const double *__restrict__ part, int nb, int ncols, double *__restrict__ out) {
  for (int k = 0; k < ncols; ++k) {
    double a = 0.0;
    for (int b = threadIdx.x; b < nb; b += kBlock) a += part[(size_t)b * ncols + k];
    const double tot = block_sum(a);
    if (threadIdx.x == 0) out[k] = tot;
  }
}
void launch_sum_cols(const double *part, int nb, int ncols, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_cols, dim3(1), dim3(kBlock), 0, st, part, nb, ncols, out);
}

// The five masked sums of the even/odd slice-scaling Gauss-Newton step
// (unires/_update.py:310-336): msk = x != 0, slices split along dim_thick as the reference's
// _even_odd does ('odd' = dat[::2], 'even' = dat[1::2], _update.py:430-445);
// out[0] = sum (x-y)^2, out[1] = sum_even y(x-y), out[2] = sum_odd y(x-y),
// out[3] = sum_even y^2, out[4] = sum_odd y^2  - float32 terms, float64 sums.
// Two stages, both in a fixed order (r1 - r3 added the workgroups' sums with float64 atomics: a result that
// depended on the order they finished in, and one 64-bit division per voxel for the slice parity): every workgroup
// walks the volume with a fixed stride keeping (x, y, z) of its voxel by carry arithmetic, writes its five sums to
// part[5 * blockIdx.x ..]; k_sum_cols adds the workgroups' sums in index order.
constexpr int kScalingBlocks = 1024;
__global__ void __launch_bounds__(kBlock)
    k_scaling_sums(const float *__restrict__ x, const float *__restrict__ y, Dim3i d, int dim_thick,
                   double *__restrict__ part) {
  const size_t n = d.numel(), stride = (size_t)gridDim.x * blockDim.x;
  // stride = sx * (dy dz) + sy * dz + sz; position of voxel i0 = blockIdx.x * blockDim.x + threadIdx.x
  const size_t plane = (size_t)d.y * d.z;
  const int sx = (int)(stride / plane), sy = (int)((stride % plane) / d.z), sz = (int)(stride % d.z);
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int px = (int)(i0 / plane), py = (int)((i0 % plane) / d.z), pz = (int)(i0 % d.z);
  double s0 = 0.0, ge = 0.0, go = 0.0, he = 0.0, ho = 0.0;
  for (size_t i = i0; i < n; i += stride) {
    const float xv = x[i];
    if (xv != 0.f) {
      const float yv = y[i], r = __fsub_rn(xv, yv);
      const bool even = (dim_thick == 2 ? pz : (dim_thick == 1 ? py : px)) & 1;  // index 1::2 along dim_thick
      s0 += (double)__fmul_rn(r, r);
      const double g = (double)__fmul_rn(yv, r), h = (double)__fmul_rn(yv, yv);
      if (even)
        ge += g, he += h;
      else
        go += g, ho += h;
    }
    pz += sz;
    if (pz >= d.z) pz -= d.z, ++py;
    py += sy;
    if (py >= d.y) py -= d.y, ++px;  // (py + carry + sy <= 2 dy - 1: once)
    px += sx;
  }
  const double v[5] = {block_sum(s0), block_sum(ge), block_sum(go), block_sum(he), block_sum(ho)};
  if (threadIdx.x == 0)
    for (int k = 0; k < 5; ++k) part[5 * blockIdx.x + k] = v[k];
}

int scaling_sums_blocks(Dim3i d) {
  size_t b = (d.numel() + kBlock - 1) / kBlock;
  return (int)(b > (size_t)kScalingBlocks ? (size_t)kScalingBlocks : b);
}

// part: 5 * scaling_sums_blocks(d) doubles of scratch
void launch_scaling_sums(const float *x, const float *y, Dim3i d, int dim_thick, double *part, double *out,
                         hipStream_t st) {
  const int nb = scaling_sums_blocks(d);
  hipLaunchKernelGGL(k_scaling_sums, dim3(nb), dim3(kBlock), 0, st, x, y, d, dim_thick, part);
  launch_sum_cols(part, nb, 5, out, st);
}

// Gauss-Newton sums of the rigid update (unires/_update.py:622-650).  With s_i(v) =
// sum_d gr_d(v) * dAff_i,d(v)  and  dAff_i,d(v) = D_i[d,0] i + D_i[d,1] j + D_i[d,2] k + D_i[d,3]:
//   out[i]            = sum_v diff(v) s_i(v)                       (gradient, i < 6)
//   out[6 + tri(i,j)] = sum_v ctc(v) s_i(v) s_j(v),  i <= j         (Hessian: Hes_m is the
//                       rank-1 outer product gr gr^T times CtC, so the reference's 9 x 21
//                       triple loop collapses to 21 products per voxel)
// gr3: (dim,3) spatial gradients of the pulled image; diff: (dim) residual (conv_transposed in
// the super-resolution case); ctc: (dim) or nullptr (= 1).  float64 accumulation.
struct RigidD {
  float d[6][12];
};
__global__ void __launch_bounds__(kBlock)
    k_rigid_sums(const float *__restrict__ gr3, const float *__restrict__ diff,
                 const float *__restrict__ ctc, Dim3i dm, RigidD D, double *__restrict__ out) {
  const size_t n = dm.numel(), stride = (size_t)gridDim.x * blockDim.x;
  double acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = 0.0;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
    const int k = (int)(v % dm.z), j = (int)((v / dm.z) % dm.y), i = (int)(v / ((size_t)dm.z * dm.y));
    const float g0 = gr3[3 * v], g1 = gr3[3 * v + 1], g2 = gr3[3 * v + 2];
    const double df = diff[v], c = ctc ? (double)ctc[v] : 1.0;
    double s[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float *m = D.d[q];
      const double a0 = (double)m[0] * i + (double)m[1] * j + (double)m[2] * k + (double)m[3];
      const double a1 = (double)m[4] * i + (double)m[5] * j + (double)m[6] * k + (double)m[7];
      const double a2 = (double)m[8] * i + (double)m[9] * j + (double)m[10] * k + (double)m[11];
      s[q] = g0 * a0 + g1 * a1 + g2 * a2;
      acc[q] += df * s[q];
    }
    int t = 6;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) acc[t++] += c * s[a] * s[b];
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const double tot = block_sum(acc[t]);
    if (threadIdx.x == 0) out[27 * blockIdx.x + t] = tot;  // (per-workgroup sums: k_sum_cols adds them in order)
  }
}

int rigid_sums_blocks(Dim3i dm) {
  size_t b = (dm.numel() + kBlock - 1) / kBlock;
  return (int)(b > 512 ? 512 : b);
}
// part: 27 * rigid_sums_blocks(dm) doubles of scratch
void launch_rigid_sums(const float *gr3, const float *diff, const float *ctc, Dim3i dm,
                       const float D[6][12], double *part, double *out, hipStream_t st) {
  RigidD R;
  memcpy(R.d, D, sizeof(R.d));
  const int b = rigid_sums_blocks(dm);
  hipLaunchKernelGGL(k_rigid_sums, dim3(b), dim3(kBlock), 0, st, gr3, diff, ctc, dm, R, part);
  launch_sum_cols(part, b, 27, out, st);
}

// y[v] = 0 where M v falls outside [0, dim_x) on any axis  (fit()'s clean_fov, run.py:150-164:
// grid = M (i,j,k), keep = 0 <= g_d < dim_x_d)
__global__ void __launch_bounds__(kBlock)
    k_clean_fov(float *__restrict__ y, Dim3i d, Affine M, float nx, float ny, float nz) {
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * (kBlock / kWave) + threadIdx.y,
            i = blockIdx.z;
  if (k >= d.z || j >= d.y) return;
  float gx, gy, gz;
  affine_point(M, (float)i, (float)j, (float)k, gx, gy, gz);
  const bool keep = gx >= 0.f && gx < nx && gy >= 0.f && gy < ny && gz >= 0.f && gz < nz;
  if (!keep) y[((size_t)i * d.y + j) * d.z + k] = 0.f;
}

static inline dim3 vblock() { return dim3(kWave, kBlock / kWave, 1); }

void launch_clean_fov(float *y, Dim3i d, const Affine &M, Dim3i dx, hipStream_t st) {
  const dim3 grid((d.z + kWave - 1) / kWave, (d.y + 3) / 4, d.x);
  hipLaunchKernelGGL(k_clean_fov, grid, vblock(), 0, st, y, d, M, (float)dx.x, (float)dx.y,
                     (float)dx.z);
}
// workgroups of a k_jtv_scale launch: (z, y) patches x as many x slots as keep the launch at ~4096 workgroups
static dim3 jtv_grid(Dim3i d) {
  const int tz = (d.z + kWave - 1) / kWave, ty = (d.y + 3) / 4;
  int gx = (int)std::min<long long>(d.x, std::max<long long>(1, 4096 / ((long long)tz * ty)));
  if ((long long)tz * ty * gx > 65535ll * 16) gx = 1;
  return dim3(tz, ty, gx);
}
int jtv_scale_blocks(Dim3i d) {
  const dim3 g = jtv_grid(d);
  return (int)(g.x * g.y * g.z);
}

int launch_jtv_scale(const float *const *y, const float *lam, int nc, const float *w,
                     const float *z_old, Dim3i d, const float vx[3], float rho, float alpha,
                     float *scale, double *part, double *out, int norm_only, hipStream_t st) {
  // The joint-TV magnitude couples all channels of a voxel (unires/_update.py:166-173 loops over any
  // C); the kernel takes 8 channel pointers by value, so more channels run as a chain of launches
  // that carry the running sum of squares in `scale`.
  if (nc > 8 && !scale) return -1;
  // part (jtv_scale_blocks(d) doubles of scratch) + out: the sum of the norms, added in workgroup order
  const dim3 grid = jtv_grid(d);
  const int g = (int)(grid.x * grid.y * grid.z);
  double *partials = out ? part : nullptr;
  for (int c0 = 0; c0 < nc; c0 += 8) {
    ChanPtrs C;
    C.n = nc - c0 < 8 ? nc - c0 : 8;
    C.c0 = c0, C.first = c0 == 0, C.last = c0 + 8 >= nc;
    for (int c = 0; c < 8; ++c) C.y[c] = y[c0 + (c < C.n ? c : 0)], C.lam[c] = lam[c0 + (c < C.n ? c : 0)];
#define JTV_LAUNCH(NCV)                                                                               \
  hipLaunchKernelGGL(k_jtv_scale<NCV>, grid, vblock(), 0, st, C, w, z_old, d, 1.f / vx[0], 1.f / vx[1], \
                     1.f / vx[2], rho, alpha, scale, partials, norm_only)
    switch (C.n) {
      case 1: JTV_LAUNCH(1); break;
      case 2: JTV_LAUNCH(2); break;
      case 3: JTV_LAUNCH(3); break;
      case 4: JTV_LAUNCH(4); break;
      default: JTV_LAUNCH(0); break;
    }
#undef JTV_LAUNCH
  }
  if (out) launch_sum_cols(part, g, 1, out, st);
  return g;
}

void launch_zw_update(const float *y, float lam, const float *scale, float *z, float *w, Dim3i d,
                      const float vx[3], float rho, float alpha, hipStream_t st) {
  const dim3 grid((d.z + kWave - 1) / kWave, (d.y + 3) / 4, d.x);
  hipLaunchKernelGGL(k_zw_update, grid, vblock(), 0, st, y, lam, scale, z, w, d, 1.f / vx[0],
                     1.f / vx[1], 1.f / vx[2], rho, alpha);
}

int masked_sse_blocks(size_t n) {
  size_t b = (n + kBlock - 1) / kBlock;
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}
// part: masked_sse_blocks(n) doubles of scratch
int launch_masked_sse(const float *x, const float *ay, size_t n, double *part, double *out, hipStream_t st) {
  const int b = masked_sse_blocks(n);
  hipLaunchKernelGGL(k_masked_sse, dim3(b), dim3(kBlock), 0, st, x, ay, n, part);
  launch_sum_cols(part, b, 1, out, st);
  return b;
}

}  // namespace unires
