// ops.hip - op-level kernels: affine trilinear pull / push, strided separable
// conv down / up, forward-difference gradient / divergence / DtD.
//
// Layout: float32 volumes, (X,Y,Z) C-contiguous, Z fastest.  Every kernel puts
// the 64 lanes of a wave along Z so that HBM/L2 requests are coalesced.
// Launch shape: block (64,4,1) -> grid (ceil(Z/64), ceil(Y/4), X).
#include "common.hpp"
#include <stdint.h>

#include <algorithm>

#include "ops.hpp"

namespace unires {

static inline dim3 vol_block() { return dim3(kWave, kBlock / kWave, 1); }
static inline dim3 vol_grid(const Dim3i &d) {
  return dim3((d.z + kWave - 1) / kWave, (d.y + 3) / 4, d.x);
}

// --------------------------------------------------------------------------
// pull: dst[g] = mask(g) * sum_8 w_c * src[corner_c(M g)]
// (nitorch grid_pull linear / zero / extrapolate=False)
// --------------------------------------------------------------------------
constexpr int kPullChunksMax = 4;  // z chunks of 64 per thread: up to 16 eight-byte loads in flight

// block = 4 waves = 4 consecutive grid rows j (same i); each lane takes kPullChunks grid-z
// positions 64 apart.  Blocks whose whole footprint is inside the volume (all but a thin
// shell) take the interior path.
// kPullChunks = chunks per thread, chosen so that the last one is not mostly idle lanes
// (a 181-long row takes 3 chunks, not 4)
template <int kPullChunks>
__global__ void __launch_bounds__(kBlock) k_pull(const float *__restrict__ src, Dim3i sd, Affine A,
                                                 float *__restrict__ dst, Dim3i gd, float tol,
                                                 const int *__restrict__ done) {
  if (done && *done) return;
  const int lane = threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z;
  const int kbase = blockIdx.x * (kWave * kPullChunks);
  if (j >= gd.y) return;
  const RowBase rb = affine_row(A, (float)i, (float)j);
  float *row = dst + ((size_t)i * gd.y + j) * gd.z;
  const unsigned ny = sd.y, nz = sd.z, nynz = ny * nz;
  const float bx = (float)(sd.x - 1), by = (float)(sd.y - 1), bz = (float)(sd.z - 1);
  float g[kPullChunks][3];
  bool inside = sd.z >= 2 && fits_fast_index(sd);
#pragma unroll
  for (int u = 0; u < kPullChunks; ++u) {
    const int k = min(kbase + u * kWave + lane, gd.z - 1);
    affine_along(A, rb, (float)k, g[u][0], g[u][1], g[u][2]);
    inside = inside && g[u][0] >= 0.f && g[u][0] < bx && g[u][1] >= 0.f && g[u][1] < by &&
             g[u][2] >= 0.f && g[u][2] < bz;
  }
  if (__all(inside)) {  // every corner of every sample of this wave is inside the volume
#pragma unroll
    for (int u = 0; u < kPullChunks; ++u) {
      const int k = kbase + u * kWave + lane;
      const float v = pull_interior(src, ny, nz, nynz, g[u][0], g[u][1], g[u][2]);
      if (k < gd.z) row[k] = v;
    }
    return;
  }
  PullLoads L[kPullChunks];
#pragma unroll
  for (int u = 0; u < kPullChunks; ++u) pull_issue(src, sd, g[u][0], g[u][1], g[u][2], tol, L[u]);
#pragma unroll
  for (int u = 0; u < kPullChunks; ++u) {
    const int k = kbase + u * kWave + lane;
    if (k < gd.z) row[k] = pull_finish(L[u]);
  }
}

__global__ void __launch_bounds__(kBlock)
    k_conv_up(const float *__restrict__ xs, Dim3i xd, Taps T, Scaling S, float *__restrict__ dst,
              Dim3i gd) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z;
  if (k >= gd.z || j >= gd.y) return;
  dst[((size_t)i * gd.y + j) * gd.z + k] = conv_up_sample(xs, xd, T, S, i, j, k);
}

// --------------------------------------------------------------------------
// conv_down: dst[i,j,k] = S(i,j,k) * sum_abc kx[a]ky[b]kz[c] src[rx i+a, ry j+b, rz k+c]
// (F.conv3d, cross-correlation, no padding) + _apply_scaling epilogue
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_conv_down(const float *__restrict__ src, Dim3i gd, Taps T, Scaling S,
                float *__restrict__ dst, Dim3i xd, const int *__restrict__ done) {
  if (done && *done) return;
  const int k = blockIdx.x * kWave + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z;
  if (k >= xd.z || j >= xd.y) return;
  float acc = 0.f;
  for (int a = 0; a < T.n[0]; ++a) {
    for (int b = 0; b < T.n[1]; ++b) {
      const float wab = T.t[0][a] * T.t[1][b];
      const float *row = src + ((size_t)(T.s[0] * i + a) * gd.y + (T.s[1] * j + b)) * gd.z +
                         (size_t)T.s[2] * k;
      for (int c = 0; c < T.n[2]; ++c) acc += row[c] * (wab * T.t[2][c]);
    }
  }
  if (S.dim >= 0) {
    const int par = (S.dim == 0 ? i : (S.dim == 1 ? j : k)) & 1;
    acc *= par ? S.o : S.e;
  }
  dst[((size_t)i * xd.y + j) * xd.z + k] = acc;
}

// --------------------------------------------------------------------------
// forward differences, zero bound
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    k_grad(const float *__restrict__ src, Dim3i d, float ivx, float ivy, float ivz,
           float *__restrict__ dst) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z;
  if (k >= d.z || j >= d.y) return;
  const size_t n = d.numel();
  const size_t idx = ((size_t)i * d.y + j) * d.z + k;
  const float c = src[idx];
  const float xn = i + 1 < d.x ? src[idx + (size_t)d.y * d.z] : 0.f;
  const float yn = j + 1 < d.y ? src[idx + d.z] : 0.f;
  const float zn = k + 1 < d.z ? src[idx + 1] : 0.f;
  dst[idx] = (xn - c) * ivx;
  dst[n + idx] = (yn - c) * ivy;
  dst[2 * n + idx] = (zn - c) * ivz;
}

// dst = [add +] scale * Dt(u), u = a*src_a (+ b*src_b if src_b != NULL)
__global__ void __launch_bounds__(kBlock)
    k_div(const float *__restrict__ ua, const float *__restrict__ ub, float ca, float cb, Dim3i d,
          float ivx, float ivy, float ivz, float scale, const float *__restrict__ add,
          float *__restrict__ dst) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int i = blockIdx.z;
  if (k >= d.z || j >= d.y) return;
  const size_t n = d.numel();
  const size_t idx = ((size_t)i * d.y + j) * d.z + k;
  const size_t sx = (size_t)d.y * d.z, sy = d.z;
  // unconditional (clamped) loads so all of them are in flight together
  const bool lx = i > 0, ly = j > 0, lz = k > 0;
  const size_t ox = idx, oy = n + idx, oz = 2 * n + idx;
  const size_t oxm = lx ? ox - sx : ox, oym = ly ? oy - sy : oy, ozm = lz ? oz - 1 : oz;
  float vx = ca * ua[ox], vxm = ca * ua[oxm], vy = ca * ua[oy], vym = ca * ua[oym];
  float vz = ca * ua[oz], vzm = ca * ua[ozm];
  if (ub) {
    vx += cb * ub[ox], vxm += cb * ub[oxm], vy += cb * ub[oy], vym += cb * ub[oym];
    vz += cb * ub[oz], vzm += cb * ub[ozm];
  }
  float acc = ((lx ? vxm : 0.f) - vx) * ivx;
  acc += ((ly ? vym : 0.f) - vy) * ivy;
  acc += ((lz ? vzm : 0.f) - vz) * ivz;
  dst[idx] = (add ? add[idx] : 0.f) + scale * acc;
}

// dst = a*src + c*DtD(src): 7-point stencil with Neumann row at 0 and Dirichlet
// row at n-1 along every axis.  Optional fused float64 partial of sum(src*dst).
template <bool DOT>
__global__ void __launch_bounds__(kBlock)
    k_dtd(const float *__restrict__ src, Dim3i d, float cx, float cy, float cz, float a,
          float *__restrict__ dst, double *__restrict__ partials,
          const float *__restrict__ objb, const int *__restrict__ done) {
  if (done && *done) return;
  // tiles of 4 y-rows x 64 z; a bounded grid (<= kMaxPartials blocks) strides over them
  const int tz = (d.z + kWave - 1) / kWave, ty = (d.y + 3) / 4;
  const long long ntiles = (long long)tz * ty * d.x;
  double prod = 0.0;
  for (long long t = xcd_chunked_block(blockIdx.x, gridDim.x); t < ntiles; t += gridDim.x) {
    const int kc = (int)(t % tz);
    const long long t2 = t / tz;
    const int jq = (int)(t2 % ty);
    const int i = (int)(t2 / ty);
    const int k = kc * kWave + threadIdx.x;
    const int j = jq * 4 + threadIdx.y;
    if (k < d.z && j < d.y) {
      const size_t idx = ((size_t)i * d.y + j) * d.z + k;
      float c;
      const float st = dtd_at(src, idx, i, j, k, d, cx, cy, cz, c);
      const float q = a * c + st;
      matvec_emit(dst, idx, q, c, DOT ? objb : nullptr, DOT, prod);
    }
  }
  if (DOT) {
    const double tot = block_sum(prod);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[blockIdx.x] = tot;
  }
}

// --------------------------------------------------------------------------
// host launchers
// --------------------------------------------------------------------------
void launch_pull(const float *src, Dim3i sd, const Affine &A, float *dst, Dim3i gd, float tol,
                 const int *done, hipStream_t st) {
  // fewest idle lanes: chunks per thread = the count (<= 4) that wastes least of the row's tail
  int best = kPullChunksMax, waste = 1 << 30;
  for (int c = kPullChunksMax; c >= 1; --c) {
    const int span = kWave * c, w = (gd.z + span - 1) / span * span - gd.z;
    if (w < waste) waste = w, best = c;
  }
  const int zspan = kWave * best;
  const dim3 grid((gd.z + zspan - 1) / zspan, (gd.y + 3) / 4, gd.x);
  if (best == 4)
    hipLaunchKernelGGL(k_pull<4>, grid, vol_block(), 0, st, src, sd, A, dst, gd, tol, done);
  else if (best == 3)
    hipLaunchKernelGGL(k_pull<3>, grid, vol_block(), 0, st, src, sd, A, dst, gd, tol, done);
  else if (best == 2)
    hipLaunchKernelGGL(k_pull<2>, grid, vol_block(), 0, st, src, sd, A, dst, gd, tol, done);
  else
    hipLaunchKernelGGL(k_pull<1>, grid, vol_block(), 0, st, src, sd, A, dst, gd, tol, done);
}

// --------------------------------------------------------------------------
// Separable form of the slice-profile convolutions: one 1-D pass per non-dirac axis.
// The fused kernels apply the 3-D kernel directly (fine for a thick-slice profile: 7 x 1 x 1
// taps); a Gaussian in-plane profile on top (reference default, struct.py:95: 5 x 11 x 11 taps
// for ratio 2) costs 605 taps per output directly and 27 as three passes.
// --------------------------------------------------------------------------
struct Taps1 {
  float t[UNIRES_MAX_TAPS];
};
// dst[.., o, ..] = S(o) sum_t ker[t] src[.., s o + t, ..]   along `axis`
__global__ void __launch_bounds__(kBlock)
    k_conv1d_down(const float *__restrict__ src, Dim3i sd, int axis, Taps1 K, int n, int s,
                  float se, float so, float *__restrict__ dst, Dim3i dd, const int *__restrict__ done) {
  if (done && *done) return;
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (k >= dd.z || j >= dd.y) return;
  const size_t sstr = axis == 0 ? (size_t)sd.y * sd.z : (axis == 1 ? (size_t)sd.z : 1);
  for (int i = blockIdx.z; i < dd.x; i += gridDim.z) {  // several x slabs per workgroup
    const int o = axis == 0 ? i : (axis == 1 ? j : k);
    const size_t base = ((size_t)(axis == 0 ? s * i : i) * sd.y + (axis == 1 ? s * j : j)) * sd.z +
                        (axis == 2 ? s * k : k);
    // loads batched four deep (the sum keeps its order): a one-load-per-iteration loop is a
    // chain of full memory round trips
    float acc = 0.f;
    int t = 0;
    for (; t + 4 <= n; t += 4) {
      const float v0 = src[base + (size_t)t * sstr], v1 = src[base + (size_t)(t + 1) * sstr],
                  v2 = src[base + (size_t)(t + 2) * sstr], v3 = src[base + (size_t)(t + 3) * sstr];
      acc = fmaf(K.t[t], v0, acc), acc = fmaf(K.t[t + 1], v1, acc);
      acc = fmaf(K.t[t + 2], v2, acc), acc = fmaf(K.t[t + 3], v3, acc);
    }
    for (; t < n; ++t) acc = fmaf(K.t[t], src[base + (size_t)t * sstr], acc);
    dst[((size_t)i * dd.y + j) * dd.z + k] = acc * ((o & 1) ? so : se);
  }
}
// dst[.., u, ..] = sum_k ker[u - s k] S(k) src[.., k, ..]   along `axis` (transposed conv)
__global__ void __launch_bounds__(kBlock)
    k_conv1d_up(const float *__restrict__ src, Dim3i sd, int axis, Taps1 K, int n, int s, float se,
                float so, float *__restrict__ dst, Dim3i dd) {
  __shared__ float taps[UNIRES_MAX_TAPS];
  const int tid = threadIdx.y * kWave + threadIdx.x;
  if (tid < UNIRES_MAX_TAPS) taps[tid] = K.t[tid];
  __syncthreads();
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (k >= dd.z || j >= dd.y) return;
  const size_t sstr = axis == 0 ? (size_t)sd.y * sd.z : (axis == 1 ? (size_t)sd.z : 1);
  const int nsrc = axis == 0 ? sd.x : (axis == 1 ? sd.y : sd.z);
  const float inv_s = 1.f / (float)s;
  for (int i = blockIdx.z; i < dd.x; i += gridDim.z) {  // several x slabs per workgroup
    const int u = axis == 0 ? i : (axis == 1 ? j : k);
    int lo, hi;
    up_range_f(u, n, s, inv_s, nsrc, lo, hi);
    const size_t base = ((size_t)(axis == 0 ? 0 : i) * sd.y + (axis == 1 ? 0 : j)) * sd.z + (axis == 2 ? 0 : k);
    float acc = 0.f;
    int c = lo;
    for (; c + 2 <= hi + 1; c += 2) {  // two loads in flight
      const float v0 = src[base + (size_t)c * sstr], v1 = src[base + (size_t)(c + 1) * sstr];
      acc = fmaf(taps[u - s * c] * ((c & 1) ? so : se), v0, acc);
      acc = fmaf(taps[u - s * (c + 1)] * (((c + 1) & 1) ? so : se), v1, acc);
    }
    for (; c <= hi; ++c) acc = fmaf(taps[u - s * c] * ((c & 1) ? so : se), src[base + (size_t)c * sstr], acc);
    dst[((size_t)i * dd.y + j) * dd.z + k] = acc;
  }
}

// The same two passes along x or y with four z per lane (16-byte loads and stores; z length a
// multiple of 4, 16-byte aligned volumes): the one-float forms above move 3 TB/s, a quarter of the
// instructions per byte gets them to the streaming kernels' rate.  Same order of operations per
// output, so the results are bit-identical.
__device__ __forceinline__ float4 fma4(float w, float4 v, float4 a) {
  return make_float4(fmaf(w, v.x, a.x), fmaf(w, v.y, a.y), fmaf(w, v.z, a.z), fmaf(w, v.w, a.w));
}
__global__ void __launch_bounds__(kBlock)
    k_conv1d_down_v4(const float4 *__restrict__ src, Dim3i sd, int axis, Taps1 K, int n, int s, float se,
                     float so, float4 *__restrict__ dst, Dim3i dd, const int *__restrict__ done) {
  if (done && *done) return;
  const int z4 = dd.z >> 2;
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (k >= z4 || j >= dd.y) return;
  const size_t sstr = axis == 0 ? (size_t)sd.y * z4 : (size_t)z4;
  for (int i = blockIdx.z; i < dd.x; i += gridDim.z) {
    const int o = axis == 0 ? i : j;
    const size_t base = ((size_t)(axis == 0 ? s * i : i) * sd.y + (axis == 1 ? s * j : j)) * z4 + k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int t = 0;
    for (; t + 4 <= n; t += 4) {
      const float4 v0 = src[base + (size_t)t * sstr], v1 = src[base + (size_t)(t + 1) * sstr],
                   v2 = src[base + (size_t)(t + 2) * sstr], v3 = src[base + (size_t)(t + 3) * sstr];
      acc = fma4(K.t[t], v0, acc), acc = fma4(K.t[t + 1], v1, acc);
      acc = fma4(K.t[t + 2], v2, acc), acc = fma4(K.t[t + 3], v3, acc);
    }
    for (; t < n; ++t) acc = fma4(K.t[t], src[base + (size_t)t * sstr], acc);
    const float sc = (o & 1) ? so : se;
    dst[((size_t)i * dd.y + j) * z4 + k] = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
  }
}
__global__ void __launch_bounds__(kBlock)
    k_conv1d_up_v4(const float4 *__restrict__ src, Dim3i sd, int axis, Taps1 K, int n, int s, float se,
                   float so, float4 *__restrict__ dst, Dim3i dd) {
  __shared__ float taps[UNIRES_MAX_TAPS];
  const int tid = threadIdx.y * kWave + threadIdx.x;
  if (tid < UNIRES_MAX_TAPS) taps[tid] = K.t[tid];
  __syncthreads();
  const int z4 = dd.z >> 2;
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (k >= z4 || j >= dd.y) return;
  const size_t sstr = axis == 0 ? (size_t)sd.y * z4 : (size_t)z4;
  const int nsrc = axis == 0 ? sd.x : sd.y;
  const float inv_s = 1.f / (float)s;
  for (int i = blockIdx.z; i < dd.x; i += gridDim.z) {
    const int u = axis == 0 ? i : j;
    int lo, hi;
    up_range_f(u, n, s, inv_s, nsrc, lo, hi);
    const size_t base = ((size_t)(axis == 0 ? 0 : i) * sd.y + (axis == 1 ? 0 : j)) * z4 + k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = lo;
    for (; c + 2 <= hi + 1; c += 2) {
      const float4 v0 = src[base + (size_t)c * sstr], v1 = src[base + (size_t)(c + 1) * sstr];
      acc = fma4(taps[u - s * c] * ((c & 1) ? so : se), v0, acc);
      acc = fma4(taps[u - s * (c + 1)] * (((c + 1) & 1) ? so : se), v1, acc);
    }
    for (; c <= hi; ++c) acc = fma4(taps[u - s * c] * ((c & 1) ? so : se), src[base + (size_t)c * sstr], acc);
    dst[((size_t)i * dd.y + j) * z4 + k] = acc;
  }
}
// The x and y passes of conv_down / conv_up as ONE kernel (r3): the separable passes of an isotropic
// down-sampling (config 4: 3 taps, stride 2 along x and y) write and re-read a scratch volume between
// them - 113 + 57 + 57 + 28 MB for what needs 113 + 28.  Here a lane forms the intermediate values of
// its output in registers.  Same products, same order (y chain, scaling, then x chain for conv_down;
// x chain then y chain for conv_up), so the results are bit-identical to the two passes.
struct Taps2 {
  float x[8], y[8];
};
constexpr int kConvXYMax = 8;  // taps (conv_down) / fan-in (conv_up) per axis the fused forms take
__global__ void __launch_bounds__(kBlock)
    k_conv2d_down_xy_v4(const float4 *__restrict__ src, Dim3i sd, Taps2 K, int nx, int sx, int ny, int sy,
                        float sex, float sox, float sey, float soy, float4 *__restrict__ dst, Dim3i dd,
                        const int *__restrict__ done) {
  if (done && *done) return;
  // (threads run over the flattened (y, z / 4) plane of an x slab: rows of 192 voxels are 48 float4 -
  // a 64-lane row per z line left a quarter of the lanes idle)
  const int z4 = dd.z >> 2;
  const unsigned t = blockIdx.x * (unsigned)kBlock + threadIdx.y * kWave + threadIdx.x;
  if (t >= (unsigned)dd.y * (unsigned)z4) return;
  const int j = (int)(t / (unsigned)z4), k = (int)(t - (unsigned)j * (unsigned)z4);
  const size_t sstr_x = (size_t)sd.y * z4, sstr_y = (size_t)z4;
  const float scy = (j & 1) ? soy : sey;
  for (int i = blockIdx.y; i < dd.x; i += gridDim.y) {
    const float4 *base = src + ((size_t)(sx * i) * sd.y + (size_t)sy * j) * z4 + k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ta = 0; ta < nx; ++ta) {
      const float4 *row = base + (size_t)ta * sstr_x;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      int tb = 0;
      for (; tb + 4 <= ny; tb += 4) {
        const float4 v0 = row[(size_t)tb * sstr_y], v1 = row[(size_t)(tb + 1) * sstr_y],
                     v2 = row[(size_t)(tb + 2) * sstr_y], v3 = row[(size_t)(tb + 3) * sstr_y];
        t = fma4(K.y[tb], v0, t), t = fma4(K.y[tb + 1], v1, t);
        t = fma4(K.y[tb + 2], v2, t), t = fma4(K.y[tb + 3], v3, t);
      }
      for (; tb < ny; ++tb) t = fma4(K.y[tb], row[(size_t)tb * sstr_y], t);
      t = make_float4(t.x * scy, t.y * scy, t.z * scy, t.w * scy);
      acc = fma4(K.x[ta], t, acc);
    }
    const float sc = (i & 1) ? sox : sex;
    dst[((size_t)i * dd.y + j) * z4 + k] = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
  }
}
// conv_up: fan-ins <= FX x FY known at compile time, R consecutive x slabs per thread: every load of
// the R outputs is issued before the first product (a run-time form with one output per thread had one
// short dependent chain per thread - 27 648 workgroups that live 3.4 us each at 384 x 384 x 192: 46 us,
// 3 TB/s; this one 29 us; with fan-ins 2 x 6 it lost to the two passes, 78 vs 65 us, and was dropped).
// Taps beyond a voxel's range are skipped, not multiplied by zero: bit-identical to the two passes.
template <int FX, int FY, int R>
__global__ void __launch_bounds__(kBlock)
    k_conv2d_up_xy_v4_t(const float4 *__restrict__ src, Dim3i sd, Taps1 KX, Taps1 KY, int nx, int sx, int ny,
                        int sy, float sex, float sox, float sey, float soy, float4 *__restrict__ dst, Dim3i dd) {
  __shared__ float tx[UNIRES_MAX_TAPS], ty[UNIRES_MAX_TAPS];
  const int tid = threadIdx.y * kWave + threadIdx.x;
  if (tid < UNIRES_MAX_TAPS) tx[tid] = KX.t[tid], ty[tid] = KY.t[tid];
  __syncthreads();
  const int z4 = dd.z >> 2;
  const unsigned t = blockIdx.x * (unsigned)kBlock + (unsigned)tid;
  if (t >= (unsigned)dd.y * (unsigned)z4) return;
  const int j = (int)(t / (unsigned)z4), k = (int)(t - (unsigned)j * (unsigned)z4);
  int lo_y, hi_y;
  up_range_f(j, ny, sy, 1.f / (float)sy, sd.y, lo_y, hi_y);
  float wy[FY];
  unsigned rowy[FY];  // float4 offset of source row lo_y + cy (clamped into the range) at this lane's k
#pragma unroll
  for (int c = 0; c < FY; ++c) {
    const int cy = min(lo_y + c, hi_y);
    wy[c] = ty[j - sy * cy] * ((cy & 1) ? soy : sey);
    rowy[c] = (unsigned)cy * (unsigned)z4 + (unsigned)k;
  }
  const int i0 = (int)blockIdx.y * R;
  const float inv_sx = 1.f / (float)sx;
  const unsigned sstr_x = (unsigned)sd.y * (unsigned)z4;  // (source volumes < 2^32 float4: checked by the launcher)
  float4 v[R][FY][FX];
  float wx[R][FX];
  int nxr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = min(i0 + r, dd.x - 1);
    int lo, hi;
    up_range_f(i, nx, sx, inv_sx, sd.x, lo, hi);
    nxr[r] = hi - lo + 1;
#pragma unroll
    for (int cx = 0; cx < FX; ++cx) {
      const int c = min(lo + cx, hi);
      wx[r][cx] = tx[i - sx * c] * ((c & 1) ? sox : sex);
#pragma unroll
      for (int cy = 0; cy < FY; ++cy) v[r][cy][cx] = src[(size_t)((unsigned)c * sstr_x + rowy[cy])];
    }
  }
  const int nyr = hi_y - lo_y + 1;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (i0 + r >= dd.x) break;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int cy = 0; cy < FY; ++cy) {
      if (cy >= nyr) break;
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int cx = 0; cx < FX; ++cx)
        if (cx < nxr[r]) u = fma4(wx[r][cx], v[r][cy][cx], u);
      acc = fma4(wy[cy], u, acc);
    }
    dst[((size_t)(i0 + r) * dd.y + j) * z4 + k] = acc;
  }
}
static inline bool conv1d_v4_ok(const void *a, const void *b, const Dim3i &sd, const Dim3i &dd) {
  return (sd.z & 3) == 0 && sd.z == dd.z && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

// z-axis forms: a wave stages the contiguous piece of the input row it needs in LDS with
// coalesced loads (the generic kernels above issue one strided global load per tap and lane).
constexpr int kConvZStage = 64 * 8 + UNIRES_MAX_TAPS;  // stride <= 8
__global__ void __launch_bounds__(kBlock)
    k_conv1d_down_z(const float *__restrict__ src, Dim3i sd, Taps1 K, int n, int s, float se, float so,
                    float *__restrict__ dst, Dim3i dd, const int *__restrict__ done) {
  if (done && *done) return;
  __shared__ float stage[kBlock / kWave][kConvZStage];
  const int lane = threadIdx.x, w = threadIdx.y;
  const int k0 = blockIdx.x * kWave, j = blockIdx.y * 4 + w;
  if (j >= dd.y) return;
  const int z0 = s * k0, need = min(s * kWave + n - s, sd.z - z0);
  const int k = k0 + lane;
  for (int i = blockIdx.z; i < dd.x; i += gridDim.z) {  // several x slabs per workgroup
    const float *row = src + ((size_t)i * sd.y + j) * sd.z;
    asm volatile("" ::: "memory");
    for (int t = lane; t < need; t += kWave) stage[w][t] = row[z0 + t];
    asm volatile("" ::: "memory");  // one wave, LDS ops in order
    if (k < dd.z) {
      float acc = 0.f;
      for (int t = 0; t < n; ++t) acc = fmaf(K.t[t], stage[w][s * lane + t], acc);
      dst[((size_t)i * dd.y + j) * dd.z + k] = acc * ((k & 1) ? so : se);
    }
  }
}
// (r3: kUpZRows x slabs per trip - their row loads are issued together.  With one row per trip a wave
// had ONE 256-byte load in flight: 186 us for the 384 x 384 x 192 -> 387 volume of config 4, 1.8 TB/s.)
#ifndef UNIRES_UPZ_ROWS
#define UNIRES_UPZ_ROWS 4
#endif
constexpr int kUpZRows = UNIRES_UPZ_ROWS;
__global__ void __launch_bounds__(kBlock)
    k_conv1d_up_z(const float *__restrict__ src, Dim3i sd, Taps1 K, int n, int s, float se, float so,
                  float *__restrict__ dst, Dim3i dd) {
  constexpr int SW = kWave + UNIRES_MAX_TAPS + 2;
  __shared__ float taps[UNIRES_MAX_TAPS];
  __shared__ float stage[kBlock / kWave][kUpZRows][SW];
  const int lane = threadIdx.x, w = threadIdx.y;
  const int tid = w * kWave + lane;
  if (tid < UNIRES_MAX_TAPS) taps[tid] = K.t[tid];
  __syncthreads();
  const int u0 = blockIdx.x * kWave, j = blockIdx.y * 4 + w;
  if (j >= dd.y) return;
  int c0, c1, dummy;
  up_range(u0, n, s, sd.z, c0, dummy);                           // first slice feeding this piece
  up_range(min(u0 + kWave - 1, dd.z - 1), n, s, sd.z, dummy, c1);  // last one
  const int u = u0 + lane;
  int lo = 0, hi = -1;
  if (u < dd.z) up_range_f(u, n, s, 1.f / (float)s, sd.z, lo, hi);
  float wgt[UNIRES_MAX_TAPS / 4];  // this lane's (<= 8) weights are the same for every row
  const int nw = min(hi - lo + 1, UNIRES_MAX_TAPS / 4);
  for (int c = 0; c < UNIRES_MAX_TAPS / 4; ++c)
    wgt[c] = c < nw ? taps[u - s * (lo + c)] * (((lo + c) & 1) ? so : se) : 0.f;
  const int span = c1 - c0 + 1;  // <= kWave + taps: at most two loads per lane and row
  const int G = (int)gridDim.z;
  for (int i = blockIdx.z; i < dd.x; i += kUpZRows * G) {  // several x slabs per workgroup
    float v0[kUpZRows], v1[kUpZRows];
#pragma unroll
    for (int r = 0; r < kUpZRows; ++r) {
      const int ir = min(i + r * G, dd.x - 1);
      const float *row = src + ((size_t)ir * sd.y + j) * sd.z + c0;
      v0[r] = lane < span ? row[lane] : 0.f;
      v1[r] = lane + kWave < span ? row[lane + kWave] : 0.f;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < kUpZRows; ++r) {
      stage[w][r][lane] = v0[r];
      if (lane + kWave < SW) stage[w][r][lane + kWave] = v1[r];
    }
    asm volatile("" ::: "memory");  // one wave, LDS ops in order
    if (u < dd.z) {
#pragma unroll
      for (int r = 0; r < kUpZRows; ++r) {
        const int ir = i + r * G;
        if (ir >= dd.x) break;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < UNIRES_MAX_TAPS / 4; ++c)
          if (c < nw) acc = fmaf(wgt[c], stage[w][r][lo - c0 + c], acc);
        for (int c = lo + UNIRES_MAX_TAPS / 4; c <= hi; ++c)  // fan-in beyond 8: generic tail
          acc = fmaf(taps[u - s * c] * ((c & 1) ? so : se), stage[w][r][c - c0], acc);
        dst[((size_t)ir * dd.y + j) * dd.z + u] = acc;
      }
    }
    asm volatile("" ::: "memory");
  }
}

// grid of the 1-D conv passes: x slabs are looped inside the kernel (<= 48 workgroups along x)
static inline dim3 conv1d_grid(const Dim3i &d) {
  return dim3((d.z + kWave - 1) / kWave, (d.y + 3) / 4, d.x < 48 ? d.x : 48);
}
// grid of the fused x-y passes: the (y, z / 4) plane flattened along x, x slabs along y
static inline dim3 conv2d_grid(const Dim3i &d) {
  const unsigned plane = (unsigned)d.y * (unsigned)(d.z / 4);
  return dim3((plane + kBlock - 1) / kBlock, d.x < 65535 ? d.x : 65535, 1);
}
static inline Dim3i with_axis(Dim3i d, int axis, int v) {
  if (axis == 0) d.x = v;
  if (axis == 1) d.y = v;
  if (axis == 2) d.z = v;
  return d;
}
static inline int axis_len(const Dim3i &d, int axis) { return axis == 0 ? d.x : (axis == 1 ? d.y : d.z); }
static inline bool axis_is_dirac(const Taps &T, int a) {
  return T.n[a] == 1 && T.s[a] == 1 && T.t[a][0] == 1.f;
}

// xs = S conv_down(g): passes z, y, x through two scratch volumes (a and b, each >= numel(gd));
// the last pass writes dst.  `g` may be `a`.
// x / y passes of a stride-2 separable conv with many taps (the Gaussian profile of BASELINE config 4: 11 taps,
// fan-in 6), MARCHING along the pass axis (round 4).  k_conv1d_down_v4 / k_conv1d_up_v4 gather: eleven (five to
// six) 16-byte loads per output, cache hits that still pass the L2 -> L1 path, with their address arithmetic.  Here
// a thread keeps a sliding window of the pass axis in registers and walks a run of outputs: two new inputs per
// output (down), one new input per TWO outputs (up).  sa / sm: strides (in float4) of the thread's fixed axis and
// of the marching axis.
struct March2 {
  int na, z4;                // threads: na x z4
  long long sa_s, sm_s;      // source strides
  long long sa_d, sm_d;      // destination strides
  int n_in, n_out, run;      // extents along the pass axis, outputs (down) / input steps (up) per run
  float se, so;              // even / odd slice factors along the pass axis (1, 1: none)
  float k[12];               // taps (down) - or ke[6], ko[6] (up)
};

template <int NT>
__global__ void __launch_bounds__(kBlock) k_conv1d_down2_m(const float4 *__restrict__ src, float4 *__restrict__ dst, March2 M,
                                                          const int *__restrict__ done) {
  if (done && *done) return;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.y * kWave + threadIdx.x;
  const int a = (int)(tid / M.z4), kz = (int)(tid - (long long)a * M.z4);
  if (a >= M.na) return;
  const int oa = blockIdx.y * M.run, ob = min(oa + M.run, M.n_out);
  if (oa >= ob) return;
  const float4 *p = src + (long long)a * M.sa_s + kz + (long long)(2 * oa) * M.sm_s;
  float4 *q = dst + (long long)a * M.sa_d + kz + (long long)oa * M.sm_d;
  float4 w[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) w[t] = p[(long long)t * M.sm_s];  // (2 (n_out - 1) + NT - 1 = n_in - 1: inside)
  p += (long long)NT * M.sm_s;
  for (int o = oa; o < ob; ++o) {
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
    if (o + 1 < ob) n0 = p[0], n1 = p[M.sm_s];  // the next output's two new inputs, in flight over this one
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc = fma4(M.k[t], w[t], acc);
    const float sc = (o & 1) ? M.so : M.se;
    *q = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
#pragma unroll
    for (int t = 0; t + 2 < NT; ++t) w[t] = w[t + 2];
    w[NT - 2] = n0, w[NT - 1] = n1;
    p += 2 * M.sm_s, q += M.sm_d;
  }
}

// up: out[2m] = sum_i ke[i] S(m - i) in[m - i], out[2m + 1] = sum_i ko[i] S(m - i) in[m - i]; window w[i] = in[m - i].
// Same products (tap x slice factor, then x input) accumulated in the same order (ascending source index) as
// k_conv1d_up_v4's gather: bit-identical results.
template <int F>
__global__ void __launch_bounds__(kBlock) k_conv1d_up2_m(const float4 *__restrict__ src, float4 *__restrict__ dst, March2 M) {
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.y * kWave + threadIdx.x;
  const int a = (int)(tid / M.z4), kz = (int)(tid - (long long)a * M.z4);
  if (a >= M.na) return;
  const int nm = (M.n_out + 1) / 2;
  const int ma = blockIdx.y * M.run, mb = min(ma + M.run, nm);  // (runs are even: ma is)
  if (ma >= mb) return;
  const float4 *p = src + (long long)a * M.sa_s + kz;
  float4 *q = dst + (long long)a * M.sa_d + kz + (long long)(2 * ma) * M.sm_d;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // (value-returning: `cond ? zero : p[i]` of two lvalues selects between ADDRESSES and keeps `zero` in scratch memory)
  auto in_at = [&](int c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c >= 0 && c < M.n_in) v = p[(long long)c * M.sm_s];
    return v;
  };
  // taps x slice factor of source m - i, for even and for odd m
  float te[2][F], to[2][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const float sc = ((par ^ i) & 1) ? M.so : M.se;
      te[par][i] = M.k[i] * sc, to[par][i] = M.k[6 + i] * sc;
    }
  float4 w[F];
#pragma unroll
  for (int i = 0; i < F; ++i) w[i] = in_at(ma - i);
  auto step = [&](int m, const float (&ke)[F], const float (&ko)[F]) {
    float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m + 1 < mb) nx = in_at(m + 1);  // in flight over this step
    float4 e = zero, o = zero;
#pragma unroll
    for (int i = F - 1; i >= 0; --i) e = fma4(ke[i], w[i], e), o = fma4(ko[i], w[i], o);
    q[0] = e;
    if (2 * m + 1 < M.n_out) q[M.sm_d] = o;
#pragma unroll
    for (int i = F - 1; i > 0; --i) w[i] = w[i - 1];
    w[0] = nx;
    q += 2 * M.sm_d;
  };
  for (int m = ma; m < mb; m += 2) {
    step(m, te[0], to[0]);
    if (m + 1 < mb) step(m + 1, te[1], to[1]);
  }
}

// down then up along the same axis in ONE marching pass (A^T A of a stride-2 axis: the x pair of BASELINE
// config 4 with the default Gaussian in-plane profile): mid[j] = S(j) sum_t k[t] in[2 j + t] lives in
// registers only - out[2m + par] = sum_i k_up[2 i + par] mid[m - i] - so the (n / 2)-long intermediate is
// neither written nor read back (2 x 28 MB of 172 MB at config 4).  The products and their order are those of
// k_conv1d_down2_m followed by k_conv1d_up2_m with unit slice factors: bit-identical results.
template <int NT, int F>
__global__ void __launch_bounds__(kBlock) k_conv1d_downup2_m(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                            March2 M, int n_mid, const int *__restrict__ done) {
  if (done && *done) return;
  const long long tid = (long long)blockIdx.x * kBlock + threadIdx.y * kWave + threadIdx.x;
  const int a = (int)(tid / M.z4), kz = (int)(tid - (long long)a * M.z4);
  if (a >= M.na) return;
  const int nm = (M.n_out + 1) / 2;
  const int ma = blockIdx.y * M.run, mb = min(ma + M.run, nm);
  if (ma >= mb) return;
  const float4 *p = src + (long long)a * M.sa_s + kz;
  float4 *q = dst + (long long)a * M.sa_d + kz + (long long)(2 * ma) * M.sm_d;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float ke[F], ko[F];
#pragma unroll
  for (int i = 0; i < F; ++i) ke[i] = 2 * i < NT ? M.k[2 * i] : 0.f, ko[i] = 2 * i + 1 < NT ? M.k[2 * i + 1] : 0.f;
  auto mid_of = [&](const float4 (&w)[NT], int j) {
    float4 acc = zero;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc = fma4(M.k[t], w[t], acc);
    const float sc = (j & 1) ? M.so : M.se;
    return make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
  };
  // md[i] = mid[m - i] after step m; w[t] = in[2 m + t] at its start.  F - 1 warm-up steps (no stores) fill md.
  float4 md[F], w[NT];
#pragma unroll
  for (int i = 0; i < F; ++i) md[i] = zero;
  const int m0 = max(ma - (F - 1), 0);
  const bool any = m0 < n_mid;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    w[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any) w[t] = p[(long long)(2 * m0 + t) * M.sm_s];
  }
  p += (long long)(2 * (m0 + 1) + NT - 2) * M.sm_s;  // the first of the two inputs that mid[m0 + 1] adds
  for (int m = m0; m < mb; ++m) {
    float4 n0 = zero, n1 = zero;
    if (m + 1 < n_mid && m + 1 < mb) n0 = p[0], n1 = p[M.sm_s];  // in flight over this step
    float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < n_mid) nx = mid_of(w, m);
#pragma unroll
    for (int i = F - 1; i > 0; --i) md[i] = md[i - 1];
    md[0] = nx;
    if (m >= ma) {
      float4 e = zero, o = zero;
#pragma unroll
      for (int i = F - 1; i >= 0; --i) e = fma4(ke[i], md[i], e), o = fma4(ko[i], md[i], o);
      q[0] = e;
      if (2 * m + 1 < M.n_out) q[M.sm_d] = o;
      q += 2 * M.sm_d;
    }
#pragma unroll
    for (int t = 0; t + 2 < NT; ++t) w[t] = w[t + 2];
    w[NT - 2] = n0, w[NT - 1] = n1;
    p += 2 * M.sm_s;
  }
}

static bool march2_ok(const Taps &T, int ax) {
  static const bool off = getenv("UNIRES_CONV_MARCH") && atoi(getenv("UNIRES_CONV_MARCH")) == 0;
  return !off && ax != 2 && T.s[ax] == 2 && (T.n[ax] == 11 || T.n[ax] == 9 || T.n[ax] == 5 || T.n[ax] == 3);  // (Gaussian as built / as trimmed by the plan, triangle, trimmed rect at ratio 2)
}

// geometry shared by the two marching passes along `ax` (0 or 1) between volumes sd -> dd (float4 along z)
static March2 march2_args(Dim3i sd, Dim3i dd, int ax, int n_in, int n_out, int steps, float se, float so) {
  March2 M;
  M.z4 = dd.z / 4;
  M.na = ax == 0 ? dd.y : dd.x;  // (the fixed axis has the same extent in sd and dd)
  M.sa_s = ax == 0 ? (long long)M.z4 : (long long)sd.y * M.z4, M.sm_s = ax == 0 ? (long long)sd.y * M.z4 : (long long)M.z4;
  M.sa_d = ax == 0 ? (long long)M.z4 : (long long)dd.y * M.z4, M.sm_d = ax == 0 ? (long long)dd.y * M.z4 : (long long)M.z4;
  M.n_in = n_in, M.n_out = n_out, M.se = se, M.so = so;
  // runs: enough threads for the chip (~4096 waves), at least 8 steps each
  const long long lanes = (long long)M.na * M.z4;
  long long runs = std::max<long long>(1, (4096ll * kWave + lanes - 1) / lanes);
  M.run = (int)std::max<long long>(8, (steps + runs - 1) / runs);
  M.run += M.run & 1;  // (even: the up pass alternates two tap sets with the parity of its step)
  return M;
}

// dst (gd_ax long along ax) = conv_up_ax(S conv_down_ax(src)) with the stride-2 taps of axis ax; the volumes
// differ from the x-space one only along ax.  Non-zero: not available (taps, alignment) - nothing launched.
int launch_conv_downup2(const float *src, Dim3i sd, const Taps &T, const Scaling &S, int ax, int n_mid, float *dst,
                        const int *done, hipStream_t st) {
  static const bool off = getenv("UNIRES_CONV_DOWNUP") && atoi(getenv("UNIRES_CONV_DOWNUP")) == 0;
  if (off || !march2_ok(T, ax) || !conv1d_v4_ok(src, dst, sd, sd)) return 1;
  const int n = axis_len(sd, ax);
  if (2 * (n_mid - 1) + T.n[ax] - 1 > n - 1) return 1;
  const bool sc = S.dim == ax;
  March2 M = march2_args(sd, sd, ax, n, n, (n + 1) / 2, sc ? S.e : 1.f, sc ? S.o : 1.f);
  for (int t = 0; t < 12; ++t) M.k[t] = t < T.n[ax] ? T.t[ax][t] : 0.f;
  const int nm = (n + 1) / 2;
  const dim3 grid((unsigned)(((long long)M.na * M.z4 + kBlock - 1) / kBlock), (unsigned)((nm + M.run - 1) / M.run));
  if (T.n[ax] == 11)
    hipLaunchKernelGGL((k_conv1d_downup2_m<11, 6>), grid, vol_block(), 0, st, (const float4 *)src, (float4 *)dst, M, n_mid, done);
  else if (T.n[ax] == 9)
    hipLaunchKernelGGL((k_conv1d_downup2_m<9, 5>), grid, vol_block(), 0, st, (const float4 *)src, (float4 *)dst, M, n_mid, done);
  else if (T.n[ax] == 5)
    hipLaunchKernelGGL((k_conv1d_downup2_m<5, 3>), grid, vol_block(), 0, st, (const float4 *)src, (float4 *)dst, M, n_mid, done);
  else
    hipLaunchKernelGGL((k_conv1d_downup2_m<3, 2>), grid, vol_block(), 0, st, (const float4 *)src, (float4 *)dst, M, n_mid, done);
  return 0;
}

// conv_down_y in front of the one-pass x pair, in the same kernel (A^T A of BASELINE config 4 with the default
// Gaussian profile): a workgroup owns kYXRows x-space rows x kYXLanes float4 of z and walks along x.  Per input
// plane it stages the 2 kYXRows + NTY - 2 rows its conv_down_y needs in LDS (each input element fetched once by its
// workgroup; the next plane's loads travel while this one is reduced), every thread forms the y-reduced value
// of its (row, z) from NTY LDS reads, and that value enters the sliding x window of k_conv1d_downup2_m.  The
// (nx, ny / 2, nz) intermediate between the y pass and the x pair is never written: 177 MB instead of 177 + 116
// at config 4.  Same products in the same order as the separate passes: bit-identical results.
constexpr int kYXRows = 32, kYXLanes = 8, kYXPitch = 12;  // (pitch 12 float4: rows 2 apart land 32 banks apart)
struct YX2 {
  int nx, ny_in, ny_mid, z4;  // input planes / rows, output rows, float4 per row
  int nx_mid, run;            // x-space extent along x, x-space steps per run (even)
  int nyb, nzb;               // workgroups along y and z
  float ky[12], kx[12];
  float sey, soy, sex, sox;   // even / odd slice factors along y and x (1, 1: none)
  int gy;                     // FY > 0: output rows (conv_up_y as well: dst is (nx, gy, nz))
};

// FY > 0: conv_up_y as well (fan-in FY: A^T A of the x AND the y pair in one kernel, for the regime whose z part
// lives in the pull and the splat).  The x-complete values of a step's two planes pass through LDS once more: a
// thread adds those of the FY - 1 rows below its own, so a workgroup computes kYXRows x-space rows and owns the
// output rows of the upper kYXRows - (FY - 1) of them.  Products and order of k_conv1d_up2_m with unit slice
// factors: bit-identical to the four passes.
template <int NTY, int NTX, int FX, int FY>
__global__ void __launch_bounds__(kBlock) k_conv_ydown_xdownup2(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                               YX2 A, const int *__restrict__ done) {
  if (done && *done) return;
  constexpr int ROWS = 2 * kYXRows + NTY - 2, NLOAD = (ROWS * kYXLanes + kBlock - 1) / kBlock;
  constexpr int HALO = FY > 0 ? FY - 1 : 0, OWN = kYXRows - HALO;
  __shared__ float4 buf[2][ROWS * kYXPitch];
  __shared__ float4 vb[FY > 0 ? 2 : 1][FY > 0 ? kYXRows * kYXPitch : 1];
  const int tid = threadIdx.y * kWave + threadIdx.x;
  const int tz = tid % kYXLanes, ty = tid / kYXLanes;
  const int zb = blockIdx.x % A.nzb, yb = blockIdx.x / A.nzb;
  const int y0 = yb * OWN - HALO, z = zb * kYXLanes + tz, ym = y0 + ty;
  const bool mid_ok = ym >= 0 && ym < A.ny_mid;  // (x-space rows outside the volume are zeros)
  const bool own = FY > 0 ? (ty >= HALO && 2 * ym < A.gy && z < A.z4) : (mid_ok && z < A.z4);
  const int nm = (A.nx + 1) / 2;
  const int ma = blockIdx.y * A.run, mb = min(ma + A.run, nm);
  if (ma >= mb) return;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // staging: element e = (row, lane) of the plane's patch; thread t takes e = t, t + 256, ...
  int soff[NLOAD], loff[NLOAD];
  bool sok[NLOAD];
#pragma unroll
  for (int n = 0; n < NLOAD; ++n) {
    const int e = tid + n * kBlock, row = e / kYXLanes, lane = e - row * kYXLanes;
    const int yi = 2 * y0 + row, zi = zb * kYXLanes + lane;
    sok[n] = e < ROWS * kYXLanes && yi >= 0 && yi < A.ny_in && zi < A.z4;
    soff[n] = sok[n] ? yi * A.z4 + zi : 0;
    loff[n] = e < ROWS * kYXLanes ? row * kYXPitch + lane : -1;
  }
  const long long plane = (long long)A.ny_in * A.z4;
  const float4 *rd = &buf[0][0] + 2 * ty * kYXPitch + tz;
  const float scy = mid_ok ? ((ym & 1) ? A.soy : A.sey) : 0.f;
  float kex[FX], kox[FX];
#pragma unroll
  for (int i = 0; i < FX; ++i) kex[i] = 2 * i < NTX ? A.kx[2 * i] : 0.f, kox[i] = 2 * i + 1 < NTX ? A.kx[2 * i + 1] : 0.f;
  const int m0 = max(ma - (FX - 1), 0);
  const int m_end = min(mb, A.nx_mid);  // x-space steps that read planes: m0 .. m_end - 1
  const bool any = m0 < m_end;
  int p = 2 * m0;
  const int plast = 2 * (m_end - 1) + NTX - 1;
  // (value-returning: a `cond ? src[i] : zero` of two lvalues selects between addresses and keeps `zero` in scratch)
  auto ldz = [&](bool ok, long long i) __attribute__((always_inline)) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = src[i];
    return v;
  };
  float4 fly[NLOAD];  // the plane after the one in LDS, in flight
  if (any) {
#pragma unroll
    for (int n = 0; n < NLOAD; ++n)
      if (loff[n] >= 0) buf[p & 1][loff[n]] = ldz(sok[n], (long long)p * plane + soff[n]);
#pragma unroll
    for (int n = 0; n < NLOAD; ++n) fly[n] = ldz(sok[n] && p + 1 <= plast, (long long)(p + 1) * plane + soff[n]);
  }
  __syncthreads();
  // consumes plane p: its y-reduced value for this thread's (row, z).  Plane p + 1 (in registers since the last
  // call) goes to the other LDS buffer, plane p + 2 is requested: two planes of loads in flight per workgroup.
  auto feed = [&]() __attribute__((always_inline)) {
    float4 nxt[NLOAD];
    const bool more2 = p + 2 <= plast;
#pragma unroll
    for (int n = 0; n < NLOAD; ++n) nxt[n] = ldz(more2 && sok[n], (long long)(p + 2) * plane + soff[n]);
    const float4 *r = rd + (p & 1) * (ROWS * kYXPitch);
    float4 acc = zero;
#pragma unroll
    for (int t = 0; t < NTY; ++t) acc = fma4(A.ky[t], r[t * kYXPitch], acc);
    if (p + 1 <= plast) {
#pragma unroll
      for (int n = 0; n < NLOAD; ++n)
        if (loff[n] >= 0) buf[(p + 1) & 1][loff[n]] = fly[n];
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NLOAD; ++n) fly[n] = nxt[n];
    ++p;
    return make_float4(acc.x * scy, acc.y * scy, acc.z * scy, acc.w * scy);
  };
  float4 md[FX], w[NTX];
#pragma unroll
  for (int i = 0; i < FX; ++i) md[i] = zero;
#pragma unroll
  for (int t = 0; t < NTX; ++t) w[t] = zero;
  if (any) {
#pragma unroll
    for (int t = 0; t + 2 < NTX; ++t) w[t] = feed();
  }
  const int orows = FY > 0 ? A.gy : A.ny_mid;
  float4 *q = dst + ((long long)(2 * ma) * orows + (FY > 0 ? 2 * ym : ym)) * A.z4 + z;
  const long long oplane = (long long)orows * A.z4;
  float kye[FY > 0 ? FY : 1], kyo[FY > 0 ? FY : 1];  // conv_up_y taps at even / odd offsets (= A.ky: same profile)
#pragma unroll
  for (int i = 0; i < FY; ++i) kye[i] = 2 * i < NTY ? A.ky[2 * i] : 0.f, kyo[i] = 2 * i + 1 < NTY ? A.ky[2 * i + 1] : 0.f;
  for (int m = m0; m < mb; ++m) {
    float4 nx = zero;
    if (m < m_end) {
      w[NTX - 2] = feed(), w[NTX - 1] = feed();
      float4 acc = zero;
#pragma unroll
      for (int t = 0; t < NTX; ++t) acc = fma4(A.kx[t], w[t], acc);
      const float sc = (m & 1) ? A.sox : A.sex;
      nx = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
    }
#pragma unroll
    for (int i = FX - 1; i > 0; --i) md[i] = md[i - 1];
    md[0] = nx;
    if (m >= ma) {
      float4 e = zero, o = zero;
#pragma unroll
      for (int i = FX - 1; i >= 0; --i) e = fma4(kex[i], md[i], e), o = fma4(kox[i], md[i], o);
      if constexpr (FY > 0) {
        float4 *v0 = &vb[0][ty * kYXPitch + tz], *v1 = &vb[1][ty * kYXPitch + tz];
        *v0 = e, *v1 = o;
        __syncthreads();
        if (own) {
          float4 ee = zero, eo = zero, oe = zero, oo = zero;  // plane 2m rows 2ym / 2ym + 1, plane 2m + 1 likewise
#pragma unroll
          for (int i = FY - 1; i >= 0; --i) {
            const float4 a = i ? v0[-i * kYXPitch] : e, b = i ? v1[-i * kYXPitch] : o;
            ee = fma4(kye[i], a, ee), eo = fma4(kyo[i], a, eo), oe = fma4(kye[i], b, oe), oo = fma4(kyo[i], b, oo);
          }
          const bool row1 = 2 * ym + 1 < A.gy;
          q[0] = ee;
          if (row1) q[A.z4] = eo;
          if (2 * m + 1 < A.nx) {
            q[oplane] = oe;
            if (row1) q[oplane + A.z4] = oo;
          }
        }
        __syncthreads();  // (vb is rewritten by the next step)
      } else if (own) {
        q[0] = e;
        if (2 * m + 1 < A.nx) q[oplane] = o;
      }
      q += 2 * oplane;
    }
#pragma unroll
    for (int t = 0; t + 2 < NTX; ++t) w[t] = w[t + 2];
  }
}

// dst (nx, ny_mid, nz) = conv_up_x(Sx conv_down_x(Sy conv_down_y(src))) for stride-2 profiles along x and y; src is
// (nx, ny, nz), the x-space extents are nx_mid / ny_mid.  Non-zero: not available, nothing launched.
int launch_conv_ydown_xdownup2(const float *src, Dim3i sd, const Taps &T, const Scaling &S, int nx_mid, int ny_mid,
                               int gy, float *dst, const int *done, hipStream_t st) {
  static const bool off = getenv("UNIRES_CONV_YX") && atoi(getenv("UNIRES_CONV_YX")) == 0;
  if (off || !march2_ok(T, 0) || !march2_ok(T, 1) || (sd.z & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return 1;
  if (2 * (nx_mid - 1) + T.n[0] - 1 > sd.x - 1 || 2 * (ny_mid - 1) + T.n[1] - 1 > sd.y - 1) return 1;
  if ((long long)sd.x * sd.y * (sd.z / 4) >= (1ll << 31)) return 1;
  YX2 A;
  A.nx = sd.x, A.ny_in = sd.y, A.ny_mid = ny_mid, A.z4 = sd.z / 4, A.nx_mid = nx_mid;
  A.gy = gy;
  const int fy = (T.n[1] + 1) / 2, own = gy > 0 ? kYXRows - (fy - 1) : kYXRows;
  A.nyb = gy > 0 ? ((gy + 1) / 2 + own - 1) / own : (ny_mid + own - 1) / own;
  A.nzb = (A.z4 + kYXLanes - 1) / kYXLanes;
  for (int t = 0; t < 12; ++t) A.ky[t] = t < T.n[1] ? T.t[1][t] : 0.f, A.kx[t] = t < T.n[0] ? T.t[0][t] : 0.f;
  A.sey = S.dim == 1 ? S.e : 1.f, A.soy = S.dim == 1 ? S.o : 1.f;
  A.sex = S.dim == 0 ? S.e : 1.f, A.sox = S.dim == 0 ? S.o : 1.f;
  // runs: ~768 workgroups (three per CU), at least 6 steps each
  static const int want = getenv("UNIRES_CONV_YX_BLOCKS") ? atoi(getenv("UNIRES_CONV_YX_BLOCKS")) : 768;
  const int nm = (sd.x + 1) / 2, cols = A.nyb * A.nzb;
  const int runs = std::max(1, (want + cols - 1) / cols);
  A.run = std::max(6, (nm + runs - 1) / runs);
  A.run += A.run & 1;
  const dim3 grid((unsigned)cols, (unsigned)((nm + A.run - 1) / A.run));
#define YX_CASE(NY_, NX_, FX_)                                                                                     \
  if (T.n[1] == NY_ && T.n[0] == NX_) {                                                                            \
    if (gy > 0)                                                                                                    \
      hipLaunchKernelGGL((k_conv_ydown_xdownup2<NY_, NX_, FX_, (NY_ + 1) / 2>), grid, vol_block(), 0, st,           \
                         (const float4 *)src, (float4 *)dst, A, done);                                             \
    else                                                                                                           \
      hipLaunchKernelGGL((k_conv_ydown_xdownup2<NY_, NX_, FX_, 0>), grid, vol_block(), 0, st, (const float4 *)src, \
                         (float4 *)dst, A, done);                                                                  \
    return 0;                                                                                                      \
  }
  YX_CASE(11, 3, 2) YX_CASE(11, 5, 3) YX_CASE(11, 11, 6) YX_CASE(5, 3, 2) YX_CASE(5, 5, 3) YX_CASE(5, 11, 6)
  YX_CASE(3, 3, 2) YX_CASE(3, 5, 3) YX_CASE(3, 11, 6)
  YX_CASE(9, 3, 2) YX_CASE(9, 5, 3) YX_CASE(9, 9, 5) YX_CASE(5, 9, 5) YX_CASE(3, 9, 5)  // (the Gaussian as the plan trims it)
#undef YX_CASE
  return 1;
}

void launch_conv_down_sep(const float *g, Dim3i gd, const Taps &T, const Scaling &S, float *dst,
                          Dim3i xd, float *a, float *b, const int *done, hipStream_t st) {
  const float *cur = g;
  Dim3i cd = gd;
  int todo = 0;
  for (int ax = 0; ax < 3; ++ax) todo += !axis_is_dirac(T, ax) || S.dim == ax;
  if (todo == 0) {  // identity: plain copy
    (void)hipMemcpyAsync(dst, g, gd.numel() * sizeof(float), hipMemcpyDeviceToDevice, st);
    return;
  }
  static const bool fuse_xy = !(getenv("UNIRES_CONV_XY") && atoi(getenv("UNIRES_CONV_XY")) == 0);
  for (int ax = 2; ax >= 0; --ax) {
    if (axis_is_dirac(T, ax) && S.dim != ax) continue;
    if (ax == 1 && fuse_xy && todo == 2 && T.n[0] <= kConvXYMax && T.n[1] <= kConvXYMax && T.n[0] * T.n[1] <= 16 &&
        conv1d_v4_ok(cur, dst, cd, xd)) {
      // y and x passes in one kernel (cur is z-complete: cd.z == xd.z)
      Taps2 K2;
      for (int t = 0; t < 8; ++t) K2.x[t] = T.t[0][t], K2.y[t] = T.t[1][t];
      hipLaunchKernelGGL(k_conv2d_down_xy_v4, conv2d_grid(xd), vol_block(), 0, st,
                         (const float4 *)cur, cd, K2, T.n[0], T.s[0], T.n[1], T.s[1], S.dim == 0 ? S.e : 1.f,
                         S.dim == 0 ? S.o : 1.f, S.dim == 1 ? S.e : 1.f, S.dim == 1 ? S.o : 1.f, (float4 *)dst, xd,
                         done);
      return;
    }
    const Dim3i od = with_axis(cd, ax, axis_len(xd, ax));
    float *out = --todo == 0 ? dst : (cur == a ? b : a);
    Taps1 K;
    for (int t = 0; t < UNIRES_MAX_TAPS; ++t) K.t[t] = T.t[ax][t];
    const bool sc = S.dim == ax;
    if (march2_ok(T, ax) && conv1d_v4_ok(cur, out, cd, od)) {
      const int n_in = axis_len(cd, ax), n_out = axis_len(od, ax);
      March2 M = march2_args(cd, od, ax, n_in, n_out, n_out, sc ? S.e : 1.f, sc ? S.o : 1.f);
      for (int t = 0; t < 12; ++t) M.k[t] = t < T.n[ax] ? T.t[ax][t] : 0.f;
      const dim3 grid((unsigned)(((long long)M.na * M.z4 + kBlock - 1) / kBlock), (unsigned)((n_out + M.run - 1) / M.run));
      if (T.n[ax] == 11)
        hipLaunchKernelGGL((k_conv1d_down2_m<11>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M, done);
      else if (T.n[ax] == 9)
        hipLaunchKernelGGL((k_conv1d_down2_m<9>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M, done);
      else if (T.n[ax] == 5)
        hipLaunchKernelGGL((k_conv1d_down2_m<5>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M, done);
      else
        hipLaunchKernelGGL((k_conv1d_down2_m<3>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M, done);
    } else if (ax == 2 && T.s[2] <= 8)
      hipLaunchKernelGGL(k_conv1d_down_z, conv1d_grid(od), vol_block(), 0, st, cur, cd, K, T.n[2], T.s[2],
                         sc ? S.e : 1.f, sc ? S.o : 1.f, out, od, done);
    else if (ax != 2 && conv1d_v4_ok(cur, out, cd, od))
      hipLaunchKernelGGL(k_conv1d_down_v4, conv1d_grid(Dim3i{od.x, od.y, od.z / 4}), vol_block(), 0, st,
                         (const float4 *)cur, cd, ax, K, T.n[ax], T.s[ax], sc ? S.e : 1.f, sc ? S.o : 1.f,
                         (float4 *)out, od, done);
    else
      hipLaunchKernelGGL(k_conv1d_down, conv1d_grid(od), vol_block(), 0, st, cur, cd, ax, K, T.n[ax],
                         T.s[ax], sc ? S.e : 1.f, sc ? S.o : 1.f, out, od, done);
    cur = out, cd = od;
  }
}

// g = conv_up(S xs): passes x, y, z; returns the buffer (a or b) that holds the grid volume.
// conv_up along z for STRIDE 2 (isotropic 2 x down-sampling: BASELINE config 4, Gaussian profile = 11 taps,
// fan-in 6) without the LDS stage (round 4).  out[2m] = sum_i ker[2i] s[m - i], out[2m + 1] = sum_i ker[2i + 1] s[m - i]:
// a lane loads TWO source voxels (8 bytes), takes the five older ones from its lower neighbours by wave shifts and
// writes FOUR outputs as one 16-byte store; three lanes of halo per pass of 64.  ~8 instructions per output where
// the staged form has ~25 (a dozen LDS reads among them) - and these passes are bound by what a wave issues.
__device__ __forceinline__ float dpp_shr1(float v) {  // lane l gets lane l - 1's value (lane 0: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}

struct UpZ2Taps {
  float ke[6], ko[6];
};

// FY: the stride-2 conv_up along y fused in front (the source row of output row (x, u_y) is formed on the fly from
// the <= 6 x-space rows that feed it: six 8-byte loads per lane, cache hits, instead of a pass that writes and
// re-reads the (X, gy, sz) intermediate - 119 MB each way at BASELINE config 4).
struct UpY2 {
  float ke[6], ko[6];  // y taps at even / odd offsets
  float se, so;        // even / odd slice factors along y
  int ny_src;          // source rows
};

template <bool FY>
__global__ void __launch_bounds__(kBlock)
    k_conv1d_up_z2(const float *__restrict__ src, Dim3i sd, UpZ2Taps K, float se, float so, float *__restrict__ dst,
                   Dim3i dd, UpY2 Y) {
  const int lane = threadIdx.x, w = threadIdx.y;
  const long long nrows = (long long)dd.x * dd.y;
  const int nm = (dd.z + 1) / 2, npair = (nm + 1) / 2;
  constexpr int H = 3, U = kWave - H;  // halo lanes, useful lanes per pass
  for (long long row = (long long)blockIdx.x * (kBlock / kWave) + w; row < nrows; row += (long long)gridDim.x * (kBlock / kWave)) {
    const float *srow = src + row * sd.z;
    // FY: output row (x, uy) = sum_i ty[i] S(my - i) source row (x, my - i), my = uy / 2, taps by the parity of uy
    float ty[6];
    const float *yrow[6];
    if (FY) {
      const int x = (int)(row / dd.y), uy = (int)(row - (long long)x * dd.y), my = uy >> 1;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int c = my - i;
        const bool ok = c >= 0 && c < Y.ny_src;
        ty[i] = ok ? ((uy & 1) ? Y.ko[i] : Y.ke[i]) * ((c & 1) ? Y.so : Y.se) : 0.f;
        yrow[i] = src + ((long long)x * Y.ny_src + (ok ? c : 0)) * sd.z;
      }
    }
    float *drow = dst + row * dd.z;
    for (int jb = 0; jb < npair; jb += U) {
      const int j = jb - H + lane, c = 2 * j;
      float s0 = 0.f, s1 = 0.f;
      if (FY) {
        if (c >= 0 && c + 1 < sd.z) {
          float2 v[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) v[i] = ld2_u(yrow[i] + c);
#pragma unroll
          for (int i = 5; i >= 0; --i) s0 = fmaf(ty[i], v[i].x, s0), s1 = fmaf(ty[i], v[i].y, s1);  // (ascending source row)
        } else if (c >= 0 && c < sd.z) {
#pragma unroll
          for (int i = 5; i >= 0; --i) s0 = fmaf(ty[i], yrow[i][c], s0);
        }
        s0 *= se, s1 *= so;
      } else if (c >= 0 && c + 1 < sd.z) {
        const float2 v = ld2_u(srow + c);
        s0 = v.x * se, s1 = v.y * so;
      } else if (c >= 0 && c < sd.z) {
        s0 = srow[c] * se;
      }
      const float b0 = dpp_shr1(s0), b1 = dpp_shr1(s1);  // s[2j - 2], s[2j - 1]
      const float c0 = dpp_shr1(b0), c1 = dpp_shr1(b1);  // s[2j - 4], s[2j - 3]
      const float d1 = dpp_shr1(c1);                      // s[2j - 5]
      // m = 2j reads s[2j - i], m = 2j + 1 reads s[2j + 1 - i], i = 0 .. 5
      const float o0 = K.ke[0] * s0 + K.ke[1] * b1 + K.ke[2] * b0 + K.ke[3] * c1 + K.ke[4] * c0 + K.ke[5] * d1;
      const float o1 = K.ko[0] * s0 + K.ko[1] * b1 + K.ko[2] * b0 + K.ko[3] * c1 + K.ko[4] * c0 + K.ko[5] * d1;
      const float o2 = K.ke[0] * s1 + K.ke[1] * s0 + K.ke[2] * b1 + K.ke[3] * b0 + K.ke[4] * c1 + K.ke[5] * c0;
      const float o3 = K.ko[0] * s1 + K.ko[1] * s0 + K.ko[2] * b1 + K.ko[3] * b0 + K.ko[4] * c1 + K.ko[5] * c0;
      const int u = 4 * j;
      if (lane >= H && j < npair) {
        if (u + 3 < dd.z) {
          __builtin_memcpy(drow + u, &(const float4 &)make_float4(o0, o1, o2, o3), sizeof(float4));
        } else {
          if (u < dd.z) drow[u] = o0;
          if (u + 1 < dd.z) drow[u + 1] = o1;
          if (u + 2 < dd.z) drow[u + 2] = o2;
        }
      }
    }
  }
}

// conv_up along y AND z at stride 2 in one kernel, the y part through LDS (round 4; k_conv1d_up_z2<true> gathers its
// six source rows with 8-byte global loads per lane and pass and lost to the two passes).  A workgroup takes one x
// plane and kUpYZRows x-space rows: it stages the kUpYZRows + FY - 1 source rows their 2 kUpYZRows output rows read
// (each source element fetched once per workgroup, 16-byte loads), then every wave forms output rows: the y sum of
// a lane's two source voxels from FY 8-byte LDS reads, the z part by wave shifts exactly as k_conv1d_up_z2, one
// 16-byte store per lane.  The (X, gy, sz) intermediate (119 MB each way at BASELINE config 4) is never written.
// Same products in the same order as k_conv1d_up2_m followed by k_conv1d_up_z2: bit-identical results.
#ifndef UNIRES_UPYZ_ROWS
#define UNIRES_UPYZ_ROWS 16
#endif
constexpr int kUpYZRows = UNIRES_UPYZ_ROWS;
struct UpYZ {
  float kye[6], kyo[6];  // y taps at even / odd offsets
  float sey, soy;        // even / odd slice factors along y
  int ny_src, pitch;     // source rows; LDS floats per staged row (sz rounded up to a multiple of 4)
};

template <int FY>
__global__ void __launch_bounds__(kBlock)
    k_conv_up_yz2(const float *__restrict__ src, Dim3i sd, UpZ2Taps K, float se, float so, float *__restrict__ dst, Dim3i dd,
                  UpYZ Y, int nyb) {
  extern __shared__ __align__(16) float rows[];  // (kUpYZRows + FY - 1) x pitch
  constexpr int NR = kUpYZRows + FY - 1;
  const int lane = threadIdx.x, w = threadIdx.y, tid = w * kWave + lane;
  const int x = blockIdx.x / nyb, yb = blockIdx.x - x * nyb;
  const int my0 = yb * kUpYZRows, c0 = my0 - (FY - 1);  // first x-space row of the block, first staged row
  const int p4 = Y.pitch / 4;
  for (int e = tid; e < NR * p4; e += kBlock) {
    const int r = e / p4, q = e - r * p4, c = c0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c >= 0 && c < Y.ny_src && 4 * q < sd.z)
      v = *reinterpret_cast<const float4 *>(src + ((long long)x * Y.ny_src + c) * sd.z + 4 * q);  // (sd.z % 4 == 0)
    reinterpret_cast<float4 *>(rows)[e] = v;
  }
  __syncthreads();
  const int nm = (dd.z + 1) / 2, npair = (nm + 1) / 2;
  constexpr int H = 3, U = kWave - H;  // halo lanes, useful lanes per pass (as k_conv1d_up_z2)
  for (int ro = w; ro < 2 * kUpYZRows; ro += kBlock / kWave) {
    const int uy = 2 * my0 + ro;
    if (uy >= dd.y) break;
    const int my = uy >> 1;
    float ty[FY];
    const float *yrow[FY];
#pragma unroll
    for (int i = 0; i < FY; ++i) {
      const int c = my - i;
      ty[i] = ((uy & 1) ? Y.kyo[i] : Y.kye[i]) * ((c & 1) ? Y.soy : Y.sey);
      yrow[i] = rows + (c - c0) * Y.pitch;  // (c - c0 in [0, NR): rows outside the volume were staged as zeros)
    }
    float *drow = dst + ((long long)x * dd.y + uy) * dd.z;
    for (int jb = 0; jb < npair; jb += U) {
      const int j = jb - H + lane, c = 2 * j;
      float s0 = 0.f, s1 = 0.f;
      if (c >= 0 && c + 1 < sd.z) {
#pragma unroll
        for (int i = FY - 1; i >= 0; --i) {  // (ascending source row)
          const float2 v = *reinterpret_cast<const float2 *>(yrow[i] + c);
          s0 = fmaf(ty[i], v.x, s0), s1 = fmaf(ty[i], v.y, s1);
        }
      } else if (c >= 0 && c < sd.z) {
#pragma unroll
        for (int i = FY - 1; i >= 0; --i) s0 = fmaf(ty[i], yrow[i][c], s0);
      }
      s0 *= se, s1 *= so;
      const float b0 = dpp_shr1(s0), b1 = dpp_shr1(s1);  // s[2j - 2], s[2j - 1]
      const float c0v = dpp_shr1(b0), c1v = dpp_shr1(b1);  // s[2j - 4], s[2j - 3]
      const float d1 = dpp_shr1(c1v);                      // s[2j - 5]
      const float o0 = K.ke[0] * s0 + K.ke[1] * b1 + K.ke[2] * b0 + K.ke[3] * c1v + K.ke[4] * c0v + K.ke[5] * d1;
      const float o1 = K.ko[0] * s0 + K.ko[1] * b1 + K.ko[2] * b0 + K.ko[3] * c1v + K.ko[4] * c0v + K.ko[5] * d1;
      const float o2 = K.ke[0] * s1 + K.ke[1] * s0 + K.ke[2] * b1 + K.ke[3] * b0 + K.ke[4] * c1v + K.ke[5] * c0v;
      const float o3 = K.ko[0] * s1 + K.ko[1] * s0 + K.ko[2] * b1 + K.ko[3] * b0 + K.ko[4] * c1v + K.ko[5] * c0v;
      const int u = 4 * j;
      if (lane >= H && j < npair) {
        if (u + 3 < dd.z) {
          __builtin_memcpy(drow + u, &(const float4 &)make_float4(o0, o1, o2, o3), sizeof(float4));
        } else {
          if (u < dd.z) drow[u] = o0;
          if (u + 1 < dd.z) drow[u + 1] = o1;
          if (u + 2 < dd.z) drow[u + 2] = o2;
        }
      }
    }
  }
}

float *launch_conv_up_sep(const float *xs, Dim3i xd, const Taps &T, const Scaling &S, Dim3i gd,
                          float *a, float *b, hipStream_t st) {
  const float *cur = xs;
  Dim3i cd = xd;
  float *out = nullptr;
  static const bool fuse_xy = !(getenv("UNIRES_CONV_XY") && atoi(getenv("UNIRES_CONV_XY")) == 0);
  auto active = [&](int ax) { return !(axis_is_dirac(T, ax) && S.dim != ax); };
  auto fan = [&](int ax) { return (T.n[ax] + T.s[ax] - 1) / T.s[ax]; };
  for (int ax = 0; ax < 3; ++ax) {
    if (!active(ax)) continue;
    if (ax == 0 && fuse_xy && active(1) && fan(0) <= 3 && fan(1) <= 3 && cd.numel() / 4 < (1ull << 32)) {
      // x and y passes in one kernel
      const Dim3i od = Dim3i{gd.x, gd.y, cd.z};
      float *o2 = cur == a ? b : a;
      if (conv1d_v4_ok(cur, o2, cd, od)) {
        Taps1 KX, KY;
        for (int t = 0; t < UNIRES_MAX_TAPS; ++t) KX.t[t] = T.t[0][t], KY.t[t] = T.t[1][t];
        const float sex = S.dim == 0 ? S.e : 1.f, sox = S.dim == 0 ? S.o : 1.f, sey = S.dim == 1 ? S.e : 1.f,
                    soy = S.dim == 1 ? S.o : 1.f;
        dim3 g = conv2d_grid(od);
        if (fan(0) <= 2 && fan(1) <= 2) {
          g.y = (od.x + 3) / 4;
          hipLaunchKernelGGL((k_conv2d_up_xy_v4_t<2, 2, 4>), g, vol_block(), 0, st, (const float4 *)cur, cd, KX, KY,
                             T.n[0], T.s[0], T.n[1], T.s[1], sex, sox, sey, soy, (float4 *)o2, od);
        } else {
          g.y = (od.x + 1) / 2;
          hipLaunchKernelGGL((k_conv2d_up_xy_v4_t<3, 3, 2>), g, vol_block(), 0, st, (const float4 *)cur, cd, KX, KY,
                             T.n[0], T.s[0], T.n[1], T.s[1], sex, sox, sey, soy, (float4 *)o2, od);
        }
        out = o2, cur = o2, cd = od;
        ax = 1;  // (the loop continues with z)
        continue;
      }
    }
    const Dim3i od = with_axis(cd, ax, axis_len(gd, ax));
    out = cur == a ? b : a;
    Taps1 K;
    for (int t = 0; t < UNIRES_MAX_TAPS; ++t) K.t[t] = T.t[ax][t];
    const bool sc = S.dim == ax;
    static const bool no_z2 = getenv("UNIRES_UPZ2") && atoi(getenv("UNIRES_UPZ2")) == 0;
    // (measured and left off, UNIRES_UPYZ=1: the y pass fused into the z kernel - 104 us for the pair's 31 + 67:
    // six 8-byte row loads per lane and pass cost more than the 119 MB intermediate they spare)
    static const bool no_yz = !(getenv("UNIRES_UPYZ") && atoi(getenv("UNIRES_UPYZ")) == 1);
    const bool z2 = !no_z2 && T.s[2] == 2 && T.n[2] <= 12 && cd.z >= 2 && active(2);
    static const bool no_yz_lds = getenv("UNIRES_UPYZ_LDS") && atoi(getenv("UNIRES_UPYZ_LDS")) == 0;
    if (ax == 1 && z2 && !no_yz_lds && march2_ok(T, 1) && (cd.z & 3) == 0 && cd.z <= 512 && (((uintptr_t)cur) & 15) == 0) {
      // y and z passes in one kernel, the y part through an LDS stage: no (X, gy, sz) intermediate
      const Dim3i oz = Dim3i{cd.x, gd.y, gd.z};
      UpZ2Taps Z;
      UpYZ Y;
      for (int i = 0; i < 6; ++i) {
        Z.ke[i] = 2 * i < T.n[2] ? T.t[2][2 * i] : 0.f, Z.ko[i] = 2 * i + 1 < T.n[2] ? T.t[2][2 * i + 1] : 0.f;
        Y.kye[i] = 2 * i < T.n[1] ? T.t[1][2 * i] : 0.f, Y.kyo[i] = 2 * i + 1 < T.n[1] ? T.t[1][2 * i + 1] : 0.f;
      }
      Y.sey = S.dim == 1 ? S.e : 1.f, Y.soy = S.dim == 1 ? S.o : 1.f, Y.ny_src = cd.y, Y.pitch = cd.z;
      const int nyb = (gd.y + 2 * kUpYZRows - 1) / (2 * kUpYZRows), fy = (T.n[1] + 1) / 2;
      const dim3 grid((unsigned)(cd.x * nyb));
      const size_t lds = (size_t)(kUpYZRows + fy - 1) * Y.pitch * sizeof(float);
      const float sez = S.dim == 2 ? S.e : 1.f, soz = S.dim == 2 ? S.o : 1.f;
      if (fy == 6)
        hipLaunchKernelGGL((k_conv_up_yz2<6>), grid, vol_block(), lds, st, cur, cd, Z, sez, soz, out, oz, Y, nyb);
      else if (fy == 5)
        hipLaunchKernelGGL((k_conv_up_yz2<5>), grid, vol_block(), lds, st, cur, cd, Z, sez, soz, out, oz, Y, nyb);
      else if (fy == 3)
        hipLaunchKernelGGL((k_conv_up_yz2<3>), grid, vol_block(), lds, st, cur, cd, Z, sez, soz, out, oz, Y, nyb);
      else
        hipLaunchKernelGGL((k_conv_up_yz2<2>), grid, vol_block(), lds, st, cur, cd, Z, sez, soz, out, oz, Y, nyb);
      cur = out, cd = oz;
      ax = 2;  // (z is done too)
      continue;
    }
    if (ax == 1 && z2 && !no_yz && T.s[1] == 2 && T.n[1] <= 12) {
      // y and z passes in one kernel: no (X, gy, sz) intermediate
      const Dim3i oz = Dim3i{cd.x, gd.y, gd.z};
      UpZ2Taps Z;
      UpY2 Y;
      for (int i = 0; i < 6; ++i) {
        Z.ke[i] = 2 * i < T.n[2] ? T.t[2][2 * i] : 0.f, Z.ko[i] = 2 * i + 1 < T.n[2] ? T.t[2][2 * i + 1] : 0.f;
        Y.ke[i] = 2 * i < T.n[1] ? T.t[1][2 * i] : 0.f, Y.ko[i] = 2 * i + 1 < T.n[1] ? T.t[1][2 * i + 1] : 0.f;
      }
      Y.se = S.dim == 1 ? S.e : 1.f, Y.so = S.dim == 1 ? S.o : 1.f, Y.ny_src = cd.y;
      const long long rows = (long long)oz.x * oz.y;
      const unsigned blocks = (unsigned)std::min<long long>((rows + 3) / 4, 16384);
      hipLaunchKernelGGL((k_conv1d_up_z2<true>), dim3(blocks), vol_block(), 0, st, cur, cd, Z, S.dim == 2 ? S.e : 1.f,
                         S.dim == 2 ? S.o : 1.f, out, oz, Y);
      cur = out, cd = oz;
      ax = 2;  // (z is done too)
      continue;
    }
    if (ax == 2 && z2) {
      UpZ2Taps Z;
      for (int i = 0; i < 6; ++i) Z.ke[i] = 2 * i < T.n[2] ? T.t[2][2 * i] : 0.f, Z.ko[i] = 2 * i + 1 < T.n[2] ? T.t[2][2 * i + 1] : 0.f;
      const long long rows = (long long)od.x * od.y;
      const unsigned blocks = (unsigned)std::min<long long>((rows + 3) / 4, 16384);
      hipLaunchKernelGGL((k_conv1d_up_z2<false>), dim3(blocks), vol_block(), 0, st, cur, cd, Z, sc ? S.e : 1.f, sc ? S.o : 1.f, out,
                         od, UpY2());
    } else if (march2_ok(T, ax) && conv1d_v4_ok(cur, out, cd, od)) {
      const int n_in = axis_len(cd, ax), n_out = axis_len(od, ax);
      March2 M = march2_args(cd, od, ax, n_in, n_out, (n_out + 1) / 2, sc ? S.e : 1.f, sc ? S.o : 1.f);
      for (int i = 0; i < 6; ++i)
        M.k[i] = 2 * i < T.n[ax] ? T.t[ax][2 * i] : 0.f, M.k[6 + i] = 2 * i + 1 < T.n[ax] ? T.t[ax][2 * i + 1] : 0.f;
      const int nm = (n_out + 1) / 2;
      const dim3 grid((unsigned)(((long long)M.na * M.z4 + kBlock - 1) / kBlock), (unsigned)((nm + M.run - 1) / M.run));
      if (T.n[ax] == 11)
        hipLaunchKernelGGL((k_conv1d_up2_m<6>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M);
      else if (T.n[ax] == 9)
        hipLaunchKernelGGL((k_conv1d_up2_m<5>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M);
      else if (T.n[ax] == 5)
        hipLaunchKernelGGL((k_conv1d_up2_m<3>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M);
      else
        hipLaunchKernelGGL((k_conv1d_up2_m<2>), grid, vol_block(), 0, st, (const float4 *)cur, (float4 *)out, M);
    } else if (ax == 2)
      hipLaunchKernelGGL(k_conv1d_up_z, conv1d_grid(od), vol_block(), 0, st, cur, cd, K, T.n[2], T.s[2],
                         sc ? S.e : 1.f, sc ? S.o : 1.f, out, od);
    else if (conv1d_v4_ok(cur, out, cd, od))
      hipLaunchKernelGGL(k_conv1d_up_v4, conv1d_grid(Dim3i{od.x, od.y, od.z / 4}), vol_block(), 0, st,
                         (const float4 *)cur, cd, ax, K, T.n[ax], T.s[ax], sc ? S.e : 1.f, sc ? S.o : 1.f,
                         (float4 *)out, od);
    else
      hipLaunchKernelGGL(k_conv1d_up, conv1d_grid(od), vol_block(), 0, st, cur, cd, ax, K, T.n[ax], T.s[ax],
                         sc ? S.e : 1.f, sc ? S.o : 1.f, out, od);
    cur = out, cd = od;
  }
  if (!out) {  // identity
    (void)hipMemcpyAsync(a, xs, xd.numel() * sizeof(float), hipMemcpyDeviceToDevice, st);
    out = a;
  }
  return out;
}

// dst[(i,j,k), 0..2] = gradient of the trilinear sample at A (i,j,k)  (nitorch grid_grad layout)
__global__ void __launch_bounds__(kBlock) k_pull_grad(const float *__restrict__ src, Dim3i sd,
                                                      Affine A, float *__restrict__ dst, Dim3i gd,
                                                      float tol) {
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (k >= gd.z || j >= gd.y) return;
  float gx, gy, gz, dx, dy, dz;
  affine_point(A, (float)i, (float)j, (float)k, gx, gy, gz);
  pull_grad_sample(src, sd, gx, gy, gz, tol, dx, dy, dz);
  float *o = dst + (((size_t)i * gd.y + j) * gd.z + k) * 3;
  o[0] = dx, o[1] = dy, o[2] = dz;
}

void launch_pull_grad(const float *src, Dim3i sd, const Affine &A, float *dst, Dim3i gd, float tol,
                      hipStream_t st) {
  hipLaunchKernelGGL(k_pull_grad, vol_grid(gd), vol_block(), 0, st, src, sd, A, dst, gd, tol);
}

void launch_conv_down(const float *src, Dim3i gd, const Taps &T, const Scaling &S, float *dst,
                      Dim3i xd, const int *done, hipStream_t st) {
  hipLaunchKernelGGL(k_conv_down, vol_grid(xd), vol_block(), 0, st, src, gd, T, S, dst, xd, done);
}

void launch_conv_up(const float *xs, Dim3i xd, const Taps &T, const Scaling &S, float *dst,
                    Dim3i gd, hipStream_t st) {
  hipLaunchKernelGGL(k_conv_up, vol_grid(gd), vol_block(), 0, st, xs, xd, T, S, dst, gd);
}

void launch_grad(const float *src, Dim3i d, const float vx[3], float *dst3, hipStream_t st) {
  hipLaunchKernelGGL(k_grad, vol_grid(d), vol_block(), 0, st, src, d, 1.f / vx[0], 1.f / vx[1],
                     1.f / vx[2], dst3);
}

void launch_div(const float *ua, const float *ub, float ca, float cb, Dim3i d, const float vx[3],
                float scale, const float *add, float *dst, hipStream_t st) {
  hipLaunchKernelGGL(k_div, vol_grid(d), vol_block(), 0, st, ua, ub, ca, cb, d, 1.f / vx[0],
                     1.f / vx[1], 1.f / vx[2], scale, add, dst);
}

int dtd_num_blocks(Dim3i d) {
  const long long ntiles = (long long)((d.z + kWave - 1) / kWave) * ((d.y + 3) / 4) * d.x;
  return (int)(ntiles < kMaxPartials ? ntiles : kMaxPartials);
}

// partials (nullable) must hold dtd_num_blocks(d) doubles.
void launch_dtd(const float *src, Dim3i d, const float vx[3], float a, float c, float *dst,
                double *partials, const float *objb, const int *done, hipStream_t st) {
  const float cx = c / (vx[0] * vx[0]), cy = c / (vx[1] * vx[1]), cz = c / (vx[2] * vx[2]);
  const dim3 grid(dtd_num_blocks(d));
  if (partials)
    hipLaunchKernelGGL(k_dtd<true>, grid, vol_block(), 0, st, src, d, cx, cy, cz, a, dst,
                       partials, objb, done);
  else
    hipLaunchKernelGGL(k_dtd<false>, grid, vol_block(), 0, st, src, d, cx, cy, cz, a, dst,
                       partials, nullptr, done);
}

}  // namespace unires
