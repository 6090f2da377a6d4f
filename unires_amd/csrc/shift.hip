// shift.hip - k_ata_shift: the whole CG matvec  q = tau AtA p + c DtD p (+ sum p*q)  in ONE streaming
// kernel for observations that are TRANSLATED against the reconstruction grid by a non-integer number
// of voxels and not rotated (the commonest residual misalignment after coregistration), slice profile
// along z.  (Integer translations take aligned.hip's kernel, any rotation the pull / splat pair.)
//
// Without rotation the trilinear weights of grid point (i, j, k) factorise into per-axis weights that
// depend on i, j, k alone, and so does the in-FOV mask (an AND of per-axis tests), so
//     A   = (Mx Tx) (x) (My Ty) (x) (K S Mz Tz)         T: 2-tap interpolation rows, M: 0 / 1 masks,
//     AtA = (Tx' Mx Tx) (x) (Ty' My Ty) (x) (Tz' Mz K' S^2 K Mz Tz)   K: the strided slice-profile conv
// i.e. a tridiagonal blend along x, one along y, and a banded operator along every z line
// (unires/_project.py:161-179 composes the same factors voxel by voxel through a dense grid).  The host
// composes the three factors ONCE per operator, in double precision, from the float32 coordinates the
// kernels (and the reference's float32 grid) would compute - fl(i + t), its floor, its fraction - so
// the per-index rounding of the weights is the reference's:
//     cx[x][3], cy[y][3] : rows of Tx' Mx Tx and Ty' My Ty (zero outside the volume)
//     f[kk][NF]          : x-space voxel kk  = sum_s f[kk][s] B[s0 kk + oz + s]   (K S^2 Mz Tz, NF = taps + 1)
//     e[z][3], kb[z]     : output voxel z   += sum_{m<3} e[z][m] xs[kb[z] + m]    (Tz' Mz K')
// and the kernel is aligned.hip's 16-byte line kernel with two more steps: the line that goes through the
// z operator is the 3 x 3 blend B of nine neighbouring lines (eight of them L2 hits: every line is
// somebody's centre), and the z operator reads its taps per x-space voxel from a table.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "shift.hpp"

namespace unires {

constexpr int kShiftMaxTaps = 16;  // taps of f (profile taps + 1)
constexpr int kShiftLines = kBlock / kWave;

struct ShiftArgs {
  const float *p;
  float *q;
  const float *objb;
  double *partials;
  Dim3i dd;
  int xdz, nf, s, oz;      // x-space z length, taps of f, stride, first B index of x-space voxel 0
  const float *cx, *cy;    // [nx][4], [ny][4]  {minus, centre, plus, -}
  const float *f;          // [xdz][kShiftMaxTaps]
  const float4 *e;         // [nz] {bits(kb), e0, e1, e2}
  float tau, a0, sx, sy, sz;  // sx.. = c / vx^2 of the stencil term
  int padl, padr, wave_floats;
};

typedef float sf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float s4_lower(float v) {  // lane l gets lane l - 1's value (lane 0: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float s4_upper(float v) {  // lane l gets lane l + 1's value (lane 63: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// nz % 4 == 0, nz <= 256: one 16-byte vector per lane and line
template <bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock) k_ata_shift(ShiftArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ __align__(16) float smem[];
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  float *fl = smem;                                            // f: xdz x kShiftMaxTaps, shared by the workgroup
  float *buf = smem + A.xdz * kShiftMaxTaps + w * A.wave_floats;  // (wave_floats, padl: multiples of 4)
  for (int i = threadIdx.y * kWave + lane; i < A.xdz * kShiftMaxTaps; i += kBlock) fl[i] = A.f[i];
  float *pl = buf + A.padl;               // the blended line with zero aprons
  float *xs = pl + nz + A.padr;           // x-space line + two zero pads
  for (int i = lane; i < A.wave_floats; i += kWave) buf[i] = 0.f;
  __syncthreads();
  // lane constants: the conv_up triple of each of the lane's four voxels
  float e0[4], e1[4], e2[4];
  int kb[4];
  const int z0 = 4 * (int)lane;
  const bool in = z0 < nz;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float4 t = A.e[in ? z0 + e : 0];
    kb[e] = __float_as_int(t.x), e0[e] = in ? t.y : 0.f, e1[e] = in ? t.z : 0.f, e2[e] = in ? t.w : 0.f;
  }
  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int nlines = dd.x * dd.y;
  const size_t sxl = (size_t)dd.y * nz, syl = nz;
  double dot = 0.0;
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);  // neighbouring lines stay in one L2
  const int line_step = gridDim.x * kShiftLines;
  const int zc = in ? z0 : 0;
  for (int line = lb * kShiftLines + w; line < nlines; line += line_step) {
    const int vx = line / dd.y, vy = line - vx * dd.y;
    const size_t base = (size_t)line * nz;
    const bool hx = vx + 1 < dd.x, lx = vx > 0, hy = vy + 1 < dd.y, ly = vy > 0;
    // the nine lines of the blend (a missing neighbour reads the centre line; its coefficient is 0)
    const float *pc = p + base + zc;
    const long long ox[3] = {lx ? -(long long)sxl : 0, 0, hx ? (long long)sxl : 0};
    const long long oy[3] = {ly ? -(long long)syl : 0, 0, hy ? (long long)syl : 0};
    sf4 v[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) v[a][b] = *reinterpret_cast<const sf4 *>(pc + ox[a] + oy[b]);
    sf4 rb = {0.f, 0.f, 0.f, 0.f};
    if (OBJ) rb = *reinterpret_cast<const sf4 *>(A.objb + base + zc);
    const float4 cxv = *reinterpret_cast<const float4 *>(A.cx + 4 * vx);
    const float4 cyv = *reinterpret_cast<const float4 *>(A.cy + 4 * vy);
    const float cxa[3] = {cxv.x, cxv.y, cxv.z}, cya[3] = {cyv.x, cyv.y, cyv.z};
    const sf4 zero = {0.f, 0.f, 0.f, 0.f};
    sf4 B = zero;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      sf4 row = cya[0] * v[a][0] + cya[1] * v[a][1] + cya[2] * v[a][2];
      B += cxa[a] * row;
    }
    sf4 rc = v[1][1];
    if (!in) rc = zero, B = zero;
    if (in) *reinterpret_cast<sf4 *>(pl + z0) = B;
    asm volatile("" ::: "memory");  // single wave: LDS ops execute in order
    for (int k0 = 0; k0 < A.xdz; k0 += kWave) {
      const int k = k0 + (int)lane;
      if (k < A.xdz) {
        const float *bin = pl + (k * A.s + A.oz);  // aprons: no bounds checks on the taps
        const float *fk = fl + k * kShiftMaxTaps;
        float acc = 0.f;
        for (int t = 0; t < A.nf; ++t) acc = fmaf(fk[t], bin[t], acc);
        xs[k] = acc;
      }
    }
    asm volatile("" ::: "memory");
    // z neighbours of the stencil across lanes; the line's first voxel has no backward term (zlo := c)
    float zlo = s4_lower(rc.w), zhi = s4_upper(rc.x);
    zlo = lane == 0 ? rc.x : zlo;
    const float c4[4] = {rc.x, rc.y, rc.z, rc.w};
    const float zm4[4] = {zlo, rc.x, rc.y, rc.z}, zp4[4] = {rc.y, rc.z, rc.w, zhi};
    const float xp4[4] = {v[2][1].x, v[2][1].y, v[2][1].z, v[2][1].w}, xm4[4] = {v[0][1].x, v[0][1].y, v[0][1].z, v[0][1].w};
    const float yp4[4] = {v[1][2].x, v[1][2].y, v[1][2].z, v[1][2].w}, ym4[4] = {v[1][0].x, v[1][0].y, v[1][0].z, v[1][0].w};
    const float rb4[4] = {rb.x, rb.y, rb.z, rb.w};
    float o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = c4[e];
      const float *xo = xs + kb[e];
      const float h = e0[e] * xo[0] + e1[e] * xo[1] + e2[e] * xo[2];
      const float xf = (hx ? xp4[e] : 0.f) - c, xb = lx ? c - xm4[e] : 0.f;
      const float yf = (hy ? yp4[e] : 0.f) - c, yb = ly ? c - ym4[e] : 0.f;
      const float zf = zp4[e] - c, zb = c - zm4[e];
      o4[e] = A.tau * h + A.a0 * c + (A.sx * (xb - xf) + A.sy * (yb - yf) + A.sz * (zb - zf));
    }
    asm volatile("" ::: "memory");  // the line buffers are reused by the next line
    if (in) {
      if (OBJ) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += (double)obj_term(o4[e], rb4[e], c4[e]);
      } else {
        __builtin_nontemporal_store(sf4{o4[0], o4[1], o4[2], o4[3]}, reinterpret_cast<sf4 *>(q + base + z0));
        if (DOT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dot += (double)__fmul_rn(c4[e], o4[e]);
        }
      }
    }
  }
  if (DOT) {
    const double tot = block_sum(dot);
    if (threadIdx.x == 0 && threadIdx.y == 0) A.partials[blockIdx.x] = tot;
  }
}

// Two y-adjacent lines per trip (ny even): twelve loads serve two lines instead of eighteen, twice the
// bytes in flight per wave, one store drain per pair (see k_ata_aligned4x2).
template <bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock) k_ata_shift2(ShiftArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ __align__(16) float smem[];
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  float *fl = smem;
  float *buf = smem + A.xdz * kShiftMaxTaps + w * 2 * A.wave_floats;
  for (int i = threadIdx.y * kWave + lane; i < A.xdz * kShiftMaxTaps; i += kBlock) fl[i] = A.f[i];
  float *pl[2] = {buf + A.padl, buf + A.wave_floats + A.padl};
  float *xs[2] = {pl[0] + nz + A.padr, pl[1] + nz + A.padr};
  for (int i = lane; i < 2 * A.wave_floats; i += kWave) buf[i] = 0.f;
  __syncthreads();
  float e0[4], e1[4], e2[4];
  int kb[4];
  const int z0 = 4 * (int)lane;
  const bool in = z0 < nz;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float4 t = A.e[in ? z0 + e : 0];
    kb[e] = __float_as_int(t.x), e0[e] = in ? t.y : 0.f, e1[e] = in ? t.z : 0.f, e2[e] = in ? t.w : 0.f;
  }
  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int hy2 = dd.y / 2, npairs = dd.x * hy2;
  const size_t sxl = (size_t)dd.y * nz, syl = nz;
  double dot = 0.0;
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);
  const int pair_step = gridDim.x * kShiftLines;
  const int zc = in ? z0 : 0;
  const sf4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int pr = lb * kShiftLines + w; pr < npairs; pr += pair_step) {
    const int vx = pr / hy2, vy0 = 2 * (pr - vx * hy2);
    const size_t base = ((size_t)vx * dd.y + vy0) * nz;
    const bool hx = vx + 1 < dd.x, lx = vx > 0, ly = vy0 > 0, hy = vy0 + 2 < dd.y;
    const float *pc = p + base + zc;
    const long long ox[3] = {lx ? -(long long)sxl : 0, 0, hx ? (long long)sxl : 0};
    // four rows of y: vy0 - 1, vy0, vy0 + 1, vy0 + 2 (missing ones read line vy0; their coefficients are 0)
    const long long oy[4] = {ly ? -(long long)syl : 0, 0, (long long)syl, hy ? 2 * (long long)syl : 0};
    sf4 v[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) v[a][b] = *reinterpret_cast<const sf4 *>(pc + ox[a] + oy[b]);
    sf4 rb[2] = {zero, zero};
    if (OBJ) {
      rb[0] = *reinterpret_cast<const sf4 *>(A.objb + base + zc);
      rb[1] = *reinterpret_cast<const sf4 *>(A.objb + base + syl + zc);
    }
    const float4 cxv = *reinterpret_cast<const float4 *>(A.cx + 4 * vx);
    const float4 cy0 = *reinterpret_cast<const float4 *>(A.cy + 4 * vy0);
    const float4 cy1 = *reinterpret_cast<const float4 *>(A.cy + 4 * (vy0 + 1));
    const float cxa[3] = {cxv.x, cxv.y, cxv.z};
    sf4 B[2] = {zero, zero};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      B[0] += cxa[a] * (cy0.x * v[a][0] + cy0.y * v[a][1] + cy0.z * v[a][2]);
      B[1] += cxa[a] * (cy1.x * v[a][1] + cy1.y * v[a][2] + cy1.z * v[a][3]);
    }
    sf4 rc[2] = {v[1][1], v[1][2]};
    if (!in) rc[0] = rc[1] = B[0] = B[1] = zero;
    if (in) *reinterpret_cast<sf4 *>(pl[0] + z0) = B[0], *reinterpret_cast<sf4 *>(pl[1] + z0) = B[1];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int l = 0; l < 2; ++l)
      for (int k0 = 0; k0 < A.xdz; k0 += kWave) {
        const int k = k0 + (int)lane;
        if (k < A.xdz) {
          const float *bin = pl[l] + (k * A.s + A.oz);
          const float *fk = fl + k * kShiftMaxTaps;
          float acc = 0.f;
          for (int t = 0; t < A.nf; ++t) acc = fmaf(fk[t], bin[t], acc);
          xs[l][k] = acc;
        }
      }
    asm volatile("" ::: "memory");
    sf4 out[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      float zlo = s4_lower(rc[l].w), zhi = s4_upper(rc[l].x);
      zlo = lane == 0 ? rc[l].x : zlo;
      const float c4[4] = {rc[l].x, rc[l].y, rc[l].z, rc[l].w};
      const float zm4[4] = {zlo, rc[l].x, rc[l].y, rc[l].z}, zp4[4] = {rc[l].y, rc[l].z, rc[l].w, zhi};
      const sf4 xp = v[2][1 + l], xm = v[0][1 + l], yp = v[1][2 + l], ym = v[1][l];
      const float xp4[4] = {xp.x, xp.y, xp.z, xp.w}, xm4[4] = {xm.x, xm.y, xm.z, xm.w};
      const float yp4[4] = {yp.x, yp.y, yp.z, yp.w}, ym4[4] = {ym.x, ym.y, ym.z, ym.w};
      const bool lyl = l == 0 ? ly : true, hyl = l == 0 ? true : hy;
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = c4[e];
        const float *xo = xs[l] + kb[e];
        const float h = e0[e] * xo[0] + e1[e] * xo[1] + e2[e] * xo[2];
        const float xf = (hx ? xp4[e] : 0.f) - c, xb = lx ? c - xm4[e] : 0.f;
        const float yf = (hyl ? yp4[e] : 0.f) - c, yb = lyl ? c - ym4[e] : 0.f;
        const float zf = zp4[e] - c, zb = c - zm4[e];
        o4[e] = A.tau * h + A.a0 * c + (A.sx * (xb - xf) + A.sy * (yb - yf) + A.sz * (zb - zf));
      }
      out[l] = sf4{o4[0], o4[1], o4[2], o4[3]};
    }
    asm volatile("" ::: "memory");
    if (in) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        if (OBJ) {
          dot += (double)obj_term(out[l].x, rb[l].x, rc[l].x) + (double)obj_term(out[l].y, rb[l].y, rc[l].y) +
                 (double)obj_term(out[l].z, rb[l].z, rc[l].z) + (double)obj_term(out[l].w, rb[l].w, rc[l].w);
        } else {
          __builtin_nontemporal_store(out[l], reinterpret_cast<sf4 *>(q + base + l * syl + z0));
          if (DOT)
            dot += (double)__fmul_rn(rc[l].x, out[l].x) + (double)__fmul_rn(rc[l].y, out[l].y) +
                   (double)__fmul_rn(rc[l].z, out[l].z) + (double)__fmul_rn(rc[l].w, out[l].w);
        }
      }
    }
  }
  if (DOT) {
    const double tot = block_sum(dot);
    if (threadIdx.x == 0 && threadIdx.y == 0) A.partials[blockIdx.x] = tot;
  }
}

// --------------------------------------------------------------------------
// host: the three factors of AtA
// --------------------------------------------------------------------------
void shift_free(ShiftPlan &S) {
  if (S.dev) (void)hipFree(S.dev);
  S = ShiftPlan();
}

int shift_blocks(Dim3i dd) {
  const long long nb = ((long long)dd.x * dd.y + kShiftLines - 1) / kShiftLines;
  return (int)(nb < 4096 ? nb : 4096);
}

// one axis: per grid index i the float32 coordinate fl(i + t) (what affine_point computes for an identity
// linear part), its mask, floor and weights -> rows of T with the mask folded in
struct AxisRows {
  std::vector<int> b;          // floor
  std::vector<double> w0, w1;  // weights on b and b + 1 (0 where masked / out of the volume)
};
static AxisRows axis_rows(int gn, int n, float t, float tol) {
  AxisRows R;
  R.b.resize(gn), R.w0.resize(gn), R.w1.resize(gn);
  for (int i = 0; i < gn; ++i) {
    const float c = (float)i + t;  // fmaf(0, k, fmaf(0, j, 1 * i)) + t, bit for bit
    const float fl = floorf(c);
    const float w1 = c - fl, w0 = 1.f - w1;
    const bool m = c > -tol && c < (float)(n - 1) + tol;
    const int b = (int)fl;
    R.b[i] = b;
    R.w0[i] = (m && b >= 0 && b < n) ? (double)w0 : 0.0;
    R.w1[i] = (m && b + 1 >= 0 && b + 1 < n) ? (double)w1 : 0.0;
  }
  return R;
}

// Builds the tables of an operator; non-zero: outside the kernel's domain (plan left invalid).
int shift_build(ShiftPlan &S, Dim3i dd, Dim3i gd, Dim3i xd, const Taps &T, const Scaling &S2, const Affine &A,
                float tol) {
  S.valid = false;
  static const bool off = getenv("UNIRES_NO_SHIFT") != nullptr;
  if (off) return 1;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      if (A.m[4 * r + c] != (r == c ? 1.f : 0.f)) return 1;
  const float t[3] = {A.m[3], A.m[7], A.m[11]};
  for (int d = 0; d < 3; ++d)
    if (!(fabsf(t[d]) < 1e5f)) return 1;
  if (t[0] == floorf(t[0]) && t[1] == floorf(t[1]) && t[2] == floorf(t[2])) return 1;  // aligned.hip's case
  for (int d = 0; d < 2; ++d)
    if (T.n[d] != 1 || T.s[d] != 1 || T.t[d][0] != 1.f) return 1;
  if (S2.dim >= 0 && S2.dim != 2) return 1;
  if (dd.z % 4 != 0 || dd.z > 4 * kWave || dd.z < 8) return 1;
  const int nk = T.n[2], s = T.s[2];
  if (nk + 1 > kShiftMaxTaps || s < 1 || s > nk) return 1;
  if (gd.z != (xd.z - 1) * s + nk || gd.x != xd.x || gd.y != xd.y) return 1;
  const AxisRows X = axis_rows(gd.x, dd.x, t[0], tol), Y = axis_rows(gd.y, dd.y, t[1], tol),
                 Z = axis_rows(gd.z, dd.z, t[2], tol);
  // x / y: rows of T' T; the band must be tridiagonal (floor(i + t) - i constant)
  auto band = [&](const AxisRows &R, int gn, int n, std::vector<float> &out) {
    std::vector<double> c((size_t)n * 3, 0.0);
    for (int i = 0; i < gn; ++i) {
      const int b = R.b[i];
      const double w[2] = {R.w0[i], R.w1[i]};
      for (int u = 0; u < 2; ++u)
        for (int v = 0; v < 2; ++v) {
          if (w[u] == 0.0 || w[v] == 0.0) continue;
          const int x = b + u, d = v - u;  // entry (x, x + d)
          if (x < 0 || x >= n) return false;
          c[(size_t)x * 3 + d + 1] += w[u] * w[v];
        }
    }
    out.assign((size_t)n * 4, 0.f);
    for (int x = 0; x < n; ++x)
      for (int d = 0; d < 3; ++d) out[(size_t)x * 4 + d] = (float)c[(size_t)x * 3 + d];
    return true;
  };
  std::vector<float> cx, cy;
  if (!band(X, gd.x, dd.x, cx) || !band(Y, gd.y, dd.y, cy)) return 1;
  // z: constant offset between grid index and floor
  const int oz = Z.b[0];
  for (int k = 0; k < gd.z; ++k)
    if (Z.b[k] != k + oz) return 1;
  const int nf = nk + 1;
  std::vector<float> f((size_t)xd.z * kShiftMaxTaps, 0.f);
  std::vector<double> fd((size_t)xd.z * kShiftMaxTaps, 0.0);
  for (int kk = 0; kk < xd.z; ++kk) {
    const double sc = S2.dim == 2 ? ((kk & 1) ? (double)S2.o : (double)S2.e) : 1.0;
    for (int tt = 0; tt < nk; ++tt) {
      const int k = kk * s + tt;
      fd[(size_t)kk * kShiftMaxTaps + tt] += sc * (double)T.t[2][tt] * Z.w0[k];
      fd[(size_t)kk * kShiftMaxTaps + tt + 1] += sc * (double)T.t[2][tt] * Z.w1[k];
    }
    for (int u = 0; u < nf; ++u) f[(size_t)kk * kShiftMaxTaps + u] = (float)fd[(size_t)kk * kShiftMaxTaps + u];
  }
  // output z: sum over x-space voxels, E[z][kk] = sum_k Tz[k][z] kz[k - s kk]
  std::vector<float> e((size_t)dd.z * 4, 0.f);
  std::vector<double> row((size_t)xd.z);
  for (int z = 0; z < dd.z; ++z) {
    std::fill(row.begin(), row.end(), 0.0);
    for (int u = 0; u < 2; ++u) {
      const int k = z - oz - u;  // grid points whose corner b + u is z
      if (k < 0 || k >= gd.z) continue;
      const double wz = u ? Z.w1[k] : Z.w0[k];
      if (wz == 0.0) continue;
      for (int kk = 0; kk < xd.z; ++kk) {
        const int tt = k - s * kk;
        if (tt >= 0 && tt < nk) row[kk] += wz * (double)T.t[2][tt];
      }
    }
    int first = -1, last = -1;
    for (int kk = 0; kk < xd.z; ++kk)
      if (row[kk] != 0.0) {
        if (first < 0) first = kk;
        last = kk;
      }
    if (first < 0) first = last = 0;
    if (last - first > 2) return 1;  // conv_up fan-in beyond the three entries a voxel reads
    memcpy(&e[(size_t)z * 4], &first, 4);
    for (int m = 0; m < 3; ++m) e[(size_t)z * 4 + 1 + m] = first + m < xd.z ? (float)row[first + m] : 0.f;
  }
  // aprons of the blended line: x-space voxel kk reads B[s kk + oz, s kk + oz + nf)
  const int lo = oz, hi = (xd.z - 1) * s + oz + nf - 1;
  int padl = lo < 0 ? -lo : 0, padr = hi >= dd.z ? hi - dd.z + 1 : 0;
  padl = (padl + 3) & ~3;
  if (padl > 256 || padr > 256) return 1;
  const int wave_floats = (padl + dd.z + padr + xd.z + 2 + 3) & ~3;
  if (((size_t)xd.z * kShiftMaxTaps + (size_t)kShiftLines * wave_floats) * sizeof(float) > 60 * 1024) return 1;
  const size_t n_cx = cx.size(), n_cy = cy.size(), n_f = f.size(), n_e = e.size();
  const size_t total = n_cx + n_cy + n_f + n_e;
  if (total > S.cap) {
    if (S.dev) (void)hipFree(S.dev);
    S.dev = nullptr;
    if (hipMalloc((void **)&S.dev, total * sizeof(float)) != hipSuccess) return 1;
    S.cap = total;
  }
  std::vector<float> all;
  all.reserve(total);
  all.insert(all.end(), e.begin(), e.end());  // (16-byte aligned tables first)
  all.insert(all.end(), cx.begin(), cx.end());
  all.insert(all.end(), cy.begin(), cy.end());
  all.insert(all.end(), f.begin(), f.end());
  if (hipMemcpy(S.dev, all.data(), total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
  S.o_e = 0, S.o_cx = n_e, S.o_cy = n_e + n_cx, S.o_f = n_e + n_cx + n_cy;
  S.nf = nf, S.s = s, S.oz = oz, S.padl = padl, S.padr = padr, S.wave_floats = wave_floats;
  S.dd = dd, S.xdz = xd.z;
  memcpy(S.key, A.m, sizeof(A.m));
  S.valid = true;
  return 0;
}

int launch_ata_shift(const ShiftPlan &S, const float *p, float *q, Dim3i dd, const Affine &A, float tau, float a0,
                     float cx, float cy, float cz, double *partials, const float *objb, const int *done,
                     hipStream_t st) {
  if (!S.valid || memcmp(S.key, A.m, sizeof(A.m)) != 0) return 1;
  if (dd.x != S.dd.x || dd.y != S.dd.y || dd.z != S.dd.z) return 1;
  if (objb && !partials) return 1;
  const uintptr_t al = (uintptr_t)p | (uintptr_t)q | (uintptr_t)(objb ? objb : p);
  if (al & 15u) return 1;
  ShiftArgs G;
  G.p = p, G.q = q, G.objb = objb, G.partials = partials, G.dd = dd;
  G.xdz = S.xdz, G.nf = S.nf, G.s = S.s, G.oz = S.oz;
  G.e = reinterpret_cast<const float4 *>(S.dev + S.o_e);
  G.cx = S.dev + S.o_cx, G.cy = S.dev + S.o_cy, G.f = S.dev + S.o_f;
  G.tau = tau, G.a0 = a0, G.sx = cx, G.sy = cy, G.sz = cz;
  G.padl = S.padl, G.padr = S.padr, G.wave_floats = S.wave_floats;
  const size_t lds = ((size_t)S.xdz * kShiftMaxTaps + (size_t)kShiftLines * S.wave_floats) * sizeof(float);
  const dim3 grid(shift_blocks(dd)), block(kWave, kShiftLines);
  static const bool no_x2 = getenv("UNIRES_SHIFT_X2") && getenv("UNIRES_SHIFT_X2")[0] == '0';
  const size_t lds2 = ((size_t)S.xdz * kShiftMaxTaps + (size_t)2 * kShiftLines * S.wave_floats) * sizeof(float);
  if (!no_x2 && dd.y % 2 == 0 && lds2 <= 64 * 1024) {
    if (objb)
      hipLaunchKernelGGL((k_ata_shift2<true, true>), grid, block, lds2, st, G, done);
    else if (partials)
      hipLaunchKernelGGL((k_ata_shift2<true, false>), grid, block, lds2, st, G, done);
    else
      hipLaunchKernelGGL((k_ata_shift2<false, false>), grid, block, lds2, st, G, done);
    return 0;
  }
  if (objb)
    hipLaunchKernelGGL((k_ata_shift<true, true>), grid, block, lds, st, G, done);
  else if (partials)
    hipLaunchKernelGGL((k_ata_shift<true, false>), grid, block, lds, st, G, done);
  else
    hipLaunchKernelGGL((k_ata_shift<false, false>), grid, block, lds, st, G, done);
  return 0;
}

}  // namespace unires
