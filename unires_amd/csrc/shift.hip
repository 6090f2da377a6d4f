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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
  int xr;  // x-marching form: planes per run
  const float4 *e4;  // lane-window rows: [nz] weights on xs[kmin[lane] .. + 3]
  const int *kmin;   // [64]
  int lw;            // 1: lane-window form of the z operator (xdz <= 64, four-entry windows)
  int defer;         // 1: a step's stores are issued behind the next step's loads
  int dbg;           // measurement only (UNIRES_SHIFT_DBG): 1 no z operator, 2 no stores, 8 plain stores
  unsigned long long *prof;  // -DUNIRES_SHIFT_PROF builds: per wave {steps, total, load wait, z operator, stencil + store} clocks
};

typedef float sf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float s4_lower(float v) {  // lane l gets lane l - 1's value (lane 0: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float s4_upper(float v) {  // lane l gets lane l + 1's value (lane 63: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// One x-space voxel of the z operator: sum_t f[t] B[t], taps four at a time with their eight LDS reads in
// flight together (rows of the table beyond nf are zero; the reads they pair with stay inside the wave's
// buffer: pl is followed by the x-space line).  Same products in the same order as a tap-by-tap loop.
__device__ __forceinline__ float zline_taps(const float *bin, const float *fk, int xdz, int nf) {
  float acc = 0.f;
  for (int t0 = 0; t0 < nf; t0 += 4) {
    const float f0 = fk[t0 * xdz], f1 = fk[(t0 + 1) * xdz], f2 = fk[(t0 + 2) * xdz], f3 = fk[(t0 + 3) * xdz];
    const float b0 = bin[t0], b1 = bin[t0 + 1], b2 = bin[t0 + 2], b3 = bin[t0 + 3];
    acc = fmaf(f3, b3, fmaf(f2, b2, fmaf(f1, b1, fmaf(f0, b0, acc))));
  }
  return acc;
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
  // the taps of x-space voxel k in LDS as fl[t * xdz + k]: lane = k reads consecutive words (the [k][t] layout
  // of the host table put sixteen lanes on one bank)
  for (int i = threadIdx.y * kWave + lane; i < A.xdz * kShiftMaxTaps; i += kBlock) {
    const int k = i / kShiftMaxTaps, t = i - k * kShiftMaxTaps;
    fl[t * A.xdz + k] = A.f[i];
  }
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
        xs[k] = zline_taps(pl + (k * A.s + A.oz), fl + k, A.xdz, A.nf);  // aprons: no bounds checks on the taps
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
  // the taps of x-space voxel k in LDS as fl[t * xdz + k]: lane = k reads consecutive words (the [k][t] layout
  // of the host table put sixteen lanes on one bank)
  for (int i = threadIdx.y * kWave + lane; i < A.xdz * kShiftMaxTaps; i += kBlock) {
    const int k = i / kShiftMaxTaps, t = i - k * kShiftMaxTaps;
    fl[t * A.xdz + k] = A.f[i];
  }
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
          xs[l][k] = zline_taps(pl[l] + (k * A.s + A.oz), fl + k, A.xdz, A.nf);  // aprons: no bounds checks on the taps
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

// x-marching form (round 4).  A wave owns a PAIR of y-adjacent lines over a run of `xr` x planes and walks
// along x: the 3 x 3 blend factorises, B(x) = cx- R(x-1) + cx0 R(x) + cx+ R(x+1) with R(x) = the y blend of
// plane x, so a plane's four lines (the pair and its two y neighbours) are loaded ONCE, reduced to two
// rows R and kept in registers with the pair's centres for the next two planes - four vector loads per
// plane for two output lines (+ two planes of run-in per run) where k_ata_shift2 issues twelve per pair,
// every one of them a different L2 line of another wave's centre.  The next plane's loads are in flight
// while a plane is computed; the z operator and the stencil are k_ata_shift2's.
// NL = lines per wave (1 or 2): with one line a wave's plane state halves (4 waves per SIMD instead of 2); with
// two, a plane costs four loads for two lines instead of three for one.
//
// The instruction stream is what bounds this kernel (a step's loads have long arrived when it looks for them;
// -DUNIRES_SHIFT_PROF timeline, EXPERIMENTS E4), so the step is written for few instructions: four plane slots
// that change ROLES (previous / current / next / in flight) instead of registers being copied, the walk
// unrolled by four; everything in 4-vectors, which the compiler packs into v_pk_* instructions; the
// volume's faces on a path of their own, so that the interior carries no selects; pointers bumped, not
// recomputed; the lane's conv_up rows transposed (one vector per window entry).
// FAST: the common case compiled without its run-time alternatives - lane-window z operator, every lane inside
// the line (nz = 256), no measurement switches: the step then carries a handful of branches instead of ~40
// (exec-mask juggling and wait counts around them were a third of its instructions).
template <int NL, bool DOT, bool OBJ, bool FAST>
__global__ void __launch_bounds__(kBlock) k_ata_shift_m(ShiftArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ __align__(16) float smem[];
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  float *fl = smem;
  float *cxl = smem + A.xdz * kShiftMaxTaps;  // rows of the x blend (a vector load per step would be waited for
                                              // with vmcnt(0): the next plane's loads, just issued, with it)
  float4 *el = reinterpret_cast<float4 *>(cxl + 4 * dd.x);  // conv_up rows per output z
  float *buf = cxl + 4 * dd.x + 4 * nz + w * NL * A.wave_floats;
  // the taps of x-space voxel k in LDS as fl[t * xdz + k]: lane = k reads consecutive words (the [k][t] layout
  // of the host table put sixteen lanes on one bank)
  for (int i = threadIdx.y * kWave + lane; i < A.xdz * kShiftMaxTaps; i += kBlock) {
    const int k = i / kShiftMaxTaps, t = i - k * kShiftMaxTaps;
    fl[t * A.xdz + k] = A.f[i];
  }
  for (int i = threadIdx.y * kWave + lane; i < 4 * dd.x; i += kBlock) cxl[i] = A.cx[i];
  for (int i = threadIdx.y * kWave + lane; i < nz; i += kBlock) el[i] = (FAST || A.lw) ? A.e4[i] : A.e[i];
  float *pl[NL], *xs[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) pl[l] = buf + l * A.wave_floats + A.padl, xs[l] = pl[l] + nz + A.padr;
  for (int i = lane; i < NL * A.wave_floats; i += kWave) buf[i] = 0.f;
  __syncthreads();
  const int z0 = 4 * (int)lane;
  const bool in = FAST ? true : z0 < nz;
  const float4 *elz = el + (in ? z0 : 0);
  const bool lw = FAST ? true : A.lw != 0;
  const int dbg = FAST ? 0 : A.dbg;
  const bool defer = FAST ? false : A.defer != 0;
  // lane-window form: the taps of "its" x-space voxel (k = lane) are lane constants, and so are the start of
  // the x-space window its four outputs read and their rows on it - transposed: tw[j] = the weights of window
  // entry j in the lane's four outputs
  float fr[kShiftMaxTaps];
#pragma unroll
  for (int t = 0; t < kShiftMaxTaps; ++t) fr[t] = (FAST || A.lw) && (int)lane < A.xdz ? A.f[lane * kShiftMaxTaps + t] : 0.f;
  const int kmin = (FAST || A.lw) ? A.kmin[lane] : 0;
  const int bin_off = ((int)lane < A.xdz ? (int)lane : 0) * A.s + A.oz;
  sf4 tw[4];
  {
    const float4 t0 = elz[0], t1 = elz[1], t2 = elz[2], t3 = elz[3];
    tw[0] = sf4{t0.x, t1.x, t2.x, t3.x}, tw[1] = sf4{t0.y, t1.y, t2.y, t3.y};
    tw[2] = sf4{t0.z, t1.z, t2.z, t3.z}, tw[3] = sf4{t0.w, t1.w, t2.w, t3.w};
  }
  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int hyn = dd.y / NL, nxr = (dd.x + A.xr - 1) / A.xr, ntasks = hyn * nxr;
  const size_t sxl = (size_t)dd.y * nz;
  const long long syl = nz;
  double dot = 0.0;
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);
  const int task_step = gridDim.x * kShiftLines;
  const int zc = in ? z0 : 0;
  const sf4 zero = {0.f, 0.f, 0.f, 0.f};
  const bool lane0 = lane == 0;
  struct Plane {
    sf4 v[NL + 2];  // lines vy0 - 1 .. vy0 + NL
    sf4 r[NL];      // their y blends
  };
#ifdef UNIRES_SHIFT_PROF
  unsigned long long pt_steps = 0, pt_total = 0, pt_wait = 0, pt_z = 0, pt_out = 0;
  const unsigned long long pt_begin = __builtin_readcyclecounter();
#define PT_NOW(var) const unsigned long long var = __builtin_readcyclecounter()
#define PT_PIN(x) asm volatile("" ::"v"(x))
#else
#define PT_NOW(var)
#define PT_PIN(x)
#endif
  for (int task = lb * kShiftLines + w; task < ntasks; task += task_step) {
    const int xr = task / hyn, vy0 = NL * (task - xr * hyn);
    const int xa = xr * A.xr, xb = min(xa + A.xr, dd.x);
    const bool ly = vy0 > 0, hy = vy0 + NL < dd.y;
    // rows of y: vy0 - 1 .. vy0 + NL (missing ones read line vy0; their coefficients are 0)
    long long oy[NL + 2];
    oy[0] = ly ? -syl : 0;
#pragma unroll
    for (int b = 0; b < NL; ++b) oy[1 + b] = b * syl;
    oy[NL + 1] = hy ? NL * syl : 0;
    float4 cyv[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) cyv[l] = *reinterpret_cast<const float4 *>(A.cy + 4 * (vy0 + l));
    const float *pcol = p + (size_t)vy0 * nz + zc;
    // the lines of plane vx (a plane outside the volume reads the nearest one; its coefficient is 0)
    auto load_plane = [&](int vx, Plane &P) {
      const float *pp = pcol + (size_t)min(max(vx, 0), dd.x - 1) * sxl;
#pragma unroll
      for (int b = 0; b < NL + 2; ++b) P.v[b] = *reinterpret_cast<const sf4 *>(pp + oy[b]);
    };
    auto rows_of = [&](Plane &P) {
#pragma unroll
      for (int l = 0; l < NL; ++l) P.r[l] = cyv[l].x * P.v[l] + cyv[l].y * P.v[l + 1] + cyv[l].z * P.v[l + 2];
    };
    // one output plane: prev / cur / next hold planes vx - 1, vx, vx + 1 (next's rows not yet formed), fly
    // receives plane vx + 2
    // (UNIRES_SHIFT_DEFER=1, measured and left off: the stores of a step issued at the START of the next one,
    // right behind its loads, as in the copy loop of tools/mb_stream.hip that walks the same planes in 25 us -
    // 39.3 us against 37.1)
    sf4 pend[NL];
    float *pend_q = nullptr;
    auto flush = [&]() {
      if (pend_q && in) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          if (dbg & 8)
            *reinterpret_cast<sf4 *>(pend_q + l * syl) = pend[l];
          else if (!(dbg & 2))
            __builtin_nontemporal_store(pend[l], reinterpret_cast<sf4 *>(pend_q + l * syl));
        }
      }
      pend_q = nullptr;
    };
    auto step = [&](const Plane &prev, const Plane &cur, Plane &next, Plane &fly, int vx) {
      const size_t base = ((size_t)vx * dd.y + vy0) * nz;
      load_plane(vx + 2 < xb + 1 ? vx + 2 : vx + 1, fly);  // in flight over this step
      flush();
      sf4 rb[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) rb[l] = OBJ ? *reinterpret_cast<const sf4 *>(A.objb + base + l * syl + zc) : zero;
      const float4 cxv = *reinterpret_cast<const float4 *>(cxl + 4 * vx);
      PT_NOW(pt0);
      rows_of(next);
      PT_PIN(next.r[0].x);
      PT_PIN(next.r[NL - 1].w);
      PT_NOW(pt1);
      sf4 rc[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        sf4 B = cxv.x * prev.r[l] + cxv.y * cur.r[l] + cxv.z * next.r[l];
        rc[l] = cur.v[1 + l];
        if (!in) rc[l] = B = zero;
        if (in) *reinterpret_cast<sf4 *>(pl[l] + z0) = B;
      }
      asm volatile("" ::: "memory");
      if (dbg & 1) {
      } else if (lw) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const float *bin = pl[l] + bin_off;
          float acc = 0.f;
#pragma unroll
          for (int t0 = 0; t0 < kShiftMaxTaps; t0 += 4)
            if (t0 < A.nf) {
              const float b0 = bin[t0], b1 = bin[t0 + 1], b2 = bin[t0 + 2], b3 = bin[t0 + 3];
              acc = fmaf(fr[t0 + 3], b3, fmaf(fr[t0 + 2], b2, fmaf(fr[t0 + 1], b1, fmaf(fr[t0], b0, acc))));
            }
          if ((int)lane < A.xdz) xs[l][lane] = acc;
        }
      } else {
#pragma unroll
        for (int l = 0; l < NL; ++l)
          for (int k0 = 0; k0 < A.xdz; k0 += kWave) {
            const int k = k0 + (int)lane;
            if (k < A.xdz)
              xs[l][k] = zline_taps(pl[l] + (k * A.s + A.oz), fl + k, A.xdz, A.nf);  // aprons: no bounds checks on the taps
          }
      }
      asm volatile("" ::: "memory");
#ifdef UNIRES_SHIFT_PROF
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      PT_NOW(pt2);
      const bool hx = vx + 1 < dd.x, lx = vx > 0;
      sf4 out[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const sf4 c = rc[l];
        // conv_up
        sf4 h;
        if (lw && !(dbg & 1)) {  // the lane's x-space window, read once for its four outputs
          const float *xo = xs[l] + kmin;
          h = xo[0] * tw[0] + xo[1] * tw[1] + xo[2] * tw[2] + xo[3] * tw[3];
        } else {
          float h4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float4 te = elz[e];
            const float *xo = xs[l] + __float_as_int(te.x);
            h4[e] = te.y * xo[0] + te.z * xo[1] + te.w * xo[2];
          }
          h = sf4{h4[0], h4[1], h4[2], h4[3]};
        }
        // z neighbours across lanes; the line's first voxel has no backward term (zlo := c), its last no
        // forward neighbour (a zero comes in from beyond lane 63 / the zero vector of the lanes past nz)
        float zlo = s4_lower(c.w);
        const float zhi = s4_upper(c.x);
        zlo = lane0 ? c.x : zlo;
        const sf4 zm = {zlo, c.x, c.y, c.z}, zp = {c.y, c.z, c.w, zhi};
        const sf4 xp = next.v[1 + l], xm = prev.v[1 + l], yp = cur.v[2 + l], ym = cur.v[l];
        const bool lyl = l == 0 ? ly : true, hyl = l == NL - 1 ? hy : true;
        sf4 dx, dy;
        if (hx && lx && lyl && hyl) {  // interior of the volume in x and y (wave-uniform)
          dx = (c - xm) - (xp - c), dy = (c - ym) - (yp - c);
        } else {
          const sf4 xf = (hx ? xp : zero) - c, xbk = lx ? c - xm : zero;
          const sf4 yf = (hyl ? yp : zero) - c, ybk = lyl ? c - ym : zero;
          dx = xbk - xf, dy = ybk - yf;
        }
        const sf4 dz = (c - zm) - (zp - c);
        out[l] = A.tau * h + A.a0 * c + (A.sx * dx + A.sy * dy + A.sz * dz);
      }
      asm volatile("" ::: "memory");
      if (in) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          if (OBJ) {
            dot += (double)obj_term(out[l].x, rb[l].x, rc[l].x) + (double)obj_term(out[l].y, rb[l].y, rc[l].y) +
                   (double)obj_term(out[l].z, rb[l].z, rc[l].z) + (double)obj_term(out[l].w, rb[l].w, rc[l].w);
          } else {
            if (defer) {
              pend[l] = out[l];
              pend_q = q + base + z0;
            } else if (dbg & 8) {
              *reinterpret_cast<sf4 *>(q + base + l * syl + z0) = out[l];
            } else if (!(dbg & 2)) {
              __builtin_nontemporal_store(out[l], reinterpret_cast<sf4 *>(q + base + l * syl + z0));
            }
            if (DOT)
              dot += ((double)__fmul_rn(rc[l].x, out[l].x) + (double)__fmul_rn(rc[l].y, out[l].y)) +
                     ((double)__fmul_rn(rc[l].z, out[l].z) + (double)__fmul_rn(rc[l].w, out[l].w));
          }
        }
      }
#ifdef UNIRES_SHIFT_PROF
      {
        PT_NOW(pt3);
        pt_steps += 1, pt_wait += pt1 - pt0, pt_z += pt2 - pt1, pt_out += pt3 - pt2;
      }
#endif
    };
    Plane P0, P1, P2, P3;
    load_plane(xa - 1, P0);
    load_plane(xa, P1);
    load_plane(xa + 1, P2);
    rows_of(P0);
    rows_of(P1);
    for (int vx = xa; vx < xb; vx += 4) {  // the slots change roles: no register moves
      step(P0, P1, P2, P3, vx);
      if (vx + 1 < xb) step(P1, P2, P3, P0, vx + 1);
      if (vx + 2 < xb) step(P2, P3, P0, P1, vx + 2);
      if (vx + 3 < xb) step(P3, P0, P1, P2, vx + 3);
      else break;
    }
    flush();
  }
#ifdef UNIRES_SHIFT_PROF
  pt_total = __builtin_readcyclecounter() - pt_begin;
  if (A.prof && lane == 0) {
    unsigned long long *o = A.prof + ((size_t)blockIdx.x * kShiftLines + w) * 8;
    o[0] = pt_steps, o[1] = pt_total, o[2] = pt_wait, o[3] = pt_z, o[4] = pt_out, o[5] = pt_begin;
  }
#endif
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

// x-marching form: planes per run - as long as the chip still gets ~UNIRES_SHIFT_TASKS wave tasks (0: form off)
static int shift_nl(Dim3i dd) {  // lines per wave of the marching form
  static const int nl = getenv("UNIRES_SHIFT_NL") ? atoi(getenv("UNIRES_SHIFT_NL")) : 2;
  return nl == 2 && dd.y % 2 == 0 ? 2 : 1;
}

static int shift_xr(Dim3i dd) {
  static const int march = getenv("UNIRES_SHIFT_MARCH") ? atoi(getenv("UNIRES_SHIFT_MARCH")) : -1;  // 0: off, n: xr = n
  // tasks = the waves the chip holds at once: 2 per SIMD with two lines per wave (233 registers), 3 with one
  static const int tasks_env = getenv("UNIRES_SHIFT_TASKS") ? atoi(getenv("UNIRES_SHIFT_TASKS")) : 0;
  if (march == 0 || dd.x > 1024) return 0;  // (the x-blend table sits in LDS)
  const int tasks = tasks_env > 0 ? tasks_env : (shift_nl(dd) == 2 ? 2048 : 3072);
  const long long pairs = dd.y / shift_nl(dd);
  long long xr = march > 0 ? march : ((long long)dd.x * pairs + tasks - 1) / tasks;
  xr = std::max<long long>(4, std::min<long long>(xr, dd.x));
  // runs that start a multiple of 1 MB apart keep the concurrently walked planes on the same DRAM banks
  // (256^3: runs of 16 planes, 4 MB apart: 40.3 us; of 17: 37.7 us, same number of runs)
  const long long run_bytes = xr * (long long)dd.y * dd.z * 4;
  if (march <= 0 && xr < dd.x && run_bytes % (1 << 20) == 0 && (dd.x + xr) / (xr + 1) == (dd.x + xr - 1) / xr) ++xr;
  return (int)xr;
}

// true: the x-marching kernel's fast form serves this operator (lane-window z operator, 256-voxel lines)
bool shift_fast(const ShiftPlan &S, Dim3i dd) {
  static const bool no_fast = getenv("UNIRES_SHIFT_FAST") && getenv("UNIRES_SHIFT_FAST")[0] == '0';
  static const bool no_lw = getenv("UNIRES_SHIFT_LW") && getenv("UNIRES_SHIFT_LW")[0] == '0';
  const size_t lds_m = ((size_t)S.xdz * kShiftMaxTaps + (size_t)shift_nl(dd) * kShiftLines * S.wave_floats + (size_t)4 * (dd.x + dd.z)) *
                       sizeof(float);
  return S.valid && !no_fast && !no_lw && S.lane_window && dd.z == 4 * kWave && shift_xr(dd) > 0 && lds_m <= 64 * 1024;
}

int shift_blocks(Dim3i dd) {
  long long nb = ((long long)dd.x * dd.y + kShiftLines - 1) / kShiftLines;
  if (const int xr = shift_xr(dd))  // every workgroup of the marching form has tasks (its XCD chunks are of tasks)
    nb = ((long long)(dd.y / shift_nl(dd)) * ((dd.x + xr - 1) / xr) + kShiftLines - 1) / kShiftLines;
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
  // (integer shifts are aligned.hip's case; since r4 the tables are built for them too: where the x-marching
  // kernel's fast form applies it beats k_ata_aligned4x2, 31.5 vs 36 - 37 us at 256^3.  UNIRES_SHIFT_INT=0: not)
  static const bool no_int = getenv("UNIRES_SHIFT_INT") && getenv("UNIRES_SHIFT_INT")[0] == '0';
  if (no_int && t[0] == floorf(t[0]) && t[1] == floorf(t[1]) && t[2] == floorf(t[2])) return 1;
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
  // lane-window form of the same rows (x-marching kernel): a lane's four outputs z0 .. z0 + 3 read ONE window
  // xs[kmin .. kmin + 3] of the x-space line - their rows as weights on those four entries; lw = 0 where the
  // four rows of some lane span more than four entries (strides below 4: the per-voxel triples serve)
  std::vector<float> e4((size_t)dd.z * 4, 0.f);
  std::vector<int> kmin((size_t)kWave, 0);
  bool lane_window = xd.z <= kWave;
  for (int L = 0; L < kWave && lane_window; ++L) {
    int lo = 1 << 30, hi = -1;
    for (int z = 4 * L; z < std::min(4 * L + 4, dd.z); ++z) {
      int first;
      memcpy(&first, &e[(size_t)z * 4], 4);
      int used = 0;
      for (int m = 0; m < 3; ++m)
        if (e[(size_t)z * 4 + 1 + m] != 0.f) used = m + 1;
      if (used == 0) continue;
      lo = std::min(lo, first), hi = std::max(hi, first + used - 1);
    }
    if (hi < 0) lo = hi = 0;
    if (hi - lo > 3) lane_window = false;
    kmin[L] = lo;
    for (int z = 4 * L; z < std::min(4 * L + 4, dd.z) && lane_window; ++z) {
      int first;
      memcpy(&first, &e[(size_t)z * 4], 4);
      for (int m = 0; m < 3; ++m) {
        const float wv = e[(size_t)z * 4 + 1 + m];
        if (wv != 0.f) e4[(size_t)z * 4 + (first + m - lo)] = wv;
      }
    }
  }
  // aprons of the blended line: x-space voxel kk reads B[s kk + oz, s kk + oz + nf)
  const int lo = oz, hi = (xd.z - 1) * s + oz + nf - 1;
  int padl = lo < 0 ? -lo : 0, padr = hi >= dd.z ? hi - dd.z + 1 : 0;
  padl = (padl + 3) & ~3;
  if (padl > 256 || padr > 256) return 1;
  const int wave_floats = (padl + dd.z + padr + xd.z + 4 + 3) & ~3;  // (+ 4: a lane window may start at the last voxel)
  if (((size_t)xd.z * kShiftMaxTaps + (size_t)kShiftLines * wave_floats) * sizeof(float) > 60 * 1024) return 1;
  const size_t n_cx = cx.size(), n_cy = cy.size(), n_f = f.size(), n_e = e.size(), n_e4 = e4.size(), n_km = kmin.size();
  const size_t total = n_cx + n_cy + n_f + n_e + n_e4 + n_km;
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
  all.insert(all.end(), e4.begin(), e4.end());
  for (int v : kmin) {
    float fv;
    memcpy(&fv, &v, 4);
    all.push_back(fv);
  }
  if (hipMemcpy(S.dev, all.data(), total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
  S.o_e = 0, S.o_cx = n_e, S.o_cy = n_e + n_cx, S.o_f = n_e + n_cx + n_cy;
  S.o_e4 = S.o_f + n_f, S.o_kmin = S.o_e4 + n_e4, S.lane_window = lane_window;
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
  G.e4 = reinterpret_cast<const float4 *>(S.dev + S.o_e4), G.kmin = reinterpret_cast<const int *>(S.dev + S.o_kmin);
  static const bool no_lw = getenv("UNIRES_SHIFT_LW") && getenv("UNIRES_SHIFT_LW")[0] == '0';
  G.lw = S.lane_window && !no_lw ? 1 : 0;
  static const int dbg = getenv("UNIRES_SHIFT_DBG") ? atoi(getenv("UNIRES_SHIFT_DBG")) : 0;
  G.dbg = dbg;
  static const bool defer = getenv("UNIRES_SHIFT_DEFER") && getenv("UNIRES_SHIFT_DEFER")[0] == '1';
  G.defer = defer ? 1 : 0;
  const size_t lds = ((size_t)S.xdz * kShiftMaxTaps + (size_t)kShiftLines * S.wave_floats) * sizeof(float);
  const dim3 grid(shift_blocks(dd)), block(kWave, kShiftLines);
  static const bool no_x2 = getenv("UNIRES_SHIFT_X2") && getenv("UNIRES_SHIFT_X2")[0] == '0';
  G.xr = shift_xr(dd);
  const size_t lds2 = ((size_t)S.xdz * kShiftMaxTaps + (size_t)2 * kShiftLines * S.wave_floats) * sizeof(float);
  const int nl = shift_nl(dd);
  const size_t lds_m = ((size_t)S.xdz * kShiftMaxTaps + (size_t)nl * kShiftLines * S.wave_floats + (size_t)4 * (dd.x + dd.z)) *
                       sizeof(float);
  G.prof = nullptr;
#ifdef UNIRES_SHIFT_PROF
  static unsigned long long *prof_dev = nullptr;
  if (!prof_dev) (void)hipMalloc((void **)&prof_dev, (size_t)4096 * kShiftLines * 8 * sizeof(unsigned long long));
  (void)hipMemsetAsync(prof_dev, 0, (size_t)4096 * kShiftLines * 8 * sizeof(unsigned long long), st);
  G.prof = prof_dev;
#endif
  static const size_t pad_lds = getenv("UNIRES_SHIFT_PADLDS") ? (size_t)atoi(getenv("UNIRES_SHIFT_PADLDS")) : 0;  // (measurement: fewer workgroups per CU)
  if (G.xr > 0 && lds_m <= 64 * 1024) {
#define SHIFT_M_LAUNCH(NLV, FASTV)                                                                             \
  do {                                                                                                         \
    if (objb)                                                                                                  \
      hipLaunchKernelGGL((k_ata_shift_m<NLV, true, true, FASTV>), grid, block, lds_m + pad_lds, st, G, done);   \
    else if (partials)                                                                                         \
      hipLaunchKernelGGL((k_ata_shift_m<NLV, true, false, FASTV>), grid, block, lds_m + pad_lds, st, G, done);  \
    else                                                                                                       \
      hipLaunchKernelGGL((k_ata_shift_m<NLV, false, false, FASTV>), grid, block, lds_m + pad_lds, st, G, done); \
  } while (0)
    const bool fast = shift_fast(S, dd) && G.lw && G.dbg == 0 && G.defer == 0;
    if (nl == 2 && fast)
      SHIFT_M_LAUNCH(2, true);
    else if (nl == 2)
      SHIFT_M_LAUNCH(2, false);
    else if (fast)
      SHIFT_M_LAUNCH(1, true);
    else
      SHIFT_M_LAUNCH(1, false);
#undef SHIFT_M_LAUNCH
#ifdef UNIRES_SHIFT_PROF
    {
      static int shots = 0;
      if (++shots == 12) {  // one warmed-up launch: mean clocks per step and wave
        (void)hipStreamSynchronize(st);
        const size_t nw = (size_t)grid.x * kShiftLines;
        std::vector<unsigned long long> h(nw * 8);
        (void)hipMemcpy(h.data(), prof_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double steps = 0, tot = 0, wt = 0, zt = 0, ot = 0;
        unsigned long long b0 = ~0ull, b1 = 0, e1 = 0;
        for (size_t i = 0; i < nw; ++i) {
          if (!h[i * 8]) continue;
          steps += (double)h[i * 8], tot += (double)h[i * 8 + 1], wt += (double)h[i * 8 + 2], zt += (double)h[i * 8 + 3], ot += (double)h[i * 8 + 4];
          b0 = std::min(b0, h[i * 8 + 5]), b1 = std::max(b1, h[i * 8 + 5]), e1 = std::max(e1, h[i * 8 + 5] + h[i * 8 + 1]);
        }
        fprintf(stderr, "[shift_m] %zu waves, %.0f steps; clocks per step: total %.0f = load wait %.0f + z operator %.0f + stencil / store %.0f "
                        "+ rest %.0f; waves start within %llu clocks, kernel spans %llu clocks\n",
                nw, steps, tot / steps, wt / steps, zt / steps, ot / steps, (tot - wt - zt - ot) / steps, b1 - b0, e1 - b0);
      }
    }
#endif
    return 0;
  }
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
