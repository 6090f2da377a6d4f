// gather2.hip - k_gather2: the adjoint of the trilinear pull in register-blocked GATHER
// form, for operators close to the identity (the usual case: sub-voxel..few-voxel rigid
// misalignment, |rotation| < ~0.2 rad, unit voxel ratio along the grid axes).
//
//   out[v] = alpha * sum_u  hat(gx(u)-vx) hat(gy(u)-vy) hat(gz(u)-vz) mask(g(u)) h[u]
//
// (hat(d) = max(0, 1-|d|): v is a corner of g(u) iff every |g_d - v_d| < 1, and the
// trilinear weight of that corner is the product of the hats - exactly what nitorch's
// scatter-push adds into v.)  A lane owns a 2x2 patch of output (x,y) at one z, lanes
// of a wave run along z.  Candidate sources: 4x4 grid rows (ui,uj) around M^-1 v and,
// per row, the two grid-z positions bracketing v (a third one in the rare lanes where
// |dz/dk| < 1 lets three fit).  No LDS tile, no atomics, fixed summation order
// (bit-reproducible), full occupancy; the DtD stencil and the CG dot product are fused
// into the single coalesced write of the output.
//
// Unlike k_splat (one wave per LDS tile, 2-4 waves per SIMD) this kernel is limited only
// by registers, which is what an instruction-issue-bound kernel needs.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fused.hpp"

namespace unires {

struct G2Args {
  const float *src;      // grid-space volume, or x-space volume (CONVZ)
  int gx, gy, gz;        // grid dims
  int xdy, xdz;          // x-space dims (CONVZ)
  const float4 *ztab;    // CONVZ: per grid z {bits(x-space z offset), w0, w1, -}, gz entries
  int ztab_in_lds;
  Affine A, Ainv;
  float lox, loy;        // ia0 = floor(u*_x + lox) + 1  (lox = min over the patch - hx)
  float inv_c2, thr_lo, thr_hi;
  float alpha, tol;
  const float *p;
  float a0, cx, cy, cz;
  float *dst;
  Dim3i dd;
  int accumulate;
  double *partials;
  const float *objb;
};

// max(0, 1-|d|) in ONE instruction: fmed3(x,0,1) folds into v_sub_f32 |d| clamp
// (__saturatef costs five)
__device__ __forceinline__ float hat1(float d) {
  return __builtin_amdgcn_fmed3f(1.f - fabsf(d), 0.f, 1.f);
}

template <bool LDS>
__device__ __forceinline__ float4 ztab_at(const float4 *lds, const float4 *__restrict__ glb, int k) {
  if (LDS) return lds[k];
  return glb[k];
}

// CONVZ: 0 grid-space source; 1 x-space source + z table in LDS; 2 x-space source + z table
// read from global memory (grids too long for the LDS copy)
template <int CONVZ>
__global__ void __launch_bounds__(kBlock) k_gather2(G2Args G, const int *__restrict__ done) {
  if (done && *done) return;
  const Dim3i dd = G.dd;
  const float *__restrict__ src = G.src;
  const float *__restrict__ pin = G.p;
  float *__restrict__ dst = G.dst;
  const int lane = threadIdx.x, wave = threadIdx.y;
  const int nzc = (dd.z + kWave - 1) / kWave, nyp = (dd.y + 1) / 2, nxp = (dd.x + 1) / 2;
  const int ntiles = nzc * nyp * nxp;
  const float c0 = G.A.m[2], c1 = G.A.m[6], c2 = G.A.m[10];
  const float a0x = G.A.m[0], a0y = G.A.m[4], a0z = G.A.m[8];  // step of g per grid x
  const float a1x = G.A.m[1], a1y = G.A.m[5], a1z = G.A.m[9];  // step of g per grid y
  extern __shared__ float4 s_ztab[];  // CONVZ: copy of G.ztab (gz entries) when it fits
  constexpr bool tab_lds = CONVZ == 1;
  if (tab_lds) {
    for (int i = threadIdx.y * kWave + threadIdx.x; i < G.gz; i += kBlock) s_ztab[i] = G.ztab[i];
    __syncthreads();
  }
  const unsigned gyz = (unsigned)G.gy * (unsigned)G.gz;
  const unsigned xyz = CONVZ ? (unsigned)G.xdy * (unsigned)G.xdz : 0u;
  double dot = 0.0;
  for (int t = blockIdx.x * (kBlock / kWave) + wave; t < ntiles;
       t += gridDim.x * (kBlock / kWave)) {
    const int zc = t % nzc;
    const int t2 = t / nzc;
    const int yp = t2 % nyp, xp = t2 / nyp;
    const int x0 = 2 * xp, y0 = 2 * yp, z = zc * kWave + lane;
    const bool zin = z < dd.z;
    const float vx = (float)x0, vy = (float)y0, vz = (float)min(z, dd.z - 1);
    // first candidate row: smallest integer above (u* over the 2x2 patch) - h
    float ux, uy, uz;
    affine_point(G.Ainv, vx, vy, vz, ux, uy, uz);
    const int ia0 = (int)floorf(ux + G.lox) + 1, ja0 = (int)floorf(uy + G.loy) + 1;
    // row base relative to the patch origin: g(ia0, ja0, k) - (x0, y0, z) = rb + k*c
    const float fi = (float)ia0, fj = (float)ja0;
    const float rbx = fmaf(a1x, fj, fmaf(a0x, fi, G.A.m[3])) - vx;
    const float rby = fmaf(a1y, fj, fmaf(a0y, fi, G.A.m[7])) - vy;
    const float rbz = fmaf(a1z, fj, fmaf(a0z, fi, G.A.m[11])) - vz;
    // in-FOV mask (extrapolate=False) can only bite for outputs on a boundary plane
    const bool edge_xy = x0 == 0 || y0 == 0 || x0 + 2 >= dd.x || y0 + 2 >= dd.y;
    const bool zedge = zc == 0 || zc == nzc - 1;  // wave-uniform: clamps / z mask needed
    float acc00 = 0.f, acc01 = 0.f, acc10 = 0.f, acc11 = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ui = ia0 + a, uj = ja0 + b;
        const float rx = rbx + ((float)a * a0x + (float)b * a1x);
        const float ry = rby + ((float)a * a0y + (float)b * a1y);
        const float rz = rbz + ((float)a * a0z + (float)b * a1z);
        const bool rowok = (unsigned)ui < (unsigned)G.gx && (unsigned)uj < (unsigned)G.gy;
        // grid-z candidates: gz(k) - vz = rz + k*c2 ; |.| < 1  <=>  |k - tk| < 1/c2
        const float tk = -rz * G.inv_c2;
        const float ka = floorf(tk);
        const float frac = tk - ka;
        int ki = (int)ka;
        const float dz_a = fmaf(ka, c2, rz), dx_a = fmaf(ka, c0, rx), dy_a = fmaf(ka, c1, ry);
        const unsigned rowoff = CONVZ ? __umul24((unsigned)ui, xyz) + __umul24((unsigned)uj, (unsigned)G.xdz)
                                      : __umul24((unsigned)ui, gyz) + __umul24((unsigned)uj, (unsigned)G.gz);
        // up to three candidates: ka, ka+1 always; ka-1 or ka+2 when the lane's frac says so
        const bool third = frac < G.thr_lo || frac > G.thr_hi;
        // interior fast path (wave-uniform): every lane's row exists, all candidate slices
        // exist, no output of the wave sits on a boundary plane -> no masks, no clamps
        const bool fast = !zedge && !edge_xy && __all(rowok && ki >= 1 && ki + 2 < G.gz);
        if (fast) {
          float ha, hb;
          if (CONVZ) {
            const float4 ta = ztab_at<tab_lds>(s_ztab, G.ztab, ki);
            const float4 tb = ztab_at<tab_lds>(s_ztab, G.ztab, ki + 1);
            const float2 pa = ld2_u(src + rowoff + (unsigned)__float_as_int(ta.x));
            const float2 pb = ld2_u(src + rowoff + (unsigned)__float_as_int(tb.x));
            ha = ta.y * pa.x + ta.z * pa.y;
            hb = tb.y * pb.x + tb.z * pb.y;
          } else {
            const float2 pr = ld2_u(src + rowoff + (unsigned)ki);
            ha = pr.x, hb = pr.y;
          }
          {
            const float pw = hat1(dz_a) * ha;
            const float t0 = pw * hat1(dx_a), t1 = pw * hat1(dx_a - 1.f);
            const float wy0 = hat1(dy_a), wy1 = hat1(dy_a - 1.f);
            acc00 = fmaf(t0, wy0, acc00), acc01 = fmaf(t0, wy1, acc01);
            acc10 = fmaf(t1, wy0, acc10), acc11 = fmaf(t1, wy1, acc11);
          }
          {
            const float dz = dz_a + c2, dx = dx_a + c0, dy = dy_a + c1;
            const float pw = hat1(dz) * hb;
            const float t0 = pw * hat1(dx), t1 = pw * hat1(dx - 1.f);
            const float wy0 = hat1(dy), wy1 = hat1(dy - 1.f);
            acc00 = fmaf(t0, wy0, acc00), acc01 = fmaf(t0, wy1, acc01);
            acc10 = fmaf(t1, wy0, acc10), acc11 = fmaf(t1, wy1, acc11);
          }
          if (__any(third)) {  // rare third slice (|dz/dk| < 1): zero weight where not needed
            const float m = frac < 0.5f ? -1.f : 2.f;
            const int k = ki + (int)m;
            const float dz = fmaf(m, c2, dz_a), dx = fmaf(m, c0, dx_a), dy = fmaf(m, c1, dy_a);
            float h;
            if (CONVZ) {
              const float4 tb = ztab_at<tab_lds>(s_ztab, G.ztab, k);
              const float2 pr = ld2_u(src + rowoff + (unsigned)__float_as_int(tb.x));
              h = tb.y * pr.x + tb.z * pr.y;
            } else {
              h = src[rowoff + (unsigned)k];
            }
            const float pw = hat1(dz) * h;
            const float t0 = pw * hat1(dx), t1 = pw * hat1(dx - 1.f);
            const float wy0 = hat1(dy), wy1 = hat1(dy - 1.f);
            acc00 = fmaf(t0, wy0, acc00), acc01 = fmaf(t0, wy1, acc01);
            acc10 = fmaf(t1, wy0, acc10), acc11 = fmaf(t1, wy1, acc11);
          }
          continue;
        }
        // general path: volume shell (missing rows / slices, in-FOV mask)
        const int ncand = __any(third && rowok) ? 3 : 2;
        for (int cnd = 0; cnd < ncand; ++cnd) {
          int k;
          float dz, dx, dy;
          if (cnd == 0) {
            k = ki, dz = dz_a, dx = dx_a, dy = dy_a;
          } else if (cnd == 1) {
            k = ki + 1, dz = dz_a + c2, dx = dx_a + c0, dy = dy_a + c1;
          } else {
            const float m = frac < 0.5f ? -1.f : 2.f;
            k = ki + (int)m, dz = fmaf(m, c2, dz_a), dx = fmaf(m, c0, dx_a), dy = fmaf(m, c1, dy_a);
          }
          float wz = hat1(dz);
          const bool kok = (unsigned)k < (unsigned)G.gz;
          if (!(rowok && kok)) wz = 0.f;
          {
            // absolute position of the source; mask as nitorch does (g in (-tol, n-1+tol))
            const float gxa = dx + vx, gya = dy + vy, gza = dz + vz;
            if (!in_fov(gxa, gya, gza, dd, G.tol)) wz = 0.f;
          }
          const int kc = min(max(k, 0), G.gz - 1);
          float h;
          if (CONVZ) {
            const float4 tb = ztab_at<tab_lds>(s_ztab, G.ztab, kc);
            const unsigned o = (rowok ? rowoff : 0u) + (unsigned)__float_as_int(tb.x);
            const float2 pr = ld2_u(src + o);
            h = tb.y * pr.x + tb.z * pr.y;
          } else {
            h = src[(rowok ? rowoff : 0u) + (unsigned)kc];
          }
          const float pw = wz * h;
          const float wx0 = hat1(dx), wx1 = hat1(dx - 1.f), wy0 = hat1(dy), wy1 = hat1(dy - 1.f);
          const float t0 = pw * wx0, t1 = pw * wx1;
          acc00 = fmaf(t0, wy0, acc00);
          acc01 = fmaf(t0, wy1, acc01);
          acc10 = fmaf(t1, wy0, acc10);
          acc11 = fmaf(t1, wy1, acc11);
        }
      }
    }
    // ---- epilogue: 4 outputs (x0+dx, y0+dy, z) ----
    const float accs[4] = {acc00, acc01, acc10, acc11};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int i = x0 + (o >> 1), j = y0 + (o & 1);
      if (!zin || i >= dd.x || j >= dd.y) continue;
      const size_t idx = ((size_t)i * dd.y + j) * dd.z + z;
      float q = G.alpha * accs[o];
      float pc = 0.f;
      if (pin) {
        const float st = dtd_at(pin, idx, i, j, z, dd, G.cx, G.cy, G.cz, pc);
        q += G.a0 * pc + st;
      }
      if (G.accumulate) q += dst[idx];
      matvec_emit(dst, idx, q, pc, G.objb, G.partials != nullptr, dot);
    }
  }
  if (G.partials) {
    const double tot = block_sum(dot);
    if (threadIdx.x == 0 && threadIdx.y == 0) G.partials[blockIdx.x] = tot;
  }
}

int gather2_blocks(Dim3i dd) {
  const long long nt = (long long)((dd.z + kWave - 1) / kWave) * ((dd.y + 1) / 2) * ((dd.x + 1) / 2);
  const long long nb = (nt + 3) / 4;
  return (int)(nb < kMaxPartials ? nb : kMaxPartials);
}

// Host-side table for the on-the-fly conv_up along z: for grid slice u the x-space slices
// koff, koff+1 contribute with weights w0, w1 (tap * even/odd scale), packed so ONE 8-byte
// load fetches both.  `out` has gz entries of 4 floats.
void gather2_ztab(const Taps &T, const Scaling &S, int gz, int xdz, float *out) {
  const int K = T.n[2], s = T.s[2];
  for (int u = 0; u < gz; ++u) {
    int khi = u / s;
    if (khi > xdz - 1) khi = xdz - 1;
    const int tt = u - K + 1;
    int klo = tt <= 0 ? 0 : (tt + s - 1) / s;
    const int n = khi - klo + 1;
    float w0 = 0.f, w1 = 0.f;
    const float se = S.dim == 2 ? S.e : 1.f, so = S.dim == 2 ? S.o : 1.f;
    if (n >= 1) w0 = T.t[2][u - s * klo] * ((klo & 1) ? so : se);
    if (n >= 2) w1 = T.t[2][u - s * (klo + 1)] * (((klo + 1) & 1) ? so : se);
    int koff = n < 1 ? 0 : klo;
    if (koff > xdz - 2) {
      koff = xdz - 2;
      w1 = w0, w0 = 0.f;
    }
    int bits = koff;
    memcpy(&out[4 * u], &bits, 4);
    out[4 * u + 1] = w0, out[4 * u + 2] = w1, out[4 * u + 3] = 0.f;
  }
}

// Non-zero return (nothing launched): the operator is outside this kernel's domain.
int launch_gather2(const PushSrc &src, const float4 *ztab_dev, const Affine &A, const Affine &Ainv,
                   float alpha, float tol, const PushEpilogue &ep, float *dst, Dim3i dd,
                   const int *done, hipStream_t st) {
  const float c2 = A.m[10];
  if (!(c2 > 0.9f && c2 < 1.12f)) return 1;
  // candidate rows: the 2x2 patch needs every ui with |L(u - u*)| possibly < 1
  float h[2], span[2];
  for (int r = 0; r < 2; ++r) {
    h[r] = fabsf(Ainv.m[4 * r]) + fabsf(Ainv.m[4 * r + 1]) + fabsf(Ainv.m[4 * r + 2]) + 2e-3f;
    span[r] = fabsf(Ainv.m[4 * r]) + fabsf(Ainv.m[4 * r + 1]);  // u* moves this much over the patch
    if (span[r] + 2.f * h[r] > 4.f - 1e-3f) return 1;          // 4 rows per axis must cover it
  }
  if (src.gd.x > 4000 || src.gd.y > 4000 || src.gd.z > 4000) return 1;  // 24-bit index math
  G2Args G;
  G.src = src.data;
  G.gx = src.gd.x, G.gy = src.gd.y, G.gz = src.gd.z;
  G.xdy = src.xd.y, G.xdz = src.xd.z;
  G.ztab = ztab_dev;
  if (src.convup) {
    if (!ztab_dev) return 1;
    if (src.T.n[0] != 1 || src.T.n[1] != 1 || src.T.s[0] != 1 || src.T.s[1] != 1) return 1;
    if (src.T.t[0][0] != 1.f || src.T.t[1][0] != 1.f) return 1;
    if ((src.T.n[2] + src.T.s[2] - 1) / src.T.s[2] > 2 || src.xd.z < 2) return 1;
    if (src.S.dim >= 0 && src.S.dim != 2) return 1;
  }
  G.A = A, G.Ainv = Ainv;
  G.lox = fminf(0.f, Ainv.m[0]) + fminf(0.f, Ainv.m[1]) - h[0];
  G.loy = fminf(0.f, Ainv.m[4]) + fminf(0.f, Ainv.m[5]) - h[1];
  G.inv_c2 = 1.f / c2;
  const float s = 1.f / c2;  // |k - tk| < s
  G.thr_lo = (s - 1.f) + 2e-3f;         // frac below this: ka-1 may be inside
  G.thr_hi = (2.f - s) - 2e-3f;         // frac above this: ka+2 may be inside
  G.alpha = alpha, G.tol = tol;
  G.p = ep.p, G.a0 = ep.a0, G.cx = ep.cx, G.cy = ep.cy, G.cz = ep.cz;
  G.dst = dst, G.dd = dd;
  G.accumulate = ep.accumulate;
  G.partials = ep.partials;
  G.objb = ep.objb;
  const dim3 grid(gather2_blocks(dd)), block(kWave, kBlock / kWave);
  const size_t tab_bytes = src.convup ? (size_t)G.gz * sizeof(float4) : 0;
  G.ztab_in_lds = tab_bytes > 0 && tab_bytes <= 16 * 1024;
  if (src.convup && G.ztab_in_lds)
    hipLaunchKernelGGL(k_gather2<1>, grid, block, tab_bytes, st, G, done);
  else if (src.convup)
    hipLaunchKernelGGL(k_gather2<2>, grid, block, 0, st, G, done);
  else
    hipLaunchKernelGGL(k_gather2<0>, grid, block, 0, st, G, done);
  return 0;
}

}  // namespace unires
