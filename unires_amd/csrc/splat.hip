// splat.hip - k_splat: the lean, specialised form of the owner-computes push
// (same algorithm and race-freedom argument as k_push_tile in fused.hip, which stays
// as the general fallback) for the two sources the headline configurations use:
//
//   AXIS = -1    : grid-space source (denoising regime, op-level push)
//   AXIS = 0,1,2 : thick slices along that axis: conv_up along ONE axis, fan-in <= 2
//                  (every rect profile), regenerated on the fly from x-space
//   AXIS = 3     : conv_up along all three axes (isotropic down-sampling, BASELINE config 4),
//                  fan-in <= 2 per axis: 8 x-space values per grid voxel, four 8-byte loads
//
// Differences that matter for instruction count (this kernel is issue-bound, not
// HBM-bound - DESIGN.md 4): one z table instead of three, the two x-space values of a
// grid voxel fetched with ONE 8-byte load, lane hand-over with DPP wave shifts instead
// of ds_bpermute, a straight-line splat for the common no-conflict case, small kernel
// argument block (no SGPR spills).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "fused.hpp"

namespace unires {

struct SplatArgs {
  const float *src;
  int gx, gy, gz;     // grid dims
  int xdy, xdz, xdn;  // x-space dims y, z and along the thick axis (AXIS >= 0)
  int nkz, sz;        // taps / stride along the thick axis
  float kz[UNIRES_MAX_TAPS];
  float se, so;       // even/odd slice scaling along the thick axis (1,1 = none)
  // AXIS == 3 (conv_up along all three axes, e.g. isotropic down-sampling, BASELINE config 4):
  // per-axis taps live in kz[10 a .. 10 a + 9]
  int nk3[3], s3[3], xd3[3], sdim;
  Affine A, Ainv;
  float alpha, tol;
  const float *p;
  float a0, cx, cy, cz;
  float *dst;
  Dim3i dd;
  int accumulate;
  double *partials;
  const float *objb;
  int row_sep;
  int dbg;
  unsigned long long *prof;  // UNIRES_SPLAT_PROF builds only: per-phase tick sums
};

// Tile shapes.  A wave-instruction of the splat carries G = 64 / L row segments of up to L
// lanes along grid z.  Measured on config 3 (k_splat, channels with 0.06 / 0.096 / 0.094 rad of
// tilt out of the z axis; a tilted z line leaves a tile column through its sides, so the
// number of segments grows ~1.5x and so does the run time):
//   Long  8 x 4 x 30, L = 32 : 169 / 236 / 208 us   <- used
//   Short 8 x 8 x 14, L = 16 : 219 / 239 / 234 us   (less apron, but per-tile set-up doubles)
//         8 x 8 x 30, L = 32 : 219 / 281 / 267 us   (fewer segments, but 12 instead of 16 waves/CU)
//         6 x 4 x 30, L = 32 : 204 / 287 / 274 us   (20 instead of 16 waves/CU, but 34 % more tiles)
//   earlier sweeps of the long form: 6x6 230, 4x8 249, 4x4 329 (vs 8x4 218)
// Also tried and dropped: SHEARED tiles (layer z of a tile shifted by round(z c0/c2), round(z c1/c2)
// voxels so that a tilted z line stays inside one tile column).  It works (all parity and
// reproducibility tests pass) and does what it should to the splat proper - 30 full segments per
// tile instead of 41 half-empty ones, splat phase 153 -> 121 us on the 0.1 rad channel - but the
// stencil epilogue then writes rows that change (x, y) every ~10 z: its time doubles (41 -> 91 us)
// and the kernel ends up slower (203 / 246 / 251 us vs 162 / 227 / 201 us).
// Also tried and dropped: packing the short segments of a tilted grid (half of them are <= 16
// lanes) four to an instruction - 183 / 235 / 221 us with a second code instantiation, 187 / 254 /
// 235 us with a run-time group width: short segments come from rows next to the same tile face,
// so the packed rows conflict and fall into the serialised turn schedule.
// UNIRES_SPLAT_CFG=short selects the short tile (kept as a tested variant).
struct SplatLong {
  static constexpr int TX = 8, TY = 4, TZ = 30, L = 32;
};
struct SplatShort {
  static constexpr int TX = 8, TY = 8, TZ = 14, L = 16;
};

struct Seg {  // one <=L-long run of a grid row (ui,uj)
  short ui, uj, k0;
  unsigned char len, solo;  // solo: its partner row is too close -> splat in its own turn
};

#ifdef UNIRES_NO_DPP
__device__ __forceinline__ int dpp_up1(int v) { return __shfl_up(v, 1, kWave); }
__device__ __forceinline__ int dpp_dn1(int v) { return __shfl_down(v, 1, kWave); }
#else
__device__ __forceinline__ int dpp_up1(int v) {  // lane i <- lane i-1 (lane 0 keeps its own)
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_dn1(int v) {  // lane i <- lane i+1 (lane 63 keeps its own)
  return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
}
#endif
__device__ __forceinline__ float dpp_up1f(float v) {
  return __int_as_float(dpp_up1(__float_as_int(v)));
}

#ifdef UNIRES_HARD_FENCE
#define SPLAT_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define SPLAT_FENCE() asm volatile("" ::: "memory")
#endif

// phase-ablation switches (tools/abl.sh): compiled in only with -DUNIRES_ABLATE, then driven by
// the UNIRES_DBG environment variable; product builds carry none of it
#ifdef UNIRES_ABLATE
#define ABL(bit) ((P.dbg & (bit)) != 0)
#else
#define ABL(bit) (false)
#endif

#ifdef UNIRES_SPLAT_PROF  // per-phase timer sums (debug builds only)
#define SP_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define SP_ADD(slot, t0, t1) \
  if (P.prof && lane == 0) atomicAdd(P.prof + (slot), (t1) - (t0))
#else
#define SP_T(var)
#define SP_ADD(slot, t0, t1)
#endif

template <int AXIS, class CFG>
__global__ void __launch_bounds__(kWave) k_splat(SplatArgs P, const int *__restrict__ done) {
  if (done && *done) return;
  constexpr bool CONV = AXIS >= 0;
  constexpr int TX = CFG::TX, TY = CFG::TY, TZ = CFG::TZ;
  constexpr int L = CFG::L, G = kWave / L;  // lanes per segment, segments per instruction
  static_assert(TZ + 2 <= L && (L == 16 || L == 32) && TY % G == 0, "tile / lane-group shape");
  constexpr int SZ = TZ + 2, SY = TY + 2, SXd = TX + 2, N = SXd * SY * SZ;
  constexpr int XS = SY * SZ, YS = SZ;  // strides of the aproned accumulator (x, y; z = 1)
  constexpr int kSegs = 128;
  __shared__ __align__(16) float acc[N];
  __shared__ __align__(8) Seg rows[kSegs];
  __shared__ __align__(16) float4 ztab[64];  // {bits(k offset), w0, w1, -}
  __shared__ __align__(16) float4 xytab[AXIS == 3 ? 64 : 1];  // AXIS 3: x table, then y table
  const int lane = threadIdx.x, grp = lane / L, gl = lane & (L - 1);
  const Dim3i dd = P.dd;
  const float *__restrict__ src = P.src;
  const float *__restrict__ pin = P.p;
  float *__restrict__ dst = P.dst;
  const int ntx = (dd.x + TX - 1) / TX, nty = (dd.y + TY - 1) / TY, ntz = (dd.z + TZ - 1) / TZ;
  const int ntiles = ntx * nty * ntz;
  const int nxcd = 8;  // XCD-aware persistent schedule (see k_push_tile)
  const int per_xcd = (ntiles + nxcd - 1) / nxcd;
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
  const int slots = (gridDim.x + nxcd - 1 - xcd) / nxcd;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const float c0 = P.A.m[2], c1 = P.A.m[6], c2 = P.A.m[10];
  const float inv_sz = CONV ? 1.f / (float)P.sz : 1.f;
  double dot = 0.0;
  for (int tl = slot; tl < per_xcd; tl += slots) {
    const int t = xcd * per_xcd + tl;
    if (t >= ntiles) break;
    const int tzi = t % ntz, tyi = (t / ntz) % nty, txi = t / (ntz * nty);
    const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
    const int ex = min(TX, dd.x - x0), ey = min(TY, dd.y - y0), ez = min(TZ, dd.z - z0);
    SP_T(t_start);
    SPLAT_FENCE();
    for (int i = lane; i < N / 4; i += kWave)
      reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float flx = (float)(x0 - 1), fly = (float)(y0 - 1), flz = (float)(z0 - 1);
    const float fhx = (float)(x0 + ex), fhy = (float)(y0 + ey), fhz = (float)(z0 + ez);
    // per-point acceptance box: the tile's aproned cell range intersected with the field of
    // view (g > -tol  <=>  g >= nextafter(-tol); g < n - 1 + tol): folding the in-FOV mask into
    // the tile bounds costs nothing per point (30 % of the tiles touch the volume boundary)
    const float tlo = nextafterf(-P.tol, 1.f);
    const float tlx = fmaxf(flx, tlo), tly = fmaxf(fly, tlo), tlz = fmaxf(flz, tlo);
    const float thx = fminf(fhx, (float)(dd.x - 1) + P.tol), thy = fminf(fhy, (float)(dd.y - 1) + P.tol),
                thz = fminf(fhz, (float)(dd.z - 1) + P.tol);
    // grid-space bounding box of everything that can touch the tile
    float lo0 = 1e30f, lo1 = 1e30f, lo2 = 1e30f, hi0 = -1e30f, hi1 = -1e30f, hi2 = -1e30f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float ux, uy, uz;
      affine_point(P.Ainv, (c & 4) ? fhx : flx, (c & 2) ? fhy : fly, (c & 1) ? fhz : flz, ux, uy,
                   uz);
      lo0 = fminf(lo0, ux), hi0 = fmaxf(hi0, ux);
      lo1 = fminf(lo1, uy), hi1 = fmaxf(hi1, uy);
      lo2 = fminf(lo2, uz), hi2 = fmaxf(hi2, uz);
    }
    const int bx0 = max(0, (int)floorf(lo0 - 0.01f)), bx1 = min(P.gx - 1, (int)ceilf(hi0 + 0.01f));
    const int by0 = max(0, (int)floorf(lo1 - 0.01f)), by1 = min(P.gy - 1, (int)ceilf(hi1 + 0.01f));
    const int bz0 = max(0, (int)floorf(lo2 - 0.01f)), bz1 = min(P.gz - 1, (int)ceilf(hi2 + 0.01f));
    const int nby = by1 - by0 + 1;
    const int nrow_cand = max(bx1 - bx0 + 1, 0) * max(nby, 0);
    if (AXIS == 3) {
      // one table per axis: {first x-space slice, weight of it, weight of the next one}
      auto entry = [&](int u, int nk, int sz, int xdn, int tbase, bool scaled) {
        const float inv = 1.f / (float)sz;
        int khi = (int)(((float)u + 0.5f) * inv);
        if (khi * sz > u) --khi;
        if ((khi + 1) * sz <= u) ++khi;  // khi = u / sz exactly
        khi = min(khi, xdn - 1);
        const int tt = u - nk + 1;
        int klo = 0;
        if (tt > 0) {
          klo = (int)(((float)(tt + sz - 1) + 0.5f) * inv);
          if (klo * sz > tt + sz - 1) --klo;
          if ((klo + 1) * sz <= tt + sz - 1) ++klo;  // ceil(tt / sz)
        }
        const int n = khi - klo + 1;
        const float fe = scaled ? P.se : 1.f, fo = scaled ? P.so : 1.f;
        float w0 = 0.f, w1 = 0.f;
        if (n >= 1) w0 = P.kz[tbase + u - sz * klo] * ((klo & 1) ? fo : fe);
        if (n >= 2) w1 = P.kz[tbase + u - sz * (klo + 1)] * (((klo + 1) & 1) ? fo : fe);
        int koff = n < 1 ? 0 : klo;
        if (koff > xdn - 2) koff = xdn - 2, w1 = w0, w0 = 0.f;  // last slice: pair (xdn-2, xdn-1)
        return make_float4(__int_as_float(koff), w0, w1, 0.f);
      };
      const int a = lane >> 5, l5 = lane & 31;  // lanes 0..31: x table, 32..63: y table
      xytab[lane] = entry(min((a ? by0 : bx0) + l5, (a ? P.gy : P.gx) - 1), P.nk3[a], P.s3[a],
                          P.xd3[a], 10 * a, P.sdim == a);
      ztab[lane] = entry(min(bz0 + lane, P.gz - 1), P.nk3[2], P.s3[2], P.xd3[2], 20, P.sdim == 2);
    } else if (CONV) {
      // which x-space slices feed grid slice u = (box start along AXIS) + lane; for AXIS 2 the
      // pair is packed for ONE 8-byte load, else the second slice is one x/y stride away
      const int b0 = AXIS == 0 ? bx0 : (AXIS == 1 ? by0 : bz0);
      const int gn = AXIS == 0 ? P.gx : (AXIS == 1 ? P.gy : P.gz);
      const int uz = min(b0 + lane, gn - 1);
      int khi = (int)(((float)uz + 0.5f) * inv_sz);
      if (khi * P.sz > uz) --khi;
      if ((khi + 1) * P.sz <= uz) ++khi;                  // khi = uz / sz exactly
      khi = min(khi, P.xdn - 1);
      const int tt = uz - P.nkz + 1;
      int klo = 0;
      if (tt > 0) {
        klo = (int)(((float)(tt + P.sz - 1) + 0.5f) * inv_sz);
        if (klo * P.sz > tt + P.sz - 1) --klo;
        if ((klo + 1) * P.sz <= tt + P.sz - 1) ++klo;     // ceil(tt / sz)
      }
      const int n = khi - klo + 1;
      float w0 = 0.f, w1 = 0.f;
      if (n >= 1) w0 = P.kz[uz - P.sz * klo] * ((klo & 1) ? P.so : P.se);
      if (n >= 2) w1 = P.kz[uz - P.sz * (klo + 1)] * (((klo + 1) & 1) ? P.so : P.se);
      int koff = klo;
      if (n < 1) koff = 0;
      if (koff > P.xdn - 2) {  // last slice: fetch the pair (xdn-2, xdn-1), weight on the second
        koff = P.xdn - 2;
        w1 = w0, w0 = 0.f;
      }
      ztab[lane] = make_float4(__int_as_float(koff), w0, w1, 0.f);
    }
    SP_T(t_setup);
    SP_ADD(0, t_start, t_setup);
    // ---- phase A: rows (ui,uj) -> exact grid-z intervals -> <=32-long segments ----
    const int segs_per_row = (max(bz1 - bz0 + 1, 1) + L - 1) / L;
    // candidate rows per pass: never more segments than the LDS row list holds (a grid much
    // finer than the output in z yields long rows; the usual case is 2 segments per row)
    const int chunk = max(1, min(kWave, kSegs / segs_per_row));
    int nseg = 0;
    for (int rc0 = 0;;) {
      if (ABL(16)) break;
      SP_T(t_a0);
      if (rc0 < nrow_cand) {
        const int rc = rc0 + lane;
        int ui = 0, uj = 0, k0 = 0, k1 = -1;
        if (rc < nrow_cand && lane < chunk) {
          const int a = rc / nby, b = rc - a * nby;
          ui = bx0 + a, uj = by0 + b;
          float r0, r1, r2;
          affine_point(P.A, (float)ui, (float)uj, 0.f, r0, r1, r2);
          k0 = bz0, k1 = bz1;
          const float rr[3] = {r0, r1, r2}, cc[3] = {c0, c1, c2};
          const float lw[3] = {flx, fly, flz}, hg[3] = {fhx, fhy, fhz};
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            if (fabsf(cc[d]) > 1e-6f) {
              float ta = (lw[d] - rr[d]) / cc[d], tb = (hg[d] - rr[d]) / cc[d];
              const float tmin = fminf(ta, tb), tmax = fmaxf(ta, tb);
              ta = fmaxf(tmin, -1e6f), tb = fminf(tmax, 1e6f);
              k0 = max(k0, (int)ceilf(ta - 2e-3f - 1e-5f * fabsf(ta)));
              k1 = min(k1, (int)floorf(tb + 2e-3f + 1e-5f * fabsf(tb)));
            } else if (rr[d] < lw[d] - 0.01f || rr[d] >= hg[d] + 0.01f) {
              k1 = k0 - 1;
            }
          }
        }
        bool has = k1 >= k0;
        while (__any(has)) {
          const unsigned long long m = __ballot(has);
          const int pos = nseg + __popcll(m & lt_mask);
          if (has && pos < kSegs)
            rows[pos] = Seg{(short)ui, (short)uj, (short)k0, (unsigned char)min(L, k1 - k0 + 1), 0};
          nseg += __popcll(m);
          k0 += L;
          has = has && k1 >= k0;
        }
        rc0 += chunk;
      }
      const bool last = rc0 >= nrow_cand;
      if (!last && nseg + chunk * segs_per_row <= kSegs) continue;
      SPLAT_FENCE();
      const int nr = min(nseg, kSegs);
      const int npair = (nr + G - 1) / G;  // instruction p carries rows p, p + npair, ... (G of them)
      // a row closer than row_sep to a row of a LOWER group in its instruction cannot be
      // splat together with it: it gets a turn of its own
      for (int idx = npair + lane; idx < nr; idx += kWave) {
        const int p = idx % npair, g = idx / npair;
        const Seg b = rows[idx];
        bool solo = false;
        for (int g2 = 0; g2 < g; ++g2) {
          const Seg a = rows[p + g2 * npair];
          solo = solo || max(abs(a.ui - b.ui), abs(a.uj - b.uj)) < P.row_sep;
        }
        if (solo) rows[idx].solo = 1;
      }
      SPLAT_FENCE();
      SP_T(t_a1);
      SP_ADD(1, t_a0, t_a1);
      // ---- phase B: one lane group per segment, lanes along grid z, kU instructions per batch ----
      constexpr int kU = 4;
      for (int p0 = 0; p0 < (ABL(8) ? 0 : npair); p0 += kU) {
        SP_T(t_b0);
        float val[kU];
        int ui[kU], uj[kU], uk[kU];
        bool act[kU], solo[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int p = p0 + u;
          const int idx = p + grp * npair;
          const Seg R = rows[min(idx, nr - 1)];
          act[u] = p < npair && idx < nr && gl < R.len;
          solo[u] = R.solo && grp > 0;
          ui[u] = R.ui, uj[u] = R.uj, uk[u] = min(R.k0 + gl, P.gz - 1);
          if (AXIS == 3) {
            const float4 tx = xytab[min(max(ui[u] - bx0, 0), 31)];
            const float4 ty = xytab[32 + min(max(uj[u] - by0, 0), 31)];
            const float4 tz = ztab[min(uk[u] - bz0, 63)];
            const unsigned sy = (unsigned)P.xdz, sx = (unsigned)P.xdy * (unsigned)P.xdz;
            const unsigned base =
                __umul24(__umul24((unsigned)__float_as_int(tx.x), (unsigned)P.xdy) +
                             (unsigned)__float_as_int(ty.x), (unsigned)P.xdz) +
                (unsigned)__float_as_int(tz.x);
            const float2 p00 = ld2_u(src + base), p01 = ld2_u(src + base + sy),
                         p10 = ld2_u(src + base + sx), p11 = ld2_u(src + base + sx + sy);
            const float z00 = tz.y * p00.x + tz.z * p00.y, z01 = tz.y * p01.x + tz.z * p01.y;
            const float z10 = tz.y * p10.x + tz.z * p10.y, z11 = tz.y * p11.x + tz.z * p11.y;
            val[u] = tx.y * (ty.y * z00 + ty.z * z01) + tx.z * (ty.y * z10 + ty.z * z11);
          } else if (AXIS == 2) {
            const float4 tb = ztab[min(uk[u] - bz0, 63)];
            const unsigned base =
                __umul24(__umul24((unsigned)ui[u], (unsigned)P.xdy) + (unsigned)uj[u],
                         (unsigned)P.xdz) + (unsigned)__float_as_int(tb.x);
            const float2 pr = ld2_u(src + base);
            val[u] = tb.y * pr.x + tb.z * pr.y;
          } else if (AXIS == 0) {
            const float4 tb = ztab[min(max(ui[u] - bx0, 0), 63)];
            const unsigned base =
                __umul24(__umul24((unsigned)__float_as_int(tb.x), (unsigned)P.xdy) + (unsigned)uj[u],
                         (unsigned)P.xdz) + (unsigned)uk[u];
            val[u] = tb.y * src[base] + tb.z * src[base + (unsigned)P.xdy * (unsigned)P.xdz];
          } else if (AXIS == 1) {
            const float4 tb = ztab[min(max(uj[u] - by0, 0), 63)];
            const unsigned base =
                __umul24(__umul24((unsigned)ui[u], (unsigned)P.xdy) + (unsigned)__float_as_int(tb.x),
                         (unsigned)P.xdz) + (unsigned)uk[u];
            val[u] = tb.y * src[base] + tb.z * src[base + (unsigned)P.xdz];
          } else {
            const unsigned base =
                __umul24(__umul24((unsigned)ui[u], (unsigned)P.gy) + (unsigned)uj[u],
                         (unsigned)P.gz) + (unsigned)uk[u];
            val[u] = src[base];
          }
        }
        SP_T(t_b1);
        SP_ADD(2, t_b0, t_b1);
        if (ABL(4)) continue;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          float gx, gy, gz;
          affine_point(P.A, (float)ui[u], (float)uj[u], (float)uk[u], gx, gy, gz);
          bool ok = act[u] && gx >= tlx && gx < thx && gy >= tly && gy < thy && gz >= tlz && gz < thz;
          const float v = P.alpha * val[u];
          ok = ok && v != 0.f;
          if (!__any(ok)) continue;  // nothing of this instruction lands in the tile
          const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
          const int lx = (int)fx - (x0 - 1), ly = (int)fy - (y0 - 1), lz = (int)fz - (z0 - 1);
          const float wx1 = gx - fx, wy1 = gy - fy, wz1 = gz - fz;
          const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
          const int cell = ok ? (lx * SY + ly) * SZ + lz : -2 - lane;
          const int lzk = ok ? lz : -2 - lane;
          const float a00 = v * (wx0 * wy0), a01 = v * (wx0 * wy1), a10 = v * (wx1 * wy0),
                      a11 = v * (wx1 * wy1);
          float l00 = a00 * wz0, l01 = a01 * wz0, l10 = a10 * wz0, l11 = a11 * wz0;
          const float u00 = a00 * wz1, u01 = a01 * wz1, u10 = a10 * wz1, u11 = a11 * wz1;
          // a lane in the same z plane as its predecessor may overlap it inside one update
          // group: replay it in a later turn (rare: |dz/dk| < 1 by a hair)
          const int prev_lz = dpp_up1(lzk);  // executed by ALL lanes: a DPP read of a lane that
                                             // is masked off returns the reader's own value
          const bool dup = ok && gl > 0 && lzk == prev_lz;
          const bool slow = __any(dup) || __any(ok && solo[u]) || ABL(1);
          if (!slow) {
            // common case, straight line: z-adjacent lanes hand their shared plane over in
            // registers, then every lane updates ONE z plane (+ the top plane of lanes
            // with nobody above them)
            const int below = dpp_up1(cell), above = dpp_dn1(cell);
            const float p00 = dpp_up1f(u00), p01 = dpp_up1f(u01), p10 = dpp_up1f(u10),
                        p11 = dpp_up1f(u11);
            const bool recv = ok && gl > 0 && below + 1 == cell;
            const bool sent = ok && gl < L - 1 && above == cell + 1;
            if (recv) l00 += p00, l01 += p01, l10 += p10, l11 += p11;
            float *q = acc + (ok ? cell : 0);
            SPLAT_FENCE();
            if (ok) {
              const float o00 = q[0], o01 = q[YS], o10 = q[XS], o11 = q[XS + YS];
              q[0] = o00 + l00, q[YS] = o01 + l01, q[XS] = o10 + l10, q[XS + YS] = o11 + l11;
            }
            SPLAT_FENCE();
            if (ok && !sent) {
              const float o00 = q[1], o01 = q[YS + 1], o10 = q[XS + 1], o11 = q[XS + YS + 1];
              q[1] = o00 + u00, q[YS + 1] = o01 + u01, q[XS + 1] = o10 + u10,
              q[XS + YS + 1] = o11 + u11;
            }
            SPLAT_FENCE();
          } else {
            // turn schedule: (row B of a too-close pair) x (replayed same-plane neighbour);
            // within a turn no hand-over, each lane updates both of its planes
            const int myturn = (solo[u] ? 2 * grp : 0) + (dup ? 1 : 0);
            for (int turn = 0; turn < 2 * G; ++turn) {
              if (!__any(ok && myturn == turn)) continue;
              SPLAT_FENCE();
              if (ok && myturn == turn) {
                float *q = acc + cell;
                const float o00 = q[0], o01 = q[YS], o10 = q[XS], o11 = q[XS + YS];
                q[0] = o00 + l00, q[YS] = o01 + l01, q[XS] = o10 + l10, q[XS + YS] = o11 + l11;
              }
              SPLAT_FENCE();
              if (ok && myturn == turn) {
                float *q = acc + cell;
                const float o00 = q[1], o01 = q[YS + 1], o10 = q[XS + 1], o11 = q[XS + YS + 1];
                q[1] = o00 + u00, q[YS + 1] = o01 + u01, q[XS + 1] = o10 + u10,
                q[XS + YS + 1] = o11 + u11;
              }
              SPLAT_FENCE();
            }
          }
        }
        SP_T(t_b2);
        SP_ADD(3, t_b1, t_b2);
      }
      nseg = 0;
      if (last) break;
    }
    SPLAT_FENCE();
    SP_T(t_e0);
    // ---- epilogue: q = [q +] acc + a0 p + c DtD p ; dot += p*q  (one row per lane group) ----
    // one output row (lx, ly, all z) per lane group
    // Fast form for tiles whose x/y stencil neighbours are all inside the volume (91 % of the
    // tiles of a 256^3 volume): the row base is a scalar, the lane offset is computed once per
    // tile, and every load/store is "scalar base + lane offset" - no per-row index arithmetic.
    // (A wave64 VALU instruction occupies the SIMD for 4 clocks; the generic form below spends
    // ~125 of them per pair of rows, this one ~25.)
    if (ABL(2)) continue;
    constexpr int RPX = TY / G;  // instructions per x slab of the tile
    static_assert((TX * RPX) % 4 == 0, "fast epilogue unrolls four instructions");
    const bool fast_xy = pin != nullptr && !P.accumulate && x0 > 0 && y0 > 0 &&
                         x0 + TX < dd.x && y0 + TY < dd.y && dd.numel() < (1ull << 29);
    if (fast_xy) {
      const size_t sxe = (size_t)dd.y * dd.z, sye = dd.z;
      const int kc = min(z0 + gl, dd.z - 1);
      const bool act = gl < ez, lzf = kc > 0, hzf = kc + 1 < dd.z;
      // buffer addressing: per-lane byte offset (once per tile) + scalar row offset
      // (soffset is added as an unsigned value: the lane offset is taken relative to the
      // (x0-1, y0-1) row so that every scalar row offset below is non-negative)
      const unsigned e0 = 4u * (unsigned)(((x0 - 1) * dd.y + y0 - 1 + grp) * dd.z + kc);
      const unsigned em = lzf ? e0 - 4u : e0, ep = hzf ? e0 + 4u : e0;
      const unsigned sxb = 4u * (unsigned)sxe, syb = 4u * (unsigned)sye;
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(pin, dd.numel() * 4),
                                   rd = make_rsrc(dst, dd.numel() * 4),
                                   rb = make_rsrc(P.objb ? P.objb : pin, dd.numel() * 4);
      const float *arow = acc + (SY + 1 + grp) * SZ + gl + 1;
      auto rows = [&](auto obj_tag) {
        constexpr bool OBJ = decltype(obj_tag)::value;
#pragma unroll 1
        for (int it = 0; it < TX * RPX; it += 4) {
          // readfirstlane: keeps the row base in SGPRs and hides the induction variable from
          // loop strength reduction (which would turn every address into a 64-bit VGPR pointer)
          const int it0 = __builtin_amdgcn_readfirstlane(it);
          float c[4], vxp[4], vxm[4], vyp[4], vym[4], vzp[4], vzm[4], ob[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {  // all 28 loads of four row pairs in flight together
            const unsigned ro = (unsigned)((it0 + j) / RPX + 1) * sxb +
                                (unsigned)(((it0 + j) % RPX) * G + 1) * syb;
            c[j] = buf_load(rp, e0, ro);
            vxp[j] = buf_load(rp, e0, ro + sxb), vxm[j] = buf_load(rp, e0, ro - sxb);
            vyp[j] = buf_load(rp, e0, ro + syb), vym[j] = buf_load(rp, e0, ro - syb);
            vzp[j] = buf_load(rp, ep, ro), vzm[j] = buf_load(rp, em, ro);
            if (OBJ) ob[j] = buf_load(rb, e0, ro);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int lx = (it0 + j) / RPX, ly2 = ((it0 + j) % RPX) * G;  // this group's row: (lx, ly2 + grp)
            const unsigned ro = (unsigned)(lx + 1) * sxb + (unsigned)(ly2 + 1) * syb;
            float q = arow[(lx * SY + ly2) * SZ];
            const float xf = vxp[j] - c[j], xb = c[j] - vxm[j], yf = vyp[j] - c[j], yb = c[j] - vym[j];
            const float zf = (hzf ? vzp[j] : 0.f) - c[j], zb = lzf ? c[j] - vzm[j] : 0.f;
            const float st = P.cx * (xb - xf) + P.cy * (yb - yf) + P.cz * (zb - zf);
            q += P.a0 * c[j] + st;
            if (act) {
              if (OBJ) {
                dot += (double)obj_term(q, ob[j], c[j]);
              } else {
                buf_store(q, rd, e0, ro);
                dot += (double)__fmul_rn(c[j], q);  // (unused when no partials are requested)
              }
            }
          }
        }
      };
      if (P.objb)
        rows(std::true_type{});
      else
        rows(std::false_type{});
    } else {
#pragma unroll 4
      for (int r = grp; r < TX * TY; r += G) {
        const int lx = r / TY, ly = r % TY, lz = gl;
        if (lx >= ex || ly >= ey || lz >= ez) continue;
        const int i = x0 + lx, j = y0 + ly, k = z0 + lz;
        const size_t idx = ((size_t)i * dd.y + j) * dd.z + k;
        float q = acc[((lx + 1) * SY + ly + 1) * SZ + lz + 1];
        float pc = 0.f;
        if (pin) {
          const float st = dtd_at(pin, idx, i, j, k, dd, P.cx, P.cy, P.cz, pc);
          q += P.a0 * pc + st;
        }
        if (P.accumulate) q += dst[idx];
        matvec_emit(dst, idx, q, pc, P.objb, P.partials != nullptr, dot);
      }
    }
    SP_T(t_e1);
    SP_ADD(4, t_e0, t_e1);
    SP_ADD(5, t_start, t_e1);
    SP_ADD(6, 0ull, 1ull);
  }
  if (P.partials) {
    const double tot = wave_sum(dot);
    if (lane == 0) P.partials[blockIdx.x] = tot;
  }
}

// Tile shape for this operator: the short tile when a grid z line drifts sideways by more than
// ~1 voxel over a long tile column (UNIRES_SPLAT_CFG=long|short forces one).
static bool splat_use_short(const Affine &A) {
  static const char *force = getenv("UNIRES_SPLAT_CFG");
  if (force && force[0] == 'l') return false;
  if (force && force[0] == 's') return true;
  (void)A;  // measured on config 3 (channels with 0.06 / 0.096 / 0.094 rad tilt): the short tile
  return false;  // loses everywhere (219/239/234 us vs 169/236/208 us) - per-tile set-up dominates
}

template <class CFG>
static long long splat_tiles(Dim3i dd) {
  return (long long)((dd.x + CFG::TX - 1) / CFG::TX) * ((dd.y + CFG::TY - 1) / CFG::TY) *
         ((dd.z + CFG::TZ - 1) / CFG::TZ);
}

int splat_blocks(Dim3i dd, const Affine &A) {
  const long long nt = splat_use_short(A) ? splat_tiles<SplatShort>(dd) : splat_tiles<SplatLong>(dd);
  static const int cap = getenv("UNIRES_SPLAT_BLOCKS") ? atoi(getenv("UNIRES_SPLAT_BLOCKS")) : 4096;
  const int lim = cap < kMaxPartials ? cap : kMaxPartials;
  return (int)(nt < lim ? nt : lim);  // persistent grid
}

// Returns non-zero (nothing launched) when the operator is outside this kernel's domain;
// the caller then uses the general k_push_tile.
int launch_splat(const PushSrc &src, const Affine &A, const Affine &Ainv, const SplatSafety &safe,
                 float alpha, float tol, const PushEpilogue &ep, float *dst, Dim3i dd,
                 const int *done, hipStream_t st) {
  if (safe.use_atomics) return 1;
  if (dd.x > 32000 || dd.y > 32000 || dd.z > 32000) return 1;
  SplatArgs P;
  P.src = src.data;
  P.gx = src.gd.x, P.gy = src.gd.y, P.gz = src.gd.z;
  if (P.gx > 32000 || P.gy > 32000 || P.gz > 32000) return 1;
  if (!fits_fast_index(src.gd) || (src.convup && !fits_fast_index(src.xd))) return 1;
  P.xdy = src.xd.y, P.xdz = src.xd.z;
  P.nkz = 1, P.sz = 1, P.se = 1.f, P.so = 1.f;
  for (int d = 0; d < 3; ++d) P.nk3[d] = 1, P.s3[d] = 1, P.xd3[d] = 2;
  P.sdim = -1;
  int axis = -1;
  if (src.convup) {
    // conv_up along ONE axis (AXIS 0..2) or along several (AXIS 3), with at most two x-space
    // slices per grid slice and axis
    int nconv = 0;
    for (int d = 0; d < 3; ++d) {
      const bool dirac = src.T.n[d] == 1 && src.T.s[d] == 1 && src.T.t[d][0] == 1.f;
      if (dirac) continue;
      ++nconv;
      axis = d;
    }
    if (nconv > 1) {
      const int xdv[3] = {src.xd.x, src.xd.y, src.xd.z};
      for (int d = 0; d < 3; ++d) {
        if ((src.T.n[d] + src.T.s[d] - 1) / src.T.s[d] > 2 || xdv[d] < 2 || src.T.n[d] > 10) return 1;
        P.nk3[d] = src.T.n[d], P.s3[d] = src.T.s[d], P.xd3[d] = xdv[d];
      }
      // the x / y tables hold 32 grid slices: the grid-space box of an aproned tile must fit
      const float ext[3] = {(float)SplatLong::TX + 1.f, (float)SplatLong::TY + 1.f, (float)SplatLong::TZ + 1.f};
      for (int r = 0; r < 2; ++r) {
        float e = 3.f;
        for (int c = 0; c < 3; ++c) e += fabsf(Ainv.m[4 * r + c]) * ext[c];
        if (e > 30.f) return 1;
      }
      for (int i = 0; i < UNIRES_MAX_TAPS; ++i) P.kz[i] = 0.f;
      for (int d = 0; d < 3; ++d)
        for (int i = 0; i < src.T.n[d]; ++i) P.kz[10 * d + i] = src.T.t[d][i];
      P.sdim = src.S.dim;
      if (src.S.dim >= 0) P.se = src.S.e, P.so = src.S.o;
      P.xdn = 1;
      axis = 3;
    }
  }
  if (src.convup) {
    // the conv tables hold 64 grid slices along z (32 along x / y for AXIS 3): the grid-space
    // box of an aproned tile must fit, else the general kernel (which checks per tile) runs
    const float ext[3] = {(float)SplatLong::TX + 1.f, (float)SplatLong::TY + 1.f, (float)SplatLong::TZ + 1.f};
    for (int r = 0; r < 3; ++r) {
      float e = 3.f;
      for (int c = 0; c < 3; ++c) e += fabsf(Ainv.m[4 * r + c]) * ext[c];
      if (e > 62.f) return 1;
    }
  }
  if (src.convup && axis != 3) {
    if (axis < 0) axis = 2;  // all dirac: conv_up is the identity, any axis works
    const int xdv[3] = {src.xd.x, src.xd.y, src.xd.z};
    if ((src.T.n[axis] + src.T.s[axis] - 1) / src.T.s[axis] > 2 || xdv[axis] < 2) return 1;
    if (src.S.dim >= 0 && src.S.dim != axis) return 1;
    P.nkz = src.T.n[axis], P.sz = src.T.s[axis];
    P.xdn = xdv[axis];
    for (int i = 0; i < UNIRES_MAX_TAPS; ++i) P.kz[i] = src.T.t[axis][i];
    if (src.S.dim == axis) P.se = src.S.e, P.so = src.S.o;
  } else if (!src.convup) {
    P.xdn = 1;
    for (int i = 0; i < UNIRES_MAX_TAPS; ++i) P.kz[i] = 0.f;
  }
  P.A = A, P.Ainv = Ainv;
  P.alpha = alpha, P.tol = tol;
  P.p = ep.p, P.a0 = ep.a0, P.cx = ep.cx, P.cy = ep.cy, P.cz = ep.cz;
  P.dst = dst, P.dd = dd;
  P.accumulate = ep.accumulate;
  P.partials = ep.partials;
  P.objb = ep.objb;
  P.row_sep = safe.row_sep;
  static const int dbg = getenv("UNIRES_DBG") ? atoi(getenv("UNIRES_DBG")) : 0;
  P.dbg = dbg;
  P.prof = nullptr;
#ifdef UNIRES_SPLAT_PROF
  static unsigned long long *prof = nullptr;
  if (!prof) (void)hipMalloc((void **)&prof, 8 * sizeof(unsigned long long));
  (void)hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), st);
  P.prof = prof;
#endif
  const dim3 grid(splat_blocks(dd, A));
#define LAUNCH_SPLAT(CFG)                                                              \
  do {                                                                                 \
    if (axis == 0)                                                                     \
      hipLaunchKernelGGL((k_splat<0, CFG>), grid, dim3(kWave), 0, st, P, done);        \
    else if (axis == 1)                                                                \
      hipLaunchKernelGGL((k_splat<1, CFG>), grid, dim3(kWave), 0, st, P, done);        \
    else if (axis == 2)                                                                \
      hipLaunchKernelGGL((k_splat<2, CFG>), grid, dim3(kWave), 0, st, P, done);        \
    else if (axis == 3)                                                                \
      hipLaunchKernelGGL((k_splat<3, CFG>), grid, dim3(kWave), 0, st, P, done);        \
    else                                                                               \
      hipLaunchKernelGGL((k_splat<-1, CFG>), grid, dim3(kWave), 0, st, P, done);       \
  } while (0)
  if (splat_use_short(A))
    LAUNCH_SPLAT(SplatShort);
  else
    LAUNCH_SPLAT(SplatLong);
#undef LAUNCH_SPLAT
#ifdef UNIRES_SPLAT_PROF
  {
    unsigned long long h[8];
    (void)hipMemcpyAsync(h, prof, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    const double nt = h[6] ? (double)h[6] : 1.0;
    fprintf(stderr, "[splat prof] tiles %llu  ticks/tile: setup %.0f  phaseA %.0f  B-issue %.0f  B-rmw %.0f  "
            "epilogue %.0f  total %.0f\n", h[6], h[0] / nt, h[1] / nt, h[2] / nt, h[3] / nt, h[4] / nt, h[5] / nt);
  }
#endif
  return 0;
}

}  // namespace unires
