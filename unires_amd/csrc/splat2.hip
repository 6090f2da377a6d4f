// splat2.hip - schedule-driven owner-computes push (see splat2.hpp).
//
// Race freedom without atomics (LDS float atomics cost ~190 clocks per wave-instruction on
// gfx950, tools/mb_lds.hip): a tile is owned by ONE wave, whose LDS operations execute in
// order; inside one read-add-write group
//   * the lanes of a segment sit in DIFFERENT z planes (the build kernel cuts a row wherever two
//     consecutive points share floor(gz)), and a group touches one plane per lane: the lower
//     plane of every point first, the upper plane second;
//   * the two segments of an instruction come from rows >= row_sep apart (host-checked for the
//     affine: some coordinate of any two of their points differs by >= 2).
// The schedule is fixed, so results are bit-reproducible.
#include "splat2.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

namespace unires {

struct S2Tile {
  static constexpr int TX = 8, TY = 4, TZ = 30, L = 32;
  static constexpr int SX = TX + 2, SY = TY + 2, SZ = TZ + 2, N = SX * SY * SZ;
  static constexpr int XS = SY * SZ, YS = SZ;
};
constexpr int kS2Waves = 4;  // waves (= independent tiles in flight) per workgroup

#define S2_FENCE() asm volatile("" ::: "memory")

// coordinate of grid point k of a row with base (rx, ry, rz): identical, bit for bit, to
// affine_along() / affine_point() (the pull kernels), so that At is the exact adjoint of A
__device__ __forceinline__ void s2_point(const Affine &A, float rx, float ry, float rz, float kf,
                                         float &gx, float &gy, float &gz) {
  gx = fmaf(A.m[2], kf, rx) + A.m[3];
  gy = fmaf(A.m[6], kf, ry) + A.m[7];
  gz = fmaf(A.m[10], kf, rz) + A.m[11];
}

struct S2TileGeom {
  int x0, y0, z0, ex, ey, ez;
};
__device__ __forceinline__ S2TileGeom s2_tile(int t, const Dim3i &dd) {
  using T = S2Tile;
  const int nty = (dd.y + T::TY - 1) / T::TY, ntz = (dd.z + T::TZ - 1) / T::TZ;
  const int tzi = t % ntz, tyi = (t / ntz) % nty, txi = t / (ntz * nty);
  S2TileGeom g;
  g.x0 = txi * T::TX, g.y0 = tyi * T::TY, g.z0 = tzi * T::TZ;
  g.ex = min(T::TX, dd.x - g.x0), g.ey = min(T::TY, dd.y - g.y0), g.ez = min(T::TZ, dd.z - g.z0);
  return g;
}
static int s2_ntiles(Dim3i dd) {
  using T = S2Tile;
  return ((dd.x + T::TX - 1) / T::TX) * ((dd.y + T::TY - 1) / T::TY) * ((dd.z + T::TZ - 1) / T::TZ);
}

// --------------------------------------------------------------------------
// schedule build
// --------------------------------------------------------------------------
struct S2BuildArgs {
  Affine A, Ainv;
  Dim3i gd, dd;
  float tol;
  int row_sep;
  unsigned sx, sy;
  int tabsel;
};

struct S2Seg {
  short ui, uj, k0, len;
};

// One wave per tile.  FILL = false: counts[t] = number of entries; FILL = true: counts holds the
// exclusive prefix sum and the entries are written.  stats[0] += points, stats[1] += instructions.
template <bool FILL>
__global__ void __launch_bounds__(kWave)
    k_splat2_build(S2BuildArgs B, unsigned *__restrict__ counts, S2Entry *__restrict__ entries,
                   int *__restrict__ err, unsigned long long *__restrict__ stats) {
  using T = S2Tile;
  constexpr int L = T::L, kSegs = 384;
  __shared__ S2Seg segs[kSegs];
  __shared__ unsigned char flag[kSegs];
  const int lane = threadIdx.x;
  const Dim3i dd = B.dd;
  const int t = blockIdx.x;
  const S2TileGeom g = s2_tile(t, dd);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const float flx = (float)(g.x0 - 1), fly = (float)(g.y0 - 1), flz = (float)(g.z0 - 1);
  const float fhx = (float)(g.x0 + g.ex), fhy = (float)(g.y0 + g.ey), fhz = (float)(g.z0 + g.ez);
  // acceptance box of a point: floor cell inside the aproned tile AND inside the field of view
  // (g > -tol  <=>  g >= nextafter(-tol); g < n - 1 + tol)
  const float tlo = nextafterf(-B.tol, 1.f);
  const float tlx = fmaxf(flx, tlo), tly = fmaxf(fly, tlo), tlz = fmaxf(flz, tlo);
  const float thx = fminf(fhx, (float)(dd.x - 1) + B.tol), thy = fminf(fhy, (float)(dd.y - 1) + B.tol),
              thz = fminf(fhz, (float)(dd.z - 1) + B.tol);
  // grid-space bounding box of everything that can touch the tile
  float lo0 = 1e30f, lo1 = 1e30f, lo2 = 1e30f, hi0 = -1e30f, hi1 = -1e30f, hi2 = -1e30f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float ux, uy, uz;
    affine_point(B.Ainv, (c & 4) ? fhx : flx, (c & 2) ? fhy : fly, (c & 1) ? fhz : flz, ux, uy, uz);
    lo0 = fminf(lo0, ux), hi0 = fmaxf(hi0, ux);
    lo1 = fminf(lo1, uy), hi1 = fmaxf(hi1, uy);
    lo2 = fminf(lo2, uz), hi2 = fmaxf(hi2, uz);
  }
  const int bx0 = max(0, (int)floorf(lo0 - 0.01f)), bx1 = min(B.gd.x - 1, (int)ceilf(hi0 + 0.01f));
  const int by0 = max(0, (int)floorf(lo1 - 0.01f)), by1 = min(B.gd.y - 1, (int)ceilf(hi1 + 0.01f));
  const int bz0 = max(0, (int)floorf(lo2 - 0.01f) - 1), bz1 = min(B.gd.z - 1, (int)ceilf(hi2 + 0.01f) + 1);
  const int nby = by1 - by0 + 1;
  const int nrow_cand = max(bx1 - bx0 + 1, 0) * max(nby, 0);
  const float c0 = B.A.m[2], c1 = B.A.m[6], c2 = B.A.m[10];
  int nseg = 0;
  for (int rc0 = 0; rc0 < nrow_cand; rc0 += kWave) {
    const int rc = rc0 + lane;
    int ui = 0, uj = 0, k0 = 0, k1 = -1;
    RowBase rb{0.f, 0.f, 0.f};
    if (rc < nrow_cand) {
      const int a = rc / nby, b = rc - a * nby;
      ui = bx0 + a, uj = by0 + b;
      rb = affine_row(B.A, (float)ui, (float)uj);
      k0 = bz0, k1 = bz1;
      // slab clipping in real arithmetic gives a superset (with slack) of the accepted interval
      const float rr[3] = {rb.x + B.A.m[3], rb.y + B.A.m[7], rb.z + B.A.m[11]}, cc[3] = {c0, c1, c2};
      const float lw[3] = {flx, fly, flz}, hg[3] = {fhx, fhy, fhz};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (fabsf(cc[d]) > 1e-6f) {
          float ta = (lw[d] - rr[d]) / cc[d], tb = (hg[d] - rr[d]) / cc[d];
          const float tmin = fminf(ta, tb), tmax = fmaxf(ta, tb);
          ta = fmaxf(tmin, -1e6f), tb = fminf(tmax, 1e6f);
          k0 = max(k0, (int)ceilf(ta - 2e-3f - 1e-5f * fabsf(ta)) - 1);
          k1 = min(k1, (int)floorf(tb + 2e-3f + 1e-5f * fabsf(tb)) + 1);
        } else if (rr[d] < lw[d] - 0.01f || rr[d] >= hg[d] + 0.01f) {
          k1 = k0 - 1;
        }
      }
      // ... then exactly: the accepted points of a row form an interval (every rounding in
      // s2_point is monotone in k), so shrinking from both ends finds it
      auto accept = [&](int k) {
        float gx, gy, gz;
        s2_point(B.A, rb.x, rb.y, rb.z, (float)k, gx, gy, gz);
        return gx >= tlx && gx < thx && gy >= tly && gy < thy && gz >= tlz && gz < thz;
      };
      while (k0 <= k1 && !accept(k0)) ++k0;
      while (k1 >= k0 && !accept(k1)) --k1;
    }
    // cut into segments: <= L points, no two consecutive points in the same z plane
    int cur = k0;
    bool more = k1 >= k0;
    while (__any(more)) {
      int e = cur;
      if (more) {
        float gx, gy, gz;
        s2_point(B.A, rb.x, rb.y, rb.z, (float)cur, gx, gy, gz);
        float plz = floorf(gz);
        while (e + 1 <= k1 && e + 1 - cur < L) {
          s2_point(B.A, rb.x, rb.y, rb.z, (float)(e + 1), gx, gy, gz);
          const float lz = floorf(gz);
          if (lz == plz) break;
          plz = lz;
          ++e;
        }
      }
      const unsigned long long m = __ballot(more);
      const int pos = nseg + __popcll(m & lt_mask);
      if (more && pos < kSegs) segs[pos] = S2Seg{(short)ui, (short)uj, (short)cur, (short)(e - cur + 1)};
      nseg += __popcll(m);
      cur = e + 1;
      more = more && cur <= k1;
    }
  }
  if (nseg > kSegs) {
    if (lane == 0) atomicExch(err, 1);
    nseg = kSegs;
  }
  S2_FENCE();
  __syncthreads();
  // pairing: segment p with segment p + half (rows about half a tile apart); a pair whose rows
  // are closer than row_sep is split into two single-segment instructions
  const int half = (nseg + 1) / 2;
  int nconf = 0;
  for (int p0 = 0; p0 < half; p0 += kWave) {
    const int p = p0 + lane;
    bool conf = false;
    if (p < half && p + half < nseg) {
      const S2Seg a = segs[p], b = segs[p + half];
      conf = max(abs(a.ui - b.ui), abs(a.uj - b.uj)) < B.row_sep;
    }
    const unsigned long long m = __ballot(conf);
    if (p < half) flag[p] = conf ? (unsigned char)1 : (unsigned char)0;
    // index of this conflict among the tile's conflicts, stored for the fill pass
    if (conf) segs[p + half].len = (short)(segs[p + half].len | ((nconf + __popcll(m & lt_mask)) << 6));
    nconf += __popcll(m);
  }
  const int ninstr = half + nconf;
  if (!FILL) {
    if (lane == 0) counts[t] = 2u * (unsigned)ninstr;
    return;
  }
  S2_FENCE();
  __syncthreads();
  S2Entry *out = entries + counts[t];
  auto make = [&](const S2Seg s, int len) {
    const RowBase rb = affine_row(B.A, (float)s.ui, (float)s.uj);
    S2Entry e;
    e.rx = rb.x, e.ry = rb.y, e.rz = rb.z;
    e.k0f = (float)s.k0;
    e.srcoff = (unsigned)s.ui * B.sx + (unsigned)s.uj * B.sy;
    const unsigned tabidx = (unsigned)(B.tabsel ? s.uj : s.ui);
    e.kl = (unsigned)s.k0 | ((unsigned)len << 12) | (tabidx << 18);
    return e;
  };
  const S2Entry empty{0.f, 0.f, 0.f, 0.f, 0u, 0u};
  unsigned long long pts = 0;
  for (int p0 = 0; p0 < half; p0 += kWave) {
    const int p = p0 + lane;
    if (p >= half) continue;
    const S2Seg a = segs[p];
    out[2 * p] = make(a, a.len & 63);
    pts += (unsigned long long)(a.len & 63);
    S2Entry eb = empty;
    if (p + half < nseg) {
      const S2Seg b = segs[p + half];
      const int blen = b.len & 63;
      pts += (unsigned long long)blen;
      if (flag[p]) {
        const int ci = b.len >> 6;
        out[2 * (half + ci)] = make(b, blen);
        out[2 * (half + ci) + 1] = empty;
      } else {
        eb = make(b, blen);
      }
    }
    out[2 * p + 1] = eb;
  }
  if (stats) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) pts += __shfl_down(pts, off, kWave);
    if (lane == 0) {
      atomicAdd(stats, pts);
      atomicAdd(stats + 1, (unsigned long long)ninstr);
    }
  }
}

void splat2_free(SplatSched &S) {
  if (S.entries) (void)hipFree(S.entries);
  if (S.tile_off) (void)hipFree(S.tile_off);
  if (S.scratch) (void)hipFree(S.scratch);
  S = SplatSched();
}

int splat2_build(SplatSched &S, const Affine &A, const Affine &Ainv, Dim3i gd, Dim3i dd, float tol,
                 const SplatSafety &safe, int axis, unsigned sx, unsigned sy) {
  S.valid = false;
  static const bool off = getenv("UNIRES_NO_SPLAT2") != nullptr;
  if (off) return 1;
  if (safe.use_atomics) return 1;
  if (dd.x > 4000 || dd.y > 4000 || dd.z > 4000 || gd.x > 4000 || gd.y > 4000 || gd.z > 4000) return 1;
  if (!fits_fast_index(gd) || !fits_fast_index(dd) || dd.numel() >= (1ull << 30)) return 1;
  const int nt = s2_ntiles(dd);
  if ((size_t)nt + 1 > S.cap_tiles) {
    if (S.tile_off) (void)hipFree(S.tile_off);
    S.tile_off = nullptr;
    if (hipMalloc((void **)&S.tile_off, ((size_t)nt + 1) * sizeof(unsigned)) != hipSuccess) return 1;
    S.cap_tiles = (size_t)nt + 1;
  }
  if (!S.scratch && hipMalloc((void **)&S.scratch, 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
  (void)hipMemset(S.scratch, 0, 4 * sizeof(unsigned long long));
  int *err_dev = (int *)S.scratch;
  unsigned long long *stats_dev = S.scratch + 1;
  S2BuildArgs B;
  B.A = A, B.Ainv = Ainv, B.gd = gd, B.dd = dd, B.tol = tol, B.row_sep = safe.row_sep;
  B.sx = sx, B.sy = sy, B.tabsel = axis == 1 ? 1 : 0;
  hipLaunchKernelGGL(k_splat2_build<false>, dim3(nt), dim3(kWave), 0, 0, B, S.tile_off,
                     (S2Entry *)nullptr, err_dev, (unsigned long long *)nullptr);
  std::vector<unsigned> h((size_t)nt + 1);
  if (hipMemcpy(h.data(), S.tile_off, (size_t)nt * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
    return 1;
  unsigned run = 0;
  for (int i = 0; i < nt; ++i) {
    const unsigned c = h[i];
    h[i] = run;
    run += c;
  }
  h[nt] = run;
  if (run + 8 > S.cap_entries) {
    if (S.entries) (void)hipFree(S.entries);
    S.entries = nullptr;
    const size_t cap = (size_t)run + run / 8 + 64;
    if (hipMalloc((void **)&S.entries, cap * sizeof(S2Entry)) != hipSuccess) return 1;
    S.cap_entries = cap;
  }
  if (hipMemcpy(S.tile_off, h.data(), ((size_t)nt + 1) * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess)
    return 1;
  // a few entries past the end are read (never used) by the prefetch of the last tile
  (void)hipMemset(S.entries + run, 0, 8 * sizeof(S2Entry));
  hipLaunchKernelGGL(k_splat2_build<true>, dim3(nt), dim3(kWave), 0, 0, B, S.tile_off, S.entries,
                     err_dev, stats_dev);
  int herr = 0;
  unsigned long long hs[2] = {0, 0};
  if (hipMemcpy(&herr, err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (hipMemcpy(hs, stats_dev, sizeof(hs), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  static const bool verbose = getenv("UNIRES_SPLAT2_VERBOSE") != nullptr;
  if (herr) {
    if (verbose) fprintf(stderr, "[splat2] a tile exceeds the segment list: general kernel used\n");
    return 1;
  }
  S.ntiles = nt;
  S.total = run;
  S.axis = axis;
  S.fill = hs[1] ? (double)hs[0] / (64.0 * (double)hs[1]) : 0.0;
  S.valid = true;
  if (verbose)
    fprintf(stderr, "[splat2] %d tiles, %llu instructions, %llu points (%.2f per output voxel), lane fill %.3f, "
            "schedule %.1f MB, row_sep %d\n", nt, hs[1], hs[0], (double)hs[0] / (double)dd.numel(), S.fill,
            run * sizeof(S2Entry) / 1e6, safe.row_sep);
  return 0;
}

void splat2_convtab(const Taps &T, const Scaling &S, int axis, int gn, int xdn, float *out) {
  const int K = T.n[axis], s = T.s[axis];
  const float se = S.dim == axis ? S.e : 1.f, so = S.dim == axis ? S.o : 1.f;
  for (int u = 0; u < gn; ++u) {
    int khi = u / s;
    if (khi > xdn - 1) khi = xdn - 1;
    const int tt = u - K + 1;
    const int klo = tt <= 0 ? 0 : (tt + s - 1) / s;
    const int n = khi - klo + 1;
    float w0 = 0.f, w1 = 0.f;
    if (n >= 1) w0 = T.t[axis][u - s * klo] * ((klo & 1) ? so : se);
    if (n >= 2) w1 = T.t[axis][u - s * (klo + 1)] * (((klo + 1) & 1) ? so : se);
    int koff = n < 1 ? 0 : klo;
    if (koff > xdn - 2) {  // last slice: the pair (xdn-2, xdn-1), weight on the second
      koff = xdn - 2;
      w1 = w0, w0 = 0.f;
    }
    memcpy(&out[4 * u], &koff, 4);
    out[4 * u + 1] = w0, out[4 * u + 2] = w1, out[4 * u + 3] = 0.f;
  }
}

// --------------------------------------------------------------------------
// the splat
// --------------------------------------------------------------------------
struct S2Args {
  const float *src;
  size_t src_bytes;
  const float4 *tab;  // conv_up table (AXIS >= 0)
  int gn;             // entries in it
  int tabn;           // LDS table length (gn + L, padded with zeros)
  unsigned tab_step;  // elements between the two x-space values of a grid voxel (AXIS 0 / 1)
  const S2Entry *entries;
  const unsigned *tile_off;
  int ntiles;
  Affine A;
  float alpha;
  const float *p;
  float a0, cx, cy, cz;
  float *dst;
  Dim3i dd;
  int accumulate;
  double *partials;
  const float *objb;
  int dbg;  // UNIRES_S2_DBG ablation bits (0 in production): 1 no splat, 2 no epilogue, 4 no LDS updates, 8 no source loads
};

__device__ __forceinline__ S2Entry s2_load_entry(const S2Entry *p) {
  S2Entry e;
  __builtin_memcpy(&e, p, sizeof(e));
  return e;
}

template <int AXIS>
__global__ void __launch_bounds__(kWave *kS2Waves) k_splat2(S2Args P, const int *__restrict__ done) {
  if (done && *done) return;
  using T = S2Tile;
  constexpr int TX = T::TX, TY = T::TY, TZ = T::TZ, L = T::L, G = kWave / L;
  constexpr int SY = T::SY, SZ = T::SZ, N = T::N, XS = T::XS, YS = T::YS;
  constexpr bool CONV = AXIS >= 0;
  __shared__ __align__(16) float acc_all[kS2Waves][N];
  extern __shared__ float tabs[];  // CONV: byte offsets | w0 | w1, tabn entries each
  const int lane = threadIdx.x & (kWave - 1), grp = lane / L, gl = lane & (L - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *acc = acc_all[wave];
  const Dim3i dd = P.dd;
  const float *__restrict__ pin = P.p;
  float *__restrict__ dst = P.dst;
  if (CONV) {
    const unsigned step = AXIS == 2 ? 1u : P.tab_step;
    for (int i = threadIdx.x; i < P.tabn; i += kWave * kS2Waves) {
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < P.gn) e = P.tab[i];
      tabs[i] = __int_as_float((int)(4u * step * (unsigned)__float_as_int(e.x)));
      tabs[P.tabn + i] = P.alpha * e.y;
      tabs[2 * P.tabn + i] = P.alpha * e.z;
    }
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(P.src, P.src_bytes);
  const int ntiles = P.ntiles;
  // XCD-aware persistent schedule: workgroup b sits on XCD b % 8; each XCD walks one contiguous
  // run of tiles so that neighbouring tiles (shared stencil halos, schedule lines) share an L2
  const int nxcd = min(8, (int)gridDim.x);
  const int per_xcd = (ntiles + nxcd - 1) / nxcd;
  const int xcd = blockIdx.x % nxcd;
  const int slot = (blockIdx.x / nxcd) * kS2Waves + wave;
  const int slots = ((gridDim.x + nxcd - 1 - xcd) / nxcd) * kS2Waves;
  const float c0 = P.A.m[2], c1 = P.A.m[6], c2 = P.A.m[10];
  const float t0 = P.A.m[3], t1 = P.A.m[7], t2 = P.A.m[11];
  const float glf = (float)gl;
  const unsigned gl4 = 4u * (unsigned)gl;
  double dot = 0.0;
  for (int tl = slot; tl < per_xcd; tl += slots) {
    const int t = xcd * per_xcd + tl;
    if (t >= ntiles) break;
    const S2TileGeom g = s2_tile(t, dd);
    const int x0 = g.x0, y0 = g.y0, z0 = g.z0, ex = g.ex, ey = g.ey, ez = g.ez;
    S2_FENCE();
    for (int i = lane; i < N / 4; i += kWave)
      reinterpret_cast<float4 *>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float xb = (float)(x0 - 1), yb = (float)(y0 - 1), zb = (float)(z0 - 1);
    const unsigned off0 = P.tile_off[t], off1 = P.tile_off[t + 1];
    const int ninstr = (int)((off1 - off0) >> 1);
    const S2Entry *E = P.entries + off0 + grp;
    S2_FENCE();
    constexpr int kU = 4;
    auto load = [&](S2Entry(&e)[kU], int p0) {
#pragma unroll
      for (int u = 0; u < kU; ++u) e[u] = s2_load_entry(E + 2 * (p0 + u));  // (8 spare entries past the end)
    };
    auto process = [&](const S2Entry(&e)[kU], int p0) {
      float val[kU];
      bool act[kU];
      float kf[kU];
      // ---- source values: conv_up regenerated on the fly from x-space ----
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const unsigned kl = e[u].kl;
        const int len = (p0 + u < ninstr) ? (int)((kl >> 12) & 63u) : 0;
        act[u] = gl < len;
        kf[u] = e[u].k0f + glf;
        const unsigned k4 = 4u * (kl & 0xfffu) + gl4;  // byte offset of grid z
        const unsigned so4 = 4u * e[u].srcoff;
        if (P.dbg & 8) {
          val[u] = 1.f;
        } else if (AXIS == 2) {
          const float koff = tabs[k4 >> 2], w0 = tabs[P.tabn + (k4 >> 2)], w1 = tabs[2 * P.tabn + (k4 >> 2)];
          const unsigned a = so4 + (unsigned)__float_as_int(koff);
          const float pa = buf_load(rsrc, a, 0), pb = buf_load(rsrc, a + 4u, 0);
          val[u] = w0 * pa + w1 * pb;
        } else if (AXIS == 0 || AXIS == 1) {
          const unsigned ti = kl >> 18;
          const float koff = tabs[ti], w0 = tabs[P.tabn + ti], w1 = tabs[2 * P.tabn + ti];
          const unsigned a = so4 + (unsigned)__float_as_int(koff) + k4;
          const float pa = buf_load(rsrc, a, 0), pb = buf_load(rsrc, a + 4u * P.tab_step, 0);
          val[u] = w0 * pa + w1 * pb;
        } else {
          val[u] = P.alpha * buf_load(rsrc, so4 + k4, 0);
        }
      }
      // ---- the splat proper ----
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        float gx, gy, gz;
        gx = fmaf(c0, kf[u], e[u].rx) + t0;
        gy = fmaf(c1, kf[u], e[u].ry) + t1;
        gz = fmaf(c2, kf[u], e[u].rz) + t2;
        // local coordinates: the subtraction of the (integer) tile base is exact
        const float lxf = gx - xb, lyf = gy - yb, lzf = gz - zb;
        const float wx1 = __builtin_amdgcn_fractf(lxf), wy1 = __builtin_amdgcn_fractf(lyf),
                    wz1 = __builtin_amdgcn_fractf(lzf);
        const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
        // cell index in float (exact: small integers), one conversion
        const float cf = fmaf(lxf - wx1, (float)XS, fmaf(lyf - wy1, (float)YS, lzf - wz1));
        const int cell = (int)cf;
        const float v = val[u];
        const float vx0 = v * wx0, vx1 = v * wx1;
        const float a00 = vx0 * wy0, a01 = vx0 * wy1, a10 = vx1 * wy0, a11 = vx1 * wy1;
        S2_FENCE();
        if (P.dbg & 4) dot += (double)(a00 + a01 + a10 + a11 + (float)cell);
        if (act[u] && !(P.dbg & 4)) {
          float *q = acc + cell;
          {
            const float o00 = q[0], o01 = q[YS], o10 = q[XS], o11 = q[XS + YS];
            q[0] = o00 + a00 * wz0, q[YS] = o01 + a01 * wz0, q[XS] = o10 + a10 * wz0,
            q[XS + YS] = o11 + a11 * wz0;
          }
          S2_FENCE();
          {
            const float o00 = q[1], o01 = q[YS + 1], o10 = q[XS + 1], o11 = q[XS + YS + 1];
            q[1] = o00 + a00 * wz1, q[YS + 1] = o01 + a01 * wz1, q[XS + 1] = o10 + a10 * wz1,
            q[XS + YS + 1] = o11 + a11 * wz1;
          }
        }
        S2_FENCE();
      }
    };
    if (ninstr > 0 && !(P.dbg & 1)) {
      S2Entry ea[kU], eb[kU];
      load(ea, 0);
      for (int p0 = 0; p0 < ninstr; p0 += 2 * kU) {
        const bool more = p0 + kU < ninstr;
        if (more) load(eb, p0 + kU);
        process(ea, p0);
        if (more) {
          if (p0 + 2 * kU < ninstr) load(ea, p0 + 2 * kU);
          process(eb, p0 + kU);
        }
      }
    }
    S2_FENCE();
    // ---- epilogue: q = [q +] acc + a0 p + c DtD p ; dot += p*q  (one row per lane group) ----
    constexpr int RPX = TY / G;  // instructions per x slab of the tile
    static_assert((TX * RPX) % 4 == 0, "fast epilogue unrolls four instructions");
    if (P.dbg & 2) continue;
    const bool fast_xy = pin != nullptr && !P.accumulate && x0 > 0 && y0 > 0 &&
                         x0 + TX < dd.x && y0 + TY < dd.y && dd.numel() < (1ull << 29);
    if (fast_xy) {
      // interior tiles (all x/y stencil neighbours inside the volume): buffer addressing with a
      // per-lane byte offset computed once per tile + scalar row offsets; 28 loads in flight
      const size_t sxe = (size_t)dd.y * dd.z, sye = dd.z;
      const int kc = min(z0 + gl, dd.z - 1);
      const bool actz = gl < ez, lzf = kc > 0, hzf = kc + 1 < dd.z;
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
          const int it0 = __builtin_amdgcn_readfirstlane(it);
          float c[4], vxp[4], vxm[4], vyp[4], vym[4], vzp[4], vzm[4], ob[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
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
            const int lx = (it0 + j) / RPX, ly2 = ((it0 + j) % RPX) * G;
            const unsigned ro = (unsigned)(lx + 1) * sxb + (unsigned)(ly2 + 1) * syb;
            float q = arow[(lx * SY + ly2) * SZ];
            const float xf = vxp[j] - c[j], xbk = c[j] - vxm[j], yf = vyp[j] - c[j], ybk = c[j] - vym[j];
            const float zf = (hzf ? vzp[j] : 0.f) - c[j], zbk = lzf ? c[j] - vzm[j] : 0.f;
            const float st = P.cx * (xbk - xf) + P.cy * (ybk - yf) + P.cz * (zbk - zf);
            q += P.a0 * c[j] + st;
            if (actz) {
              if (OBJ) {
                dot += (double)obj_term(q, ob[j], c[j]);
              } else {
                buf_store(q, rd, e0, ro);
                dot += (double)__fmul_rn(c[j], q);
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
  }
  if (P.partials) {
    const double tot = wave_sum(dot);
    if (lane == 0) P.partials[blockIdx.x * kS2Waves + wave] = tot;
  }
}

static int s2_grid(Dim3i dd) {
  const int nt = s2_ntiles(dd);
  static const int cap = getenv("UNIRES_SPLAT2_BLOCKS") ? atoi(getenv("UNIRES_SPLAT2_BLOCKS")) : 1024;
  const int want = (nt + kS2Waves - 1) / kS2Waves;
  return want < cap ? want : cap;
}

int splat2_blocks(Dim3i dd) { return s2_grid(dd) * kS2Waves; }  // partials written

int launch_splat2(const SplatSched &S, const float *src, size_t src_numel, const float4 *tab_dev, int gn,
                  unsigned tab_step, const Affine &A, float alpha, float tol, const PushEpilogue &ep, float *dst,
                  Dim3i dd, const int *done, hipStream_t st) {
  (void)tol;
  if (!S.valid || S.ntiles != s2_ntiles(dd)) return 1;
  if (S.axis >= 0 && !tab_dev) return 1;
  S2Args P;
  if (src_numel >= (1ull << 30)) return 1;  // 32-bit byte offsets into the source
  P.src = src;
  P.src_bytes = src_numel * sizeof(float);  // buffer range check: idle lanes may point past a row
  P.tab = tab_dev, P.gn = gn, P.tabn = S.axis >= 0 ? gn + S2Tile::L : 0, P.tab_step = tab_step;
  P.entries = S.entries, P.tile_off = S.tile_off, P.ntiles = S.ntiles;
  P.A = A, P.alpha = alpha;
  P.p = ep.p, P.a0 = ep.a0, P.cx = ep.cx, P.cy = ep.cy, P.cz = ep.cz;
  P.dst = dst, P.dd = dd, P.accumulate = ep.accumulate, P.partials = ep.partials, P.objb = ep.objb;
  static const int dbg = getenv("UNIRES_S2_DBG") ? atoi(getenv("UNIRES_S2_DBG")) : 0;
  P.dbg = dbg;
  const dim3 grid(s2_grid(dd)), block(kWave * kS2Waves);
  const size_t lds = S.axis >= 0 ? 3 * (size_t)P.tabn * sizeof(float) : 0;
  if (lds > 48 * 1024) return 1;
  switch (S.axis) {
    case 0: hipLaunchKernelGGL((k_splat2<0>), grid, block, lds, st, P, done); break;
    case 1: hipLaunchKernelGGL((k_splat2<1>), grid, block, lds, st, P, done); break;
    case 2: hipLaunchKernelGGL((k_splat2<2>), grid, block, lds, st, P, done); break;
    default: hipLaunchKernelGGL((k_splat2<-1>), grid, block, lds, st, P, done); break;
  }
  return 0;
}

}  // namespace unires
