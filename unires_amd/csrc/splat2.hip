// splat2.hip - schedule-driven owner-computes push (see splat2.hpp).
//
// Race freedom without atomics (LDS float atomics cost ~190 clocks per wave-instruction on
// gfx950, tools/mb_lds.hip): a tile is owned by ONE wave, whose LDS operations execute in
// order; inside one read-add-write group
//   * the lanes of a segment sit in DIFFERENT z planes (the build kernel cuts a row wherever two
//     consecutive points share floor(gz)), and a group touches one plane per lane: the lower
//     plane of every point first, the upper plane second;
//   * the segments packed into one instruction either come from rows >= row_sep apart
//     (host-checked for the affine: some coordinate of any two of their points differs by >= 2)
//     or occupy z-plane ranges that do not touch.
// The schedule is fixed, so results are bit-reproducible.
//
// What the hardware charges (tools/mb_valu.hip, tools/mb_lds.hip, MI355X): a wave64 add / mul
// issues in 2 clocks, FMA / convert / shift / compare in 4; an 8-cell read-add-write costs the CU's
// LDS ~43 clocks; every VMEM wave-instruction costs the texture addresser ~16 clocks however
// well it coalesces.  Hence: segment descriptors go through an LDS ring (one coalesced global load
// per 32 segments), not per-lane global loads; ONE 8-byte source load per lane and instruction.
#include "splat2.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace unires {

struct S2Tile {
  static constexpr int TX = 8, TY = 4, TZ = 30, L = 32;
  static constexpr int SX = TX + 2, SY = TY + 2, SZ = TZ + 2, N = SX * SY * SZ;
  static constexpr int XS = SY * SZ, YS = SZ;
};
constexpr int kS2Waves = 4;   // waves (= independent tiles in flight) per workgroup
constexpr int kS2MaxSeg = 8;  // segments per instruction

#define S2_FENCE() asm volatile("" ::: "memory")
#ifndef S2_INTERLEAVE
#define S2_INTERLEAVE 1
#endif
// phase-ablation switches: compiled in only with -DUNIRES_ABLATE (then UNIRES_S2_DBG selects bits:
// 1 no splat, 2 no epilogue, 4 no LDS updates, 8 no source loads); product builds carry none of it
#ifdef UNIRES_ABLATE
#define S2_ABL(bit) ((P.dbg & (bit)) != 0)
#else
#define S2_ABL(bit) (false)
#endif

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

// workgroups of the persistent launch; `share_cap` > 0: the caller runs other work next to it (channels of a y-update
// on streams of their own) and wants room left on the CUs - see unires_plan_set_concurrency (api.hip)
static int s2_grid(Dim3i dd, int share_cap = 0) {
  const int nt = s2_ntiles(dd);
  static const int cap_env = getenv("UNIRES_SPLAT2_BLOCKS") ? atoi(getenv("UNIRES_SPLAT2_BLOCKS")) : 0;
  const int cap = cap_env > 0 ? cap_env : (share_cap >= 8 ? share_cap : 1024);
  const int want = (nt + kS2Waves - 1) / kS2Waves;
  return want < cap ? want : cap;
}


// --------------------------------------------------------------------------
// schedule build
// --------------------------------------------------------------------------
struct S2BuildArgs {
  Affine A, Ainv;
  Dim3i gd, dd;
  float tol;
  int row_sep;
  int axis, rows_y;
  const float4 *xtab, *ytab;  // axis 3
  Dim3i xd;
  int exact;  // close rows tested point by point (aligned schedules) instead of kept apart wholesale
};

struct S2Seg {
  short ui, uj, k0;
  unsigned char len, lzmin, lzmax, pad;  // local z planes touched: [lzmin, lzmax + 1]
};

__device__ __forceinline__ unsigned s2_rowcode(const S2BuildArgs &B, int ui, int uj) {
  return (B.axis == 0 || B.axis == 1 || B.axis == 3) ? ((unsigned)ui << 9) | (unsigned)uj
                                      : (unsigned)ui * (unsigned)B.rows_y + (unsigned)uj;
}

// One wave per tile.  MODE 0: counts[t] = {entries, instructions}; MODE 1: counts holds the exclusive prefix
// sums and the entries / masks are written where they belong (the two-pass build of rounds 2-4, kept as the
// fallback); MODE 2 (round 5): ONE pass - counts[t] as in MODE 0 AND the tile's entries / masks into its staging
// slot (kS2StageE entries, 64 masks per tile: entries, ext and masks point at the staging arrays), from where
// k_splat2_compact moves them once the host has fixed the processing order.  A tile with more entries than a slot
// holds raises bit 1 of *err: the caller falls back to MODE 1.
constexpr int kS2StageE = 128;
template <int MODE>
__global__ void __launch_bounds__(kWave)
    k_splat2_build(S2BuildArgs B, uint2 *__restrict__ counts, const int *__restrict__ geom,
                   S2Entry *__restrict__ entries, S2Ext *__restrict__ ext, ulonglong2 *__restrict__ masks,
                   int *__restrict__ err, unsigned long long *__restrict__ stats) {
  using T = S2Tile;
  constexpr int L = T::L, kSegs = 384;
  __shared__ S2Seg segs[kSegs];
  __shared__ unsigned short member[kWave][kS2MaxSeg];
  const int lane = threadIdx.x;
  const Dim3i dd = B.dd;
  const int slot = blockIdx.x;                  // where this tile's counts / offsets live
  const int t = geom ? geom[slot] : slot;       // the output tile (FILL: processing order -> tile)
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
      float za = 0.f, zb = 0.f;
      if (more) {
        float gx, gy, gz;
        s2_point(B.A, rb.x, rb.y, rb.z, (float)cur, gx, gy, gz);
        float plz = floorf(gz);
        za = plz;
        while (e + 1 <= k1 && e + 1 - cur < L) {
          s2_point(B.A, rb.x, rb.y, rb.z, (float)(e + 1), gx, gy, gz);
          const float lz = floorf(gz);
          if (lz == plz) break;
          plz = lz;
          ++e;
        }
        zb = plz;
      }
      const unsigned long long m = __ballot(more);
      const int pos = nseg + __popcll(m & lt_mask);
      if (more && pos < kSegs) {
        const int la = (int)za - (g.z0 - 1), lb = (int)zb - (g.z0 - 1);
        segs[pos] = S2Seg{(short)ui, (short)uj, (short)cur, (unsigned char)(e - cur + 1),
                          (unsigned char)min(la, lb), (unsigned char)max(la, lb), 0};
      }
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
  // (r6) the slice of the conv_up table this tile needs: 64 consecutive entries from `kb` on - the tile's first grid
  // index along the table's axis (grid z for axis 2 / 3, ui / uj for axis 0 / 1).  The splat keeps one slice per wave in
  // LDS instead of the whole table per workgroup (6 - 8 KB: config 4's 385 entries pushed four 4-wave workgroups over
  // a CU's 160 KB).  Axis 2 / 3 entries carry k0 relative to kb.  A tile that spans more than 64: general kernels.
  int kb = 0;
  if (B.axis >= 0) {
    int lo = 1 << 30, hi = -(1 << 30);
    for (int s0 = 0; s0 < nseg; s0 += kWave) {
      const int s = s0 + lane;
      if (s < nseg) {
        const S2Seg q = segs[s];
        const int a = B.axis == 0 ? (int)q.ui : (B.axis == 1 ? (int)q.uj : (int)q.k0);
        const int b = (B.axis == 0 || B.axis == 1) ? a : a + (int)q.len - 1;
        lo = min(lo, a), hi = max(hi, b);
      }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) lo = min(lo, __shfl_xor(lo, off, kWave)), hi = max(hi, __shfl_xor(hi, off, kWave));
    if (nseg > 0) {
      kb = lo;
      if (hi - lo >= kWave && lane == 0) atomicOr(err, 1);
    }
  }
  const int kb_k = (B.axis == 2 || B.axis == 3) ? kb : 0;  // what the entries' k field is relative to
  // ---- packing into 64-lane instructions: lane b is instruction b of the tile ----
  // ALIGNED mode (every segment walks one z plane per point, i.e. |dz/dk| <= 1): within each
  // 32-lane half, lane position = z plane (31 - plane when z decreases along the row).  The
  // accumulator's x / y strides are multiples of 32 words, so the LDS bank of an update is its z
  // plane: lanes of one half then never collide on a bank, whatever rows they come from.
  // Otherwise segments are simply laid end to end.  Either way a segment joins the first
  // instruction that has room, fewer than kS2MaxSeg members and no member it could collide with
  // (rows closer than row_sep whose plane ranges touch).
  __shared__ unsigned short order[kSegs];
  // (lane 0's work arrays and the lanes' start lists live in LDS - the block is one wave; indexed at run time they
  // sat in scratch memory: 928 bytes per lane, build kernels 494 + 678 -> 454 + 556 us at 256^3)
  __shared__ int cnt[33];
  __shared__ unsigned short tmp[kSegs];
  __shared__ int start_all[kWave][kS2MaxSeg];
  bool al = true;
  for (int s0 = 0; s0 < nseg; s0 += kWave) {
    const int s = s0 + lane;
    if (s < nseg) al = al && ((int)segs[s].lzmax - (int)segs[s].lzmin + 1 == (int)segs[s].len);
  }
  const bool aligned = __all(al) && T::SZ == 32;
  const bool zdown = c2 < 0.f;
  auto seg_pos = [&](const S2Seg q) { return aligned ? (zdown ? 31 - (int)q.lzmax : (int)q.lzmin) : 0; };
  // stable counting sort by first lane position (one lane: deterministic schedule)
  if (lane == 0) {
    for (int i = 0; i < 33; ++i) cnt[i] = 0;
    for (int s = 0; s < nseg; ++s) ++cnt[seg_pos(segs[s]) + 1];
    for (int i = 0; i < 32; ++i) cnt[i + 1] += cnt[i];
    for (int s = 0; s < nseg; ++s) order[cnt[seg_pos(segs[s])]++] = (unsigned short)s;
#if S2_INTERLEAVE
    // ... offered to the first-fit below as 0, h, 1, h + 1, ... (h = half the count).  Most segments are
    // whole rows of the tile's ~9 x 5 cross-section - one per 32-lane half - and two rows share an
    // instruction only if they are >= row_sep apart: in scan order a row's successors are its
    // neighbours, the partner had to be found among rows offered much later, and the last ones found
    // none (26 instructions per config-3 tile for 44 segments).  Offered in this order the row half the
    // cross-section away - 4 to 5 rows over - comes right behind its partner.
    {
      const int h = (nseg + 1) / 2;
      for (int s = 0; s < nseg; ++s) tmp[s] = order[s];
      for (int s = 0; s < nseg; ++s) order[s] = tmp[(s & 1) ? h + (s >> 1) : (s >> 1)];
    }
#endif
  }
  S2_FENCE();
  __syncthreads();
  unsigned occ0 = 0u, occ1 = 0u;  // aligned: occupied lanes of each half; else: lanes used so far in occ0
  int nmem = 0, nbins = 0;
  for (int si = 0; si < nseg; ++si) {
    const int s = order[si];
    const S2Seg q = segs[s];
    const int a = seg_pos(q);
    const unsigned pm = (q.len >= 32 ? 0xffffffffu : ((1u << q.len) - 1u)) << a;
    bool free = true;
    for (int j = 0; j < nmem && free; ++j) {
      const S2Seg m = segs[member[lane][j] & 0x7fff];
      const bool rows_close = max(abs(m.ui - q.ui), abs(m.uj - q.uj)) < B.row_sep;
      const bool planes_touch = (int)m.lzmin <= (int)q.lzmax + 1 && (int)q.lzmin <= (int)m.lzmax + 1;
      if (!(rows_close && planes_touch)) continue;
      // (adjacent rows - at most one apart in both indices - overlap wherever they share a plane: no need to look)
      if (!(B.exact && aligned && !zdown) || max(abs(m.ui - q.ui), abs(m.uj - q.uj)) < 2) {
        free = false;
        break;
      }
      // (r5, as ata1.hip's packing) ... else exactly: two lanes meet only in the same read-add-write group on the
      // same z plane - two points with the same floor plane whose 2 x 2 cell footprints overlap.  Aligned segments
      // walk one plane per point, so the points that share plane pl are known: their floor cells are compared, in
      // the kernel's own arithmetic.  Rows two apart, which the blanket rule (row_sep 3 for a rotated operator)
      // keeps out of each other's instructions, almost always pass.
      const int lo = max((int)m.lzmin, (int)q.lzmin), hi = min((int)m.lzmax, (int)q.lzmax);
      const RowBase ra = affine_row(B.A, (float)m.ui, (float)m.uj), rq = affine_row(B.A, (float)q.ui, (float)q.uj);
      for (int pl = lo; pl <= hi; ++pl) {
        float ax, ay, az, bx, by, bz;
        s2_point(B.A, ra.x, ra.y, ra.z, (float)((int)m.k0 + pl - (int)m.lzmin), ax, ay, az);
        s2_point(B.A, rq.x, rq.y, rq.z, (float)((int)q.k0 + pl - (int)q.lzmin), bx, by, bz);
        if (fabsf(floorf(ax) - floorf(bx)) < 2.f && fabsf(floorf(ay) - floorf(by)) < 2.f) {
          free = false;
          break;
        }
      }
    }
    free = free && lane <= nbins && nmem < kS2MaxSeg;
    bool ok0, ok1;
    if (aligned) {
      ok0 = free && (occ0 & pm) == 0u, ok1 = free && (occ1 & pm) == 0u;
    } else {
      ok0 = free && (int)occ0 + q.len <= kWave, ok1 = false;
    }
    const unsigned long long m = __ballot(ok0 || ok1);
    if (m == 0ull) {  // more than 64 instructions in one tile
      if (lane == 0) atomicOr(err, 1);
      break;
    }
    const int chosen = __ffsll((long long)m) - 1;
    if (lane == chosen) {
      const int half = ok0 ? 0 : 1;
      member[lane][nmem++] = (unsigned short)(s | (half << 15));
      if (aligned) {
        if (half == 0) occ0 |= pm; else occ1 |= pm;
      } else {
        occ0 += q.len;
      }
    }
    nbins = max(nbins, chosen + 1);
  }
  const int nent = lane < nbins ? nmem : 0;
  // inclusive prefix sum of nent over lanes
  int incl = nent;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int v = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += v;
  }
  const int total_ent = __shfl(incl, kWave - 1, kWave);
  if (MODE != 1) {
    if (lane == 0) counts[slot] = make_uint2((unsigned)total_ent, (unsigned)nbins | ((unsigned)kb << 8));
    if (MODE == 0) return;
    if (total_ent > kS2StageE) {
      if (lane == 0) atomicOr(err, 2);
      return;
    }
  }
  const uint2 base = MODE == 1 ? counts[slot] : make_uint2((unsigned)slot * (unsigned)kS2StageE, (unsigned)slot * 64u);
  unsigned long long pts = 0;
  if (lane < nbins) {
    S2Entry *out = entries + base.x + (incl - nent);
    // members in lane order (aligned: by half, then position; else: as packed)
    int *start_of = start_all[lane];
    int run = 0;
    for (int j = 0; j < nmem; ++j) {
      const S2Seg q = segs[member[lane][j] & 0x7fff];
      start_of[j] = aligned ? 32 * (member[lane][j] >> 15) + seg_pos(q) : run;
      run += q.len;
    }
    for (int i = 1; i < nmem; ++i)  // insertion sort of (start, member) pairs
      for (int j = i; j > 0 && start_of[j] < start_of[j - 1]; --j) {
        const int ts = start_of[j];
        start_of[j] = start_of[j - 1], start_of[j - 1] = ts;
        const unsigned short tm = member[lane][j];
        member[lane][j] = member[lane][j - 1], member[lane][j - 1] = tm;
      }
    unsigned long long starts = 0ull, active = 0ull;
    for (int j = 0; j < nmem; ++j) {
      const S2Seg q = segs[member[lane][j] & 0x7fff];
      const int start = start_of[j];
      const RowBase rb = affine_row(B.A, (float)q.ui, (float)q.uj);
      S2Entry e;
      e.rx = rb.x, e.ry = rb.y, e.rz = rb.z;
      e.pk = s2_rowcode(B, q.ui, q.uj) | ((unsigned)(q.k0 - kb_k - start + 64) << kS2RowBits);
      out[j] = e;
      if (B.axis == 3) {  // x / y part of the conv_up of this row: 2 x 2 x-space columns
        const float4 tx = B.xtab[q.ui], ty = B.ytab[q.uj];
        S2Ext x;
        x.base4 = 4u * (((unsigned)__float_as_int(tx.x) * (unsigned)B.xd.y + (unsigned)__float_as_int(ty.x)) *
                        (unsigned)B.xd.z);
        x.w00 = tx.y * ty.y, x.w01 = tx.y * ty.z, x.w10 = tx.z * ty.y, x.w11 = tx.z * ty.z;
        x.pad[0] = x.pad[1] = x.pad[2] = 0u;
        ext[base.x + (incl - nent) + j] = x;
      }
      if (j > 0) starts |= 1ull << (start - 1);  // lanes >= start count it: slot = popcount below lane
      active |= (q.len >= 64 ? ~0ull : ((1ull << q.len) - 1ull)) << start;
      pts += q.len;
    }
    masks[base.y + lane] = make_ulonglong2(starts, active);
  }
  if (stats) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) pts += __shfl_down(pts, off, kWave);
    if (lane == 0) {
      atomicAdd(stats, pts);
      atomicAdd(stats + 1, (unsigned long long)nbins);
    }
  }
}

// Staging slot of tile geom[u] -> the schedule's arrays at processing slot u's offsets.
__global__ void __launch_bounds__(kWave)
    k_splat2_compact(const uint2 *__restrict__ stage_cnt, const S2Entry *__restrict__ stage_e,
                     const S2Ext *__restrict__ stage_x, const ulonglong2 *__restrict__ stage_m,
                     const int *__restrict__ geom, const uint2 *__restrict__ tile_off, S2Entry *__restrict__ entries,
                     S2Ext *__restrict__ ext, ulonglong2 *__restrict__ masks) {
  const int u = blockIdx.x, lane = threadIdx.x;
  const int t = geom[u];
  const uint2 c = stage_cnt[t], o = tile_off[u];
  const uint4 *se = reinterpret_cast<const uint4 *>(stage_e) + (size_t)t * kS2StageE;
  uint4 *de = reinterpret_cast<uint4 *>(entries) + o.x;
  for (unsigned i = lane; i < c.x; i += kWave) de[i] = se[i];
  if (stage_x) {
    const uint4 *sx = reinterpret_cast<const uint4 *>(stage_x) + 2 * (size_t)t * kS2StageE;  // 32 bytes per entry
    uint4 *dx = reinterpret_cast<uint4 *>(ext) + 2 * (size_t)o.x;
    for (unsigned i = lane; i < 2u * c.x; i += kWave) dx[i] = sx[i];
  }
  if ((unsigned)lane < (c.y & 0xffu)) masks[o.y + lane] = stage_m[(size_t)t * 64 + lane];  // (.y: instructions | table base << 8)
}

// Staging arrays of the single-pass build: one set per device, grown on demand, shared by every schedule built on it
// (a build holds the lock from its first kernel to its last: builds synchronise with the host anyway).
struct S2Stage {
  uint2 *cnt = nullptr;
  S2Entry *e = nullptr;
  S2Ext *x = nullptr;
  ulonglong2 *m = nullptr;
  size_t tiles = 0, tiles_x = 0;
};
static std::mutex g_stage_mu;
static std::map<int, S2Stage> g_stage;

static S2Stage *s2_stage(int nt, bool want_ext) {  // (call with g_stage_mu held) nullptr: use the two-pass build
  static const bool two_pass = getenv("UNIRES_S2_BUILD_2PASS") != nullptr;
  if (two_pass) return nullptr;
  const size_t bytes = (size_t)nt * (kS2StageE * (sizeof(S2Entry) + (want_ext ? sizeof(S2Ext) : 0)) + 64 * sizeof(ulonglong2));
  if (bytes > (512ull << 20)) return nullptr;  // (a 1 000^3 volume: two passes rather than half a gigabyte of staging)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  S2Stage &G = g_stage[dev];
  if ((size_t)nt > G.tiles) {
    if (G.cnt) (void)hipFree(G.cnt);
    if (G.e) (void)hipFree(G.e);
    if (G.m) (void)hipFree(G.m);
    G.cnt = nullptr, G.e = nullptr, G.m = nullptr, G.tiles = 0;
    const size_t n = (size_t)nt + nt / 8;
    if (hipMalloc((void **)&G.cnt, n * sizeof(uint2)) != hipSuccess ||
        hipMalloc((void **)&G.e, n * kS2StageE * sizeof(S2Entry)) != hipSuccess ||
        hipMalloc((void **)&G.m, n * 64 * sizeof(ulonglong2)) != hipSuccess) {
      (void)hipGetLastError();
      if (G.cnt) (void)hipFree(G.cnt);
      if (G.e) (void)hipFree(G.e);
      if (G.m) (void)hipFree(G.m);
      G.cnt = nullptr, G.e = nullptr, G.m = nullptr;
      return nullptr;
    }
    G.tiles = n;
  }
  if (want_ext && (size_t)nt > G.tiles_x) {
    if (G.x) (void)hipFree(G.x);
    G.x = nullptr, G.tiles_x = 0;
    const size_t n = (size_t)nt + nt / 8;
    if (hipMalloc((void **)&G.x, n * kS2StageE * sizeof(S2Ext)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    G.tiles_x = n;
  }
  return &G;
}

static int s2_active(int axis, int grid);
static thread_local bool t_thorough = true;
void sched_set_thorough(bool on) { t_thorough = on; }
bool sched_thorough() { return t_thorough; }

void splat2_free(SplatSched &S) {
  if (S.entries) (void)hipFree(S.entries);
  if (S.ext) (void)hipFree(S.ext);
  if (S.masks) (void)hipFree(S.masks);
  if (S.tile_off) (void)hipFree(S.tile_off);
  if (S.tile_geom) (void)hipFree(S.tile_geom);
  if (S.recs) (void)hipFree(S.recs);
  if (S.scratch) (void)hipFree(S.scratch);
  S = SplatSched();
}

int splat2_build(SplatSched &S, const Affine &A, const Affine &Ainv, Dim3i gd, Dim3i dd, float tol,
                 const SplatSafety &safe, int axis, int rows_y, const float4 *xtab, const float4 *ytab,
                 Dim3i xd) {
  S.valid = false;
  static const bool off = getenv("UNIRES_NO_SPLAT2") != nullptr;
  static const bool verbose = getenv("UNIRES_SPLAT2_VERBOSE") != nullptr;
  if (off) return 1;
  if (safe.use_atomics) return 1;
  if (dd.x > 4000 || dd.y > 4000 || dd.z > 4000 || gd.x > 4000 || gd.y > 4000 || gd.z > 4000) return 1;
  if (!fits_fast_index(gd) || !fits_fast_index(dd) || dd.numel() >= (1ull << 30)) return 1;
  // the row code must fit its 19 bits (all ones is reserved)
  if (axis == 3 && (!xtab || !ytab)) return 1;
  if (axis == 0 || axis == 1 || axis == 3) {
    if (gd.x > 1023 || gd.y > 511) return 1;
  } else if ((long long)gd.x * rows_y >= (long long)kS2RowIdle || rows_y < gd.y) {
    return 1;
  }
  const int nt = s2_ntiles(dd);
  if ((size_t)nt + 1 > S.cap_tiles) {
    if (S.tile_off) (void)hipFree(S.tile_off);
    S.tile_off = nullptr;
    if (hipMalloc((void **)&S.tile_off, ((size_t)nt + 1) * sizeof(uint2)) != hipSuccess) return 1;
    if (S.tile_geom) (void)hipFree(S.tile_geom);
    S.tile_geom = nullptr;
    if (hipMalloc((void **)&S.tile_geom, ((size_t)nt + 1) * sizeof(int)) != hipSuccess) return 1;
    S.cap_tiles = (size_t)nt + 1;
  }
  if (!S.scratch && hipMalloc((void **)&S.scratch, 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
  (void)hipMemset(S.scratch, 0, 4 * sizeof(unsigned long long));
  int *err_dev = (int *)S.scratch;
  unsigned long long *stats_dev = S.scratch + 1;
  S2BuildArgs B;
  B.A = A, B.Ainv = Ainv, B.gd = gd, B.dd = dd, B.tol = tol, B.row_sep = safe.row_sep;
  B.axis = axis, B.rows_y = rows_y;
  B.xtab = xtab, B.ytab = ytab, B.xd = xd;
  static const int exact = getenv("UNIRES_S2_EXACT") ? atoi(getenv("UNIRES_S2_EXACT")) : -1;  // (-1: as the plan asks)
  B.exact = exact >= 0 ? exact : (sched_thorough() ? 1 : 0);
  // One pass into per-tile staging slots + a compaction once the order is known (round 5; the schedule is the
  // two-pass build's, bit for bit - tests/test_gpu_selection.py); two passes where there is no staging or a tile
  // does not fit its slot.  unires_plan_set_repeat runs this for every channel after every rigid update.
  std::lock_guard<std::mutex> stage_lock(g_stage_mu);
  S2Stage *stage = s2_stage(nt, axis == 3);
  std::vector<uint2> cnt((size_t)nt), h((size_t)nt + 1);
  if (stage) {
    hipLaunchKernelGGL(k_splat2_build<2>, dim3(nt), dim3(kWave), 0, 0, B, stage->cnt, (const int *)nullptr, stage->e,
                       axis == 3 ? stage->x : (S2Ext *)nullptr, stage->m, err_dev, stats_dev);
    int e0 = 0;
    if (hipMemcpy(cnt.data(), stage->cnt, (size_t)nt * sizeof(uint2), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (hipMemcpy(&e0, err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (e0 & 2) {  // a tile overflowed its slot: two passes (the counts are good)
      stage = nullptr;
      (void)hipMemset(S.scratch, 0, 4 * sizeof(unsigned long long));
      if (e0 & 1) (void)hipMemset(err_dev, 1, 1);
    }
  } else {
    hipLaunchKernelGGL(k_splat2_build<0>, dim3(nt), dim3(kWave), 0, 0, B, S.tile_off, (const int *)nullptr,
                       (S2Entry *)nullptr, (S2Ext *)nullptr, (ulonglong2 *)nullptr, err_dev,
                       (unsigned long long *)nullptr);
    if (hipMemcpy(cnt.data(), S.tile_off, (size_t)nt * sizeof(uint2), hipMemcpyDeviceToHost) != hipSuccess)
      return 1;
  }
  // (the build kernel packs the tile's table base above its instruction count)
  std::vector<int> kbv((size_t)nt);
  for (int i = 0; i < nt; ++i) kbv[i] = (int)(cnt[i].y >> 8), cnt[i].y &= 0xffu;
  // Processing order.  (1) Contiguous runs of tiles per XCD with equal cost: a tile costs its
  // instructions + 11 for the epilogue (~5 us against ~0.45 us per instruction, tools/s2_timeline.py);
  // with equal tile COUNTS the XCD that holds the volume's first x slabs had a third less to do.
  // (2) Inside a run, the tiles with (next to) no instructions - the part of the volume the
  // observation does not see, 14 % of config 3 - go LAST, emptiest at the very end: a wave gets 4.5
  // tiles, so the last round is half empty and as long as its longest tile; made of 5 us tiles
  // instead of 16 us ones the kernel's tail shrinks by two thirds.  The other tiles keep their index
  // order (neighbours share halos and schedule lines in the XCD's L2).
  constexpr double kEpilogueCost = 11.0;
  std::vector<int> geom((size_t)nt + 1);
  // the launch this schedule is laid out for: `nwg` workgroups take tiles, partition x (an XCD when there are >= 8)
  // is served by the workgroups b with b % np == x
  const int nwg = s2_active(axis, s2_grid(dd));
  const int np = std::min(8, std::max(nwg, 1));
  std::vector<int> part_lo((size_t)np + 1, nt);
  {
    double total = 0.0;
    unsigned imax = 0;
    for (int i = 0; i < nt; ++i) total += (double)cnt[i].y + kEpilogueCost, imax = std::max(imax, cnt[i].y);
    int x = 1;
    double cum = 0.0;
    part_lo[0] = 0;
    for (int i = 0; i < nt && x < np; ++i) {
      cum += (double)cnt[i].y + kEpilogueCost;
      while (x < np && cum >= total * x / (double)np) part_lo[x++] = i + 1;
    }
    for (; x <= np; ++x) part_lo[x] = nt;
    static const bool keep_order = getenv("UNIRES_SPLAT2_INDEX_ORDER") != nullptr;
    const unsigned cheap = keep_order ? 0u : imax / 2u;
    // (3) y strips.  In index order (z fastest, then y, then x) a tile's x neighbour comes a whole yz slab of
    // tiles later: 8 x 256 x 256 floats of p = 2.1 MB at 256^3, 4.7 MB at 384^3 - with q, the x-space rows and the
    // schedule lines that go by in between, more than the XCD's 4 MB of L2 keeps: the x halos of the epilogue's
    // stencil window came from the fabric again.  Walking the run in strips of `strip` tiles along y (z fastest,
    // then y inside the strip, then x, then the next strip) puts the x neighbour one strip-slab (~1 MB) away; only
    // the strips' border rows are fetched twice.  Measured (channel 1, PMC): k_splat2 fetches 169.7 -> 143.4 MB
    // per launch at config 3, 677 -> 561 MB at 384^3; its time does not move (it is not bandwidth-bound) - this is
    // about not wasting fabric traffic.  (The same walk on k_ata1's 4 x 4 tiles: no change, 1 MB slabs fit anyway.)
    using T = S2Tile;
    const int nty = (dd.y + T::TY - 1) / T::TY, ntz = (dd.z + T::TZ - 1) / T::TZ;
    static const int strip_env = getenv("UNIRES_S2_STRIP") ? atoi(getenv("UNIRES_S2_STRIP")) : -1;
    int strip = strip_env >= 0 ? strip_env
                               : (int)std::max<long long>(4, (1ll << 20) / ((long long)T::TX * T::TY * dd.z * 8));
    if (strip <= 0 || strip >= nty || keep_order) strip = nty;  // (= index order)
    for (int xc = 0; xc < np; ++xc) {
      const int lo = part_lo[xc], hi = part_lo[xc + 1];
      int u = lo;
      if (strip < nty && hi > lo) {
        // (generated, not sorted: this runs inside every unires_plan_set_repeat - a rigid update per channel and
        // ADMM iteration - and a comparison sort of 17 k tiles cost 1.5 ms there)
        const int tx_lo = lo / (ntz * nty), tx_hi = (hi - 1) / (ntz * nty);
        for (int s0 = 0; s0 < nty; s0 += strip)
          for (int txi = tx_lo; txi <= tx_hi; ++txi)
            for (int tyi = s0; tyi < std::min(s0 + strip, nty); ++tyi) {
              const int base = (txi * nty + tyi) * ntz;
              for (int g = std::max(base, lo); g < std::min(base + ntz, hi); ++g)
                if (cnt[g].y > cheap) geom[u++] = g;
            }
      } else {
        for (int g = lo; g < hi; ++g)
          if (keep_order || cnt[g].y > cheap) geom[u++] = g;
      }
      const int first_cheap = u;
      for (int g = lo; g < hi; ++g)
        if (!keep_order && cnt[g].y <= cheap) geom[u++] = g;
      std::stable_sort(geom.begin() + first_cheap, geom.begin() + hi,
                       [&](int a, int b) { return cnt[a].y > cnt[b].y; });
    }
    geom[nt] = 0;
  }
  // (4) Who takes which tile: wave slot s of a partition (S slots) walks positions s, s + S, s + 2 S, ... of the order
  // above - waves sweep the run together, and a wave knows its next tile a tile ahead (its header is prefetched).
  // Round 6 measured two ways of levelling the waves' ends (4.33 tiles per wave at config 3: a third of the waves have a
  // fifth 14 us tile, the rest end at 59 of 72 us, and a wave's end follows the predicted sum of its tiles' costs,
  // correlation 0.94 over 4 096 waves - tools/s2_timeline.py) and dropped both:
  //  * dealing the last (partial, or also the last whole) round AT BUILD TIME to the slots with the least predicted
  //    work, longest first: 72.2 us where round-robin has 70.2 on the unrotated channel, no change on the rotated ones;
  //  * a ticket queue for the tiles that do not fill a round (one L2 atomic per draw, 16 counters per XCD - atomics on
  //    ONE address are served ~30 ns apart: 512 waves drawing from one counter cost the kernel 17 - 50 us): 72.1 / 72.4 us
  //    against 71.2 / 68.7 static.
  // A CU's 16 waves share its issue slots: what a wave does not use the others get, so what counts is the work per CU -
  // which round-robin over workgroups that the dispatcher deals over the CUs levels by itself.
  std::vector<int> pos_tile(geom.begin(), geom.begin() + nt);  // position -> tile
  for (int xc = 0; xc <= 8; ++xc) S.pos_lo[xc] = part_lo[(size_t)std::min(xc, np)];
  S.nwg = nwg;
  // offsets in position order (geom: the u-th tile in that order, for the compaction / fill kernels)
  std::vector<uint4> recs(pos_tile.size() + 1, make_uint4(0xffffffffu, 0u, 0u, 0u));
  unsigned re = 0, ri = 0;
  {
    using T = S2Tile;
    const int nty = (dd.y + T::TY - 1) / T::TY, ntz = (dd.z + T::TZ - 1) / T::TZ;
    int u = 0;
    for (size_t pp = 0; pp < pos_tile.size(); ++pp) {
      const int g = pos_tile[pp];
      if (g < 0) continue;
      const uint2 cg = cnt[g];
      const unsigned tzi = (unsigned)(g % ntz), tyi = (unsigned)((g / ntz) % nty), txi = (unsigned)(g / (ntz * nty));
      recs[pp] = make_uint4(txi | (tyi << 10) | (tzi << 20), re, ri, cg.y | ((unsigned)kbv[g] << 8));
      geom[u] = g;
      h[u] = make_uint2(re, ri);
      re += cg.x, ri += cg.y;
      ++u;
    }
    if (u != nt) return 1;  // (every tile exactly once)
    h[nt] = make_uint2(re, ri);
    geom[nt] = 0;
  }
  constexpr size_t kPad = 160;  // entries read (never used) past the end by the ring prefetch
  if ((size_t)re + kPad > S.cap_entries) {
    if (S.entries) (void)hipFree(S.entries);
    S.entries = nullptr;
    const size_t cap = (size_t)re + re / 8 + kPad;
    if (hipMalloc((void **)&S.entries, cap * sizeof(S2Entry)) != hipSuccess) return 1;
    if (S.ext) (void)hipFree(S.ext);
    S.ext = nullptr;
    S.cap_entries = cap;
  }
  if (axis == 3 && !S.ext) {
    if (hipMalloc((void **)&S.ext, S.cap_entries * sizeof(S2Ext)) != hipSuccess) return 1;
    (void)hipMemset(S.ext, 0, S.cap_entries * sizeof(S2Ext));
  }
  if ((size_t)ri + 8 > S.cap_instr) {
    if (S.masks) (void)hipFree(S.masks);
    S.masks = nullptr;
    const size_t cap = (size_t)ri + ri / 8 + 8;
    if (hipMalloc((void **)&S.masks, cap * sizeof(ulonglong2)) != hipSuccess) return 1;
    S.cap_instr = cap;
  }
  if (hipMemcpy(S.tile_off, h.data(), ((size_t)nt + 1) * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess)
    return 1;
  if (hipMemcpy(S.tile_geom, geom.data(), ((size_t)nt + 1) * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return 1;
  if (recs.size() > S.cap_recs) {
    if (S.recs) (void)hipFree(S.recs);
    S.recs = nullptr;
    const size_t cap = recs.size() + recs.size() / 8;
    if (hipMalloc((void **)&S.recs, cap * sizeof(uint4)) != hipSuccess) return 1;
    S.cap_recs = cap;
  }
  if (hipMemcpy(S.recs, recs.data(), recs.size() * sizeof(uint4), hipMemcpyHostToDevice) != hipSuccess) return 1;
  (void)hipMemset(S.entries + re, 0xff, kPad * sizeof(S2Entry));
  (void)hipMemset(S.masks + ri, 0, 8 * sizeof(ulonglong2));
  if (stage)
    hipLaunchKernelGGL(k_splat2_compact, dim3(nt), dim3(kWave), 0, 0, (const uint2 *)stage->cnt, (const S2Entry *)stage->e,
                       axis == 3 ? (const S2Ext *)stage->x : (const S2Ext *)nullptr, (const ulonglong2 *)stage->m,
                       (const int *)S.tile_geom, (const uint2 *)S.tile_off, S.entries, S.ext, S.masks);
  else
    hipLaunchKernelGGL(k_splat2_build<1>, dim3(nt), dim3(kWave), 0, 0, B, S.tile_off, (const int *)S.tile_geom,
                       S.entries, S.ext, S.masks, err_dev, stats_dev);
  int herr = 0;
  unsigned long long hs[2] = {0, 0};
  if (hipMemcpy(&herr, err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (hipMemcpy(hs, stats_dev, sizeof(hs), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (herr & 1) {
    if (verbose) fprintf(stderr, "[splat2] a tile exceeds the segment / instruction lists: general kernel used\n");
    return 1;
  }
  S.ntiles = nt;
  S.axis = axis;
  S.fill = hs[1] ? (double)hs[0] / (64.0 * (double)hs[1]) : 0.0;
  S.valid = true;
  if (verbose)
    fprintf(stderr, "[splat2] %d tiles, %llu instructions, %u segments, %llu points (%.2f per output voxel), "
            "lane fill %.3f, schedule %.1f MB, row_sep %d\n", nt, hs[1], re, hs[0],
            (double)hs[0] / (double)dd.numel(), S.fill, (re * sizeof(S2Entry) + ri * 16.0) / 1e6, safe.row_sep);
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
  const float4 *tab;     // conv_up table (AXIS >= 0)
  int gn;                // entries in it
  unsigned row_stride4;  // bytes per unit of the row code (see launch_splat2)
  unsigned tab_step4;    // bytes between the two x-space values of a grid voxel
  const S2Entry *entries;
  size_t ent_bytes;      // (bytes behind it: LDS-DMA goes through a buffer resource)
  const S2Ext *ext;      // AXIS 3
  unsigned xs_sy4, xs_sx4;  // AXIS 3: bytes between x-space rows / slabs
  const ulonglong2 *masks;  // per instruction: {segment starts, active lanes}
  const uint4 *recs;     // per position of the walk: {tile, entry offset, instruction offset, instructions | table base << 8}
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
  int dbg;  // UNIRES_S2_DBG ablation bits (read only by -DUNIRES_ABLATE builds)
  int prio_rot;  // rotate the waves' issue priority per tile (UNIRES_S2_PRIO=0 switches it off)
  int xlo[9];  // position range [xlo[x], xlo[x + 1]) of partition x (an XCD when the grid has >= 8 workgroups)
  int active;  // workgroups that take tiles (the rest of the grid only clears its partials)
  unsigned long long *prof;  // -DUNIRES_S2_PROF builds: per-wave timeline (100 MHz ticks)
};

// OBJK: the launch evaluates the CG objective (P.objb set: q is not stored) - a kernel of its own since r6, so that the
// sixteen extra rows its epilogue holds do not decide the register allocation of the form every CG iteration runs
template <int AXIS, int NW, bool OBJK>
__global__ void __launch_bounds__(kWave *NW) __attribute__((amdgpu_waves_per_eu(4, 4))) k_splat2(S2Args P, const int *__restrict__ done) {
  if (done && *done) return;
  using T = S2Tile;
  constexpr int TX = T::TX, TY = T::TY, L = T::L, G = kWave / L;
  constexpr int SY = T::SY, SZ = T::SZ, N = T::N, XS = T::XS, YS = T::YS;
  constexpr bool CONV = AXIS >= 0;
  // (one struct, so that the layout is the one written: each wave's ring and table slice are 1 KB on a 1 KB boundary,
  // and slot (i & 63) of either is  base | ((i << 4) & 0x3f0)  - an add-shift and an and-or where the indexed form costs
  // an and, a shift-add and, for the ring, an add)
  struct __align__(1024) Lds {
    uint4 ring[NW][kWave];  // 2 chunks of 32 segment entries
    // CONV: the 64 entries of the conv_up table from the tile's base on, {grid coordinate as a float, byte offset,
    // alpha w0, alpha w1}, one slice per wave, refilled per tile (r6; was the whole table per workgroup)
    float4 tabs[CONV ? NW : 1][kWave];
    float acc[NW][N];
    uint4 ring2[AXIS == 3 ? NW : 1][AXIS == 3 ? 2 * kWave : 1];  // AXIS 3: the entries' S2Ext
  };
  __shared__ Lds lds;
  static_assert(sizeof(uint4) * kWave == 1024, "ring / table slices are addressed as 1 KB blocks");
  auto &acc_all = lds.acc;
  auto &ring_all = lds.ring;
  auto &ring2_all = lds.ring2;
  auto &tabs_all = lds.tabs;
  const int lane = threadIdx.x & (kWave - 1), grp = lane / L, gl = lane & (L - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *acc = acc_all[wave];
  uint4 *ring = ring_all[wave];
  uint4 *ring2 = ring2_all[AXIS == 3 ? wave : 0];
  float4 *tabs = tabs_all[CONV ? wave : 0];
  const Dim3i dd = P.dd;
  // (r6) What only a tile's header and epilogue want - pointers, buffer ranges, the stencil's coefficients - is NOT held
  // in scalar registers through the instruction stream: held, it did not fit (96 scalar values parked in the lanes of
  // two VGPRs, ~75 v_readlane + their hazard waits per tile - VALU issues, what this kernel runs out of).  It is read
  // again from the kernel-argument segment, by scalar load, through a pointer laundered per tile (S2Args is the
  // kernel's first parameter: offset 0 of the segment).
  typedef const __attribute__((address_space(4))) S2Args *KArgs;
  auto kargs = [&]() {
    KArgs k = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
#ifndef UNIRES_S2_NO_KARG
    asm volatile("" : "+s"(k));
#endif
    return k;
  };
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(P.src, P.src_bytes);
  // XCD-aware persistent schedule: workgroup b sits on XCD b % 8; each XCD walks one contiguous
  // run of tiles so that neighbouring tiles (shared stencil halos, schedule lines) share an L2.
  // (Tried and dropped: a contiguous, cost-balanced range of tiles per wave - 103 us instead of
  // 95 us: waves that sweep the run together share cache lines, waves far apart do not.)
  // Only as many workgroups as the chip holds AT ONCE take tiles (P.active, a multiple of 8): the table
  // above is part of the workgroup's LDS, and with 384 + 128 entries a CU holds three workgroups, not
  // four - the fourth quarter of a 1024-workgroup grid then started when the first three were done and
  // the kernel took twice a workgroup's time (config 4: 424 us where 16.5 M points cost 81 us in config 3).
  if ((int)blockIdx.x >= P.active) {
    if (P.partials && lane == 0) P.partials[blockIdx.x * NW + wave] = 0.0;
    return;
  }
  const int nwg = P.active;
  const int nxcd = min(8, nwg);
  const int xcd = blockIdx.x % nxcd;
  const int t_lo = P.xlo[xcd], t_hi = P.xlo[xcd + 1];
  const int slot = (blockIdx.x / nxcd) * NW + wave;
  const int slots = ((nwg + nxcd - 1 - xcd) / nxcd) * NW;
  const float c0 = P.A.m[2], c1 = P.A.m[6], c2 = P.A.m[10];
  const float t0 = P.A.m[3], t1 = P.A.m[7], t2 = P.A.m[11];
  const int lane_m64 = lane - 64;
  typedef unsigned u4r __attribute__((ext_vector_type(4)));
  typedef float f4r __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) u4r *LdsU4;
  typedef const __attribute__((address_space(3))) f4r *LdsF4;
  const unsigned ring_b = (unsigned)(__UINTPTR_TYPE__)(LdsU4)(const void *)ring, tabs_b = (unsigned)(__UINTPTR_TYPE__)(LdsF4)(const void *)tabs;
  unsigned m3f0 = 0x3f0u;  // (in a register: with the base in a scalar register the literal would be a second constant)
  asm volatile("" : "+v"(m3f0));
  typedef float v2k __attribute__((ext_vector_type(2)));
  v2k k_one0 = {1.f, 0.f};  // (likewise: the packed weight pairs' second constant, else re-made in front of every use)
  asm volatile("" : "+v"(k_one0));
  double dot = 0.0;
#ifdef UNIRES_S2_PROF
  unsigned long long *pw = P.prof ? P.prof + (size_t)(blockIdx.x * NW + wave) * 32 : nullptr;
  int ptile = 0;
  if (pw && lane == 0) pw[0] = wall_clock64();
#endif
  // The SIMD's arbiter issues from its OLDEST wave first: of the four waves that share a SIMD the
  // first to arrive ran its instruction stream at 0.43 us per instruction, the last at 0.62 us - same
  // work, and the youngest waves set the kernel time.  User priority beats age, so every wave walks
  // through the four priorities, one per tile, starting from its slot on the SIMD: over a wave's 4.5
  // tiles the four of a SIMD are each first, second, third and last once: 81.8 -> 76.8 us.  (Rotating
  // every 8 instructions instead: 80.7 us - strict priorities are the efficient way to run, it is
  // the fixed ranking that unbalances.)
  const int hw_slot = (int)(__builtin_amdgcn_s_getreg(6148) & 3u);  // HW_ID.wave_id
  int round = 0;
  // (r6) The walk.  A tile's 16-byte record comes by SCALAR load (constant address space: the schedule is read-only),
  // the next tile's one is requested as a tile starts and travels under its stream.  What a tile needs per LANE before
  // its first instruction - the masks of instruction `lane`, the first 64 segment entries, its slice of the conv_up
  // table - is requested at the head of the PREVIOUS tile's epilogue and lands under it: three dependent trips to
  // memory (tile -> offsets -> masks / entries) stood at the head of every tile, 1.6 of its 14 us.
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const __attribute__((address_space(4))) u4v *recs = (const __attribute__((address_space(4))) u4v *)P.recs;
  const u4v rec_none = {0xffffffffu, 0u, 0u, 0u};
  int t = t_lo + slot;
  u4v rec = t < t_hi ? recs[t] : rec_none;
  // The entries and the table slice go straight into LDS (LDS-DMA: lane l's 16 bytes land on bytes 16 l of the ring /
  // the slice - no register carried through the epilogue); the masks' load is issued AFTER them, so that the wait
  // the compiler puts in front of the masks' first use covers both (loads return in order).
  ulonglong2 h_mk = make_ulonglong2(0ull, 0ull);
  struct Hdr {  // what a header request wants of the kernel arguments
    const S2Entry *entries;
    size_t ent_bytes;
    const float4 *tab;
    int gn;
    const ulonglong2 *masks;
  };
  auto load_hdr = [&](KArgs K) { return Hdr{K->entries, K->ent_bytes, K->tab, K->gn, K->masks}; };
  auto issue_header = [&](const Hdr &H, const u4v &rc, int ln) {
    const int ni = (int)(rc.w & 0xffu);
    const int gn = H.gn;
    const __amdgpu_buffer_rsrc_t rs_ent = make_rsrc(H.entries, H.ent_bytes);
    S2_FENCE();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_ent, (__attribute__((address_space(3))) void *)ring, 16,
                                             16u * (rc.y + (unsigned)ln), 0, 0, 0);
    if (CONV) {
      const __amdgpu_buffer_rsrc_t rs_tab = make_rsrc(H.tab, (size_t)gn * sizeof(float4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_tab, (__attribute__((address_space(3))) void *)tabs, 16,
                                               16u * (unsigned)min(max((int)(rc.w >> 8) + ln, 0), gn - 1), 0, 0, 0);
    }
    S2_FENCE();
    // (every lane loads - lanes past the tile's last instruction read the zero padding or a later tile's masks, never
    // used: an unconditional load is always issued, and the wait on it is what orders the DMA above)
    h_mk = H.masks[rc.z + (unsigned)min(ln, max(ni - 1, 0))];
    (void)ni;
  };
  if (rec.x != 0xffffffffu) issue_header(load_hdr(kargs()), rec, lane);
  for (; rec.x != 0xffffffffu; ++round) {
    if (P.prio_rot) {
      switch ((hw_slot + round) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
    const int x0 = (int)(rec.x & 1023u) * TX, y0 = (int)((rec.x >> 10) & 1023u) * TY, z0 = (int)(rec.x >> 20) * T::TZ;
    const int ex = min(TX, dd.x - x0), ey = min(TY, dd.y - y0), ez = min(T::TZ, dd.z - z0);
    const int ninstr = (int)(rec.w & 0xffu);
    const int tab_base = (int)(rec.w >> 8);
    const unsigned ent0 = rec.y;
    const uint4 *E = reinterpret_cast<const uint4 *>(P.entries) + ent0;  // (used through the stream: stays in registers)
    const int t_next = t + slots;
    const u4v rec_next = t_next < t_hi ? recs[t_next] : rec_none;
    // lane l keeps the segment-start mask of instruction l (a tile has at most 64 of them)
    int lp = lane;  // (laundered per tile, as in the epilogue: the prologue's addresses are not held through the stream)
    asm volatile("" : "+v"(lp));
    const ulonglong2 mymask = h_mk;
    const int mlo_v = (int)(unsigned)mymask.x, mhi_v = (int)(unsigned)(mymask.x >> 32);
    const int alo_v = (int)(unsigned)mymask.y, ahi_v = (int)(unsigned)(mymask.y >> 32);
    // the masks are here => so are the ring's chunks 0 and 1 and the table slice (issued before them)
    asm volatile("" ::"v"(mlo_v), "v"(mhi_v), "v"(alo_v), "v"(ahi_v) : "memory");
    // segment ring: chunk 2 in flight in registers
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);
    if (lp < 32) pre = E[64 + lp];
    if (CONV) {
      // (.x = the grid coordinate the entry belongs to, as a float: the z-profile kernels take k from the table
      // instead of converting it, and the 16-byte read costs the LDS 4 cycles where the 12-byte one costs 8)
      const unsigned step4 = (AXIS == 2 || AXIS == 3) ? 4u : P.tab_step4;  // the per-lane table runs along z
      const float4 raw = tabs[lp];
      // (as the stream reads it: {grid coordinate, byte offset, alpha w0, alpha w1} - the coordinate FIRST, in the even
      // register of the 16-byte read, where the packed coordinate arithmetic takes it without a move)
      tabs[lp] = make_float4((float)(tab_base + lp), __int_as_float((int)(step4 * (unsigned)__float_as_int(raw.x))),
                             P.alpha * raw.y, P.alpha * raw.z);
    }
#ifdef UNIRES_S2_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // per tile: {header here, stream end, p window here, end, instructions}
    if (pw && lane == 0 && ptile < 5) pw[3 + 5 * ptile] = wall_clock64();
#endif
    const uint4 *X = reinterpret_cast<const uint4 *>(P.ext) + 2 * (size_t)ent0;  // AXIS 3: 2 uint4 per entry
    uint4 pre2 = make_uint4(0u, 0u, 0u, 0u);
    if (AXIS == 3) {
      ring2[lane] = X[lane], ring2[kWave + lane] = X[kWave + lane];
      pre2 = X[2 * kWave + lane];  // chunk 2 = 32 entries = 64 uint4
    }
    {
      // (the zeros and the lane's address are made HERE, per tile: kept live across the instruction stream as loop
      // invariants they cost five of its 128 registers, and the allocator parked them in scratch memory)
      int lz = lane;
      float zr = 0.f;
      asm volatile("" : "+v"(lz), "+v"(zr));
      for (int i = lz; i < N / 4; i += kWave) reinterpret_cast<float4 *>(acc)[i] = make_float4(zr, zr, zr, zr);
    }
    // accumulator cell of global floor cell (fx, fy, fz): acc_t[(fx * SY + fy) * SZ + fz]
    // (the offset is laundered: left to itself the compiler folds its constant part, XS + YS + 1, into the immediate
    // offsets of the eight LDS accesses, which then no longer fit their 8 bits - two more address additions per
    // splat instruction; opaque, the four cells sit at 0 / YS / XS / XS + YS (+ 1) <= 225 dwords from ONE address)
    // (... and the accumulator's own LDS address goes in with it: one scalar byte address per tile)
    typedef __attribute__((address_space(3))) float *LdsF1;
    unsigned acc_addr = (unsigned)(__UINTPTR_TYPE__)(LdsF1)(void *)acc - 4u * (unsigned)(((x0 - 1) * SY + (y0 - 1)) * SZ + (z0 - 1));
    asm volatile("" : "+s"(acc_addr));
    int chunk_lo = 0;  // the ring holds chunks chunk_lo and chunk_lo + 1 (32 entries each)
    int eb = 0;        // first entry of the next instruction, relative to the tile
    S2_FENCE();
    constexpr int kU = 4;
    // (Tried and dropped: a schedule whose groups of 2 / 4 consecutive instructions are mutually
    // conflict-free, so that their LDS updates run as two rounds with all reads of a round in
    // flight - the extra packing constraint costs lane fill (0.75 -> 0.65 / 0.40) and the round form
    // 20-30 more VGPRs (occupancy 4 -> 3 waves per SIMD): 155 us instead of 88 us.)
    // A batch = kU instructions.  fetch(): segment-start masks -> mbcnt -> segment entry (LDS ring)
    // -> conv_up table -> ONE source load per lane and instruction, all issued back to back;
    // splat(): coordinates, weights and the two LDS update groups.  Two batches are in flight: the
    // source loads of batch b + 1 travel while batch b is splatted.
    struct Batch {
      float w0[kU], w1[kU], s0[kU], s1[kU], gx[kU], gy[kU], gz[kU];  // (coordinates, not their four ingredients: a register less per slot)
      unsigned long long amask[kU];
    };
    auto fetch = [&](Batch &Bt, int p0) {
      unsigned mlo[kU], mhi[kU];
      int ebu[kU];
      int need = eb;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        // (a batch slot past the tile's last instruction replays that instruction, switched off)
        const int pi = min(p0 + u, ninstr - 1);
        mlo[u] = (unsigned)__builtin_amdgcn_readlane(mlo_v, pi);
        mhi[u] = (unsigned)__builtin_amdgcn_readlane(mhi_v, pi);
        Bt.amask[u] = p0 + u < ninstr ? ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(ahi_v, pi) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane(alo_v, pi)
                                      : 0ull;
        ebu[u] = (p0 + u < ninstr || u == 0) ? need : ebu[u > 0 ? u - 1 : 0];
        if (p0 + u < ninstr) need += __popc(mlo[u]) + __popc(mhi[u]) + 1;
      }
      if (need > 32 * (chunk_lo + 2)) {  // advance the ring by one chunk (the oldest one is dead)
        S2_FENCE();
        if (lane < 32) ring[(chunk_lo & 1) * 32 + lane] = pre;
        if (AXIS == 3) ring2[(chunk_lo & 1) * 64 + lane] = pre2;
        ++chunk_lo;
        int la = lane;  // (laundered: 16 * lane is not to be held through the tile loop - it ended up in scratch memory)
        asm volatile("" : "+v"(la));
        if (la < 32) pre = E[32 * (chunk_lo + 2) + la];
        if (AXIS == 3) pre2 = X[64 * (chunk_lo + 2) + lane];
        S2_FENCE();
      }
      eb = need;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int sl = (int)__builtin_amdgcn_mbcnt_hi(mhi[u], __builtin_amdgcn_mbcnt_lo(mlo[u], 0u));
        const u4r e = *(LdsU4)(__UINTPTR_TYPE__)((((unsigned)(ebu[u] + sl) << 4) & m3f0) | ring_b);
        const float rx = __uint_as_float(e.x), ry = __uint_as_float(e.y), rz = __uint_as_float(e.z);
        float kf;
        const unsigned code = e.w & kS2RowIdle;
        // (idle lanes in front of an instruction's first segment see k < its k0: the tables carry 64
        // zero entries in front for them)
        // (axis 2 / 3: k relative to the tile's table base - the index into its slice for every active lane)
        const int k = (int)(e.w >> kS2RowBits) + lane_m64;
        kf = (float)k;
        Bt.w0[u] = P.alpha, Bt.w1[u] = 0.f, Bt.s0[u] = 1.f, Bt.s1[u] = 0.f;
        if (S2_ABL(8)) {
        } else if (AXIS == 3) {
          // conv_up along x, y and z: 2 x 2 x-space columns (per segment) x the z pair (per lane)
          const uint4 xa = ring2[2 * ((ebu[u] + sl) & 63)], xb = ring2[2 * ((ebu[u] + sl) & 63) + 1];
          const f4r tb = *(LdsF4)(__UINTPTR_TYPE__)((((unsigned)k << 4) & m3f0) | tabs_b);
          const unsigned a = xa.x + (unsigned)__float_as_int(tb.y);
          const uint2 p00 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, a, 0, 0));
          const uint2 p01 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, a + P.xs_sy4, 0, 0));
          const uint2 p10 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, a + P.xs_sx4, 0, 0));
          const uint2 p11 =
              __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, a + P.xs_sx4 + P.xs_sy4, 0, 0));
          const float z00 = tb.z * __uint_as_float(p00.x) + tb.w * __uint_as_float(p00.y);
          const float z01 = tb.z * __uint_as_float(p01.x) + tb.w * __uint_as_float(p01.y);
          const float z10 = tb.z * __uint_as_float(p10.x) + tb.w * __uint_as_float(p10.y);
          const float z11 = tb.z * __uint_as_float(p11.x) + tb.w * __uint_as_float(p11.y);
          Bt.w0[u] = 1.f;
          kf = tb.x;
          Bt.s0[u] = __uint_as_float(xa.y) * z00 + __uint_as_float(xa.z) * z01 + __uint_as_float(xa.w) * z10 +
                     __uint_as_float(xb.x) * z11;
          (void)code;
        } else if (AXIS == 2) {
          const f4r tb = *(LdsF4)(__UINTPTR_TYPE__)((((unsigned)k << 4) & m3f0) | tabs_b);
          kf = tb.x;
          const unsigned a = __umul24(code, P.row_stride4) + (unsigned)__float_as_int(tb.y);
          const uint2 pr = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, a, 0, 0));
          Bt.w0[u] = tb.z, Bt.w1[u] = tb.w, Bt.s0[u] = __uint_as_float(pr.x), Bt.s1[u] = __uint_as_float(pr.y);
        } else if (AXIS == 0 || AXIS == 1) {
          const unsigned ui = code >> 9, uj = code & 511u;
          const float4 tb = tabs[((int)(AXIS == 0 ? ui : uj) - tab_base) & 63];
          const unsigned a = __umul24(AXIS == 0 ? uj : ui, P.row_stride4) + (unsigned)__float_as_int(tb.y) +
                             4u * (unsigned)k;
          Bt.w0[u] = tb.z, Bt.w1[u] = tb.w;
          Bt.s0[u] = buf_load(rsrc, a, 0), Bt.s1[u] = buf_load(rsrc, a + P.tab_step4, 0);
        } else {
          Bt.s0[u] = buf_load(rsrc, __umul24(code, P.row_stride4) + 4u * (unsigned)k, 0);
        }
#ifndef UNIRES_S2_NO_PKXY
        {
          // x and y as ONE packed pair per instruction (the entry's {rx, ry} arrive adjacent from the ring): left to
          // itself the vectoriser pairs the same coordinate of two batch slots and pays eleven moves per batch for it
          typedef float v2f __attribute__((ext_vector_type(2)));
          const v2f cxy = {c0, c1}, txy = {t0, t1}, rxy = {rx, ry}, kk = {kf, kf};
          const v2f gxy = __builtin_elementwise_fma(cxy, kk, rxy) + txy;
          Bt.gx[u] = gxy.x, Bt.gy[u] = gxy.y;
        }
#else
        Bt.gx[u] = fmaf(c0, kf, rx) + t0;
        Bt.gy[u] = fmaf(c1, kf, ry) + t1;
#endif
        Bt.gz[u] = fmaf(c2, kf, rz) + t2;
      }
    };
    // (Tried and dropped: the four weight products and the eight accumulate FMAs as v_pk_mul_f32 /
    // v_pk_fma_f32 on the (y, y + 1) pairs the ds_read2 / ds_write2 carry - 6 VALU instructions less
    // per splat instruction, 129 VGPRs (128 when forced), and 79-83 us where this form has 76-83.)
    auto splat = [&](const Batch &Bt) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float gx = Bt.gx[u], gy = Bt.gy[u], gz = Bt.gz[u];
        // floor cell and weights exactly as the pull computes them (floor, then g - floor: the build kernel's
        // collision tests floor the same global coordinate); the tile's base enters once, as an integer offset of
        // the accumulator pointer (r6: was a subtraction per coordinate in front of v_fract - three instructions
        // more, and exact only for tiles off the volume's low faces)
#ifndef UNIRES_S2_NO_PKW
        // (r6, with the registers the build without the SLP vectoriser left free - packed, this form needed 129 of 128 in
        // round 3.)  The weights as PAIRS: {wx0, wx1} and {wy0, wy1} come out of one packed fma each ({-1, 1} w + {1, 0}: the
        // same single rounding as 1 - w, and w itself), v {wx0, wx1}, then {a00, a01} = vx0 {wy0, wy1} and {a10, a11} = vx1
        // {wy0, wy1} are packed multiplies, and the accumulator's (y, y + 1) pairs arrive from ds_read2_b32 as register
        // pairs: one packed fma per pair where there were two.  21 vector instructions per splat instruction instead of 29,
        // every product and sum rounded exactly as before.
        typedef float v2f __attribute__((ext_vector_type(2)));
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const v2f gxy = {gx, gy}, fxy = {fx, fy};
        const v2f w1 = gxy - fxy;  // {wx1, wy1}
        const float wz1 = gz - fz, wz0 = 1.f - wz1;
        const v2f m11 = {-1.f, 1.f}, one0 = k_one0;
        const v2f wx = __builtin_elementwise_fma(m11, (v2f){w1.x, w1.x}, one0);  // {wx0, wx1}
        const v2f wy = __builtin_elementwise_fma(m11, (v2f){w1.y, w1.y}, one0);  // {wy0, wy1}
        // cell index in float (exact: integers below 2^24), one conversion
        const float cf = fmaf(fx, (float)XS, fmaf(fy, (float)YS, fz));
        const int cell = (int)cf;
        const float v = Bt.w0[u] * Bt.s0[u] + Bt.w1[u] * Bt.s1[u];
        const v2f vx = (v2f){v, v} * wx;                 // {vx0, vx1}
        const v2f a0 = (v2f){vx.x, vx.x} * wy, a1 = (v2f){vx.y, vx.y} * wy;  // {a00, a01}, {a10, a11}
        S2_FENCE();
        if (S2_ABL(4)) dot += (double)(a0.x + a0.y + a1.x + a1.y + (float)cell);
        if (__builtin_amdgcn_inverse_ballot_w64(Bt.amask[u]) && !S2_ABL(4)) {
          const LdsF1 q = (LdsF1)(__UINTPTR_TYPE__)(acc_addr + 4u * (unsigned)cell);
          {
            const v2f o0 = {q[0], q[YS]}, o1 = {q[XS], q[XS + YS]};
            const v2f r0 = __builtin_elementwise_fma(a0, (v2f){wz0, wz0}, o0), r1 = __builtin_elementwise_fma(a1, (v2f){wz0, wz0}, o1);
            q[0] = r0.x, q[YS] = r0.y, q[XS] = r1.x, q[XS + YS] = r1.y;
          }
          S2_FENCE();
          {
            const v2f o0 = {q[1], q[YS + 1]}, o1 = {q[XS + 1], q[XS + YS + 1]};
            const v2f r0 = __builtin_elementwise_fma(a0, (v2f){wz1, wz1}, o0), r1 = __builtin_elementwise_fma(a1, (v2f){wz1, wz1}, o1);
            q[1] = r0.x, q[YS + 1] = r0.y, q[XS + 1] = r1.x, q[XS + YS + 1] = r1.y;
          }
        }
#else
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const float wx1 = gx - fx, wy1 = gy - fy, wz1 = gz - fz;
        const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
        // cell index in float (exact: integers below 2^24), one conversion
        const float cf = fmaf(fx, (float)XS, fmaf(fy, (float)YS, fz));
        const int cell = (int)cf;
        const float v = Bt.w0[u] * Bt.s0[u] + Bt.w1[u] * Bt.s1[u];
        const float vx0 = v * wx0, vx1 = v * wx1;
        const float a00 = vx0 * wy0, a01 = vx0 * wy1, a10 = vx1 * wy0, a11 = vx1 * wy1;
        S2_FENCE();
        if (S2_ABL(4)) dot += (double)(a00 + a01 + a10 + a11 + (float)cell);
        if (__builtin_amdgcn_inverse_ballot_w64(Bt.amask[u]) && !S2_ABL(4)) {
          const LdsF1 q = (LdsF1)(__UINTPTR_TYPE__)(acc_addr + 4u * (unsigned)cell);
          {
            const float o00 = q[0], o01 = q[YS], o10 = q[XS], o11 = q[XS + YS];
            q[0] = o00 + a00 * wz0, q[YS] = o01 + a01 * wz0, q[XS] = o10 + a10 * wz0,
            q[XS + YS] = o11 + a11 * wz0;
          }
          S2_FENCE();
#ifdef S2_EXP_MASKB
          if ((lane & (S2_EXP_MASKB - 1)) == 0)  // (measurement only, wrong results: what a mostly-masked second group costs)
#endif
          {
            const float o00 = q[1], o01 = q[YS + 1], o10 = q[XS + 1], o11 = q[XS + YS + 1];
            q[1] = o00 + a00 * wz1, q[YS + 1] = o01 + a01 * wz1, q[XS + 1] = o10 + a10 * wz1,
            q[XS + YS + 1] = o11 + a11 * wz1;
          }
        }
#endif
        S2_FENCE();
      }
    };
    if (ninstr > 0 && !S2_ABL(1)) {
      // (Measured, r2: the branches around the fetches make the compiler's wait-count pass put
      // s_waitcnt vmcnt(0) in front of every splat, so a batch also waits for the source loads of
      // the next one.  A branch-free body - fetch / splat / fetch / splat with slots past the end
      // switched off - gets vmcnt(4..7) there and is no faster, 92 vs 90 us: the stream is not bound
      // by that latency but by VALU issue (~58 % of a SIMD) and LDS (~57 % of a CU) together, see
      // tools/s2_timeline.py.  Same outcome for a dynamic tile queue (also on top of the cheap-tiles-last
      // order: ends bunch up, 64 .. 80 us, but the ticket in front of the epilogue's loads costs each
      // epilogue 1 us), a staggered start of half the waves and a prefetch of the next tile's schedule.)
      Batch ba, bb;
      fetch(ba, 0);
      for (int p0 = 0; p0 < ninstr; p0 += 2 * kU) {
        const bool more = p0 + kU < ninstr;
        if (more) fetch(bb, p0 + kU);
        splat(ba);
        if (more) {
          if (p0 + 2 * kU < ninstr) fetch(ba, p0 + 2 * kU);
          splat(bb);
        }
      }
    }
    S2_FENCE();
#ifdef UNIRES_S2_PROF
    if (pw && lane == 0 && ptile < 5) pw[4 + 5 * ptile] = wall_clock64(), pw[7 + 5 * ptile] = (unsigned long long)ninstr;
#endif
    // ---- epilogue: q = [q +] acc + a0 p + c DtD p ; dot += p*q  (one row per lane group) ----
    // the next tile's per-lane header: requested now, it lands under this epilogue
    // (ALL of the epilogue's scalars are requested here, ahead of the branch around the header request, and pinned: one
    // trip to the scalar cache per tile - they came in two, the second one behind the header's loads)
    const KArgs K = kargs();
    const Hdr H = load_hdr(K);
    const float *__restrict__ pin = K->p;
    float *__restrict__ dst = K->dst;
    const float *objb = OBJK ? K->objb : nullptr;
    const int accumulate = K->accumulate;
    const float kcx = K->cx, kcy = K->cy, kcz = K->cz, ka0 = K->a0;
    asm volatile("" ::"s"(H.entries), "s"(H.ent_bytes), "s"(H.tab), "s"(H.gn), "s"(H.masks), "s"(pin), "s"(dst), "s"(objb),
                 "s"(accumulate), "s"(kcx), "s"(kcy), "s"(kcz), "s"(ka0));
    {
      // (the ring's chunk in flight may never have been used: let it land NOW, or the compiler waits for everything -
      // the header included - the first time the epilogue writes one of its registers)
      asm volatile("" ::"v"(pre.x), "v"(pre.y), "v"(pre.z), "v"(pre.w));
      if (AXIS == 3) asm volatile("" ::"v"(pre2.x), "v"(pre2.y), "v"(pre2.z), "v"(pre2.w));
      int lh = lane;
      asm volatile("" : "+v"(lh));
      if (rec_next.x != 0xffffffffu) issue_header(H, rec_next, lh);
    }
    double dtile = 0.0;  // this tile's part of the dot
    const bool fast_xy = !S2_ABL(2) && pin != nullptr && !accumulate && ex == TX && ey == TY && dd.numel() < (1ull << 29);
    if (fast_xy) {
      // Whole tiles, wherever they lie: x / y stencil neighbours outside the volume read as zeros
      // through an out-of-range buffer offset (the volume's first and last x slabs are a quarter of
      // the tiles of the XCDs that own them, and a wave whose stride lands it on the first or last
      // y row of tiles got four 12 us generic epilogues in a row).  Lane gl of a group holds
      // z plane z0 - 1 + gl of the aproned tile; group g owns x slabs 4g .. 4g + 3.  Every row of p
      // the group's stencils touch is loaded ONCE (32 loads per tile instead of 7 per output row);
      // the z neighbours come from the adjacent lanes (DPP wave shifts), the x / y neighbours from
      // the other registers.  Buffer addressing: a per-lane byte offset computed once per tile +
      // scalar row offsets.
      static_assert(TX == 8 && TY == 4 && G == 2, "epilogue register window is written for 8 x 4 tiles");
      // (r6) Instruction count is what this kernel runs out of, so the epilogue is written to be short:
      //  * no selects for the volume's faces: a neighbour that the reference's forward differences do not see is
      //    loaded from the CENTRE's own address (c - c = 0: the absent backward term of the first slab / row / plane),
      //    a neighbour beyond the last slab / row / plane from an out-of-range offset (0: the zero bound);
      //  * every difference once: p[s + 1] - p[s] is the forward term of s and the backward term of s + 1; along z the
      //    difference of adjacent lanes is one DPP subtraction, and (backward - forward) a second one;
      //  * the 16 accumulator reads issued together, ahead of the arithmetic; no branch per output row (lanes of the
      //    z apron store to an out-of-range offset and add nothing to the dot).
      // (lane-dependent values of the epilogue are derived from a laundered copy of the lane index, so that none of
      // them is hoisted out of the tile loop and held in a register through the instruction stream)
      int le = lane;
      asm volatile("" : "+v"(le));
      const int gl = le & (L - 1), grp = le / L;
      const int kz = z0 - 1 + gl;
      const bool out_z = gl >= 1 && gl <= ez;
      // (the 24 row offsets below are scalar sums of these two: as loop invariants of the tile loop they were computed
      // once, could not all be held through the stream and came back from VGPR lanes - 58 v_readlane + their hazard
      // waits per epilogue; laundered per tile they are made here by the scalar unit)
      unsigned sxb = 4u * (unsigned)(dd.y * dd.z), syb = 4u * (unsigned)dd.z;
      asm volatile("" : "+s"(sxb), "+s"(syb));
      constexpr unsigned kOob = 0x80000000u;
      const int xg = x0 + 4 * grp;
      // byte offset of (first owned slab, first owned row, plane kz): plane -1 reads plane 0 (so that the backward
      // difference of plane 0 comes out as zero), planes >= dd.z read zeros
      const unsigned e0 = 4u * (unsigned)((xg * dd.y + y0) * dd.z + max(kz, 0));
      const unsigned e1 = kz < dd.z ? e0 : kOob;
      const unsigned elo = xg > 0 ? e1 - sxb : e1, ehi = xg + 4 < dd.x ? e1 : kOob;
      const unsigned eyl = y0 > 0 ? e1 - syb : e1, eyh = y0 + TY < dd.y ? e1 : kOob;
      const unsigned est = out_z ? e0 : kOob;
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(pin, dd.numel() * 4),
                                   rd = make_rsrc(dst, dd.numel() * 4),
                                   rb = make_rsrc(objb ? objb : pin, dd.numel() * 4);
      const float *arow = acc + ((4 * grp + 1) * SY + 1) * SZ + gl;
      float pv[6][6];  // pv[s + 1][ly + 1]: x slab s = -1 .. 4, row ly = -1 .. 4 (corners unused)
#pragma unroll
      for (int sa = 0; sa < 6; ++sa)
#pragma unroll
        for (int la = 0; la < 6; ++la) {
          const bool halo_x = sa == 0 || sa == 5, halo_y = la == 0 || la == 5;
          // (the low halos sit at the centre's own row offsets when they alias it: first slab / first row)
          pv[sa][la] = (halo_x && halo_y) ? 0.f
                       : sa == 0 ? buf_load(rp, elo, (unsigned)(la - 1) * syb)
                       : sa == 5 ? buf_load(rp, ehi, 4u * sxb + (unsigned)(la - 1) * syb)
                       : la == 0 ? buf_load(rp, eyl, (unsigned)(sa - 1) * sxb)
                       : la == 5 ? buf_load(rp, eyh, (unsigned)(sa - 1) * sxb + 4u * syb)
                                 : buf_load(rp, e1, (unsigned)(sa - 1) * sxb + (unsigned)(la - 1) * syb);
        }
#ifdef UNIRES_S2_PROF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (pw && lane == 0 && ptile < 5) pw[5 + 5 * ptile] = wall_clock64();
#endif
      auto rows = [&](auto obj_tag) {
        constexpr bool OBJ = decltype(obj_tag)::value;
        // x and y differences, each once: p[s + 1][l] - p[s][l] is the forward term of slab s and the backward term of s + 1
        float dxp[4];
#pragma unroll
        for (int la = 1; la <= 4; ++la) dxp[la - 1] = pv[1][la] - pv[0][la];
        double dt = 0.0;
#pragma unroll
        for (int sa = 1; sa <= 4; ++sa) {
          float dy[5], av[4];  // (the slab's four accumulator reads issued together, ahead of its arithmetic)
#pragma unroll
          for (int la = 1; la <= 4; ++la) av[la - 1] = arow[((sa - 1) * SY + (la - 1)) * SZ];
          float ob[4];  // (objective form: b of the slab's rows, requested per slab - all sixteen up front did not fit the registers)
          if (OBJ) {
#pragma unroll
            for (int la = 1; la <= 4; ++la) ob[la - 1] = buf_load(rb, e1, (unsigned)(sa - 1) * sxb + (unsigned)(la - 1) * syb);
          }
#pragma unroll
          for (int la = 0; la <= 4; ++la) dy[la] = pv[sa][la + 1] - pv[sa][la];
#pragma unroll
          for (int la = 1; la <= 4; ++la) {
            const float c = pv[sa][la];
            // zf = p[k + 1] - p[k] (lane + 1 minus this lane); zbk = the same difference one lane down
            const float vzp = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x130, 0xf, 0xf, true));
            const float zf = vzp - c;
            const float zbk = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(zf), 0x138, 0xf, 0xf, true));
            const float xf = pv[sa + 1][la] - c, xbk = dxp[la - 1];
            dxp[la - 1] = xf;
            const float yf = dy[la], ybk = dy[la - 1];
            const float st = kcx * (xbk - xf) + kcy * (ybk - yf) + kcz * (zbk - zf);
            const float q = av[la - 1] + (ka0 * c + st);
            if (OBJ) {
              dt += (double)obj_term(q, ob[la - 1], c);
            } else {
              buf_store(q, rd, est, (unsigned)(sa - 1) * sxb + (unsigned)(la - 1) * syb);
              dt += (double)__fmul_rn(c, q);
            }
          }
        }
        dtile = out_z ? dt : 0.0;
      };
      rows(std::integral_constant<bool, OBJK>{});
    } else if (!S2_ABL(2)) {
      int lg = lane;  // (laundered: see the fast form)
      asm volatile("" : "+v"(lg));
      const int gl = lg & (L - 1), grp = lg / L;
#pragma unroll 2
      for (int r = grp; r < TX * TY; r += G) {
        const int lx = r / TY, ly = r % TY, lz = gl;
        if (lx >= ex || ly >= ey || lz >= ez) continue;
        const int i = x0 + lx, j = y0 + ly, k = z0 + lz;
        const size_t idx = ((size_t)i * dd.y + j) * dd.z + k;
        float q = acc[((lx + 1) * SY + ly + 1) * SZ + lz + 1];
        float pc = 0.f;
        if (pin) {
          const float st = dtd_at(pin, idx, i, j, k, dd, kcx, kcy, kcz, pc);
          q += ka0 * pc + st;
        }
        if (accumulate) q += dst[idx];
        matvec_emit(dst, idx, q, pc, objb, P.partials != nullptr, dtile);
      }
    }
#ifdef UNIRES_S2_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pw && lane == 0 && ptile < 5) pw[6 + 5 * ptile] = wall_clock64();
    ++ptile;
#endif
    dot += dtile;
    t = t_next, rec = rec_next;
  }
#ifdef UNIRES_S2_PROF
  if (pw && lane == 0) pw[1] = wall_clock64(), pw[2] = (unsigned long long)ptile;
#endif
  if (P.partials) {
    const double tot = wave_sum(dot);
    if (lane == 0) P.partials[blockIdx.x * NW + wave] = tot;
  }
}

int splat2_blocks(Dim3i dd, int grid_cap) { return s2_grid(dd, grid_cap) * kS2Waves; }  // partials written

// Workgroups of k_splat2<axis> that take tiles: as many of the grid as the device holds at once (asked of the
// runtime once per kernel), rounded down to whole rounds over the 8 XCDs.  (All LDS is static since r6: four
// 4-wave workgroups per CU for every axis but 3, whose per-segment x / y tables make it three.)
static int s2_active(int axis, int grid) {
  static std::map<int, int> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(axis);
  if (it == cache.end()) {
    int per_cu = 0, dev = 0, ncu = 0;
    const void *fn = axis == 0   ? (const void *)k_splat2<0, kS2Waves, false>
                     : axis == 1 ? (const void *)k_splat2<1, kS2Waves, false>
                     : axis == 2 ? (const void *)k_splat2<2, kS2Waves, false>
                     : axis == 3 ? (const void *)k_splat2<3, kS2Waves, false>
                                 : (const void *)k_splat2<-1, kS2Waves, false>;
    static const int force = getenv("UNIRES_SPLAT2_RESIDENT") ? atoi(getenv("UNIRES_SPLAT2_RESIDENT")) : 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kWave * kS2Waves, 0) != hipSuccess) per_cu = 0;
    if (force > 0) per_cu = force;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
    const int n = per_cu > 0 && ncu > 0 ? per_cu * ncu : 1 << 30;
    it = cache.emplace(axis, n).first;
  }
  int n = std::min(grid, it->second);
  if (n >= 8) n -= n % 8;
  return n;
}

int launch_splat2(const SplatSched &S, const float *src, size_t src_numel, const float4 *tab_dev, int gn,
                  unsigned row_stride, unsigned tab_step, unsigned xs_sy, unsigned xs_sx, const Affine &A,
                  float alpha,
                  const PushEpilogue &ep, float *dst, Dim3i dd, const int *done, hipStream_t st) {
  if (!S.valid || S.ntiles != s2_ntiles(dd) || !S.recs) return 1;
  if (S.axis >= 0 && (!tab_dev || gn < 1)) return 1;
  if (S.axis == 3 && !S.ext) return 1;
  if (src_numel >= (1ull << 30)) return 1;  // 32-bit byte offsets into the source
  if ((unsigned long long)row_stride * 4ull >= (1ull << 24)) return 1;  // 24-bit multiply
  S2Args P;
  P.src = src;
  P.src_bytes = src_numel * sizeof(float);  // buffer range check: idle lanes may point anywhere
  P.tab = tab_dev, P.gn = gn;
  P.row_stride4 = 4u * row_stride, P.tab_step4 = 4u * tab_step;
  P.entries = S.entries, P.ent_bytes = S.cap_entries * sizeof(S2Entry), P.ext = S.ext, P.xs_sy4 = 4u * xs_sy, P.xs_sx4 = 4u * xs_sx, P.masks = S.masks, P.recs = S.recs, P.ntiles = S.ntiles;
  P.A = A, P.alpha = alpha;
  P.p = ep.p, P.a0 = ep.a0, P.cx = ep.cx, P.cy = ep.cy, P.cz = ep.cz;
  P.dst = dst, P.dd = dd, P.accumulate = ep.accumulate, P.partials = ep.partials, P.objb = ep.objb;
  static const int dbg = getenv("UNIRES_S2_DBG") ? atoi(getenv("UNIRES_S2_DBG")) : 0;
  P.dbg = dbg;
  const dim3 grid(s2_grid(dd, ep.grid_cap)), block(kWave * kS2Waves);
  P.active = s2_active(S.axis, (int)grid.x);
  // (the walk's layout depends on the launch only through the number of partitions, min(8, workgroups): a smaller
  // persistent grid - ep.grid_cap - walks the same records with fewer wave slots per partition)
  if (std::min(8, P.active) != std::min(8, S.nwg) || (P.active >= 8 && P.active % 8)) return 1;
  for (int x = 0; x <= 8; ++x) P.xlo[x] = S.pos_lo[x];
  static const int prio_rot = getenv("UNIRES_S2_PRIO") ? atoi(getenv("UNIRES_S2_PRIO")) : 1;
  P.prio_rot = prio_rot;
  P.prof = nullptr;
#ifdef UNIRES_S2_PROF
  static unsigned long long *prof_dev = nullptr;
  const size_t nprof = (size_t)grid.x * kS2Waves * 32;
  if (!prof_dev) (void)hipMalloc((void **)&prof_dev, 4096 * 4 * 32 * sizeof(unsigned long long));
  (void)hipMemsetAsync(prof_dev, 0, nprof * sizeof(unsigned long long), st);
  P.prof = prof_dev;
#endif
#define S2_LAUNCH(AX)                                                                          \
  do {                                                                                         \
    if (P.objb)                                                                                \
      hipLaunchKernelGGL((k_splat2<AX, kS2Waves, true>), grid, block, 0, st, P, done);         \
    else                                                                                       \
      hipLaunchKernelGGL((k_splat2<AX, kS2Waves, false>), grid, block, 0, st, P, done);        \
  } while (0)
  switch (S.axis) {
    case 0: S2_LAUNCH(0); break;
    case 1: S2_LAUNCH(1); break;
    case 2: S2_LAUNCH(2); break;
    case 3: S2_LAUNCH(3); break;
    default: S2_LAUNCH(-1); break;
  }
#undef S2_LAUNCH
#ifdef UNIRES_S2_PROF
  {
    static int shots = 0;
    if (++shots == 12 && getenv("UNIRES_S2_PROF_OUT")) {  // one warmed-up launch
      std::vector<unsigned long long> h(nprof);
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h.data(), prof_dev, nprof * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      FILE *f = fopen(getenv("UNIRES_S2_PROF_OUT"), "w");
      for (size_t w = 0; w < nprof / 32; ++w) {
        for (int i = 0; i < 32; ++i) fprintf(f, "%llu ", h[w * 32 + i]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  return 0;
}

}  // namespace unires
