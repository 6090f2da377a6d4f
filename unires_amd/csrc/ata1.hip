// ata1.hip - single-pass A^T A = push_M . pull_M (+ stencil, + dot) for operators without a slice
// profile (see ata1.hpp; unires/_project.py:180-188).
//
// Budget per wave (tile 4 x 4 x 30): p window 6 x 6 x 32 floats + accumulator 6 x 6 x 32 floats + a ring of
// 64 segment entries = 10 KB of LDS, so a CU holds 16 waves (4 per SIMD - a wave alone issues one instruction
// every ~6.8 clocks, four of them one every ~2.1: profiles/r04_mb_valu2.txt; the 8 x 4 x 30 tile of the splat
// with a window next to it would leave 2.5 waves per SIMD).  Per 64-lane instruction: one 16-byte header
// (scalar), the lanes' segment entries from the LDS ring, ~55 VALU, 4 ds_read2 (gather), 4 + 4 ds_read2 /
// ds_write2 (two dense read-add-write groups: lower z plane, then upper) - the x-space round trip, the second
// kernel's decode / source loads and half of the coordinate arithmetic of the pull + splat pair are gone.
//
// Race freedom without atomics, as in splat2.hip: a tile is owned by ONE wave whose LDS operations execute in
// order; the lanes of a segment sit in consecutive z planes (the build kernel cuts a row wherever the plane
// does not advance by exactly one), a read-add-write group touches one plane per lane, and two segments that
// share an instruction AND a z plane have footprints that cannot meet: rows >= row_sep apart (the quick rule,
// rebuilds in set_repeat) or, point by point on the shared planes, cells >= 2 apart in x or y (the exact
// rule, plan creation - k_f1_build).  The schedule is fixed: results are bit-reproducible.
#include "ata1.hpp"
#include "splat2.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace unires {

#define F1_FENCE() asm volatile("" ::: "memory")
// phase-ablation switches: compiled in only with -DUNIRES_ABLATE (then UNIRES_F1_DBG selects bits: 1 no
// instruction stream, 2 no epilogue arithmetic / stores, 4 no window staging, 8 no LDS updates, 16 no gather);
// product builds carry none of it
#ifdef UNIRES_ABLATE
#define F1_ABL(bit) ((P.dbg & (bit)) != 0)
#else
#define F1_ABL(bit) (false)
#endif

constexpr int kF1Waves = 4;  // waves (= independent tiles in flight) per workgroup
constexpr int kF1TZ = 30, kF1SZ = 32;
constexpr unsigned kF1KBias = 64u, kF1KMask = 0x1fffu;  // pk = (k of lane 0 + bias) | first lane << 13 | len << 19
constexpr int kF1MaxSeg = 8;  // segments per instruction

struct F1TileGeom {
  int x0, y0, z0, ex, ey, ez;
};
template <int TX, int TY>
__host__ __device__ __forceinline__ F1TileGeom f1_tile(int t, const Dim3i &dd) {
  const int nty = (dd.y + TY - 1) / TY, ntz = (dd.z + kF1TZ - 1) / kF1TZ;
  const int tzi = t % ntz, tyi = (t / ntz) % nty, txi = t / (ntz * nty);
  F1TileGeom g;
  g.x0 = txi * TX, g.y0 = tyi * TY, g.z0 = tzi * kF1TZ;
  g.ex = min(TX, dd.x - g.x0), g.ey = min(TY, dd.y - g.y0), g.ez = min(kF1TZ, dd.z - g.z0);
  return g;
}
static int f1_ntiles(Dim3i dd, int tx, int ty) {
  return ((dd.x + tx - 1) / tx) * ((dd.y + ty - 1) / ty) * ((dd.z + kF1TZ - 1) / kF1TZ);
}

// coordinate of grid point k of a row with base (rx, ry, rz): affine_along() / s2_point(), bit for bit
__device__ __forceinline__ void f1_point(const Affine &A, float rx, float ry, float rz, float kf, float &gx,
                                         float &gy, float &gz) {
  gx = fmaf(A.m[2], kf, rx) + A.m[3];
  gy = fmaf(A.m[6], kf, ry) + A.m[7];
  gz = fmaf(A.m[10], kf, rz) + A.m[11];
}

// --------------------------------------------------------------------------
// schedule build: one wave per tile
// --------------------------------------------------------------------------
struct F1BuildArgs {
  Affine A, Ainv;
  Dim3i gd, dd;
  float tol;
  int row_sep;
  int pack;  // 1: segments may sit on lanes other than their z plane's (see the packing)
  int exact;  // 1: close rows are tested point by point instead of being kept apart wholesale
};
struct F1Seg {
  short ui, uj, k0;
  unsigned char len, pos;  // points; lane (= local z plane) of the first one
};

// FILL = false: counts[slot] = {segments, instructions} of the tile; FILL = true: counts = their exclusive
// prefix sums (in processing order) and the segment entries / instruction headers are written.
template <int TX, int TY, bool FILL>
__global__ void __launch_bounds__(kWave)
    k_f1_build(F1BuildArgs B, uint2 *__restrict__ counts, const int *__restrict__ geom, uint4 *__restrict__ desc,
               uint4 *__restrict__ hdr, int *__restrict__ err, unsigned long long *__restrict__ stats) {
  constexpr int kSegs = 256;
  __shared__ F1Seg segs[kSegs];
  __shared__ unsigned short order[kSegs], tmp[kSegs];
  const int lane = threadIdx.x;
  const Dim3i dd = B.dd;
  const int slot = blockIdx.x;
  const int t = geom ? geom[slot] : slot;
  const F1TileGeom g = f1_tile<TX, TY>(t, dd);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const float flx = (float)(g.x0 - 1), fly = (float)(g.y0 - 1), flz = (float)(g.z0 - 1);
  const float fhx = (float)(g.x0 + g.ex), fhy = (float)(g.y0 + g.ey), fhz = (float)(g.z0 + g.ez);
  // acceptance box of a point: floor cell inside the aproned tile AND inside the field of view
  const float tlo = nextafterf(-B.tol, 1.f);
  const float tlx = fmaxf(flx, tlo), tly = fmaxf(fly, tlo), tlz = fmaxf(flz, tlo);
  const float thx = fminf(fhx, (float)(dd.x - 1) + B.tol), thy = fminf(fhy, (float)(dd.y - 1) + B.tol),
              thz = fminf(fhz, (float)(dd.z - 1) + B.tol);
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
      // slab clipping in real arithmetic gives a superset (with slack) of the accepted interval ...
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
      // ... then exactly: the accepted points of a row form an interval (every rounding in f1_point is
      // monotone in k), so shrinking from both ends finds it
      auto accept = [&](int k) {
        float gx, gy, gz;
        f1_point(B.A, rb.x, rb.y, rb.z, (float)k, gx, gy, gz);
        return gx >= tlx && gx < thx && gy >= tly && gy < thy && gz >= tlz && gz < thz;
      };
      while (k0 <= k1 && !accept(k0)) ++k0;
      while (k1 >= k0 && !accept(k1)) --k1;
    }
    // cut into segments: <= 32 points, every point exactly one z plane above its predecessor
    int cur = k0;
    bool more = k1 >= k0;
    while (__any(more)) {
      int e = cur;
      float za = 0.f;
      if (more) {
        float gx, gy, gz;
        f1_point(B.A, rb.x, rb.y, rb.z, (float)cur, gx, gy, gz);
        float plz = floorf(gz);
        za = plz;
        while (e + 1 <= k1 && e + 1 - cur < kF1SZ) {
          f1_point(B.A, rb.x, rb.y, rb.z, (float)(e + 1), gx, gy, gz);
          const float lz = floorf(gz);
          if (lz != plz + 1.f) break;
          plz = lz;
          ++e;
        }
      }
      const unsigned long long m = __ballot(more);
      const int pos = nseg + __popcll(m & lt_mask);
      if (more && pos < kSegs)
        segs[pos] = F1Seg{(short)ui, (short)uj, (short)cur, (unsigned char)(e - cur + 1),
                          (unsigned char)((int)za - (g.z0 - 1))};
      nseg += __popcll(m);
      cur = e + 1;
      more = more && cur <= k1;
    }
  }
  if (nseg > kSegs) {
    if (lane == 0) atomicExch(err, 1);
    nseg = kSegs;
  }
  F1_FENCE();
  __syncthreads();
  // ---- packing: lane b is instruction b of the tile; a segment joins the first instruction that has room
  // for it (below) and none of whose segments it can collide with.  The segments are offered sorted by first
  // plane, then as 0, h, 1, h + 1, ... (h = half the count): in scan order a row's successors are its
  // neighbours, the row half the cross-section away comes right behind its partner (what took the splat's
  // lane fill from 0.75 to 0.84, EXPERIMENTS E3).  Lane fill on config 2 (unrotated / rotated channels): one
  // segment per 32-lane half 0.84 / 0.44 - 0.51; any plane-disjoint set per half 0.85 / 0.65 - 0.69; close
  // rows tested point by point instead of kept apart wholesale 0.89 / 0.80 - 0.82.
  __shared__ int cnt[33];
  __shared__ unsigned short member[kWave][kF1MaxSeg];
  if (lane == 0) {
    for (int i = 0; i < 33; ++i) cnt[i] = 0;
    for (int s = 0; s < nseg; ++s) ++cnt[segs[s].pos + 1];
    for (int i = 0; i < 32; ++i) cnt[i + 1] += cnt[i];
    for (int s = 0; s < nseg; ++s) tmp[cnt[segs[s].pos]++] = (unsigned short)s;
    const int h = (nseg + 1) / 2;
    for (int s = 0; s < nseg; ++s) order[s] = tmp[(s & 1) ? h + (s >> 1) : (s >> 1)];
  }
  F1_FENCE();
  __syncthreads();
  // Lane occupancy of this instruction (64 bits).  A segment is placed, in this order of preference: (1) in the
  // first EXISTING instruction that has its plane-aligned lanes free (lane = z plane in half 0 or 1: no LDS
  // bank is asked twice), (2) - B.pack - in the first existing instruction with ANY run of free lanes long
  // enough (a rotated subject's rows cross a 4-wide tile in ~12 planes, all at different heights: plane-aligned
  // alone filled 0.65 - 0.69 of the lanes; the misplaced lanes pay a two-way bank conflict on their reads, the
  // stream has a quarter fewer instructions), (3) aligned in a new instruction.  Always subject to the
  // collision rule against every segment already there.
  unsigned long long occ = 0ull;
  unsigned char start_lane[kF1MaxSeg];
  int nmem = 0, nbins = 0;
  for (int si = 0; si < nseg; ++si) {
    const int s = order[si];
    const F1Seg q = segs[s];
    const unsigned long long lm = q.len >= 64 ? ~0ull : ((1ull << q.len) - 1ull);
    bool free = lane <= nbins && nmem < kF1MaxSeg;
    for (int j = 0; j < nmem && free; ++j) {
      const F1Seg m = segs[member[lane][j]];
      const bool rows_close = max(abs(m.ui - q.ui), abs(m.uj - q.uj)) < B.row_sep;
      const int lo = max((int)m.pos, (int)q.pos), hi = min((int)m.pos + m.len, (int)q.pos + q.len) - 1;
      if (!rows_close || lo > hi) continue;  // far apart whatever the planes, or no plane in common
      // (adjacent rows - at most one apart in both indices - overlap wherever they share a plane: no need to look)
      if (!B.exact || max(abs(m.ui - q.ui), abs(m.uj - q.uj)) < 2) {
        free = false;
        break;
      }
      // ... else exactly: two lanes meet only in the same read-add-write group on the same z plane, i.e. two
      // points with the same floor plane whose 2 x 2 cell footprints overlap.  Both segments walk one plane per
      // point, so the points that share plane pl are known; their floor cells are compared, computed with the
      // kernel's own arithmetic (rows two apart - which the blanket rule, row_sep = 3 for a rotated operator,
      // keeps out of each other's instructions - almost always pass)
      const RowBase ra = affine_row(B.A, (float)m.ui, (float)m.uj), rq = affine_row(B.A, (float)q.ui, (float)q.uj);
      for (int pl = lo; pl <= hi; ++pl) {
        float ax, ay, az, bx, by, bz;
        f1_point(B.A, ra.x, ra.y, ra.z, (float)((int)m.k0 + pl - (int)m.pos), ax, ay, az);
        f1_point(B.A, rq.x, rq.y, rq.z, (float)((int)q.k0 + pl - (int)q.pos), bx, by, bz);
        // (the kernel floors g - (tile0 - 1): for a tile on the volume's low faces the base is -1 and an accepted
        // in-tolerance point with g in (-6e-8, 0) rounds g + 1 to 1.0 - its cell moves from -1 to 0; the test uses
        // the kernel's local arithmetic, not the floor of the global coordinate)
        if (fabsf(floorf(ax - flx) - floorf(bx - flx)) < 2.f && fabsf(floorf(ay - fly) - floorf(by - fly)) < 2.f) {
          free = false;
          break;
        }
      }
    }
    const bool ok0 = free && (occ & (lm << q.pos)) == 0ull, ok1 = free && (occ & (lm << (32 + q.pos))) == 0ull;
    int any = -1;  // first run of free lanes anywhere
    if (free && B.pack && !(ok0 || ok1) && lane < nbins) {
      for (int st = 0; st + q.len <= kWave && any < 0; ++st)
        if ((occ & (lm << st)) == 0ull) any = st;
    }
    const unsigned long long m_al = __ballot((ok0 || ok1) && lane < nbins);
    const unsigned long long m_any = __ballot(any >= 0);
    const unsigned long long m_new = __ballot((ok0 || ok1) && lane == nbins);
    const unsigned long long m = m_al ? m_al : (m_any ? m_any : m_new);
    if (m == 0ull) {  // more than 64 instructions in one tile
      if (lane == 0) atomicExch(err, 1);
      break;
    }
    const int chosen = __ffsll((long long)m) - 1;
    if (lane == chosen) {
      const int st = (m_al || !m_any) ? (ok0 ? q.pos : 32 + q.pos) : any;
      member[lane][nmem] = (unsigned short)s;
      start_lane[nmem++] = (unsigned char)st;
      occ |= lm << st;
    }
    nbins = max(nbins, chosen + 1);
  }
  const int nent = lane < nbins ? nmem : 0;
  int incl = nent;  // inclusive prefix sum of the entries over the tile's instructions
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int v = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += v;
  }
  const int total_ent = __shfl(incl, kWave - 1, kWave);
  if (!FILL) {
    if (lane == 0) counts[slot] = make_uint2((unsigned)total_ent, (unsigned)nbins);
    return;
  }
  const uint2 base = counts[slot];
  unsigned long long pts = 0;
  if (lane < nbins) {
    // members in lane order (insertion sort by first lane)
    int start_of[kF1MaxSeg];
#pragma unroll
    for (int j = 0; j < kF1MaxSeg; ++j) start_of[j] = j < nmem ? (int)start_lane[j] : 1 << 20;
    unsigned long long starts = 0ull;
    uint4 *out = desc + base.x + (incl - nent);
    for (int j = 0; j < nmem; ++j) {
      // the j-th smallest start (nmem <= 8: selection by counting)
      int pick = 0;
#pragma unroll
      for (int a = 0; a < kF1MaxSeg; ++a) {
        int rank = 0;
#pragma unroll
        for (int b = 0; b < kF1MaxSeg; ++b) rank += (start_of[b] < start_of[a]) ? 1 : 0;
        if (rank == j && a < nmem) pick = a;
      }
      const F1Seg q = segs[member[lane][pick]];
      const int start = start_of[pick];
      const RowBase rb = affine_row(B.A, (float)q.ui, (float)q.uj);
      uint4 e;
      e.x = __float_as_uint(rb.x), e.y = __float_as_uint(rb.y), e.z = __float_as_uint(rb.z);
      e.w = ((unsigned)((int)q.k0 - start + (int)kF1KBias) & kF1KMask) | ((unsigned)start << 13) | ((unsigned)q.len << 19);
      out[j] = e;
      if (j > 0) starts |= 1ull << (start - 1);  // lanes >= start count it: entry = popcount below the lane
      pts += q.len;
    }
    hdr[base.y + lane] = make_uint4((unsigned)starts, (unsigned)(starts >> 32), (unsigned)(incl - nent), (unsigned)nmem);
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

void ata1_free(F1Sched &S) {
  if (S.desc) (void)hipFree(S.desc);
  if (S.hdr) (void)hipFree(S.hdr);
  if (S.tile_off) (void)hipFree(S.tile_off);
  if (S.tile_geom) (void)hipFree(S.tile_geom);
  if (S.tile_org) (void)hipFree(S.tile_org);
  if (S.scratch) (void)hipFree(S.scratch);
  S = F1Sched();
}

// tile shape: UNIRES_F1_TILE = 44 (default) | 64 | 84  (TX TY)
static void f1_shape(int &tx, int &ty) {
  static const int v = getenv("UNIRES_F1_TILE") ? atoi(getenv("UNIRES_F1_TILE")) : 44;
  tx = v / 10, ty = v % 10;
  if (!((tx == 4 || tx == 6 || tx == 8) && ty == 4)) tx = 4, ty = 4;
}

template <int TX, int TY>
static void f1_launch_build(bool fill, const F1BuildArgs &B, int nt, uint2 *counts, const int *geom, uint4 *desc,
                            uint4 *hdr, int *err, unsigned long long *stats) {
  if (fill)
    hipLaunchKernelGGL((k_f1_build<TX, TY, true>), dim3(nt), dim3(kWave), 0, 0, B, counts, geom, desc, hdr, err, stats);
  else
    hipLaunchKernelGGL((k_f1_build<TX, TY, false>), dim3(nt), dim3(kWave), 0, 0, B, counts, geom, desc, hdr, err, stats);
}

int ata1_build(F1Sched &S, const Affine &A, const Affine &Ainv, Dim3i gd, Dim3i dd, float tol,
               const SplatSafety &safe) {
  S.valid = false;
  static const bool off = getenv("UNIRES_NO_ATA1") != nullptr;
  static const bool verbose = getenv("UNIRES_ATA1_VERBOSE") != nullptr;
  if (off) return 1;
  if (safe.use_atomics) return 1;
  if (!(A.m[10] > 0.f)) return 1;  // lane = z plane needs grid z to run up the output's z (canonical layout)
  if (dd.x > 4000 || dd.y > 4000 || dd.z > 4000 || gd.x > 4000 || gd.y > 4000 || gd.z > 4000) return 1;
  if (!fits_fast_index(gd) || !fits_fast_index(dd) || dd.numel() >= (1ull << 30)) return 1;
  int tx, ty;
  f1_shape(tx, ty);
  const int nt = f1_ntiles(dd, tx, ty);
  if ((size_t)nt + 1 > S.cap_tiles) {
    if (S.tile_off) (void)hipFree(S.tile_off);
    S.tile_off = nullptr;
    if (hipMalloc((void **)&S.tile_off, ((size_t)nt + 1) * sizeof(uint2)) != hipSuccess) return 1;
    if (S.tile_geom) (void)hipFree(S.tile_geom);
    S.tile_geom = nullptr;
    if (hipMalloc((void **)&S.tile_geom, ((size_t)nt + 1) * sizeof(int)) != hipSuccess) return 1;
    if (S.tile_org) (void)hipFree(S.tile_org);
    S.tile_org = nullptr;
    if (hipMalloc((void **)&S.tile_org, ((size_t)nt + 1) * sizeof(uint2)) != hipSuccess) return 1;
    S.cap_tiles = (size_t)nt + 1;
  }
  if (!S.scratch && hipMalloc((void **)&S.scratch, 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
  (void)hipMemset(S.scratch, 0, 4 * sizeof(unsigned long long));
  int *err_dev = (int *)S.scratch;
  unsigned long long *stats_dev = S.scratch + 1;
  F1BuildArgs B;
  B.A = A, B.Ainv = Ainv, B.gd = gd, B.dd = dd, B.tol = tol, B.row_sep = safe.row_sep;
  static const int pack = getenv("UNIRES_F1_PACK") ? atoi(getenv("UNIRES_F1_PACK")) : 1;
  B.pack = pack;
  static const int exact = getenv("UNIRES_F1_EXACT") ? atoi(getenv("UNIRES_F1_EXACT")) : -1;  // (-1: as the plan asks)
  B.exact = exact >= 0 ? exact : (sched_thorough() ? 1 : 0);
  auto run = [&](bool fill, const int *geom) {
    if (tx == 8)
      f1_launch_build<8, 4>(fill, B, nt, S.tile_off, geom, S.desc, S.hdr, err_dev, fill ? stats_dev : nullptr);
    else if (tx == 6)
      f1_launch_build<6, 4>(fill, B, nt, S.tile_off, geom, S.desc, S.hdr, err_dev, fill ? stats_dev : nullptr);
    else
      f1_launch_build<4, 4>(fill, B, nt, S.tile_off, geom, S.desc, S.hdr, err_dev, fill ? stats_dev : nullptr);
  };
  run(false, nullptr);
  std::vector<uint2> cnt2((size_t)nt), h((size_t)nt + 1);
  if (hipMemcpy(cnt2.data(), S.tile_off, (size_t)nt * sizeof(uint2), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  std::vector<unsigned> cnt((size_t)nt);
  for (int i = 0; i < nt; ++i) cnt[i] = cnt2[i].y;
  // Processing order, as the splat's (splat2_build): contiguous runs of tiles per XCD with equal COST (a
  // tile costs its instructions + a constant for staging and epilogue); inside a run the tiles with (next
  // to) no instructions go last, emptiest at the very end.
  static const double kTileCost = getenv("UNIRES_F1_TILE_COST") ? atof(getenv("UNIRES_F1_TILE_COST")) : 5.0;
  std::vector<int> geom((size_t)nt + 1);
  {
    double total = 0.0;
    unsigned imax = 0;
    for (int i = 0; i < nt; ++i) total += (double)cnt[i] + kTileCost, imax = std::max(imax, cnt[i]);
    int x = 1;
    double cum = 0.0;
    S.xcd_lo[0] = 0;
    for (int i = 0; i < nt && x < 8; ++i) {
      cum += (double)cnt[i] + kTileCost;
      while (x < 8 && cum >= total * x / 8.0) S.xcd_lo[x++] = i + 1;
    }
    for (; x <= 8; ++x) S.xcd_lo[x] = nt;
    const unsigned cheap = imax / 2u;
    for (int xc = 0; xc < 8; ++xc) {
      const int lo = S.xcd_lo[xc], hi = S.xcd_lo[xc + 1];
      int u = lo;
      for (int g = lo; g < hi; ++g)
        if (cnt[g] > cheap) geom[u++] = g;
      const int first_cheap = u;
      for (int g = lo; g < hi; ++g)
        if (cnt[g] <= cheap) geom[u++] = g;
      std::stable_sort(geom.begin() + first_cheap, geom.begin() + hi, [&](int a, int b) { return cnt[a] > cnt[b]; });
    }
    geom[nt] = 0;
  }
  unsigned re = 0, ri = 0;
  for (int u = 0; u < nt; ++u) {
    h[u] = make_uint2(re, ri);
    re += cnt2[geom[u]].x, ri += cnt2[geom[u]].y;
  }
  h[nt] = make_uint2(re, ri);
  constexpr size_t kPad = 16;  // entries / headers read (never used) past the end by the prefetch
  if ((size_t)re + kPad > S.cap_entries) {
    if (S.desc) (void)hipFree(S.desc);
    S.desc = nullptr;
    const size_t cap = (size_t)re + re / 8 + kPad;
    if (hipMalloc((void **)&S.desc, cap * sizeof(uint4)) != hipSuccess) return 1;
    S.cap_entries = cap;
  }
  if ((size_t)ri + 64 + kPad > S.cap_instr) {
    if (S.hdr) (void)hipFree(S.hdr);
    S.hdr = nullptr;
    const size_t cap = (size_t)ri + ri / 8 + 64 + kPad;  // (a wave reads 64 headers from its tile's first on)
    if (hipMalloc((void **)&S.hdr, cap * sizeof(uint4)) != hipSuccess) return 1;
    S.cap_instr = cap;
  }
  if (hipMemcpy(S.tile_off, h.data(), ((size_t)nt + 1) * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess)
    return 1;
  if (hipMemcpy(S.tile_geom, geom.data(), ((size_t)nt + 1) * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return 1;
  {
    std::vector<uint2> org((size_t)nt + 1);
    for (int u = 0; u <= nt; ++u) {
      const F1TileGeom g = tx == 8 ? f1_tile<8, 4>(geom[u], dd) : tx == 6 ? f1_tile<6, 4>(geom[u], dd) : f1_tile<4, 4>(geom[u], dd);
      org[u] = make_uint2((unsigned)g.x0 | ((unsigned)g.y0 << 16), (unsigned)g.z0);
    }
    if (hipMemcpy(S.tile_org, org.data(), ((size_t)nt + 1) * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) return 1;
  }
  (void)hipMemset(S.desc + re, 0, kPad * sizeof(uint4));
  (void)hipMemset(S.hdr + ri, 0, (64 + kPad) * sizeof(uint4));
  run(true, S.tile_geom);
  int herr = 0;
  unsigned long long hs[2] = {0, 0};
  if (hipMemcpy(&herr, err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (hipMemcpy(hs, stats_dev, sizeof(hs), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (herr) {
    if (verbose) fprintf(stderr, "[ata1] a tile exceeds the segment / instruction lists: two-kernel path used\n");
    return 1;
  }
  S.ntiles = nt, S.tx = tx, S.ty = ty;
  S.fill = hs[1] ? (double)hs[0] / (64.0 * (double)hs[1]) : 0.0;
  S.visits = (double)hs[0] / (double)dd.numel();
  // (a grid much finer or coarser than the output cuts every row into crumbs: the pair of kernels does better)
  static const double min_fill = getenv("UNIRES_F1_MIN_FILL") ? atof(getenv("UNIRES_F1_MIN_FILL")) : 0.45;
  if (verbose) {
    unsigned emax = 0;
    for (int i = 0; i < nt; ++i) emax = std::max(emax, cnt2[i].x);
    fprintf(stderr, "[ata1] tile %d x %d x %d: %d tiles, %llu instructions, %llu points (%.2f per output voxel), "
            "lane fill %.3f, schedule %.1f MB, row_sep %d, max segments per tile %u\n", tx, ty, kF1TZ, nt, hs[1], hs[0],
            S.visits, S.fill, (re + ri) * 16.0 / 1e6, safe.row_sep, emax);
  }
  // (lane = z plane: a volume whose z extent does not fill its layers of tiles cannot fill the lanes whatever the
  // operator - 33 planes are a full layer and one of 3 planes: judged against what its layers can hold)
  const int ntz = (dd.z + kF1TZ - 1) / kF1TZ;
  const double reach = std::min(1.0, (double)(dd.z + ntz) / (double)(ntz * kF1SZ));
  if (hs[1] > 0 && S.fill < min_fill * reach) return 1;
  S.valid = true;
  return 0;
}

// --------------------------------------------------------------------------
// the kernel
// --------------------------------------------------------------------------
struct F1Args {
  const float *src;
  const uint4 *desc;
  size_t desc_bytes;
  const uint4 *hdr;
  const uint2 *tile_off;
  const uint2 *tile_org;  // {x0 | y0 << 16, z0} of the tile in processing slot u
  int ntiles;
  Affine A;
  float alpha;
  float a0, cx, cy, cz;
  float *dst;
  Dim3i dd;
  int accumulate;
  double *partials;
  const float *objb;
  int want_dot;
  int xlo[9];
  int active;
  int prio_rot;
  int dbg;  // UNIRES_F1_DBG ablation bits (read only by -DUNIRES_ABLATE builds)
};

// MODE 0: dst = q (+ dot p q if asked); 1: objective mode (partials = sum (q - 2 objb) p, q not stored);
// 2: general (dst += q first - a later repeat of a multi-repeat operator - then as 0 or 1)
template <int TX, int TY, int NW, int MODE>
__global__ void __launch_bounds__(kWave *NW) k_ata1(F1Args P, const int *__restrict__ done) {
  if (done && *done) return;
  constexpr int SX = TX + 2, SY = TY + 2, SZ = kF1SZ, N = SX * SY * SZ, XS = SY * SZ, YS = SZ;
  constexpr int HX = TX / 2;
  static_assert(TX % 2 == 0 && TY % 2 == 0 && N % 4 == 0, "tile shape");
  static_assert(XS + YS + 1 < 256, "ds_read2 / ds_write2 offsets are 8 bits");
  __shared__ __align__(16) float win_all[NW][N];
  __shared__ __align__(16) float acc_all[NW][N];
  __shared__ __align__(16) uint4 ring_all[NW][kWave];  // segment entries: 2 chunks of 32
  const int lane = threadIdx.x & (kWave - 1), grp = lane >> 5, gl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *win = win_all[wave];
  float *acc = acc_all[wave];
  uint4 *ring = ring_all[wave];
  const Dim3i dd = P.dd;
  float *__restrict__ dst = P.dst;
  if ((int)blockIdx.x >= P.active) {
    if (P.partials && lane == 0) P.partials[blockIdx.x * NW + wave] = 0.0;
    return;
  }
  // XCD-aware persistent schedule (as k_splat2): workgroup b sits on XCD b % 8; each XCD walks one
  // contiguous run of tiles of equal cost, so neighbouring tiles' windows share an L2
  const int nwg = P.active;
  const int nxcd = min(8, nwg);
  const int xcd = blockIdx.x % nxcd;
  const int t_lo = P.xlo[xcd], t_hi = P.xlo[xcd + 1];
  const int slot = (blockIdx.x / nxcd) * NW + wave;
  const int slots = ((nwg + nxcd - 1 - xcd) / nxcd) * NW;
  const float c0 = P.A.m[2], c1 = P.A.m[6], c2 = P.A.m[10];
  const float t0 = P.A.m[3], t1 = P.A.m[7], t2 = P.A.m[11];
  const size_t nbytes = dd.numel() * sizeof(float);
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(P.src, nbytes), rd = make_rsrc(dst, nbytes),
                               rb = make_rsrc(P.objb ? P.objb : P.src, nbytes),
                               re = make_rsrc(P.desc, P.desc_bytes);
  const unsigned sxb = 4u * (unsigned)(dd.y * dd.z), syb = 4u * (unsigned)dd.z;
  constexpr unsigned kOob = 0x80000000u;
  double dot = 0.0;
  const int hw_slot = (int)(__builtin_amdgcn_s_getreg(6148) & 3u);  // HW_ID.wave_id
  // ---- stage a tile's window: the tile + one cell all round, planes z0 - 1 .. z0 + 30 in lanes 0 .. 31 of each
  // half; one LDS-DMA dword per lane moves two y rows per instruction (no VGPR round trip, no ds_write).  Cells
  // outside the volume come from an out-of-range buffer offset: zeros, which is the reference's zero bound for
  // the gather AND what the stencil's forward differences want at the volume's far faces.
  auto stage = [&](int x0, int y0, int z0) {
    if (F1_ABL(4)) return;
    const int kz = z0 - 1 + gl;
    const bool zok = (unsigned)kz < (unsigned)dd.z;
    unsigned voff[SY / 2];
#pragma unroll
    for (int m = 0; m < SY / 2; ++m) {
      const int y = y0 - 1 + 2 * m + grp;
      voff[m] = (zok && (unsigned)y < (unsigned)dd.y) ? (unsigned)y * syb + 4u * (unsigned)kz : kOob;
    }
    if (x0 >= 1 && x0 + TX < dd.x) {  // every x slab of the window inside the volume
      const unsigned s0 = (unsigned)(x0 - 1) * sxb;
#pragma unroll
      for (int lx = 0; lx < SX; ++lx)
#pragma unroll
        for (int m = 0; m < SY / 2; ++m)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(win + (lx * SY + 2 * m) * SZ),
                                                   4, voff[m], s0 + (unsigned)lx * sxb, 0, 0);
    } else {
#pragma unroll
      for (int lx = 0; lx < SX; ++lx) {
        const int x = x0 - 1 + lx;
        const bool xok = (unsigned)x < (unsigned)dd.x;  // wave-uniform
        const unsigned soff = xok ? (unsigned)x * sxb : 0u;
        const unsigned kill = xok ? 0u : kOob;
#pragma unroll
        for (int m = 0; m < SY / 2; ++m)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(win + (lx * SY + 2 * m) * SZ),
                                                   4, voff[m] | kill, soff, 0, 0);
      }
    }
  };
  // ... and its first 64 segment entries into the ring (one 16-byte LDS-DMA piece per lane)
  auto stage_entries = [&](unsigned first) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (__attribute__((address_space(3))) void *)ring, 16,
                                             (first + (unsigned)lane) * 16u, 0, 0, 0);
  };
  auto origin = [&](int t, int &x0, int &y0, int &z0) {  // (host-tabulated: the index arithmetic costs two divisions)
    const uint2 g = P.tile_org[t];
    x0 = __builtin_amdgcn_readfirstlane((int)(g.x & 0xffffu)), y0 = __builtin_amdgcn_readfirstlane((int)(g.x >> 16)),
    z0 = __builtin_amdgcn_readfirstlane((int)g.y);
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < (N / 4 + kWave - 1) / kWave; ++j)
      if (lane + j * kWave < N / 4) reinterpret_cast<float4 *>(acc)[lane + j * kWave] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  int t = t_lo + slot;
  int x0 = 0, y0 = 0, z0 = 0;
  uint4 myh = make_uint4(0u, 0u, 0u, 0u);
  uint2 off0 = make_uint2(0u, 0u);
  int nent = 0, ninstr = 0;
  if (t < t_hi) {
    origin(t, x0, y0, z0);
    off0 = P.tile_off[t];
    const uint2 off1 = P.tile_off[t + 1];
    nent = (int)(off1.x - off0.x), ninstr = (int)(off1.y - off0.y);
    stage(x0, y0, z0);
    stage_entries(off0.x);
    myh = P.hdr[off0.y + lane];
  }
  zero_acc();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first tile: nothing younger than its requests is in flight)
  int round = 0;
  for (; t < t_hi; t += slots, ++round) {
    if (P.prio_rot) {  // (the SIMD issues from its oldest wave first: rotate the ranking, see k_splat2)
      switch ((hw_slot + round) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
    const int ey = min(TY, dd.y - y0), ez = min(kF1TZ, dd.z - z0);
    // lane l keeps the header of instruction l (a tile has at most 64): segment-start mask, first entry
    const int hlo_v = (int)myh.x, hhi_v = (int)myh.y, hfe_v = (int)myh.z;
    // The window, the ring and the header were requested one tile ahead (at the previous epilogue's head);
    // what may still be in flight behind them are that epilogue's stores, which nobody waits for.
    if (MODE == 0)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HX * TY) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    F1_FENCE();
    // ---- the instruction stream: gather + scatter of up to 64 grid points at a time.
    // (Tried and dropped: the stream software-pipelined - instruction i + 1 decoded and its gather issued
    // between the two read-add-write groups of instruction i, the entry of i + 2 requested from the ring: a
    // wave's chain per instruction is then two LDS round trips instead of five, and the kernel is SLOWER,
    // 47.1 / 49.3 us against 44.4 / 47.3 (config 2, hot / cold operands): the carried state costs ~15 more
    // instructions per stream instruction, and issue slots - not a wave's latency chain - are what four waves
    // per SIMD run out of: 3.6e7 issues per launch x 2.1 clocks = 31 us of the kernel's 45.)
    const float xb = (float)(x0 - 1), yb = (float)(y0 - 1), zb = (float)(z0 - 1);
    int loaded_hi = 64;  // entries of this tile requested so far (the ring holds the last 64 of them)
    bool pending = false;
    auto one = [&](int i, auto big_tag) {
      constexpr bool BIG = decltype(big_tag)::value;
      const unsigned mlo = (unsigned)__builtin_amdgcn_readlane(hlo_v, i), mhi = (unsigned)__builtin_amdgcn_readlane(hhi_v, i);
      const int fe = __builtin_amdgcn_readlane(hfe_v, i);
      // (tiles with more than 64 segments: once the stream has moved into the younger chunk the older one
      // is refilled - 32 entries, lanes 0 .. 31 - and waited for when an instruction first reaches past
      // what has landed)
      if (BIG) {
        const int fe_next = fe + __popc(mlo) + __popc(mhi) + 1;
        if (pending && fe_next > loaded_hi - 32) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          pending = false;
        }
        if (!pending && fe >= loaded_hi - 32 && loaded_hi < nent) {
          if (lane < 32)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (__attribute__((address_space(3))) void *)(ring + (loaded_hi & 32)), 16,
                                                     (off0.x + (unsigned)loaded_hi + (unsigned)lane) * 16u, 0, 0, 0);
          loaded_hi += 32;
          pending = true;
        }
      }
      const int sl = fe + (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
      uint4 d = ring[sl & 63];
      asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));  // (one 16-byte read, not a dword + 12 bytes with a wait between)
      const unsigned pk = d.w;
      const int kb = (int)(pk & kF1KMask) - (int)kF1KBias;
      const unsigned pos = (pk >> 13) & 63u, len = pk >> 19;
      const float kf = (float)(kb + lane);
      const float gx = fmaf(c0, kf, __uint_as_float(d.x)) + t0;
      const float gy = fmaf(c1, kf, __uint_as_float(d.y)) + t1;
      const float gz = fmaf(c2, kf, __uint_as_float(d.z)) + t2;
      // local coordinates: the subtraction of the (integer) tile base is exact
      const float lxf = gx - xb, lyf = gy - yb, lzf = gz - zb;
      const float wx1 = __builtin_amdgcn_fractf(lxf), wy1 = __builtin_amdgcn_fractf(lyf),
                  wz1 = __builtin_amdgcn_fractf(lzf);
      const float wz0 = 1.f - wz1;
      const float cf = fmaf(lxf - wx1, (float)XS, fmaf(lyf - wy1, (float)YS, lzf - wz1));
      const int cell = (int)cf;
      if ((unsigned)(lane - (int)pos) < len) {
        const float *w = win + (F1_ABL(16) ? 0 : cell);
        // gather: four z pairs of the window (ds_read2_b32 offsets 0, 1) -> the trilinear sample (lerps)
        const float p000 = w[0], p001 = w[1], p010 = w[YS], p011 = w[YS + 1];
        const float p100 = w[XS], p101 = w[XS + 1], p110 = w[XS + YS], p111 = w[XS + YS + 1];
        const float z00 = fmaf(wz1, p001 - p000, p000), z01 = fmaf(wz1, p011 - p010, p010);
        const float z10 = fmaf(wz1, p101 - p100, p100), z11 = fmaf(wz1, p111 - p110, p110);
        const float y0v = fmaf(wy1, z01 - z00, z00), y1v = fmaf(wy1, z11 - z10, z10);
        const float v = fmaf(wx1, y1v - y0v, y0v);
        // scatter weights: v wx wy, with 1 - f formed as v - v f (one FMA less per product pair)
        const float vx1 = v * wx1, vx0 = v - vx1;
        const float a01 = vx0 * wy1, a00 = vx0 - a01, a11 = vx1 * wy1, a10 = vx1 - a11;
        float *q = acc + cell;
        if (F1_ABL(8)) {
          if (a00 + a01 + a10 + a11 == 123.f) q[0] = 1.f;
        } else {
        {
          const float o00 = q[0], o01 = q[YS], o10 = q[XS], o11 = q[XS + YS];
          q[0] = o00 + a00 * wz0, q[YS] = o01 + a01 * wz0, q[XS] = o10 + a10 * wz0, q[XS + YS] = o11 + a11 * wz0;
        }
        F1_FENCE();
        {
          const float o00 = q[1], o01 = q[YS + 1], o10 = q[XS + 1], o11 = q[XS + YS + 1];
          q[1] = o00 + a00 * wz1, q[YS + 1] = o01 + a01 * wz1, q[XS + 1] = o10 + a10 * wz1,
          q[XS + YS + 1] = o11 + a11 * wz1;
        }
        }
      }
      F1_FENCE();
    };
    if (F1_ABL(1)) {
    } else if (nent > 64) {
      for (int i = 0; i < ninstr; ++i) one(i, std::true_type{});
    } else {
      for (int i = 0; i < ninstr; ++i) one(i, std::false_type{});
    }
    if (pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a refill nobody used: the ring is rewritten below)
    F1_FENCE();
    // ---- epilogue: q = [q +] alpha acc + a0 p + c DtD p ; dot += p q.  Lane gl of a half holds z plane
    // z0 - 1 + gl; half h owns x slabs h HX .. h HX + HX - 1.  p comes from the window (read once into a
    // register window), its z neighbours from the adjacent lanes (DPP wave shifts).  Once the register
    // window is read the LDS window is dead: the NEXT tile's window, entries and header are requested right
    // here (LDS-DMA) and travel while this tile's outputs are formed and stored.
    {
      const int kz = z0 - 1 + gl;
      const bool out_z = gl >= 1 && gl <= ez, lz_ok = kz > 0, hz_ok = kz + 1 < dd.z;
      const int xg = x0 + HX * grp;
      const float *wrow = win + (HX * grp * SY) * SZ + gl;
      const float *arow = acc + (HX * grp * SY) * SZ + gl;
      float pv[HX + 2][SY], av[HX][TY];
#pragma unroll
      for (int sa = 0; sa < HX + 2; ++sa)
#pragma unroll
        for (int la = 0; la < SY; ++la) {
          const bool corner = (sa == 0 || sa == HX + 1) && (la == 0 || la == SY - 1);
          pv[sa][la] = corner ? 0.f : wrow[(sa * SY + la) * SZ];
        }
#pragma unroll
      for (int sa = 1; sa <= HX; ++sa)
#pragma unroll
        for (int la = 1; la <= TY; ++la) av[sa - 1][la - 1] = arow[(sa * SY + la) * SZ];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      F1_FENCE();
      const int tn = t + slots;
      int x0n = 0, y0n = 0, z0n = 0, nentn = 0, ninstrn = 0;
      uint2 off0n = make_uint2(0u, 0u);
      if (tn < t_hi) {
        origin(tn, x0n, y0n, z0n);
        off0n = P.tile_off[tn];
        const uint2 off1n = P.tile_off[tn + 1];
        nentn = (int)(off1n.x - off0n.x), ninstrn = (int)(off1n.y - off0n.y);
        stage(x0n, y0n, z0n);
        stage_entries(off0n.x);
        myh = P.hdr[off0n.y + lane];
      }
      zero_acc();
      if (F1_ABL(2)) {
        if (pv[1][1] + av[0][0] == 123.f) dst[0] = 1.f;
        x0 = x0n, y0 = y0n, z0 = z0n, off0 = off0n, nent = nentn, ninstr = ninstrn;
        continue;
      }
      // (per-lane byte offset of (first owned slab, first owned row, own plane): the half's slab goes into the
      // VECTOR offset - a scalar offset that differs between the halves makes every store a waterfall loop)
      const unsigned e1 = (unsigned)xg * sxb + (unsigned)y0 * syb + 4u * (unsigned)kz;
      float ob[HX][TY], od[HX][TY];
      if (MODE != 0) {  // the epilogue's loads, all issued before the first is used
#pragma unroll
        for (int sa = 1; sa <= HX; ++sa)
#pragma unroll
          for (int la = 1; la <= TY; ++la) {
            const bool ok = out_z && xg + sa - 1 < dd.x && la - 1 < ey;
            const unsigned so = (unsigned)(sa - 1) * sxb + (unsigned)(la - 1) * syb;
            ob[sa - 1][la - 1] = P.objb ? buf_load(rb, ok ? e1 : kOob, so) : 0.f;
            od[sa - 1][la - 1] = P.accumulate ? buf_load(rd, ok ? e1 : kOob, so) : 0.f;
          }
      }
#pragma unroll
      for (int sa = 1; sa <= HX; ++sa) {
        const int x = xg + sa - 1;
        const bool x_ok = x < dd.x;
#pragma unroll
        for (int la = 1; la <= TY; ++la) {
          const bool ok = out_z && x_ok && la - 1 < ey;
          const unsigned vo = ok ? e1 : kOob;
          const unsigned so = (unsigned)(sa - 1) * sxb + (unsigned)(la - 1) * syb;  // wave-uniform
          const float c = pv[sa][la];
          const float vzm = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x138, 0xf, 0xf, false));
          const float vzp = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x130, 0xf, 0xf, false));
          float q = P.alpha * av[sa - 1][la - 1];
          // (forward differences with a zero bound: the backward term of the volume's first slab is absent,
          // the forward term of its last slab sees the window's zero)
          const float xf = pv[sa + 1][la] - c, xbk = x == 0 ? 0.f : c - pv[sa - 1][la];
          const float yf = pv[sa][la + 1] - c, ybk = (la == 1 && y0 == 0) ? 0.f : c - pv[sa][la - 1];
          const float zf = (hz_ok ? vzp : 0.f) - c, zbk = lz_ok ? c - vzm : 0.f;
          const float st = P.cx * (xbk - xf) + P.cy * (ybk - yf) + P.cz * (zbk - zf);
          q += P.a0 * c + st;
          if (MODE == 2) q += od[sa - 1][la - 1];
          if (MODE == 1 || (MODE == 2 && P.objb)) {
            if (ok) dot += (double)obj_term(q, ob[sa - 1][la - 1], c);
          } else {
            buf_store(q, rd, vo, so);  // (lanes that own nothing point out of range: dropped)
            if (ok && P.want_dot) dot += (double)__fmul_rn(c, q);
          }
        }
      }
      x0 = x0n, y0 = y0n, z0 = z0n, off0 = off0n, nent = nentn, ninstr = ninstrn;
    }
    F1_FENCE();
  }
  if (P.partials) {
    const double tot = wave_sum(dot);
    if (lane == 0) P.partials[blockIdx.x * NW + wave] = tot;
  }
}

static int f1_grid(Dim3i dd, int tx, int ty, int share_cap = 0) {
  const int nt = f1_ntiles(dd, tx, ty);
  static const int cap_env = getenv("UNIRES_F1_BLOCKS") ? atoi(getenv("UNIRES_F1_BLOCKS")) : 0;
  const int cap = cap_env > 0 ? cap_env : (share_cap >= 8 ? share_cap : 1024);
  const int want = (nt + kF1Waves - 1) / kF1Waves;
  return want < cap ? want : cap;
}

int ata1_blocks(Dim3i dd, int grid_cap) {
  int tx, ty;
  f1_shape(tx, ty);
  return f1_grid(dd, tx, ty, grid_cap) * kF1Waves;
}

static const void *f1_fn(int tx) {
  return tx == 8   ? (const void *)k_ata1<8, 4, kF1Waves, 0>
         : tx == 6 ? (const void *)k_ata1<6, 4, kF1Waves, 0>
                   : (const void *)k_ata1<4, 4, kF1Waves, 0>;
}

// workgroups the device holds at once, rounded down to whole rounds over the 8 XCDs
static int f1_active(int tx, int grid) {
  static std::map<int, int> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(tx);
  if (it == cache.end()) {
    int per_cu = 0, dev = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f1_fn(tx), kWave * kF1Waves, 0) != hipSuccess) per_cu = 0;
    static const int force = getenv("UNIRES_F1_RESIDENT") ? atoi(getenv("UNIRES_F1_RESIDENT")) : 0;
    if (force > 0) per_cu = force;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
    const int n = per_cu > 0 && ncu > 0 ? per_cu * ncu : 1 << 30;
    static const bool verbose = getenv("UNIRES_ATA1_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[ata1] %d workgroups per CU resident (tile %d x 4)\n", per_cu, tx);
    it = cache.emplace(tx, n).first;
  }
  int n = std::min(grid, it->second);
  if (n >= 8) n -= n % 8;
  return n;
}

int launch_ata1(const F1Sched &S, const float *src, const Affine &A, float alpha, const PushEpilogue &ep,
                float *dst, Dim3i dd, const int *done, hipStream_t st) {
  int tx, ty;
  f1_shape(tx, ty);
  if (!S.valid || S.tx != tx || S.ty != ty || S.ntiles != f1_ntiles(dd, tx, ty)) return 1;
  if (ep.p && ep.p != src) return 1;
  if (dd.numel() >= (1ull << 30)) return 1;
  F1Args P;
  P.src = src, P.desc = S.desc, P.desc_bytes = S.cap_entries * sizeof(uint4), P.hdr = S.hdr, P.tile_off = S.tile_off, P.tile_org = S.tile_org, P.ntiles = S.ntiles;
  P.A = A, P.alpha = alpha;
  P.a0 = ep.p ? ep.a0 : 0.f, P.cx = ep.p ? ep.cx : 0.f, P.cy = ep.p ? ep.cy : 0.f, P.cz = ep.p ? ep.cz : 0.f;
  P.dst = dst, P.dd = dd, P.accumulate = ep.accumulate, P.partials = ep.partials, P.objb = ep.partials ? ep.objb : nullptr;
  P.want_dot = ep.partials != nullptr;
  const dim3 grid(f1_grid(dd, tx, ty, ep.grid_cap)), block(kWave * kF1Waves);
  if (grid.x >= 8) {
    for (int x = 0; x <= 8; ++x) P.xlo[x] = S.xcd_lo[x];
  } else {
    for (int x = 0; x <= 8; ++x) P.xlo[x] = (int)std::min<long long>(S.ntiles, ((long long)S.ntiles * x + grid.x - 1) / grid.x);
  }
  static const int prio_rot = getenv("UNIRES_F1_PRIO") ? atoi(getenv("UNIRES_F1_PRIO")) : 1;
  P.prio_rot = prio_rot;
  static const int dbg = getenv("UNIRES_F1_DBG") ? atoi(getenv("UNIRES_F1_DBG")) : 0;
  P.dbg = dbg;
  P.active = f1_active(tx, (int)grid.x);
  const int mode = P.accumulate ? 2 : (P.objb ? 1 : 0);
#define F1_LAUNCH(TX_)                                                                                   \
  do {                                                                                                   \
    if (mode == 0)                                                                                       \
      hipLaunchKernelGGL((k_ata1<TX_, 4, kF1Waves, 0>), grid, block, 0, st, P, done);                    \
    else if (mode == 1)                                                                                  \
      hipLaunchKernelGGL((k_ata1<TX_, 4, kF1Waves, 1>), grid, block, 0, st, P, done);                    \
    else                                                                                                 \
      hipLaunchKernelGGL((k_ata1<TX_, 4, kF1Waves, 2>), grid, block, 0, st, P, done);                    \
  } while (0)
  if (tx == 8)
    F1_LAUNCH(8);
  else if (tx == 6)
    F1_LAUNCH(6);
  else
    F1_LAUNCH(4);
#undef F1_LAUNCH
  return 0;
}

}  // namespace unires
