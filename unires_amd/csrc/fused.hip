// fused.hip - the two kernels of the fused CG matvec (regimes 1 and 2):
//
//   k_pull_conv : xs = S . conv_down . pull_M (p)            y-space -> x-space
//   k_push_tile : q  = [q +] alpha * push_M(conv_up(S xs)) + a0 p + c DtD p  (+ sum p*q)
//
// so one application of  tau AtA + rho lam^2 DtD  reads p once, writes/reads
// the small x-space intermediate once, and writes q once (B_mv of SURVEY 8(d)).
// The grid-space (yx) volume never exists in HBM: k_pull_conv keeps its pulled
// tile in LDS, k_push_tile regenerates conv_up values on the fly.
//
// k_push_tile is an OWNER-COMPUTES scatter: a workgroup owns a TXxTYxTZ tile of
// the output, enumerates the grid rows (ui,uj) whose image crosses the tile and,
// per row, the exact interval of grid-z that lands in it, then splats the 8
// trilinear corners of every such source voxel into an LDS accumulator with
// ds_add_f32.  No global atomics; q is written exactly once, coalesced, with
// the DtD stencil and the CG dot product fused into the same epilogue.
#include "fused.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

namespace unires {

// exact for 0 <= n < 2^22, d >= 1 when inv = 1/d rounded to nearest
__device__ __forceinline__ int fast_div(int n, int d, float inv) {
  int q = (int)(((float)n + 0.5f) * inv);
  // one correction step makes it exact regardless of rounding
  if (q * d > n) --q;
  if ((q + 1) * d <= n) ++q;
  return q;
}

// --------------------------------------------------------------------------
// k_pull_conv
// --------------------------------------------------------------------------
// A workgroup owns an x-space tile ot; the grid-space box it needs (pt = (ot-1)*s + K
// per axis) is pulled into LDS row by row - one wave per grid row (ux,uy), lanes along
// grid z, so the y-space reads of a wave are (nearly) contiguous - then reduced by the
// separable taps.  Tiles whose whole footprint is inside the volume take the
// `pull_interior` path (no masks / clamps).
__global__ void __launch_bounds__(kBlock)
    k_pull_conv(const float *__restrict__ src, Dim3i sd, Affine A, Taps T, Scaling S,
                float *__restrict__ dst, Dim3i xd, Dim3i gd, Dim3i ot, float tol,
                const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ float smem[];
  const int ptx = (ot.x - 1) * T.s[0] + T.n[0];
  const int pty = (ot.y - 1) * T.s[1] + T.n[1];
  const int ptz = (ot.z - 1) * T.s[2] + T.n[2];
  const int ptot = ptx * pty * ptz;
  float *tile = smem;
  float *taps = smem + ptot;  // 3 * UNIRES_MAX_TAPS
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row loop runs on the SALU
  if (tid < 3 * UNIRES_MAX_TAPS) taps[tid] = (&T.t[0][0])[tid];
  const int o0x = blockIdx.z * ot.x, o0y = blockIdx.y * ot.y, o0z = blockIdx.x * ot.z;
  const int p0x = o0x * T.s[0], p0y = o0y * T.s[1], p0z = o0z * T.s[2];
  const unsigned ny = sd.y, nz = sd.z, nynz = ny * nz;
  const int nrows = ptx * pty;
  const float bx = (float)(sd.x - 1), by = (float)(sd.y - 1), bz = (float)(sd.z - 1);
  for (int row = wave; row < nrows; row += kBlock / kWave) {
    const int a = row / pty, b = row - a * pty;
    const int ux = p0x + a, uy = p0y + b;
    const bool row_in = ux < gd.x && uy < gd.y;
    const RowBase rb = affine_row(A, (float)min(ux, gd.x - 1), (float)min(uy, gd.y - 1));
    float *trow = tile + row * ptz;
    for (int c0 = 0; c0 < ptz; c0 += kWave) {
      const int c = c0 + lane;
      const int uz = p0z + min(c, ptz - 1);
      float gx, gy, gz;
      affine_along(A, rb, (float)min(uz, gd.z - 1), gx, gy, gz);
      // all 8 corners of every lane's sample inside the volume (then the FOV mask is 1 too):
      // decided per wave-pass, so only the thin boundary shell takes the general path
      const bool inside = gx >= 0.f && gx < bx && gy >= 0.f && gy < by && gz >= 0.f && gz < bz;
      // ... and passes whose samples are ALL outside the field of view are exact zeros
      const bool outside = gx <= -tol || gx >= bx + tol || gy <= -tol || gy >= by + tol ||
                           gz <= -tol || gz >= bz + tol;
      float v;
      if (__all(inside) && sd.z >= 2 && fits_fast_index(sd))
        v = pull_interior(src, ny, nz, nynz, gx, gy, gz);
      else if (__all(outside))
        v = 0.f;
      else
        v = pull_sample(src, sd, gx, gy, gz, tol);
      if (c < ptz) trow[c] = (row_in && p0z + c < gd.z) ? v : 0.f;
    }
  }
  __syncthreads();
  const int otot = ot.x * ot.y * ot.z;
  const float inv_oz = 1.f / (float)ot.z, inv_oy = 1.f / (float)ot.y;
  for (int o = tid; o < otot; o += kBlock) {
    const int ij = fast_div(o, ot.z, inv_oz);
    const int ok = o - ij * ot.z;
    const int oi = fast_div(ij, ot.y, inv_oy);
    const int oj = ij - oi * ot.y;
    const int i = o0x + oi, j = o0y + oj, k = o0z + ok;
    if (i >= xd.x || j >= xd.y || k >= xd.z) continue;
    float acc = 0.f;
    for (int a = 0; a < T.n[0]; ++a) {
      const float wa = taps[a];
      for (int b = 0; b < T.n[1]; ++b) {
        const float wab = wa * taps[UNIRES_MAX_TAPS + b];
        const float *row = tile + ((oi * T.s[0] + a) * pty + (oj * T.s[1] + b)) * ptz + ok * T.s[2];
        for (int c = 0; c < T.n[2]; ++c) acc += row[c] * (wab * taps[2 * UNIRES_MAX_TAPS + c]);
      }
    }
    if (S.dim >= 0) {
      const int par = (S.dim == 0 ? i : (S.dim == 1 ? j : k)) & 1;
      acc *= par ? S.o : S.e;
    }
    dst[((size_t)i * xd.y + j) * xd.z + k] = acc;
  }
}

// --------------------------------------------------------------------------
// k_push_tile
// --------------------------------------------------------------------------
struct PushArgs {
  const float *src;  // grid-space values (direct) or x-space values (conv_up source)
  Dim3i xd;          // x-space dims (conv_up source)
  Dim3i gd;          // grid dims
  Taps T;
  Scaling S;
  Affine A;     // grid voxel -> output voxel coordinates
  Affine Ainv;  // output voxel -> grid voxel coordinates (bounding boxes only)
  float alpha;
  float tol;
  // fused epilogue
  const float *p;  // stencil input (nullptr: no DtD / a0 term)
  float a0, cx, cy, cz;
  float *dst;
  Dim3i dd;
  int accumulate;    // dst += instead of dst =
  double *partials;  // nullptr: no dot; else partial of sum(p * dst) per block
  const float *objb;
  int dbg;           // ablation bitmask (UNIRES_DBG env; 0 in production)
  int row_sep;       // min |d(ui,uj)|_inf for two grid rows to be splatted together
  int use_atomics;   // grid-z step too short for the plain read-add-write splat
  unsigned long long *prof;  // debug: per-phase cycle sums (UNIRES_PROF=1), else nullptr
};

// Aproned LDS accumulator: storage index 0 <-> output index (tile origin - 1), so
// every corner of every accepted source voxel lands inside the storage and the 8
// corner updates need no predicates (apron cells are dropped by the epilogue).
template <int TX, int TY, int TZ>
struct PushTile {
  static constexpr int SZ = TZ + 2, SY = TY + 2, SXd = TX + 2;
  static constexpr int N = SXd * SY * SZ;
  static constexpr int kSegs = 256;  // row segments (<= 32 long) buffered before a flush
  static constexpr int kTab = 64;    // conv_up table entries per axis (>= bbox extent)
  static constexpr int kMaxC = 4;    // max x-space voxels feeding one grid voxel per axis
};

struct RowSeg {
  short ui, uj, k0, len;
};

// conv_up tables for the part of the grid a tile touches: for grid index u (axis d)
// the x-space indices lo..lo+n-1 contribute with weights w[0..n-1] (tap * even/odd scale)
struct UpTab {
  short lo[3][64];
  short n[3][64];
  float w[3][64][4];
};

constexpr int kPushThreads = kWave;  // ONE wave per tile (see below)

// phase-ablation switches: compiled in only with -DUNIRES_ABLATE (UNIRES_DBG selects the bits)
#ifdef UNIRES_ABLATE
#define ABL(bit) ((P.dbg & (bit)) != 0)
#else
#define ABL(bit) (false)
#endif

// compiler-only memory barrier for wave-synchronous LDS code (no instruction emitted)
#define WAVE_FENCE() asm volatile("" ::: "memory")
#define PROF_T(var) const unsigned long long var = P.prof ? __builtin_readcyclecounter() : 0ull
#define PROF_ADD(slot, t0, t1) \
  if (P.prof && lane == 0) atomicAdd(P.prof + (slot), (t1) - (t0))

// LDS float atomics (ds_add_f32) retire about one lane per clock per CU on gfx950
// (measured: 8 per source voxel -> 0.9 ms per push), so the splat is done with plain
// ds_read / add / ds_write instead.  That is race-free because
//  * a tile is owned by ONE wave, whose LDS operations execute in program order;
//  * within one read-modify-write group no two lanes touch the same address:
//      - the two half-waves work on grid rows >= `row_sep` apart, far enough that their
//        2x2x2 footprints cannot meet (host-checked for the actual affine);
//      - along a row, z-adjacent lanes hand their shared plane over in registers, so
//        a group only updates ONE z plane per lane, and lanes of a segment sit in
//        different planes - except a neighbour pair that lands in the same plane
//        (|dz/dk| < 1), which is detected with a lane shuffle and replayed in a later turn;
//      - operators whose grid-z axis is far from the output z axis (|dz/dk| <= 0.76) fall
//        back to ds_add_f32 atomics (correct, slow).
// Fixed schedule -> bit-reproducible results (the reference's atomic push is not).
template <int SRC, int TX, int TY, int TZ>
__global__ void __launch_bounds__(kPushThreads)
    k_push_tile(PushArgs P, const int *__restrict__ done) {
  if (done && *done) return;
  using Tile = PushTile<TX, TY, TZ>;
  __shared__ float acc[Tile::N];
  __shared__ RowSeg rows[Tile::kSegs];
  __shared__ UpTab tab;
  const int lane = threadIdx.x;
  const Dim3i dd = P.dd, gd = P.gd;
  const float *__restrict__ src = P.src;
  const float *__restrict__ pin = P.p;
  float *__restrict__ dst = P.dst;
  const int ntx = (dd.x + TX - 1) / TX, nty = (dd.y + TY - 1) / TY, ntz = (dd.z + TZ - 1) / TZ;
  const int ntiles = ntx * nty * ntz;
  // XCD-aware persistent schedule: block b sits on XCD b%8 (observed, perf only);
  // give each XCD one contiguous run of tiles so neighbouring tiles share an L2.
  const int nxcd = 8;
  const int per_xcd = (ntiles + nxcd - 1) / nxcd;
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
  const int slots = (gridDim.x + nxcd - 1 - xcd) / nxcd;  // blocks that share this xcd id
  const int half = lane >> 5, hl = lane & 31;              // half-wave id / lane in half-wave
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  double dot = 0.0;
  for (int tl = slot; tl < per_xcd; tl += slots) {
    const int t = xcd * per_xcd + tl;
    if (t >= ntiles) break;
    const int tzi = t % ntz, tyi = (t / ntz) % nty, txi = t / (ntz * nty);
    const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
    const int ex = min(TX, dd.x - x0), ey = min(TY, dd.y - y0), ez = min(TZ, dd.z - z0);
    WAVE_FENCE();  // previous tile's epilogue done before re-zeroing
    PROF_T(t_start);
    for (int i = lane; i < Tile::N; i += kPushThreads) acc[i] = 0.f;
    // a source voxel touches the tile <=> floor(g) in [lo, hi-1] <=> g in [lo, hi)
    const float flx = (float)(x0 - 1), fly = (float)(y0 - 1), flz = (float)(z0 - 1);
    const float fhx = (float)(x0 + ex), fhy = (float)(y0 + ey), fhz = (float)(z0 + ez);
    // tiles strictly inside the volume cannot fail the in-FOV test
    const bool edge = x0 == 0 || y0 == 0 || z0 == 0 || x0 + ex >= dd.x || y0 + ey >= dd.y ||
                      z0 + ez >= dd.z;
    // ---- grid-space bounding box of everything that can touch this tile ----
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float ux, uy, uz;
      affine_point(P.Ainv, (c & 4) ? fhx : flx, (c & 2) ? fhy : fly, (c & 1) ? fhz : flz, ux, uy,
                   uz);
      lo[0] = fminf(lo[0], ux), hi[0] = fmaxf(hi[0], ux);
      lo[1] = fminf(lo[1], uy), hi[1] = fmaxf(hi[1], uy);
      lo[2] = fminf(lo[2], uz), hi[2] = fmaxf(hi[2], uz);
    }
    const int bx0 = max(0, (int)floorf(lo[0] - 0.01f)),
              bx1 = min(gd.x - 1, (int)ceilf(hi[0] + 0.01f));
    const int by0 = max(0, (int)floorf(lo[1] - 0.01f)),
              by1 = min(gd.y - 1, (int)ceilf(hi[1] + 0.01f));
    const int bz0 = max(0, (int)floorf(lo[2] - 0.01f)),
              bz1 = min(gd.z - 1, (int)ceilf(hi[2] + 0.01f));
    const int nbx = bx1 - bx0 + 1, nby = by1 - by0 + 1;
    // table-driven conv_up needs the box to fit the tables; else the general path
    // recomputes ranges per voxel (slow, only for extreme anisotropic scalings)
    const bool tab_ok = nbx <= Tile::kTab && nby <= Tile::kTab && (bz1 - bz0 + 1) <= Tile::kTab;
    if (SRC != 0 && tab_ok) {
      // conv_up tables over the bounding box (per axis: which x-space voxels feed grid voxel u)
      const int b0[3] = {bx0, by0, bz0}, b1[3] = {bx1, by1, bz1};
      const int xdv[3] = {P.xd.x, P.xd.y, P.xd.z};
      for (int e = lane; e < 3 * Tile::kTab; e += kPushThreads) {
        const int d = e / Tile::kTab, o = e % Tile::kTab;
        const int u = b0[d] + o;
        int klo = 0, n = 0;
        if (u <= b1[d]) {
          int khi;
          up_range(u, P.T.n[d], P.T.s[d], xdv[d], klo, khi);
          n = max(0, min(khi - klo + 1, Tile::kMaxC));
          if (n == 0) klo = 0;  // keeps the (zero-weighted) loads in bounds
        }
        tab.lo[d][o] = (short)klo;
        tab.n[d][o] = (short)n;
        for (int c = 0; c < Tile::kMaxC; ++c) {
          float w = 0.f;
          if (c < n) {
            w = P.T.t[d][u - P.T.s[d] * (klo + c)];
            if (P.S.dim == d) w *= ((klo + c) & 1) ? P.S.o : P.S.e;
          }
          tab.w[d][o][c] = w;
        }
      }
    }
    PROF_T(t_setup);
    PROF_ADD(0, t_start, t_setup);
    const float c0 = P.A.m[2], c1 = P.A.m[6], c2 = P.A.m[10];  // step of g along grid z
    const int nrow_cand = ABL(8) ? 0 : max(nbx, 0) * max(nby, 0);
    const int segs_per_row = (max(bz1 - bz0 + 1, 1) + 31) / 32;
    int nseg = 0;
    // candidate rows per pass: never more segments than the row list holds (long rows when the
    // grid is much finer than the output along z)
    const int chunk = max(1, min(kPushThreads, Tile::kSegs / segs_per_row));
    for (int rc0 = 0;;) {
      // ---- phase A: up to 64 candidate rows (ui,uj) per pass -> exact grid-z intervals ----
      PROF_T(t_a0);
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
              if (ta > tb) {
                const float tmp = ta;
                ta = tb;
                tb = tmp;
              }
              // clamp before the int conversion; the slack only has to cover float rounding
              // of ta/tb (membership itself is decided exactly per voxel in phase B)
              ta = fmaxf(ta, -1e6f), tb = fminf(tb, 1e6f);
              k0 = max(k0, (int)ceilf(ta - 2e-3f - 1e-5f * fabsf(ta)));
              k1 = min(k1, (int)floorf(tb + 2e-3f + 1e-5f * fabsf(tb)));
            } else if (rr[d] < lw[d] - 0.01f || rr[d] >= hg[d] + 0.01f) {
              k1 = k0 - 1;
            }
          }
        }
        // compact into the segment list, one entry per <=32-long piece, by ballot
        bool has = k1 >= k0;
        while (__any(has)) {
          const unsigned long long m = __ballot(has);
          const int pos = nseg + __popcll(m & lt_mask);
          if (has && pos < Tile::kSegs)
            rows[pos] = RowSeg{(short)ui, (short)uj, (short)k0, (short)min(32, k1 - k0 + 1)};
          nseg += __popcll(m);
          k0 += 32;
          has = has && k1 >= k0;
        }
        rc0 += chunk;
      }
      PROF_T(t_a1);
      PROF_ADD(1, t_a0, t_a1);
      const bool last = rc0 >= nrow_cand;
      if (!last && nseg + chunk * segs_per_row <= Tile::kSegs) continue;
      WAVE_FENCE();  // segment list / tables / zeroed acc written before being read
      // ---- phase B: the two half-waves take segments p and p + npair, lanes along grid z.
      // Batches of kU segment pairs, software-pipelined: the global loads of batch b+1 are
      // issued before the LDS updates of batch b.
      const int nr = ABL(1) ? 0 : min(nseg, Tile::kSegs);
      const int npair = (nr + 1) / 2;
      constexpr int kU = 4;
      struct Batch {
        float s0[kU], s1[kU], w0[kU], w1[kU];
        int ui[kU], uj[kU], uk[kU];
        bool act[kU], solo[kU];
      };
      auto load_batch = [&](int p0, Batch &B) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int p = p0 + u;
          const int ia = max(min(p, nr - 1), 0), ib = max(min(p + npair, nr - 1), 0);
          const RowSeg Ra = rows[ia], Rb = rows[ib];
          // rows too close to be splatted in the same instruction -> B waits for a solo turn
          B.solo[u] = p + npair < nr && max(abs(Ra.ui - Rb.ui), abs(Ra.uj - Rb.uj)) < P.row_sep;
          const RowSeg R = half ? Rb : Ra;
          B.act[u] = p < npair && (half ? p + npair < nr : true) && hl < R.len;
          B.ui[u] = R.ui, B.uj[u] = R.uj, B.uk[u] = min(R.k0 + hl, gd.z - 1);
          const int ui = B.ui[u], uj = B.uj[u], uk = B.uk[u];
          if (SRC == 0) {
            B.s0[u] = src[((size_t)ui * gd.y + uj) * gd.z + uk];
            B.s1[u] = 0.f, B.w0[u] = 1.f, B.w1[u] = 0.f;
          } else if (SRC == 1 && tab_ok) {
            // thick-slice case: at most 2 x-space voxels along ONE axis feed a grid voxel;
            // both loads are unconditional (a zero weight covers the absent second one)
            const int ox = ui - bx0, oy = uj - by0, oz = min(uk - bz0, Tile::kTab - 1);
            const int ix = tab.lo[0][ox], iy = tab.lo[1][oy], iz = tab.lo[2][oz];
            const bool dx = tab.n[0][ox] > 1, dy = tab.n[1][oy] > 1, dz = tab.n[2][oz] > 1;
            const size_t base = ((size_t)ix * P.xd.y + iy) * P.xd.z + iz;
            const size_t step = dx ? (size_t)P.xd.y * P.xd.z : (dy ? (size_t)P.xd.z : (size_t)dz);
            B.s0[u] = src[base], B.s1[u] = src[base + step];
            const float wx = tab.w[0][ox][0], wy = tab.w[1][oy][0], wz = tab.w[2][oz][0];
            B.w0[u] = wx * wy * wz;
            B.w1[u] = dx ? tab.w[0][ox][1] * wy * wz
                         : (dy ? wx * tab.w[1][oy][1] * wz : wx * wy * tab.w[2][oz][1]);
          } else if (tab_ok) {
            const int ox = ui - bx0, oy = uj - by0, oz = min(uk - bz0, Tile::kTab - 1);
            float v = 0.f;
            for (int a = 0; a < tab.n[0][ox]; ++a)
              for (int b = 0; b < tab.n[1][oy]; ++b) {
                const float wab = tab.w[0][ox][a] * tab.w[1][oy][b];
                const float *row =
                    src + ((size_t)(tab.lo[0][ox] + a) * P.xd.y + (tab.lo[1][oy] + b)) * P.xd.z +
                    tab.lo[2][oz];
                for (int c = 0; c < tab.n[2][oz]; ++c) v += row[c] * (wab * tab.w[2][oz][c]);
              }
            B.s0[u] = v, B.s1[u] = 0.f, B.w0[u] = 1.f, B.w1[u] = 0.f;
          } else {
            B.s0[u] = conv_up_sample(src, P.xd, P.T, P.S, ui, uj, uk);
            B.s1[u] = 0.f, B.w0[u] = 1.f, B.w1[u] = 0.f;
          }
        }
      };
      auto splat_batch = [&](const Batch &B) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          float gx, gy, gz;
          affine_point(P.A, (float)B.ui[u], (float)B.uj[u], (float)B.uk[u], gx, gy, gz);
          bool ok = B.act[u] && gx >= flx && gx < fhx && gy >= fly && gy < fhy && gz >= flz && gz < fhz;
          if (edge) ok = ok && in_fov(gx, gy, gz, dd, P.tol);
          const float v = P.alpha * (B.s0[u] * B.w0[u] + B.s1[u] * B.w1[u]);
          ok = ok && v != 0.f && !ABL(2);
          const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
          const int lx = (int)fx - (x0 - 1), ly = (int)fy - (y0 - 1), lz = (int)fz - (z0 - 1);
          const float wx1 = gx - fx, wy1 = gy - fy, wz1 = gz - fz;
          const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
          const int cell = (lx * Tile::SY + ly) * Tile::SZ + lz;
          // a lane whose z plane equals its predecessor's could overlap it inside one
          // update group (same or xy-adjacent cell): it waits for the next turn.  With
          // |dz/dk| > 0.76 (host-checked) no two such lanes of a segment share a plane.
          const int prev_lz = __shfl_up(ok ? lz : -1 - lane, 1, kWave);
          const bool dup = ok && hl > 0 && lz == prev_lz;
          constexpr int SX = Tile::SY * Tile::SZ, SY = Tile::SZ;
          const float a00 = v * (wx0 * wy0), a01 = v * (wx0 * wy1), a10 = v * (wx1 * wy0),
                      a11 = v * (wx1 * wy1);
          // Turn schedule: lanes that could touch the footprint of another lane of the same
          // instruction wait for a later turn: (row B of a too-close pair) x (replayed
          // same-plane neighbour).  Turn 0 is the only one that normally runs.
          const int myturn = ((B.solo[u] && half) ? 2 : 0) + (dup ? 1 : 0);
#pragma unroll
          for (int turn = 0; turn < 4; ++turn) {
            if (!__any(ok && myturn == turn)) continue;  // wave-uniform
            // WAVE_FENCE pins the order of the LDS read-modify-write groups for the compiler;
            // the hardware executes one wave's LDS instructions in order.
            WAVE_FENCE();
            const bool on = ok && myturn == turn;
            const int mycell = on ? cell : -1 - lane;
            float l00 = a00 * wz0, l01 = a01 * wz0, l10 = a10 * wz0, l11 = a11 * wz0;  // z  plane
            const float u00 = a00 * wz1, u01 = a01 * wz1, u10 = a10 * wz1, u11 = a11 * wz1;  // z+1
            // z-adjacent lanes of a segment: the upper plane of lane l IS the lower plane of
            // lane l+1 -> hand it over in registers (half the LDS traffic, and the remaining
            // updates of one group never share an address)
            const int below = __shfl_up(mycell, 1, kWave), above = __shfl_down(mycell, 1, kWave);
            const float p00 = __shfl_up(u00, 1, kWave), p01 = __shfl_up(u01, 1, kWave),
                        p10 = __shfl_up(u10, 1, kWave), p11 = __shfl_up(u11, 1, kWave);
            const bool recv = on && hl > 0 && below + 1 == mycell;
            const bool sent = on && hl < 31 && above == mycell + 1;
            if (recv) l00 += p00, l01 += p01, l10 += p10, l11 += p11;
            float *q = acc + (on ? cell : 0);
            if (P.use_atomics) {  // grid step too short for the neighbour-only argument
              if (on) {
                atomicAdd(q, l00), atomicAdd(q + SY, l01), atomicAdd(q + SX, l10),
                    atomicAdd(q + SX + SY, l11);
                if (!sent)
                  atomicAdd(q + 1, u00), atomicAdd(q + SY + 1, u01), atomicAdd(q + SX + 1, u10),
                      atomicAdd(q + SX + SY + 1, u11);
              }
              continue;
            }
            if (on) {  // group 1: z plane (4 distinct cells per lane, no overlap across lanes)
              const float o00 = q[0], o01 = q[SY], o10 = q[SX], o11 = q[SX + SY];
              q[0] = o00 + l00, q[SY] = o01 + l01, q[SX] = o10 + l10, q[SX + SY] = o11 + l11;
            }
            WAVE_FENCE();
            if (on && !sent) {  // group 2: z+1 plane of lanes with nobody above them
              const float o00 = q[1], o01 = q[SY + 1], o10 = q[SX + 1], o11 = q[SX + SY + 1];
              q[1] = o00 + u00, q[SY + 1] = o01 + u01, q[SX + 1] = o10 + u10,
              q[SX + SY + 1] = o11 + u11;
            }
            WAVE_FENCE();
          }
        }
      };
      {
        PROF_T(t_b0);
        Batch B0, B1;
        if (npair > 0) load_batch(0, B0);
        for (int p0 = 0; p0 < npair; p0 += 2 * kU) {
          PROF_T(t_l0);
          if (p0 + kU < npair) load_batch(p0 + kU, B1);
          PROF_T(t_l1);
          splat_batch(B0);
          PROF_T(t_l2);
          PROF_ADD(2, t_l0, t_l1);
          PROF_ADD(3, t_l1, t_l2);
          if (p0 + kU < npair) {
            if (p0 + 2 * kU < npair) load_batch(p0 + 2 * kU, B0);
            splat_batch(B1);
          }
        }
        PROF_T(t_b1);
        PROF_ADD(4, t_b0, t_b1);
        if (P.prof && lane == 0) atomicAdd(P.prof + 7, (unsigned long long)nr);
      }
      nseg = 0;
      if (last) break;
    }
    WAVE_FENCE();
    PROF_T(t_e0);
    // ---- epilogue: q = [q +] acc + a0 p + c DtD p ; dot += p*q --------------
    // (dst never aliases p: the stencil loads of several outputs are hoisted together)
    // one output row (lx,ly) per half-wave and pass: contiguous 4*TZ-byte stores, no div/mod
    static_assert(TZ <= 32, "epilogue maps one row to a half-wave");
#pragma unroll 4
    for (int r = half; r < TX * TY; r += 2) {
      const int lx = r / TY, ly = r % TY, lz = hl;
      if (lx >= ex || ly >= ey || lz >= ez) continue;
      const int i = x0 + lx, j = y0 + ly, k = z0 + lz;
      const size_t idx = ((size_t)i * dd.y + j) * dd.z + k;
      float q = acc[((lx + 1) * Tile::SY + ly + 1) * Tile::SZ + lz + 1];
      float pc = 0.f;
      if (pin && !ABL(4)) {  // dbg 4: no stencil
        const float st = dtd_at(pin, idx, i, j, k, dd, P.cx, P.cy, P.cz, pc);
        q += P.a0 * pc + st;
      }
      if (P.accumulate) q += dst[idx];
      matvec_emit(dst, idx, q, pc, P.objb, P.partials != nullptr, dot);
    }
    PROF_T(t_e1);
    PROF_ADD(5, t_e0, t_e1);
    PROF_ADD(6, t_start, t_e1);
  }
  if (P.partials) {
    const double tot = wave_sum(dot);
    if (lane == 0) P.partials[blockIdx.x] = tot;
  }
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
static void pick_out_tile(const Taps &T, const Dim3i &xd, Dim3i &ot, size_t &lds_bytes) {
  // z: as many outputs as keep the pulled row within two 64-lane passes (a dirac axis:
  // one pass); x,y: 4x4 rows, halved until the pulled tile fits ~24 KB of LDS.
  const int xdv[3] = {xd.x, xd.y, xd.z};
  static const int ex = getenv("UNIRES_PC_TX") ? atoi(getenv("UNIRES_PC_TX")) : 4;
  static const int ey = getenv("UNIRES_PC_TY") ? atoi(getenv("UNIRES_PC_TY")) : 4;
  static const int ez = getenv("UNIRES_PC_PZ") ? atoi(getenv("UNIRES_PC_PZ")) : 128;
  int t[3] = {ex, ey, 1};
  t[2] = (T.s[2] == 1 && T.n[2] == 1) ? 64 : std::max(1, (ez - T.n[2]) / T.s[2] + 1);
  for (int d = 0; d < 3; ++d)
    if (t[d] > xdv[d]) t[d] = xdv[d];
  auto pt = [&](int d) { return (size_t)(t[d] - 1) * T.s[d] + T.n[d]; };
  while (pt(0) * pt(1) * pt(2) * 4 > 24 * 1024) {
    int big = pt(0) >= pt(1) ? 0 : 1;
    if (t[big] == 1) big = 1 - big;
    if (t[big] == 1) big = 2;
    if (t[big] == 1) break;
    t[big] = (t[big] + 1) / 2;
  }
  ot = Dim3i{t[0], t[1], t[2]};
  lds_bytes = (pt(0) * pt(1) * pt(2) + 3 * UNIRES_MAX_TAPS) * sizeof(float);
}

int launch_pull_conv(const float *src, Dim3i sd, const Affine &A, const Taps &T, const Scaling &S,
                     float *dst, Dim3i xd, Dim3i gd, float tol, const int *done, hipStream_t st) {
  Dim3i ot;
  size_t lds;
  pick_out_tile(T, xd, ot, lds);
  if (lds > 64 * 1024) return 1;  // caller falls back to the unfused path
  const dim3 grid((xd.z + ot.z - 1) / ot.z, (xd.y + ot.y - 1) / ot.y, (xd.x + ot.x - 1) / ot.x);
  hipLaunchKernelGGL(k_pull_conv, grid, dim3(kBlock), lds, st, src, sd, A, T, S, dst, xd, gd, ot,
                     tol, done);
  return 0;
}

constexpr int kTX = 8, kTY = 8, kTZ = 30;

int push_tile_blocks(Dim3i dd) {
  const long long nt = (long long)((dd.x + kTX - 1) / kTX) * ((dd.y + kTY - 1) / kTY) *
                       ((dd.z + kTZ - 1) / kTZ);
  return (int)(nt < kMaxPartials ? nt : kMaxPartials);
}

// Geometry of the race-free splat (see k_push_tile).
//  * use_atomics: lanes two apart along a grid row must never share a cell, i.e. the
//    image of two grid-z steps must be longer than a cell diagonal;
//  * row_sep: smallest n such that rows (ui,uj) and (ui+di,uj+dj), max(|di|,|dj|) >= n,
//    have disjoint 2x2x2 footprints for EVERY pair of lanes: for every offset tau along
//    the rows some coordinate differs by >= 2 (then its floor differs by >= 2).
void splat_safety(const Affine &A, int &row_sep, int &use_atomics) {
  const double c[3] = {A.m[2], A.m[6], A.m[10]};  // image of one grid-z step
  const double cn = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  // z planes must advance by >= 1 every two lanes (so replayed lanes, which are >= 2
  // apart, never share a plane) and never by more than one cell diagonal per lane
  use_atomics = !(fabs(c[2]) > 0.76 && cn < 1.9);
  row_sep = 1 << 14;  // "never pair rows"
  if (use_atomics) return;
  auto disjoint = [&](int di, int dj) {
    double v[3];
    for (int r = 0; r < 3; ++r) v[r] = di * (double)A.m[4 * r] + dj * (double)A.m[4 * r + 1];
    // min over tau in [-48, 48] of max_r |v_r + tau c_r|: convex and piecewise linear, so the minimum sits at an
    // end, at a zero of one line or where two of the six lines +-(v_r + tau c_r) cross (r1 - r3 sampled tau in
    // steps of 0.02: 1.2 ms per operator, most of what unires_plan_set_repeat cost the host)
    auto f = [&](double tau) {
      double m = 0;
      for (int r = 0; r < 3; ++r) m = std::max(m, fabs(v[r] + tau * c[r]));
      return m;
    };
    double best = std::min(f(-48.0), f(48.0));
    auto consider = [&](double tau) {
      if (tau > -48.0 && tau < 48.0) best = std::min(best, f(tau));
    };
    for (int r = 0; r < 3; ++r) {
      if (c[r] != 0.0) consider(-v[r] / c[r]);
      for (int q = r + 1; q < 3; ++q) {
        if (c[r] != c[q]) consider((v[q] - v[r]) / (c[r] - c[q]));
        if (c[r] != -c[q]) consider((-v[q] - v[r]) / (c[r] + c[q]));
      }
    }
    return !(best < 2.0 + 0.03);  // (0.03: the margin the sampled form carried for its tau step, kept)
  };
  for (int n = 1; n <= 6; ++n) {
    bool ok = true;
    for (int di = -n - 3; di <= n + 3 && ok; ++di)
      for (int dj = -n - 3; dj <= n + 3 && ok; ++dj)
        if (std::max(abs(di), abs(dj)) >= n && !disjoint(di, dj)) ok = false;
    if (ok) {
      row_sep = n;
      return;
    }
  }
}

// 0: direct source; 1: conv_up where <= 2 x-space voxels along one axis feed a grid
// voxel (every thick-slice rect profile); 2: general separable conv_up
static int push_src_kind(const PushSrc &src) {
  if (!src.convup) return 0;
  int multi = 0, worst = 1;
  for (int d = 0; d < 3; ++d) {
    const int c = (src.T.n[d] + src.T.s[d] - 1) / src.T.s[d];
    if (c > 1) ++multi;
    if (c > worst) worst = c;
  }
  return (multi <= 1 && worst <= 2) ? 1 : 2;
}

int launch_push_tile(const PushSrc &src, const Affine &A, const Affine &Ainv,
                     const SplatSafety &safe, float alpha, float tol, const PushEpilogue &ep,
                     float *dst, Dim3i dd, const int *done, hipStream_t st) {
  PushArgs P;
  P.src = src.data;
  P.xd = src.xd;
  P.gd = src.gd;
  P.T = src.T;
  P.S = src.S;
  P.A = A;
  P.Ainv = Ainv;
  P.alpha = alpha;
  P.tol = tol;
  P.p = ep.p;
  P.a0 = ep.a0;
  P.cx = ep.cx;
  P.cy = ep.cy;
  P.cz = ep.cz;
  P.dst = dst;
  P.dd = dd;
  P.accumulate = ep.accumulate;
  P.partials = ep.partials;
  P.objb = ep.objb;
  static const int dbg = getenv("UNIRES_DBG") ? atoi(getenv("UNIRES_DBG")) : 0;
  P.dbg = dbg;
  P.row_sep = safe.row_sep;
  P.use_atomics = safe.use_atomics;
  static unsigned long long *prof = nullptr;
  static const bool want_prof = getenv("UNIRES_PROF") != nullptr;
  if (want_prof && !prof) (void)hipMalloc((void **)&prof, 8 * sizeof(unsigned long long));
  if (want_prof) (void)hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), st);
  P.prof = want_prof ? prof : nullptr;
  const int kind = push_src_kind(src);
  if (kind == 2)
    for (int d = 0; d < 3; ++d)
      if ((src.T.n[d] + src.T.s[d] - 1) / src.T.s[d] > PushTile<kTX, kTY, kTZ>::kMaxC) return 1;
  const dim3 grid(push_tile_blocks(dd));
  if (kind == 0)
    hipLaunchKernelGGL((k_push_tile<0, kTX, kTY, kTZ>), grid, dim3(kPushThreads), 0, st, P, done);
  else if (kind == 1)
    hipLaunchKernelGGL((k_push_tile<1, kTX, kTY, kTZ>), grid, dim3(kPushThreads), 0, st, P, done);
  else
    hipLaunchKernelGGL((k_push_tile<2, kTX, kTY, kTZ>), grid, dim3(kPushThreads), 0, st, P, done);
  if (want_prof) {
    unsigned long long h[8];
    (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr,
            "[push prof kind %d] Mcycles: setup %.1f phaseA %.1f loadB %.1f splatB %.1f phaseB %.1f "
            "epilogue %.1f total %.1f | segments %llu waves %d\n",
            kind, h[0] * 1e-6, h[1] * 1e-6, h[2] * 1e-6, h[3] * 1e-6, h[4] * 1e-6, h[5] * 1e-6,
            h[6] * 1e-6, h[7], (int)grid.x);
  }
  return 0;
}

}  // namespace unires
