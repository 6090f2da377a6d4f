// pull2.hip - k_pull_conv2: xs = S . conv_down_z . pull_M (p) with the trilinear gather served
// from LDS instead of the texture addresser.
//
// What the hardware charges (tools/mb_pull.hip, MI355X): a wave-level global gather costs the
// texture addresser ~16 clocks however well its 64 addresses coalesce, so the 4 corner-pair loads
// of a trilinear sample bound a 256^3 pull at ~50 us; the same 8 values read from LDS cost the
// CU ~18 clocks per 64 samples.  So a workgroup stages the part of p its grid rows can touch with
// coalesced 16-byte loads and samples from LDS.
//
//   workgroup = TI x TJ grid rows x one 64-lane chunk of grid z (lanes along grid z)
//   window    = for every group of 4 z planes of p, a W x H patch of (x, y) columns whose origin
//               follows the rows as they drift (a SHEARED box: an axis-aligned bounding box of
//               rows tilted by 0.1 rad would be 3-4x larger).  Origins come from the closed-form
//               extremes of  x = px_i i + px_j j + s_x gz + c_x  over the workgroup's rows and the
//               group's planes.  They depend only on the operator, so a one-off kernel tabulates
//               them per workgroup (PullPlan, built with the plan) and the hot kernel starts with
//               three scalar loads instead of a page of uniform float arithmetic.
//   staging   = buffer_load_dwordx4 ... lds (LDS-DMA): the window is laid out so that item n of
//               the load order is 16 bytes n of the LDS window - no VGPR round trip, no ds_write.
//               The byte offset of an item is  boff[plane group] + coff[item]:  boff (window origin of
//               the group, per workgroup) comes with the plan's record, coff and the item -> group map
//               from a per-operator table (r3: the in-kernel form - two divisions by constants, a
//               64-bit multiply-add and four range tests per item, five quarter-rate integer
//               multiplies among them - was a quarter of the kernel's VALU time).  Plane groups that
//               lie outside the volume along z carry an out-of-range origin, so every workgroup takes
//               the LDS-DMA path (the volume's last, partial group is zero-filled after the barrier).
//   conv_down = the slice profile runs on the pulled values of 8 rows at a time through a small
//               padded LDS scratch; chunks are cut at multiples of the stride so every x-space
//               voxel is produced by exactly one workgroup.
// The coordinate arithmetic is affine_row / affine_along, bit for bit the splat's, so the pair
// stays an exact adjoint.
#include <math.h>

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "pull2.hpp"

namespace unires {

// phase-ablation switches: compiled in only with -DUNIRES_ABLATE (then UNIRES_P2_DBG selects bits: 1 no staging,
// 2 no sampling, 4 no conv / store); product builds carry none of it (r6: the run-time test cost every sample
// two scalar and two vector instructions and a branch)
#ifdef UNIRES_ABLATE
#define P2_ABL(bit) ((P.dbg & (bit)) != 0)
#else
#define P2_ABL(bit) (false)
#endif

constexpr int kP2SZ = 72, kP2SZ4 = kP2SZ / 4;  // z planes of a window (64 + drift + 2 + alignment)
constexpr int kP2Items = 10;                    // 16-byte window pieces staged per thread (at most)
#ifndef UNIRES_P2_TI
#define UNIRES_P2_TI 4
#endif
#ifndef UNIRES_P2_TJ
#define UNIRES_P2_TJ 8
#endif
constexpr int kP2TI = UNIRES_P2_TI, kP2TJ = UNIRES_P2_TJ;  // grid rows per workgroup (profile along z only)
constexpr int kP2Rows = 64;                     // ... and at most, in any layout
constexpr int kP2Scr = kWave + 9;               // floats per row of the z-only conv scratch (64 lanes + <= 8 extras, odd)
constexpr int kP2MaxExt = 8;                    // extras per row at most (8 rows per wave x 8 = one pass of 64 lanes)
#ifndef UNIRES_P2_BAND
#define UNIRES_P2_BAND 4
#endif
constexpr int kP2Band = UNIRES_P2_BAND;         // block rows walked together (see the kernel's block mapping)

struct P2Geom {
  Affine A;
  Dim3i sd, gd;
  int sk, m;         // stride along grid z, conv windows (x-space voxels) per chunk
  // grid points a chunk samples: (m - 1) sk + taps.  <= 64: one per lane.  A z-only profile may take ONE
  // window more than 64 lanes hold where that saves a chunk (config 3: 42 slices = 11 + 11 + 11 + 9
  // instead of 10 + 10 + 10 + 10 + 2): the span - 64 <= 8 points beyond the lanes ("extras", 3 for 7 taps
  // at stride 6) of a wave's 8 rows are sampled together in one more pass, lane = (row, extra).
  int span;
  int nbj, nbc;      // workgroups along j and chunks along z (block = (bi * nbj + bj) * nbc + bc)
  // rows of a workgroup: pi x pj grid rows = what oi x oj x-space rows need (stride si / sj,
  // ni / nj taps along x / y; 8 x 8 rows, strides 1, when the profile runs along z only)
  int pi, pj, oi, oj, si, sj, ni, nj;
  // sheared-plane form of the affine: x = pxi i + pxj j + sx gz + cx  (same for y)
  float pxi, pxj, sx, cx, pyi, pyj, sy, cy;
};

// per-workgroup record: {Z0, plane groups in use, all-inside flag, flags} then the 18 window origins,
// then the 18 byte offsets of the origins (kP2OobBase: the group is not staged)
//   flags: 1 all samples outside the field of view, 2 every staged column inside the volume in x / y,
//          4 the volume's last plane group is partial (planes >= sd.z to be zero-filled)
constexpr int kP2Rec = 4 + 3 * kP2SZ4 + 2;
constexpr unsigned kP2OobBase = 0x80000000u;  // + any in-window offset (< 2^31) stays out of range

__global__ void k_pull2_plan(P2Geom G, int nblk, float tol, int W, int H, int *__restrict__ rec,
                             unsigned char *__restrict__ cls) {
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;
  if (blk >= nblk) return;
  constexpr int SZ4 = kP2SZ4;
  const int bc = blk % G.nbc, bj = (blk / G.nbc) % G.nbj, bi = blk / (G.nbc * G.nbj);
  const int i0 = bi * G.oi * G.si, j0 = bj * G.oj * G.sj;
  const int i1 = min(i0 + G.pi, G.gd.x) - 1, j1 = min(j0 + G.pj, G.gd.y) - 1;
  const int k0 = bc * G.m * G.sk;
  const int npts = min(max(kWave, G.span), G.gd.z - k0);
  // z extent of the workgroup's samples: 8 vertices of the (i, j, k) box
  float zmin = 1e30f, zmax = -1e30f;
  bool inside = true;  // every sample has all 8 corners inside the volume (no FOV mask needed)
  // ... or every sample is outside the field of view: the 8 vertices of the box of grid points lie
  // beyond one face of the volume (an affine image of a box is convex), with a margin well above the
  // rounding of the coordinates
  int below[3] = {0, 0, 0}, above[3] = {0, 0, 0};
  const float lim[3] = {(float)(G.sd.x - 1), (float)(G.sd.y - 1), (float)(G.sd.z - 1)};
  for (int c = 0; c < 8; ++c) {
    float gx, gy, gz;
    affine_point(G.A, (float)((c & 4) ? i1 : i0), (float)((c & 2) ? j1 : j0),
                 (float)((c & 1) ? k0 + npts - 1 : k0), gx, gy, gz);
    zmin = fminf(zmin, gz), zmax = fmaxf(zmax, gz);
    inside = inside && gx >= 0.01f && gx <= (float)(G.sd.x - 1) - 0.01f && gy >= 0.01f &&
             gy <= (float)(G.sd.y - 1) - 0.01f && gz >= 0.01f && gz <= (float)(G.sd.z - 1) - 0.01f;
    const float gv[3] = {gx, gy, gz};
    for (int d = 0; d < 3; ++d) below[d] += gv[d] < -tol - 0.01f, above[d] += gv[d] > lim[d] + tol + 0.01f;
  }
  bool empty = false;
  for (int d = 0; d < 3; ++d) empty = empty || below[d] == 8 || above[d] == 8;
  const int Z0 = 4 * (int)floorf(floorf(zmin - 0.01f) * 0.25f);
  const int ngrp = min(SZ4, ((int)floorf(zmax + 0.01f) + 1 - Z0) / 4 + 1);
  int *r = rec + (size_t)blk * kP2Rec;
  const float fi0 = (float)i0, fi1 = (float)i1, fj0 = (float)j0, fj1 = (float)j1;
  bool xyin = true, partial = false;
  for (int g = 0; g < SZ4; ++g) {
    // samples whose lower or upper corner plane falls in this group: gz in [Zg - 1, Zg + 4)
    const float za = (float)(Z0 + 4 * g - 1), zb = (float)(Z0 + 4 * g + 4);
    const float xlo = G.cx + fminf(G.pxi * fi0, G.pxi * fi1) + fminf(G.pxj * fj0, G.pxj * fj1) +
                      fminf(G.sx * za, G.sx * zb);
    const float ylo = G.cy + fminf(G.pyi * fi0, G.pyi * fi1) + fminf(G.pyj * fj0, G.pyj * fj1) +
                      fminf(G.sy * za, G.sy * zb);
    const int ox = (int)floorf(xlo - 0.02f), oy = (int)floorf(ylo - 0.02f);
    r[4 + 2 * g] = ox;
    r[5 + 2 * g] = oy;
    // byte offset of the group's origin column (wraps for negative origins: the sum with an in-volume
    // column's offset is exact mod 2^32); groups beyond ngrp or wholly outside the volume along z are
    // not staged (Z0 is a multiple of 4: a group below z = 0 is wholly below)
    const int z = Z0 + 4 * g;
    const bool staged = g < ngrp && z >= 0 && z < G.sd.z;
    unsigned boff = kP2OobBase;
    if (staged) {
      boff = 4u * (unsigned)((ox * G.sd.y + oy) * G.sd.z + z);
      xyin = xyin && ox >= 0 && ox + W <= G.sd.x && oy >= 0 && oy + H <= G.sd.y;
      partial = partial || z + 3 >= G.sd.z;
    }
    r[4 + 2 * SZ4 + g] = (int)boff;
  }
  r[0] = Z0, r[1] = ngrp, r[2] = inside ? 1 : 0;
  r[3] = (empty ? 1 : 0) | (xyin ? 2 : 0) | (partial ? 4 : 0);
  r[4 + 3 * SZ4] = (int)kP2OobBase;  // slot the padding items of the table point at
  r[5 + 3 * SZ4] = 0;
  // what the workgroup will cost (pull2_build deals the workgroups over the XCDs by it): 0 all outside the field of
  // view (it writes zeros), 1 plain, 2 with the FOV mask and / or range tests on its staging items
  cls[blk] = empty ? 0 : ((inside && xyin) ? 1 : 2);
}

struct P2Args {
  const float *src;
  const int *rec;
  const int4 *wtab;  // dispatch index -> {record, bi, bj, bc} (pull2_build: equal COST per XCD); nullptr: by formula
  const int2 *itab;  // per staging item: {4 * plane group, byte offset inside the window} ...
  const int *itab2;  // ... and cxl | cyl << 16 (column inside the window; read by workgroups at the volume's x / y faces)
  float inv_m;       // 1 / G.m
  P2Geom G;
  float kz[UNIRES_MAX_TAPS], kx[UNIRES_MAX_TAPS], ky[UNIRES_MAX_TAPS];
  int nk;            // taps along grid z (1 with sk 1: no conv)
  float se, so;      // even / odd x-space slice scaling (1, 1: none) ...
  int sdim;          // ... along this x-space axis (-1: none)
  float *dst;
  Dim3i xd;
  float tol;
  int W;             // window extent along x (cells); along y it is the template parameter H
  int band;          // block rows walked together (see the kernel's block mapping)
  unsigned long long *prof;  // -DUNIRES_P2_PROF builds: per-workgroup timeline (100 MHz ticks)
  int dbg;           // UNIRES_P2_DBG ablation bits (read only by -DUNIRES_ABLATE builds): 1 no staging, 2 no sampling, 4 no conv / store
};

// NK, SK > 0: compile-time slice profile (7 taps stride 6 is the 6 mm / 1 mm case); 0: run-time
// GEN: profile along x and / or y as well (rows laid out pi x pj, separable conv over a
// workgroup-wide scratch); otherwise 8 x 8 rows and a wave-private conv along z.
template <int H, int NK, int SK, bool GEN>
__global__ void __launch_bounds__(kBlock) k_pull_conv2(P2Args P, const int *__restrict__ done) {
  if (done && *done) return;
#ifdef UNIRES_P2_PROF
  unsigned long long *pw = P.prof ? P.prof + (size_t)blockIdx.x * 8 : nullptr;
  if (pw && threadIdx.x == 0) pw[0] = wall_clock64(), pw[4] = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);
#endif
  constexpr int SZ = kP2SZ, SZ4 = kP2SZ4, NW = kBlock / kWave, TI = kP2TI, TJ = kP2TJ;
  constexpr int ROWS = GEN ? kP2Rows : TI * TJ, RPW = ROWS / NW, SCR = GEN ? kWave + 1 : kP2Scr;
  constexpr int HALF = RPW % 8 == 0 ? 8 : (RPW % 6 == 0 ? 6 : 4);
  static_assert(TI * TJ <= kP2Rows && ROWS % NW == 0 && RPW % HALF == 0, "rows per wave");
  // the extended chunk's extra grid points of a wave's rows are sampled in ONE pass, lane = (row, extra)
  static_assert(GEN || RPW * kP2MaxExt <= kWave, "extras of a wave's rows must fit one pass of 64 lanes");
  extern __shared__ __align__(16) float win[];  // W * H columns x SZ planes
  __shared__ int tab[SZ4 + 2];                  // per plane group: -(ox * H + oy) * SZ, in floats
  __shared__ int2 org[SZ4];
  __shared__ unsigned boff[SZ4 + 1];            // per plane group: byte offset of the window origin; [SZ4]: out of range
  __shared__ unsigned char rowi[kP2Rows], rowj[kP2Rows];  // GEN: row -> (ri, rj)
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const P2Geom &G = P.G;
  // (each XCD walks one contiguous run of workgroups: neighbours share window columns in its L2)
  // Inside the run the (bi, bj) pairs are walked in BANDS of kP2Band values of bi (bc fastest, then bi
  // inside the band, then bj): a workgroup shares window columns with its neighbours in i and in j
  // (each column of p is wanted by ~2.2 workgroups), and the row-major walk puts the i-neighbour
  // nbj * nbc workgroups = 7 MB of windows away at 384^3 - beyond the XCD's 4 MB L2: the pull fetched
  // 508 MB for a 226 MB volume there (profiles/r03_traffic_other_configs.jsonl), and HBM is what bounds
  // it at that size.  (Where one block row's windows DO fit - config 3: 3.2 MB - the plain order is kept:
  // bands cut by the XCD runs' ends fetched 79 MB instead of 72 there.)
  // (r6) With the plan's dispatch table the workgroup reads {record, bi, bj, bc} in one 16-byte scalar load: the walk's
  // arithmetic below is five integer divisions by run-time values - ~150 dependent scalar instructions between a
  // workgroup's first cycle and the address of its record, at the head of 10 240 workgroups that live ~7 us each.
  int bi, bj, bc, blk;
  if (P.wtab) {
    typedef int i4v __attribute__((ext_vector_type(4)));
    const i4v e = ((const __attribute__((address_space(4))) i4v *)P.wtab)[blockIdx.x];
    blk = e.x, bi = e.y, bj = e.z, bc = e.w;
  } else {
    const int blk0 = xcd_chunked_block((int)blockIdx.x, (int)gridDim.x);
    const int pair = blk0 / G.nbc;
    bc = blk0 % G.nbc;
    const int nbi = (int)gridDim.x / (G.nbc * G.nbj);
    const int bandh = P.band;  // kP2Band, or nbi (= plain row-major order) where a block row's windows fit the L2
    const int band = pair / (bandh * G.nbj), rem = pair - band * (bandh * G.nbj);
    const int bh = min(bandh, nbi - band * bandh);  // height of this band (the last one may be short)
    bj = rem / bh, bi = band * bandh + (rem - bj * bh);
    blk = (bi * G.nbj + bj) * G.nbc + bc;  // index of the plan's record
  }
  const int i0 = GEN ? bi * G.oi * G.si : bi * TI, j0 = GEN ? bj * G.oj * G.sj : bj * TJ;
  const int nrows = GEN ? G.pi * G.pj : ROWS;
  const int kk0 = bc * G.m, k0 = kk0 * G.sk;
  const int npts = min(kWave, G.gd.z - k0);  // grid points of this chunk along z
  const Dim3i sd = G.sd;
  const int *rec = P.rec + (size_t)blk * kP2Rec;
  const int Z0 = rec[0];
  const bool inside = rec[2] != 0;
  const int flags = rec[3];
  // this thread's staging items: issued now, they travel while the record is read and the
  // workgroup meets (the table is padded to whole waves; padding items point at the record's
  // out-of-range slot)
  const int nitem = P.W * H * SZ4;
  int2 ie[kP2Items];
#pragma unroll
  for (int n = 0; n < kP2Items; ++n) {
    const int base = n * kBlock + wave * kWave;
    ie[n] = make_int2(4 * SZ4, 0);
    if (base < nitem) ie[n] = P.itab[base + lane];
  }
  if (flags & 1) {
    // every sample of this workgroup lies outside the field of view (the part of the observation's
    // grid that sticks out of the volume: ~10 % of config 3's workgroups): its outputs are zeros
    const int xdy = P.xd.y, xdz = P.xd.z;
    if (GEN) {
      const int ojm = G.oj * G.m, nitem = G.oi * ojm;
      for (int it = tid; it < nitem; it += kBlock) {
        const int a = it / ojm, rem = it - a * ojm, b = rem / G.m, c = rem - b * G.m;
        const int io = bi * G.oi + a, jo = bj * G.oj + b, ko = kk0 + c;
        if (io < P.xd.x && jo < xdy && ko < xdz) P.dst[((size_t)io * xdy + jo) * xdz + ko] = 0.f;
      }
    } else {
      const bool plain0 = (NK > 0 ? NK : P.nk) == 1 && (SK > 0 ? SK : G.sk) == 1;
      const int per_row = plain0 ? npts : min(G.m, xdz - kk0), first = plain0 ? k0 : kk0;
      for (int it = tid; it < ROWS * per_row; it += kBlock) {
        const int row = it / per_row, c = it - row * per_row;
        const int i = i0 + row / TJ, j = j0 + row % TJ;
        if (i < G.gd.x && j < G.gd.y) P.dst[((size_t)i * xdy + j) * xdz + first + c] = 0.f;
      }
    }
#ifdef UNIRES_P2_PROF
    if (pw && threadIdx.x == 0) pw[1] = pw[2] = wall_clock64();
#endif
    return;
  }
  if (tid < SZ4) {
    const int ox = rec[4 + 2 * tid], oy = rec[5 + 2 * tid];
    org[tid] = make_int2(ox, oy);
    tab[tid] = -(ox * H + oy) * SZ;
  }
  if (tid >= kWave && tid < kWave + SZ4 + 1) boff[tid - kWave] = (unsigned)rec[4 + 2 * SZ4 + tid - kWave];
  if (GEN && tid < ROWS) {
    const int ri = tid / G.pj;
    rowi[tid] = (unsigned char)ri, rowj[tid] = (unsigned char)(tid - ri * G.pj);
  }
  __syncthreads();
  // ---- stage the window: one 16-byte piece (4 planes of one column) per item; item n of the
  // load order lands on bytes 16 n of the window.  Pieces outside the volume come from an
  // out-of-range buffer offset (zeros, no branch). ----
  if (!P2_ABL(1)) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(P.src, sd.numel() * sizeof(float));
    // (all origin reads first, then the loads: one LDS round trip per wave instead of one per item)
    unsigned offs[kP2Items];
#pragma unroll
    for (int n = 0; n < kP2Items; ++n)
      offs[n] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(boff) + ie[n].x) + (unsigned)ie[n].y;
    if (!(flags & 2)) {  // the window sticks out of the volume in x / y: range test on every item's column
#pragma unroll
      for (int n = 0; n < kP2Items; ++n) {
        const int base = n * kBlock + wave * kWave;
        if (base >= nitem) break;
        const int cc = P.itab2[base + lane];
        const int2 o = *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(org) + 2 * min(ie[n].x, 4 * (SZ4 - 1)));
        const unsigned x = (unsigned)(o.x + (cc & 0xffff)), y = (unsigned)(o.y + (cc >> 16));
        offs[n] = (x < (unsigned)sd.x && y < (unsigned)sd.y) ? offs[n] : kP2OobBase;
      }
    }
#pragma unroll
    for (int n = 0; n < kP2Items; ++n) {
      const int base = n * kBlock + wave * kWave;  // wave-uniform: LDS-DMA writes base + lane
      if (base >= nitem) break;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs, (__attribute__((address_space(3))) void *)(win + 4 * base), 16, offs[n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (flags & 4) {
      // the volume's last plane group is partial (sd.z not a multiple of 4): its 16-byte pieces run
      // into the next column - zero bound wants zeros there.  After every wave's pieces have landed.
      __syncthreads();
      const int gl = (sd.z - 1 - Z0) >> 2, first = sd.z - (Z0 + 4 * gl);  // planes first .. 3 of group gl
      for (int c = tid; c < P.W * H; c += kBlock)
        for (int q = first; q < 4; ++q) win[c * SZ + 4 * gl + q] = 0.f;
    }
  }
  __syncthreads();
#ifdef UNIRES_P2_PROF
  if (pw && threadIdx.x == 0) pw[1] = wall_clock64();
#endif
  // ---- pull: one grid row per wave pass, lanes along grid z; conv_down every 8 rows ----
  // Measured alternatives that did NOT pay on config 3 (this form: 55 us):
  //  * a 96-word column stride (bank = z plane, LDS bank conflicts 48 % -> 4 % of the LDS cycles):
  //    64 us - the window grows by a third, fewer workgroups fit, and LDS was not the limiter;
  //  * computing all addresses / weights before the barrier, while the LDS-DMA is in flight: 65 us -
  //    the waves of a workgroup then do their VALU work together and their LDS reads together
  //    instead of interleaving them;
  //  * persistent workgroups looping over work items (1024 - 2560 of them): 61 - 70 us - the
  //    hardware dispatcher balances 10 240 short workgroups better than a static loop does;
  //  * (r2, with the conv scratch still in LDS of its own) 4 x 8 instead of 8 x 8 rows per workgroup:
  //    the same 55 us.  Once the scratch aliases the window, LDS is the window alone and 4 x 8 rows
  //    (7 x 10 columns = 20 KB, eight workgroups = every wave slot of a CU) beat 8 x 8 (29 KB, five):
  //    config 3 47.6 vs 49.4 us, config 2 22 / 24 vs 26 / 31 us, config 4 155 vs 167 us;
  //  * the rows of a half in groups of 2 / 4 / 8 with their table reads, then their window reads
  //    issued together (sched_barrier between the phases; row by row the compiler serialises two
  //    dependent LDS round trips per row): 65 us - 148 VGPRs, one workgroup fewer per CU, and capped
  //    at 128 it spills;
  //  * the workgroup's geometry (z range, flags, window origins) computed at its head, lane-parallel,
  //    instead of read from the plan's table: 53 / 57 / 60 us for the three channels against 52.5 /
  //    55 / 56 with the table - the table read costs less than ~45 vector instructions and three
  //    cross-lane reductions in front of the first barrier;
  //  * the row part of the coordinates computed by lane r for row r and broadcast with v_readlane:
  //    the same 55 us, and SQ_INSTS_VALU went UP 7 % (the compiler already shares the products of a
  //    row's i with the 8 rows that have it).
  const float kf = (float)(k0 + min(lane, npts - 1));
  const float c0 = G.A.m[2], c1 = G.A.m[6], c2 = G.A.m[10];
  const float t0 = G.A.m[3], t1 = G.A.m[7], t2 = G.A.m[11];
  const float bx = (float)(sd.x - 1), by = (float)(sd.y - 1), bz = (float)(sd.z - 1);
  const float fZ0 = (float)Z0;
  const int xdy = P.xd.y, xdz = P.xd.z;
  const int nk = NK > 0 ? NK : P.nk, sk = SK > 0 ? SK : G.sk;
  const bool plain = nk == 1 && sk == 1;
  const int nout = min(G.m, xdz - kk0);
  const float inv_m = P.inv_m;
  // A wave first pulls all of its rows (one value per lane and row stays in a register), then the
  // workgroup meets once and the rows go through the conv in a scratch that ALIASES the window -
  // dead by then: no LDS of its own for the scratch (it was 8.3 KB, 16.6 KB with profiles along x / y),
  // so a fifth workgroup fits a CU where the window is <= 32 KB.
  // one sample: the trilinear value of p at grid point (i, j, kfv), from the window (masked by the field
  // of view where the workgroup is not wholly inside)
  auto sample = [&](auto i, auto j, float kfv) {
    // affine_row / affine_along, bit for bit, with x and y as ONE packed pair (written out: the file is compiled without
    // the SLP vectoriser, whose other pairings - the y and x lerps below - cost more moves than they save)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float fi = (float)i, fj = (float)j;
    const v2f ai = {G.A.m[0] * fi, G.A.m[4] * fi}, aj = {G.A.m[1], G.A.m[5]}, jj = {fj, fj};
    const v2f cxy = {c0, c1}, txy = {t0, t1}, kk = {kfv, kfv};
    const v2f gxy = __builtin_elementwise_fma(cxy, kk, __builtin_elementwise_fma(aj, jj, ai)) + txy;
    const float gx = gxy.x, gy = gxy.y;
    const float gz = fmaf(c2, kfv, fmaf(G.A.m[9], fj, G.A.m[8] * fi)) + t2;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const float wx = gx - fx, wy = gy - fy, wz = gz - fz;
    const int zl = (int)(fz - fZ0);
    const int xy = (int)fmaf(fx, (float)(H * SZ), fy * (float)SZ);
    const int a0 = xy + zl + tab[zl >> 2], a1 = xy + zl + 1 + tab[(zl + 1) >> 2];
    float v = 0.f;
    if (!P2_ABL(2)) {
      // the four ds_read2_b32 return (y, y + 1) pairs: the z interpolation runs on the pairs as they
      // come (v_pk_add_f32 / v_pk_fma_f32, no register shuffles), y and x on scalars
      const v2f P0 = {win[a0], win[a0 + SZ]}, R0 = {win[a0 + H * SZ], win[a0 + (H + 1) * SZ]};
      const v2f P1 = {win[a1], win[a1 + SZ]}, R1 = {win[a1 + H * SZ], win[a1 + (H + 1) * SZ]};
      const v2f wz2 = {wz, wz};
      const v2f Q0 = __builtin_elementwise_fma(wz2, P1 - P0, P0), Q1 = __builtin_elementwise_fma(wz2, R1 - R0, R0);
      const float q0 = fmaf(wy, Q0.y - Q0.x, Q0.x), q1 = fmaf(wy, Q1.y - Q1.x, Q1.x);
      v = fmaf(wx, q1 - q0, q0);
    }
    if (!inside) {  // zero bound comes from the zero-filled window; the in-FOV mask is explicit
      const bool in = gx > -P.tol && gx < bx + P.tol && gy > -P.tol && gy < by + P.tol && gz > -P.tol &&
                      gz < bz + P.tol;
      v = in ? v : 0.f;
    }
    return v;
  };
  float hvall[RPW];
#pragma unroll
  for (int h0 = 0; h0 < RPW; h0 += HALF) {
    float hv[HALF];
#pragma unroll
    for (int r = 0; r < HALF; ++r) {
      const int row = wave * RPW + h0 + r;
      const int rr = GEN ? min(row, nrows - 1) : row;  // (rows past the layout replay the last one)
      const int i = min(i0 + (GEN ? (int)rowi[rr] : row / TJ), G.gd.x - 1),
                j = min(j0 + (GEN ? (int)rowj[rr] : row % TJ), G.gd.y - 1);
      const float v = sample(i, j, kf);
      hv[r] = lane < npts ? v : 0.f;
    }
#pragma unroll
    for (int r = 0; r < HALF; ++r) hvall[h0 + r] = hv[r];
  }
  // the chunk's grid points beyond the 64 lanes (extended chunks only): lane = (row of this wave, extra)
  const int next = GEN ? 0 : max(0, min(G.span, G.gd.z - k0) - kWave);
  float hx = 0.f;
  int xrow = 0, xcol = 0;
  bool xact = false;
  if (!GEN && next > 0) {
    const int r = (int)(((float)lane + 0.5f) / (float)next), t = lane - r * next;
    xact = r < RPW;
    xrow = wave * RPW + min(r, RPW - 1), xcol = kWave + t;
    const int i = min(i0 + xrow / TJ, G.gd.x - 1), j = min(j0 + xrow % TJ, G.gd.y - 1);
    const float v = sample(i, j, (float)(k0 + kWave + t));
    hx = xact ? v : 0.f;
  }
  float (*scr)[SCR] = reinterpret_cast<float (*)[SCR]>(win);
  if (P2_ABL(4)) {
    if (hvall[0] + hvall[RPW - 1] == 123.f) P.dst[0] = 1.f;
  } else if (GEN) {  // keep the rows for the workgroup-wide separable conv below
    __syncthreads();  // every wave is done with the window
#pragma unroll
    for (int r = 0; r < RPW; ++r) scr[wave * RPW + r][lane] = hvall[r];
  } else if (plain) {  // no slice profile: the pulled rows are the output
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = wave * RPW + r;
      const int i = i0 + row / TJ, j = j0 + row % TJ;
      if (i < G.gd.x && j < G.gd.y && lane < npts)
        P.dst[((size_t)i * xdy + j) * xdz + k0 + lane] = hvall[r] * P.se;
    }
  } else {
    __syncthreads();  // every wave is done with the window; the scratch rows below are wave-private
#pragma unroll
    for (int r = 0; r < RPW; ++r) scr[wave * RPW + r][lane] = hvall[r];
    if (xact) scr[xrow][xcol] = hx;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = lane; it < RPW * G.m; it += kWave) {
      const int r = (int)(((float)it + 0.5f) * inv_m), win_i = it - __mul24(r, G.m);  // exact: it < 2^20
      const int row = wave * RPW + r;
      const int i = i0 + row / TJ, j = j0 + row % TJ;
      const float *h = &scr[0][0] + __mul24(row, SCR) + __mul24(win_i, sk);
      float acc = 0.f;
      if (NK > 0) {
#pragma unroll
        for (int t = 0; t < NK; ++t) acc += h[t] * P.kz[t];
      } else {
        for (int t = 0; t < nk; ++t) acc += h[t] * P.kz[t];
      }
      const int kk = kk0 + win_i;
      acc *= (kk & 1) ? P.so : P.se;
      // (x-space indices fit 24-bit multiplies: p2_geometry checks fits_fast_index(xd))
      if (win_i < nout && i < G.gd.x && j < G.gd.y)
        P.dst[__umul24(__umul24((unsigned)i, (unsigned)xdy) + (unsigned)j, (unsigned)xdz) + (unsigned)kk] = acc;
    }
  }
#ifdef UNIRES_P2_PROF
  if (pw && threadIdx.x == 0) pw[2] = wall_clock64();
#endif
  if (GEN) {
    // ---- separable conv_down over the workgroup's rows: oi x oj x m x-space voxels ----
    __syncthreads();
    const int ojm = G.oj * G.m, nitem = G.oi * ojm;
    const float inv_ojm = 1.f / (float)ojm;
    const int ib = bi * G.oi, jb = bj * G.oj;
    for (int it = tid; it < nitem; it += kBlock) {
      const int a = (int)(((float)it + 0.5f) * inv_ojm), rem = it - a * ojm;
      const int b = (int)(((float)rem + 0.5f) * inv_m), c = rem - b * G.m;
      const int io = ib + a, jo = jb + b, ko = kk0 + c;
      if (io >= P.xd.x || jo >= xdy || ko >= xdz) continue;
      float acc = 0.f;
      const int r0 = a * G.si * G.pj + b * G.sj;  // first row of the footprint
      if (nk == 1 && G.nj == 1) {                 // profile along x only
        const float *h = &scr[r0][c * sk];
        for (int ta = 0; ta < G.ni; ++ta) acc += P.kx[ta] * h[ta * G.pj * SCR];
      } else if (nk == 1 && G.ni == 1) {          // along y only
        const float *h = &scr[r0][c * sk];
        for (int tb = 0; tb < G.nj; ++tb) acc += P.ky[tb] * h[tb * SCR];
      } else {
        for (int ta = 0; ta < G.ni; ++ta) {
          const float wa = P.kx[ta];
          for (int tb = 0; tb < G.nj; ++tb) {
            const float wab = wa * P.ky[tb];
            const float *h = &scr[r0 + ta * G.pj + tb][c * sk];
            float az = 0.f;
            for (int tc = 0; tc < nk; ++tc) az += h[tc] * P.kz[tc];
            acc += wab * az;
          }
        }
      }
      if (P.sdim >= 0) acc *= ((P.sdim == 0 ? io : (P.sdim == 1 ? jo : ko)) & 1) ? P.so : P.se;
      P.dst[((size_t)io * xdy + jo) * xdz + ko] = acc;
    }
  }
}

static bool same_geom(const P2Geom &a, const P2Geom &b) { return memcmp(&a, &b, sizeof(P2Geom)) == 0; }

void pull2_free(PullPlan &Q) {
  if (Q.rec) (void)hipFree(Q.rec);
  if (Q.itab) (void)hipFree(Q.itab);
  if (Q.wtab) (void)hipFree(Q.wtab);
  Q = PullPlan();
}

// Geometry of the operator for this kernel; false: outside its domain.
static bool p2_geometry(Dim3i sd, const Affine &A, const Taps &T, const Scaling &S, Dim3i xd, Dim3i gd,
                        P2Geom &G, int &W, int &H) {
  (void)S;
  for (int d = 0; d < 3; ++d)
    if (T.n[d] > 32 || T.s[d] > T.n[d] || T.n[d] < 1 || T.s[d] < 1) return false;
  if (T.n[2] > kWave - 8) return false;
  const int gdv[3] = {gd.x, gd.y, gd.z}, xdv[3] = {xd.x, xd.y, xd.z};
  for (int d = 0; d < 3; ++d)
    if (gdv[d] != (xdv[d] - 1) * T.s[d] + T.n[d]) return false;
  if (sd.numel() >= (1ull << 29) || !fits_fast_index(sd) || !fits_fast_index(xd)) return false;  // (byte offsets + kP2OobBase stay below 2^32)
  const double a22 = A.m[10];
  if (!(fabs(a22) > 0.5)) return false;
  memset(&G, 0, sizeof(G));
  G.A = A, G.sd = sd, G.gd = gd;
  const double sxd = A.m[2] / a22, syd = A.m[6] / a22;
  G.pxi = (float)(A.m[0] - sxd * A.m[8]), G.pxj = (float)(A.m[1] - sxd * A.m[9]);
  G.pyi = (float)(A.m[4] - syd * A.m[8]), G.pyj = (float)(A.m[5] - syd * A.m[9]);
  G.sx = (float)sxd, G.sy = (float)syd;
  G.cx = (float)(A.m[3] - sxd * A.m[11]), G.cy = (float)(A.m[7] - syd * A.m[11]);
  G.si = T.s[0], G.sj = T.s[1], G.ni = T.n[0], G.nj = T.n[1];
  G.sk = T.s[2];
  G.m = (kWave - T.n[2]) / T.s[2] + 1;  // whole conv windows inside 64 grid points
  G.span = (G.m - 1) * T.s[2] + T.n[2];
  const bool zonly = T.n[0] == 1 && T.s[0] == 1 && T.n[1] == 1 && T.s[1] == 1;
  // rows of a workgroup: the (oi, oj) with the most x-space rows per sampled grid row whose window
  // fits; window extents = span of the rows over a plane group (z in [Zg - 1, Zg + 4)), + floor, +
  // the upper corner, + rounding margin
  auto extents = [&](int pi, int pj, int &w, int &hn) {
    const double ex = (pi - 1) * fabs(G.pxi) + (pj - 1) * fabs(G.pxj) + 5.0 * fabs(sxd) + 0.05;
    const double ey = (pi - 1) * fabs(G.pyi) + (pj - 1) * fabs(G.pyj) + 5.0 * fabs(syd) + 0.05;
    w = (int)floor(ex) + 3, hn = (int)floor(ey) + 3;
  };
  auto fits = [&](int pi, int pj, int &w, int &h) {
    int hn;
    extents(pi, pj, w, hn);
    // z planes: span of gz over the workgroup + floor + upper corner + alignment of Z0 to 4
    const double ez = (pi - 1) * fabs((double)A.m[8]) + (pj - 1) * fabs((double)A.m[9]) +
                      (double)(std::max(G.span, (int)kWave) - 1) * fabs(a22) + 0.05;
    if ((int)floor(ez) + 2 + 3 + 1 > kP2SZ) return false;
    if (w > 24 || hn > 16) return false;
    h = hn <= 10 ? 10 : (hn <= 12 ? 12 : 16);
    if (w * h * kP2SZ4 > kP2Items * kBlock) return false;
    return (size_t)w * h * kP2SZ * sizeof(float) <= (zonly ? 56u : 40u) * 1024u;
  };
  if (zonly) {
    G.pi = G.oi = kP2TI, G.pj = G.oj = kP2TJ;
    // one window more per chunk where that saves a chunk and the taller window still fits
    static const bool ext_ok = !(getenv("UNIRES_P2_EXT") && atoi(getenv("UNIRES_P2_EXT")) == 0);
    const int m0 = G.m, span0 = G.span, span1 = m0 * T.s[2] + T.n[2];
    bool ext = ext_ok && !(T.n[2] == 1 && T.s[2] == 1) && span1 - kWave <= kP2MaxExt &&
               (xd.z + m0) / (m0 + 1) < (xd.z + m0 - 1) / m0;
    if (ext) {
      G.m = m0 + 1, G.span = span1;
      ext = fits(G.pi, G.pj, W, H);
    }
    if (!ext) {
      G.m = m0, G.span = span0;
      if (!fits(G.pi, G.pj, W, H)) return false;
    }
  } else {
    double best = 0.0;
    for (int oi = 1; oi <= 16; ++oi)
      for (int oj = 1; oj <= 16; ++oj) {
        const int pi = (oi - 1) * T.s[0] + T.n[0], pj = (oj - 1) * T.s[1] + T.n[1];
        int w, h;
        if (pi > 16 || pj > 16 || pi * pj > kP2Rows || !fits(pi, pj, w, h)) continue;
        const double score = (double)(oi * T.s[0]) * (oj * T.s[1]) / (pi * pj) + 1e-3 * (oi * oj);
        if (score > best) best = score, G.oi = oi, G.oj = oj, G.pi = pi, G.pj = pj, W = w, H = h;
      }
    if (best == 0.0) return false;
  }
  G.nbj = (xd.y + G.oj - 1) / G.oj;
  G.nbc = (xd.z + G.m - 1) / G.m;
  return true;
}

static long long p2_blocks(const P2Geom &G, Dim3i xd) {
  return (long long)((xd.x + G.oi - 1) / G.oi) * G.nbj * G.nbc;
}

// dynamic LDS of a workgroup: the window, or the conv scratch that aliases it (64 rows x 65 floats) if that is larger
static size_t p2_lds(int W, int H, bool gen) {
  const size_t scratch = gen ? (size_t)kP2Rows * (kWave + 1) : (size_t)(kP2TI * kP2TJ) * kP2Scr;
  return std::max((size_t)W * H * kP2SZ, scratch) * sizeof(float);
}

// block rows walked together (see the kernel's block mapping): all of them where a block row's windows fit the L2
static int p2_band(const P2Geom &G, long long nblk, size_t lds) {
  const long long nbi = nblk / ((long long)G.nbc * G.nbj);
  const bool row_fits_l2 = (double)G.nbj * G.nbc * (double)lds < 3.5e6;
  int band = (int)(row_fits_l2 ? std::max<long long>(nbi, 1) : kP2Band);
  static const int band_env = getenv("UNIRES_P2_BAND") ? atoi(getenv("UNIRES_P2_BAND")) : 0;  // (measurement)
  if (band_env > 0) band = (int)std::min<long long>(band_env, std::max<long long>(nbi, 1));
  return band;
}

int pull2_build(PullPlan &Q, Dim3i sd, const Affine &A, const Taps &T, Dim3i xd, Dim3i gd, float tol) {
  Q.valid = false;
  static const bool off = getenv("UNIRES_NO_PULL2") != nullptr;
  if (off) return 1;
  P2Geom G;
  int W, H;
  if (!p2_geometry(sd, A, T, Scaling{1.f, 1.f, -1}, xd, gd, G, W, H)) return 1;
  const long long nblk = p2_blocks(G, xd);
  if (nblk > 0x3fffffffll) return 1;
  if ((size_t)nblk > Q.cap) {
    if (Q.rec) (void)hipFree(Q.rec);
    Q.rec = nullptr;
    if (hipMalloc((void **)&Q.rec, (size_t)nblk * kP2Rec * sizeof(int)) != hipSuccess) return 1;
    Q.cap = (size_t)nblk;
  }
  // staging items of a window (the same for every workgroup of the operator), padded to whole waves
  const int nitem = W * H * kP2SZ4, npad = (nitem + kWave - 1) / kWave * kWave;
  if ((size_t)npad > Q.itab_cap) {
    if (Q.itab) (void)hipFree(Q.itab);
    Q.itab = nullptr;
    if (hipMalloc((void **)&Q.itab, (size_t)npad * 3 * sizeof(int)) != hipSuccess) return 1;
    Q.itab_cap = (size_t)npad;
  }
  {
    std::vector<int> host((size_t)Q.itab_cap * 3, 0);
    int *fast = host.data(), *col = host.data() + 2 * Q.itab_cap;
    for (int it = 0; it < npad; ++it) {
      const int zg = it % kP2SZ4, slot = it / kP2SZ4, cxl = slot / H, cyl = slot - cxl * H;
      const bool real = it < nitem;
      fast[2 * it] = real ? 4 * zg : 4 * kP2SZ4;
      fast[2 * it + 1] = real ? (int)(4u * (unsigned)((cxl * sd.y + cyl) * sd.z)) : 0;
      col[it] = real ? (cxl | (cyl << 16)) : 0;
    }
    if (hipMemcpy(Q.itab, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return 1;
  }
  if ((size_t)nblk > Q.wtab_cap) {
    if (Q.wtab) (void)hipFree(Q.wtab);
    Q.wtab = nullptr;
    // (the table, then one class byte per workgroup)
    if (hipMalloc((void **)&Q.wtab, (size_t)nblk * (4 * sizeof(int) + 1)) != hipSuccess) return 1;
    Q.wtab_cap = (size_t)nblk;
  }
  unsigned char *cls_dev = reinterpret_cast<unsigned char *>(Q.wtab + 4 * Q.wtab_cap);
  hipLaunchKernelGGL(k_pull2_plan, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, 0, G, (int)nblk, tol, W, H,
                     Q.rec, cls_dev);
  Q.tol = tol;
  // Who runs where (r6).  Workgroup b runs on XCD b % 8, so every XCD gets the same NUMBER of workgroups; walking the
  // volume in one contiguous run per XCD (shared window columns stay in its L2) gave the XCDs at the volume's ends
  // the workgroups that see nothing - config 4: XCD 7 done at 114 of 168 us - and those at its faces the expensive
  // ones (FOV mask, range tests on staging: config 3: XCD 0 / 1 done at 45 us, the others at 40.5;
  // tools/p2_timeline.py).  Here the non-empty workgroups are cut into 8 contiguous runs of equal COST (plain 8,
  // masked / range-tested 10), and each XCD's count is made up with empty ones.
  Q.use_wtab = false;
  {
    std::vector<unsigned char> cls((size_t)nblk);
    if (hipMemcpy(cls.data(), cls_dev, (size_t)nblk, hipMemcpyDeviceToHost) != hipSuccess) return 1;  // (synchronises)
    static const bool off_tab = getenv("UNIRES_P2_WTAB") && atoi(getenv("UNIRES_P2_WTAB")) == 0;
    const bool gen = !(T.n[0] == 1 && T.s[0] == 1 && T.n[1] == 1 && T.s[1] == 1);
    const int bandh = p2_band(G, nblk, p2_lds(W, H, gen));
    const long long nbi = nblk / ((long long)G.nbc * G.nbj);
    if (!off_tab && nblk >= 64) {
      // the walk: position w -> workgroup record (the kernel's own mapping)
      long long w_bi = 0, w_bj = 0, w_bc = 0;
      auto blk_of = [&](long long w) {
        const long long bc = w % G.nbc, pair = w / G.nbc;
        const long long band = pair / ((long long)bandh * G.nbj), rem = pair - band * ((long long)bandh * G.nbj);
        const long long bh = std::min<long long>(bandh, nbi - band * bandh);
        const long long bj = rem / bh, bi = band * bandh + (rem - bj * bh);
        w_bi = bi, w_bj = bj, w_bc = bc;
        return (bi * G.nbj + bj) * G.nbc + bc;
      };
      std::vector<int> ne, em;
      ne.reserve((size_t)nblk);
      long long cost_all = 0;
      std::vector<unsigned char> wc((size_t)nblk);
      for (long long w = 0; w < nblk; ++w) {
        const unsigned char c = cls[(size_t)blk_of(w)];
        wc[(size_t)w] = c;
        if (c == 0) em.push_back((int)w); else ne.push_back((int)w), cost_all += c == 1 ? 8 : 10;
      }
      std::vector<int> tab((size_t)nblk, 0);
      size_t in = 0, ie = 0;
      long long cum = 0;
      bool ok = true;
      for (int x = 0; x < 8 && ok; ++x) {
        const long long cnt = nblk / 8 + (x < (int)(nblk % 8) ? 1 : 0);
        // what the XCDs after this one can still take bounds how few this one may take
        long long later = 0;
        for (int y = x + 1; y < 8; ++y) later += nblk / 8 + (y < (int)(nblk % 8) ? 1 : 0);
        long long i = 0;
        const long long target = cost_all * (x + 1) / 8;
        while (i < cnt && in < ne.size() && (cum < target || (long long)(ne.size() - in) > later || x == 7)) {
          const int w = ne[in++];
          cum += wc[(size_t)w] == 1 ? 8 : 10;
          tab[(size_t)(x + 8 * i++)] = w;
        }
        while (i < cnt && ie < em.size()) tab[(size_t)(x + 8 * i++)] = em[ie++];
        while (i < cnt && in < ne.size()) {  // (no empty one left: non-empty ones after all)
          const int w = ne[in++];
          cum += wc[(size_t)w] == 1 ? 8 : 10;
          tab[(size_t)(x + 8 * i++)] = w;
        }
        ok = i == cnt;
      }
      ok = ok && in == ne.size() && ie == em.size();
      // what the kernel reads: {record, bi, bj, bc} of the walk's position
      std::vector<int> tab4((size_t)nblk * 4);
      for (long long d = 0; d < nblk && ok; ++d) {
        const long long b = blk_of(tab[(size_t)d]);
        tab4[4 * (size_t)d] = (int)b, tab4[4 * (size_t)d + 1] = (int)w_bi, tab4[4 * (size_t)d + 2] = (int)w_bj,
                         tab4[4 * (size_t)d + 3] = (int)w_bc;
      }
      if (ok && hipMemcpy(Q.wtab, tab4.data(), tab4.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess)
        Q.use_wtab = true;
    }
  }
  static const bool verbose = getenv("UNIRES_PULL2_VERBOSE") != nullptr;
  if (verbose) {  // one line per plan: what the workgroups of this operator look like
    std::vector<int> rec((size_t)nblk * kP2Rec);
    if (hipMemcpy(rec.data(), Q.rec, rec.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
      long long empty = 0, masked = 0, slow = 0, partial = 0;
      double groups = 0.0;
      for (long long b = 0; b < nblk; ++b) {
        const int *r = rec.data() + (size_t)b * kP2Rec;
        const bool e = (r[3] & 1) != 0;
        empty += e;
        if (e) continue;
        masked += r[2] == 0, slow += (r[3] & 2) == 0, partial += (r[3] & 4) != 0, groups += r[1];
      }
      fprintf(stderr,
              "[pull2] %lld workgroups (%d x %d rows, %d chunks along z of %d windows), window %d x %d columns x %d planes; "
              "%lld all outside the field of view, of the rest %lld with the FOV mask, %lld staging with x / y range tests, "
              "%lld with a partial last plane group, %.1f of %d plane groups staged on average\n",
              nblk, G.pi, G.pj, G.nbc, G.m, W, H, kP2SZ, empty, masked, slow, partial,
              groups / (double)std::max<long long>(1, nblk - empty), kP2SZ4);
    }
  }
  static_assert(sizeof(P2Geom) <= sizeof(Q.key), "PullPlan key too small");
  memset(Q.key, 0, sizeof(Q.key));
  memcpy(Q.key, &G, sizeof(G));
  Q.valid = true;
  return 0;
}

// Non-zero return: operator outside this kernel's domain / no plan for it (nothing launched).
int launch_pull_conv2(const PullPlan &Q, const float *src, Dim3i sd, const Affine &A, const Taps &T,
                      const Scaling &S, float *dst, Dim3i xd, Dim3i gd, float tol, const int *done,
                      hipStream_t st) {
  if (!Q.valid) return 1;
  P2Args P;
  int W, H;
  if (!p2_geometry(sd, A, T, S, xd, gd, P.G, W, H)) return 1;
  P2Geom key;
  memcpy(&key, Q.key, sizeof(key));
  if (!same_geom(key, P.G) || tol != Q.tol) return 1;  // the plan was built for another operator / mask tolerance
  P.src = src, P.rec = Q.rec;
  P.itab = reinterpret_cast<const int2 *>(Q.itab), P.itab2 = Q.itab + 2 * Q.itab_cap;
  P.inv_m = 1.f / (float)P.G.m;
  for (int i = 0; i < UNIRES_MAX_TAPS; ++i) {
    P.kz[i] = i < T.n[2] ? T.t[2][i] : 0.f;
    P.kx[i] = i < T.n[0] ? T.t[0][i] : 0.f;
    P.ky[i] = i < T.n[1] ? T.t[1][i] : 0.f;
  }
  P.nk = T.n[2];
  const bool gen = !(T.n[0] == 1 && T.s[0] == 1 && T.n[1] == 1 && T.s[1] == 1);
  if (!gen && S.dim >= 0 && S.dim != 2) return 1;  // (the wave-private conv scales along z only)
  P.sdim = S.dim;
  P.se = S.dim >= 0 ? S.e : 1.f, P.so = S.dim >= 0 ? S.o : 1.f;
  P.dst = dst, P.xd = xd, P.tol = tol, P.W = W;
  static const int dbg = getenv("UNIRES_P2_DBG") ? atoi(getenv("UNIRES_P2_DBG")) : 0;
  P.dbg = dbg;
  // (UNIRES_P2_LDS_PAD=<bytes>: extra dynamic LDS per workgroup = fewer workgroups per CU - measurement of how the
  // kernel shares a CU with other channels' kernels)
  static const size_t lds_pad = getenv("UNIRES_P2_LDS_PAD") ? (size_t)atol(getenv("UNIRES_P2_LDS_PAD")) : 0;
  const size_t lds = p2_lds(W, H, gen) + lds_pad;
  const dim3 grid((unsigned)p2_blocks(P.G, xd)), block(kBlock);
  P.band = p2_band(P.G, (long long)grid.x, p2_lds(W, H, gen));
  P.wtab = Q.use_wtab ? reinterpret_cast<const int4 *>(Q.wtab) : nullptr;
  P.prof = nullptr;
#ifdef UNIRES_P2_PROF
  static unsigned long long *prof_dev = nullptr;
  const size_t nprof = (size_t)grid.x * 8;
  if (!prof_dev) (void)hipMalloc((void **)&prof_dev, (size_t)(1 << 20) * 8 * sizeof(unsigned long long));
  if (grid.x <= (1u << 20)) {
    (void)hipMemsetAsync(prof_dev, 0, nprof * sizeof(unsigned long long), st);
    P.prof = prof_dev;
  }
#endif
  const bool k76 = T.n[2] == 7 && T.s[2] == 6;
  static const bool no_k112 = getenv("UNIRES_P2_K112") && atoi(getenv("UNIRES_P2_K112")) == 0;
  const bool k112 = !no_k112 && T.n[2] == 11 && T.s[2] == 2;  // the default Gaussian profile at ratio 2 (BASELINE config 4) ...
  const bool k92 = !no_k112 && T.n[2] == 9 && T.s[2] == 2;    // ... and as the plan trims it (its +-5 taps are 3e-8 of the sum)
  const bool k32 = !no_k112 && T.n[2] == 3 && T.s[2] == 2;     // the (trimmed) rect profile at ratio 2
#define P2_LAUNCH(HH)                                                                        \
  do {                                                                                       \
    if (gen)                                                                                 \
      hipLaunchKernelGGL((k_pull_conv2<HH, 0, 0, true>), grid, block, lds, st, P, done);     \
    else if (k76)                                                                            \
      hipLaunchKernelGGL((k_pull_conv2<HH, 7, 6, false>), grid, block, lds, st, P, done);    \
    else if (k112)                                                                           \
      hipLaunchKernelGGL((k_pull_conv2<HH, 11, 2, false>), grid, block, lds, st, P, done);   \
    else if (k92)                                                                            \
      hipLaunchKernelGGL((k_pull_conv2<HH, 9, 2, false>), grid, block, lds, st, P, done);    \
    else if (k32)                                                                            \
      hipLaunchKernelGGL((k_pull_conv2<HH, 3, 2, false>), grid, block, lds, st, P, done);    \
    else                                                                                     \
      hipLaunchKernelGGL((k_pull_conv2<HH, 0, 0, false>), grid, block, lds, st, P, done);    \
  } while (0)
  if (H == 10)
    P2_LAUNCH(10);
  else if (H == 12)
    P2_LAUNCH(12);
  else
    P2_LAUNCH(16);
#undef P2_LAUNCH
#ifdef UNIRES_P2_PROF
  {
    static int shots = 0;
    if (P.prof && ++shots == 12 && getenv("UNIRES_P2_PROF_OUT")) {  // one warmed-up launch
      std::vector<unsigned long long> h(nprof);
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h.data(), prof_dev, nprof * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      FILE *f = fopen(getenv("UNIRES_P2_PROF_OUT"), "w");
      fprintf(f, "# lds_bytes %zu\n", lds);
      for (size_t w = 0; w < nprof / 8; ++w) {
        for (int i = 0; i < 8; ++i) fprintf(f, "%llu ", h[w * 8 + i]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  return 0;
}

}  // namespace unires
