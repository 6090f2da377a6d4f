// stencil.hip - k_dtd_flat: q = a0 p + c DtD p (+ sum p*q, or the CG objective) for regime A = I
// (UniRes' denoising of registered images: unires/_project.py:76-77 returns dat, :300-317 _DtD),
// as ONE streaming pass over the volume seen as a flat array.
//
// r1 / r2 ran this regime through the aligned line kernel: one wave per z line, dword loads, five
// global loads per output voxel (centre + four x / y neighbours) and 64-lane passes over lines
// that are not multiples of 64 long (181 -> 3 passes, 94 % of the third idle): 19.7 us for
// 181 x 217 x 181 = 0.36 of the HBM peak.  Here
//   * a lane owns FOUR consecutive voxels of the flat array - lines are ignored, every lane of every
//     wave is busy whatever the line length - and q goes out as aligned 16-byte stores;
//   * centre, y-1, y+1, x-1, x+1 are 16-byte loads at flat offsets 0, -nz, +nz, -ny nz, +ny nz
//     (five vector-memory instructions per 256 outputs instead of 20); the y neighbours are L1 hits
//     (the same lines a neighbouring lane loads as its centre), the x neighbours L2 hits: each XCD
//     walks ONE contiguous range of the volume, so a line is fetched from HBM once per XCD range;
//   * the z neighbours never touch memory: inside the lane's four values, across lanes by DPP wave
//     shifts; only lanes 0 and 63 of a wave load one extra dword each;
//   * which of a lane's voxels sit on a volume face follows from ONE position per lane (k of its
//     first voxel; a line can end at most once inside four voxels) - exact float reciprocals of small
//     integers, no integer division.
// The stencil is evaluated in difference form ((c - lower) - (upper - c), zero bound above, no
// backward term on the first plane), like dtd_at() of the other kernels.
#include <stdlib.h>

#include <algorithm>

#include "stencil.hpp"

namespace unires {

#ifndef UNIRES_FLAT_VECS
#define UNIRES_FLAT_VECS 1
#endif
constexpr int kFlatVecs = UNIRES_FLAT_VECS;            // float4 per thread and chunk
constexpr int kFlatChunk = kBlock * 4 * kFlatVecs;     // voxels per chunk (2048)

struct FlatArgs {
  const float *p;
  float *q;
  const float *objb;
  double *partials;
  unsigned n, nz, ny, nynz;
  float inv_nz, inv_ny;
  unsigned head;    // voxels in front of the first vector (q + head is 16-byte aligned)
  unsigned nvec;    // whole vectors after the head
  unsigned nchunk;  // ceil(nvec * 4 / kFlatChunk)
  float a0, cx, cy, cz;
  // chunk ranges of the XCDs, composed on the host (no division in the kernel's prologue): XCD x walks
  // chunks cb[x] .. cb[x + 1] with its G / nx (+ 1 for x < rem) workgroups; nx = 8, or 1 for tiny grids
  unsigned nx, per_xcd, rem, cb[9];
};

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int kAuxNt = 2;  // cache-policy bit of the buffer instructions: non-temporal (q is not read again soon)

__device__ __forceinline__ f4 ld4_fast(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
// element-wise form for vectors that stick out of the array: voxels below 0 or at / beyond n read as
// 0.  The range test is explicit: the compiler folds "+ 4 e" into the instruction's immediate offset,
// and the hardware adds that WITHOUT wrapping at 32 bits, so a negative (wrapped) base does not come
// back into range the way 32-bit arithmetic would suggest.
__device__ __forceinline__ f4 ld4_safe(__amdgpu_buffer_rsrc_t r, int idx, int n) {
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = idx + e;
    v[e] = buf_load(r, (i >= 0 && i < n) ? 4u * (unsigned)i : 0x80000000u, 0);
  }
  return f4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ float dpp_from_lower_lane(float v) {  // lane l gets lane l - 1's value (lane 0: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float v) {  // lane l gets lane l + 1's value (lane 63: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// v / d for v < 2^24 with the reciprocal passed in: exact after the two fix-ups
__device__ __forceinline__ unsigned div_small(unsigned v, unsigned d, float inv_d) {
  unsigned q = (unsigned)(((float)v + 0.5f) * inv_d);
  if (__umul24(q, d) > v) --q;
  if (__umul24(q + 1u, d) <= v) ++q;
  return q;
}

// One vector (four consecutive voxels of a lane).  XY = false: the caller guarantees that none of the
// wave's voxels lies on an x or y face of the volume and that all five vectors are inside the array
// (the common case: whole chunks away from the first / last rows and slabs), so only the z faces - a
// line end somewhere inside the wave - are tested.
struct FlatVec {
  f4 cc, xm, xp, ym, yp, ob;
  float edge;
};

template <bool XY, bool DOT, bool OBJ>
__device__ __forceinline__ void flat_vec(const FlatArgs &A, const FlatVec &L, unsigned lane, unsigned idx0,
                                         unsigned k0, unsigned j0, bool valid, __amdgpu_buffer_rsrc_t rq,
                                         double &dot) {
  const unsigned nz = A.nz, ny = A.ny;
  float zlo = dpp_from_lower_lane(L.cc.w), zhi = dpp_from_upper_lane(L.cc.x);
  zlo = lane == 0u ? L.edge : zlo;
  zhi = lane == (unsigned)kWave - 1u ? L.edge : zhi;
  // faces.  A line ends at most once inside the lane's four voxels (nz >= 4): after voxel w.
  const unsigned w = nz - 1u - k0;
  const float c4[4] = {L.cc.x, L.cc.y, L.cc.z, L.cc.w};
  const float zm4[4] = {zlo, L.cc.x, L.cc.y, L.cc.z}, zp4[4] = {L.cc.y, L.cc.z, L.cc.w, zhi};
  const float xm4[4] = {L.xm.x, L.xm.y, L.xm.z, L.xm.w}, xp4[4] = {L.xp.x, L.xp.y, L.xp.z, L.xp.w};
  const float ym4[4] = {L.ym.x, L.ym.y, L.ym.z, L.ym.w}, yp4[4] = {L.yp.x, L.yp.y, L.yp.z, L.yp.w};
  const float ob4[4] = {L.ob.x, L.ob.y, L.ob.z, L.ob.w};
  float out[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ce = c4[e];
    const bool zlo_ok = e == 0 ? k0 != 0u : (unsigned)(e - 1) != w;  // not the first voxel of a line
    const bool zhi_ok = (unsigned)e != w;                              // not the last voxel of a line
    float xb = ce - xm4[e], yb = ce - ym4[e], yp = yp4[e];
    if (XY) {
      const bool wrapped = (unsigned)e > w;  // voxel e lies on the next line
      const unsigned je = wrapped ? (j0 + 1u == ny ? 0u : j0 + 1u) : j0;
      xb = idx0 + (unsigned)e >= A.nynz ? xb : 0.f;  // (x upper face: the load returned 0)
      yb = je != 0u ? yb : 0.f;
      yp = je + 1u != ny ? yp : 0.f;
    }
    const float xf = xp4[e] - ce, yf = yp - ce;
    const float zb = zlo_ok ? ce - zm4[e] : 0.f, zf = (zhi_ok ? zp4[e] : 0.f) - ce;
    const float o = A.a0 * ce + (A.cx * (xb - xf) + A.cy * (yb - yf) + A.cz * (zb - zf));
    out[e] = o;
    if (valid) {
      if (OBJ)
        dot += (double)obj_term(o, ob4[e], ce);
      else if (DOT)
        dot += (double)__fmul_rn(ce, o);
    }
  }
  if (!OBJ && valid)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, f4{out[0], out[1], out[2], out[3]}), rq, 4u * idx0, 0, kAuxNt);
}

template <bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock) k_dtd_flat(FlatArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1);
  const unsigned n = A.n, nz = A.nz, ny = A.ny, nynz = A.nynz;
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(A.p, (size_t)n * 4);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(OBJ ? A.objb : A.p, (size_t)n * 4);
  float *__restrict__ q = A.q;
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(A.q, (size_t)n * 4);
  double dot = 0.0;
  // every XCD (workgroup b sits on XCD b % 8) walks one contiguous range of chunks: the x / y halo
  // lines of a chunk are then in that XCD's L2 already, or will be used from it next
  const unsigned xcd = A.nx == 8u ? blockIdx.x & 7u : 0u, slot = A.nx == 8u ? blockIdx.x >> 3 : blockIdx.x;
  const unsigned cnt = A.per_xcd + (xcd < A.rem ? 1u : 0u);      // workgroups of this XCD
  const unsigned c_lo = A.cb[xcd], c_hi = A.cb[xcd + 1u];
  for (unsigned c = c_lo + slot; c < c_hi; c += cnt) {
    const unsigned e0 = A.head + c * (unsigned)kFlatChunk;       // first voxel of the chunk (wave-uniform)
    const unsigned line0 = e0 / nz, kb = e0 - line0 * nz, jb = line0 % ny;
    // whole chunk away from the x / y faces, every vector of every lane inside the array: the rows it
    // touches are jb .. jb + (kb + chunk - 1) / nz, its slabs lie strictly between the first and last
    const unsigned rows = (kb + (unsigned)kFlatChunk - 1u) / nz;
    const bool plain = jb >= 1u && jb + rows + 1u < ny && e0 >= nynz &&
                       (unsigned long long)e0 + kFlatChunk + nynz <= n &&
                       (c + 1u) * (unsigned)(kFlatChunk / 4) <= A.nvec;
    if (plain) {
      FlatVec L[kFlatVecs];
      unsigned k0[kFlatVecs];
#pragma unroll
      for (int v = 0; v < kFlatVecs; ++v) {  // all loads of the chunk first
        const unsigned t = (unsigned)v * kBlock + tid, bo = 4u * (e0 + 4u * t);
        L[v].cc = ld4_fast(rp, bo), L[v].ym = ld4_fast(rp, bo - 4u * nz), L[v].yp = ld4_fast(rp, bo + 4u * nz);
        L[v].xm = ld4_fast(rp, bo - 4u * nynz), L[v].xp = ld4_fast(rp, bo + 4u * nynz);
        L[v].ob = f4{0.f, 0.f, 0.f, 0.f};
        if (OBJ) L[v].ob = ld4_fast(rb, bo);
        // the voxel below lane 0's first / above lane 63's last; issued by every lane, after the vectors
        // (a load under a lane mask in front of them made the compiler drain it before the other four)
        L[v].edge = buf_load(rp, lane == (unsigned)kWave - 1u ? bo + 16u : bo - 4u, 0);
        const unsigned u = kb + 4u * t;
        k0[v] = u - __umul24(div_small(u, nz, A.inv_nz), nz);
      }
#pragma unroll
      for (int v = 0; v < kFlatVecs; ++v)
        flat_vec<false, DOT, OBJ>(A, L[v], lane, e0 + 4u * ((unsigned)v * kBlock + tid), k0[v], 1u, true, rq, dot);
      continue;
    }
#pragma unroll
    for (int v = 0; v < kFlatVecs; ++v) {
      const unsigned t = (unsigned)v * kBlock + tid;
      const unsigned vi = c * (unsigned)(kFlatChunk / 4) + t;    // vector index
      const bool valid = vi < A.nvec;
      const unsigned idx0 = e0 + 4u * t, bo = 4u * idx0;
      // position of the lane's first voxel: k0 along z, j0 along y
      const unsigned u = kb + 4u * t, ql = div_small(u, nz, A.inv_nz), k0 = u - __umul24(ql, nz);
      const unsigned lj = jb + ql, qj = div_small(lj, ny, A.inv_ny), j0 = lj - __umul24(qj, ny);
      // all five vectors of every lane wholly inside the array?  (false only in the first / last x
      // slab and in the last, partial wave: those take the element-wise loads)
      const bool inner = valid && idx0 >= nynz && idx0 + nynz + 4u <= n;
      FlatVec L;
      L.ob = f4{0.f, 0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(!inner) == 0ull) {
        L.cc = ld4_fast(rp, bo), L.ym = ld4_fast(rp, bo - 4u * nz), L.yp = ld4_fast(rp, bo + 4u * nz);
        L.xm = ld4_fast(rp, bo - 4u * nynz), L.xp = ld4_fast(rp, bo + 4u * nynz);
        if (OBJ) L.ob = ld4_fast(rb, bo);
      } else {
        const int i0 = (int)idx0, in = (int)n;
        L.cc = ld4_safe(rp, i0, in), L.ym = ld4_safe(rp, i0 - (int)nz, in), L.yp = ld4_safe(rp, i0 + (int)nz, in);
        L.xm = ld4_safe(rp, i0 - (int)nynz, in), L.xp = ld4_safe(rp, i0 + (int)nynz, in);
        if (OBJ) L.ob = ld4_safe(rb, i0, in);
      }
      L.edge = 0.f;  // lane 0: the voxel below its first, lane 63: the voxel above its last
      if (lane == 0u || lane == (unsigned)kWave - 1u) {
        const int ie = lane == 0u ? (int)idx0 - 1 : (int)idx0 + 4;
        L.edge = buf_load(rp, (ie >= 0 && ie < (int)n) ? 4u * (unsigned)ie : 0x80000000u, 0);
      }
      flat_vec<true, DOT, OBJ>(A, L, lane, idx0, k0, j0, valid, rq, dot);
    }
  }
  // the few voxels in front of the first and behind the last vector
  if (blockIdx.x == 0) {
    const unsigned tail0 = A.head + 4u * A.nvec, nedge = A.head + (n - tail0);
    if (tid < nedge) {
      const unsigned idx = tid < A.head ? tid : tail0 + (tid - A.head);
      const unsigned line = idx / nz, k = idx - line * nz, i = line / ny, j = line - i * ny;
      const Dim3i dd{(int)(n / nynz), (int)ny, (int)nz};
      float pc;
      const float st = dtd_at(A.p, idx, (int)i, (int)j, (int)k, dd, A.cx, A.cy, A.cz, pc);
      matvec_emit(q, idx, A.a0 * pc + st, pc, OBJ ? A.objb : nullptr, DOT, dot);
    }
  }
  if (DOT || OBJ) {
    const double tot = block_sum(dot);
    if (tid == 0) A.partials[blockIdx.x] = tot;
  }
}

// x-marching form (round 4).  The five 16-byte loads per vector of k_dtd_flat are cache hits but still pass the
// L2 -> L1 path (tools/mb_stream.hip: 16.9 us for that pattern at 181 x 217 x 181 where a copy takes 8.9).  Here a
// lane owns four consecutive voxels of a PLANE and walks along x: the x neighbours are the previous / next
// plane's own vectors, kept in registers (four plane slots that change roles, the walk unrolled by four); a plane
// costs three 16-byte loads (centre, y - 1, y + 1) and one edge dword, issued two planes ahead.  Planes are
// not multiples of 16 bytes in general (217 x 181 floats), so every access but plane 0's is 4-byte aligned
// only - measured: no penalty.  The 0 .. 3 voxels a plane has beyond its whole vectors go through dtd_at().
struct FlatMArgs {
  FlatArgs F;
  unsigned nx, nvp, ncw, xr, nxr;  // planes; whole vectors per plane; 64-vector chunks per plane; planes per run; runs
  int dbg;                          // UNIRES_FLAT_DBG (measurement only): 1 no edge load, 2 no stencil arithmetic, 4 centre loads only
};

template <bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock) k_dtd_flat_m(FlatMArgs M, const int *__restrict__ done) {
  if (done && *done) return;
  const FlatArgs &A = M.F;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), w = tid >> 6;
  const unsigned n = A.n, nz = A.nz, nynz = A.nynz;
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(A.p, (size_t)n * 4);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(OBJ ? A.objb : A.p, (size_t)n * 4);
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(A.q, (size_t)n * 4);
  double dot = 0.0;
  const unsigned ntasks = M.ncw * M.nxr;
  struct Plane {
    f4 cc, ym, yp, ob;
    float edge;
  };
  // (the last workgroup takes the voxels beyond the planes' whole vectors and nothing else: its element-wise loads
  // run next to the walk instead of behind it)
  const unsigned nwg = gridDim.x - 1u;
  const bool tail_wg = blockIdx.x == nwg;
  const int lb = tail_wg ? 0 : xcd_chunked_block(blockIdx.x, nwg);
  for (unsigned task = tail_wg ? ntasks : (unsigned)lb * (kBlock / kWave) + w; task < ntasks; task += nwg * (kBlock / kWave)) {
    const unsigned r = task / M.ncw, cw = task - r * M.ncw;
    const unsigned v = cw * kWave + lane;
    const bool valid = v < M.nvp;
    const unsigned o = 4u * v;  // in-plane offset of the lane's first voxel
    const unsigned j0 = div_small(o, nz, A.inv_nz), k0 = o - __umul24(j0, nz);
    const unsigned xa = r * M.xr, xb = min(xa + M.xr, M.nx);
    const unsigned eoff = lane == (unsigned)kWave - 1u ? 16u : 0xfffffffcu;  // the voxel above lane 63's last / below lane 0's first
    // does any of the wave's voxels lie on a y face?  (a vector spans at most two lines: j0 and j0 + 1)
    const bool yface = j0 == 0u || j0 + 2u >= A.ny;
    const bool y_inside = __builtin_amdgcn_ballot_w64(yface) == 0ull;
    const unsigned pstep = 4u * nynz;
    // a plane's centre vector is requested TWO steps ahead, its y neighbours / edge voxel one step ahead: by then
    // the lines they share with the centres are on their way or in the cache (requested together with the centre
    // they were three misses on the same lines: 3.8 us for them where tools/mb_stream.hip pays 1.6)
    auto load_rest = [&](unsigned bo, bool first, Plane &P) {  // (planes -1 and nx: offsets outside the array read zeros)
      if (M.dbg & 4) { P.ym = P.cc, P.yp = P.cc, P.ob = P.cc, P.edge = 0.f; return; }
      P.ym = ld4_fast(rp, bo - 4u * nz), P.yp = ld4_fast(rp, bo + 4u * nz);
      // (plane 0: the y - 1 vector of the lane whose four voxels straddle the end of line 0 starts below the
      // array; the hardware adds the components' offsets without 32-bit wrap-around, so its in-range half would
      // read as zeros too)
      if (first && o < nz) P.ym = ld4_safe(rp, (int)o - (int)nz, (int)n);
      P.ob = OBJ ? ld4_fast(rb, bo) : f4{0.f, 0.f, 0.f, 0.f};
      P.edge = (M.dbg & 1) ? 0.f : buf_load(rp, bo + eoff, 0);
    };
    unsigned bo = 4u * (xa * nynz + o);  // byte offset of the lane's vector in the plane being computed
    auto step = [&](const Plane &prev, const Plane &cur, Plane &next, Plane &fly, unsigned vx) {
      // (beyond the run only the centre of its first plane is wanted, as x + 1)
      if (vx + 2u <= xb) fly.cc = ld4_fast(rp, bo + 2u * pstep);
      if (vx + 1u < xb) load_rest(bo + pstep, false, next);
      FlatVec L;
      L.cc = cur.cc, L.xm = prev.cc, L.xp = next.cc, L.ym = cur.ym, L.yp = cur.yp, L.ob = cur.ob, L.edge = cur.edge;
      if (M.dbg & 2) {
        const f4 o = cur.cc + prev.cc + next.cc + cur.ym + cur.yp;
        if (valid) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), rq, bo, 0, kAuxNt);
        bo += pstep;
        return;
      }
      // the common case - no voxel of the wave on an x or y face - carries the z tests only
      if (y_inside && vx > 0u && vx + 1u < M.nx)
        flat_vec<false, DOT, OBJ>(A, L, lane, bo >> 2, k0, j0, valid, rq, dot);
      else
        flat_vec<true, DOT, OBJ>(A, L, lane, bo >> 2, k0, j0, valid, rq, dot);
      bo += pstep;
    };
    Plane P0, P1, P2, P3;
    P0.cc = f4{0.f, 0.f, 0.f, 0.f};
    if (xa > 0u) P0.cc = ld4_fast(rp, bo - pstep);
    P1.cc = ld4_fast(rp, bo);
    load_rest(bo, xa == 0u, P1);
    P2.cc = ld4_fast(rp, bo + pstep);
    for (unsigned vx = xa; vx < xb; vx += 4u) {  // the slots change roles: no register moves
      step(P0, P1, P2, P3, vx);
      if (vx + 1u < xb) step(P1, P2, P3, P0, vx + 1u);
      if (vx + 2u < xb) step(P2, P3, P0, P1, vx + 2u);
      if (vx + 3u < xb) step(P3, P0, P1, P2, vx + 3u);
      else break;
    }
  }
  // the voxels a plane has beyond its whole vectors, and nothing else is left
  if (tail_wg) {
    const unsigned tl = nynz - 4u * M.nvp, ntail = M.nx * tl;
    const Dim3i dd{(int)M.nx, (int)A.ny, (int)nz};
    for (unsigned i = tid; i < ntail; i += kBlock) {
      const unsigned x = i / tl, idx = x * nynz + 4u * M.nvp + (i - x * tl);
      const unsigned line = idx / nz, k = idx - line * nz, j = line - x * A.ny;
      float pc;
      const float st = dtd_at(A.p, idx, (int)x, (int)j, (int)k, dd, A.cx, A.cy, A.cz, pc);
      matvec_emit(A.q, idx, A.a0 * pc + st, pc, OBJ ? A.objb : nullptr, DOT, dot);
    }
  }
  if (DOT || OBJ) {
    const double tot = block_sum(dot);
    if (tid == 0) A.partials[blockIdx.x] = tot;
  }
}

// marching form: geometry of the launch; false: the flat kernel serves the volume
static bool flat_m_geometry(Dim3i dd, FlatMArgs &M) {
  // Measured and left OFF (UNIRES_FLAT_MARCH=-1: on with automatic runs, n: planes per run): at 181 x 217 x 181
  // the walk takes 14.2 - 15.2 us (plain launches, 2048 - 4096 wave tasks; 21 with 1024) where k_dtd_flat takes 12.7
  // in the same harness.  By ablation (UNIRES_FLAT_DBG): the bare walk - centre loads and stores only - 10.8 - 11.2
  // (tools/mb_stream.hip's marching copy: 8.9), y neighbours + edge + the stencil arithmetic 3 us on top: with two
  // to four waves per SIMD a wave's ~86 instructions per plane run at the single-wave issue interval, and shorter
  // runs pay more run-in planes.  (Before the y neighbours moved one step behind the centre loads and the planes'
  // last voxels to a workgroup of their own: 17.5.)
  static const int mode = getenv("UNIRES_FLAT_MARCH") ? atoi(getenv("UNIRES_FLAT_MARCH")) : 0;
  if (mode == 0) return false;
  const unsigned long long nynz = (unsigned long long)dd.y * dd.z;
  if (nynz < 4u * kWave || dd.x < 4 || nynz >= (1u << 24)) return false;
  M.nx = (unsigned)dd.x, M.nvp = (unsigned)(nynz / 4u), M.ncw = (M.nvp + kWave - 1u) / kWave;
  static const int tasks = getenv("UNIRES_FLAT_TASKS") ? atoi(getenv("UNIRES_FLAT_TASKS")) : 2048;
  unsigned long long xr = mode > 0 ? (unsigned long long)mode : ((unsigned long long)M.nx * M.ncw + tasks - 1) / tasks;
  xr = std::max<unsigned long long>(mode > 0 ? 2 : 4, std::min<unsigned long long>(xr, M.nx));
  // (runs a multiple of 1 MB apart keep the concurrently walked planes on the same DRAM banks)
  if (mode <= 0 && xr < M.nx && (xr * nynz * 4u) % (1u << 20) == 0) ++xr;
  M.xr = (unsigned)xr, M.nxr = (M.nx + M.xr - 1u) / M.xr;
  return true;
}

static int flat_m_blocks(const FlatMArgs &M) {
  const unsigned long long nb = ((unsigned long long)M.ncw * M.nxr + (kBlock / kWave) - 1) / (kBlock / kWave);
  return (int)std::min<unsigned long long>(nb, 4096) + 1;  // (+ the workgroup of the planes' last voxels)
}

static int flat_grid(unsigned nchunk) {
  // every workgroup the same number of chunks (a grid capped at 4096 gave 356 of an XCD's 512
  // workgroups two chunks and the others one); UNIRES_FLAT_BLOCKS overrides the cap (<= kMaxPartials)
  static const unsigned cap = getenv("UNIRES_FLAT_BLOCKS") ? (unsigned)atoi(getenv("UNIRES_FLAT_BLOCKS")) : 4096u;
  if (nchunk < 1) return 1;
  const unsigned per = (nchunk + cap - 1) / cap;
  return (int)((nchunk + per - 1) / per);
}

int dtd_flat_blocks(Dim3i dd) {
  FlatMArgs M;
  if (flat_m_geometry(dd, M)) return flat_m_blocks(M);
  const size_t n = dd.numel();
  return flat_grid((unsigned)((n / 4 * 4 + kFlatChunk - 1) / kFlatChunk));
}

// Non-zero return: outside the kernel's domain (tiny or huge volumes), nothing launched.
int launch_dtd_flat(const float *p, float *q, Dim3i dd, float a0, float cx, float cy, float cz,
                    double *partials, const float *objb, const int *done, hipStream_t st) {
  const size_t n = dd.numel();
  if (dd.z < 4 || n >= (1ull << 29) || n < 64 || (objb && !partials)) return 1;
  if ((size_t)dd.z + kFlatChunk >= (1u << 24) || (size_t)dd.y + kFlatChunk >= (1u << 24)) return 1;
  FlatArgs A;
  A.p = p, A.q = q, A.objb = objb, A.partials = partials;
  A.n = (unsigned)n, A.nz = (unsigned)dd.z, A.ny = (unsigned)dd.y, A.nynz = (unsigned)dd.y * (unsigned)dd.z;
  A.inv_nz = 1.f / (float)dd.z, A.inv_ny = 1.f / (float)dd.y;
  A.head = (unsigned)(((16u - (unsigned)((uintptr_t)q & 15u)) & 15u) / 4u);
  A.nvec = (A.n - A.head) / 4u;
  A.nchunk = (A.nvec * 4u + kFlatChunk - 1u) / (unsigned)kFlatChunk;
  A.a0 = a0, A.cx = cx, A.cy = cy, A.cz = cz;
  FlatMArgs M;
  if (flat_m_geometry(dd, M)) {
    M.F = A;
    static const int fdbg = getenv("UNIRES_FLAT_DBG") ? atoi(getenv("UNIRES_FLAT_DBG")) : 0;
    M.dbg = fdbg;
    const dim3 grid(flat_m_blocks(M)), block(kBlock);
    if (objb)
      hipLaunchKernelGGL((k_dtd_flat_m<true, true>), grid, block, 0, st, M, done);
    else if (partials)
      hipLaunchKernelGGL((k_dtd_flat_m<true, false>), grid, block, 0, st, M, done);
    else
      hipLaunchKernelGGL((k_dtd_flat_m<false, false>), grid, block, 0, st, M, done);
    return 0;
  }
  const unsigned G = (unsigned)dtd_flat_blocks(dd);
  A.nx = G >= 8u ? 8u : 1u, A.per_xcd = G / A.nx, A.rem = G % A.nx;
  for (unsigned x = 0, before = 0; x <= A.nx; ++x) {
    A.cb[x] = (unsigned)((unsigned long long)A.nchunk * before / G);
    before += A.per_xcd + (x < A.rem ? 1u : 0u);
  }
  // the number of partials must not depend on q's alignment: callers size their reduction with
  // dtd_flat_blocks(dd)
  const dim3 grid(G), block(kBlock);
  if (objb)
    hipLaunchKernelGGL((k_dtd_flat<true, true>), grid, block, 0, st, A, done);
  else if (partials)
    hipLaunchKernelGGL((k_dtd_flat<true, false>), grid, block, 0, st, A, done);
  else
    hipLaunchKernelGGL((k_dtd_flat<false, false>), grid, block, 0, st, A, done);
  return 0;
}

}  // namespace unires
