// aligned.hip - k_ata_aligned: the whole CG matvec  q = tau AtA p + c DtD p (+ sum p*q)  in ONE
// streaming kernel for grid-aligned observations: the affine from grid to output voxels is
// the identity plus an INTEGER translation (UniRes' single-image / no-motion case: rigid = I,
// mat_yx built from mat_y, unires/_project.py:266-285) and the slice profile acts along z.
//
// Then pull and push are exact shifted copies (all trilinear weights are 0 or 1, the in-FOV
// mask equals the in-bounds test), and AtA acts independently on every z line:
//     xs[k]   = S2(k) * sum_t kz[t] * p[.., 6k + t + oz]          (x-space line, 42 values)
//     (AtA p)[z] = sum_{k : 0 <= z-oz-6k < K} kz[z - oz - 6k] * xs[k]
// One wave owns one output z line: the line is read once into LDS, its x-space line is built
// in LDS (it never goes to HBM), and q is written once with the DtD stencil and the float64
// dot fused.  HBM traffic = read p + write q.
//
// What bounds it: a wave64 VALU instruction occupies its SIMD for 4 clocks, so the chip issues
// 6.1e11 wave-instructions/s; the first version spent 325 VALU instructions per line (35 us of
// issue time for 65 536 lines - its whole run time), mostly per-lane copies of wave-uniform
// index arithmetic.  Here everything uniform across the wave lives in SGPRs (the wave index
// goes through readfirstlane; loads and stores are scalar base + lane offset), everything that
// depends on the lane but not on the line is hoisted out of the line loop, and the LDS line
// carries a zero apron so that no tap needs a bounds check.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "aligned.hpp"

namespace unires {

constexpr int kAlignedMaxTaps = 16;
#ifndef UNIRES_ALIGNED_PF
#define UNIRES_ALIGNED_PF 1
#endif

struct AlignedArgs {
  const float *p;
  float *q;
  Dim3i dd;           // output dims
  int gx, gy, gz;     // grid dims (trimmed taps)
  int xdz;            // x-space z length
  int ox, oy, oz;     // output voxel = grid voxel + o
  int nk, s;          // taps / stride along z
  float kz[kAlignedMaxTaps];
  float se2, so2;     // even/odd slice scaling squared (S(2 scl)), 1,1 = none
  float tau, a0, cx, cy, cz;
  double *partials;
  const float *objb;  // objective mode (see matvec_emit)
  int padl, padr;     // zero apron of the LDS line (floats)
  int wave_floats;    // LDS floats per wave: padl + nz + padr + xdz + 1
  unsigned long long *prof;
};

constexpr int kLinesPerBlock = kBlock / kWave;

#ifdef UNIRES_ALIGNED_PROF  // debug builds: per-phase wall-clock sums (100 MHz s_memrealtime ticks)
#define AP_T(var) const unsigned long long var = wall_clock64()
#define AP_ADD(slot, t0, t1) \
  if (A.prof && lane == 0) atomicAdd(A.prof + (slot), (t1) - (t0))
#else
#define AP_T(var)
#define AP_ADD(slot, t0, t1)
#endif

// NP = ceil(nz / 64) z-passes per line, known at compile time so that ALL global loads of a
// line (its own values and its four x/y neighbours) are issued before anything is consumed.
template <int NP, bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock)
    k_ata_aligned(AlignedArgs A, const int *__restrict__ done) {
  AP_T(t_entry);
  if (done && *done) return;
  extern __shared__ float smem[];
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);  // wave-uniform -> SGPR
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  // block-shared: per-output-z table of the transposed conv, (AtA p)[z] = w0 xs[ko] + w1 xs[ko+1]
  float4 *ztab = reinterpret_cast<float4 *>(smem);
  float *kzs = smem + 4 * NP * kWave;
  float *buf = kzs + kAlignedMaxTaps + w * A.wave_floats;
  float *pl = buf + A.padl;          // pl[-padl, nz + padr): this wave's copy of the p line
  float *xs = pl + nz + A.padr;      // xs[0, xdz]: its x-space line (+ one zero pad)
  if (w == 0 && lane < kAlignedMaxTaps) kzs[lane] = A.kz[lane];
  for (int i = lane; i < A.wave_floats; i += kWave) buf[i] = 0.f;  // aprons stay zero
  __syncthreads();
  for (int z = w * kWave + (int)lane; z < NP * kWave; z += kBlock) {
    const int uz = z - A.oz;
    float w0 = 0.f, w1 = 0.f;
    int koff = 0;
    if (uz >= 0 && uz < A.gz && z < nz) {
      int klo, khi;
      up_range(uz, A.nk, A.s, A.xdz, klo, khi);
      const int n = khi - klo + 1;
      if (n >= 1) w0 = kzs[uz - A.s * klo], koff = klo;
      if (n >= 2) w1 = kzs[uz - A.s * (klo + 1)];
    }
    ztab[z] = make_float4(__int_as_float(koff), w0, w1, 0.f);
  }
  __syncthreads();
  // ---- lane constants (independent of the line) ----
  float ew0[NP], ew1[NP];
  const float *exs[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const float4 tb = ztab[u * kWave + lane];
    ew0[u] = tb.y, ew1[u] = tb.z, exs[u] = xs + __float_as_int(tb.x);  // xs[xdz] is a zero pad
  }

  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int nlines = dd.x * dd.y;
  const size_t sx = (size_t)dd.y * nz, sy = nz;
  double dot = 0.0;
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);  // x/y halo lines stay in one L2
  const int line_step = gridDim.x * kLinesPerBlock;
  AP_T(t_loop);
  AP_ADD(0, t_entry, t_loop);
  for (int line = lb * kLinesPerBlock + w; line < nlines; line += line_step) {
    AP_T(t0);
    // everything down to the loads is scalar (SGPR) arithmetic
    const int vx = line / dd.y, vy = line - vx * dd.y;
    const size_t base = (size_t)line * nz;
    const bool hx = vx + 1 < dd.x, lx = vx > 0, hy = vy + 1 < dd.y, ly = vy > 0;
    const float *pc = p + base;
    const float *pxp = hx ? pc + sx : pc, *pxm = lx ? pc - sx : pc;
    const float *pyp = hy ? pc + sy : pc, *pym = ly ? pc - sy : pc;
    const float *ob = OBJ ? A.objb + base : pc;
    float rc[NP], rxp[NP], rxm[NP], ryp[NP], rym[NP], rb[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const unsigned z = min(u * kWave + lane, (unsigned)nz - 1u);
      rc[u] = pc[z], rxp[u] = pxp[z], rxm[u] = pxm[z], ryp[u] = pyp[z], rym[u] = pym[z];
      if (OBJ) rb[u] = ob[z];
    }
#pragma unroll
    for (int u = 0; u < NP; ++u)
      if (u * kWave + (int)lane < nz) pl[u * kWave + lane] = rc[u];
    asm volatile("" ::: "memory");  // single wave: LDS ops execute in order
    AP_T(t1);
    AP_ADD(1, t0, t1);
    const int ux = vx - A.ox, uy = vy - A.oy;
    const bool has = ux >= 0 && ux < A.gx && uy >= 0 && uy < A.gy;  // wave-uniform
    if (has) {
      for (int k0 = 0; k0 < A.xdz; k0 += kWave) {
        const int k = k0 + (int)lane;
        if (k < A.xdz) {
          const float *in = pl + (k * A.s + A.oz);  // apron: no bounds checks on the taps
          float acc = 0.f;
          for (int t = 0; t < A.nk; ++t) acc = fmaf(kzs[t], in[t], acc);  // taps: LDS broadcast
          xs[k] = acc * ((k & 1) ? A.so2 : A.se2);
        }
      }
      asm volatile("" ::: "memory");
    }
    AP_T(t2);
    AP_ADD(2, t1, t2);
    // All NP outputs are computed BEFORE the first store.  gfx9 counts loads and stores in
    // one in-order-per-kind counter, so after a store the compiler must drain it completely
    // (vmcnt(0)) to be sure of an older load: interleaving stores and load uses cost one
    // store round trip per pass.
    float out[NP];
    auto compute = [&](auto edge_tag, auto has_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value, HAS = decltype(has_tag)::value;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const unsigned z = u * kWave + lane;
        const float c = rc[u];
        float h = 0.f;
        if (HAS) h = ew0[u] * exs[u][0] + ew1[u] * exs[u][1];
        const float vzp = pl[z + 1];            // zero apron at z = nz
        float vzm = pl[(int)z - 1];
        if (u == 0) vzm = lane == 0 ? c : vzm;  // Neumann row at z = 0
        float xf, xb, yf, yb;
        if (EDGE) {
          xf = (hx ? rxp[u] : 0.f) - c, xb = lx ? c - rxm[u] : 0.f;
          yf = (hy ? ryp[u] : 0.f) - c, yb = ly ? c - rym[u] : 0.f;
        } else {
          xf = rxp[u] - c, xb = c - rxm[u], yf = ryp[u] - c, yb = c - rym[u];
        }
        const float zf = vzp - c, zb = c - vzm;
        out[u] = A.tau * h + A.a0 * c + (A.cx * (xb - xf) + A.cy * (yb - yf) + A.cz * (zb - zf));
      }
    };
    const bool edge = !(hx && lx && hy && ly);
    if (!edge && has)
      compute(std::false_type{}, std::true_type{});
    else if (!edge)
      compute(std::false_type{}, std::false_type{});
    else if (has)
      compute(std::true_type{}, std::true_type{});
    else
      compute(std::true_type{}, std::false_type{});
    asm volatile("" ::: "memory");  // the line buffers are reused by the next line
    AP_T(t3);
    AP_ADD(3, t2, t3);
    float *qc = q + base;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const unsigned z = u * kWave + lane;
      if ((int)z < nz) {
        if (OBJ) {
          dot += (double)obj_term(out[u], rb[u], rc[u]);
        } else {
          qc[z] = out[u];
          if (DOT) dot += (double)__fmul_rn(rc[u], out[u]);
        }
      }
    }
    AP_T(t4);
    AP_ADD(4, t3, t4);
    AP_ADD(5, 0ull, 1ull);
  }
  AP_T(t_end);
  AP_ADD(6, t_entry, t_end);
  AP_ADD(7, 0ull, 1ull);
  if (DOT) {
    const double tot = block_sum(dot);
    if (threadIdx.x == 0 && threadIdx.y == 0) A.partials[blockIdx.x] = tot;
  }
}

// ---- 16-byte form (r3): nz a multiple of 4, one z pass of 256 voxels per wave -------------------
// The dword form above spends 5 x NP global loads per 64 outputs and reaches the z neighbours
// through LDS.  Here a lane owns FOUR consecutive voxels of the line: the line and its four x / y
// neighbour lines are one 16-byte load each (five vector-memory instructions per 256 outputs
// instead of twenty), the z neighbours are the lane's own values or the adjacent lane's (DPP wave
// shift), q leaves as one 16-byte store.  LDS only serves the slice-profile part: the line is
// written once (ds_write_b128), 42 lanes build the x-space line, every lane reads its four
// (xs[k], xs[k + 1]) pairs back.
typedef float af4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float a4_lower(float v) {  // lane l gets lane l - 1's value (lane 0: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float a4_upper(float v) {  // lane l gets lane l + 1's value (lane 63: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

#ifndef UNIRES_ALIGNED4_WAVES
#define UNIRES_ALIGNED4_WAVES 6
#endif
template <int NP4, bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock, NP4 == 1 ? UNIRES_ALIGNED4_WAVES : 4)
    k_ata_aligned4(AlignedArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ __align__(16) float smem[];
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  constexpr int ZP = NP4 * 4 * kWave;  // z positions a wave covers
  float4 *ztab = reinterpret_cast<float4 *>(smem);  // per output z: {x-space offset, w0, w1, -}
  float *kzs = smem + 4 * ZP;
  float *buf = kzs + kAlignedMaxTaps + w * A.wave_floats;  // (wave_floats, padl: multiples of 4)
  float *pl = buf + A.padl;
  float *xs = pl + nz + A.padr;
  if (w == 0 && lane < kAlignedMaxTaps) kzs[lane] = A.kz[lane];
  for (int i = lane; i < A.wave_floats; i += kWave) buf[i] = 0.f;  // aprons stay zero
  __syncthreads();
  for (int z = w * kWave + (int)lane; z < ZP; z += kBlock) {
    const int uz = z - A.oz;
    float w0 = 0.f, w1 = 0.f;
    int koff = 0;
    if (uz >= 0 && uz < A.gz && z < nz) {
      int klo, khi;
      up_range(uz, A.nk, A.s, A.xdz, klo, khi);
      const int n = khi - klo + 1;
      if (n >= 1) w0 = kzs[uz - A.s * klo], koff = klo;  // (koff + 1 <= xdz: xs[xdz] is a zero pad)
      if (n >= 2) w1 = kzs[uz - A.s * (klo + 1)];
    }
    ztab[z] = make_float4(__int_as_float(koff), w0, w1, 0.f);
  }
  __syncthreads();
  // ---- lane constants (independent of the line): the conv_up pair of each of the lane's voxels ----
  // (Measured, r3: re-reading them from the LDS table for every line (-DUNIRES_ALIGNED4_LDSTAB) to get
  // from 76 to 64 registers / from 6 to 8 waves per SIMD: 33.6 - 34.2 us instead of 31.0; requesting
  // the x + 1 line one line ahead: 35.2 us - loads and stores share one in-order counter on gfx9, so
  // every wait drains the look-ahead load and the previous store as well.)
#ifndef UNIRES_ALIGNED4_LDSTAB
  float ew0[NP4][4], ew1[NP4][4];
  int exo[NP4][4];
#pragma unroll
  for (int u = 0; u < NP4; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4 tb = ztab[u * 4 * kWave + 4 * lane + e];
      ew0[u][e] = tb.y, ew1[u][e] = tb.z, exo[u][e] = __float_as_int(tb.x);
    }
#endif
  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int nlines = dd.x * dd.y;
  const size_t sx = (size_t)dd.y * nz, sy = nz;
  double dot = 0.0;
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);  // x/y halo lines stay in one L2
  const int line_step = gridDim.x * kLinesPerBlock;
  for (int line = lb * kLinesPerBlock + w; line < nlines; line += line_step) {
    const int vx = line / dd.y, vy = line - vx * dd.y;
    const size_t base = (size_t)line * nz;
    const bool hx = vx + 1 < dd.x, lx = vx > 0, hy = vy + 1 < dd.y, ly = vy > 0;
    const float *pc = p + base;
    const float *pxp = hx ? pc + sx : pc, *pxm = lx ? pc - sx : pc;
    const float *pyp = hy ? pc + sy : pc, *pym = ly ? pc - sy : pc;
    const float *ob = OBJ ? A.objb + base : pc;
    af4 rc[NP4], rxp[NP4], rxm[NP4], ryp[NP4], rym[NP4], rb[NP4];
#pragma unroll
    for (int u = 0; u < NP4; ++u) {
      const int z = u * 4 * kWave + 4 * (int)lane;
      const bool in = z < nz;  // nz % 4 == 0: a vector is inside or outside as a whole
      const int zc = in ? z : 0;
      const af4 zero = {0.f, 0.f, 0.f, 0.f};
      rc[u] = *reinterpret_cast<const af4 *>(pc + zc), rxp[u] = *reinterpret_cast<const af4 *>(pxp + zc);
      rxm[u] = *reinterpret_cast<const af4 *>(pxm + zc), ryp[u] = *reinterpret_cast<const af4 *>(pyp + zc);
      rym[u] = *reinterpret_cast<const af4 *>(pym + zc);
      if (OBJ) rb[u] = *reinterpret_cast<const af4 *>(ob + zc);
      if (!in) rc[u] = zero;  // (the z + 1 neighbour of the line's last voxel reads this zero)
    }
    const int ux = vx - A.ox, uy = vy - A.oy;
    const bool has = ux >= 0 && ux < A.gx && uy >= 0 && uy < A.gy;  // wave-uniform
    if (has) {
#pragma unroll
      for (int u = 0; u < NP4; ++u)
        if (u * 4 * kWave + 4 * (int)lane < nz) *reinterpret_cast<af4 *>(pl + u * 4 * kWave + 4 * lane) = rc[u];
      asm volatile("" ::: "memory");  // single wave: LDS ops execute in order
      for (int k0 = 0; k0 < A.xdz; k0 += kWave) {
        const int k = k0 + (int)lane;
        if (k < A.xdz) {
          const float *in = pl + (k * A.s + A.oz);  // apron: no bounds checks on the taps
          float acc = 0.f;
          for (int t = 0; t < A.nk; ++t) acc = fmaf(kzs[t], in[t], acc);  // taps: LDS broadcast
          xs[k] = acc * ((k & 1) ? A.so2 : A.se2);
        }
      }
      asm volatile("" ::: "memory");
    }
    af4 out[NP4];
    auto compute = [&](auto edge_tag, auto has_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value, HAS = decltype(has_tag)::value;
#pragma unroll
      for (int u = 0; u < NP4; ++u) {
        // z neighbours across lanes: the lane below holds z - 1 in its .w, the lane above z + 4 in its .x;
        // across passes through readlane; the line's first voxel has no backward term (zlo := c)
        float zlo = a4_lower(rc[u].w), zhi = a4_upper(rc[u].x);
        if (u > 0) {
          const float prev = __builtin_amdgcn_readlane(rc[u > 0 ? u - 1 : 0].w, kWave - 1);
          zlo = lane == 0 ? prev : zlo;
        } else {
          zlo = lane == 0 ? rc[0].x : zlo;
        }
        if (u + 1 < NP4) {
          const float next = __builtin_amdgcn_readlane(rc[u + 1 < NP4 ? u + 1 : 0].x, 0);
          zhi = lane == kWave - 1 ? next : zhi;
        }
        const float c4[4] = {rc[u].x, rc[u].y, rc[u].z, rc[u].w};
        const float zm4[4] = {zlo, rc[u].x, rc[u].y, rc[u].z}, zp4[4] = {rc[u].y, rc[u].z, rc[u].w, zhi};
        const float xp4[4] = {rxp[u].x, rxp[u].y, rxp[u].z, rxp[u].w}, xm4[4] = {rxm[u].x, rxm[u].y, rxm[u].z, rxm[u].w};
        const float yp4[4] = {ryp[u].x, ryp[u].y, ryp[u].z, ryp[u].w}, ym4[4] = {rym[u].x, rym[u].y, rym[u].z, rym[u].w};
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float c = c4[e];
          float h = 0.f;
#ifndef UNIRES_ALIGNED4_LDSTAB
          if (HAS) h = ew0[u][e] * xs[exo[u][e]] + ew1[u][e] * xs[exo[u][e] + 1];
#else
          if (HAS) {
            const float4 tb = ztab[u * 4 * kWave + 4 * lane + e];
            const float *xo = xs + __float_as_int(tb.x);
            h = tb.y * xo[0] + tb.z * xo[1];
          }
#endif
          float xf, xb, yf, yb;
          if (EDGE) {
            xf = (hx ? xp4[e] : 0.f) - c, xb = lx ? c - xm4[e] : 0.f;
            yf = (hy ? yp4[e] : 0.f) - c, yb = ly ? c - ym4[e] : 0.f;
          } else {
            xf = xp4[e] - c, xb = c - xm4[e], yf = yp4[e] - c, yb = c - ym4[e];
          }
          const float zf = zp4[e] - c, zb = c - zm4[e];
          o4[e] = A.tau * h + A.a0 * c + (A.cx * (xb - xf) + A.cy * (yb - yf) + A.cz * (zb - zf));
        }
        out[u] = af4{o4[0], o4[1], o4[2], o4[3]};
      }
    };
    const bool edge = !(hx && lx && hy && ly);
    if (!edge && has)
      compute(std::false_type{}, std::true_type{});
    else if (!edge)
      compute(std::false_type{}, std::false_type{});
    else if (has)
      compute(std::true_type{}, std::true_type{});
    else
      compute(std::true_type{}, std::false_type{});
    asm volatile("" ::: "memory");  // the line buffers are reused by the next line
    float *qc = q + base;
#pragma unroll
    for (int u = 0; u < NP4; ++u) {
      const int z = u * 4 * kWave + 4 * (int)lane;
      if (z < nz) {
        if (OBJ) {
          dot += (double)obj_term(out[u].x, rb[u].x, rc[u].x) + (double)obj_term(out[u].y, rb[u].y, rc[u].y) +
                 (double)obj_term(out[u].z, rb[u].z, rc[u].z) + (double)obj_term(out[u].w, rb[u].w, rc[u].w);
        } else {
          // (nt store: 29.7 vs 30.9 us - q is not read again before the next kernel)
          __builtin_nontemporal_store(out[u], reinterpret_cast<af4 *>(qc + z));
          if (DOT)
            dot += (double)__fmul_rn(rc[u].x, out[u].x) + (double)__fmul_rn(rc[u].y, out[u].y) +
                   (double)__fmul_rn(rc[u].z, out[u].z) + (double)__fmul_rn(rc[u].w, out[u].w);
        }
      }
    }
  }
  if (DOT) {
    const double tot = block_sum(dot);
    if (threadIdx.x == 0 && threadIdx.y == 0) A.partials[blockIdx.x] = tot;
  }
}

// ---- two lines per trip (r3): the 16-byte form on PAIRS of y-adjacent lines (ny even, nz <= 256) -------
// The line kernel's time is latency, not work: per line a wave issues its loads, waits, goes through four
// dependent LDS round trips, stores, and - loads and stores sharing one in-order counter - drains that
// store at the next line's first wait.  Two y-adjacent lines per trip halve the trips: eight loads serve
// two lines (each is the other's y neighbour) instead of ten, twice the bytes are in flight per wave, and
// the drain is paid once per pair.
template <bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock) k_ata_aligned4x2(AlignedArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ __align__(16) float smem[];
  const unsigned lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  constexpr int ZP = 4 * kWave;
  float4 *ztab = reinterpret_cast<float4 *>(smem);  // per output z: {x-space offset, w0, w1, -}
  float *kzs = smem + 4 * ZP;
  float *buf = kzs + kAlignedMaxTaps + w * 2 * A.wave_floats;  // two line buffers per wave
  float *pl[2] = {buf + A.padl, buf + A.wave_floats + A.padl};
  float *xs[2] = {pl[0] + nz + A.padr, pl[1] + nz + A.padr};
  if (w == 0 && lane < kAlignedMaxTaps) kzs[lane] = A.kz[lane];
  for (int i = lane; i < 2 * A.wave_floats; i += kWave) buf[i] = 0.f;  // aprons stay zero
  __syncthreads();
  for (int z = w * kWave + (int)lane; z < ZP; z += kBlock) {
    const int uz = z - A.oz;
    float w0 = 0.f, w1 = 0.f;
    int koff = 0;
    if (uz >= 0 && uz < A.gz && z < nz) {
      int klo, khi;
      up_range(uz, A.nk, A.s, A.xdz, klo, khi);
      const int n = khi - klo + 1;
      if (n >= 1) w0 = kzs[uz - A.s * klo], koff = klo;
      if (n >= 2) w1 = kzs[uz - A.s * (klo + 1)];
    }
    ztab[z] = make_float4(__int_as_float(koff), w0, w1, 0.f);
  }
  __syncthreads();
  float ew0[4], ew1[4];
  int exo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float4 tb = ztab[4 * lane + e];
    ew0[e] = tb.y, ew1[e] = tb.z, exo[e] = __float_as_int(tb.x);
  }
  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int hy2 = dd.y / 2, npairs = dd.x * hy2;
  const size_t sx = (size_t)dd.y * nz, sy = nz;
  double dot = 0.0;
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);
  const int pair_step = gridDim.x * kLinesPerBlock;
  const int z0 = 4 * (int)lane;
  const bool in = z0 < nz;
  const int zc = in ? z0 : 0;
  const af4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int pr = lb * kLinesPerBlock + w; pr < npairs; pr += pair_step) {
    const int vx = pr / hy2, vy0 = 2 * (pr - vx * hy2);
    const size_t base = ((size_t)vx * dd.y + vy0) * nz;
    const bool hx = vx + 1 < dd.x, lx = vx > 0, ly = vy0 > 0, hy = vy0 + 2 < dd.y;
    const float *pc = p + base + zc;
    af4 rc[2], rxp[2], rxm[2], ryo[2], rb[2];  // ryo: line 0's y - 1, line 1's y + 1 (the other y neighbour is the partner)
    rc[0] = *reinterpret_cast<const af4 *>(pc), rc[1] = *reinterpret_cast<const af4 *>(pc + sy);
    ryo[0] = *reinterpret_cast<const af4 *>(ly ? pc - sy : pc);
    ryo[1] = *reinterpret_cast<const af4 *>(hy ? pc + 2 * sy : pc);
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      rxm[l] = *reinterpret_cast<const af4 *>(lx ? pc + l * sy - sx : pc);
      rxp[l] = *reinterpret_cast<const af4 *>(hx ? pc + l * sy + sx : pc);
      rb[l] = zero;
      if (OBJ) rb[l] = *reinterpret_cast<const af4 *>(A.objb + base + l * sy + zc);
    }
    if (!in) rc[0] = zero, rc[1] = zero;
    const int ux = vx - A.ox, uy0 = vy0 - A.oy;
    const bool hasx = ux >= 0 && ux < A.gx;
    const bool has[2] = {hasx && uy0 >= 0 && uy0 < A.gy, hasx && uy0 + 1 >= 0 && uy0 + 1 < A.gy};
#pragma unroll
    for (int l = 0; l < 2; ++l)
      if (has[l] && in) *reinterpret_cast<af4 *>(pl[l] + z0) = rc[l];
    asm volatile("" ::: "memory");  // single wave: LDS ops execute in order
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (!has[l]) continue;
      for (int k0 = 0; k0 < A.xdz; k0 += kWave) {
        const int k = k0 + (int)lane;
        if (k < A.xdz) {
          const float *bin = pl[l] + (k * A.s + A.oz);
          float acc = 0.f;
          for (int t = 0; t < A.nk; ++t) acc = fmaf(kzs[t], bin[t], acc);
          xs[l][k] = acc * ((k & 1) ? A.so2 : A.se2);
        }
      }
    }
    asm volatile("" ::: "memory");
    af4 out[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      float zlo = a4_lower(rc[l].w), zhi = a4_upper(rc[l].x);
      zlo = lane == 0 ? rc[l].x : zlo;  // the line's first voxel has no backward term
      const float c4[4] = {rc[l].x, rc[l].y, rc[l].z, rc[l].w};
      const float zm4[4] = {zlo, rc[l].x, rc[l].y, rc[l].z}, zp4[4] = {rc[l].y, rc[l].z, rc[l].w, zhi};
      const float xp4[4] = {rxp[l].x, rxp[l].y, rxp[l].z, rxp[l].w}, xm4[4] = {rxm[l].x, rxm[l].y, rxm[l].z, rxm[l].w};
      const af4 ym = l == 0 ? ryo[0] : rc[0], yp = l == 0 ? rc[1] : ryo[1];
      const float yp4[4] = {yp.x, yp.y, yp.z, yp.w}, ym4[4] = {ym.x, ym.y, ym.z, ym.w};
      const bool lyl = l == 0 ? ly : true, hyl = l == 0 ? true : hy;
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = c4[e];
        float h = 0.f;
        if (has[l]) h = ew0[e] * xs[l][exo[e]] + ew1[e] * xs[l][exo[e] + 1];
        const float xf = (hx ? xp4[e] : 0.f) - c, xb = lx ? c - xm4[e] : 0.f;
        const float yf = (hyl ? yp4[e] : 0.f) - c, yb = lyl ? c - ym4[e] : 0.f;
        const float zf = zp4[e] - c, zb = c - zm4[e];
        o4[e] = A.tau * h + A.a0 * c + (A.cx * (xb - xf) + A.cy * (yb - yf) + A.cz * (zb - zf));
      }
      out[l] = af4{o4[0], o4[1], o4[2], o4[3]};
    }
    asm volatile("" ::: "memory");  // the line buffers are reused by the next pair
    if (in) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        if (OBJ) {
          dot += (double)obj_term(out[l].x, rb[l].x, rc[l].x) + (double)obj_term(out[l].y, rb[l].y, rc[l].y) +
                 (double)obj_term(out[l].z, rb[l].z, rc[l].z) + (double)obj_term(out[l].w, rb[l].w, rc[l].w);
        } else {
          __builtin_nontemporal_store(out[l], reinterpret_cast<af4 *>(q + base + l * sy + z0));
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

int aligned_blocks(Dim3i dd) {
  const long long nb = ((long long)dd.x * dd.y + kLinesPerBlock - 1) / kLinesPerBlock;
  // 4096 workgroups = 4 lines per wave: amortises the per-wave set-up (table, LDS aprons) and
  // still load-balances (measured: 2048 38.8 us, 4096 35.2 us, 8192 39.0 us)
  static const int cap =
      getenv("UNIRES_ALIGNED_BLOCKS") ? atoi(getenv("UNIRES_ALIGNED_BLOCKS")) : 4096;
  const long long lim = cap < kMaxPartials ? cap : kMaxPartials;
  return (int)(nb < lim ? nb : lim);
}

// True iff A is the identity plus an integer translation, bit for bit (then every trilinear
// weight the reference computes is exactly 0 or 1).
bool affine_is_integer_shift(const Affine &A, int off[3]) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      if (A.m[4 * r + c] != (r == c ? 1.f : 0.f)) return false;
    const float t = A.m[4 * r + 3];
    if (t != floorf(t) || fabsf(t) > 1e6f) return false;
    off[r] = (int)t;
  }
  return true;
}

// Launches the line kernel for filled-in arguments (npt = 2, 4, 6 or 8 z-passes per line).
static void launch_lines(AlignedArgs &G, int npt, size_t lds, const int *done, hipStream_t st) {
  const Dim3i dd = G.dd;
  const double *partials = G.partials;
  const float *objb = G.objb;
  const dim3 grid(aligned_blocks(dd)), block(kWave, kLinesPerBlock);
  // 16-byte form: lines that are whole vectors (nz % 4 == 0, 16-byte aligned volumes), <= 512 long
  static const bool no_v4 = getenv("UNIRES_ALIGNED_V4") && getenv("UNIRES_ALIGNED_V4")[0] == '0';
  const uintptr_t al = (uintptr_t)G.p | (uintptr_t)G.q | (uintptr_t)(objb ? objb : G.p);
  if (!no_v4 && dd.z % 4 == 0 && dd.z <= 8 * kWave && (al & 15u) == 0) {
    AlignedArgs V = G;
    const int np4 = dd.z <= 4 * kWave ? 1 : 2;
    V.padl = (G.padl + 3) & ~3;
    V.wave_floats = (V.padl + dd.z + G.padr + G.xdz + 1 + 3) & ~3;
    const size_t lds4 = ((size_t)4 * np4 * 4 * kWave + kAlignedMaxTaps + (size_t)kLinesPerBlock * V.wave_floats) *
                        sizeof(float);
    static const bool no_x2 = getenv("UNIRES_ALIGNED_X2") && getenv("UNIRES_ALIGNED_X2")[0] == '0';
    const size_t lds42 = ((size_t)4 * 4 * kWave + kAlignedMaxTaps + (size_t)2 * kLinesPerBlock * V.wave_floats) * sizeof(float);
    if (!no_x2 && np4 == 1 && dd.y % 2 == 0 && lds42 <= 64 * 1024) {
      if (objb)
        hipLaunchKernelGGL((k_ata_aligned4x2<true, true>), grid, block, lds42, st, V, done);
      else if (partials)
        hipLaunchKernelGGL((k_ata_aligned4x2<true, false>), grid, block, lds42, st, V, done);
      else
        hipLaunchKernelGGL((k_ata_aligned4x2<false, false>), grid, block, lds42, st, V, done);
      return;
    }
    if (lds4 <= 64 * 1024) {
#define LAUNCH_ALIGNED4(NPV)                                                                    \
  do {                                                                                          \
    if (objb)                                                                                   \
      hipLaunchKernelGGL((k_ata_aligned4<NPV, true, true>), grid, block, lds4, st, V, done);    \
    else if (partials)                                                                          \
      hipLaunchKernelGGL((k_ata_aligned4<NPV, true, false>), grid, block, lds4, st, V, done);   \
    else                                                                                        \
      hipLaunchKernelGGL((k_ata_aligned4<NPV, false, false>), grid, block, lds4, st, V, done);  \
  } while (0)
      if (np4 == 1)
        LAUNCH_ALIGNED4(1);
      else
        LAUNCH_ALIGNED4(2);
#undef LAUNCH_ALIGNED4
      return;
    }
  }
#define LAUNCH_ALIGNED(NPV)                                                                  \
  do {                                                                                       \
    if (objb)                                                                                \
      hipLaunchKernelGGL((k_ata_aligned<NPV, true, true>), grid, block, lds, st, G, done);   \
    else if (partials)                                                                       \
      hipLaunchKernelGGL((k_ata_aligned<NPV, true, false>), grid, block, lds, st, G, done);  \
    else                                                                                     \
      hipLaunchKernelGGL((k_ata_aligned<NPV, false, false>), grid, block, lds, st, G, done); \
  } while (0)
  if (npt == 2)
    LAUNCH_ALIGNED(2);
  else if (npt == 4)
    LAUNCH_ALIGNED(4);
  else if (npt == 6)
    LAUNCH_ALIGNED(6);
  else
    LAUNCH_ALIGNED(8);
#undef LAUNCH_ALIGNED
}

// Regime A = I: q = a0 p + c DtD p (+ dot / objective) with the same line kernel (no AtA term).
// Non-zero return: lines too long for the kernel's LDS budget, nothing launched.
int launch_dtd_lines(const float *p, float *q, Dim3i dd, float a0, float cx, float cy, float cz,
                     double *partials, const float *objb, const int *done, hipStream_t st) {
  if (dd.z > 8 * kWave || (objb && !partials)) return 1;
  const int np = (dd.z + kWave - 1) / kWave;
  const int npt = np <= 2 ? 2 : (np <= 4 ? 4 : (np <= 6 ? 6 : 8));
  AlignedArgs G;
  G.p = p, G.q = q, G.dd = dd;
  G.gx = G.gy = G.gz = 0;  // no voxel belongs to a grid: the AtA branch is never taken
  G.xdz = 1, G.ox = G.oy = G.oz = 0, G.nk = 1, G.s = 1;
  for (int i = 0; i < kAlignedMaxTaps; ++i) G.kz[i] = 0.f;
  G.se2 = G.so2 = 1.f;
  G.tau = 0.f, G.a0 = a0, G.cx = cx, G.cy = cy, G.cz = cz;
  G.partials = partials, G.objb = objb;
  G.padl = G.padr = 1;
  G.wave_floats = 1 + dd.z + 1 + 2;
  G.prof = nullptr;
  const size_t lds =
      ((size_t)4 * npt * kWave + kAlignedMaxTaps + (size_t)kLinesPerBlock * G.wave_floats) * sizeof(float);
  launch_lines(G, npt, lds, done, st);
  return 0;
}

// Non-zero return (nothing launched): outside this kernel's domain.
int launch_ata_aligned(const float *p, float *q, Dim3i dd, Dim3i gd, Dim3i xd, const Taps &T,
                       const Scaling &S2, const Affine &A, float tau, float a0, float cx,
                       float cy, float cz, double *partials, const float *objb, const int *done,
                       hipStream_t st) {
  int off[3];
  if (!affine_is_integer_shift(A, off)) return 1;
  for (int d = 0; d < 2; ++d)
    if (T.n[d] != 1 || T.s[d] != 1 || T.t[d][0] != 1.f) return 1;
  if (S2.dim >= 0 && S2.dim != 2) return 1;
  if (xd.z < 2 || (T.n[2] + T.s[2] - 1) / T.s[2] > 2) return 1;  // fan-in <= 2 (rect profiles)
  if (T.n[2] > kAlignedMaxTaps || dd.z > 8 * kWave) return 1;
  if (objb && !partials) return 1;
  // taps of slice k read output voxels [k s + oz, k s + oz + nk): zero apron for the part
  // outside the line (bound 'zero'; in this geometry the in-FOV mask is the in-bounds test)
  const int lo = off[2], hi = (xd.z - 1) * T.s[2] + off[2] + T.n[2] - 1;
  const int padl = lo < -1 ? -lo : 1, padr = hi > dd.z ? hi - dd.z + 1 : 1;
  if (padl > 256 || padr > 256) return 1;
  const int np = (dd.z + kWave - 1) / kWave;
  const int npt = np <= 2 ? 2 : (np <= 4 ? 4 : (np <= 6 ? 6 : 8));
  AlignedArgs G;
  G.padl = padl, G.padr = padr;
  G.wave_floats = padl + dd.z + padr + xd.z + 1;
  const size_t lds =
      ((size_t)4 * npt * kWave + kAlignedMaxTaps + (size_t)kLinesPerBlock * G.wave_floats) * sizeof(float);
  if (lds > 64 * 1024) return 1;
  G.p = p, G.q = q, G.dd = dd;
  G.gx = gd.x, G.gy = gd.y, G.gz = gd.z;
  G.xdz = xd.z;
  G.ox = off[0], G.oy = off[1], G.oz = off[2];
  G.nk = T.n[2], G.s = T.s[2];
  for (int i = 0; i < kAlignedMaxTaps; ++i) G.kz[i] = T.t[2][i];
  G.se2 = S2.dim == 2 ? S2.e : 1.f, G.so2 = S2.dim == 2 ? S2.o : 1.f;
  G.tau = tau, G.a0 = a0, G.cx = cx, G.cy = cy, G.cz = cz;
  G.partials = partials;
  G.objb = objb;
  G.prof = nullptr;
#ifdef UNIRES_ALIGNED_PROF
  static unsigned long long *prof = nullptr;
  if (!prof) (void)hipMalloc((void **)&prof, 8 * sizeof(unsigned long long));
  (void)hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), st);
  G.prof = prof;
#endif
  launch_lines(G, npt, lds, done, st);
#ifdef UNIRES_ALIGNED_PROF
  {
    unsigned long long h[8];
    (void)hipMemcpyAsync(h, prof, sizeof(h), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    const double nl = h[5] ? (double)h[5] : 1.0, nw = h[7] ? (double)h[7] : 1.0;
    fprintf(stderr, "[aligned prof] waves %llu lines %llu | ns per wave: setup %.0f total %.0f | ns per line: "
            "load+lds %.0f  xs %.0f  compute %.0f  store %.0f\n", h[7], h[5], 10 * h[0] / nw, 10 * h[6] / nw,
            10 * h[1] / nl, 10 * h[2] / nl, 10 * h[3] / nl, 10 * h[4] / nl);
  }
#endif
  return 0;
}

}  // namespace unires
