// common.hpp - shared device helpers for libunires_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/unires_hip.h"

namespace unires {

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kBlock = 256;         // 4 waves, one per SIMD
constexpr int kMaxPartials = 8192;  // blocks of a dot-producing kernel (<= this)

struct Dim3i {
  int x, y, z;
  __host__ __device__ size_t numel() const { return (size_t)x * y * z; }
};

struct Affine {  // row-major 3x4, grid voxel -> source voxel coordinates
  float m[12];
};

struct Taps {  // separable slice-profile kernel + stride
  float t[3][UNIRES_MAX_TAPS];
  int n[3];
  int s[3];
};

struct Scaling {  // _apply_scaling: even slices * e, odd slices * o along dim
  float e, o;
  int dim;  // -1: no scaling
};

// ---- wave / block reductions (float64) ----------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Sum over a kBlock-thread block (1-D or (64,4) shaped); result valid in thread (0,0).
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double s_part[kBlock / kWave];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  v = wave_sum(v);
  __syncthreads();  // protect s_part reuse across successive calls
  if (lane == 0) s_part[wave] = v;
  __syncthreads();
  double tot = 0.0;
  if (tid == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) tot += s_part[w];
  }
  return tot;
}

// Affine coordinate of grid voxel (i,j,k).  The reference builds the dense grid
// in float32 as lin @ ijk + off (unires/_project.py:159 -> nitorch affine_grid).
__device__ __forceinline__ void affine_point(const Affine &A, float i, float j, float k, float &gx,
                                             float &gy, float &gz) {
  gx = fmaf(A.m[2], k, fmaf(A.m[1], j, A.m[0] * i)) + A.m[3];
  gy = fmaf(A.m[6], k, fmaf(A.m[5], j, A.m[4] * i)) + A.m[7];
  gz = fmaf(A.m[10], k, fmaf(A.m[9], j, A.m[8] * i)) + A.m[11];
}

__device__ __forceinline__ bool in_fov(float gx, float gy, float gz, const Dim3i &d, float tol) {
  return gx > -tol && gx < (float)(d.x - 1) + tol && gy > -tol && gy < (float)(d.y - 1) + tol &&
         gz > -tol && gz < (float)(d.z - 1) + tol;
}

// Trilinear sample of src at voxel coordinate (gx,gy,gz): zero bound + in-FOV mask
// (nitorch grid_pull linear / zero / extrapolate=False).
// Split in two so callers can issue the loads of SEVERAL samples before consuming any
// (these kernels are latency-bound unless many loads are in flight per lane).
//   pull_issue : 4 unconditional 8-byte loads (z-adjacent corner pairs, addresses
//                clamped into the volume)
//   pull_finish: weights (zero for out-of-bound corners / out-of-FOV samples) and sum
struct PullLoads {
  float2 v00, v01, v10, v11;  // (x0,y0) (x0,y1) (x1,y0) (x1,y1) rows, z pair
  float wx0, wx1, wy0, wy1, wz0, wz1;
};

__device__ __forceinline__ float2 ld_pair(const float *p) {
  // 4-byte aligned 8-byte load (gfx950 global memory allows unaligned dwordx2)
  float2 v;
  __builtin_memcpy(&v, p, sizeof(v));
  return v;
}

__device__ __forceinline__ void pull_issue(const float *__restrict__ src, const Dim3i &sd,
                                           float gx, float gy, float gz, float tol,
                                           PullLoads &L) {
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  float wx1 = gx - fx, wy1 = gy - fy, wz1 = gz - fz;
  float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
  const float m = in_fov(gx, gy, gz, sd, tol) ? 1.f : 0.f;
  L.wx0 = (ix >= 0 && ix < sd.x) ? wx0 * m : 0.f;
  L.wx1 = (ix + 1 >= 0 && ix + 1 < sd.x) ? wx1 * m : 0.f;
  L.wy0 = (iy >= 0 && iy < sd.y) ? wy0 : 0.f;
  L.wy1 = (iy + 1 >= 0 && iy + 1 < sd.y) ? wy1 : 0.f;
  // z pair (b, b+1) with b clamped into [0, nz-2]; remap the weights onto the pair
  const int b = min(max(iz, 0), max(sd.z - 2, 0));
  wz0 = (iz >= 0 && iz < sd.z) ? wz0 : 0.f;
  wz1 = (iz + 1 >= 0 && iz + 1 < sd.z) ? wz1 : 0.f;
  // pair.x holds index b, pair.y index b+1:  iz==b -> (wz0,wz1); iz==b-1 -> z1 is b -> (wz1,0);
  // iz==b+1 -> z0 is b+1 -> (0,wz0); anything else has both weights zero already
  L.wz0 = iz == b ? wz0 : (iz == b - 1 ? wz1 : 0.f);
  L.wz1 = iz == b ? wz1 : (iz == b + 1 ? wz0 : 0.f);
  const int cx0 = min(max(ix, 0), sd.x - 1), cx1 = min(max(ix + 1, 0), sd.x - 1);
  const int cy0 = min(max(iy, 0), sd.y - 1), cy1 = min(max(iy + 1, 0), sd.y - 1);
  const float *r00 = src + ((size_t)cx0 * sd.y + cy0) * sd.z + b;
  const float *r01 = src + ((size_t)cx0 * sd.y + cy1) * sd.z + b;
  const float *r10 = src + ((size_t)cx1 * sd.y + cy0) * sd.z + b;
  const float *r11 = src + ((size_t)cx1 * sd.y + cy1) * sd.z + b;
  if (sd.z >= 2) {
    L.v00 = ld_pair(r00), L.v01 = ld_pair(r01), L.v10 = ld_pair(r10), L.v11 = ld_pair(r11);
  } else {  // single-slice volume: no pair to load
    L.v00 = make_float2(*r00, 0.f), L.v01 = make_float2(*r01, 0.f);
    L.v10 = make_float2(*r10, 0.f), L.v11 = make_float2(*r11, 0.f);
  }
}

__device__ __forceinline__ float pull_finish(const PullLoads &L) {
  float acc = L.v00.x * (L.wx0 * L.wy0 * L.wz0);
  acc += L.v00.y * (L.wx0 * L.wy0 * L.wz1);
  acc += L.v01.x * (L.wx0 * L.wy1 * L.wz0);
  acc += L.v01.y * (L.wx0 * L.wy1 * L.wz1);
  acc += L.v10.x * (L.wx1 * L.wy0 * L.wz0);
  acc += L.v10.y * (L.wx1 * L.wy0 * L.wz1);
  acc += L.v11.x * (L.wx1 * L.wy1 * L.wz0);
  acc += L.v11.y * (L.wx1 * L.wy1 * L.wz1);
  return acc;
}

// Spatial gradient of the trilinear sample w.r.t. the voxel coordinate (nitorch grid_grad,
// linear / zero bound / extrapolate=False: out-of-volume corners count as zeros, the in-FOV
// mask multiplies the result; call site unires/_update.py:508).  d/dx of sum v wx wy wz is
// sum v (+-1) wy wz with the same validity rules as the weights.
__device__ __forceinline__ void pull_grad_sample(const float *__restrict__ src, const Dim3i &sd,
                                                 float gx, float gy, float gz, float tol,
                                                 float &dx, float &dy, float &dz) {
  PullLoads L;
  pull_issue(src, sd, gx, gy, gz, tol, L);  // L.wx* carry the FOV mask
  const int ix = (int)floorf(gx), iy = (int)floorf(gy), iz = (int)floorf(gz);
  const float m = in_fov(gx, gy, gz, sd, tol) ? 1.f : 0.f;
  const float ax0 = (ix >= 0 && ix < sd.x) ? -m : 0.f, ax1 = (ix + 1 >= 0 && ix + 1 < sd.x) ? m : 0.f;
  const float ay0 = (iy >= 0 && iy < sd.y) ? -1.f : 0.f, ay1 = (iy + 1 >= 0 && iy + 1 < sd.y) ? 1.f : 0.f;
  const float bz0 = (iz >= 0 && iz < sd.z) ? -1.f : 0.f, bz1 = (iz + 1 >= 0 && iz + 1 < sd.z) ? 1.f : 0.f;
  const int b = min(max(iz, 0), max(sd.z - 2, 0));  // same pair remap as pull_issue
  const float az0 = iz == b ? bz0 : (iz == b - 1 ? bz1 : 0.f);
  const float az1 = iz == b ? bz1 : (iz == b + 1 ? bz0 : 0.f);
  // z-interpolated / z-differentiated values of the four (x,y) rows
  const float s00 = L.v00.x * L.wz0 + L.v00.y * L.wz1, s01 = L.v01.x * L.wz0 + L.v01.y * L.wz1;
  const float s10 = L.v10.x * L.wz0 + L.v10.y * L.wz1, s11 = L.v11.x * L.wz0 + L.v11.y * L.wz1;
  const float d00 = L.v00.x * az0 + L.v00.y * az1, d01 = L.v01.x * az0 + L.v01.y * az1;
  const float d10 = L.v10.x * az0 + L.v10.y * az1, d11 = L.v11.x * az0 + L.v11.y * az1;
  dx = ax0 * (s00 * L.wy0 + s01 * L.wy1) + ax1 * (s10 * L.wy0 + s11 * L.wy1);
  dy = L.wx0 * (s00 * ay0 + s01 * ay1) + L.wx1 * (s10 * ay0 + s11 * ay1);
  dz = L.wx0 * (d00 * L.wy0 + d01 * L.wy1) + L.wx1 * (d10 * L.wy0 + d11 * L.wy1);
}

// 32-bit / 24-bit index arithmetic of the fast paths (pull_interior, the splat's x-space
// gathers) is exact for volumes below these sizes; larger ones take the size_t paths.
__host__ __device__ __forceinline__ bool fits_fast_index(const Dim3i &d) {
  return (long long)d.x * d.y < (1ll << 24) && d.z < (1 << 24) && d.numel() < (1ull << 31);
}

// Interior fast path: all 8 corners inside the volume (so also inside the FOV): no
// clamps, no masks, 32-bit offsets, lerp form (~46 VALU per sample vs ~140).
__device__ __forceinline__ float2 ld2_u(const float *p) {
  return *reinterpret_cast<const float2 *>(p);  // gfx950 global loads may be 4-byte aligned
}

__device__ __forceinline__ float pull_interior(const float *__restrict__ src, unsigned ny,
                                               unsigned nz, unsigned nynz, float gx, float gy,
                                               float gz) {
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  const float wx = gx - fx, wy = gy - fy, wz = gz - fz;
  const unsigned ix = (unsigned)(int)fx, iy = (unsigned)(int)fy, iz = (unsigned)(int)fz;
  const unsigned off = __umul24(__umul24(ix, ny) + iy, nz) + iz;
  const float2 a = ld2_u(src + off), b = ld2_u(src + off + nz), c = ld2_u(src + off + nynz),
               d = ld2_u(src + off + nynz + nz);
  const float a1 = fmaf(wz, a.y - a.x, a.x), b1 = fmaf(wz, b.y - b.x, b.x),
              c1 = fmaf(wz, c.y - c.x, c.x), d1 = fmaf(wz, d.y - d.x, d.x);
  const float ab = fmaf(wy, b1 - a1, a1), cd = fmaf(wy, d1 - c1, c1);
  return fmaf(wx, cd - ab, ab);
}

// Canonical coordinate arithmetic split so the row part can be hoisted:
// affine_point(A,i,j,k) == affine_row(A,i,j) then affine_along(A,row,k), bit for bit.
struct RowBase {
  float x, y, z;
};
__device__ __forceinline__ RowBase affine_row(const Affine &A, float i, float j) {
  return RowBase{fmaf(A.m[1], j, A.m[0] * i), fmaf(A.m[5], j, A.m[4] * i),
                 fmaf(A.m[9], j, A.m[8] * i)};
}
__device__ __forceinline__ void affine_along(const Affine &A, const RowBase &r, float k, float &gx,
                                             float &gy, float &gz) {
  gx = fmaf(A.m[2], k, r.x) + A.m[3];
  gy = fmaf(A.m[6], k, r.y) + A.m[7];
  gz = fmaf(A.m[10], k, r.z) + A.m[11];
}

// True iff the image of the grid box [lo, hi] (inclusive corners) lies inside
// [margin, n-1-margin]^3, i.e. every sample of the box has all 8 corners in the volume.
// Affine maps are convex: testing the 8 box corners suffices.  One corner per lane,
// combined by the caller (block- or wave-wide AND).
__device__ __forceinline__ bool corner_inside(const Affine &A, int c, const int lo[3],
                                              const int hi[3], const Dim3i &sd) {
  float gx, gy, gz;
  affine_point(A, (float)((c & 4) ? hi[0] : lo[0]), (float)((c & 2) ? hi[1] : lo[1]),
               (float)((c & 1) ? hi[2] : lo[2]), gx, gy, gz);
  const float m = 0.01f;
  return gx >= m && gx <= (float)(sd.x - 1) - m && gy >= m && gy <= (float)(sd.y - 1) - m &&
         gz >= m && gz <= (float)(sd.z - 1) - m;
}

__device__ __forceinline__ float pull_sample(const float *__restrict__ src, const Dim3i &sd,
                                             float gx, float gy, float gz, float tol) {
  PullLoads L;
  pull_issue(src, sd, gx, gy, gz, tol, L);
  return pull_finish(L);
}

// Buffer addressing (buffer_load/store_dword v, voffset, s[rsrc], soffset): a 128-bit resource
// in SGPRs + a scalar byte offset + a 32-bit per-lane byte offset.  Wave-uniform row/tile bases
// go into soffset and cost no VALU instruction (global_load needs a 64-bit VGPR address unless
// the compiler can prove the scalar-base form, which it loses across loop back-edges).
// Volumes must be < 4 GB; reads beyond num_records return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff,
                                          unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each with its own
// L2.  For kernels whose neighbouring blocks share halo lines, renumber so that every XCD
// works on ONE contiguous chunk of the volume: halos are then re-read from that XCD's L2
// instead of crossing the fabric (measured on k_ata_aligned: fetch 98 MB -> see DESIGN 4).
// (r6: for ANY grid size - XCD x takes nb / 8 blocks, one more if x < nb % 8.  The first form did this for multiples
// of 8 only and left every other grid dealt round-robin: config 4's pull, 28 518 workgroups, had each window column
// fetched by every XCD that met it - 643 MB for a 226 MB volume.)
__device__ __forceinline__ int xcd_chunked_block(int b, int nb) {
  const int x = b & 7, r = nb & 7;
  return x * (nb >> 3) + (x < r ? x : r) + (b >> 3);
}

// One term of the CG objective sum x (Ax - 2b): A(x).sub_(2*b).mul_(x), nitorch
// optim.py cg() stop='max_gain' (rounded like the reference's three elementwise ops).
__device__ __forceinline__ float obj_term(float ax, float b, float x) {
  return __fmul_rn(__fsub_rn(ax, __fmul_rn(2.f, b)), x);
}

// Epilogue of every matvec kernel.  Normal mode: store q, partial += p*q.  Objective mode
// (objb != nullptr): partial += (q - 2 b) * p and q is NOT stored - the CG objective needs
// A(x) only inside this sum, so the vector never goes to HBM.
__device__ __forceinline__ void matvec_emit(float *__restrict__ dst, size_t idx, float q, float pc,
                                            const float *__restrict__ objb, bool want_dot,
                                            double &dot) {
  if (objb) {
    dot += (double)obj_term(q, objb[idx], pc);
  } else {
    dst[idx] = q;
    if (want_dot) dot += (double)__fmul_rn(pc, q);
  }
}

// (c . DtD p)[i,j,k] with per-axis weights cx,cy,cz = c/vx^2: forward differences, zero
// bound -> rows [1,-1] at 0, [-1,2,-1] inside, [-1,2] at n-1.  Loads are unconditional
// (clamped addresses) so the 7 of them issue together; `pc` returns the centre value.
__device__ __forceinline__ float dtd_at(const float *__restrict__ p, size_t idx, int i, int j,
                                        int k, const Dim3i &d, float cx, float cy, float cz,
                                        float &pc) {
  const size_t sx = (size_t)d.y * d.z, sy = d.z;
  const bool hx = i + 1 < d.x, lx = i > 0, hy = j + 1 < d.y, ly = j > 0, hz = k + 1 < d.z,
             lz = k > 0;
  const float c = p[idx];
  const float vxp = p[hx ? idx + sx : idx], vxm = p[lx ? idx - sx : idx];
  const float vyp = p[hy ? idx + sy : idx], vym = p[ly ? idx - sy : idx];
  const float vzp = p[hz ? idx + 1 : idx], vzm = p[lz ? idx - 1 : idx];
  const float xf = (hx ? vxp : 0.f) - c, xb = lx ? c - vxm : 0.f;
  const float yf = (hy ? vyp : 0.f) - c, yb = ly ? c - vym : 0.f;
  const float zf = (hz ? vzp : 0.f) - c, zb = lz ? c - vzm : 0.f;
  pc = c;
  return cx * (xb - xf) + cy * (yb - yf) + cz * (zb - zf);
}

// conv_transpose index range: all k with 0 <= u - r*k < K and 0 <= k < n
__device__ __forceinline__ void up_range(int u, int K, int r, int n, int &lo, int &hi) {
  hi = u / r;
  if (hi > n - 1) hi = n - 1;
  const int t = u - K + 1;
  lo = t <= 0 ? 0 : (t + r - 1) / r;
}

// Same range with the two integer divisions done as float multiply + fix-up (exact for
// u < 2^23; an integer division by a run-time stride costs ~20 VALU instructions, this ~6).
__device__ __forceinline__ int div_exact(int v, int r, float inv_r) {
  int q = (int)(((float)v + 0.5f) * inv_r);
  if (q * r > v) --q;
  if ((q + 1) * r <= v) ++q;
  return q;
}
__device__ __forceinline__ void up_range_f(int u, int K, int r, float inv_r, int n, int &lo, int &hi) {
  hi = min(div_exact(u, r, inv_r), n - 1);
  const int t = u - K + 1;
  lo = t <= 0 ? 0 : div_exact(t + r - 1, r, inv_r);
}

// conv_up gather: h[u] = sum_k ker[u - r k] * S(k) * xs[k]   (F.conv_transpose3d)
__device__ __forceinline__ float conv_up_sample(const float *__restrict__ xs, const Dim3i &xd,
                                                const Taps &T, const Scaling &S, int ux, int uy,
                                                int uz) {
  int ilo, ihi, jlo, jhi, klo, khi;
  up_range(ux, T.n[0], T.s[0], xd.x, ilo, ihi);
  up_range(uy, T.n[1], T.s[1], xd.y, jlo, jhi);
  up_range(uz, T.n[2], T.s[2], xd.z, klo, khi);
  float acc = 0.f;
  for (int i = ilo; i <= ihi; ++i) {
    const float wi = T.t[0][ux - T.s[0] * i];
    for (int j = jlo; j <= jhi; ++j) {
      const float wij = wi * T.t[1][uy - T.s[1] * j];
      const float *row = xs + ((size_t)i * xd.y + j) * xd.z;
      for (int k = klo; k <= khi; ++k) {
        float v = row[k] * (wij * T.t[2][uz - T.s[2] * k]);
        if (S.dim >= 0) {
          const int par = (S.dim == 0 ? i : (S.dim == 1 ? j : k)) & 1;
          v *= par ? S.o : S.e;
        }
        acc += v;
      }
    }
  }
  return acc;
}

}  // namespace unires
