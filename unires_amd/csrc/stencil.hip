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
#include "stencil.hpp"

namespace unires {

constexpr int kFlatVecs = 2;                           // float4 per thread and chunk
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
};

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4_fast(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
// element-wise form: every dword range-checked on its own (offsets that wrapped below zero or run
// past the end read as 0)
__device__ __forceinline__ f4 ld4_safe(__amdgpu_buffer_rsrc_t r, unsigned off) {
  f4 v;
  v.x = buf_load(r, off, 0), v.y = buf_load(r, off + 4u, 0), v.z = buf_load(r, off + 8u, 0),
  v.w = buf_load(r, off + 12u, 0);
  return v;
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

template <bool DOT, bool OBJ>
__global__ void __launch_bounds__(kBlock) k_dtd_flat(FlatArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1);
  const unsigned n = A.n, nz = A.nz, ny = A.ny, nynz = A.nynz;
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(A.p, (size_t)n * 4);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(OBJ ? A.objb : A.p, (size_t)n * 4);
  float *__restrict__ q = A.q;
  double dot = 0.0;
  // every XCD (workgroup b sits on XCD b % 8) walks one contiguous range of chunks: the x / y halo
  // lines of a chunk are then in that XCD's L2 already, or will be used from it next
  const unsigned G = gridDim.x, nx = G < 8u ? G : 8u;
  const unsigned xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
  const unsigned cnt = (G - xcd + nx - 1u) / nx;                 // workgroups of this XCD
  const unsigned before = xcd * (G / nx) + (xcd < G % nx ? xcd : G % nx);  // workgroups of the XCDs below
  const unsigned c_lo = (unsigned)((unsigned long long)A.nchunk * before / G),
                 c_hi = (unsigned)((unsigned long long)A.nchunk * (before + cnt) / G);
  for (unsigned c = c_lo + slot; c < c_hi; c += cnt) {
    const unsigned e0 = A.head + c * (unsigned)kFlatChunk;       // first voxel of the chunk (wave-uniform)
    const unsigned line0 = e0 / nz, kb = e0 - line0 * nz, jb = line0 % ny;
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
      f4 cc, xm, xp, ym, yp, ob = {0.f, 0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(!inner) == 0ull) {
        cc = ld4_fast(rp, bo), ym = ld4_fast(rp, bo - 4u * nz), yp = ld4_fast(rp, bo + 4u * nz);
        xm = ld4_fast(rp, bo - 4u * nynz), xp = ld4_fast(rp, bo + 4u * nynz);
        if (OBJ) ob = ld4_fast(rb, bo);
      } else {
        cc = ld4_safe(rp, bo), ym = ld4_safe(rp, bo - 4u * nz), yp = ld4_safe(rp, bo + 4u * nz);
        xm = ld4_safe(rp, bo - 4u * nynz), xp = ld4_safe(rp, bo + 4u * nynz);
        if (OBJ) ob = ld4_safe(rb, bo);
      }
      float edge = 0.f;  // lane 0: the voxel below its first, lane 63: the voxel above its last
      if (lane == 0u || lane == (unsigned)kWave - 1u) edge = buf_load(rp, lane == 0u ? bo - 4u : bo + 16u, 0);
      float zlo = dpp_from_lower_lane(cc.w), zhi = dpp_from_upper_lane(cc.x);
      zlo = lane == 0u ? edge : zlo;
      zhi = lane == (unsigned)kWave - 1u ? edge : zhi;
      // faces.  A line ends at most once inside the lane's four voxels (nz >= 4): after voxel w.
      const unsigned w = nz - 1u - k0;
      const float c4[4] = {cc.x, cc.y, cc.z, cc.w};
      const float zm4[4] = {zlo, cc.x, cc.y, cc.z}, zp4[4] = {cc.y, cc.z, cc.w, zhi};
      const float xm4[4] = {xm.x, xm.y, xm.z, xm.w}, xp4[4] = {xp.x, xp.y, xp.z, xp.w};
      const float ym4[4] = {ym.x, ym.y, ym.z, ym.w}, yp4[4] = {yp.x, yp.y, yp.z, yp.w};
      const float ob4[4] = {ob.x, ob.y, ob.z, ob.w};
      float out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ce = c4[e];
        const bool wrapped = (unsigned)e > w;                  // voxel e lies on the next line
        const bool zlo_ok = e == 0 ? k0 != 0u : (unsigned)(e - 1) != w;  // not the first voxel of a line
        const bool zhi_ok = (unsigned)e != w;                  // not the last voxel of a line
        const unsigned je = wrapped ? (j0 + 1u == ny ? 0u : j0 + 1u) : j0;
        const bool ylo_ok = je != 0u, yhi_ok = je + 1u != ny;
        const bool xlo_ok = idx0 + (unsigned)e >= nynz;        // (x upper face: the load returned 0)
        const float xb = xlo_ok ? ce - xm4[e] : 0.f, xf = xp4[e] - ce;
        const float yb = ylo_ok ? ce - ym4[e] : 0.f, yf = (yhi_ok ? yp4[e] : 0.f) - ce;
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
      if (!OBJ && valid) *reinterpret_cast<float4 *>(q + idx0) = make_float4(out[0], out[1], out[2], out[3]);
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

static int flat_grid(unsigned nchunk) {
  const unsigned cap = 4096;  // (kMaxPartials = 8192 partial sums at most)
  return (int)(nchunk < cap ? (nchunk < 1 ? 1 : nchunk) : cap);
}

int dtd_flat_blocks(Dim3i dd) {
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
  // the number of partials must not depend on q's alignment: callers size their reduction with
  // dtd_flat_blocks(dd)
  const dim3 grid(dtd_flat_blocks(dd)), block(kBlock);
  if (objb)
    hipLaunchKernelGGL((k_dtd_flat<true, true>), grid, block, 0, st, A, done);
  else if (partials)
    hipLaunchKernelGGL((k_dtd_flat<true, false>), grid, block, 0, st, A, done);
  else
    hipLaunchKernelGGL((k_dtd_flat<false, false>), grid, block, 0, st, A, done);
  return 0;
}

}  // namespace unires
