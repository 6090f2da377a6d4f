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
// dot fused.  HBM traffic = read p + write q: this is the kernel that can sit on the HBM
// roofline; the general kernels (fused.hip, splat.hip) are instruction-issue bound.
#include "aligned.hpp"

namespace unires {

struct AlignedArgs {
  const float *p;
  float *q;
  Dim3i dd;           // output dims
  int gx, gy, gz;     // grid dims (trimmed taps)
  int xdz;            // x-space z length
  int ox, oy, oz;     // output voxel = grid voxel + o
  int nk, s;          // taps / stride along z
  float kz[UNIRES_MAX_TAPS];
  float se2, so2;     // even/odd slice scaling squared (S(2 scl)), 1,1 = none
  float tau, a0, cx, cy, cz;
  double *partials;
  const float *objb;  // objective mode (see matvec_emit)
  int tab_pad;  // floats of padding so that the z table is 16-byte aligned in LDS
};

constexpr int kLinesPerBlock = kBlock / kWave;

// NP = ceil(nz / 64) z-passes per line, known at compile time so that ALL global loads of a
// line (its own values and its four x/y neighbours) are issued before anything is consumed:
// one memory round trip per line.  NP = 0: generic loop version for longer lines.
template <int NP>
__global__ void __launch_bounds__(kBlock) k_ata_aligned(AlignedArgs A, const int *__restrict__ done) {
  if (done && *done) return;
  extern __shared__ float smem[];
  const int lane = threadIdx.x, w = threadIdx.y;
  const Dim3i dd = A.dd;
  const int nz = dd.z;
  float *kz = smem;                                          // taps (per-lane indexed below)
  float *pl = smem + UNIRES_MAX_TAPS + w * (nz + A.xdz);     // this wave's copy of the p line
  float *xs = pl + nz;                                       // and its x-space line
  // per-output-z table of the transposed conv: (AtA p)[z] = w0 * xs[koff] + w1 * xs[koff+1]
  float4 *ztab = reinterpret_cast<float4 *>(smem + UNIRES_MAX_TAPS + kLinesPerBlock * (nz + A.xdz) +
                                            A.tab_pad);
  if (w == 0 && lane < UNIRES_MAX_TAPS) kz[lane] = A.kz[lane];
  __syncthreads();
  for (int z = w * kWave + lane; z < nz; z += kBlock) {
    const int uz = z - A.oz;
    float w0 = 0.f, w1 = 0.f;
    int koff = 0;
    if (uz >= 0 && uz < A.gz) {
      int klo, khi;
      up_range(uz, A.nk, A.s, A.xdz, klo, khi);
      const int n = khi - klo + 1;
      if (n >= 1) w0 = kz[uz - A.s * klo];
      if (n >= 2) w1 = kz[uz - A.s * (klo + 1)];
      koff = n >= 1 ? klo : 0;
      if (koff > A.xdz - 2) {  // last slice: use the pair (xdz-2, xdz-1), weight on the second
        koff = A.xdz - 2;
        w1 = w0, w0 = 0.f;
      }
    }
    ztab[z] = make_float4(__int_as_float(koff), w0, w1, 0.f);
  }
  __syncthreads();
  const float *__restrict__ p = A.p;
  float *__restrict__ q = A.q;
  const int nlines = dd.x * dd.y;
  const size_t sx = (size_t)dd.y * nz, sy = nz;
  constexpr int MAXP = NP > 0 ? NP : 1;
  double dot = 0.0;
  for (int line = blockIdx.x * kLinesPerBlock + w; line < nlines;
       line += gridDim.x * kLinesPerBlock) {
    const int vx = line / dd.y, vy = line - vx * dd.y;
    const size_t base = (size_t)line * nz;
    const bool hx = vx + 1 < dd.x, lx = vx > 0, hy = vy + 1 < dd.y, ly = vy > 0;
    const size_t bxp = hx ? base + sx : base, bxm = lx ? base - sx : base;
    const size_t byp = hy ? base + sy : base, bym = ly ? base - sy : base;
    float rc[MAXP], rxp[MAXP], rxm[MAXP], ryp[MAXP], rym[MAXP];
    if (NP > 0) {
#pragma unroll
      for (int u = 0; u < MAXP; ++u) {
        const int z = min(u * kWave + lane, nz - 1);
        rc[u] = p[base + z], rxp[u] = p[bxp + z], rxm[u] = p[bxm + z], ryp[u] = p[byp + z],
        rym[u] = p[bym + z];
      }
#pragma unroll
      for (int u = 0; u < MAXP; ++u)
        if (u * kWave + lane < nz) pl[u * kWave + lane] = rc[u];
    } else {
      for (int z = lane; z < nz; z += kWave) pl[z] = p[base + z];
    }
    asm volatile("" ::: "memory");  // single wave: LDS ops execute in order
    const int ux = vx - A.ox, uy = vy - A.oy;
    const bool has = ux >= 0 && ux < A.gx && uy >= 0 && uy < A.gy;  // wave-uniform
    if (has) {
      for (int k = lane; k < A.xdz; k += kWave) {
        float acc = 0.f;
        const int z0 = k * A.s + A.oz;
        for (int t = 0; t < A.nk; ++t) {
          const int z = z0 + t;
          const float v = (z >= 0 && z < nz) ? pl[min(max(z, 0), nz - 1)] : 0.f;
          acc = fmaf(kz[t], v, acc);
        }
        xs[k] = acc * ((k & 1) ? A.so2 : A.se2);
      }
      asm volatile("" ::: "memory");
    }
    auto emit = [&](int z, float c, float vxp, float vxm, float vyp, float vym) {
      float h = 0.f;
      if (has) {
        const float4 tb = ztab[z];
        const int ko = __float_as_int(tb.x);
        h = tb.y * xs[ko] + tb.z * xs[ko + 1];
      }
      const size_t idx = base + z;
      const float vzp = z + 1 < nz ? pl[z + 1] : 0.f, vzm = z > 0 ? pl[z - 1] : c;
      const float xf = (hx ? vxp : 0.f) - c, xb = lx ? c - vxm : 0.f;
      const float yf = (hy ? vyp : 0.f) - c, yb = ly ? c - vym : 0.f;
      const float zf = vzp - c, zb = z > 0 ? c - vzm : 0.f;
      const float out = A.tau * h + A.a0 * c + (A.cx * (xb - xf) + A.cy * (yb - yf) + A.cz * (zb - zf));
      matvec_emit(q, idx, out, c, A.objb, A.partials != nullptr, dot);
    };
    if (NP > 0) {
#pragma unroll
      for (int u = 0; u < MAXP; ++u) {
        const int z = u * kWave + lane;
        if (z < nz) emit(z, rc[u], rxp[u], rxm[u], ryp[u], rym[u]);
      }
    } else {
      for (int z = lane; z < nz; z += kWave) emit(z, pl[z], p[bxp + z], p[bxm + z], p[byp + z], p[bym + z]);
    }
    asm volatile("" ::: "memory");  // the line buffers are reused by the next line
  }
  if (A.partials) {
    const double tot = block_sum(dot);
    if (threadIdx.x == 0 && threadIdx.y == 0) A.partials[blockIdx.x] = tot;
  }
}

int aligned_blocks(Dim3i dd) {
  const long long nb = ((long long)dd.x * dd.y + kLinesPerBlock - 1) / kLinesPerBlock;
  return (int)(nb < kMaxPartials ? nb : kMaxPartials);
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
  const size_t nf = (size_t)kLinesPerBlock * (dd.z + xd.z) + UNIRES_MAX_TAPS;
  const int pad = (int)((4 - nf % 4) % 4);
  const size_t lds = (nf + pad + (size_t)dd.z * 4) * sizeof(float);
  if (lds > 48 * 1024) return 1;
  AlignedArgs G;
  G.p = p, G.q = q, G.dd = dd;
  G.gx = gd.x, G.gy = gd.y, G.gz = gd.z;
  G.xdz = xd.z;
  G.ox = off[0], G.oy = off[1], G.oz = off[2];
  G.nk = T.n[2], G.s = T.s[2];
  for (int i = 0; i < UNIRES_MAX_TAPS; ++i) G.kz[i] = T.t[2][i];
  G.se2 = S2.dim == 2 ? S2.e : 1.f, G.so2 = S2.dim == 2 ? S2.o : 1.f;
  G.tau = tau, G.a0 = a0, G.cx = cx, G.cy = cy, G.cz = cz;
  G.partials = partials;
  G.objb = objb;
  G.tab_pad = pad;
  const dim3 grid(aligned_blocks(dd)), block(kWave, kLinesPerBlock);
  const int np = (dd.z + kWave - 1) / kWave;
  if (np <= 2)
    hipLaunchKernelGGL(k_ata_aligned<2>, grid, block, lds, st, G, done);
  else if (np <= 4)
    hipLaunchKernelGGL(k_ata_aligned<4>, grid, block, lds, st, G, done);
  else if (np <= 6)
    hipLaunchKernelGGL(k_ata_aligned<6>, grid, block, lds, st, G, done);
  else
    hipLaunchKernelGGL(k_ata_aligned<0>, grid, block, lds, st, G, done);
  return 0;
}

}  // namespace unires
