// What a streaming 7-point-stencil pass over a 256^3 float volume can reach on MI355X, by construction:
// from a plain copy up to the neighbour loads the one-kernel matvecs issue (aligned.hip, stencil.hip, shift.hip).
// Every variant moves the same algorithmic bytes (read p once, write q once: 134 MB); operands cycle through a
// ring of volumes larger than the 256 MB Infinity Cache ("cold").
// build: hipcc -O3 --offload-arch=gfx950 tools/mb_stream.hip -o tools/mb_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int NX = 256, NY = 256, NZ = 256;
constexpr size_t N = (size_t)NX * NY * NZ;

// MODE 0: copy, one line per wave and trip          1: + x / y neighbour loads (5 loads per line)
//      2: pairs of lines (8 loads per pair)         3: copy, two lines per wave in flight
//      4: flat float4 copy (grid-stride, no lines)  5: flat + the four neighbour loads at +-NZ, +-NY*NZ
//      6: mode 1 with the store delayed by an LDS round-trip chain (what a z operator costs a wave)
// marching copy: a wave owns NL y-adjacent lines and walks XR planes along x (the access pattern of
// k_ata_shift_m), one plane of loads in flight ahead of the stores; YN: also load the two y neighbours
template <int NL, bool YN>
__global__ void __launch_bounds__(256) k_march(const float *__restrict__ p, float *__restrict__ q, int xr, int nt_store) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t sx = (size_t)NY * NZ, sy = NZ;
  const int hyn = NY / NL, nxr = (NX + xr - 1) / xr, ntasks = hyn * nxr;
  for (int task = blockIdx.x * 4 + w; task < ntasks; task += gridDim.x * 4) {
    const int r = task / hyn, vy0 = NL * (task - r * hyn);
    const int xa = r * xr, xb = min(xa + xr, NX);
    const float *pc = p + ((size_t)xa * NY + vy0) * NZ + 4 * lane;
    float *qc = q + ((size_t)xa * NY + vy0) * NZ + 4 * lane;
    f4 cur[NL + 2], nxt[NL + 2];
    auto load = [&](const float *pp, f4 (&v)[NL + 2]) {
#pragma unroll
      for (int b = 0; b < NL + 2; ++b)
        if (YN || (b >= 1 && b <= NL)) v[b] = *reinterpret_cast<const f4 *>(pp + ((long long)b - 1) * (long long)sy * ((b == 0 && vy0 == 0) || (b == NL + 1 && vy0 + NL >= NY) ? 0 : 1));
    };
    load(pc, cur);
    for (int vx = xa; vx < xb; ++vx) {
      if (vx + 1 < xb) load(pc + sx, nxt);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        f4 o = cur[1 + l];
        if (YN) o = 2.f * o - cur[l] - cur[2 + l];
        if (nt_store) __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(qc + l * sy));
        else *reinterpret_cast<f4 *>(qc + l * sy) = o;
      }
#pragma unroll
      for (int b = 0; b < NL + 2; ++b) cur[b] = nxt[b];
      pc += sx, qc += sx;
    }
  }
}

// flat marching over a volume whose planes are NOT a multiple of 16 bytes (181 x 217 x 181: plane = 39277
// floats): a lane owns four consecutive in-plane voxels and walks along x with stride `plane`; every 16-byte
// access except in plane 0 is 4-byte aligned only.  YN: + the y neighbours at +-nz.  (What k_dtd_flat would do
// if it marched; k_flat5 below is what it does now: five 16-byte loads at 0, +-nz, +-plane per vector.)
template <bool YN>
__global__ void __launch_bounds__(256) k_flat_march(const float *__restrict__ p, float *__restrict__ q, unsigned nx, unsigned plane,
                                                    unsigned nz, int xr) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned nvp = plane / 4, ncw = (nvp + 63) / 64, nxr = (nx + xr - 1) / xr, ntasks = ncw * nxr;
  const size_t n = (size_t)nx * plane;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (unsigned)(n * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(q, 0, (unsigned)(n * 4), 0x00020000);
  for (unsigned task = blockIdx.x * 4 + w; task < ntasks; task += gridDim.x * 4) {
    const unsigned r = task / ncw, cw = task - r * ncw;
    const unsigned v = cw * 64 + lane;
    if (v >= nvp) continue;
    const unsigned xa = r * xr, xb = min(xa + (unsigned)xr, nx);
    unsigned off = 4u * (xa * plane + 4u * v);
    f4 cur = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, 0));
    f4 ym = cur, yp = cur;
    for (unsigned vx = xa; vx < xb; ++vx) {
      f4 nxt = cur;
      if (vx + 1 < xb) nxt = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rp, off + 4u * plane, 0, 0));
      if (YN) {
        ym = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rp, off - 4u * nz, 0, 0));
        yp = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rp, off + 4u * nz, 0, 0));
      }
      f4 o = cur;
      if (YN) o = 2.f * cur - ym - yp;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), rq, off, 0, 2);
      cur = nxt;
      off += 4u * plane;
    }
  }
}

__global__ void __launch_bounds__(256) k_flat5(const float *__restrict__ p, float *__restrict__ q, unsigned n, unsigned plane, unsigned nz) {
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, n * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(q, 0, n * 4u, 0x00020000);
  const unsigned n4 = n / 4, stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const unsigned off = 16u * i;
    auto ld = [&](unsigned o) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rp, o, 0, 0)); };
    const f4 c = ld(off), a = ld(off - 4u * nz), b = ld(off + 4u * nz), d = ld(off - 4u * plane), g = ld(off + 4u * plane);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, 4.f * c - a - b - d - g), rq, off, 0, 2);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k(const float *__restrict__ p, float *__restrict__ q, int nt_store) {
  __shared__ float lds[4][320];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nlines = NX * NY;
  const size_t sx = (size_t)NY * NZ, sy = NZ;
  auto st = [&](f4 v, float *dst) {
    if (nt_store) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(dst));
    else *reinterpret_cast<f4 *>(dst) = v;
  };
  if (MODE == 4 || MODE == 5) {
    const size_t n4 = N / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float *pc = p + 4 * i;
      f4 c = *reinterpret_cast<const f4 *>(pc);
      if (MODE == 5) {
        const size_t e = 4 * i;
        const f4 a = *reinterpret_cast<const f4 *>(e >= sy ? pc - sy : pc), b = *reinterpret_cast<const f4 *>(e + sy < N ? pc + sy : pc);
        const f4 d = *reinterpret_cast<const f4 *>(e >= sx ? pc - sx : pc), g = *reinterpret_cast<const f4 *>(e + sx < N ? pc + sx : pc);
        c = 4.f * c - a - b - d - g;
      }
      st(c, q + 4 * i);
    }
    return;
  }
  const int step = gridDim.x * 4;
  if (MODE == 2) {
    const int npairs = NX * NY / 2;
    for (int pr = blockIdx.x * 4 + w; pr < npairs; pr += step) {
      const int vx = pr / (NY / 2), vy = 2 * (pr - vx * (NY / 2));
      const float *pc = p + ((size_t)vx * NY + vy) * NZ + 4 * lane;
      const bool lx = vx > 0, hx = vx + 1 < NX, ly = vy > 0, hy = vy + 2 < NY;
      const f4 c0 = *reinterpret_cast<const f4 *>(pc), c1 = *reinterpret_cast<const f4 *>(pc + sy);
      const f4 ym = *reinterpret_cast<const f4 *>(ly ? pc - sy : pc), yp = *reinterpret_cast<const f4 *>(hy ? pc + 2 * sy : pc);
      const f4 xm0 = *reinterpret_cast<const f4 *>(lx ? pc - sx : pc), xp0 = *reinterpret_cast<const f4 *>(hx ? pc + sx : pc);
      const f4 xm1 = *reinterpret_cast<const f4 *>(lx ? pc + sy - sx : pc), xp1 = *reinterpret_cast<const f4 *>(hx ? pc + sy + sx : pc);
      float *qc = q + ((size_t)vx * NY + vy) * NZ + 4 * lane;
      st(4.f * c0 - ym - c1 - xm0 - xp0, qc);
      st(4.f * c1 - c0 - yp - xm1 - xp1, qc + sy);
    }
    return;
  }
  for (int line = blockIdx.x * 4 + w; line < nlines; line += step * (MODE == 3 ? 2 : 1)) {
    const int vx = line / NY, vy = line - vx * NY;
    const float *pc = p + (size_t)line * NZ + 4 * lane;
    f4 c = *reinterpret_cast<const f4 *>(pc);
    if (MODE == 3) {
      const int l2 = line + step;
      f4 c2 = {0, 0, 0, 0};
      if (l2 < nlines) c2 = *reinterpret_cast<const f4 *>(p + (size_t)l2 * NZ + 4 * lane);
      st(c, q + (size_t)line * NZ + 4 * lane);
      if (l2 < nlines) st(c2, q + (size_t)l2 * NZ + 4 * lane);
      continue;
    }
    if (MODE == 1 || MODE == 6) {
      const bool lx = vx > 0, hx = vx + 1 < NX, ly = vy > 0, hy = vy + 1 < NY;
      const f4 a = *reinterpret_cast<const f4 *>(ly ? pc - sy : pc), b = *reinterpret_cast<const f4 *>(hy ? pc + sy : pc);
      const f4 d = *reinterpret_cast<const f4 *>(lx ? pc - sx : pc), g = *reinterpret_cast<const f4 *>(hx ? pc + sx : pc);
      c = 4.f * c - a - b - d - g;
    }
    if (MODE == 6) {  // a dependent LDS chain: write the line, 8 dependent read / write trips, read it back
      float *l = lds[w];
      *reinterpret_cast<f4 *>(l + 4 * lane) = c;
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        asm volatile("" ::: "memory");
        acc += l[(lane * 6 + t) & 255];
        asm volatile("" ::: "memory");
        l[256 + lane] = acc;
      }
      asm volatile("" ::: "memory");
      c += l[256 + (lane >> 2)] * 1e-30f;
    }
    st(c, q + (size_t)line * NZ + 4 * lane);
  }
}

template <int MODE> void run(const char *nm, float **bufs, int nb, int blocks, int nt) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 24;
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, bufs[(2 * i) % nb], bufs[(2 * i + 1) % nb], nt);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, bufs[(2 * i) % nb], bufs[(2 * i + 1) % nb], nt);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("  %-58s %5d blocks %s: %6.1f us  %.2f TB/s (134 MB)\n", nm, blocks, nt ? "nt" : "  ", us, 2.0 * N * 4 / us / 1e6);
}

template <int NL, bool YN> void run_march(const char *nm, float **bufs, int nb, int xr, int nt) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 24, tasks = (NY / NL) * ((NX + xr - 1) / xr), blocks = (tasks + 3) / 4;
  for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k_march<NL, YN>), dim3(blocks), dim3(256), 0, 0, bufs[(2 * i) % nb], bufs[(2 * i + 1) % nb], xr, nt);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_march<NL, YN>), dim3(blocks), dim3(256), 0, 0, bufs[(2 * i) % nb], bufs[(2 * i + 1) % nb], xr, nt);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("  %-44s runs of %3d planes, %5d wave tasks %s: %6.1f us  %.2f TB/s (134 MB)\n", nm, xr, tasks, nt ? "nt" : "  ", us, 2.0 * N * 4 / us / 1e6);
}

void run_flat(float **bufs, int nb) {
  const unsigned nx = 181, ny = 217, nz = 181, plane = ny * nz, n = nx * plane;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 40;
  auto time = [&](const char *nm, auto launch) {
    for (int i = 0; i < 4; ++i) launch(bufs[(2 * i) % nb], bufs[(2 * i + 1) % nb]);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(bufs[(2 * i) % nb], bufs[(2 * i + 1) % nb]);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("  181 x 217 x 181: %-60s %6.1f us  %.2f TB/s (56.9 MB)\n", nm, us, 2.0 * n * 4 / us / 1e6);
  };
  time("flat float4 copy (aligned)", [&](float *a, float *b) { hipLaunchKernelGGL((k<4>), dim3(2048), dim3(256), 0, 0, a, b, 1); });
  time("five 16-byte loads per vector (k_dtd_flat's pattern)", [&](float *a, float *b) { hipLaunchKernelGGL(k_flat5, dim3(2048), dim3(256), 0, 0, a, b, n, plane, nz); });
  for (int xr : {6, 9, 12, 16}) {
    const unsigned tasks = ((plane / 4 + 63) / 64) * ((nx + xr - 1) / xr);
    char nm[128];
    snprintf(nm, sizeof(nm), "marching, unaligned, runs of %d planes (%u tasks): copy", xr, tasks);
    time(nm, [&](float *a, float *b) { hipLaunchKernelGGL((k_flat_march<false>), dim3((tasks + 3) / 4), dim3(256), 0, 0, a, b, nx, plane, nz, xr); });
    snprintf(nm, sizeof(nm), "marching, unaligned, runs of %d planes (%u tasks): + y neighbours", xr, tasks);
    time(nm, [&](float *a, float *b) { hipLaunchKernelGGL((k_flat_march<true>), dim3((tasks + 3) / 4), dim3(256), 0, 0, a, b, nx, plane, nz, xr); });
  }
}

int main() {
  const int nb = 6;  // 6 x 67 MB = 403 MB > Infinity Cache
  float *bufs[nb];
  for (int i = 0; i < nb; ++i) { CK(hipMalloc(&bufs[i], N * 4)); CK(hipMemset(bufs[i], 0, N * 4)); }
  run_flat(bufs, nb);
  for (int xr : {16, 17}) {
    run_march<1, false>("marching copy, 1 line per wave", bufs, nb, xr, 1);
    run_march<2, false>("marching copy, 2 lines per wave", bufs, nb, xr, 1);
    run_march<2, true>("marching, 2 lines + y neighbours (4 loads)", bufs, nb, xr, 1);
  }
  for (int nt : {1})
    for (int blocks : {1024, 4096}) {
      run<4>("flat float4 copy (grid-stride)", bufs, nb, blocks, nt);
      run<5>("flat + 4 neighbour loads (+-NZ, +-NY NZ)", bufs, nb, blocks, nt);
      run<0>("line copy: one line per wave and trip", bufs, nb, blocks, nt);
      run<3>("line copy: two lines in flight per wave", bufs, nb, blocks, nt);
      run<1>("line + x / y neighbours: 5 loads per line", bufs, nb, blocks, nt);
      run<2>("pairs of lines: 8 loads per pair", bufs, nb, blocks, nt);
      run<6>("5 loads per line + a dependent LDS chain before the store", bufs, nb, blocks, nt);
    }
  return 0;
}
