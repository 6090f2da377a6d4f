// Wave64 VALU issue cost on gfx950, measured two ways at once (reconcile DESIGN's 4 clk per
// v_fma_f32 with the guide's "2 cyc (SIMD-32)" row):
//   (a) wall time of the launch at a NOMINAL 2.4 GHz (the r2 method, profiles/r02_mb_valu.txt), and
//   (b) s_memtime ticks (= shader cycles) counted by the waves themselves, so the figure does not depend on the
//       clock the chip chooses under the load (DVFS),
// with every source operand of every instruction a DISTINCT register (16 independent accumulator chains, sources
// from two other 16-register arrays: no operand is read twice, consecutive instructions touch different VGPR
// banks), VOP2 and VOP3 encodings, and 1 / 2 / 4 / 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/mb_valu2.hip -o tools/mb_valu2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *ticks, int n) {
  float r[16], a[16], b[16];
  v2 pr[8], pa[8], pb[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.f + 1e-6f * (threadIdx.x + i), a[i] = 1.f + 1e-7f * i, b[i] = 1e-9f * (i + 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) pr[i] = v2{r[2 * i], r[2 * i + 1]}, pa[i] = v2{a[2 * i], a[2 * i + 1]}, pb[i] = v2{b[2 * i], b[2 * i + 1]};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; ++it) {
#define OP(i)                                                                                              \
  if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a[i]), "v"(b[i]));            \
  else if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a[i]), "v"(b[i]));          \
  else if (MODE == 2) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(a[i]));                  \
  else if (MODE == 3) asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(r[i]) : "v"(a[i]));                  \
  else if (MODE == 4) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(b[i]));                  \
  else if (MODE == 5) asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(r[i]) : "v"(b[i]));                  \
  else if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a[(i + 1) & 15]), "v"(b[(i + 2) & 15])); \
  else if (MODE == 7) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a[i]), "v"(b[i]));   \
  else if (MODE == 8) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(r[i]) : "v"(b[i]));
    REP16(OP)
#undef OP
    if (MODE == 20) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pr[i]) : "v"(pa[i]), "v"(pb[i]));
    } else if (MODE == 21) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pr[i]) : "v"(pa[i]));
    } else if (MODE == 22) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pr[i]) : "v"(pb[i]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += pr[i].x + pr[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char *nm, float *out, unsigned long long *ticks, int blocks) {
  const int n = 4096, waves = blocks * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, n);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  static unsigned long long h[65536];
  CK(hipMemcpy(h, ticks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
  double tk = 0; for (int i = 0; i < waves; ++i) tk += (double)h[i];
  tk /= waves;
  const double per_simd = waves / 1024.0;  // waves sharing a SIMD
  const double instr = (double)n * 16;
  printf("  %-34s %d waves/SIMD: %6.2f ticks per instruction and wave -> %5.2f cycles of its SIMD per wave-instruction; "
         "wall %7.1f us = %5.2f clk at 2.4 GHz (clock by ticks: %.2f GHz)\n",
         nm, (int)per_simd, tk / instr, tk / instr / per_simd, ms * 1e3, ms * 1e-3 * 2.4e9 / (instr * per_simd), tk / (ms * 1e-3) / 1e9);
}

int main() {
  float *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&ticks, sizeof(unsigned long long) * 65536));
  for (int blocks : {256, 512, 1024, 2048}) {  // 1, 2, 4, 8 waves per SIMD: all resident at once
    run<0>("v_fma_f32 d,a,b  (VOP3, distinct)", out, ticks, blocks);
    run<6>("v_fma_f32 a',b',d (other banks)", out, ticks, blocks);
    run<1>("v_fmac_f32 d,a,b (VOP2)", out, ticks, blocks);
    run<2>("v_mul_f32_e32 (VOP2)", out, ticks, blocks);
    run<3>("v_mul_f32_e64 (VOP3)", out, ticks, blocks);
    run<4>("v_add_f32_e32 (VOP2)", out, ticks, blocks);
    run<5>("v_add_f32_e64 (VOP3)", out, ticks, blocks);
    run<8>("v_sub_f32_e32 (VOP2)", out, ticks, blocks);
    run<7>("v_mad_u32_u24", out, ticks, blocks);
    run<20>("v_pk_fma_f32 (2 FMAs per lane)", out, ticks, blocks);
    run<21>("v_pk_mul_f32", out, ticks, blocks);
    run<22>("v_pk_add_f32", out, ticks, blocks);
  }
  return 0;
}
