// VALU / LDS issue-rate table for gfx950 (what one wave64 instruction costs a SIMD / the CU's LDS)
// build: hipcc -O3 --offload-arch=gfx950 tools/mb_valu.hip -o tools/mb_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float *out, int n, float sm) {
  float r[16];
  int ri[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + 1e-6f * (threadIdx.x + i), ri[i] = threadIdx.x + i;
  const float m = 1.0000001f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 *p = reinterpret_cast<v2 *>(r);
  v2 mm = {m, m};
  for (int it = 0; it < n; ++it) {
#define OP(i)                                                                                       \
  if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m));                       \
  else if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m));                  \
  else if (MODE == 2) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[i]) : "v"(m));                 \
  else if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));              \
  else if (MODE == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "s"(sm));                 \
  else if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "s"(sm), "v"(m));     \
  else if (MODE == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(ri[i]) : "v"(ri[15 - i])); \
  else if (MODE == 7) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));                           \
  else if (MODE == 8) asm volatile("v_fract_f32 %0, %0" : "+v"(r[i]));                             \
  else if (MODE == 9) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));                             \
  else if (MODE == 10) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ri[i]) : "v"(ri[15 - i])); \
  else if (MODE == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m) : "vcc"); \
  else if (MODE == 12) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(r[i]));                         \
  else if (MODE == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ri[i]) : "v"(ri[15 - i]));       \
  else if (MODE == 14) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i])); \
  else if (MODE == 15) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(ri[i]));                         \
  else if (MODE == 16) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(m));             \
  else if (MODE == 17) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(ri[i]));                      \
  else if (MODE == 18) asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(ri[i]));                     \
  else if (MODE == 19) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(ri[i]));                       \
  else if (MODE == 20) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));            \
  else if (MODE == 21) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(m) : "vcc");
    REP16(OP)
#undef OP
    if (MODE == 30) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
    } else if (MODE == 31) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(mm));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(mm));
    } else if (MODE == 32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
    } else if (MODE == 40) {  // SALU
      int s = it;
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
      ri[0] += s;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i] + (float)ri[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS instruction mixes, per wave: MODE 0: 4x ds_read2_b32 + 4x ds_write2_b32 (8 cells RMW)
//  1: 8x ds_read_b32 + 8x ds_write_b32 ; 2: 8x ds_read_b32 only ; 3: 4x ds_read2_b32 only ; 4: 2x ds_read_b128
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(float *out, int n) {
  __shared__ float acc[4][1920];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float *a = acc[w];
  for (int i = lane; i < 1920; i += 64) a[i] = 0.f;
  __syncthreads();
  float v = 1.f + lane, s = 0.f;
  for (int it = 0; it < n; ++it) {
    float *q = a + ((it * 37 + lane) & 1023);
    asm volatile("" ::: "memory");
    if (MODE == 0 || MODE == 1) {
      const float o0 = q[0], o1 = q[32], o2 = q[192], o3 = q[224];
      q[0] = o0 + v, q[32] = o1 + v, q[192] = o2 + v, q[224] = o3 + v;
      asm volatile("" ::: "memory");
      const float u0 = q[1], u1 = q[33], u2 = q[193], u3 = q[225];
      q[1] = u0 + v, q[33] = u1 + v, q[193] = u2 + v, q[225] = u3 + v;
    } else if (MODE == 2 || MODE == 3) {
      s += q[0] + q[32] + q[192] + q[224] + q[1] + q[33] + q[193] + q[225];
    } else {
      const float4 *q4 = reinterpret_cast<const float4 *>(a) + ((it * 5 + lane) & 127);
      const float4 x = q4[0], y = q4[128];
      s += x.x + x.w + y.y + y.z;
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  for (int i = lane; i < 1920; i += 64) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

template <int MODE> void valu(const char *nm, float *out, int blocks) {
  const int n = 2048;
  const float us = timeit([&] { hipLaunchKernelGGL((k_valu<MODE>), dim3(blocks), dim3(256), 0, 0, out, n, 1.0000001f); }, 5);
  const double wi = (double)blocks * 4 * n * 16;
  printf("  %-26s %8.1f us -> %.2f clk / wave-instr / SIMD (2.4 GHz nominal)\n", nm, us, us * 1e-6 * 2.4e9 / (wi / 1024.0));
}
template <int MODE> void lds(const char *nm, float *out, int ops) {
  const int n = 2048, blocks = 2048;
  const float us = timeit([&] { hipLaunchKernelGGL((k_lds<MODE>), dim3(blocks), dim3(256), 0, 0, out, n); }, 5);
  const double wi = (double)blocks * 4 * n;
  printf("  %-34s %8.1f us -> %.1f clk / iteration / CU (%d DS instr)\n", nm, us, us * 1e-6 * 2.4e9 / (wi / 256.0), ops);
}

int main() {
  float *out; CK(hipMalloc(&out, 1 << 24));
  for (int blocks : {8192, 1024}) {
    printf("VALU, %d blocks x 256 threads (%s):\n", blocks, blocks > 2048 ? "8 waves/SIMD" : "1 wave/SIMD");
    valu<0>("v_mul_f32 (VOP2)", out, blocks);
    valu<16>("v_mul_f32_e64 (VOP3)", out, blocks);
    valu<1>("v_add_f32", out, blocks);
    valu<12>("v_sub_f32 1.0, v", out, blocks);
    valu<2>("v_fmac_f32 (VOP2)", out, blocks);
    valu<3>("v_fma_f32 v,v,v", out, blocks);
    valu<4>("v_mul_f32 s,v", out, blocks);
    valu<5>("v_fma_f32 v,s,v", out, blocks);
    valu<6>("v_mad_u32_u24", out, blocks);
    valu<13>("v_add_u32", out, blocks);
    valu<10>("v_lshl_add_u32", out, blocks);
    valu<17>("v_lshlrev_b32", out, blocks);
    valu<18>("v_and_b32 lit", out, blocks);
    valu<19>("v_bfe_u32", out, blocks);
    valu<7>("v_cvt_f32_i32", out, blocks);
    valu<15>("v_cvt_u32_f32", out, blocks);
    valu<8>("v_fract_f32", out, blocks);
    valu<9>("v_floor_f32", out, blocks);
    valu<11>("v_cndmask_b32", out, blocks);
    valu<20>("v_med3_f32", out, blocks);
    valu<21>("v_cmp_lt_f32", out, blocks);
    valu<14>("v_mov_b32_dpp wave_shr", out, blocks);
    valu<30>("v_pk_mul_f32", out, blocks);
    valu<31>("v_pk_fma_f32", out, blocks);
    valu<32>("v_pk_add_f32", out, blocks);
    valu<40>("s_add_u32 (per CU: x4)", out, blocks);
  }
  printf("LDS (2048 blocks x 4 waves, 7.7 KB/wave):\n");
  lds<0>("8-cell RMW (compiler's choice)", out, 16);
  lds<2>("8 reads", out, 8);
  lds<4>("2x ds_read_b128", out, 2);
  return 0;
}
