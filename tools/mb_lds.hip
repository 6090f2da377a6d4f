// micro-benchmarks that decide the splat design on gfx950:
//   (1) ds_add_f32 (LDS float atomic, no return) vs plain ds_read/add/ds_write, lanes on consecutive words
//   (2) v_pk_mul_f32 vs v_mul_f32 issue rate
//   (3) DPP-modified v_add_f32 rate
// build: hipcc -O3 --offload-arch=gfx950 tools/mb_lds.hip -o tools/mb_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

constexpr int kIters = 256;

// MODE 0: ds_add_f32 x8 per iteration; 1: read/add/write x8 (4 then 4, like the splat); 2: ds_add_rtn_f32
template <int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_lds(float *out, int stride) {
  __shared__ float acc[WAVES][2048];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float *a = acc[w];
  for (int i = lane; i < 2048; i += 64) a[i] = 0.f;
  __syncthreads();
  float v = 1.0f + lane;
  float s = 0.f;
  for (int it = 0; it < kIters; ++it) {
    const int base = (it * 7) & 1023;  // moving window; lanes consecutive
    float *q = a + ((base + lane) & 1023);
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        __hip_atomic_fetch_add(q + c * stride, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        s += __hip_atomic_fetch_add(q + c * stride, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      asm volatile("" ::: "memory");
      {
        const float o0 = q[0], o1 = q[stride], o2 = q[2 * stride], o3 = q[3 * stride];
        q[0] = o0 + v, q[stride] = o1 + v, q[2 * stride] = o2 + v, q[3 * stride] = o3 + v;
      }
      asm volatile("" ::: "memory");
      {
        const float o0 = q[4 * stride], o1 = q[5 * stride], o2 = q[6 * stride], o3 = q[7 * stride];
        q[4 * stride] = o0 + v, q[5 * stride] = o1 + v, q[6 * stride] = o2 + v, q[7 * stride] = o3 + v;
      }
      asm volatile("" ::: "memory");
    }
    v += 0.5f;
  }
  __syncthreads();
  float t = s;
  for (int i = lane; i < 2048; i += 64) t += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// VALU issue rate: 0 v_mul_f32 x16, 1 v_pk_mul_f32 x8 (same flops), 2 v_pk_mul_f32 x16, 3 v_add_f32 dpp x16, 4 v_fma x16
template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float *out, int n) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + 1e-6f * (threadIdx.x + i);
  const float m = 1.0000001f;
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
    } else if (MODE == 1 || MODE == 2) {
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 *p = reinterpret_cast<v2 *>(r);
      v2 mm = {m, m};
#pragma unroll
      for (int rep = 0; rep < (MODE == 2 ? 2 : 1); ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r[i]) : "v"(m));
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
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

int main() {
  float *out; CK(hipMalloc(&out, 1 << 24));
  const int nb = 256 * 8;
  const double clk = 2.4e9;
  for (int stride : {34, 64, 1}) {
    printf("LDS stride %d (per wave-instruction CU clocks, %d blocks x 4 waves, 16KB LDS/wave -> ~2 waves/SIMD):\n", stride, nb);
    auto rep = [&](const char *nm, float us, int ops) {
      // wave-instr total = nb*4*kIters*ops ; per CU = /256
      const double per = us * 1e-6 * clk / ((double)nb * 4 * kIters * ops / 256.0);
      printf("  %-28s %8.1f us  -> %.2f clk per wave-op per CU\n", nm, us, per);
    };
    rep("ds_add_f32 x8", timeit([&] { hipLaunchKernelGGL((k_lds<0, 4>), dim3(nb), dim3(256), 0, 0, out, stride); }, 10), 8);
    rep("ds_add_rtn_f32 x8", timeit([&] { hipLaunchKernelGGL((k_lds<2, 4>), dim3(nb), dim3(256), 0, 0, out, stride); }, 10), 8);
    rep("read+add+write x8 (r+w=2)", timeit([&] { hipLaunchKernelGGL((k_lds<1, 4>), dim3(nb), dim3(256), 0, 0, out, stride); }, 10), 16);
  }
  const int n = 4096;
  auto repv = [&](const char *nm, float us, int per_it) {
    const double wi = (double)4096 * 4 * n * per_it;  // wave-instr
    printf("  %-28s %8.1f us -> %.2f clk per wave-instr per SIMD\n", nm, us, us * 1e-6 * clk / (wi / 1024.0));
  };
  printf("VALU issue (4096 blocks x 256 thr):\n");
  repv("v_mul_f32 x16", timeit([&] { hipLaunchKernelGGL((k_valu<0>), dim3(4096), dim3(256), 0, 0, out, n); }, 5), 16);
  repv("v_pk_mul_f32 x8", timeit([&] { hipLaunchKernelGGL((k_valu<1>), dim3(4096), dim3(256), 0, 0, out, n); }, 5), 8);
  repv("v_pk_mul_f32 x16", timeit([&] { hipLaunchKernelGGL((k_valu<2>), dim3(4096), dim3(256), 0, 0, out, n); }, 5), 16);
  repv("v_add_f32_dpp x16", timeit([&] { hipLaunchKernelGGL((k_valu<3>), dim3(4096), dim3(256), 0, 0, out, n); }, 5), 16);
  repv("v_fma_f32 x16", timeit([&] { hipLaunchKernelGGL((k_valu<4>), dim3(4096), dim3(256), 0, 0, out, n); }, 5), 16);
  return 0;
}
