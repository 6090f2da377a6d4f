// Known-bytes kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths our kernels use (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of 16 B/lane
// streaming reads; other widths are uncalibrated).  Each kernel reads N floats and writes N floats
// (N = 2^26: 268 MB each way, larger than the 256 MB Infinity Cache).
// build: hipcc -O3 --offload-arch=gfx950 tools/mb_traffic.hip -o tools/mb_traffic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <class V>
__global__ void __launch_bounds__(256) k_copy(const V *__restrict__ a, V *__restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}
// read-only (sum) and write-only (fill) forms
template <class V>
__global__ void __launch_bounds__(256) k_read(const V *__restrict__ a, float *__restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i < n; i += st) {
    const V v = a[i];
    s += reinterpret_cast<const float *>(&v)[0];
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_fill1(float *__restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = 1.f;
}
int main() {
  const size_t N = (size_t)1 << 26;
  float *a, *b;
  CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4));
  CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k_copy<float>), dim3(8192), dim3(256), 0, 0, a, b, N);
    hipLaunchKernelGGL((k_copy<float2>), dim3(8192), dim3(256), 0, 0, (const float2 *)a, (float2 *)b, N / 2);
    hipLaunchKernelGGL((k_copy<float4>), dim3(8192), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, N / 4);
    hipLaunchKernelGGL((k_read<float>), dim3(8192), dim3(256), 0, 0, a, b, N);
    hipLaunchKernelGGL((k_read<float4>), dim3(8192), dim3(256), 0, 0, (const float4 *)a, b, N / 4);
    hipLaunchKernelGGL(k_fill1, dim3(8192), dim3(256), 0, 0, b, N);
  }
  CK(hipDeviceSynchronize());
  printf("bytes each way per kernel: %zu\n", N * 4);
  return 0;
}
