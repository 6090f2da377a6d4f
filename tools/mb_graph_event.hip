// do event-record nodes captured into a hipGraph give kernel timings on replay?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float *p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
  float *d; hipMalloc(&d, 4096);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  hipGraph_t g; hipGraphExec_t ge;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { printf("capture refused\n"); return 1; }
  hipError_t r0 = hipEventRecord(e0, st);
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 200000);
  hipError_t r1 = hipEventRecord(e1, st);
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 400000);
  hipError_t r2 = hipEventRecord(e2, st);
  hipError_t ce = hipStreamEndCapture(st, &g);
  printf("record during capture: %d %d %d  end %d\n", r0, r1, r2, ce);
  if (ce != hipSuccess) return 1;
  size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn); printf("nodes %zu\n", nn);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
  for (int rep = 0; rep < 3; ++rep) {
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    float a = -1, b = -1; hipError_t x = hipEventElapsedTime(&a, e0, e1), y = hipEventElapsedTime(&b, e1, e2);
    printf("replay %d: %d %.3f ms  %d %.3f ms\n", rep, x, a, y, b);
  }
  return 0;
}
