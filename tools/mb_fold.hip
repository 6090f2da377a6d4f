// mb_fold.hip - can the last workgroup of a producer kernel reduce the kernel's per-wave partials WITHOUT an
// agent-scope release fence (which on gfx950 writes back the XCD's whole L2: EXPERIMENTS E2, 4 200 -> 1 550 it/s)?
// Form tested: partials written with agent-scope atomic stores (sc1: write-through), s_waitcnt vmcnt(0), ticket by
// agent-scope atomic add, the last workgroup reads the partials with agent-scope atomic loads.  Checks every
// iteration's sum bit for bit against a one-block kernel behind a kernel boundary, and times both chains.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_fold.hip -o tools/mb_fold && tools/mb_fold
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kB = 256, kW = 4;
struct State { double sum, alpha; unsigned ticket; unsigned bad; };

__device__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ double block_sum_fixed(double v) {
  __shared__ double s[kW];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < kW; ++w) t += s[w];
  __syncthreads();
  return t;
}
__device__ double sum_partials(const double *part, int g, bool coherent) {
  double v = 0.0;
  for (int i = threadIdx.x; i < g; i += kB)
    v += coherent ? __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[i];
  return block_sum_fixed(v);
}

// producer: q = s * p (a volume pass that dirties the L2), partial = sum p * q per wave
template <bool FOLD>
__global__ void __launch_bounds__(kB) k_prod(const float *__restrict__ p, float *__restrict__ q, size_t n, float s,
                                              double *part, State *st) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * kB + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * kB) {
    float4 v = reinterpret_cast<const float4 *>(p)[i];
    float4 o = make_float4(s * v.x, s * v.y, s * v.z, s * v.w);
    reinterpret_cast<float4 *>(q)[i] = o;
    acc += (double)(v.x * o.x) + (double)(v.y * o.y) + (double)(v.z * o.z) + (double)(v.w * o.w);
  }
  const double w = wave_sum(acc);
  const int slot = blockIdx.x * kW + (threadIdx.x >> 6);
  if (!FOLD) {
    if ((threadIdx.x & 63) == 0) part[slot] = w;
    return;
  }
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_store(part + slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __shared__ unsigned last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  const double tot = sum_partials(part, (int)gridDim.x * kW, true);
  if (threadIdx.x == 0) {
    st->sum = tot;
    st->alpha = 1.0 / tot;
    __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void __launch_bounds__(kB) k_scalar(const double *part, int g, State *st) {
  const double tot = sum_partials(part, g, false);
  if (threadIdx.x == 0) st->sum = tot, st->alpha = 1.0 / tot;
}
// consumer: r -= alpha * q  (reads the scalar the chain produced), and records the sum it saw
__global__ void __launch_bounds__(kB) k_cons(const State *st, const float *__restrict__ q, float *__restrict__ r, size_t n,
                                              double *seen, int it) {
  const float a = (float)st->alpha;
  for (size_t i = (size_t)blockIdx.x * kB + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * kB) {
    float4 v = reinterpret_cast<const float4 *>(q)[i], o = reinterpret_cast<float4 *>(r)[i];
    o.x -= a * v.x, o.y -= a * v.y, o.z -= a * v.z, o.w -= a * v.w;
    reinterpret_cast<float4 *>(r)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) seen[it] = st->sum;
}

int main(int argc, char **argv) {
  const size_t n = (size_t)256 * 256 * 256;
  const int grid = argc > 1 ? atoi(argv[1]) : 1024, iters = 200;
  float *p, *q, *r;
  double *part, *seen0, *seen1;
  State *st;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&q, n * 4)); CK(hipMalloc(&r, n * 4));
  CK(hipMalloc(&part, grid * kW * 8)); CK(hipMalloc(&seen0, iters * 8)); CK(hipMalloc(&seen1, iters * 8));
  CK(hipMalloc(&st, sizeof(State)));
  CK(hipMemset(st, 0, sizeof(State))); CK(hipMemset(r, 0, n * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int it = 0; it < iters; ++it) {
        const float sc = 1.f + 0.01f * (float)(it % 37);
        if (mode == 0) {
          hipLaunchKernelGGL(k_prod<false>, dim3(grid), dim3(kB), 0, s, p, q, n, sc, part, st);
          hipLaunchKernelGGL(k_scalar, dim3(1), dim3(kB), 0, s, part, grid * kW, st);
        } else {
          hipLaunchKernelGGL(k_prod<true>, dim3(grid), dim3(kB), 0, s, p, q, n, sc, part, st);
        }
        hipLaunchKernelGGL(k_cons, dim3(grid), dim3(kB), 0, s, st, q, r, n, mode ? seen1 : seen0, it);
      }
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s grid %d: %.1f us per iteration\n", mode ? "folded (last workgroup)" : "separate scalar kernel", grid, 1e3 * ms / iters);
    }
  }
  std::vector<double> a(iters), b(iters);
  CK(hipMemcpy(a.data(), seen0, iters * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), seen1, iters * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int it = 0; it < iters; ++it) bad += a[it] != b[it];
  printf("sums differing between the two chains: %d of %d (first: %.17g vs %.17g)\n", bad, iters, a[0], b[0]);
  return bad != 0;
}
