// common.hpp - shared device helpers for libunires_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/unires_hip.h"

namespace unires {

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kBlock = 256;         // 4 waves, one per SIMD
constexpr int kMaxPartials = 1024;  // blocks of a dot-producing kernel (<= this)

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

}  // namespace unires
