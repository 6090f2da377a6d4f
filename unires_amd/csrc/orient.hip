// orient.hip - axis re-ordering of x-space volumes between the caller's voxel layout and the
// plan's canonical one (orient.hpp).  One pass over the (small: 11 MB at BASELINE config 3)
// observation per 'A' / 'At' / RHS call; A^T A never needs it.
#include <math.h>

#include "orient.hpp"

namespace unires {

Orient orient_of(const Affine &A) {
  static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  double norm[3];
  for (int c = 0; c < 3; ++c) {
    const double a = A.m[c], b = A.m[4 + c], d = A.m[8 + c];
    norm[c] = sqrt(a * a + b * b + d * d);
    if (!(norm[c] > 0.0)) norm[c] = 1.0;
  }
  Orient best;
  double best_score = -1.0;
  for (int p = 0; p < 6; ++p) {
    double score = 0.0;
    for (int j = 0; j < 3; ++j) score += fabs((double)A.m[4 * j + perms[p][j]]) / norm[perms[p][j]];
    if (score > best_score * (1.0 + 1e-12) + 1e-12) {  // strictly better: the identity (p = 0) keeps ties
      best_score = score;
      for (int j = 0; j < 3; ++j) best.perm[j] = perms[p][j];
    }
  }
  for (int j = 0; j < 3; ++j) best.flip[j] = A.m[4 * j + best.perm[j]] < 0.f ? 1 : 0;
  return best;
}

namespace {

struct PermArgs {
  int d[3];        // destination dims (z fastest)
  long long s[3];  // source stride (elements, signed) per destination axis
  long long off;   // source offset of destination voxel (0, 0, 0)
};

// destination-linear copy: coalesced on both sides when the source's fastest axis is the destination's
__global__ void __launch_bounds__(kBlock) k_permute_lin(const float *__restrict__ src, float *__restrict__ dst,
                                                        PermArgs P) {
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * (kBlock / kWave) + threadIdx.y, i = blockIdx.z;
  if (k >= P.d[2] || j >= P.d[1]) return;
  dst[((size_t)i * P.d[1] + j) * P.d[2] + k] = src[P.off + i * P.s[0] + j * P.s[1] + k * P.s[2]];
}

// the source's fastest axis is destination axis F (0 or 1): 64 x 64 tiles over (F, z) through LDS, lanes
// along the source's fastest axis when reading and along the destination's when writing
template <int F>
__global__ void __launch_bounds__(kBlock) k_permute_tr(const float *__restrict__ src, float *__restrict__ dst,
                                                       PermArgs P) {
  __shared__ float tile[kWave][kWave + 1];
  constexpr int R = 1 - F;
  const int k0 = blockIdx.x * kWave, f0 = blockIdx.y * kWave, r = blockIdx.z;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const long long base = P.off + (long long)r * P.s[R];
  for (int yy = ty; yy < kWave; yy += kBlock / kWave) {
    const int a = f0 + tx, k = k0 + yy;
    if (a < P.d[F] && k < P.d[2]) tile[yy][tx] = src[base + a * P.s[F] + k * P.s[2]];
  }
  __syncthreads();
  for (int yy = ty; yy < kWave; yy += kBlock / kWave) {
    const int a = f0 + yy, k = k0 + tx;
    if (a < P.d[F] && k < P.d[2]) {
      const size_t i = F == 0 ? a : r, j = F == 0 ? r : a;
      dst[(i * P.d[1] + j) * P.d[2] + k] = tile[tx][yy];
    }
  }
}

void launch_permute(const float *src, float *dst, const PermArgs &P, hipStream_t st) {
  int f = 2;
  for (int a = 0; a < 3; ++a)
    if (P.s[a] == 1 || P.s[a] == -1) f = a;  // (an axis of extent 1 may tie: any choice is correct)
  if (P.s[2] == 1 || P.s[2] == -1) f = 2;
  const dim3 block(kWave, kBlock / kWave);
  if (f == 2) {
    const dim3 grid((P.d[2] + kWave - 1) / kWave, (P.d[1] + block.y - 1) / block.y, P.d[0]);
    hipLaunchKernelGGL(k_permute_lin, grid, block, 0, st, src, dst, P);
  } else if (f == 0) {
    const dim3 grid((P.d[2] + kWave - 1) / kWave, (P.d[0] + kWave - 1) / kWave, P.d[1]);
    hipLaunchKernelGGL(k_permute_tr<0>, grid, block, 0, st, src, dst, P);
  } else {
    const dim3 grid((P.d[2] + kWave - 1) / kWave, (P.d[1] + kWave - 1) / kWave, P.d[0]);
    hipLaunchKernelGGL(k_permute_tr<1>, grid, block, 0, st, src, dst, P);
  }
}

}  // namespace

void launch_to_canonical(const Orient &O, const float *src, Dim3i du, float *dst, hipStream_t st) {
  const int nu[3] = {du.x, du.y, du.z};
  const long long su[3] = {(long long)du.y * du.z, du.z, 1};
  PermArgs P;
  P.off = 0;
  for (int j = 0; j < 3; ++j) {
    const int a = O.perm[j];
    P.d[j] = nu[a];
    P.s[j] = O.flip[j] ? -su[a] : su[a];
    if (O.flip[j]) P.off += (long long)(nu[a] - 1) * su[a];
  }
  launch_permute(src, dst, P, st);
}

void launch_from_canonical(const Orient &O, const float *src, float *dst, Dim3i du, hipStream_t st) {
  const int nu[3] = {du.x, du.y, du.z};
  int nc[3];
  for (int j = 0; j < 3; ++j) nc[j] = nu[O.perm[j]];
  const long long sc[3] = {(long long)nc[1] * nc[2], nc[2], 1};
  PermArgs P;
  P.off = 0;
  for (int j = 0; j < 3; ++j) {
    const int a = O.perm[j];
    P.d[a] = nu[a];
    P.s[a] = O.flip[j] ? -sc[j] : sc[j];
    if (O.flip[j]) P.off += (long long)(nc[j] - 1) * sc[j];
  }
  launch_permute(src, dst, P, st);
}

}  // namespace unires
