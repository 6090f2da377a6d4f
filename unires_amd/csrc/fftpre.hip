// fftpre.hip - see fftpre.hpp.  rocFFT (through hipFFT) does the two 3-D real transforms;
// the diagonal scaling is one pass over the half-spectrum with the per-axis eigenvalues in LDS.
#include <math.h>

#include <vector>

#include "fftpre.hpp"

namespace unires {

// freq[i,j,k] *= 1 / (N (a + cx lx[i] + cy ly[j] + cz lz[k]))   (unnormalised inverse FFT)
__global__ void __launch_bounds__(kBlock)
    k_fft_scale(float2 *__restrict__ f, Dim3i d, int nzh, const float *__restrict__ lx,
                const float *__restrict__ ly, const float *__restrict__ lz, float a, float cx,
                float cy, float cz, float inv_n) {
  const int k = blockIdx.x * kWave + threadIdx.x, j = blockIdx.y * (kBlock / kWave) + threadIdx.y,
            i = blockIdx.z;
  if (k >= nzh || j >= d.y) return;
  const float den = a + cx * lx[i] + cy * ly[j] + cz * lz[k];
  const float w = inv_n / den;
  const size_t idx = ((size_t)i * d.y + j) * nzh + k;
  float2 v = f[idx];
  v.x *= w, v.y *= w;
  f[idx] = v;
}

int fftpre_setup(FftPre &F, Dim3i d) {
  if (F.have_plans && F.d.x == d.x && F.d.y == d.y && F.d.z == d.z) return 0;
  fftpre_destroy(F);
  F.d = d;
  const int nzh = d.z / 2 + 1;
  if (hipfftPlan3d(&F.fwd, d.x, d.y, d.z, HIPFFT_R2C) != HIPFFT_SUCCESS) return 1;
  if (hipfftPlan3d(&F.inv, d.x, d.y, d.z, HIPFFT_C2R) != HIPFFT_SUCCESS) return 1;
  F.have_plans = true;
  if (hipMalloc((void **)&F.freq, (size_t)d.x * d.y * nzh * sizeof(float2)) != hipSuccess) return 2;
  if (hipMalloc((void **)&F.z, d.numel() * sizeof(float)) != hipSuccess) return 2;
  const int n[3] = {d.x, d.y, d.z};
  for (int a = 0; a < 3; ++a) {
    std::vector<float> h((size_t)n[a]);
    for (int k = 0; k < n[a]; ++k) h[k] = (float)(2.0 - 2.0 * cos(2.0 * M_PI * (double)k / (double)n[a]));
    if (hipMalloc((void **)&F.lam[a], h.size() * sizeof(float)) != hipSuccess) return 2;
    if (hipMemcpy(F.lam[a], h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return 2;
  }
  return 0;
}

void fftpre_destroy(FftPre &F) {
  if (F.have_plans) {
    (void)hipfftDestroy(F.fwd);
    (void)hipfftDestroy(F.inv);
    F.have_plans = false;
  }
  if (F.freq) (void)hipFree(F.freq);
  if (F.z) (void)hipFree(F.z);
  for (int a = 0; a < 3; ++a)
    if (F.lam[a]) (void)hipFree(F.lam[a]);
  F.freq = nullptr, F.z = nullptr, F.lam[0] = F.lam[1] = F.lam[2] = nullptr;
}

int fftpre_apply(FftPre &F, const float *in, float *out, hipStream_t st) {
  if (!F.have_plans) return 1;
  const Dim3i d = F.d;
  const int nzh = d.z / 2 + 1;
  if (hipfftSetStream(F.fwd, st) != HIPFFT_SUCCESS || hipfftSetStream(F.inv, st) != HIPFFT_SUCCESS)
    return 1;
  if (hipfftExecR2C(F.fwd, const_cast<float *>(in), reinterpret_cast<hipfftComplex *>(F.freq)) !=
      HIPFFT_SUCCESS)
    return 1;
  const dim3 grid((nzh + kWave - 1) / kWave, (d.y + 3) / 4, d.x), block(kWave, kBlock / kWave);
  hipLaunchKernelGGL(k_fft_scale, grid, block, 0, st, F.freq, d, nzh, F.lam[0], F.lam[1], F.lam[2],
                     F.a, F.c[0], F.c[1], F.c[2], 1.f / (float)d.numel());
  if (hipfftExecC2R(F.inv, reinterpret_cast<hipfftComplex *>(F.freq), out) != HIPFFT_SUCCESS) return 1;
  return 0;
}

}  // namespace unires
