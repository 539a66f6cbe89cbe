// Bench driver: generation-3 forward FFT kernel with compile-time ablations.
#include "../neuraloperator_amd/csrc/sc_kernels_fft3.h"
#include <cstdio>
#include <vector>
int main() {
  const int NIMG = 2048, H = 256;
  float* x; cf32 *xh, *tW, *tH;
  hipMalloc(&x, (size_t)NIMG * H * 256 * 4); hipMalloc(&xh, (size_t)NIMG * 64 * 33 * 8);
  hipMemset(x, 1, (size_t)NIMG * H * 256 * 4);
  std::vector<void*> owned; Fft2dPlan fp;
  fft2d_upload(&owned, 256, &tW); fft2d_upload(&owned, H, &tH);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&] { hipLaunchKernelGGL((k_fft2d_fwd3<256, float>), dim3(NIMG), dim3(256), 0, 0, (const float*)x, xh, (const cf32*)tW, (const cf32*)tH, 64, 33, 1.f, 1.f, F3Shard{0, 0}); };
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s: %7.1f us\n", ABL_NAME, ms * 100.f);
  return 0;
}
