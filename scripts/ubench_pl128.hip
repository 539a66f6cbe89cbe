// A-B driver for the 128 x 128 plane kernels (FNO3d 128^3: 32768 planes per transform): forward and inverse kernel of
// ONE build of sc_kernels_plane.h, timed back to back at settled clocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DABL_NAME='"name"' [-DPL_OLD_SIG] scripts/ubench_pl128.hip -o scripts/pl128_name.bin
// PL_OLD_SIG: the one-plane-per-workgroup kernels the library ships (a persistent variant measured slower: profiles/r03s2_pl128_persistent_ab.txt).
// env PL_GRID: workgroups of the persistent kernels (default SC_PL_WGS x compute units).
#include "../neuraloperator_amd/csrc/sc_kernels_plane.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef ABL_NAME
#define ABL_NAME "default"
#endif
int main(int argc, char** argv) {
  const int NPL = 8 * 32 * 128, K0 = 32, J = 17, C = 32;
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  float *x, *y, *bias, *cs; cf32 *xh, *t128;
  hipMalloc(&x, (size_t)NPL * 128 * 128 * 4); hipMalloc(&y, (size_t)NPL * 128 * 128 * 4);
  hipMalloc(&xh, (size_t)NPL * K0 * J * 8); hipMalloc(&bias, C * 4); hipMalloc(&cs, 32 * 4);
  {
    std::vector<float> hx((size_t)NPL * 128 * 128);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  }
  std::vector<float> hcs(32, 1.f / 16384.f);
  hipMemcpy(cs, hcs.data(), 32 * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, C * 4);
  std::vector<void*> owned;
  fft2d_upload(&owned, 128, &t128);
#ifdef PL_OLD_SIG
  const int PPW = getenv("PL_PPW") ? atoi(getenv("PL_PPW")) : 1;      // planes per workgroup of the forward kernel
  auto fwd = [&] { hipLaunchKernelGGL(k_pl128_fwd, dim3((NPL + PPW - 1) / PPW), dim3(256), 0, 0, (const float*)x, xh, (const cf32*)t128,
                                      (const float*)cs, K0, J, (int64_t)NPL, PPW); };
  auto inv = [&] { hipLaunchKernelGGL(k_pl128_inv, dim3(NPL), dim3(256), 0, 0, (const cf32*)xh, y, (const cf32*)t128,
                                      (const float*)cs, (const float*)bias, (int64_t)128, C, K0, J); };
#else
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int GRID = getenv("PL_GRID") ? atoi(getenv("PL_GRID")) : SC_PL_WGS * cus;
  auto fwd = [&] { hipLaunchKernelGGL(k_pl128_fwd, dim3(GRID), dim3(256), 0, 0, (const float*)x, xh, (const cf32*)t128,
                                      (const float*)cs, K0, J, (int64_t)NPL, GRID); };
  auto inv = [&] { hipLaunchKernelGGL(k_pl128_inv, dim3(GRID), dim3(256), 0, 0, (const cf32*)xh, y, (const cf32*)t128,
                                      (const float*)cs, (const float*)bias, (int64_t)128, C, K0, J, (int64_t)NPL, GRID); };
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, int n) {
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n; };
  for (int i = 0; i < 30; ++i) { fwd(); inv(); }             // settle the clocks
  hipDeviceSynchronize();
  const float tf = timeit(fwd, reps);
  const float ti = timeit(inv, reps);
  std::vector<float> hy(4096);
  hipMemcpy(hy.data(), y + 1234567, 4096 * 4, hipMemcpyDeviceToHost);
  double cs2 = 0; for (float v : hy) cs2 += (double)v * v;
  std::vector<float> hz(4096);
  hipMemcpy(hz.data(), (float*)xh + 7654321, 4096 * 4, hipMemcpyDeviceToHost);
  double cs3 = 0; for (float v : hz) cs3 += (double)v * v;
  const double gb = (double)NPL * 128 * 128 * 4 / 1e9;
  printf("%-22s fwd %7.1f us (%5.2f TB/s)   inv %7.1f us (%5.2f TB/s)   checksums %.9e %.9e\n", ABL_NAME, tf, gb / tf * 1e3,
         ti, gb / ti * 1e3, cs2, cs3);
  return 0;
}
