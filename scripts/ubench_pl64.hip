// A-B driver for the 64 x 64 plane kernels: forward and inverse kernel of ONE build of sc_kernels_plane64.h, timed back
// to back at settled clocks.   usage: pl64.bin [reps] ;  env PL_PLANES (default 16384 = FNO3d 64^3, B = 8, C = 32),
// PL_PPW (planes per workgroup of the forward kernel)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DABL_NAME='"name"' scripts/ubench_pl64.hip -o scripts/pl64_name.bin
#include "../neuraloperator_amd/csrc/sc_kernels_plane64.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef ABL_NAME
#define ABL_NAME "default"
#endif
int main(int argc, char** argv) {
  const int NPL = getenv("PL_PLANES") ? atoi(getenv("PL_PLANES")) : 16384, K0 = 32, J = 17, C = 32;
  const int PPW = getenv("PL_PPW") ? atoi(getenv("PL_PPW")) : 1;
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  float *x, *y, *bias, *cs; cf32 *xh, *t64;
  hipMalloc(&x, (size_t)NPL * 4096 * 4); hipMalloc(&y, (size_t)NPL * 4096 * 4);
  hipMalloc(&xh, (size_t)NPL * K0 * J * 8); hipMalloc(&bias, C * 4); hipMalloc(&cs, 32 * 4); hipMalloc(&t64, 64 * 8);
  {
    std::vector<float> hx((size_t)NPL * 4096);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<cf32> ht(64);
    for (int m = 0; m < 64; ++m) ht[m] = cf_make((float)cos(-2.0 * M_PI * m / 64), (float)sin(-2.0 * M_PI * m / 64));
    hipMemcpy(t64, ht.data(), 64 * 8, hipMemcpyHostToDevice);
  }
  std::vector<float> hcs(32, 1.f / 4096.f);
  hipMemcpy(cs, hcs.data(), 32 * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, C * 4);
  auto fwd = [&] { hipLaunchKernelGGL(k_pl64_fwd, dim3((NPL + PPW - 1) / PPW), dim3(256), 0, 0, (const float*)x, xh, (const cf32*)t64,
                                      (const float*)cs, K0, J, (int64_t)NPL, PPW); };
  auto inv = [&] { hipLaunchKernelGGL(k_pl64_inv, dim3(NPL), dim3(256), 0, 0, (const cf32*)xh, y, (const cf32*)t64,
                                      (const float*)cs, (const float*)bias, (int64_t)64, C, K0, J); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, int n) {
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n; };
  for (int i = 0; i < 200; ++i) { fwd(); inv(); }            // settle the clocks
  hipDeviceSynchronize();
  const float tf = timeit(fwd, reps);
  const float ti = timeit(inv, reps);
  std::vector<float> hy(4096);
  hipMemcpy(hy.data(), y + 1234567, 4096 * 4, hipMemcpyDeviceToHost);
  double cs2 = 0; for (float v : hy) cs2 += (double)v * v;
  std::vector<float> hz(4096);
  hipMemcpy(hz.data(), (float*)xh + 765432, 4096 * 4, hipMemcpyDeviceToHost);
  double cs3 = 0; for (float v : hz) cs3 += (double)v * v;
  const double gb = ((double)NPL * 4096 * 4 + (double)NPL * K0 * J * 8) / 1e9;
  printf("%-18s planes %6d ppw %d  fwd %7.1f us (%5.2f TB/s)   inv %7.1f us (%5.2f TB/s)   checksums %.9e %.9e\n", ABL_NAME, NPL, PPW,
         tf, gb / tf * 1e3, ti, gb / ti * 1e3, cs2, cs3);
  return 0;
}
