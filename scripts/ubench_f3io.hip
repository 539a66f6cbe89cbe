// A-B driver for the generation-3 fused FFT kernels (round 3): forward and inverse kernel of ONE build variant
// (compile-time switches of sc_kernels_fft3.h), timed back to back at settled clocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DABL_NAME='"name"' [-DSC_F3_...] scripts/ubench_f3ab.hip -o f3ab_name
// Prints: name, forward us, inverse us, (forward -> inverse alternating) us per pair.
#include "../neuraloperator_amd/csrc/sc_kernels_fft3.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef F3_IO
#define F3_IO float
#endif
static void f3_set(float& d, float v) { d = v; }
static void f3_set(sc_bf16& d, float v) { uint32_t u; memcpy(&u, &v, 4); d.v = (uint16_t)(u >> 16); }
static float f3_get(float v) { return v; }
static float f3_get(sc_bf16 v) { uint32_t u = (uint32_t)v.v << 16; float f; memcpy(&f, &u, 4); return f; }
#ifndef ABL_NAME
#define ABL_NAME "default"
#endif
int main(int argc, char** argv) {
  const int NIMG = 2048, H = 256, C = 64;
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  F3_IO *x, *y; float *bias; cf32 *xh, *tW, *tH;
  hipMalloc(&x, (size_t)NIMG * H * 256 * sizeof(F3_IO)); hipMalloc(&y, (size_t)NIMG * H * 256 * sizeof(F3_IO));
  hipMalloc(&xh, (size_t)NIMG * 64 * 33 * 8); hipMalloc(&bias, C * 4);
  std::vector<F3_IO> hx((size_t)NIMG * H * 256);
  unsigned s = 12345u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; f3_set(v, (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22)); }
  hipMemcpy(x, hx.data(), hx.size() * sizeof(F3_IO), hipMemcpyHostToDevice);
  hipMemset(bias, 0, C * 4);
  std::vector<void*> owned;
  fft2d_upload(&owned, 256, &tW); fft2d_upload(&owned, H, &tH);
  auto fwd = [&] { hipLaunchKernelGGL((k_fft2d_fwd3<256, F3_IO>), dim3(NIMG), dim3(256), 0, 0, (const F3_IO*)x, xh,
                                      (const cf32*)tW, (const cf32*)tH, 64, 33, 1.f / 65536.f, 1.f / 65536.f, F3Shard{0, 0}); };
  const int IGRID = getenv("F3_INV_GRID") ? atoi(getenv("F3_INV_GRID")) : SC_F3_INV_WGS * sc_cu_count();
  auto inv = [&] { hipLaunchKernelGGL((k_fft2d_inv3<256, F3_IO, 0>), dim3(IGRID), dim3(256), 0, 0, (const cf32*)xh, y,
                                      (const float*)bias, C, (const cf32*)tW, (const cf32*)tH, 64, 33, 1.f, 2.f,
                                      (const F3_IO*)nullptr, (F3_IO*)nullptr, F3Shard{0, 0}, (int64_t)NIMG, IGRID); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, int n) {
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n; };
  for (int i = 0; i < 150; ++i) { fwd(); inv(); }            // settle the clocks (~35 ms of load)
  hipDeviceSynchronize();
  const float tf = timeit(fwd, reps);
  const float ti = timeit(inv, reps);
  const float tp = timeit([&] { fwd(); inv(); }, reps / 2);
  // checksum of the round trip (x -> xhat -> y with all 64 x 33 modes kept is NOT the identity; just a fingerprint
  // that must agree between variants of one kernel)
  std::vector<F3_IO> hy(4096);
  hipMemcpy(hy.data(), y + 12345, 4096 * sizeof(F3_IO), hipMemcpyDeviceToHost);
  double cs = 0; for (auto v : hy) cs += (double)f3_get(v) * f3_get(v);
  printf("%-26s fwd %7.2f us   inv %7.2f us   fwd+inv %7.2f us   checksum %.9e\n", ABL_NAME, tf, ti, tp, cs);
  return 0;
}
