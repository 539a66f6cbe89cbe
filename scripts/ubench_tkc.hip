// Stand-alone driver of the fused Tucker chain kernels (csrc/sc_kernels_tkchain.h, round 5) at BASELINE configs[2]:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] scripts/ubench_tkc.hip -o scripts/ubench_tkc.bin
//   ubench_tkc.bin [name] [abl] [n_wg] [reps]
// abl (as SC_TK_ABL): 1 = no k loops, 2 = no tile stores.  Prints forward / backward us and a checksum of yhat / gxhat.
#include "../neuraloperator_amd/csrc/sc_kernels_tkchain.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
static uint32_t inv32(int64_t n) { return (uint32_t)((((uint64_t)1 << 32) + (uint64_t)n - 1) / (uint64_t)n); }
int main(int argc, char** argv) {
  const char* name = argc > 1 ? argv[1] : "default";
  const int abl = argc > 2 ? atoi(argv[2]) : 0;
  const int B = 32, Ci = 64, Co = 64, R1 = 36, R2 = 36;
  const int64_t M = 2112;
  int n_wg = argc > 3 ? atoi(argv[3]) : 256;
  const int reps = argc > 4 ? atoi(argv[4]) : 100;
  auto dev = [&](size_t n, bool fill) {
    cf32* p; hipMalloc(&p, n * 8);
    if (fill) {
      std::vector<float> h(2 * n); static unsigned s = 12345u;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
      hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice);
    } else hipMemset(p, 0, n * 8);
    return p; };
  TkcArgs g; std::memset((void*)&g, 0, sizeof(g));
  g.xhat = dev(B * Ci * M, true); g.u_in = dev(Ci * R1, true); g.t3m = dev(M * R1 * R2, true); g.u_out = dev(Co * R2, true);
  g.z = dev(B * R1 * M, false); g.t = dev(B * R2 * M, false); g.yhat = dev(B * Co * M, false);
  g.zin = g.z; g.tin = g.t; g.gy = dev(B * Co * M, true); g.gxhat = dev(B * Ci * M, false); g.gt3m = dev(M * R1 * R2, false);
  g.partial = dev((size_t)256 * (Co * R2 + Ci * R1), false);
  g.B = B; g.Ci = Ci; g.Co = Co; g.R1 = R1; g.R2 = R2; g.M = M; g.n_tiles = (int)(M / 4);
  if (n_wg > g.n_tiles) n_wg = g.n_tiles;
  g.n_wg = n_wg; g.abl = abl;
  g.inv_ci = inv32(Ci); g.inv_co = inv32(Co); g.inv_r1 = inv32(R1); g.inv_r2 = inv32(R2); g.inv_r12 = inv32(R1 * R2);
  const size_t lf = (size_t)tkc_layout(B, Ci, Co, R1, R2, false).total * 8, lb = (size_t)tkc_layout(B, Ci, Co, R1, R2, true).total * 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_tkc_fwd<16, 18>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_tkc_bwd<16, 12, 18>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
  auto fwd = [&] { hipLaunchKernelGGL((k_tkc_fwd<16, 18>), dim3(n_wg), dim3(256), lf, 0, g); };
  auto bwd = [&] { hipLaunchKernelGGL((k_tkc_bwd<16, 12, 18>), dim3(n_wg), dim3(256), lb, 0, g); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, int n) {
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n; };
  for (int i = 0; i < 100; ++i) { fwd(); bwd(); }
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return 1; }
  const float tf = timeit(fwd, reps), tb = timeit(bwd, reps);
  std::vector<float> hy(8192), hx(8192);
  hipMemcpy(hy.data(), g.yhat + 54321, 8192 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), g.gxhat + 54321, 8192 * 4, hipMemcpyDeviceToHost);
  double cy = 0, cx = 0; for (float v : hy) cy += (double)v * v; for (float v : hx) cx += (double)v * v;
  printf("%-28s abl %d wgs %3d lds %zu/%zu: fwd %7.1f us  bwd %7.1f us   checksums %.9e %.9e\n", name, abl, n_wg, lf, lb, tf, tb, cy, cx);
  return 0;
}
