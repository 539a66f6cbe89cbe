// Stand-alone driver of the fused Tucker chain kernels (csrc/sc_kernels_tkchain.h, round 5) at BASELINE configs[2]:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] scripts/ubench_tkc.hip -o scripts/ubench_tkc.bin
//   ubench_tkc.bin [name] [abl] [n_wg] [reps]
// abl (as SC_TK_ABL): 1 = no k loops, 2 = no tile stores.  Prints forward / backward us and a checksum of yhat / gxhat.
#include "../neuraloperator_amd/csrc/sc_kernels_tkchain.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <vector>
// ------------------------------------------------------------------------------------------
// forward, second form (round 5; lives in this harness only: measured, not taken -- profiles/r05_tkchain_ab.txt): a work unit = (four modes, a CHUNK of g.B batch rows
// out of g.Btot), T3 as the B operand straight from its mode-major copy (no LDS) -- 53 KB of LDS at 16 rows: three
// workgroups per compute unit, whose staging / emit phases overlap each other's k loops.
// ------------------------------------------------------------------------------------------
struct TkcLayout2 {
  int ldi, ldo, ld1, ld2, PA, PB, oA, oB, total;
};
SC_TK_HD TkcLayout2 tkc_layout2(const int B, const int Ci, const int Co, const int R1, const int R2) {
  TkcLayout2 L;
  L.ldi = tkm_ld_rows(Ci); L.ldo = tkm_ld_rows(Co); L.ld1 = tkm_ld_rows(R1); L.ld2 = tkm_ld_rows(R2);
  L.PA = B * tkc_max(tkc_max(L.ldi, L.ld2), L.ldo);         // X -> t -> yhat planes (wave-private)
  L.PB = B * L.ld1;                                         // z planes
  L.oA = 0; L.oB = 4 * L.PA; L.total = L.oB + 4 * L.PB;
  return L;
}
template <int PFX, int OCC, int RT>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, OCC)
k_tkc_fwd2(TkcArgs g, int Btot) {
  SC_DYN_SHARED(cf32, lds);
  const TkcLayout2 L = tkc_layout2(g.B, g.Ci, g.Co, g.R1, g.R2);
  const int lane = SC_TID & 63, w = SC_UNIFORM(SC_TID >> 6);
  cf32* XA = lds + L.oA;
  cf32* ZB = lds + L.oB;
  const int R12 = g.R1 * g.R2;
  const int n_ch = Btot / g.B;                                // chunks of the batch
  for (int round = 0;; ++round) {
    const int unit = tkc_tile(round, g.n_wg);
    if (unit >= g.n_tiles * n_ch) break;
    const int tile = unit / n_ch, ch = unit - tile * n_ch;
    const int64_t m0 = (int64_t)tile * 4;
    const int64_t b0 = (int64_t)ch * g.B;
    {
      const int tid = sc_opaque(SC_TID);
      sc_f4 vx[PFX];
      tkc_fetch<PFX>(g.xhat + b0 * g.Ci * g.M, g.B * g.Ci, g.M, m0, tid, vx);
      tkc_plant<PFX>(XA, L.PA, L.ldi, g.B * g.Ci, g.Ci, g.inv_ci, tid, vx);
    }
    SC_SYNC();
    cf32* Xw = XA + w * L.PA;
    cf32* Zw = ZB + w * L.PB;
    const cf32* T3w = g.t3m + (m0 + w) * R12;               // [f][g] of this wave's mode, global
#pragma unroll 1
    for (int rt = 0; rt < RT; ++rt) {
      TkAcc a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_zero(a[k]);
      tk_multi<3, true, false, false, false, true>(Xw, L.ldi, 1, g.u_in, g.R1, 1, 16 * rt, 0, g.B, g.R1, g.Ci, lane, a, g.abl);
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_store<false>(a[k], Zw, L.ld1, 16 * rt, 16 * k, g.B, g.R1, lane, g.abl);
    }
    SC_WAVE_SYNC();
#pragma unroll 1
    for (int rt = 0; rt < RT; ++rt) {
      TkAcc a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_zero(a[k]);
      tk_multi<3, true, false, false, false, true>(Zw, L.ld1, 1, T3w, g.R2, 1, 16 * rt, 0, g.B, g.R2, g.R1, lane, a, g.abl);
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_store<false>(a[k], Xw, L.ld2, 16 * rt, 16 * k, g.B, g.R2, lane, g.abl);   // t over X
    }
    SC_SYNC();
    { const int tid = sc_opaque(SC_TID);
    tkc_emit(ZB, L.PB, L.ld1, g.B * g.R1, g.R1, g.inv_r1, g.z + b0 * g.R1 * g.M, g.M, m0, tid);
    tkc_emit(XA, L.PA, L.ld2, g.B * g.R2, g.R2, g.inv_r2, g.t + b0 * g.R2 * g.M, g.M, m0, tid); }
    SC_SYNC();
    // yhat_w = t_w u_out^T: all tiles in registers first (t_w and the yhat plane share the wave's region)
    {
      TkAcc a[RT][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tk_zero(a[rt][k]);
        tk_multi<4, true, false, false, false, true>(Xw, L.ld2, 1, g.u_out, 1, g.R2, 16 * rt, 0, g.B, g.Co, g.R2, lane, a[rt], g.abl);
      }
      SC_WAVE_SYNC();
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
          for (int k = 0; k < 4; ++k) tk_store<false>(a[rt][k], Xw, L.ldo, 16 * rt, 16 * k, g.B, g.Co, lane, g.abl);
        }
    }
    SC_SYNC();
    tkc_emit(XA, L.PA, L.ldo, g.B * g.Co, g.Co, g.inv_co, g.yhat + b0 * g.Co * g.M, g.M, m0, sc_opaque(SC_TID));
    SC_SYNC();
  }
}

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
  // second form: TKC_BT batch rows per unit (default 16), TKC_OCC workgroups per compute unit
#ifndef TKC_BT
#define TKC_BT 16
#endif
#ifndef TKC_OCC
#define TKC_OCC 3
#endif
  TkcArgs g2 = g; g2.B = TKC_BT; g2.n_wg = getenv("TKC_WGS") ? atoi(getenv("TKC_WGS")) : 256 * TKC_OCC;
  { const int units = g2.n_tiles * (B / TKC_BT); if (g2.n_wg > units) g2.n_wg = units; }
  const size_t lf2 = (size_t)tkc_layout2(TKC_BT, Ci, Co, R1, R2).total * 8;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_tkc_fwd2<(2 * TKC_BT * 64 + 255) / 256, TKC_OCC, (TKC_BT + 15) / 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf2);
  auto fwd2 = [&] { hipLaunchKernelGGL((k_tkc_fwd2<(2 * TKC_BT * 64 + 255) / 256, TKC_OCC, (TKC_BT + 15) / 16>), dim3(g2.n_wg), dim3(256), lf2, 0, g2, B); };
  auto fwd = [&] { hipLaunchKernelGGL((k_tkc_fwd<16, 18>), dim3(n_wg), dim3(256), lf, 0, g); };
  auto bwd = [&] { hipLaunchKernelGGL((k_tkc_bwd<16, 12, 18>), dim3(n_wg), dim3(256), lb, 0, g); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, int n) {
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n; };
  for (int i = 0; i < 100; ++i) { fwd(); bwd(); }
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return 1; }
  const float tf = timeit(fwd, reps), tb = timeit(bwd, reps);
  std::vector<float> hy0(8192);
  hipMemcpy(hy0.data(), g.yhat + 54321, 8192 * 4, hipMemcpyDeviceToHost);
  hipMemset(g.yhat, 0, (size_t)B * Co * M * 8);
  for (int i = 0; i < 20; ++i) fwd2();
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: fwd2 failed: %s\n", name, hipGetErrorString(hipGetLastError())); return 1; }
  const float tf2 = timeit(fwd2, reps);
  { std::vector<float> h2(8192); hipMemcpy(h2.data(), g.yhat + 54321, 8192 * 4, hipMemcpyDeviceToHost);
    double d = 0, n = 0; for (int i = 0; i < 8192; ++i) { d += (double)(h2[i] - hy0[i]) * (h2[i] - hy0[i]); n += (double)hy0[i] * hy0[i]; }
    printf("%-28s fwd2 (BT %d, occ %d, wgs %d, lds %zu): %7.1f us   rel diff to fwd %.2e\n", name, TKC_BT, TKC_OCC, g2.n_wg, lf2, tf2, n > 0 ? sqrt(d / n) : -1.0); }
  std::vector<float> hy(8192), hx(8192);
  hipMemcpy(hy.data(), g.yhat + 54321, 8192 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), g.gxhat + 54321, 8192 * 4, hipMemcpyDeviceToHost);
  double cy = 0, cx = 0; for (float v : hy) cy += (double)v * v; for (float v : hx) cx += (double)v * v;
  printf("%-28s abl %d wgs %3d lds %zu/%zu: fwd %7.1f us  bwd %7.1f us   checksums %.9e %.9e\n", name, abl, n_wg, lf, lb, tf, tb, cy, cx);
  return 0;
}
