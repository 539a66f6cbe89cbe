// A-B driver for the small-extent streaming contraction (sc_kernels_sb.h) at BASELINE configs[4]: B = 4 rows against
// 128 x 128 x 33024 complex weights (4.33 GB).  fwd (sum over i), gx (sum over o, conj), gw (R = 4, weight-sized result).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DABL_NAME='"name"' [-DSB_ST=n ...] scripts/ubench_sb.hip -o scripts/sb_name.bin
#include "../neuraloperator_amd/csrc/sc_kernels_sb.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef ABL_NAME
#define ABL_NAME "default"
#endif
#ifndef SB_ST
#define SB_ST 3
#endif
#ifndef SB_ST_GW
#define SB_ST_GW 2
#endif
#ifndef SB_QT
#define SB_QT 4
#endif
#ifndef SB_ST_BWD
#define SB_ST_BWD 3
#endif
#ifndef SB_BATCH
#define SB_BATCH 4
#endif
#ifndef SB_PT
#define SB_PT 4
#endif
#ifndef SB_PT_GW
#define SB_PT_GW 4
#endif
#ifndef SB_QT_GW
#define SB_QT_GW 4
#endif
int main() {
  const int64_t Bn = SB_BATCH, C = 128, M = 33024;
  cf32 *W, *gW, *xh, *gh, *yh;
  hipMalloc(&W, C * C * M * 8); hipMalloc(&gW, C * C * M * 8);
  hipMalloc(&xh, Bn * C * M * 8); hipMalloc(&gh, Bn * C * M * 8); hipMalloc(&yh, Bn * C * M * 8);
  {
    std::vector<float> h((size_t)(64 << 20));
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
    for (size_t off = 0; off < (size_t)C * C * M * 8; off += h.size() * 4)
      hipMemcpy((char*)W + off, h.data(), std::min(h.size() * 4, (size_t)C * C * M * 8 - off), hipMemcpyHostToDevice);
    hipMemcpy(xh, h.data(), Bn * C * M * 8, hipMemcpyHostToDevice);
    hipMemcpy(gh, h.data() + 12345, Bn * C * M * 8, hipMemcpyHostToDevice);
  }
  auto args = [&](int64_t P, int64_t Q, int64_t R, int64_t a_sp, int64_t a_sr, int64_t b_sr, int64_t b_sq, int64_t c_sp,
                  int64_t c_sq, int PT, int QT, int WP, int WQ, int nt_c) {
    SbGemmArgs g;
    g.P = P; g.Q = Q; g.R = R; g.M = M; g.a_sp = a_sp; g.a_sr = a_sr; g.b_sr = b_sr; g.b_sq = b_sq; g.c_sp = c_sp; g.c_sq = c_sq;
    g.n_mt = (int)((M + 127) / 128); g.n_pt = (int)((P + PT - 1) / PT); g.n_qt = (int)((Q + QT - 1) / QT);
    const int64_t total = (int64_t)g.n_mt * ((g.n_pt + WP - 1) / WP) * ((g.n_qt + WQ - 1) / WQ);
    g.per_xcd = (int)((total + 7) / 8); g.nt_a = g.n_qt == 1; g.nt_b = g.n_pt == 1; g.nt_c = nt_c;
    return g;
  };
  // fwd: yhat[b,o,m] = sum_i xhat[b,i,m] W[i,o,m];  gx: gxhat[b,i,m] = sum_o ghat[b,o,m] conj(W[i,o,m]);  gw[i,o,m] = sum_b conj(xhat[b,i,m]) ghat[b,o,m]
  const SbGemmArgs gf = args(Bn, C, C, C * M, M, C * M, M, C * M, M, SB_PT, SB_QT, 1, 4, 0);
  const SbGemmArgs gx = args(Bn, C, C, C * M, M, M, C * M, C * M, M, SB_PT, SB_QT, 1, 4, 0);
  const SbGemmArgs gw = args(C, C, Bn, M, C * M, C * M, M, C * M, M, SB_PT_GW, SB_QT_GW, 2, 2, 1);
  auto fwd = [&] { hipLaunchKernelGGL((k_modegemm_sb<SB_PT, SB_QT, SB_ST, 1, 1, 4, false, false>), dim3(8 * gf.per_xcd), dim3(256), 0, 0, gf, (const cf32*)xh, (const cf32*)W, yh); };
  auto bgx = [&] { hipLaunchKernelGGL((k_modegemm_sb<SB_PT, SB_QT, SB_ST, 1, 1, 4, false, true>), dim3(8 * gx.per_xcd), dim3(256), 0, 0, gx, (const cf32*)gh, (const cf32*)W, yh); };
  auto bgw = [&] { hipLaunchKernelGGL((k_modegemm_sb<SB_PT_GW, SB_QT_GW, SB_ST_GW, 1, 2, 2, true, false>), dim3(8 * gw.per_xcd), dim3(256), 0, 0, gw, (const cf32*)xh, (const cf32*)gh, gW); };
  cf32* gxh; hipMalloc(&gxh, Bn * C * M * 8);
  SbBwdArgs gb;
  gb.B = Bn; gb.Ci = C; gb.Co = C; gb.M = M; gb.x_sb = C * M; gb.x_si = M; gb.g_sb = C * M; gb.g_so = M; gb.w_si = C * M; gb.w_so = M;
  gb.gw_si = C * M; gb.gw_so = M; gb.gx_sb = C * M; gb.gx_si = M; gb.n_mt = (int)((M + 127) / 128); gb.n_itg = (int)((C / 4 + 3) / 4);
  gb.per_xcd = (int)(((int64_t)gb.n_mt * gb.n_itg + 7) / 8); gb.nt_gw = 1;
  auto bwd = [&] { hipLaunchKernelGGL((k_modegemm_sb_bwd<SB_BATCH, 4, SB_ST_BWD>), dim3(8 * gb.per_xcd), dim3(256), 0, 0, gb, (const cf32*)xh, (const cf32*)gh, (const cf32*)W, gW, gxh); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto f, int n) {
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n; };
  const float tf = timeit(fwd, 6), tx = timeit(bgx, 6), tw = timeit(bgw, 6);
  std::vector<float> hy(4096), hx(4096), hy2(4096), hx2(4096);
  hipMemcpy(hy.data(), (float*)gW + 123456789, 4096 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), (float*)yh + 1234567, 4096 * 4, hipMemcpyDeviceToHost);          // gx result of the separate launch
  double cs = 0; for (float v : hy) cs += (double)v * v;
  hipMemset(gW, 0, C * C * M * 8);
  const float tb = timeit(bwd, 6);
  hipMemcpy(hy2.data(), (float*)gW + 123456789, 4096 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx2.data(), (float*)gxh + 1234567, 4096 * 4, hipMemcpyDeviceToHost);
  int same = 1; for (int i = 0; i < 4096; ++i) same &= (hy[i] == hy2[i]) && (hx[i] == hx2[i]);
  {
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto both = [&] {
      hipLaunchKernelGGL((k_modegemm_sb<SB_PT, SB_QT, SB_ST, 1, 1, 4, false, true>), dim3(8 * gx.per_xcd), dim3(256), 0, s1, gx, (const cf32*)gh, (const cf32*)W, yh);
      hipLaunchKernelGGL((k_modegemm_sb<SB_PT_GW, SB_QT_GW, SB_ST_GW, 1, 2, 2, true, false>), dim3(8 * gw.per_xcd), dim3(256), 0, s2, gw, (const cf32*)xh, (const cf32*)gh, gW);
    };
    hipDeviceSynchronize();
    for (int i = 0; i < 3; ++i) both();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0); hipStreamWaitEvent(s1, e0, 0); hipStreamWaitEvent(s2, e0, 0);
    for (int i = 0; i < 6; ++i) both();
    hipEvent_t f1, f2; hipEventCreate(&f1); hipEventCreate(&f2);
    hipEventRecord(f1, s1); hipEventRecord(f2, s2); hipStreamWaitEvent(0, f1, 0); hipStreamWaitEvent(0, f2, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("two streams, gx || gw: %7.1f us per pair\n", ms * 1e3f / 6);
  }
  printf("%-22s fwd %7.1f us   gx %7.1f us   gw %7.1f us   gx+gw fused %7.1f us (bits equal: %d)   checksum %.9e\n", ABL_NAME, tf, tx, tw, tb, same, cs);
  return 0;
}
