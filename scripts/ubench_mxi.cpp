// Repeatability harness for k_fft2d_inv_mx<H> (round 6: the H = 64 question of DESIGN 3.5).
// The kernels come from a code object built apart (scripts/mxi_kern.hip through scripts/mxi_variants.sh: compile-time
// switches of sc_kernels_fft3mx.h, compiler flags, or the generated assembly edited before it is assembled):
//   hipcc -O2 -std=c++17 scripts/ubench_mxi.cpp -o mxi          (host only)
//   ./mxi variant.hsaco H [launches] [grid (0 = 2 per unit)] [images]
// The reference result is the SAME kernel launched with one workgroup per compute unit (never seen to differ); every
// further launch is compared with it bit for bit on the device.  For the first differing images the program prints which
// output rows differ and the 2-D spectrum of the difference (which kept mode carries it), which tells a damaged
// spectrum entry (all 64 rows, one mode) from a damaged tile entry (one row) from a damaged output store.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../neuraloperator_amd/csrc/sc_kernels_fft3mx.h"      // host side: tables, F3Shard
static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: %s variant.hsaco H [launches] [grid] [images]\n", argv[0]); return 2; }
  const char* path = argv[1];
  const int H = atoi(argv[2]);
  const int reps = argc > 3 ? atoi(argv[3]) : 30;
  int grid = argc > 4 ? atoi(argv[4]) : 0;
  const int NIMG = argc > 5 ? atoi(argv[5]) : 2048;
  int MX = 64, MY = 33, C = 64;
  int cus = sc_cu_count();
  if (grid <= 0) grid = 2 * cus;
  hipModule_t mod; hipFunction_t fk, fc;
  if (hipModuleLoad(&mod, path) != hipSuccess) { printf("%s: cannot load\n", path); return 1; }
  char sym[256];
  snprintf(sym, sizeof sym, "_Z14k_fft2d_inv_mxILi%dEEvPK4cf32P7sc_bf16PKfiS2_S2_PKtiiff7F3Shardli", H);
  if (hipModuleGetFunction(&fk, mod, sym) != hipSuccess || hipModuleGetFunction(&fc, mod, "k_cmp") != hipSuccess) { printf("%s: kernels not found\n", path); return 1; }
  const char* ABL_NAME = strrchr(path, '/') ? strrchr(path, '/') + 1 : path;
  if (grid > NIMG) grid = NIMG;
  sc_bf16 *y, *y0; float* bias; cf32 *yh, *tW, *tH; uint16_t* tG; unsigned* bad;
  const size_t per = (size_t)H * 256;
  hipMalloc(&y, NIMG * per * 2); hipMalloc(&y0, NIMG * per * 2); hipMalloc(&yh, (size_t)NIMG * MX * MY * 8);
  hipMalloc(&bias, C * 4); hipMalloc(&bad, NIMG * 4);
  std::vector<float> hs((size_t)NIMG * MX * MY * 2);
  unsigned s = 12345u;
  for (auto& v : hs) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
  hipMemcpy(yh, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hb(C); for (int i = 0; i < C; ++i) hb[i] = 0.01f * i;
  hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice);
  std::vector<void*> owned; fft2d_upload(&owned, 256, &tW); fft2d_upload(&owned, H, &tH);
  std::vector<uint16_t> hg; fft3mxi_build_table(&hg);
  hipMalloc(&tG, hg.size() * 2); hipMemcpy(tG, hg.data(), hg.size() * 2, hipMemcpyHostToDevice);
  auto launch = [&](sc_bf16* out, int g) {
    float s_dc = 1.f, s_other = 2.f; F3Shard sh{0, 0}; int64_t n = NIMG;
    void* args[] = {&yh, &out, &bias, &C, &tW, &tH, &tG, &MX, &MY, &s_dc, &s_other, &sh, &n, &g};
    if (hipModuleLaunchKernel(fk, g, 1, 1, 256, 1, 1, 0, 0, args, nullptr) != hipSuccess) { printf("launch failed\n"); exit(1); } };
  cf32* dbg = nullptr;                                       // -DSC_MXI_X_DBG builds: the spectrum registers of every lane, stored by the kernel
  { hipDeviceptr_t gp; size_t gs;
    if (hipModuleGetGlobal(&gp, &gs, mod, "g_mxi_dbg") == hipSuccess) {
      hipMalloc(&dbg, (size_t)NIMG * 256 * 2 * 8 * 8); hipMemset(dbg, 0xff, (size_t)NIMG * 256 * 2 * 8 * 8);
      hipMemcpy((void*)gp, &dbg, 8, hipMemcpyHostToDevice); } }
  launch(y0, cus < NIMG ? cus : NIMG);                       // reference: one workgroup per unit
  hipDeviceSynchronize();
  int hist_q2[8] = {0}, hist_mu[8] = {0}, hist_cl[8] = {0}, hist_w[5] = {0}, hist_part[2] = {0};
  long by_first[2] = {0, 0}, by_num[8] = {0};
  int bad_launches = 0; long bad_images = 0; bool printed = false;
  std::vector<unsigned> hbad(NIMG);
  for (int it = 0; it < reps; ++it) {
    hipMemset(bad, 0, NIMG * 4);
    launch(y, grid);
    { int pi = (int)per; void* ca[] = {&y, &y0, &pi, &bad}; hipModuleLaunchKernel(fc, NIMG, 1, 1, 256, 1, 1, 0, 0, ca, nullptr); }
    hipMemcpy(hbad.data(), bad, NIMG * 4, hipMemcpyDeviceToHost);
    int nb = 0, first = -1, lo = NIMG, hi = -1;
    for (int i = 0; i < NIMG; ++i) if (hbad[i]) { ++nb; if (first < 0) first = i; if (i < lo) lo = i; if (i > hi) hi = i; }
    bad_launches += nb > 0; bad_images += nb;
    for (int i = 0; i < NIMG; ++i) if (hbad[i]) { by_first[(i % grid) < cus ? 0 : 1]++; by_num[(i / grid) < 8 ? i / grid : 7]++; }
    if (nb && dbg && !printed) {                            // which stored register differs from scale * input
      std::vector<float> hd((size_t)NIMG * 256 * 2 * 8 * 2);
      hipMemcpy(hd.data(), dbg, hd.size() * 4, hipMemcpyDeviceToHost);
      long nraw = 0, nsc = 0; int shown_d = 0;
      for (int i = 0; i < NIMG; ++i) for (int t = 0; t < 256; ++t) for (int q2 = 0; q2 < 8; ++q2) {
        const int wv = t >> 6, ln = t & 63, cl_ = ln >> 3, mu_ = ln & 7, c = 8 * wv + cl_, q = mu_ + 8 * q2, fx = q < 32 ? q : q - 64, row = fx + MX / 2;
        const float* se = &hs[(((size_t)i * MX + row) * MY + c) * 2]; const float sc = c == 0 ? 1.f : 2.f;
        const float* raw = &hd[((((size_t)i * 256 + t) * 2 + 0) * 8 + q2) * 2]; const float* scd = &hd[((((size_t)i * 256 + t) * 2 + 1) * 8 + q2) * 2];
        uint32_t rb[2]; memcpy(rb, raw, 8);
        const bool raw_stored = !(rb[0] == 0xffffffffu && rb[1] == 0xffffffffu);
        const bool braw = raw_stored && (raw[0] != se[0] || raw[1] != se[1]);
        const bool bsc = scd[0] != sc * se[0] || scd[1] != sc * se[1];
        nraw += braw; nsc += bsc;
        if ((braw || bsc) && shown_d < 24) { ++shown_d;
          printf("  dbg image %d (bad output: %s) wave %d lane %d q2 %d: input (%.6g, %.6g)  raw register (%.6g, %.6g)%s  scaled register (%.6g, %.6g) expected (%.6g, %.6g)\n",
                 i, hbad[i] ? "yes" : "no", wv, ln, q2, se[0], se[1], raw[0], raw[1], raw_stored ? "" : " [not stored]", scd[0], scd[1], sc * se[0], sc * se[1]); } }
      printf("  dbg: %ld raw registers and %ld scaled registers differ from the input in launch %d (%d bad images)\n", nraw, nsc, it, nb);
    }
    if (nb && !printed && getenv("MXI_VERBOSE")) {
      printed = true;
      printf("  launch %d: %d bad images, range %d..%d\n", it, nb, lo, hi);
      int shown = 0;
      std::vector<uint16_t> a(per), b(per);
      for (int i = 0; i < NIMG && shown < (getenv("MXI_HIST") ? atoi(getenv("MXI_HIST")) : 6); ++i) {
        if (!hbad[i]) continue;
        ++shown;
        hipMemcpy(a.data(), (uint16_t*)y + (size_t)i * per, per * 2, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), (uint16_t*)y0 + (size_t)i * per, per * 2, hipMemcpyDeviceToHost);
        std::vector<double> d(per); int rows_bad = 0, first_row = -1, last_row = -1; double dmax = 0;
        for (int r = 0; r < H; ++r) { bool rb = false;
          for (int c = 0; c < 256; ++c) { d[r * 256 + c] = (double)bf(a[r * 256 + c]) - bf(b[r * 256 + c]); if (a[r * 256 + c] != b[r * 256 + c]) rb = true; if (fabs(d[r * 256 + c]) > dmax) dmax = fabs(d[r * 256 + c]); }
          if (rb) { ++rows_bad; if (first_row < 0) first_row = r; last_row = r; } }
        // spectrum of the difference on the kept block: which (row frequency fx, column k) carries it
        double best = 0, tot = 0; int bfx = 0, bk = 0; double bre = 0, bim = 0; int ndam = 0;
        for (int q = 0; q < 64; ++q) { const int fx = q < 32 ? q : q - 64; if (fx >= H / 2 || fx < -H / 2) continue;
          for (int k = 0; k <= 32; ++k) { double re = 0, im = 0;
            for (int r = 0; r < H; ++r) for (int c = 0; c < 256; ++c) { const double v = d[r * 256 + c]; if (v == 0) continue;
              const double th = -6.283185307179586 * ((double)(fx * r) / H + (double)(k * c) / 256.0); re += v * cos(th); im += v * sin(th); }
            const double p = re * re + im * im; tot += p; if (p > best) { best = p; bfx = fx; bk = k; bre = re; bim = im; }
            // a damaged entry: the difference spectrum there is minus the scaled real (or imaginary) part of the input
            const int row = fx + MX / 2; const float* se = &hs[(((size_t)i * MX + row) * MY + k) * 2];
            const double sc = (k == 0 ? 1.0 : 2.0), dr = re / (H * 256.0), di = im / (H * 256.0);
            const bool lost_re = fabs(se[0]) > 0.05 && fabs(dr + sc * se[0]) < 0.02 + 0.02 * fabs(se[0]) && fabs(di) < 0.02;
            const bool lost_im = fabs(se[1]) > 0.05 && fabs(di + sc * se[1]) < 0.02 + 0.02 * fabs(se[1]) && fabs(dr) < 0.02;
            if (lost_re || lost_im) { ++ndam; const int mu = q & 7, q2 = q >> 3; hist_q2[q2]++; hist_mu[mu]++; hist_cl[k & 7]++; hist_w[k >> 3]++; hist_part[lost_im ? 1 : 0]++; } } }
        // the spectrum entry the kernel was given there
        const int row = bfx + MX / 2; const float* se = &hs[(((size_t)i * MX + row) * MY + bk) * 2];
        if (shown <= (getenv("MXI_SHOW") ? atoi(getenv("MXI_SHOW")) : 6)) printf("  image %4d (wg %3d, its image #%d): %u values differ, %d rows (%d..%d), max |d| %.3g; %d entries lost one part; strongest mode fx %d k %d: "
               "d-hat / (H W) = (%.4g, %.4g); input entry there (%.4g, %.4g)\n",
               i, i % grid, i / grid, hbad[i], rows_bad, first_row, last_row, dmax, ndam, bfx, bk,
               bre / (H * 256.0), bim / (H * 256.0), se[0], se[1]);
      }
    }
  }
  if (printed) {
    auto pr = [](const char* n, const int* h, int c) { printf("  lost entries by %s:", n); for (int i = 0; i < c; ++i) printf(" %d", h[i]); printf("\n"); };
    pr("q2 (row q = mu + 8 q2)", hist_q2, 8); pr("mu", hist_mu, 8); pr("cl (column = 8 w + cl)", hist_cl, 8); pr("w (4 = column 32)", hist_w, 5); pr("part (re, im)", hist_part, 2);
  }
  printf("%-28s H %3d grid %4d images %d: %d of %d launches differ (%ld images in all; workgroups < units: %ld, the others: %ld; by a workgroup's image number:",
         ABL_NAME, H, grid, NIMG, bad_launches, reps, bad_images, by_first[0], by_first[1]);
  for (int i = 0; i < 8; ++i) printf(" %ld", by_num[i]);
  printf(")\n");
  return 0;
}
