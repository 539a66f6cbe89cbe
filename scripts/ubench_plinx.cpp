// Timing harness for build variants of k_plinx_bwd at the 128-channel block's shapes (B = 8, 256 x 256; round 6): loads a code
// object built by scripts/plinx_variants.sh and times each launch of the block's three 1 x 1 maps alone.
//   hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 -w scripts/ubench_plinx.cpp -o plinx_bench
//   ./plinx_bench variant.hsaco [B (8)] [side (256)]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../neuraloperator_amd/csrc/sc_kernels_plinx.h"
int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s variant.hsaco [B] [side]\n", argv[0]); return 2; }
  const std::string path = argv[1];
  const int B = argc > 2 ? atoi(argv[2]) : 8, side = argc > 3 ? atoi(argv[3]) : 256;
  hipModule_t mod;
  if (hipModuleLoad(&mod, path.c_str()) != hipSuccess) { printf("cannot load %s\n", path.c_str()); return 1; }
  const int64_t S = (int64_t)side * side; const size_t n = (size_t)B * 128 * S;
  float *x, *g, *pre, *xg, *ad, *gx, *w, *part;
  hipMalloc(&x, n * 4); hipMalloc(&g, n * 4); hipMalloc(&pre, n * 4); hipMalloc(&xg, n * 4); hipMalloc(&ad, n * 4); hipMalloc(&gx, n * 4);
  hipMalloc(&w, 128 * 128 * 4); hipMalloc(&part, (size_t)256 * 40000 * 4);
  std::vector<float> h(n); unsigned s = 777u;
  auto fill = [&](float* d, size_t m, float sc) { for (size_t i = 0; i < m; ++i) { s = s * 1664525u + 1013904223u; h[i] = sc * (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); } hipMemcpy(d, h.data(), m * 4, hipMemcpyHostToDevice); };
  fill(x, n, 1.f); fill(g, n, 1.f); fill(pre, n, 1.f); fill(xg, n, 1.f); fill(ad, n, 1.f); fill(w, 128 * 128, .1f);
  struct Case { const char* sym; const char* what; int ci, co, om0, do_gx, flags; bool addend; };
  const Case cases[] = {
    {"_Z11k_plinx_bwdILi4ELi4ELi2ELb0EEv9PlinxArgsi", "<4,4,2> skip 128->128, tiles 0-1 + gx (PRO)", 4, 4, 0, 1, SC_PLX_PRO, false},
    {"_Z11k_plinx_bwdILi4ELi4ELi2ELb0EEv9PlinxArgsi", "<4,4,2> skip 128->128, tiles 2-3, weight gradient only (PRO)", 4, 4, 2, 0, SC_PLX_PRO, false},
    {"_Z11k_plinx_bwdILi2ELi4ELi4ELb0EEv9PlinxArgsi", "<2,4,4> fc2 64->128 (XACT | XGRAD)", 2, 4, 0, 1, SC_PLX_XACT | SC_PLX_XGRAD, false},
    {"_Z11k_plinx_bwdILi4ELi2ELi2ELb0EEv9PlinxArgsi", "<4,2,2> fc1 128->64 (addend)", 4, 2, 0, 1, 0, true},
  };
  for (const Case& c : cases) {
    hipFunction_t fk;
    if (hipModuleGetFunction(&fk, mod, c.sym) != hipSuccess) { printf("no %s\n", c.sym); continue; }
    PlinxArgs a; memset(&a, 0, sizeof a);
    a.x = x; a.w = w; a.gout = g; a.pre = pre; a.xg = xg; a.addend = c.addend ? ad : nullptr; a.out = gx; a.partial = part;
    a.spatial = S; a.tiles_per_sample = (int)(S / 32); a.n_tiles = (int64_t)B * a.tiles_per_sample; a.n_wg = 256; a.flags = c.flags; a.do_gx = c.do_gx;
    int om0 = c.om0; void* args[] = {&a, &om0};
    auto launch = [&] { if (hipModuleLaunchKernel(fk, 256, 1, 1, 256, 1, 1, 0, 0, args, nullptr) != hipSuccess) { printf("launch failed\n"); exit(1); } };
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms / 10); }
    std::vector<float> p(4096); hipMemcpy(p.data(), part + 1000, 4096 * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (int i = 0; i < 4096; ++i) cs += (double)p[i] * (i % 7 + 1);
    printf("%-10s %-62s %8.1f us   checksum %.6e\n", path.substr(path.rfind('/') + 1).c_str(), c.what, best * 1e3f, cs);
  }
  return 0;
}
