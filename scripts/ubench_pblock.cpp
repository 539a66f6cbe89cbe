// Timing harness for build variants of k_pblock_fwd at the metric block's shape (B = 32, 64 channels, hidden 32, 256 x 256;
// round 6): loads a code object built by scripts/pblock_variants.sh and times the launch alone.
//   hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 scripts/ubench_pblock.cpp -o pblock_bench
//   ./pblock_bench variant.hsaco [n_wg (default 512)] [launches] [pre_is_grad]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../neuraloperator_amd/csrc/sc_kernels_pmlp.h"
int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s variant.hsaco [n_wg] [launches] [pre_is_grad]\n", argv[0]); return 2; }
  const std::string path = argv[1];
  const int n_wg = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 10, dgrad = argc > 4 ? atoi(argv[4]) : 1;
  std::string sym; { std::ifstream f(path.substr(0, path.size() - 6) + ".sym"); std::getline(f, sym); }
  hipModule_t mod; hipFunction_t fk;
  if (hipModuleLoad(&mod, path.c_str()) != hipSuccess || hipModuleGetFunction(&fk, mod, sym.c_str()) != hipSuccess) { printf("%s: cannot load %s\n", path.c_str(), sym.c_str()); return 1; }
  const int B = 32, C = 64, Hd = 32; const int64_t S = 256 * 256; const size_t n = (size_t)B * C * S;
  float *x, *conv, *y, *pre, *out, *ws, *bs, *w1, *b1, *w2, *b2, *gt;
  hipMalloc(&x, n * 4); hipMalloc(&conv, n * 4); hipMalloc(&y, n * 4); hipMalloc(&pre, n * 4); hipMalloc(&out, n * 4);
  hipMalloc(&ws, C * C * 4); hipMalloc(&bs, C * 4); hipMalloc(&w1, Hd * C * 4); hipMalloc(&w2, C * Hd * 4); hipMalloc(&b1, Hd * 4); hipMalloc(&b2, C * 4); hipMalloc(&gt, C * 4);
  std::vector<float> h(n); unsigned s = 777u;
  auto fill = [&](float* d, size_t m, float sc) { for (size_t i = 0; i < m; ++i) { s = s * 1664525u + 1013904223u; h[i] = sc * (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); } hipMemcpy(d, h.data(), m * 4, hipMemcpyHostToDevice); };
  fill(x, n, 1.f); fill(conv, n, 1.f); fill(ws, C * C, .125f); fill(bs, C, 1.f); fill(w1, Hd * C, .125f); fill(w2, C * Hd, .17f); fill(b1, Hd, 1.f); fill(b2, C, 1.f); fill(gt, C, 1.f);
  PblockArgs g; memset(&g, 0, sizeof g);
  g.conv = conv; g.x = x; g.ws = ws; g.bs = bs; g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.gate = gt; g.y = y; g.pre = pre; g.pre_is_grad = dgrad; g.out = out;
  g.spatial = S; g.tiles_per_sample = (int)(S / 32); g.n_tiles = (int64_t)B * g.tiles_per_sample; g.n_wg = n_wg;
  void* args[] = {&g};
  auto launch = [&] { if (hipModuleLaunchKernel(fk, n_wg, 1, 1, 256, 1, 1, 0, 0, args, nullptr) != hipSuccess) { printf("launch failed\n"); exit(1); } };
  for (int i = 0; i < 30; ++i) launch();                    // settle the clocks
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) { hipEventRecord(e0); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms / reps); }
  std::vector<float> a(4096), b(4096); hipMemcpy(a.data(), out + 123456, 4096 * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), pre + 654321, 4096 * 4, hipMemcpyDeviceToHost);
  double ca = 0, cb = 0; for (int i = 0; i < 4096; ++i) { ca += (double)a[i] * (i % 7 + 1); cb += (double)b[i] * (i % 5 + 1); }
  printf("%-28s n_wg %3d dgrad %d: %8.1f us per launch   checksums out %.6e  pre %.6e\n", path.substr(path.rfind('/') + 1).c_str(), n_wg, dgrad, best * 1e3f, ca, cb);
  return 0;
}
