// Bench driver: bare MFMA loop of k_modegemm_mfma (dbg = 14) with compile-time ablations.
#include "../neuraloperator_amd/csrc/sc_kernels_mfma.h"
#include <cstdio>
#include <vector>
int main() {
  const int B = 32, C = 64, M = 2112;
  cf32 *A, *Bm, *Cm;
  hipMalloc(&A, (size_t)B * C * M * 8); hipMalloc(&Bm, (size_t)C * C * M * 8); hipMalloc(&Cm, (size_t)B * C * M * 8);
  std::vector<float> h((size_t)C * C * M * 2);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(A, h.data(), (size_t)B * C * M * 8, hipMemcpyHostToDevice);
  hipMemcpy(Bm, h.data(), (size_t)C * C * M * 8, hipMemcpyHostToDevice);
  MfmaGemmArgs g;
  g.P = B; g.Q = C; g.R = C; g.M = M; g.G = 256;
  g.a_sp = C * M; g.a_sr = M; g.a_sm = 1; g.b_sr = C * M; g.b_sq = M; g.b_sm = 1; g.c_sp = C * M; g.c_sq = M; g.c_sm = 1;
  g.b_idx = nullptr; g.c_idx = nullptr; g.dbg = 14;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_modegemm_mfma<1, 4, 9, false, false>), dim3(256), dim3(256), 0, 0, g, A, Bm, Cm);
  hipDeviceSynchronize();
  long long c[4]; hipMemcpy(c, Cm, 32, hipMemcpyDeviceToHost);
  printf("%s: %.1f clock64 ticks per MFMA (%.2f GHz)\n", ABL_NAME, c[0] / 576.0, c[0] / (c[1] / 100e6) / 1e9);
  return 0;
}
