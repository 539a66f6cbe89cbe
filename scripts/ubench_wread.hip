// How fast can the dense weight of configs[4] (128 x 128 rows of 33024 complex modes, 4.33 GB) be READ in the order the
// small-batch contraction walks it?  Pure load kernels (one fmaf per loaded value so that nothing is elided):
//   walk<QT, MC, ST, FWD>: a wave owns QT weight rows q and MC consecutive 1 KB pieces of each; it steps r = 0..127
//      FWD: rows (r, q0 + q): q-neighbours 264 KB apart, r-steps 33.8 MB apart   (forward: sum over i)
//     !FWD: rows (q0 + q, r): q-neighbours 33.8 MB apart, r-steps 264 KB apart   (gradient of the spectrum: sum over o)
//   with ST steps of loads in flight; the four waves of a workgroup take four neighbouring q tiles (as k_modegemm_sb's
//   1 x 1 x 4 arrangement), items dealt mode-tile-major, consecutive items to one XCD.
//   linear: a grid-stride float4 reader.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NI = 128, NO = 128;
constexpr long M = 33024;
template <int QT, int MC, int ST, bool FWD, bool NT>
__global__ void __launch_bounds__(256) k_walk(const float* w, float* out, int n_mt, int per_xcd) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long item = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_qg = 128 / (4 * QT);
  if (item >= (long)n_mt * n_qg) return;
  const int mt = (int)(item / n_qg), q0 = ((int)(item % n_qg) * 4 + wv) * QT;
  const long rs = FWD ? (long)NO * M : M, qs = FWD ? M : (long)NO * M;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 ring[ST][QT][MC];
  auto req = [&](int r, f4 (&dst)[QT][MC]) {
    r = r < 128 ? r : 127;
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        long m = ((long)mt * MC + c) * 128 + 2 * lane;
        m = m < M ? m : M - 2;
        const f4* src = (const f4*)(w + 2 * ((long)r * rs + (long)(q0 + q) * qs + m));
        dst[q][c] = NT ? __builtin_nontemporal_load(src) : *src;
      }
  };
#pragma unroll
  for (int s = 0; s < ST; ++s) req(s, ring[s]);
#pragma unroll 1
  for (int r0 = 0; r0 < 128; r0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
#pragma unroll
      for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int c = 0; c < MC; ++c) acc += ring[s][q][c];
      req(r0 + s + ST, ring[s]);
    }
  }
  if (acc.x == 1234.5f) out[threadIdx.x] = acc.y + acc.z + acc.w;
}
template <bool NT>
__global__ void __launch_bounds__(256) k_linear(const float* w, float* out, long n4) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    acc += NT ? __builtin_nontemporal_load((const f4*)w + i) : ((const f4*)w)[i];
  if (acc.x == 1234.5f) out[threadIdx.x] = acc.y + acc.z + acc.w;
}
int main() {
  float *w, *out; const size_t bytes = (size_t)NI * NO * M * 8;
  hipMalloc(&w, bytes); hipMalloc(&out, 4096); hipMemset(w, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto f) {
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < 6; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6;
    printf("%-44s %8.1f us  %5.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9);
  };
#define WALK(QT, MC, ST, FWD, NT) { const int n_mt = (int)((M + 128 * MC - 1) / (128 * MC)); const long items = (long)n_mt * (128 / (4 * QT)); \
    const int pxc = (int)((items + 7) / 8); timeit("walk<QT " #QT ",MC " #MC ",ST " #ST "," #FWD "," #NT ">", [&] { hipLaunchKernelGGL((k_walk<QT, MC, ST, FWD, NT>), dim3(8 * pxc), dim3(256), 0, 0, w, out, n_mt, pxc); }); }
  timeit("linear nt grid 2048", [&] { hipLaunchKernelGGL((k_linear<true>), dim3(2048), dim3(256), 0, 0, w, out, (long)(bytes / 16)); });
  timeit("linear plain grid 2048", [&] { hipLaunchKernelGGL((k_linear<false>), dim3(2048), dim3(256), 0, 0, w, out, (long)(bytes / 16)); });
  WALK(4, 1, 3, true, true) WALK(4, 1, 3, false, true) WALK(4, 1, 3, true, false)
  WALK(4, 1, 6, true, true) WALK(4, 1, 6, false, true)
  WALK(2, 2, 4, true, true) WALK(2, 2, 4, false, true)
  WALK(1, 4, 4, true, true) WALK(1, 4, 4, false, true)
  WALK(2, 4, 3, true, true) WALK(2, 4, 3, false, true)
  WALK(4, 2, 3, true, true) WALK(4, 2, 3, false, true)
  WALK(1, 8, 3, true, true) WALK(1, 8, 3, false, true)
  return 0;
}
