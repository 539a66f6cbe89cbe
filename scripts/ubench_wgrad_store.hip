// How fast can a weight-gradient-shaped result be WRITTEN?  configs[4]: gW (128 x 128 rows of 33024 complex modes =
// 264 KB each, 4.33 GB) leaves k_modegemm_sb as 1 KB pieces, one per (i, o) row and 128-mode tile, at 3.0 TB/s.
// Pure store kernels (no loads, no arithmetic) with the same ownership patterns:
//   pieces<PT, QT, MC>: a wave owns a PT x QT tile of rows and MC consecutive 1 KB pieces of each (MC KB contiguous
//                       per row); a workgroup = 2 x 2 waves (rows i, o) as in k_modegemm_sb's weight-gradient shape;
//                       work items are dealt mode-tile-major, consecutive items to one XCD
//   linear            : a plain grid-stride float4 writer of the same bytes
// nt = non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NI = 128, NO = 128;
constexpr long M = 33024;                     // complex modes per row (8 B each): 264 192 B
template <int PT, int QT, int MC, bool NT>
__global__ void __launch_bounds__(256) k_pieces(float* w, int n_mt, int per_xcd) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wp = wv >> 1, wq = wv & 1;
  const long item = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_pt = NI / (2 * PT), n_qt = NO / (2 * QT);
  if (item >= (long)n_mt * n_pt * n_qt) return;
  const int mt = (int)(item / (n_pt * n_qt));
  const int rem = (int)(item % (n_pt * n_qt));
  const int i0 = ((rem % n_pt) * 2 + wp) * PT, o0 = ((rem / n_pt) * 2 + wq) * QT;
  const f4 v = {1.f, 2.f, 3.f, (float)lane};
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int c = 0; c < MC; ++c) {
        const long m = ((long)mt * MC + c) * 128 + 2 * lane;          // this lane's two modes
        if (m < M) {
          f4* dst = (f4*)(w + 2 * (((long)(i0 + p) * NO + (o0 + q)) * M + m));
          if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
      }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_linear(float* w, long n4) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, (f4*)w + i); else ((f4*)w)[i] = v;
  }
}
int main() {
  float* w; const size_t bytes = (size_t)NI * NO * M * 8;
  hipMalloc(&w, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto f) {
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < 6; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6;
    printf("%-44s %8.1f us  %5.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9);
  };
#define PIECES(PT, QT, MC, NT) { const int n_mt = (int)((M + 128 * MC - 1) / (128 * MC)); const long items = (long)n_mt * (NI / (2 * PT)) * (NO / (2 * QT)); \
    const int pxc = (int)((items + 7) / 8); timeit("pieces<" #PT "," #QT "," #MC "," #NT ">", [&] { hipLaunchKernelGGL((k_pieces<PT, QT, MC, NT>), dim3(8 * pxc), dim3(256), 0, 0, w, n_mt, pxc); }); }
  timeit("linear nt grid 2048", [&] { hipLaunchKernelGGL((k_linear<true>), dim3(2048), dim3(256), 0, 0, w, (long)(bytes / 16)); });
  timeit("linear plain grid 2048", [&] { hipLaunchKernelGGL((k_linear<false>), dim3(2048), dim3(256), 0, 0, w, (long)(bytes / 16)); });
  PIECES(4, 4, 1, true) PIECES(4, 4, 1, false)
  PIECES(2, 2, 4, true) PIECES(2, 2, 4, false)
  PIECES(2, 4, 2, true) PIECES(1, 2, 8, true) PIECES(1, 1, 16, true) PIECES(4, 4, 4, true) PIECES(2, 2, 16, true)
  return 0;
}
