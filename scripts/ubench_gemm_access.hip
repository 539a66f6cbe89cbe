// Micro-benchmark: HBM/L2 behaviour of the mode-GEMM operand access patterns.
//   operand T[rows][M] complex64, rows = (p, r) pairs, M modes contiguous (row = M*8 bytes)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_gemm_access.hip -o scripts/ubench_gemm_access.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// A: each block owns a contiguous mode range [m0, m0+nm); lanes = (segment, mode); every wave-load
// covers spi = 64 / nm rows; rows are visited in order; DEPTH loads in flight per wave.
template <int DEPTH>
__global__ void __launch_bounds__(256) seg_read(const float2* __restrict__ T, int rows, int M, int G, float* out, int xcdmap) {
  int gid = blockIdx.x;
  if (xcdmap && (G & 7) == 0) gid = (gid & 7) * (G >> 3) + (gid >> 3);
  const int m0 = (int)(((long)gid * M) / G), nm = (int)(((long)(gid + 1) * M) / G) - m0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int spi = 64 / nm, sl = lane / nm, jl = lane - sl * nm;
  const bool lv = sl < spi;
  const float2* base = T + (long)(lv ? sl : 0) * M + m0 + (lv ? jl : 0);
  float acc = 0.f;
  const int ngroups = (rows + spi - 1) / spi;        // wave-loads in total, split over 4 waves
  for (int g0 = w; g0 < ngroups; g0 += 4 * DEPTH) {
    float2 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      int gg = g0 + 4 * d;
      if (gg >= ngroups) gg = ngroups - 1;
      int rb = gg * spi;
      if (rb + sl >= rows) rb = rows - 1 - (lv ? sl : 0);
      v[d] = base[(long)rb * M];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// B: the lanes-are-modes pattern of the VALU kernel: a wave owns 64 consecutive modes (512 B) and walks
// RPW rows; grid = (M/64 mode tiles) x (rows / RPW row groups)
template <int DEPTH>
__global__ void __launch_bounds__(256) tile_read(const float2* __restrict__ T, int rows, int M, int rpw, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ntile = M / 64;
  const int item = blockIdx.x * 4 + w;
  const int mt = item % ntile, rg = item / ntile;
  const float2* base = T + (long)rg * rpw * M + mt * 64 + lane;
  float acc = 0.f;
  for (int r0 = 0; r0 < rpw; r0 += DEPTH) {
    float2 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = base[(long)(r0 + d) * M];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// C: plain streaming of the same bytes (float4 per lane)
__global__ void __launch_bounds__(256) stream_read(const float4* __restrict__ a, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

__global__ void flush(float4* p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1, 2, 3, 4); }

static float4* g_flush; static size_t g_nflush;
template <class F>
float timeit(F launch, bool cold, int iters = 10) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  float tot = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (cold) flush<<<2048, 256>>>(g_flush, g_nflush);
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
  }
  return tot / iters;
}

int main() {
  const int M = 2112, rows = 64 * 64;                 // W-sized operand: 64 x 64 rows of 2112 modes = 69.2 MB
  const size_t n = (size_t)rows * M;
  float2* T; float* o;
  CHECK(hipMalloc(&T, n * 8)); CHECK(hipMalloc(&o, 1 << 20));
  CHECK(hipMemset(T, 0, n * 8));
  g_nflush = (size_t)600 * 1024 * 1024 / 16; CHECK(hipMalloc(&g_flush, g_nflush * 16));
  const double gb = n * 8 / 1e9;
  float t;
#define RUN(name, call) for (int cold = 0; cold < 2; ++cold) { t = timeit([&] { call; }, cold); printf("%-44s %s: %7.1f us  %7.1f GB/s\n", name, cold ? "cold" : "warm", t * 1e3, gb / (t * 1e-3)); }
  RUN("stream float4 grid 2048", (stream_read<<<2048, 256>>>((const float4*)T, n / 2, o)))
  RUN("seg_read<8>  G=256 (8.25 modes) xcdmap", (seg_read<8><<<256, 256>>>(T, rows, M, 256, o, 1)))
  RUN("seg_read<16> G=256 (8.25 modes) xcdmap", (seg_read<16><<<256, 256>>>(T, rows, M, 256, o, 1)))
  RUN("seg_read<32> G=256 (8.25 modes) xcdmap", (seg_read<32><<<256, 256>>>(T, rows, M, 256, o, 1)))
  RUN("seg_read<16> G=256 (8.25 modes) no xcdmap", (seg_read<16><<<256, 256>>>(T, rows, M, 256, o, 0)))
  RUN("seg_read<16> G=132 (16 modes, 128B aligned)", (seg_read<16><<<132, 256>>>(T, rows, M, 132, o, 0)))
  RUN("seg_read<16> G=264 (8 modes, 64B aligned)", (seg_read<16><<<264, 256>>>(T, rows, M, 264, o, 1)))
  RUN("seg_read<16> G=528 (4 modes)", (seg_read<16><<<528, 256>>>(T, rows, M, 528, o, 1)))
  RUN("seg_read<16> G=66 (32 modes)", (seg_read<16><<<66, 256>>>(T, rows, M, 66, o, 0)))
  RUN("tile_read<8>  rpw=64 (VALU-kernel pattern)", (tile_read<8><<<33 * 64 / 4, 256>>>(T, rows, M, 64, o)))
  RUN("tile_read<16> rpw=64", (tile_read<16><<<33 * 64 / 4, 256>>>(T, rows, M, 64, o)))
  RUN("tile_read<16> rpw=16", (tile_read<16><<<33 * 256 / 4, 256>>>(T, rows, M, 16, o)))
  return 0;
}
