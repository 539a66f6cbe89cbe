// Micro-benchmark: HBM bandwidth of the fused-FFT kernels' global access patterns.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_access.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define H 256
#define W 256
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// pattern of k_fft2d_fwd: thread (f = tid>>4, t = tid&15) reads rows (A,B) at t + 16 j
__global__ void __launch_bounds__(256) rd_dword_seg64(const float* __restrict__ x, float* __restrict__ out) {
  const int tid = threadIdx.x, f = tid >> 4, t = tid & 15;
  const float* xi = x + (size_t)blockIdx.x * H * W;
  float acc = 0.f;
  for (int rr = 0; rr < 8; ++rr) {
    const int a = rr >> 1, r = rr & 1, p = r * 16 + f;
    const float* ra = xi + (size_t)(4 * (2 * p) + a) * W + t;
    const float* rb = xi + (size_t)(4 * (2 * p + 1) + a) * W + t;
    float v[32];
#pragma unroll
    for (int j = 0; j < 16; ++j) { v[j] = ra[16 * j]; v[16 + j] = rb[16 * j]; }
#pragma unroll
    for (int j = 0; j < 32; ++j) acc += v[j];
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// same bytes, float4 per lane: thread (f, t) reads rows (A,B) float4 index t + 16 j, j = 0..3
__global__ void __launch_bounds__(256) rd_float4(const float* __restrict__ x, float* __restrict__ out) {
  const int tid = threadIdx.x, f = tid >> 4, t = tid & 15;
  const float* xi = x + (size_t)blockIdx.x * H * W;
  float acc = 0.f;
  for (int rr = 0; rr < 8; ++rr) {
    const int a = rr >> 1, r = rr & 1, p = r * 16 + f;
    const float4* ra = reinterpret_cast<const float4*>(xi + (size_t)(4 * (2 * p) + a) * W) + t;
    const float4* rb = reinterpret_cast<const float4*>(xi + (size_t)(4 * (2 * p + 1) + a) * W) + t;
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = ra[16 * j]; v[4 + j] = rb[16 * j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) wr_dword_seg64(float* __restrict__ y) {
  const int tid = threadIdx.x, f = tid >> 4, t = tid & 15;
  float* yi = y + (size_t)blockIdx.x * H * W;
  for (int rr = 0; rr < 8; ++rr) {
    const int a = rr >> 1, r = rr & 1, p = r * 16 + f;
    float* ra = yi + (size_t)(4 * (2 * p) + a) * W + t;
    float* rb = yi + (size_t)(4 * (2 * p + 1) + a) * W + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) { ra[16 * j] = (float)(j + rr); rb[16 * j] = (float)(j - rr); }
  }
}

__global__ void __launch_bounds__(256) wr_float4(float* __restrict__ y) {
  const int tid = threadIdx.x, f = tid >> 4, t = tid & 15;
  float* yi = y + (size_t)blockIdx.x * H * W;
  for (int rr = 0; rr < 8; ++rr) {
    const int a = rr >> 1, r = rr & 1, p = r * 16 + f;
    float4* ra = reinterpret_cast<float4*>(yi + (size_t)(4 * (2 * p) + a) * W) + t;
    float4* rb = reinterpret_cast<float4*>(yi + (size_t)(4 * (2 * p + 1) + a) * W) + t;
#pragma unroll
    for (int j = 0; j < 4; ++j) { ra[16 * j] = make_float4(j, rr, 1.f, 2.f); rb[16 * j] = make_float4(rr, j, 3.f, 4.f); }
  }
}

// plain streaming copy for the ceiling
__global__ void __launch_bounds__(256) copy_f4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}


// ---- candidate patterns for the next kernel generation --------------------------------------
// whole-row wave loads: one wave-instruction = one 1 KB row (lane l reads float4 l);
// each wave keeps DEPTH rows in flight; LDSB bytes of dummy LDS cap the occupancy
template <int DEPTH, int LDSB>
__global__ void __launch_bounds__(256) rd_rows(const float* __restrict__ x, float* __restrict__ out, int nimg) {
  __shared__ float dummy[LDSB / 4 + 1];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  float acc = 0.f;
  for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
    const float4* xi = reinterpret_cast<const float4*>(x + (size_t)img * H * W);
    for (int r0 = w * (H / 4); r0 < (w + 1) * (H / 4); r0 += DEPTH) {
      float4 v[DEPTH];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) v[j] = xi[(size_t)(r0 + j) * 64 + l];
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
  }
  if (acc == 12345.678f) { dummy[tid] = acc; out[blockIdx.x] = dummy[(tid + 1) & 255]; }
}

template <int DEPTH, int LDSB>
__global__ void __launch_bounds__(256) wr_rows(float* __restrict__ y, int nimg) {
  __shared__ float dummy[LDSB / 4 + 1];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  if (nimg < 0) dummy[tid] = 1.f;
  for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
    float4* yi = reinterpret_cast<float4*>(y + (size_t)img * H * W);
    for (int r0 = w * (H / 4); r0 < (w + 1) * (H / 4); r0 += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) yi[(size_t)(r0 + j) * 64 + l] = make_float4(j, r0, l, 1.f);
    }
  }
  if (nimg < 0) y[0] = dummy[(tid + 1) & 255];
}

// read with 8 B per lane (float2), 16-lane group = 128 B contiguous (row-pair FFT lanes holding 2 adjacent elements)
template <int LDSB>
__global__ void __launch_bounds__(256) rd_float2_seg128(const float* __restrict__ x, float* __restrict__ out, int nimg) {
  __shared__ float dummy[LDSB / 4 + 1];
  const int tid = threadIdx.x, f = tid >> 4, t = tid & 15;
  float acc = 0.f;
  for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
    const float* xi = x + (size_t)img * H * W;
    for (int rr = 0; rr < 8; ++rr) {
      const int p = rr * 16 + f;
      const float2* ra = reinterpret_cast<const float2*>(xi + (size_t)(2 * p) * W) + t;
      const float2* rb = reinterpret_cast<const float2*>(xi + (size_t)(2 * p + 1) * W) + t;
      float2 v[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = ra[16 * j]; v[8 + j] = rb[16 * j]; }
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += v[j].x + v[j].y;
    }
  }
  if (acc == 12345.678f) { dummy[tid] = acc; out[blockIdx.x] = dummy[(tid + 1) & 255]; }
}

// the generation-3 forward kernel's load stream, no arithmetic: half-wave = one row pair, lane reads
// x[row][32 j + lam], rows h = 4 b + a (DIT over H) or h = 64 a + b (CONSEC), DEPTH rounds in flight
template <int DEPTH, bool CONSEC, int LDSB>
__global__ void __launch_bounds__(256) rd_gen3(const float* __restrict__ x, float* __restrict__ out) {
  __shared__ float dummy[LDSB / 4 + 1];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hs = lane >> 5, lam = lane & 31, hw = w * 2 + hs;
  const float* xi = x + (size_t)blockIdx.x * H * W;
  float acc = 0.f;
  float v[DEPTH][16];
  auto issue = [&](int t, float (&q)[16]) {
    if (t < 16) {
      const int a = t >> 2, p = (t & 3) * 8 + hw;
      const int ha = CONSEC ? 64 * a + 2 * p : 4 * (2 * p) + a, hb = CONSEC ? ha + 1 : 4 * (2 * p + 1) + a;
      const float* ra = xi + (size_t)ha * W + lam;
      const float* rb = xi + (size_t)hb * W + lam;
#pragma unroll
      for (int j = 0; j < 8; ++j) { q[j] = ra[32 * j]; q[8 + j] = rb[32 * j]; }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, v[d]);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += v[t % DEPTH][j];
    issue(t + DEPTH, v[t % DEPTH]);
  }
  if (acc == 12345.678f) { dummy[tid] = acc; out[blockIdx.x] = dummy[(tid + 1) & 255]; }
}

template <class F>
float timeit(F launch, int iters = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int NIMG = 2048;
  const size_t n = (size_t)NIMG * H * W;
  float *x, *y, *o;
  CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&y, n * 4)); CHECK(hipMalloc(&o, NIMG * 4));
  CHECK(hipMemset(x, 1, n * 4));
  const double gb = n * 4 / 1e9;
  float t;
  t = timeit([&] { rd_dword_seg64<<<NIMG, 256>>>(x, o); });  printf("rd_dword_seg64 : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { rd_float4<<<NIMG, 256>>>(x, o); });       printf("rd_float4      : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { wr_dword_seg64<<<NIMG, 256>>>(y); });     printf("wr_dword_seg64 : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { wr_float4<<<NIMG, 256>>>(y); });          printf("wr_float4      : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { copy_f4<<<2048, 256>>>((const float4*)x, (float4*)y, n / 4); });
  printf("copy_f4 (r+w)  : %7.1f us  %7.1f GB/s\n", t * 1e3, 2 * gb / (t * 1e-3));

#define RUN(name, call, bytes) t = timeit([&] { call; }); printf("%-34s: %7.1f us  %7.1f GB/s\n", name, t * 1e3, (bytes) / (t * 1e-3));
  RUN("rd_rows<4,0>  grid 2048", (rd_rows<4, 0><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<8,0>  grid 2048", (rd_rows<8, 0><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<16,0> grid 2048", (rd_rows<16, 0><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<8,60000> grid 2048 (2 WG/CU)", (rd_rows<8, 60000><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<16,60000> grid 2048 (2 WG/CU)", (rd_rows<16, 60000><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<16,60000> grid 512 persistent", (rd_rows<16, 60000><<<512, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<8,36000> grid 2048 (4 WG/CU)", (rd_rows<8, 36000><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_rows<16,36000> grid 1024 persistent", (rd_rows<16, 36000><<<1024, 256>>>(x, o, NIMG)), gb)
  RUN("rd_float2_seg128<60000> grid 2048", (rd_float2_seg128<60000><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_float2_seg128<36000> grid 2048", (rd_float2_seg128<36000><<<2048, 256>>>(x, o, NIMG)), gb)
  RUN("rd_gen3<1,DIT,40000> (4 WG/CU)", (rd_gen3<1, false, 40000><<<2048, 256>>>(x, o)), gb)
  RUN("rd_gen3<2,DIT,40000> (4 WG/CU)", (rd_gen3<2, false, 40000><<<2048, 256>>>(x, o)), gb)
  RUN("rd_gen3<2,DIT,53000> (3 WG/CU)", (rd_gen3<2, false, 53000><<<2048, 256>>>(x, o)), gb)
  RUN("rd_gen3<1,CONSEC,40000> (4 WG/CU)", (rd_gen3<1, true, 40000><<<2048, 256>>>(x, o)), gb)
  RUN("rd_gen3<2,CONSEC,40000> (4 WG/CU)", (rd_gen3<2, true, 40000><<<2048, 256>>>(x, o)), gb)
  RUN("wr_rows<8,0>  grid 2048", (wr_rows<8, 0><<<2048, 256>>>(y, NIMG)), gb)
  RUN("wr_rows<16,60000> grid 2048 (2 WG/CU)", (wr_rows<16, 60000><<<2048, 256>>>(y, NIMG)), gb)
  RUN("wr_rows<16,60000> grid 512 persistent", (wr_rows<16, 60000><<<512, 256>>>(y, NIMG)), gb)
  RUN("wr_rows<8,36000> grid 2048 (4 WG/CU)", (wr_rows<8, 36000><<<2048, 256>>>(y, NIMG)), gb)
  return 0;
}
