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
  CHECK(hipMemset(x, 0, n * 4));
  const double gb = n * 4 / 1e9;
  float t;
  t = timeit([&] { rd_dword_seg64<<<NIMG, 256>>>(x, o); });  printf("rd_dword_seg64 : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { rd_float4<<<NIMG, 256>>>(x, o); });       printf("rd_float4      : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { wr_dword_seg64<<<NIMG, 256>>>(y); });     printf("wr_dword_seg64 : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { wr_float4<<<NIMG, 256>>>(y); });          printf("wr_float4      : %7.1f us  %7.1f GB/s\n", t * 1e3, gb / (t * 1e-3));
  t = timeit([&] { copy_f4<<<2048, 256>>>((const float4*)x, (float4*)y, n / 4); });
  printf("copy_f4 (r+w)  : %7.1f us  %7.1f GB/s\n", t * 1e3, 2 * gb / (t * 1e-3));
  return 0;
}
