// Micro-benchmark (round 3): does the SHAPE of a non-temporal store stream matter?  The inverse FFT kernel writes
// its 537 MB as 4-byte-per-lane stores (a half-wave = one 128-byte piece, rows in decimated order h = 4 b + a);
// the alternative would be 16-byte-per-lane stores covering a whole 1 KB row per wave instruction (through an LDS
// patch).  Each writer is followed by a 537 MB reader (what the NEXT kernel pays), events around both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define H 256
#define W 256
template <int NT>
__global__ void __launch_bounds__(256) wr_gen3_dword(float* __restrict__ y) {       // the kernel's pattern
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hs = lane >> 5, lam = lane & 31, hw = w * 2 + hs;
  float* yi = y + (size_t)blockIdx.x * H * W;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 4; ++r) {
      const int p = r * 8 + hw;
      float* ra = yi + (size_t)(4 * (2 * p) + a) * W + lam;
      float* rb = yi + (size_t)(4 * (2 * p + 1) + a) * W + lam;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (NT) { __builtin_nontemporal_store((float)(j + r), ra + 32 * j); __builtin_nontemporal_store((float)(j - a), rb + 32 * j); }
        else { ra[32 * j] = (float)(j + r); rb[32 * j] = (float)(j - a); }
      }
    }
}
template <int NT>
__global__ void __launch_bounds__(256) wr_gen3_row16(float* __restrict__ y) {       // same rows, 16 B per lane: a half-wave = half a row
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hs = lane >> 5, lam = lane & 31, hw = w * 2 + hs;
  float* yi = y + (size_t)blockIdx.x * H * W;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 4; ++r) {
      const int p = r * 8 + hw;
      v4f* ra = reinterpret_cast<v4f*>(yi + (size_t)(4 * (2 * p) + a) * W) + lam;
      v4f* rb = reinterpret_cast<v4f*>(yi + (size_t)(4 * (2 * p + 1) + a) * W) + lam;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const v4f va = {(float)j, (float)r, 1.f, 2.f}, vb = {(float)a, (float)j, 3.f, 4.f};
        if (NT) { __builtin_nontemporal_store(va, ra + 32 * j); __builtin_nontemporal_store(vb, rb + 32 * j); }
        else { ra[32 * j] = va; rb[32 * j] = vb; }
      }
    }
}
__global__ void __launch_bounds__(256) reader(const float4* __restrict__ x, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x) + i); acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}
int main() {
  const size_t n = (size_t)2048 * H * W;
  float *x, *y, *o;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&o, 1 << 20);
  hipMemset(x, 1, n * 4);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const char* names[] = {"dword pieces, plain", "dword pieces, nt", "16 B per lane, plain", "16 B per lane, nt"};
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 4; ++mode) {
    float tw = 0, tr = 0; const int it = 20;
    for (int k = 0; k < it + 10; ++k) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: wr_gen3_dword<0><<<2048, 256>>>(y); break;
        case 1: wr_gen3_dword<1><<<2048, 256>>>(y); break;
        case 2: wr_gen3_row16<0><<<2048, 256>>>(y); break;
        case 3: wr_gen3_row16<1><<<2048, 256>>>(y); break;
      }
      hipEventRecord(e1);
      reader<<<2048, 256>>>((const float4*)x, n / 4, o);
      hipEventRecord(e2); hipEventSynchronize(e2);
      float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
      if (k >= 10) { tw += a; tr += b; }
    }
    printf("%-22s writer %6.1f us  following reader %6.1f us  pair %6.1f us\n", names[mode], tw / it * 1e3, tr / it * 1e3, (tw + tr) / it * 1e3);
  }
  return 0;
}
