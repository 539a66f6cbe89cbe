// Micro-benchmark: does a 0.5 GB write leave dirty Infinity-Cache lines that the NEXT kernel pays for,
// and do non-temporal stores avoid it?  Sequence timed: [writer(y)] [reader(x)] with events around each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ void __launch_bounds__(256) writer(v4f* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    v4f v = {(float)i, 1.f, 2.f, 3.f};
    if (MODE == 0) y[i] = v;
    if (MODE == 1) __builtin_nontemporal_store(v, &y[i]);
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(&y[i]), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&y[i]), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" :: "v"(&y[i]), "v"(v) : "memory");
  }
}
__global__ void __launch_bounds__(256) reader(const float4* __restrict__ x, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}
int main() {
  const size_t n = (size_t)2048 * 256 * 256 / 4;   // 536.9 MB as float4
  float4* x; v4f* y; float* o;
  CHECK(hipMalloc(&x, n * 16)); CHECK(hipMalloc(&y, n * 16)); CHECK(hipMalloc(&o, 1 << 20));
  CHECK(hipMemset(x, 1, n * 16));
  hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  const char* names[] = {"plain", "nontemporal (nt)", "sc0 sc1", "sc1", "nt sc1"};
  for (int mode = 0; mode < 5; ++mode) {
    float tw = 0, tr = 0; const int it = 10;
    for (int k = 0; k < it + 2; ++k) {
      CHECK(hipEventRecord(e0));
      switch (mode) { case 0: writer<0><<<2048, 256>>>(y, n); break; case 1: writer<1><<<2048, 256>>>(y, n); break; case 2: writer<2><<<2048, 256>>>(y, n); break; case 3: writer<3><<<2048, 256>>>(y, n); break; case 4: writer<4><<<2048, 256>>>(y, n); break; }
      CHECK(hipEventRecord(e1));
      reader<<<2048, 256>>>(x, n, o);
      CHECK(hipEventRecord(e2)); CHECK(hipEventSynchronize(e2));
      float a, b; CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
      if (k >= 2) { tw += a; tr += b; }
    }
    printf("%-18s stores: writer %6.1f us  following reader %6.1f us  pair %6.1f us\n", names[mode], tw / it * 1e3, tr / it * 1e3, (tw + tr) / it * 1e3);
  }
  return 0;
}
