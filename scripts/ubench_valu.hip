// Micro-benchmark: issue rate of packed vs scalar fp32 VALU ops at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) kv(const float* in, float* out, long long* clk, int iters) {
  f2 acc[16];
  // register-only initial values: no loads, so no s_waitcnt lands inside the timed loop
  const float seed = 1.0f + 1e-6f * (float)(threadIdx.x + iters);
  for (int j = 0; j < 16; ++j) { acc[j].x = seed + 1e-3f * j; acc[j].y = seed - 1e-3f * j; }
  f2 a, b; a.x = seed * 0.999f; a.y = seed * 1.001f; b.x = 1e-4f * seed; b.y = -1e-4f * seed;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[j]) : "v"(a));
      if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[j]) : "v"(a));
      if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(a), "v"(b));
      if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[j].x) : "v"(a.x));
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[j].x) : "v"(a.x), "v"(b.x));
      if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[j].x) : "v"(acc[(j + 1) & 15].y));
      if (OP == 6) asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[(j + 1) & 15]), "v"(acc[(j + 2) & 15]));
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  float *in, *out; long long* clk;
  CHECK(hipMalloc(&in, 8192)); CHECK(hipMalloc(&out, 4 * 256 * 4096)); CHECK(hipMalloc(&clk, 8 * 4096));
  std::vector<float> h(2048);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 0.01f + 1.0f;
  CHECK(hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice));
  const char* names[] = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32", "v_fma_f32", "v_mov_b32", "v_pk_mov_b32"};
  const int it = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int grid = 256 * wps;
    for (int op = 0; op < 7; ++op) {
      auto launch = [&] {
        switch (op) {
          case 0: kv<0><<<grid, 256>>>(in, out, clk, it); break; case 1: kv<1><<<grid, 256>>>(in, out, clk, it); break;
          case 2: kv<2><<<grid, 256>>>(in, out, clk, it); break; case 3: kv<3><<<grid, 256>>>(in, out, clk, it); break;
          case 4: kv<4><<<grid, 256>>>(in, out, clk, it); break; case 5: kv<5><<<grid, 256>>>(in, out, clk, it); break;
          case 6: kv<6><<<grid, 256>>>(in, out, clk, it); break;
        }
      };
      launch(); CHECK(hipDeviceSynchronize()); launch(); CHECK(hipDeviceSynchronize());
      std::vector<long long> c(grid); CHECK(hipMemcpy(c.data(), clk, 8 * grid, hipMemcpyDeviceToHost));
      double cyc = 0; for (int i = 0; i < grid; ++i) cyc += c[i]; cyc /= grid;
      printf("%d wave/SIMD  %-14s: %5.2f cycles per instr per wave -> %5.2f per SIMD\n", wps, names[op], cyc / (16.0 * it), cyc / (16.0 * it) / wps);
    }
  }
  return 0;
}
