// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4 and the shader clock it sustains
// with all CUs busy (zero vs random operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k32(const float* in, float* out, long long* clk, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  __syncthreads();
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int v = 0; v < 16; ++v) s += acc[j][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// distinct A/B operand registers per MFMA (as in a real GEMM loop)
template <int NACC, int STRIDE>
__global__ void __launch_bounds__(256) k32d(const float* in, float* out, long long* clk, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  float a[NACC], b[NACC];
  for (int j = 0; j < NACC; ++j) { a[j] = in[(threadIdx.x + 7 * j) & 1023]; b[j] = in[(threadIdx.x + 256 + 13 * j) & 1023]; }
  __syncthreads();
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j * STRIDE) % NACC], b[(j * STRIDE) % NACC], acc[j], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int v = 0; v < 16; ++v) s += acc[j][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) k16(const float* in, float* out, long long* clk, int iters) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int v = 0; v < 4; ++v) acc[j][v] = 0.f;
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  __syncthreads();
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int v = 0; v < 4; ++v) s += acc[j][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// plain FMA for comparison: 64 independent chains per lane
__global__ void __launch_bounds__(256) kfma(const float* in, float* out, long long* clk, int iters) {
  float acc[32];
  for (int j = 0; j < 32; ++j) acc[j] = (float)j;
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = fmaf(a, acc[j], b);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 32; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  float *in, *out; long long* clk;
  CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, 4 * 256 * 2048)); CHECK(hipMalloc(&clk, 16 * 2048));
  std::vector<float> h(1024);
  int wallkhz = 0; CHECK(hipDeviceGetAttribute(&wallkhz, hipDeviceAttributeWallClockRate, 0));
  int clkkhz = 0; CHECK(hipDeviceGetAttribute(&clkkhz, hipDeviceAttributeClockRate, 0));
  printf("wall clock rate %d kHz, shader clock rate attr %d kHz\n", wallkhz, clkkhz);
  for (int rnd = 0; rnd < 2; ++rnd) {
    for (int i = 0; i < 1024; ++i) h[i] = rnd ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
    CHECK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    auto report = [&](const char* name, int grid, double n_inst, double flop_per_inst, float ms) {
      std::vector<long long> c(2 * grid); CHECK(hipMemcpy(c.data(), clk, 16 * grid, hipMemcpyDeviceToHost));
      double cyc = 0, wall = 0; for (int i = 0; i < grid; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
      cyc /= grid; wall /= grid;
      printf("%-7s %-26s grid %4d: %7.1f us  clock64 ticks/inst %6.1f  wall %7.1f us -> clock64 rate %5.2f GHz   %6.1f TFLOP/s\n",
             rnd ? "random" : "zero", name, grid, ms * 1e3, cyc / n_inst, wall / wallkhz * 1e3, cyc / (wall / wallkhz * 1e-3) / 1e9,
             n_inst * flop_per_inst * 4 * grid / (ms * 1e-3) / 1e12);
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    const int it = 2000;
#define RUNK(name, kern, grid, ninst, flop) kern<<<grid, 256>>>(in, out, clk, it); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); kern<<<grid, 256>>>(in, out, clk, it); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); report(name, grid, ninst, flop, ms);
    RUNK("mfma 32x32x2 x9acc", k32<9>, 256, 9.0 * it, 4096.0)
    RUNK("mfma 32x32x2 x4acc", k32<4>, 256, 4.0 * it, 4096.0)
    RUNK("mfma 32x32x2 x9acc distinct ab", (k32d<9, 1>), 256, 9.0 * it, 4096.0)
    RUNK("mfma 32x32x2 x8acc distinct ab", (k32d<8, 1>), 256, 8.0 * it, 4096.0)
    RUNK("mfma 32x32x2 x9acc", k32<9>, 32, 9.0 * it, 4096.0)
    RUNK("mfma 16x16x4 x8acc", k16<8>, 256, 8.0 * it, 2048.0)
    RUNK("mfma 16x16x4 x8acc 2wg/cu", k16<8>, 512, 8.0 * it, 2048.0)
    RUNK("v_fma x32 chains", kfma, 256, 32.0 * it, 128.0)
    RUNK("v_fma x32 chains 2wg/cu", kfma, 512, 32.0 * it, 128.0)
  }
  return 0;
}
