// fp32 products on the bf16 matrix pipe (round 6): every fp32 value as three bf16 terms (8 + 8 + 8 mantissa bits, each
// remainder exact in fp32), a product as SIX bf16 MFMAs (hh, hm, mh, hl, lh, mm: everything down to 2^-24 relative; the
// products of bf16 values are exact in the fp32 accumulator).  Questions: (1) how close to the fp32 MFMA is the result
// (both against a float64 reference)?  (2) instruction rate: 6 x v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) against
// 8 x v_mfma_f32_32x32x2_f32 (64 cycles, K = 2) for the same 32 x 32 x 16 block;  (3) do vector instructions of the same
// wave / of another wave run beside the bf16 MFMA (they do not beside the fp32 one, DESIGN 3.16 b)?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_bf16x6.hip -o x6 && ./x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ inline int row_of(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }
__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; const float r = x - (float)h;
  m = (__bf16)r; const float r2 = r - (float)m;
  l = (__bf16)r2;
}
// D (32 x 32) = A (32 x K) B (K x 32), K = 64, row-major A, B; one wave
__global__ void k_gemm_f32(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x, n = lane & 31, half = lane >> 5;
  f16v acc = {0};
  for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * 64 + 2 * s + half], B[(2 * s + half) * 32 + n], acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[row_of(v, half) * 32 + n] = acc[v];
}
template <int NP>
__global__ void k_gemm_x(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x, n = lane & 31, g = lane >> 5;
  f16v acc = {0};
  for (int s = 0; s < 4; ++s) {
    bf8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
      __bf16 h, m, l;
      split3(A[n * 64 + 16 * s + 8 * g + e], h, m, l); a[0][e] = h; a[1][e] = m; a[2][e] = l;
      split3(B[(16 * s + 8 * g + e) * 32 + n], h, m, l); b[0][e] = h; b[1][e] = m; b[2][e] = l;
    }
    // small terms first
    if (NP >= 6) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    if (NP >= 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    if (NP >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    if (NP >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    if (NP >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  for (int v = 0; v < 16; ++v) D[row_of(v, g) * 32 + n] = acc[v];
}

// rate / overlap: waves 0-3 matrix work, waves 4-7 vector work (as scripts/ubench_mfma_valu_overlap.hip); MODE 2 = one
// wave per SIMD doing BOTH (vi independent v_fma per MFMA group)
template <int KIND, int NV = 0>
__global__ void __launch_bounds__(512) k_rate(float* sink, int mi, int vi, int same_wave) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  float r[8];
  for (int k = 0; k < 8; ++k) r[k] = 0.001f * (lane + k);
  const float mm = 1.0001f, cc = 0.37f;
  if (w < 4) {
    if (mi == 0) return;
    if (KIND == 0) {
      f16v a0 = {0}, a1 = {0};
      for (int i = 0; i < mi; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                              // 8 x 32x32x2 = one 32 x 32 x 16 block, two accumulators
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.01f * lane, 0.5f, a0, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < NV; ++k) r[k & 7] = __builtin_fmaf(r[k & 7], mm, cc);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.02f * lane, 0.25f, a1, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < NV; ++k) r[k & 7] = __builtin_fmaf(r[k & 7], mm, cc);
        }
        if (same_wave)
#pragma unroll
          for (int k = 0; k < 8; ++k) r[k] = __builtin_fmaf(r[k], mm, cc);
      }
      sink[blockIdx.x * 512 + tid] = a0[0] + a1[3] + r[0] + r[7];
    } else {
      f16v a0 = {0}, a1 = {0};
      bf8 x, y;
      for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.01f * (lane + e)); y[e] = (__bf16)(0.02f * (lane - e)); }
      for (int i = 0; i < mi; ++i) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {                              // 6 x 32x32x16 bf16 = the same block from three-term operands
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < NV; ++k) r[k & 7] = __builtin_fmaf(r[k & 7], mm, cc);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < NV; ++k) r[k & 7] = __builtin_fmaf(r[k & 7], mm, cc);
        }
        if (same_wave)
#pragma unroll
          for (int k = 0; k < 8; ++k) r[k] = __builtin_fmaf(r[k], mm, cc);
      }
      sink[blockIdx.x * 512 + tid] = a0[0] + a1[3] + r[0] + r[7];
    }
    return;
  }
  if (vi == 0) return;
  float q[16];
  for (int k = 0; k < 16; ++k) q[k] = 0.001f * (lane + k);
  for (int i = 0; i < vi; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) q[k] = __builtin_fmaf(q[k], mm, cc);
  }
  float s = 0; for (int k = 0; k < 16; ++k) s += q[k];
  sink[blockIdx.x * 512 + tid] = s;
}

static float timed(void (*launch)(int, int, int), int a, int b, int c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float t = 0;
  for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); launch(a, b, c); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1); }
  return t;
}
static float* g_sink; static int g_grid;
static void l0(int a, int b, int c) { hipLaunchKernelGGL((k_rate<0>), dim3(g_grid), dim3(512), 0, 0, g_sink, a, b, c); }
static void l1(int a, int b, int c) { hipLaunchKernelGGL((k_rate<1>), dim3(g_grid), dim3(512), 0, 0, g_sink, a, b, c); }
template <int KIND, int NV> static void lv(int a, int b, int c) { hipLaunchKernelGGL((k_rate<KIND, NV>), dim3(g_grid), dim3(512), 0, 0, g_sink, a, b, c); }

int main() {
  // ---- (1) accuracy
  std::vector<float> A(32 * 64), B(64 * 32), D(32 * 32);
  unsigned s = 12345u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); };
  double errs[8] = {0}; const int trials = 20;
  float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
  for (int t = 0; t < trials; ++t) {
    for (auto& v : A) v = rnd() * (t % 2 ? 3.7f : 0.21f);
    for (auto& v : B) v = rnd();
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(32 * 32, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double a = 0; for (int k = 0; k < 64; ++k) a += (double)A[i * 64 + k] * (double)B[k * 32 + j]; ref[i * 32 + j] = a; }
    for (int kind = 0; kind < 5; ++kind) {
      if (kind == 0) hipLaunchKernelGGL(k_gemm_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      if (kind == 1) hipLaunchKernelGGL((k_gemm_x<1>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
      if (kind == 2) hipLaunchKernelGGL((k_gemm_x<3>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
      if (kind == 3) hipLaunchKernelGGL((k_gemm_x<5>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
      if (kind == 4) hipLaunchKernelGGL((k_gemm_x<6>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      double num = 0, den = 0; for (int i = 0; i < 32 * 32; ++i) { num += (D[i] - ref[i]) * (D[i] - ref[i]); den += ref[i] * ref[i]; }
      errs[kind] += std::sqrt(num / den) / trials;
    }
  }
  printf("rel-L2 against float64, 32 x 64 x 32 products, mean of %d: fp32 MFMA %.3e | bf16 terms: 1 product %.3e, 3 products %.3e, 5 products %.3e, 6 products %.3e\n",
         trials, errs[0], errs[1], errs[2], errs[3], errs[4]);
  // ---- (2), (3) rate and overlap
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); g_grid = prop.multiProcessorCount;
  hipMalloc(&g_sink, (size_t)g_grid * 512 * 4);
  for (int i = 0; i < 3; ++i) l0(5000, 5000, 0);
  hipDeviceSynchronize();
  const int MI = 5000;                                      // blocks of 32 x 32 x 16 per matrix wave... x 2 accumulators
  float a = timed(l0, MI, 0, 0), b = timed(l0, 0, 40000, 0), c = timed(l0, MI, 40000, 0), d = timed(l0, MI, 0, 1);
  printf("fp32 32x32x2 : %d x 8 MFMAs alone %.3f ms | vector waves alone %.3f | both %.3f | 8 v_fma per 8 MFMAs in the SAME wave %.3f\n", MI, a, b, c, d);
  a = timed(l1, MI, 0, 0); b = timed(l1, 0, 40000, 0); c = timed(l1, MI, 40000, 0); d = timed(l1, MI, 0, 1);
  printf("bf16 32x32x16: %d x 6 MFMAs alone %.3f ms | vector waves alone %.3f | both %.3f | 8 v_fma per 6 MFMAs in the SAME wave %.3f\n", MI, a, b, c, d);
  printf("v_fma_f32 per MFMA in the SAME wave (one wave per SIMD), ms for %d x 8 fp32 MFMAs | %d x 6 bf16 MFMAs:\n", MI, MI);
  printf("   0 per MFMA: %.3f | %.3f\n", timed(lv<0, 0>, MI, 0, 0), timed(lv<1, 0>, MI, 0, 0));
  printf("   2 per MFMA: %.3f | %.3f\n", timed(lv<0, 2>, MI, 0, 0), timed(lv<1, 2>, MI, 0, 0));
  printf("   4 per MFMA: %.3f | %.3f\n", timed(lv<0, 4>, MI, 0, 0), timed(lv<1, 4>, MI, 0, 0));
  printf("   6 per MFMA: %.3f | %.3f\n", timed(lv<0, 6>, MI, 0, 0), timed(lv<1, 6>, MI, 0, 0));
  printf("   8 per MFMA: %.3f | %.3f\n", timed(lv<0, 8>, MI, 0, 0), timed(lv<1, 8>, MI, 0, 0));
  printf("  12 per MFMA: %.3f | %.3f\n", timed(lv<0, 12>, MI, 0, 0), timed(lv<1, 12>, MI, 0, 0));
  printf("  16 per MFMA: %.3f | %.3f   (v_fma alone: 16 per slot = %.3f ms for the fp32 count, %.3f for the bf16 count)\n", timed(lv<0, 16>, MI, 0, 0), timed(lv<1, 16>, MI, 0, 0),
         MI * 8 * 16 * 4 / 2.1e6, MI * 6 * 16 * 4 / 2.1e6);
  return 0;
}
