// Which operand-select encodings of the packed-fp32 instructions misbehave while another wave of the SIMD runs MFMAs?
// (round 6, DESIGN 3.5: `v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,0]` in k_fft2d_inv_mx<64> returned 0 in the
// low result of lanes 48-63, now and then, with two workgroups per compute unit.)
//
// Every workgroup has 8 waves = two per SIMD.  Waves 0-3 ("matrix waves") run a loop of v_mfma_f32_16x16x32_bf16,
// waves 4-7 ("test waves") execute ONE encoding of v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 (inline assembly, so the
// encoding is exactly the one named) on lane-dependent non-zero operands and compare both result halves bit for bit
// with the same products from v_mul_f32 / v_fma_f32 / v_add_f32.  Mismatches are counted per (form, half, lane / 16).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_pk_forms.hip -o pk_forms && ./pk_forms [iterations] [mfma 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define FORM_LIST(X) \
  X(0, 0, 0, 0) X(0, 0, 0, 1) X(0, 0, 1, 0) X(0, 0, 1, 1) X(0, 1, 0, 0) X(0, 1, 0, 1) X(0, 1, 1, 0) X(0, 1, 1, 1) \
  X(1, 0, 0, 0) X(1, 0, 0, 1) X(1, 0, 1, 0) X(1, 0, 1, 1) X(1, 1, 0, 0) X(1, 1, 0, 1) X(1, 1, 1, 0) X(1, 1, 1, 1)
// form index = s0 + 2 s1 + 4 h0 + 8 h1  (op_sel:[s0,s1] op_sel_hi:[h0,h1])

template <int S0, int S1, int H0, int H1>
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) {
  f2 r;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(r) : "v"(a), "v"(b), "n"(S0), "n"(S1), "n"(H0), "n"(H1));
  return r;
}
template <int S0, int S1, int H0, int H1>
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) {
  f2 r;
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=v"(r) : "v"(a), "v"(b), "n"(S0), "n"(S1), "n"(H0), "n"(H1));
  return r;
}
template <int S0, int S1, int H0, int H1>
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) {      // src2 natural
  f2 r;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[%4,%5,0] op_sel_hi:[%6,%7,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(S0), "n"(S1), "n"(H0), "n"(H1));
  return r;
}
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// counters[op 3][form 16][half 2][quarter 4]
__global__ void __launch_bounds__(512) k_forms(unsigned* counters, int iters, int with_mfma, float* sink) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) lds[i] = 1.0f + 0.001f * i;
  __syncthreads();
  if (w < 4) {                                            // matrix waves
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
    if (with_mfma == 1)
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
      }
    else if (with_mfma == 2) {                            // v_mfma_f32_32x32x2_f32 (the dense contraction's instruction)
      typedef float f16v __attribute__((ext_vector_type(16)));
      f16v big = {0};
      for (int it = 0; it < iters * 24; ++it) big = __builtin_amdgcn_mfma_f32_32x32x2f32(0.01f * lane, 0.02f * lane, big, 0, 0, 0);
      acc[0][0] = big[0] + big[15];
    } else if (with_mfma == 3) {                          // v_mfma_f32_16x16x4_f32
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(0.01f * lane, 0.02f * lane, acc[t], 0, 0, 0);
      }
    } else if (with_mfma == 4) {                          // v_mfma_f32_16x16x16_f16 (gfx90a-era instruction)
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      const h4 ha = {(_Float16)0.5f, (_Float16)0.25f, (_Float16)(0.01f * lane), (_Float16)1.f}, hb = {(_Float16)0.125f, (_Float16)2.f, (_Float16)0.5f, (_Float16)(0.02f * lane)};
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, acc[t], 0, 0, 0);
      }
    } else if (with_mfma == 6) {                          // v_mfma_f32_16x16x32_f16 (gfx950)
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      h8 ha, hb;
      for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(0.01f * (lane + e)); hb[e] = (_Float16)(0.02f * (lane - e)); }
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[t], 0, 0, 0);
      }
    } else if (with_mfma == 7) {                          // v_mfma_i32_16x16x64_i8 (gfx950)
      typedef int i4v __attribute__((ext_vector_type(4)));
      i4v ia = {lane, lane + 1, lane + 2, lane + 3}, ib = {lane * 3, 7, lane, 11}, iacc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) iacc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, iacc[t], 0, 0, 0);
      }
      acc[0][0] = (float)(iacc[0][0] + iacc[1][1] + iacc[2][2] + iacc[3][3]);
    } else if (with_mfma == 8) {                          // v_mfma_f32_16x16x16_bf16 (the gfx942-era 16 x 16 bf16 shape, 4 values per lane)
      typedef __bf16 b4 __attribute__((ext_vector_type(4)));
      b4 ba, bb;
      for (int e = 0; e < 4; ++e) { ba[e] = (__bf16)(0.01f * (lane + e)); bb[e] = (__bf16)(0.02f * (lane - e)); }
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short __attribute__((ext_vector_type(4))), ba), __builtin_bit_cast(short __attribute__((ext_vector_type(4))), bb), acc[t], 0, 0, 0);
      }
    } else if (with_mfma == 5) {                          // v_mfma_f32_32x32x16_bf16 (gfx950)
      typedef float f16v __attribute__((ext_vector_type(16)));
      f16v big = {0};
      for (int it = 0; it < iters * 12; ++it) big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big, 0, 0, 0);
      acc[0][0] = big[0] + big[15];
    } else
      for (int it = 0; it < iters * 24; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = acc[t] * 1.0001f + 0.5f;
      }
    sink[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  // test waves
  unsigned bad[3][16][2] = {};
  bool have = false; float smp[8] = {};
  for (int it = 0; it < iters; ++it) {
    // operands: non-zero, lane and iteration dependent, exactly representable products are not needed (same rounding)
    const float l0 = lds[(lane * 7 + it * 13) & 4095], l1 = lds[(lane * 11 + it * 17 + 5) & 4095];
    const f2 a = {l0, -l1 * 0.75f}, b = {l1 + 0.25f, l0 * 1.5f}, c = {0.125f * l0, -0.375f * l1};
#define DO_FORM(S0, S1, H0, H1)                                                                              \
    {                                                                                                        \
      constexpr int F = S0 + 2 * S1 + 4 * H0 + 8 * H1;                                                       \
      const float a_lo = S0 ? a.y : a.x, a_hi = H0 ? a.y : a.x, b_lo = S1 ? b.y : b.x, b_hi = H1 ? b.y : b.x; \
      const f2 m = pk_mul<S0, S1, H0, H1>(a, b);                                                             \
      const f2 s = pk_add<S0, S1, H0, H1>(a, b);                                                             \
      const f2 f = pk_fma<S0, S1, H0, H1>(a, b, c);                                                          \
      if (__float_as_uint(m.x) != __float_as_uint(smul(a_lo, b_lo)) && !have) {                              \
        have = true; smp[0] = F; smp[1] = a.x; smp[2] = a.y; smp[3] = b.x; smp[4] = b.y; smp[5] = m.x; smp[6] = m.y; smp[7] = lane; }  \
      bad[0][F][0] += __float_as_uint(m.x) != __float_as_uint(smul(a_lo, b_lo));                            \
      bad[0][F][1] += __float_as_uint(m.y) != __float_as_uint(smul(a_hi, b_hi));                            \
      bad[1][F][0] += __float_as_uint(s.x) != __float_as_uint(sadd(a_lo, b_lo));                            \
      bad[1][F][1] += __float_as_uint(s.y) != __float_as_uint(sadd(a_hi, b_hi));                            \
      bad[2][F][0] += __float_as_uint(f.x) != __float_as_uint(sfma(a_lo, b_lo, c.x));                       \
      bad[2][F][1] += __float_as_uint(f.y) != __float_as_uint(sfma(a_hi, b_hi, c.y));                       \
    }
    FORM_LIST(DO_FORM)
#undef DO_FORM
  }
  if (have && atomicAdd(&counters[3 * 16 * 2 * 4], 1u) < 8) {
    const unsigned slot = atomicAdd(&counters[3 * 16 * 2 * 4 + 1], 1u);
    if (slot < 8) for (int i = 0; i < 8; ++i) sink[(size_t)gridDim.x * 512 + slot * 8 + i] = smp[i];
  }
  for (int op = 0; op < 3; ++op)
    for (int F = 0; F < 16; ++F)
      for (int h = 0; h < 2; ++h)
        if (bad[op][F][h]) atomicAdd(&counters[((op * 16 + F) * 2 + h) * 4 + (lane >> 4)], bad[op][F][h]);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int with_mfma = argc > 2 ? atoi(argv[2]) : 1;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * (argc > 3 ? atoi(argv[3]) : 1);
  unsigned* cnt; float* sink;
  hipMalloc(&cnt, (3 * 16 * 2 * 4 + 2) * 4); hipMemset(cnt, 0, (3 * 16 * 2 * 4 + 2) * 4); hipMalloc(&sink, ((size_t)grid * 512 + 64) * 4);
  hipLaunchKernelGGL(k_forms, dim3(grid), dim3(512), 0, 0, cnt, iters, with_mfma, sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  std::vector<unsigned> h(3 * 16 * 2 * 4);
  hipMemcpy(h.data(), cnt, h.size() * 4, hipMemcpyDeviceToHost);
  const char* ops[3] = {"mul", "add", "fma"}; const char code[4] = {'L', 'X', 'N', 'H'};      // (sel, hi): (0,0) L, (1,0) X, (0,1) N, (1,1) H
  long total = 0;
  printf("grid %d x 512 threads, %d iterations per test wave, matrix waves run %s\n", grid, iters, with_mfma == 1 ? "v_mfma_f32_16x16x32_bf16" : with_mfma == 2 ? "v_mfma_f32_32x32x2_f32" : with_mfma == 3 ? "v_mfma_f32_16x16x4_f32" : with_mfma == 4 ? "v_mfma_f32_16x16x16_f16" : with_mfma == 5 ? "v_mfma_f32_32x32x16_bf16" : with_mfma == 6 ? "v_mfma_f32_16x16x32_f16" : with_mfma == 7 ? "v_mfma_i32_16x16x64_i8" : with_mfma == 8 ? "v_mfma_f32_16x16x16_bf16" : "vector ALU work");
  for (int op = 0; op < 3; ++op)
    for (int F = 0; F < 16; ++F) {
      const int s0 = F & 1, s1 = (F >> 1) & 1, h0 = (F >> 2) & 1, h1 = (F >> 3) & 1;
      unsigned n = 0; for (int i = 0; i < 8; ++i) n += h[(op * 16 + F) * 8 + i];
      total += n;
      if (n) {
        printf("  v_pk_%s_f32 op_sel:[%d,%d] op_sel_hi:[%d,%d] (%s:%c%c): low half wrong by lane quarter %u %u %u %u, high half %u %u %u %u\n", ops[op], s0, s1, h0, h1,
               ops[op], code[s0 + 2 * h0], code[s1 + 2 * h1], h[(op * 16 + F) * 8 + 0], h[(op * 16 + F) * 8 + 1], h[(op * 16 + F) * 8 + 2], h[(op * 16 + F) * 8 + 3],
               h[(op * 16 + F) * 8 + 4], h[(op * 16 + F) * 8 + 5], h[(op * 16 + F) * 8 + 6], h[(op * 16 + F) * 8 + 7]);
      }
    }
  { float sm[64]; hipMemcpy(sm, sink + (size_t)grid * 512, 256, hipMemcpyDeviceToHost);
    unsigned ns[2]; hipMemcpy(ns, cnt + 3 * 16 * 2 * 4, 8, hipMemcpyDeviceToHost);
    for (unsigned i = 0; i < ns[1] && i < 4; ++i)
      printf("  sample (v_pk_mul_f32, form %d, lane %d): a = (%.6g, %.6g)  b = (%.6g, %.6g)  result = (%.6g, %.6g)\n", (int)sm[i * 8], (int)sm[i * 8 + 7], sm[i * 8 + 1], sm[i * 8 + 2], sm[i * 8 + 3], sm[i * 8 + 4], sm[i * 8 + 5], sm[i * 8 + 6]); }
  printf("total mismatches: %ld of %ld checks\n", total, (long)grid * 256 * iters * 96);
  return 0;
}
