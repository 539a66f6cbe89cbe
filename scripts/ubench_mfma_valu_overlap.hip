// Do matrix instructions and vector-ALU instructions of DIFFERENT waves of a SIMD overlap on MI355X?  (round 6: k_pmlp_bwd at
// two waves per SIMD ran exactly as fast as at one -- 1013.6 vs 1012.3 us -- although a wave's ~16 k MFMA cycles and ~15 k
// vector cycles per tile should then hide each other.)
// Workgroups of 8 waves = two per SIMD: waves 0-3 loop over one matrix instruction, waves 4-7 over dependent-free
// v_fma_f32 (or v_pk_fma_f32).  Timed: matrix waves alone, vector waves alone, both.  both ~ max(a, b): separate
// execution resources; both ~ a + b: shared.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma_valu_overlap.hip -o ovl && ./ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int VK>
__global__ void __launch_bounds__(512) k_ovl(float* sink, int mi, int vi) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  if (w < 4) {
    if (mi == 0) return;
    if (KIND == 0) {                                       // v_mfma_f32_32x32x2_f32: 4096 flop
      f16v a0 = {0}, a1 = {0};
      for (int i = 0; i < mi; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.01f * lane, 0.5f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(0.02f * lane, 0.25f, a1, 0, 0, 0);
      }
      sink[blockIdx.x * 512 + tid] = a0[0] + a1[3];
    } else if (KIND == 1) {                                // v_mfma_f32_16x16x32_bf16: 16384 flop
      f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
      bf8 x, y;
      for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.01f * (lane + e)); y[e] = (__bf16)(0.02f * (lane - e)); }
      for (int i = 0; i < mi; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
      }
      sink[blockIdx.x * 512 + tid] = a0[0] + a1[3];
    } else {                                               // v_mfma_f32_16x16x4_f32: 2048 flop
      f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
      for (int i = 0; i < mi; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.01f * lane, 0.5f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(0.02f * lane, 0.25f, a1, 0, 0, 0);
      }
      sink[blockIdx.x * 512 + tid] = a0[0] + a1[3];
    }
    return;
  }
  if (vi == 0) return;
  if (VK == 0) {                                           // 16 independent v_fma_f32 chains
    float r[16];
    for (int k = 0; k < 16; ++k) r[k] = 0.001f * (lane + k);
    const float m = 1.0001f, c = 0.37f;
    for (int i = 0; i < vi; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) r[k] = __builtin_fmaf(r[k], m, c);
    }
    float s = 0; for (int k = 0; k < 16; ++k) s += r[k];
    sink[blockIdx.x * 512 + tid] = s;
  } else if (VK == 1) {                                    // 8 independent v_pk_fma_f32 chains
    f2 r[8];
    for (int k = 0; k < 8; ++k) r[k] = f2{0.001f * (lane + k), 0.002f * lane};
    const f2 m = {1.0001f, 0.9999f}, c = {0.37f, 0.11f};
    for (int i = 0; i < vi; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = __builtin_elementwise_fma(r[k], m, c);
    }
    f2 s = {0, 0}; for (int k = 0; k < 8; ++k) s += r[k];
    sink[blockIdx.x * 512 + tid] = s.x + s.y;
  } else {                                                 // integer / logic (v_xor, v_add_u32): not the FMA datapath
    unsigned r[16];
    for (int k = 0; k < 16; ++k) r[k] = lane * 2654435761u + k;
    for (int i = 0; i < vi; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) r[k] = (r[k] ^ (r[k] >> 3)) + 0x9e3779b9u;
    }
    unsigned s = 0; for (int k = 0; k < 16; ++k) s ^= r[k];
    sink[blockIdx.x * 512 + tid] = (float)s;
  }
}

template <int KIND, int VK>
static void run(const char* mname, const char* vname, float* sink, int grid, int mi, int vi) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float t[3];
  const int cfg[3][2] = {{mi, 0}, {0, vi}, {mi, vi}};
  for (int c = 0; c < 3; ++c) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k_ovl<KIND, VK>), dim3(grid), dim3(512), 0, 0, sink, cfg[c][0], cfg[c][1]);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&t[c], e0, e1);
    }
  }
  printf("%-26s + %-14s: matrix alone %7.3f ms, vector alone %7.3f ms, both %7.3f ms  (max %.3f, sum %.3f)\n", mname, vname, t[0], t[1], t[2],
         t[0] > t[1] ? t[0] : t[1], t[0] + t[1]);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount;
  float* sink; hipMalloc(&sink, (size_t)grid * 512 * 4);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_ovl<0, 0>), dim3(grid), dim3(512), 0, 0, sink, 20000, 20000);   // settle clocks
  hipDeviceSynchronize();
  run<0, 0>("v_mfma_f32_32x32x2_f32", "v_fma_f32", sink, grid, 20000, 40000);
  run<0, 1>("v_mfma_f32_32x32x2_f32", "v_pk_fma_f32", sink, grid, 20000, 40000);
  run<0, 2>("v_mfma_f32_32x32x2_f32", "v_xor/v_add_u32", sink, grid, 20000, 40000);
  run<2, 0>("v_mfma_f32_16x16x4_f32", "v_fma_f32", sink, grid, 40000, 40000);
  run<1, 0>("v_mfma_f32_16x16x32_bf16", "v_fma_f32", sink, grid, 80000, 40000);
  run<1, 1>("v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32", sink, grid, 80000, 40000);
  return 0;
}
