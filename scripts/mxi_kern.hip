// device side of scripts/ubench_mxi.cpp: the three instantiations of k_fft2d_inv_mx + the comparison kernel, built as a
// stand-alone code object (scripts/mxi_variants.sh) so that the generated ISA can be edited before it is assembled
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include "../neuraloperator_amd/csrc/sc_kernels_fft3mx.h"
template __global__ void k_fft2d_inv_mx<64>(const cf32*, sc_bf16*, const float*, int, const cf32*, const cf32*, const uint16_t*, int, int, float, float, F3Shard, int64_t, int);
template __global__ void k_fft2d_inv_mx<128>(const cf32*, sc_bf16*, const float*, int, const cf32*, const cf32*, const uint16_t*, int, int, float, float, F3Shard, int64_t, int);
template __global__ void k_fft2d_inv_mx<256>(const cf32*, sc_bf16*, const float*, int, const cf32*, const cf32*, const uint16_t*, int, int, float, float, F3Shard, int64_t, int);
extern "C" __global__ void k_cmp(const uint16_t* a, const uint16_t* b, int per_image, unsigned* bad_per_image) {
  const int img = blockIdx.x;
  unsigned n = 0;
  for (int i = threadIdx.x; i < per_image; i += blockDim.x) n += a[(size_t)img * per_image + i] != b[(size_t)img * per_image + i];
  if (n) atomicAdd(&bad_per_image[img], n);
}
