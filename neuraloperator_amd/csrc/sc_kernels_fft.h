// sc_kernels_fft.h -- power-of-two fast path (fused, pruned 2-D FFT kernels).
// Placeholder until the fused kernels land: every plan takes the generic passes.
#pragma once
#include <string>
#include <vector>

#include "sc_device.h"

struct Fft2dPlan {
  int dummy = 0;
};

static inline bool fft2d_plan_init(Fft2dPlan*, int, const int64_t*, const int64_t*, double, double,
                                   std::vector<void*>*, std::string*) {
  return false;
}
static inline size_t fft2d_workspace_bytes(const Fft2dPlan*, int64_t) { return 0; }
static inline int fft2d_forward(const Fft2dPlan*, int, const float*, cf32*, int64_t, void*, sc_stream_t,
                                std::string*) {
  return 1;
}
static inline int fft2d_inverse(const Fft2dPlan*, int, const cf32*, const float*, int64_t, float*, int64_t,
                                void*, sc_stream_t, std::string*) {
  return 1;
}
static inline const char* fft2d_kernel_name(int) { return ""; }
