// sc_device.h -- device-side vocabulary of the engine.
//
// Product build (hipcc, gfx950): plain HIP.
// SC_EMU build (g++, tests only): the SAME kernel source is compiled for the host and
// every workgroup is executed by real OS threads with a real barrier, so that index maths,
// LDS choreography and barrier placement of each kernel can be checked in the CPU-only CI
// tier before spending GPU minutes.  The emulation library is test infrastructure
// (tests/emu/), is never loaded by neuraloperator_amd and is not a fallback: the product
// raises if libsc_engine.so (the HIP build) is missing.
#pragma once

#include <stdint.h>
#include <stddef.h>

#ifndef SC_EMU
// --------------------------------------------------------------------------- HIP (product)
#include <hip/hip_runtime.h>

#define SC_GLOBAL __global__
#define SC_DEVICE __device__ __forceinline__
#define SC_HD __host__ __device__ __forceinline__
#define SC_SHARED __shared__
#define SC_SYNC() __syncthreads()
#define SC_TID ((int)threadIdx.x)
#define SC_BID_X ((int)blockIdx.x)
#define SC_BID_Y ((int)blockIdx.y)
#define SC_BID_Z ((int)blockIdx.z)
#define SC_LAUNCH_BOUNDS(n) __launch_bounds__(n)
// second argument = minimum waves per SIMD the register allocator must leave room for
#define SC_LAUNCH_BOUNDS_OCC(n, w) __launch_bounds__(n, w)
// exchange through LDS between lanes of ONE wave: no s_barrier needed (a wave's DS operations
// execute in order); drain the wave's own LDS queue and stop the compiler moving memory
// operations across the point.
#define SC_WAVE_SYNC()                                           \
  do {                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_wave_barrier();                             \
  } while (0)
// value known to be identical in every lane of the wave -> keep it in an SGPR
#define SC_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// the instruction scheduler may not move anything across this point
#define SC_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define SC_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
  type* name = reinterpret_cast<type*>(name##_raw)

typedef hipStream_t sc_stream_t;

#define SC_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)

#else
// --------------------------------------------------------------------------- host emulation
#include <cstdlib>
#include <cstring>
#include <cmath>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace scemu {
struct ThreadCtx {
  int tid;
  int bx, by, bz;
};
extern thread_local ThreadCtx g_ctx;
extern unsigned char* g_dyn_shared;
void barrier();
void wave_barrier();
// runs fn(arg) for every thread of every block; blocks sequentially, threads concurrently
void launch(dim3 grid, dim3 block, size_t shmem, void (*fn)(void*), void* arg);
}  // namespace scemu

#define SC_GLOBAL
#define SC_DEVICE inline
#define SC_HD inline
#define SC_SHARED static
#define SC_SYNC() scemu::barrier()
#define SC_TID (scemu::g_ctx.tid)
#define SC_BID_X (scemu::g_ctx.bx)
#define SC_BID_Y (scemu::g_ctx.by)
#define SC_BID_Z (scemu::g_ctx.bz)
#define SC_LAUNCH_BOUNDS(n)
#define SC_LAUNCH_BOUNDS_OCC(n, w)
#define SC_WAVE_SYNC() scemu::wave_barrier()   /* emulated lanes are free-running threads */
#define SC_UNIFORM(x) (x)
#define SC_SCHED_BARRIER() do { } while (0)
#define SC_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(scemu::g_dyn_shared)

typedef void* sc_stream_t;

// capture the arguments by value in a lambda and hand it to the thread pool
#define SC_LAUNCH(kernel, grid, block, shmem, stream, ...)                          \
  do {                                                                              \
    auto sc_fn_ = [=]() { kernel(__VA_ARGS__); };                                   \
    scemu::launch(grid, block, shmem,                                               \
                  [](void* p) { (*static_cast<decltype(sc_fn_)*>(p))(); }, &sc_fn_); \
  } while (0)

// the few HIP runtime calls the host side of the engine uses
typedef int hipError_t;
#define hipSuccess 0
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, sc_stream_t) { std::memset(d, v, n); return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
#endif

// compile-time integer tag (selects a template body from a wave-uniform runtime value)
template <int N>
struct sc_int {
  static constexpr int value = N;
};

// --------------------------------------------------------------------------- complex helpers
struct cf32 {
  float x, y;
};

SC_HD cf32 cf_make(float x, float y) {
  cf32 r;
  r.x = x;
  r.y = y;
  return r;
}
// acc += a * b
SC_HD void cf_mac(cf32& acc, const cf32 a, const cf32 b) {
  acc.x = fmaf(a.x, b.x, acc.x);
  acc.x = fmaf(-a.y, b.y, acc.x);
  acc.y = fmaf(a.x, b.y, acc.y);
  acc.y = fmaf(a.y, b.x, acc.y);
}
// acc += conj(a) * b
SC_HD void cf_mac_conj_a(cf32& acc, const cf32 a, const cf32 b) {
  acc.x = fmaf(a.x, b.x, acc.x);
  acc.x = fmaf(a.y, b.y, acc.x);
  acc.y = fmaf(a.x, b.y, acc.y);
  acc.y = fmaf(-a.y, b.x, acc.y);
}
SC_HD cf32 cf_mul(const cf32 a, const cf32 b) {
  return cf_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
SC_HD cf32 cf_add(const cf32 a, const cf32 b) { return cf_make(a.x + b.x, a.y + b.y); }
SC_HD cf32 cf_sub(const cf32 a, const cf32 b) { return cf_make(a.x - b.x, a.y - b.y); }
SC_HD cf32 cf_conj(const cf32 a) { return cf_make(a.x, -a.y); }
// multiply by -i  /  +i
SC_HD cf32 cf_mul_mi(const cf32 a) { return cf_make(a.y, -a.x); }
SC_HD cf32 cf_mul_pi(const cf32 a) { return cf_make(-a.y, a.x); }
