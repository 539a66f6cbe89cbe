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
#include <type_traits>

#ifndef SC_EMU
// --------------------------------------------------------------------------- HIP (product)
#include <mutex>
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
// a lambda that is called from several places and must still be folded into its caller (by-reference captures of
// register arrays become scratch memory otherwise)
#define SC_ALWAYS_INLINE_LAMBDA __attribute__((always_inline))
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
// no memory operation is moved across this point by the compiler (no instruction is emitted)
#define SC_COMPILER_FENCE() asm volatile("" ::: "memory")
// returns x, but opaque to the optimiser: address arithmetic that depends on it cannot be hoisted
// above this point (epilogue store addresses computed -- and spilled -- before the main loop otherwise)
SC_DEVICE int sc_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// the same for a wave-uniform value (stays in an SGPR): loop-invariant address terms derived from it are recomputed
// where they are used instead of being hoisted into dozens of live registers
SC_DEVICE int sc_opaque_s(int x) {
  asm volatile("" : "+s"(x));
  return x;
}
// streaming (non-temporal) access to the 0.5 GB real tensors: a plain store leaves up to 256 MB of
// dirty Infinity-Cache lines whose write-back the NEXT kernel pays for (+57 us on a 537 MB reader,
// profiles/r01_writeback_ubench.txt); nt stores drain to HBM while the producing kernel computes
#define SC_STORE_STREAM(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define SC_LOAD_STREAM(ptr) __builtin_nontemporal_load(ptr)
#define SC_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
  type* name = reinterpret_cast<type*>(name##_raw)


// ---- LDS-DMA pipeline vocabulary (sc_kernels_gemm8.h) ---------------------------------------------------------
// 16 bytes per lane straight from global memory into LDS, no VGPR round trip: the wave writes ONE contiguous 1 KiB
// piece at the wave-uniform LDS address `lbase` (lane l lands at lbase + 16 l); the global source is per lane.
// hipcc neither counts these operations nor orders ds_reads behind them: the kernel waits with its own counted
// s_waitcnt vmcnt(N) and a raw s_barrier (cdna_hip_programming.md 5, "Pipelining across barriers").
#define SC_GLDS16(gptr, lbase)                                                                   \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),        \
                                   (__attribute__((address_space(3))) void*)(lbase), 16, 0, 0)
template <int N>
SC_DEVICE void sc_wait_vmcnt() {        // at most N of this wave's vector-memory operations still in flight
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define SC_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// s_barrier without the vmcnt(0) drain __syncthreads() carries while an LDS-DMA is outstanding
#define SC_BARRIER_RAW() __builtin_amdgcn_s_barrier()
typedef float sc_f4 __attribute__((ext_vector_type(4)));

// ---- cross-lane 2 x 2 transposes (gfx950: v_permlane16_swap_b32 / v_permlane32_swap_b32) ------------------------
// sc_swap16(a, b): lanes with bit 4 SET exchange their `a` with the `b` of the lane 16 below
//   (lane l, bit 4 clear:  b <- a of lane l + 16;   lane l + 16:  a <- b of lane l).
// sc_swap32(a, b): the same across bit 5 (lanes 32..63 of a <-> lanes 0..31 of b).
// One VALU instruction, no LDS.  Reduce-scatter of two values over a lane pair:  swap(a, b); s = a + b  leaves
// a_l + a_partner in the lower lane and b_l + b_partner in the upper one (sc_kernels_fft3.h, last row stage).
SC_DEVICE void sc_swap16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
SC_DEVICE void sc_swap32(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// ---- a load the compiler does not track (persistent store-bound kernels: the next work item's few inputs are
// requested while this item's rows are still being stored).  vmcnt counts loads and stores in issue order on gfx9, and
// hipcc's wait insertion, merging the loop's entry paths, drains EVERYTHING (s_waitcnt vmcnt(0): all stores of the
// previous item included) before the first use of a register that a tracked load of the previous iteration wrote --
// a full store drain per item (measured on the 128 x 128 plane kernels: 428 -> 587 us).  Issued as inline assembly the
// load is invisible to that pass; the kernel waits with its own COUNTED s_waitcnt ("at most N operations outstanding",
// N = the operations issued after the loads) and then passes every loaded value through sc_landed(), an empty
// assembly statement that is ordered after the wait and that every use depends on.  Extra operations in flight that
// the compiler does not know of can only make its own counted waits wait longer, never shorter.
// (sc_gload8_untracked / sc_landed: below, after cf32; the counted wait is sc_wait_vmcnt<N>())

// float add into LDS shared by the waves of a workgroup: ds_add_f32 (no return value)
#define SC_LDS_ADD(ptr, val) atomicAdd((ptr), (val))

typedef hipStream_t sc_stream_t;

#define SC_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)

// compute units of the current device (256 on MI355X): the grid of a PERSISTENT kernel is workgroups-per-CU x this
static inline int sc_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// ---- a second stream of the engine's own (session 2): INDEPENDENT small launches of one host call (the factor
// gradients of the Tucker chain beside its activation products) run side by side instead of in a row -- each of them
// alone leaves most of the chip waiting on its own latencies.  sc_side_fork: the side stream waits for everything
// issued to `main` so far; sc_side_join: `main` waits for everything issued to the side stream so far.  Events only,
// no host synchronisation, so the pattern also records into a hipGraph as a fork / join.  One set per device, created
// on first use; SC_NO_SIDE_STREAM=1 (environment) or any creation failure -> nullptr = the caller stays on one stream.
// Thread safety: creation is once per device under a mutex; fork / join take the set's own mutex and a
// FRESH slot of an 8-deep event ring under it, so two host threads that issue on the same device serialise their
// record + wait pairs instead of interleaving them (hipEventRecord + hipStreamWaitEvent of one pair stay atomic).
struct ScSide {
  hipStream_t stream;
  hipEvent_t fork_ev[8], join_ev[8];
  unsigned n_fork, n_join;
  std::mutex mu;
};
static inline ScSide* sc_side_get() {
  static ScSide* cached[64] = {nullptr};
  static bool tried[64] = {false};
  static std::mutex create_mu;
  static const bool off = [] { const char* e = getenv("SC_NO_SIDE_STREAM"); return e && e[0] == '1'; }();
  if (off) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(create_mu);
  if (!tried[dev]) {
    tried[dev] = true;
    ScSide* s = new ScSide();
    bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 8; ++i)
      ok = hipEventCreateWithFlags(&s->fork_ev[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&s->join_ev[i], hipEventDisableTiming) == hipSuccess;
    s->n_fork = s->n_join = 0;
    cached[dev] = ok ? s : nullptr;
  }
  return cached[dev];
}
static inline bool sc_side_fork(ScSide* s, hipStream_t main) {
  std::lock_guard<std::mutex> lock(s->mu);
  hipEvent_t e = s->fork_ev[s->n_fork++ & 7];
  return hipEventRecord(e, main) == hipSuccess && hipStreamWaitEvent(s->stream, e, 0) == hipSuccess;
}
static inline bool sc_side_join(ScSide* s, hipStream_t main) {
  std::lock_guard<std::mutex> lock(s->mu);
  hipEvent_t e = s->join_ev[s->n_join++ & 7];
  return hipEventRecord(e, s->stream) == hipSuccess && hipStreamWaitEvent(main, e, 0) == hipSuccess;
}
// joins on scope exit once armed: an early error return of a forked call must not leave the side stream running over
// a workspace the caller frees next, nor a hipGraph capture with an unjoined fork
struct ScSideJoinGuard {
  ScSide* side;
  hipStream_t main;
  bool armed;
  ScSideJoinGuard(ScSide* s, hipStream_t m) : side(s), main(m), armed(false) {}
  ~ScSideJoinGuard() {
    if (side && armed) (void)sc_side_join(side, main);
  }
};

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
// per-wave scratch for the emulated cross-lane instructions: [wave][2][64] floats
float* wave_scratch();
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
#define SC_ALWAYS_INLINE_LAMBDA
#define SC_LAUNCH_BOUNDS_OCC(n, w)
#define SC_WAVE_SYNC() scemu::wave_barrier()   /* emulated lanes are free-running threads */
#define SC_UNIFORM(x) (x)
#define SC_SCHED_BARRIER() do { } while (0)
#define SC_COMPILER_FENCE() do { } while (0)
#define SC_STORE_STREAM(ptr, val) (*(ptr) = (val))
#define SC_LOAD_STREAM(ptr) (*(ptr))
inline int sc_opaque(int x) { return x; }
inline int sc_opaque_s(int x) { return x; }
#define SC_DYN_SHARED(type, name) type* name = reinterpret_cast<type*>(scemu::g_dyn_shared)


struct alignas(16) sc_f4 {
  float x, y, z, w;
};
// LDS-DMA emulated as a synchronous copy by the lane's own thread; the waits are no-ops, the raw barrier is the
// workgroup barrier -- ordering bugs that depend on a MISSING wait are therefore invisible here (the GPU tier
// and the counted-wait arithmetic in the kernel comments cover those), index maths and buffer rotation are not
#define SC_GLDS16(gptr, lbase) std::memcpy(reinterpret_cast<unsigned char*>(lbase) + 16 * (SC_TID & 63), (gptr), 16)
template <int N>
inline void sc_wait_vmcnt() {}
#define SC_WAIT_LGKM0() do { } while (0)
#define SC_BARRIER_RAW() scemu::barrier()

typedef void* sc_stream_t;

// v_permlane16_swap / v_permlane32_swap emulated through a per-wave scratch between two wave rendezvous (every lane
// of the wave must reach the call, as on the GPU where it is one wave instruction)
inline void sc_emu_swap(float& a, float& b, const int bit) {
  float* s = scemu::wave_scratch();
  const int lane = SC_TID & 63;
  s[lane] = a;
  s[64 + lane] = b;
  scemu::wave_barrier();
  if (lane & bit) a = s[64 + (lane ^ bit)];
  else b = s[lane ^ bit];
  scemu::wave_barrier();
}
inline void sc_swap16(float& a, float& b) { sc_emu_swap(a, b, 16); }
inline void sc_swap32(float& a, float& b) { sc_emu_swap(a, b, 32); }

// emulated lanes are free-running threads: a compare-exchange loop stands in for ds_add_f32
inline void sc_emu_lds_add(float* p, const float v) {
  uint32_t* u = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
  do {
    float f;
    std::memcpy(&f, &old, 4);
    f += v;
    std::memcpy(&nw, &f, 4);
  } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
#define SC_LDS_ADD(ptr, val) sc_emu_lds_add((ptr), (val))

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
// the emulated "chip" has ONE compute unit: persistent kernels then walk several work items per workgroup in the
// CPU tier, which is what their loops and cross-item prefetches need to be tested on
inline int sc_cu_count() { return 1; }
// emulated launches are synchronous: there is no second stream
struct ScSide {
  sc_stream_t stream;
};
inline ScSide* sc_side_get() { return nullptr; }
inline bool sc_side_fork(ScSide*, sc_stream_t) { return true; }
inline bool sc_side_join(ScSide*, sc_stream_t) { return true; }
struct ScSideJoinGuard {
  bool armed = false;
  ScSideJoinGuard(ScSide*, sc_stream_t) {}
};
#endif

// compile-time integer tag (selects a template body from a wave-uniform runtime value)
template <int N>
struct sc_int {
  static constexpr int value = N;
};

// --------------------------------------------------------------------------- bfloat16 storage
// (SC_PLAN_IO_BF16: real tensors cross HBM as bfloat16, all arithmetic stays fp32)
struct sc_bf16 {
  uint16_t v;
};
#ifndef SC_EMU
SC_DEVICE float sc_bits_to_f32(const uint32_t u) { return __uint_as_float(u); }
// round to nearest even, NaN -> quiet NaN: v_cvt_pk_bf16_f32 (same results as torch's float -> bfloat16)
SC_DEVICE uint16_t sc_f32_to_bf16_bits(const float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
#else
inline float sc_bits_to_f32(const uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline uint16_t sc_f32_to_bf16_bits(const float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;            // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                // round to nearest even
  return (uint16_t)(u >> 16);
}
#endif

// --------------------------------------------------------------------------- float16 rounding
// (fno_block_precision = "half" / "mixed": values are ROUNDED to float16 where the reference holds them in
// float16 / complex32 -- spectral_convolution.py:436-459, einsum_utils.py:10-36 -- and kept in fp32 storage)
#ifndef SC_EMU
SC_DEVICE float sc_round_f16(const float f) { return (float)(_Float16)f; }      // v_cvt_f16_f32, nearest even
#else
inline float sc_round_f16(const float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint32_t sign = u & 0x80000000u;
  uint32_t a = u & 0x7fffffffu;
  float r;
  if (a >= 0x7f800000u) {                                   // inf / nan
    r = f;
    return r;
  }
  if (a >= 0x477ff000u) {                                   // >= 65520: rounds to infinity
    a = 0x7f800000u;
  } else if (a >= 0x38800000u) {                            // normal half: keep 10 mantissa bits
    a += 0xfffu + ((a >> 13) & 1u);
    a &= ~0x1fffu;
  } else {                                                  // subnormal half: multiples of 2^-24
    float m;
    std::memcpy(&m, &a, 4);
    const float q = std::nearbyint(m * 16777216.f) / 16777216.f;   // default rounding mode: nearest even
    std::memcpy(&a, &q, 4);
  }
  a |= sign;
  std::memcpy(&r, &a, 4);
  return r;
}
#endif

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
#ifndef SC_EMU
SC_DEVICE cf32 sc_gload8_untracked(const cf32* p) {
  typedef float f2v_ __attribute__((ext_vector_type(2)));
  f2v_ v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return cf_make(v.x, v.y);
}
SC_DEVICE void sc_landed(cf32& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y));
}
SC_DEVICE float sc_gload4_untracked(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// also: "use" a value the compiler DOES track, here and now -- its own wait for the load that produced it is then
// placed in front of this statement (e.g. ahead of a loop) instead of at the first real use (inside the loop, where a
// s_waitcnt vmcnt(0) would drain the loop's stores on every trip)
SC_DEVICE void sc_landed(float& v) {
  asm volatile("" : "+v"(v));
}
#else
inline cf32 sc_gload8_untracked(const cf32* p) { return *p; }
inline float sc_gload4_untracked(const float* p) { return *p; }
inline void sc_landed(cf32&) {}
inline void sc_landed(float&) {}
#endif
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
// complex product on the packed-fp32 pipe: v_pk_mul_f32 + v_pk_fma_f32 (+ one v_xor for the sign
// unless the caller keeps the twiddle pre-arranged as (c, -s, s), see cf_mul_tw).  The plain struct
// formula compiles to 4-5 VALU instructions (pk_mul, 2 pk_fma, moves).
#ifndef SC_EMU
typedef float sc_f2 __attribute__((ext_vector_type(2)));
SC_DEVICE cf32 cf_mul_pk(const cf32 a, const cf32 b) {
  const sc_f2 av = {a.x, a.y}, ayx = {a.y, a.x}, bxx = {b.x, b.x}, nb = {-b.y, b.y};
  const sc_f2 r = __builtin_elementwise_fma(ayx, nb, av * bxx);
  return cf_make(r.x, r.y);
}
// twiddle c + i s held as (c, ns = -s, s)
SC_DEVICE cf32 cf_mul_tw(const cf32 a, const float c, const float ns, const float s) {
  const sc_f2 av = {a.x, a.y}, ayx = {a.y, a.x}, cc = {c, c}, nb = {ns, s};
  const sc_f2 r = __builtin_elementwise_fma(ayx, nb, av * cc);
  return cf_make(r.x, r.y);
}
#else
inline cf32 cf_mul_pk(const cf32 a, const cf32 b) { return cf_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
inline cf32 cf_mul_tw(const cf32 a, const float c, const float ns, const float s) {
  return cf_make(a.x * c + a.y * ns, a.y * c + a.x * s);
}
#endif
// ---- LDS reads the compiler may not pair up: hipcc's load/store optimiser merges two 8-byte LDS reads into one
// ds_read2_b64, which the LDS serves at HALF the rate of two ds_read_b64 (8 vs 2 + 2 cycles per wave instruction,
// MI355X_MICROARCH.md LDS table).  A volatile access through the LDS address space is left alone (the compiler still
// inserts the waits).  sc_lds_ld128x: 16-byte reads of two neighbouring complex values (ds_read_b128, full rate);
// the address must be 16-byte aligned.
#ifndef SC_EMU
SC_DEVICE cf32 sc_lds_ld64(const cf32* p) {
  typedef const volatile __attribute__((address_space(3))) unsigned long long* lp;
  const unsigned long long u = *(lp)(p);
  return cf_make(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}
SC_DEVICE void sc_lds_ld128(const cf32* p, cf32& a, cf32& b) {
  typedef float f4_ __attribute__((ext_vector_type(4)));
  typedef const volatile __attribute__((address_space(3))) f4_* lp;
  const f4_ v = *(lp)(p);
  a = cf_make(v.x, v.y);
  b = cf_make(v.z, v.w);
}
#else
inline cf32 sc_lds_ld64(const cf32* p) { return *p; }
inline void sc_lds_ld128(const cf32* p, cf32& a, cf32& b) {
  a = p[0];
  b = p[1];
}
#endif
// a * (c + i s) with the twiddle held as the plain pair t = (c, s): the sign of the cross term is an operand
// modifier of v_pk_fma_f32, so no third register is needed (A-B: SC_F3_TW1_CS)
#ifndef SC_EMU
SC_DEVICE cf32 cf_mul_cs(const cf32 a, const cf32 t) {
  const sc_f2 av = {a.x, a.y}, ayx = {a.y, a.x}, cc = {t.x, t.x}, nb = {-t.y, t.y};
  const sc_f2 r = __builtin_elementwise_fma(ayx, nb, av * cc);
  return cf_make(r.x, r.y);
}
#else
inline cf32 cf_mul_cs(const cf32 a, const cf32 t) { return cf_make(a.x * t.x - a.y * t.y, a.y * t.x + a.x * t.y); }
#endif
struct ctw3 {   // register-resident twiddle
  float c, ns, s;
};
struct ctw4 {   // LDS-resident twiddle, one 16-byte read
  float c, pad, ns, s;
};
SC_HD ctw3 ctw3_make(const cf32 t) {
  ctw3 r;
  r.c = t.x;
  r.ns = -t.y;
  r.s = t.y;
  return r;
}
SC_HD ctw4 ctw4_make(const cf32 t) {
  ctw4 r;
  r.c = t.x;
  r.pad = t.x;
  r.ns = -t.y;
  r.s = t.y;
  return r;
}
// a + (DIR i) b and a - (DIR i) b in ONE packed instruction each: fma(swap(b), (-/+1, +/-1), a).  Multiplying by +-1
// is exact, so the bits equal those of the add / subtract pair; written as separate per-lane add and subtract the
// compiler emits two packed adds plus two register moves to re-pair the halves (10 v_mov per 8-point DFT).
#ifndef SC_EMU
template <int DIR>
SC_HD cf32 cf_add_rot(const cf32 a, const cf32 b) {              // a + w b, w = DIR i
  const sc_f2 av = {a.x, a.y}, byx = {b.y, b.x}, k = {DIR < 0 ? 1.f : -1.f, DIR < 0 ? -1.f : 1.f};
  const sc_f2 r = __builtin_elementwise_fma(byx, k, av);
  return cf_make(r.x, r.y);
}
template <int DIR>
SC_HD cf32 cf_sub_rot(const cf32 a, const cf32 b) {              // a - w b
  const sc_f2 av = {a.x, a.y}, byx = {b.y, b.x}, k = {DIR < 0 ? -1.f : 1.f, DIR < 0 ? 1.f : -1.f};
  const sc_f2 r = __builtin_elementwise_fma(byx, k, av);
  return cf_make(r.x, r.y);
}
#else
template <int DIR>
inline cf32 cf_add_rot(const cf32 a, const cf32 b) {
  return DIR < 0 ? cf_make(a.x + b.y, a.y - b.x) : cf_make(a.x - b.y, a.y + b.x);
}
template <int DIR>
inline cf32 cf_sub_rot(const cf32 a, const cf32 b) {
  return DIR < 0 ? cf_make(a.x - b.y, a.y + b.x) : cf_make(a.x + b.y, a.y - b.x);
}
#endif
#ifndef SC_EMU
// explicit packed forms: the natural (re, im) pair is the vector -- left to the SLP vectoriser, freshly loaded
// values get paired ACROSS complex numbers and every first-stage butterfly pays register moves
SC_HD cf32 cf_add(const cf32 a, const cf32 b) {
  const sc_f2 av = {a.x, a.y}, bv = {b.x, b.y};
  const sc_f2 r = av + bv;
  return cf_make(r.x, r.y);
}
SC_HD cf32 cf_sub(const cf32 a, const cf32 b) {
  const sc_f2 av = {a.x, a.y}, bv = {b.x, b.y};
  const sc_f2 r = av - bv;
  return cf_make(r.x, r.y);
}
#else
SC_HD cf32 cf_add(const cf32 a, const cf32 b) { return cf_make(a.x + b.x, a.y + b.y); }
SC_HD cf32 cf_sub(const cf32 a, const cf32 b) { return cf_make(a.x - b.x, a.y - b.y); }
#endif
SC_HD cf32 cf_conj(const cf32 a) { return cf_make(a.x, -a.y); }
// multiply by -i  /  +i
SC_HD cf32 cf_mul_mi(const cf32 a) { return cf_make(a.y, -a.x); }
SC_HD cf32 cf_mul_pi(const cf32 a) { return cf_make(-a.y, a.x); }

// base + a 32-bit BYTE offset of the lane: with a wave-uniform base the access takes the scalar-base form
// (global_load_dword v, v_off, s[base:base+1]) -- written as base + 4 * (zero-extended index) the compiler cannot prove
// that the shifted offset still fits in 32 bits and spends a v_lshl_add_u64 per access (217 per tile in k_pblock_fwd)
template <typename T>
SC_HD T* sc_at(T* base, const uint32_t byte_off) {
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
  return (T*)((B*)base + byte_off);
}

// --------------------------------------------------------------------------- activation of the block epilogue
// erfc(|x|) by A&S 7.1.26: (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p |x|); |error| <= 1.5e-7 absolute.
//
// Round 6, second pass: the instruction count of the activation IS the run time of the fp32 pointwise kernels -- every
// vector instruction runs in front of their matrix instructions, not beside them (DESIGN 3.16 b), and k_pblock_fwd held
// 3400 vector instructions for 128 MFMAs.  Two changes, same function:
//   * t = v_rcp_f32 + one Newton step (3 instructions, <= 1 ulp) instead of the correctly rounded quotient of __frcp_rn
//     (v_div_scale x 2, v_rcp, 5 fma / mul, v_div_fmas, v_div_fixup: 11); exp(-x^2) = v_exp_f32(x^2 (-log2 e)) as before;
//   * the PAIR forms (sc_gelu_pair / sc_gelu_both_pair) evaluate two values with packed-fp32 instructions: the
//     polynomial, the Newton step and the products of two values per instruction; the two transcendentals and the sign
//     select stay per value.  12 / 13 instructions per value instead of 27 / 33.
// Every operation is an IEEE fma / mul / add or one of the two transcendentals, in the same order in the scalar and the
// pair form (packed and scalar fma / mul / add round alike): the bits do not depend on which form a kernel uses
// (tests/test_gpu_parity.py::test_one_activation_on_every_route_bit_for_bit; the host-emulation tier evaluates the
// scalar form everywhere).
// 1 + erf(z) = erfc(|z|) for z < 0 and 2 - erfc(|z|) for z >= 0: the negative tail is taken from erfc DIRECTLY, so it does
// not cancel (1 + erf loses every digit for v < -4) -- as fma(-cs, q, 1 + cs) with cs = copysign(1, v), exact.
#define SC_GELU_P   0.23164189298270723f      // 0.3275911 / sqrt 2: t = 1 / (1 + P |v|)
#define SC_GELU_E  (-0.72134752044448170368f) // -log2(e) / 2: exp(-v^2 / 2) = exp2(v v E)
#define SC_GELU_D   0.39894228040143267794f   // 1 / sqrt(2 pi)
#ifndef SC_EMU
SC_DEVICE float sc_rcp_nr(const float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.f), r, r);
}
SC_DEVICE float sc_exp2_hw(const float s) { return __builtin_amdgcn_exp2f(s); }
#else
inline float sc_rcp_nr(const float d) { return 1.f / d; }
inline float sc_exp2_hw(const float s) { return exp2f(s); }
#endif
// erfc(|v| / sqrt 2) and exp(-v^2 / 2)
SC_DEVICE void sc_erfc_core(const float v, float& q, float& e) {
  const float t = sc_rcp_nr(fmaf(fabsf(v), SC_GELU_P, 1.f));
  e = sc_exp2_hw((v * v) * SC_GELU_E);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  q = (p * t) * e;
}
// gelu(v) = 0.5 v (1 + erf(v / sqrt 2)).  Absolute error of the activation <= 0.5 |v| 1.5e-7 (+ fp32 round-off); every
// engine path -- the fused store paths, the pointwise kernels, the stand-alone k_epilogue pass -- evaluates this one
// function, so a shape change never changes the activation's bits.
SC_DEVICE float sc_gelu(const float v) {
  float q, e;
  sc_erfc_core(v, q, e);
  const float cs = copysignf(1.f, v);
  const float h2 = 0.5f * fmaf(-cs, q, 1.f + cs);          // (1 + erf(v / sqrt 2)) / 2
  return v * h2;
}
// gelu(v) AND its derivative Phi(v) + v phi(v) from ONE evaluation of the two transcendentals (round 6): the exponential
// of the erfc approximation at z = |v| / sqrt 2 is exp(-v^2 / 2) -- the density's own -- so the derivative costs two
// more vector instructions instead of a second reciprocal + two exponentials.
SC_DEVICE void sc_gelu_both(const float v, float& gelu, float& grad) {
  float q, e;
  sc_erfc_core(v, q, e);
  const float cs = copysignf(1.f, v);
  const float h2 = 0.5f * fmaf(-cs, q, 1.f + cs);
  gelu = v * h2;
  grad = fmaf(v, SC_GELU_D * e, h2);
}
// two values at a time
#ifndef SC_EMU
SC_DEVICE void sc_erfc_core2(const sc_f2 v, sc_f2& q, sc_f2& e) {
  const sc_f2 av = {fabsf(v.x), fabsf(v.y)};
  const sc_f2 one = {1.f, 1.f};
  const sc_f2 d = __builtin_elementwise_fma(av, sc_f2{SC_GELU_P, SC_GELU_P}, one);
  const sc_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const sc_f2 t = __builtin_elementwise_fma(__builtin_elementwise_fma(-d, r, one), r, r);
  const sc_f2 s = (v * v) * sc_f2{SC_GELU_E, SC_GELU_E};
  e = sc_f2{__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
  sc_f2 p = __builtin_elementwise_fma(sc_f2{1.061405429f, 1.061405429f}, t, sc_f2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, sc_f2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, sc_f2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, sc_f2{0.254829592f, 0.254829592f});
  q = (p * t) * e;
}
SC_DEVICE void sc_gelu_pair(float& a, float& b) {
  const sc_f2 v = {a, b};
  sc_f2 q, e;
  sc_erfc_core2(v, q, e);
  const sc_f2 cs = {copysignf(1.f, v.x), copysignf(1.f, v.y)};
  const sc_f2 h2 = sc_f2{0.5f, 0.5f} * __builtin_elementwise_fma(-cs, q, sc_f2{1.f, 1.f} + cs);
  const sc_f2 g = v * h2;
  a = g.x; b = g.y;
}
SC_DEVICE void sc_gelu_both_pair(const float va, const float vb, float& ga, float& gb, float& da, float& db) {
  const sc_f2 v = {va, vb};
  sc_f2 q, e;
  sc_erfc_core2(v, q, e);
  const sc_f2 cs = {copysignf(1.f, v.x), copysignf(1.f, v.y)};
  const sc_f2 h2 = sc_f2{0.5f, 0.5f} * __builtin_elementwise_fma(-cs, q, sc_f2{1.f, 1.f} + cs);
  const sc_f2 g = v * h2;
  const sc_f2 d = __builtin_elementwise_fma(v, sc_f2{SC_GELU_D, SC_GELU_D} * e, h2);
  ga = g.x; gb = g.y; da = d.x; db = d.y;
}
#else
inline void sc_gelu_pair(float& a, float& b) { a = sc_gelu(a); b = sc_gelu(b); }
inline void sc_gelu_both_pair(const float va, const float vb, float& ga, float& gb, float& da, float& db) {
  sc_gelu_both(va, ga, da);
  sc_gelu_both(vb, gb, db);
}
#endif

