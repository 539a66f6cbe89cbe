// sc_kernels_peer.h -- the exchange step of the mode-parallel layer as PEER STORES (round 5, opt-in).
//
// The all-to-all of BASELINE configs[3] on 8 GPUs moves 4.46 MB per rank and direction, 557 KB per peer: 3.6 us on one
// xGMI link -- the collective is latency, not bandwidth ("or direct xGMI peer writes").  Here every rank
// owns a WINDOW (fine-grained device memory, mapped into every peer of the group through HIP IPC) and an exchange is
//   k_peer_put   block p of the send buffer -> slot `rank` of peer p's window, 16 bytes per lane straight over the
//                fabric; the LAST workgroup (device-scope ticket) advances this rank's epoch and writes it, with
//                system-scope release, into slot `rank` of every peer's flag array;
//   k_peer_wait  ONE workgroup waits (system-scope acquire loads + s_sleep) until all P flags of THIS rank carry the epoch;
//   k_peer_copy  the window -> the caller's receive tensor.
// Three plain launches on the caller's stream: stream-ordered like any kernel, capturable into a hipGraph (the epoch
// lives in device memory and advances per launch, so a replay signals a fresh value).  The window may be overwritten
// by a peer's NEXT exchange of the same kind only after that peer has waited for data this rank sent AFTER its
// k_peer_copy (the layer's exchanges alternate directions), and the two parities of a window alternate anyway.
// No reference counterpart (neuralop/mpu/helpers.py:81-99 is an unused torch.distributed all-to-all).
#pragma once
#include "sc_device.h"

struct PeerArgs {
  const sc_f4* send;            // [P][block16] 16-byte units
  sc_f4* peer_win[8];           // peer p's window base for this parity (own included): [P][block16]
  unsigned long long* peer_flag[8];   // peer p's flag array [P]
  const sc_f4* my_win;          // this rank's window (k_peer_copy)
  unsigned long long* my_flag;  // this rank's flag array [P]
  unsigned long long* epoch;    // this rank's epoch counter (device memory)
  unsigned int* ticket;         // workgroup ticket of k_peer_put
  unsigned int* error;          // set by k_peer_wait when its spin budget ran out (read by sc_peer_window_control)
  const unsigned long long* spin_budget;   // ticks of the 100 MHz wall clock k_peer_wait may spin; 0 = no bound
  sc_f4* recv;                  // [P][block16]
  long long block16;            // 16-byte units per block
  int P, rank, wg_per_peer;
};

#ifndef SC_EMU
SC_DEVICE void peer_flag_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
SC_DEVICE unsigned long long peer_flag_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
SC_DEVICE void peer_fence_system() { __threadfence_system(); }
SC_DEVICE unsigned int peer_ticket(unsigned int* t) { return atomicAdd(t, 1u); }
SC_DEVICE void peer_sleep() { __builtin_amdgcn_s_sleep(8); }
SC_DEVICE unsigned long long peer_clock() { return wall_clock64(); }      // constant 100 MHz
#else
inline void peer_flag_store(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline unsigned long long peer_flag_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void peer_fence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned int peer_ticket(unsigned int* t) { return __atomic_fetch_add(t, 1u, __ATOMIC_SEQ_CST); }
inline void peer_sleep() {}
inline unsigned long long peer_clock() { static unsigned long long t = 0; return __atomic_add_fetch(&t, 1ull, __ATOMIC_RELAXED); }
#endif

// grid = P * wg_per_peer workgroups: workgroup (p, c) copies chunk c of block p
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_peer_put(PeerArgs g) {
  SC_SHARED unsigned int last;
  const int tid = SC_TID;
  const int p = (int)SC_BID_X / g.wg_per_peer, c = (int)SC_BID_X - p * g.wg_per_peer;
  const long long per = (g.block16 + g.wg_per_peer - 1) / g.wg_per_peer;
  const long long lo = (long long)c * per, hi = lo + per < g.block16 ? lo + per : g.block16;
  const sc_f4* src = g.send + (long long)p * g.block16;
  sc_f4* dst = g.peer_win[p] + (long long)g.rank * g.block16;
  for (long long i = lo + tid; i < hi; i += 256) dst[i] = src[i];
  peer_fence_system();                                 // this thread's stores are visible system-wide ...
  SC_SYNC();                                           // ... for every thread of the workgroup
  if (tid == 0) last = peer_ticket(g.ticket) == (unsigned)(g.P * g.wg_per_peer - 1) ? 1u : 0u;
  SC_SYNC();
  if (last) {                                          // every workgroup of this launch has fenced its stores
    SC_SHARED unsigned long long e_sh;
    if (tid == 0) {
      *g.ticket = 0;                                   // (the next launch on this stream starts behind this one)
      e_sh = *g.epoch + 1;
      *g.epoch = e_sh;                                 // read by k_peer_wait, the next launch on this stream
    }
    SC_SYNC();
    peer_fence_system();
    if (tid < g.P) peer_flag_store(g.peer_flag[tid] + g.rank, e_sh);
  }
}

// ONE workgroup waits for the P flags of this rank (system-scope acquire loads + s_sleep): a spinning kernel must not
// fill the device -- on a shared device (tests: several ranks on one GPU) the peers' k_peer_put could not start behind
// a grid of spinning workgroups, and on its own device it would starve the rank's other streams
SC_GLOBAL void SC_LAUNCH_BOUNDS(64)
k_peer_wait(PeerArgs g) {
  const int tid = SC_TID;
  const unsigned long long want = *g.epoch;            // advanced by this rank's k_peer_put, earlier on this stream
  if (tid < g.P) {
    // bounded when the host set a budget (a dead peer or a window that is not coherent must end in an error
    // the host can read, not in a stream that never drains): the set-up self-test runs with 2 s, steps with the
    // budget the layer was given (default: none -- a peer that is merely late must not corrupt a step)
    const unsigned long long budget = *g.spin_budget, t0 = peer_clock();
    while (peer_flag_load(g.my_flag + tid) < want) {
      peer_sleep();
      if (budget && peer_clock() - t0 > budget) {
        *g.error = 1u + (unsigned)tid;                 // which peer's flag never came (any of them, if several)
        break;
      }
    }
  }
  peer_fence_system();
}

// window -> receive tensor (behind k_peer_wait on the stream): four 16-byte loads in flight per lane -- the window is
// fine-grained (uncached) memory, its read latency is the copy's time
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_peer_copy(PeerArgs g, int n_wg) {
  const int tid = SC_TID;
  const long long total = (long long)g.P * g.block16, stride = (long long)n_wg * 256;
  long long i = (long long)SC_BID_X * 256 + tid;
  for (; i + 3 * stride < total; i += 4 * stride) {
    const sc_f4 a = g.my_win[i], b = g.my_win[i + stride], c = g.my_win[i + 2 * stride], d = g.my_win[i + 3 * stride];
    g.recv[i] = a;
    g.recv[i + stride] = b;
    g.recv[i + 2 * stride] = c;
    g.recv[i + 3 * stride] = d;
  }
  for (; i < total; i += stride) g.recv[i] = g.my_win[i];
}
