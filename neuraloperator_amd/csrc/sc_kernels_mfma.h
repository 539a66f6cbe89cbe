// sc_kernels_mfma.h -- the dense (i,o) x mode block GEMM of the spectral layer on the CDNA4
// matrix cores (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32).
//
//   C[p, q, m] = sum_r opA(A[p, r, m]) * opB(B[r, q, m])        (complex, m = Fourier mode)
//
// replaces tl.einsum('bixy,ioxy->boxy') and its two autograd einsums
// (spectral_convolution.py:21-46) for channel counts that fill MFMA tiles; everything else
// takes the lanes-are-modes VALU kernel in sc_kernels_generic.h (same results).
//
// Decomposition (MI355X-first, not a batched-GEMM library call):
//  * every operand keeps the reference's layout: the mode index is innermost, so for one mode
//    the (p, r) / (r, q) matrix elements are 8-byte islands M*8 bytes apart.  A workgroup
//    therefore owns a CONTIGUOUS RANGE of nm <= NM modes (M modes split evenly over the grid:
//    256 workgroups x 8.25 modes at the metric shape, one workgroup per CU, no tail) and reads
//    every (p, r) / (r, q) element of its range as ONE nm*8-byte segment: each byte of A and B
//    crosses HBM exactly once, L2 only has to merge the cache lines two neighbouring ranges
//    share (neighbouring ranges are mapped to the same XCD).
//  * the complex product is evaluated as a real GEMM with K doubled:
//        A'[p][2r+c] = (Re, Im)[c] of A[p][r]           (c = lane >> 5 of the MFMA A operand)
//        B'[2r+c][2q+d] = Re B, Im B, -Im B, Re B        (n = 2q+d = lane & 31 of the B operand)
//    so one v_mfma_f32_32x32x2_f32 consumes one r for 32 p x 16 q of one mode; conjugations
//    are sign masks on the operand registers.  fp32 MFMA is a k-ordered fmaf chain: results
//    are of fp32-roundoff class like the VALU kernel (MI355X_MICROARCH.md, Matrix cores).
//  * LDS stage = RC values of r for all nm modes, split into Re/Im planes with an odd mode
//    stride so that both the staging writes (lanes = (segment, mode)) and the MFMA operand
//    reads (lanes = p or q) are bank-conflict free; two stages ping-pong, the global loads of
//    stage k+1 are in flight while stage k feeds the matrix cores; one barrier per stage.
//  * 8 waves = 2 per SIMD.  A wave owns one 32 x 32 MFMA tile column set (row tile wp, columns
//    q = 16 (wq + 4u) .. +15) for all modes; for P = 32 two waves share a tile and split the r
//    values of every stage (even / odd), their partial sums meet in the epilogue; for P = 64
//    there is one wave per tile.  In-order issue means a wave's own
//    LDS reads / sign flips / address arithmetic cannot overlap its own MFMAs (measured 86-111
//    cycles per MFMA with one wave per SIMD against the 64-cycle issue rate); the partner wave's
//    MFMAs fill those slots.  MFMAs of one r are issued back to back, accumulators stay in
//    registers (NM * 16 per column group) for the whole r loop.
#pragma once
#include "sc_device.h"

#define SC_MG_RC 8
#ifndef SC_MG_ABLATE
#define SC_MG_ABLATE 0
#endif
#ifndef SC_MG_PF
#define SC_MG_PF 1      // stages of operand loads in flight (1 or 2), see k_modegemm_mfma
#endif

#ifndef SC_EMU
typedef float sc_f32x16 __attribute__((ext_vector_type(16)));
SC_DEVICE void sc_mfma_32x32x2(sc_f32x16& acc, const float a, const float b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}
SC_DEVICE float sc_xor_sign(const float v, const uint32_t mask) {
  return __uint_as_float(__float_as_uint(v) ^ mask);
}
// bitwise (mask ? hi : lo): one v_bfi_b32
SC_DEVICE float sc_bitsel(const uint32_t mask, const float hi, const float lo) {
  return __uint_as_float((__float_as_uint(hi) & mask) | (__float_as_uint(lo) & ~mask));
}
#else
struct sc_f32x16 {
  float v[16];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
namespace scemu {
inline float g_mfma_a[16][64];
inline float g_mfma_b[16][64];
}  // namespace scemu
// lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; it owns
// D[row = (v & 3) + 8 (v >> 2) + 4 (l >> 5)][col = l & 31], v = 0..15  (cdna_hip_programming.md 3)
inline void sc_mfma_32x32x2(sc_f32x16& acc, const float a, const float b) {
  const int w = SC_TID >> 6, l = SC_TID & 63;
  scemu::g_mfma_a[w][l] = a;
  scemu::g_mfma_b[w][l] = b;
  scemu::wave_barrier();
  for (int v = 0; v < 16; ++v) {
    const int row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5), col = l & 31;
    float c = acc[v];
    for (int k = 0; k < 2; ++k) c = fmaf(scemu::g_mfma_a[w][row + 32 * k], scemu::g_mfma_b[w][col + 32 * k], c);
    acc[v] = c;
  }
  scemu::wave_barrier();
}
inline float sc_bitsel(const uint32_t mask, const float hi, const float lo) {
  uint32_t a, b;
  std::memcpy(&a, &hi, 4);
  std::memcpy(&b, &lo, 4);
  a = (a & mask) | (b & ~mask);
  float r;
  std::memcpy(&r, &a, 4);
  return r;
}
inline float sc_xor_sign(const float v, const uint32_t mask) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  u ^= mask;
  float r;
  std::memcpy(&r, &u, 4);
  return r;
}
#endif

// keep an MFMA accumulator in the AGPR file across a loop back-edge (hipcc otherwise carries it in
// VGPRs and copies every register in and out of the AGPRs each iteration)
#ifndef SC_EMU
#define SC_PIN_ACC(x) asm volatile("" : "+a"(x))
#else
#define SC_PIN_ACC(x) do { } while (0)
#endif

struct MfmaGemmArgs {
  int P, Q, R, M, G;   // G = number of mode ranges = grid size
  int64_t a_sp, a_sr, a_sm;
  int64_t b_sr, b_sq, b_sm;
  int64_t c_sp, c_sq, c_sm;
  const int32_t* b_idx;
  const int32_t* c_idx;
  // (ablation of the kernel's phases is a COMPILE-TIME switch of measurement builds: -DSC_MG_ABLATE=bits with
  //  1 skip MFMA, 2 skip C stores, 4 skip operand loads; the product library is built without it)
  int stream_c;        // 1: C is not read by the next kernel -> non-temporal stores
};

template <int PT, int QG, int NM, int NWV = 8>
struct MfmaGemmCfg {
  static constexpr int P = 32 * PT, Q = 16 * QG;
  static constexpr int RC = SC_MG_RC;                         // r values per LDS stage
  static constexpr int NW = NWV;                              // waves per workgroup (8, or 4 with 2 workgroups per CU)
  static constexpr int THREADS = 64 * NW;
  static constexpr int NT = 4 * PT;                           // (row tile, column group set) pairs
  static constexpr int MS = NW / NT;                          // waves sharing a tile = r split
  static constexpr int HL = RC / NW;                          // r values staged by one wave
  static constexpr int NMS = (NM & 1) ? NM : NM + 1;          // odd mode stride (floats)
  static constexpr int QW = QG / 4;                           // column groups per wave
  static constexpr int A_FLOATS = RC * 2 * P * NMS;           // [rr][c][p][NMS]
  static constexpr int BPAD = (16 - (Q * NMS) % 32 + 32) % 32;
  static constexpr int BPS = Q * NMS + BPAD;                  // Im plane 16 banks away from Re
  static constexpr int B_FLOATS = RC * 2 * BPS;               // [rr][plane][q][NMS]
  static constexpr int STAGE = A_FLOATS + B_FLOATS;
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  static constexpr int SPI_MIN = 64 / NM;                     // segments per wave load at nm = NM
  static constexpr int NPA = (P + SPI_MIN - 1) / SPI_MIN;     // wave loads per r for A (max)
  static constexpr int NPB = (Q + SPI_MIN - 1) / SPI_MIN;     // ... for B
  static constexpr int EP_FLOATS = 8 * 16 * NMS * 2;          // epilogue patch of one tile: 8 rows
  static_assert(NT == 4 || NT == 8, "P = 32 or 64");
  static_assert(NW % NT == 0 && NW >= NT, "whole number of waves per tile");
  static_assert(HL * NW == RC, "the waves split the r values of a stage");
  static_assert(QG % 4 == 0, "4 waves split the column groups");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(NT * EP_FLOATS <= 2 * STAGE, "epilogue patches fit in the stage buffers");
};

template <int PT, int QG, int NM, bool CA, bool CB, int NWV = 8>
SC_GLOBAL void SC_LAUNCH_BOUNDS((MfmaGemmCfg<PT, QG, NM, NWV>::THREADS))
k_modegemm_mfma(MfmaGemmArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B,
                cf32* __restrict__ C) {
  typedef MfmaGemmCfg<PT, QG, NM, NWV> K;
  constexpr int P = K::P, Q = K::Q, RC = K::RC, NMS = K::NMS, QW = K::QW, BPS = K::BPS;
  SC_SHARED __attribute__((aligned(16))) float lds[2 * K::STAGE];

  // ---- which modes: contiguous range, neighbouring ranges on one XCD (block b runs on XCD b % 8)
  int gid = SC_BID_X;
  if ((g.G & 7) == 0) gid = (gid & 7) * (g.G >> 3) + (gid >> 3);
  const int m0 = (int)(((int64_t)gid * g.M) / g.G);
  const int nm = (int)(((int64_t)(gid + 1) * g.M) / g.G) - m0;       // 1 .. NM, block-uniform
  if (nm <= 0) return;

  const int tid = SC_TID, lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);

  // ---- staging role: lane = (segment sl, mode j) of a wave load; spi segments per load
  const int spi = 64 / nm;
  const int sl = lane / nm, jl = lane - sl * nm;
  const bool lvalid = sl < spi;
  const int mj = m0 + (lvalid ? jl : 0);
  // ragged problems (g.P <= P, g.Q <= Q: Tucker / TT ranks): a group of spi rows that would reach past the
  // last row is moved back to END at it (uniform, re-stages a few rows with the same values); LDS rows
  // >= g.P keep stale data that only ever reaches output rows nobody stores
  const int slp = lvalid ? (sl < g.P ? sl : g.P - 1) : 0, slq = lvalid ? (sl < g.Q ? sl : g.Q - 1) : 0;
  const int64_t la = (int64_t)slp * g.a_sp + (int64_t)mj * g.a_sm;
  const int64_t lb = (int64_t)slq * g.b_sq + (g.b_idx ? (int64_t)g.b_idx[mj] : (int64_t)mj * g.b_sm);
  const int lds_l = sl * NMS + jl;
  const int npa = (g.P + spi - 1) / spi, npb = (g.Q + spi - 1) / spi;   // uniform
  auto group_base = [&](const int gi, const int lim) {
    int b = gi * spi;
    if (b + spi > lim) b = lim - spi;
    return b < 0 ? 0 : b;
  };

  // operand registers of the stages in flight: SC_MG_PF = 2 keeps TWO stages' loads outstanding (stage k + 2 is
  // requested before stage k feeds the matrix cores), so a stage's HBM round trip -- about as long as a stage's
  // MFMAs at the metric shape -- is covered by two compute phases instead of one
  cf32 ra0[K::HL][K::NPA], rb0[K::HL][K::NPB];
#if SC_MG_PF == 2
  cf32 ra1[K::HL][K::NPA], rb1[K::HL][K::NPB];
#endif

  // every load address stays inside the operand (no exec-masked loads); r values past g.R are zeroed
  // when the stage is committed to LDS
  auto issue = [&](const int r0, cf32 (&ra)[K::HL][K::NPA], cf32 (&rb)[K::HL][K::NPB]) {
#pragma unroll
    for (int h = 0; h < K::HL; ++h) {
      int r = r0 + w + K::NW * h;                                       // uniform
      r = r < g.R ? r : g.R - 1;
      const cf32* ar = A + (int64_t)r * g.a_sr + la;
      const cf32* br = B + (int64_t)r * g.b_sr + lb;
      // no branches around the loads (a group past the last one re-reads the last rows): with loads in
      // conditional blocks the compiler cannot count the younger loads in flight and every wait for a
      // stage's registers becomes s_waitcnt vmcnt(0), i.e. also waits for the stage requested after it
#pragma unroll
      for (int pg = 0; pg < K::NPA; ++pg) ra[h][pg] = ar[(int64_t)group_base(pg, g.P) * g.a_sp];
#pragma unroll
      for (int qg = 0; qg < K::NPB; ++qg) rb[h][qg] = br[(int64_t)group_base(qg, g.Q) * g.b_sq];
    }
  };

  auto commit = [&](float* st, const int r0, const cf32 (&ra)[K::HL][K::NPA], const cf32 (&rb)[K::HL][K::NPB]) {
#pragma unroll
    for (int h = 0; h < K::HL; ++h) {
      const int rr = w + K::NW * h;
      const float keep = (r0 + rr < g.R) ? 1.f : 0.f;                  // rows past R contribute 0
#pragma unroll
      for (int pg = 0; pg < K::NPA; ++pg) {
        const int pb = group_base(pg, g.P);
        if (lvalid && pg < npa && pb + sl < P) {
          float* o = st + (rr * 2 * P + pb) * NMS + lds_l;
          o[0] = ra[h][pg].x * keep;
          o[P * NMS] = ra[h][pg].y * (CA ? -keep : keep);          // conj(A) folded in here
        }
      }
#pragma unroll
      for (int qg = 0; qg < K::NPB; ++qg) {
        const int qb = group_base(qg, g.Q);
        if (lvalid && qg < npb && qb + sl < Q) {
          float* o = st + K::A_FLOATS + rr * 2 * BPS + qb * NMS + lds_l;
          o[0] = rb[h][qg].x;
          o[BPS] = rb[h][qg].y;
        }
      }
    }
  };

  // ---- MFMA role: lane = (c, i) for A' rows, (c, n = 2 qq + d) for B' columns
  const int c = lane >> 5, i = lane & 31, d = lane & 1, qq = (lane & 31) >> 1;
  const uint32_t bmask = (CB ? (c == 0 && d == 1) : (c == 1 && d == 0)) ? 0x80000000u : 0u;
  const int tile = w & (K::NT - 1), kh = w / K::NT;       // tile = (row tile, first column group)
  const int wp = tile >> 2, wq = tile & 3;
  const int a_off = (c * P + wp * 32 + i) * NMS;
  int b_off[QW];
#pragma unroll
  for (int u = 0; u < QW; ++u) b_off[u] = K::A_FLOATS + (c ^ d) * BPS + ((wq + 4 * u) * 16 + qq) * NMS;

  sc_f32x16 acc[NM][QW];
#pragma unroll
  for (int j = 0; j < NM; ++j)
#pragma unroll
    for (int u = 0; u < QW; ++u)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[j][u][v] = 0.f;

  // one stage: this wave's r values (kh, kh + MS, ...) for all NM mode slots (slots >= nm hold
  // stale LDS and are never stored, so there are no branches).  The MFMAs of one r are issued
  // strictly back to back after the B' sign flips of the whole batch; the partner wave on the
  // SIMD covers the gaps.
  auto compute = [&](const float* st) {
#pragma unroll
    for (int rq = 0; rq < RC / K::MS; ++rq) {
      const float* sr = st + (rq * K::MS + kh) * 2 * P * NMS;           // A' rows of this r
      const float* sb = st + (rq * K::MS + kh) * 2 * BPS;               // B  rows of this r
      float av[NM], bs[NM][QW];
#pragma unroll
      for (int j = 0; j < NM; ++j) {
        av[j] = sr[a_off + j];
#pragma unroll
        for (int u = 0; u < QW; ++u) bs[j][u] = sb[b_off[u] + j];
      }
#pragma unroll
      for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int u = 0; u < QW; ++u) bs[j][u] = sc_xor_sign(bs[j][u], bmask);
      SC_SCHED_BARRIER();
#pragma unroll
      for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int u = 0; u < QW; ++u) sc_mfma_32x32x2(acc[j][u], av[j], bs[j][u]);
      SC_SCHED_BARRIER();
    }
  };

  const int nck = (g.R + RC - 1) / RC;
  constexpr int dbg = SC_MG_ABLATE;
  auto pin_acc = [&]() {
#ifndef SC_EMU
    if constexpr (K::NW == 4) {
      // 256-thread shape: hipcc puts the accumulators in AGPRs but carries them across the loop
      // back-edge in VGPRs (copying every register in and out each stage) unless they are pinned
#pragma unroll
      for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int u = 0; u < QW; ++u) asm volatile("" : "+a"(acc[j][u]));
    }
#endif
  };
  constexpr bool ld = !(dbg & 4);
  issue(0, ra0, rb0);
#if SC_MG_PF == 2
  if (nck > 1 && ld) issue(RC, ra1, rb1);
#endif
  commit(lds, 0, ra0, rb0);
  SC_SYNC();
#if SC_MG_PF == 2
  // two stages per trip so that the register sets alternate statically: even stages come from set 0 / LDS
  // buffer 0, odd ones from set 1 / buffer 1
  float* const l0 = lds;
  float* const l1 = lds + K::STAGE;
#pragma unroll 1
  for (int ck = 0; ck < nck; ck += 2) {
    if (ck + 2 < nck && ld) issue((ck + 2) * RC, ra0, rb0);
    if (!(dbg & 1)) compute(l0);
    pin_acc();
    if (ck + 1 < nck) commit(l1, (ck + 1) * RC, ra1, rb1);
    SC_SYNC();
    if (ck + 1 >= nck) break;
    if (ck + 3 < nck && ld) issue((ck + 3) * RC, ra1, rb1);
    if (!(dbg & 1)) compute(l1);
    pin_acc();
    if (ck + 2 < nck) commit(l0, (ck + 2) * RC, ra0, rb0);
    SC_SYNC();
  }
#else
#pragma unroll 1
  for (int ck = 0; ck < nck; ++ck) {
    float* cur = lds + (ck & 1) * K::STAGE;
    float* nxt = lds + ((ck + 1) & 1) * K::STAGE;
    const bool more = ck + 1 < nck;
    if (more && ld) issue((ck + 1) * RC, ra0, rb0);
    if (!(dbg & 1)) compute(cur);
    pin_acc();
    if (more) commit(nxt, (ck + 1) * RC, ra0, rb0);
    SC_SYNC();
  }
#endif

  // ---- C: a lane owns column n = (q, d) and 16 rows of its tile, i.e. 4-byte pieces M*8 bytes
  //      apart.  Eight rows at a time are transposed through the tile's LDS patch (the stage
  //      buffers are free after the last barrier) into (row, column, mode) order and leave as
  //      nm*8-byte segments -- the same access shape as the operand loads.  The waves sharing a
  //      tile fill the patch together and split the segment stores.
  if (dbg & 2) {
    if (acc[0][0][0] != 12345.678f) return;
  }
  constexpr int NST = (128 + K::SPI_MIN - 1) / K::SPI_MIN;
  float* ep = lds + tile * K::EP_FLOATS;
  const int nst = (128 + spi - 1) / spi;
  const int64_t lc = g.c_idx ? (int64_t)g.c_idx[mj] : (int64_t)mj * g.c_sm;
#pragma unroll
  for (int u = 0; u < QW; ++u) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (K::MS == 2) {                      // partner's partial sums first
        if (kh == 1) {
#pragma unroll
          for (int j = 0; j < NM; ++j)
            if (j < nm) {
#pragma unroll
              for (int vv = 0; vv < 4; ++vv)
                ep[(((4 * c + vv) * 16 + qq) * NMS + j) * 2 + d] = acc[j][u][4 * k + vv];
            }
        }
        SC_SYNC();
      }
      if (kh == 0) {
#pragma unroll
        for (int j = 0; j < NM; ++j)
          if (j < nm) {
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
              float* e = ep + (((4 * c + vv) * 16 + qq) * NMS + j) * 2 + d;
              *e = (K::MS == 2) ? acc[j][u][4 * k + vv] + *e : acc[j][u][4 * k + vv];
            }
          }
      }
      SC_SYNC();
      cf32* crow = C + (int64_t)(wp * 32 + 8 * k) * g.c_sp + (int64_t)((wq + 4 * u) * 16) * g.c_sq + lc;
#pragma unroll
      for (int it = 0; it < NST; ++it) {
        const int sg = it * spi + sl;                   // segment = (row 0..7, column 0..15)
        if ((it % K::MS) == kh && it < nst && lvalid && sg < 128 && wp * 32 + 8 * k + (sg >> 4) < g.P &&
            (wq + 4 * u) * 16 + (sg & 15) < g.Q) {
          const cf32 val = *reinterpret_cast<const cf32*>(ep + (sg * NMS + jl) * 2);
          cf32* cd = crow + (int64_t)(sg >> 4) * g.c_sp + (int64_t)(sg & 15) * g.c_sq;
          if (g.stream_c) {
            SC_STORE_STREAM(&cd->x, val.x);
            SC_STORE_STREAM(&cd->y, val.y);
          } else {
            *cd = val;
          }
        }
      }
      SC_SYNC();
    }
  }
}
