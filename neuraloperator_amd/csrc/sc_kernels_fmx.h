// sc_kernels_fmx.h -- contractions with a small mode-INDEPENDENT matrix (Tucker / CP factor matrices, GaLore
// projections) and mode-summed contractions (their gradients) on the matrix cores (round 3).
//
//   k_modegemm_bfac_mx:  C[p, q, m] = sum_r opA(A[p, r, m]) opB(B[r, q])                    R, Q <= 64
//   k_modegemm_msum_mx:  C[p, q]    = sum_m sum_r opA(A[p, r, m]) opB(B[r, q, m])           P, Q <= 64
//
// (the factor steps of _contract_tucker / _contract_cp, spectral_convolution.py:55-103, and the factor gradients of
// their autograd).  The lanes-are-modes VALU kernels (k_modegemm_bfac, k_modegemm_msum) sit at 32-39 / 50 us for the
// 54 MB of TFNO rank 0.1 at the metric shape: a wave keeps only a handful of 512-byte loads in flight.  Here a
// workgroup stages a [rows][64 modes] chunk of the mode-dependent operand(s) in LDS with whole-row loads (the next
// chunk is in flight in registers meanwhile) and multiplies out of LDS with the 16 x 16 x 4 tile routine of
// sc_kernels_tucker.h (three real products, zero-padded k extents, conjugations as signs of the combination).
//   bfac_mx: chunk = (p, 64 modes): out[q][m] = sum_r Bt[q][r] A[r][m], Bt = the factor, LDS-resident for the launch.
//   msum_mx: chunk = (r, 64 modes): acc[p][q] += sum_m A[p][m] B[q][m] in the MFMA accumulators over all chunks of
//            the workgroup, one partial per workgroup, fixed-order reduction (k_fmx_reduce).
#pragma once
#include "sc_kernels_tucker.h"

// LDS row stride of a 64-mode chunk (8-byte units, see tkm_layout): 2 x odd.  The chunk of bfac_mx is read with k down
// the rows, for which 80 (16 mod 32) would be conflict free -- but 41 instead of 34 KB per chunk drops the kernel from
// three to two workgroups per CU, and bank conflicts are 3 % of its LDS cycles (profiles/r03_fmx_pmc.txt)
#define SC_FMX_LDK 66
#define SC_FMX_LDR 66

struct FmxArgs {
  int64_t P, Q, R, M;
  int64_t a_sp, a_sr, b_sr, b_sq, c_sp, c_sq;
  int n_mb;                         // mode blocks of 64
  int n_chunks, n_wg;
  int ldb;                          // bfac_mx: row stride of the factor table [Q4][ldb]
  uint32_t inv_q;                   // bfac_mx: ceil(2^32 / Q)
  int abl;                          // measurement only (SC_TK_ABL): 1 = no k loops, 2 = no result stores
};

// PF: prefetch registers per thread (ceil(R / 4)); TQ: 16-blocks of q a wave multiplies at once (>= ceil(Q / 16))
template <int PF, int TQ, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_modegemm_bfac_mx(FmxArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ C) {
  SC_DYN_SHARED(cf32, lds);
  const int Q = (int)g.Q, R = (int)g.R;
  const int q4 = (Q + 3) & ~3, r4 = (R + 3) & ~3;
  cf32* bt = lds;                                  // [q4][ldb]: bt[q][r] = B[r, q]
  cf32* ac = lds + q4 * g.ldb;                     // [r4][66]: the chunk, ac[r][m]
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  cf32 pf[PF];
  auto fetch = [&](const int c) {
    const int p = c / g.n_mb, mb = c - p * g.n_mb;
    const int64_t m = (int64_t)mb * 64 + lane;
    const cf32* src = A + (int64_t)p * g.a_sp + m;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int r = w + 4 * k;
      pf[k] = (r < R && m < g.M) ? src[(int64_t)r * g.a_sr] : cf_make(0.f, 0.f);
    }
  };
  // the first chunk and the factor are requested before anything else (all loads of a thread in flight together)
  if ((int)SC_BID_X < g.n_chunks) fetch(SC_BID_X);
  cf32 tb[16];                                     // Q R <= 4096 entries over 256 threads
  int to[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = tid + 256 * k;
    const int r = (int)(((uint64_t)(uint32_t)i * g.inv_q) >> 32), q = i - r * Q;       // i / Q, i % Q (i < 2^16)
    to[k] = q * g.ldb + r;
    if (i < Q * R) tb[k] = B[(int64_t)r * g.b_sr + (int64_t)q * g.b_sq];
  }
  for (int i = tid; i < q4 * g.ldb + r4 * SC_FMX_LDK; i += 256) lds[i] = cf_make(0.f, 0.f);
  SC_SYNC();
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (tid + 256 * k < Q * R) bt[to[k]] = tb[k];
  for (int c = SC_BID_X; c < g.n_chunks; c += g.n_wg) {
    SC_SYNC();                                     // the factor table (first round) / readers of the previous chunk
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int r = w + 4 * k;
      if (r < r4) ac[r * SC_FMX_LDK + lane] = pf[k];
    }
    SC_SYNC();
    if (c + g.n_wg < g.n_chunks) fetch(c + g.n_wg);
    const int p = c / g.n_mb, mb = c - p * g.n_mb;
    const int64_t left = g.M - (int64_t)mb * 64;
    const int nm = left < 64 ? (int)left : 64;
    cf32* dst = C + (int64_t)p * g.c_sp + (int64_t)mb * 64;
    // wave w: the modes 16 w .. of the chunk (the shared operand of a k step) against all TQ blocks of q
    if (16 * w < nm) {
      TkAcc a[TQ];
#pragma unroll
      for (int t = 0; t < TQ; ++t) tk_zero(a[t]);
      tk_multi<TQ, false, CB, CA>(bt, g.ldb, 1, ac, SC_FMX_LDK, 1, 0, 16 * w, Q, nm, R, lane, a, g.abl);
      // rows of the result are q (stride c_sq), columns the modes
      const int j = 16 * w + (lane & 15), ib = 4 * (lane >> 4);
      if (j < nm && !(g.abl & 2)) {
#pragma unroll
        for (int t = 0; t < TQ; ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (16 * t + ib + v < Q) dst[(int64_t)(16 * t + ib + v) * g.c_sq + j] = tk_result<CA != CB>(a[t], v);
      }
    }
  }
}

// PFA / PFB: prefetch registers per thread (ceil(P / 4), ceil(Q / 4)); SLOTS: 16-blocks of q (>= ceil(Q / 16))
template <int PFA, int PFB, int SLOTS, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_modegemm_msum_mx(FmxArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ partial) {
  SC_DYN_SHARED(cf32, lds);
  const int P = (int)g.P, Q = (int)g.Q;
  const int p4 = (P + 3) & ~3, q4 = (Q + 3) & ~3;
  cf32* ac = lds;                                  // [p4][66]: ac[p][m]
  cf32* bc = lds + p4 * SC_FMX_LDR;                // [q4][66]: bc[q][m]
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  cf32 pfa[PFA], pfb[PFB];
  auto fetch = [&](const int c) {
    const int r = c / g.n_mb, mb = c - r * g.n_mb;
    const int64_t m = (int64_t)mb * 64 + lane;
    const cf32* sa = A + (int64_t)r * g.a_sr + m;
    const cf32* sb = B + (int64_t)r * g.b_sr + m;
#pragma unroll
    for (int k = 0; k < PFA; ++k) {
      const int p = w + 4 * k;
      pfa[k] = (p < P && m < g.M) ? sa[(int64_t)p * g.a_sp] : cf_make(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < PFB; ++k) {
      const int q = w + 4 * k;
      pfb[k] = (q < Q && m < g.M) ? sb[(int64_t)q * g.b_sq] : cf_make(0.f, 0.f);
    }
  };
  if ((int)SC_BID_X < g.n_chunks) fetch(SC_BID_X);
  for (int i = tid; i < (p4 + q4) * SC_FMX_LDR; i += 256) lds[i] = cf_make(0.f, 0.f);
  TkAcc acc[SLOTS];                                // wave w: rows p = 16 w .. against all SLOTS blocks of q
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) tk_zero(acc[k]);
  const int tp = (P + 15) >> 4;
  for (int c = SC_BID_X; c < g.n_chunks; c += g.n_wg) {
    SC_SYNC();                                     // the zero fill (first round) / readers of the previous chunk
#pragma unroll
    for (int k = 0; k < PFA; ++k) {
      const int p = w + 4 * k;
      if (p < p4) ac[p * SC_FMX_LDR + lane] = pfa[k];
    }
#pragma unroll
    for (int k = 0; k < PFB; ++k) {
      const int q = w + 4 * k;
      if (q < q4) bc[q * SC_FMX_LDR + lane] = pfb[k];
    }
    SC_SYNC();
    if (c + g.n_wg < g.n_chunks) fetch(c + g.n_wg);
    if (w < tp) tk_multi<SLOTS, true, CA, CB>(ac, SC_FMX_LDR, 1, bc, 1, SC_FMX_LDR, 16 * w, 0, P, Q, 64, lane, acc, g.abl);
  }
  cf32* dst = partial + (int64_t)SC_BID_X * P * Q;
  if (w < tp) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) tk_store<CA != CB>(acc[k], dst, Q, 16 * w, 16 * k, P, Q, lane);
  }
}

// C[p, q] = sum_k partial[k][p Q + q] (fixed order); a block takes 16 entries x RG row groups (RG x 16 threads).
// Round 5: 64 row groups (1024 threads) instead of 16 -- the launch is a latency chain (each thread's n / RG loads, then
// the column's RG partial sums added in order by one thread), not bandwidth: 13.9 -> 11.2 us at TFNO rank 0.1 (partials
// of 64 x 36).  The order of the additions is fixed by (k mod RG, then the row groups ascending): reproducible.
#define SC_FMX_RED_RG 64
SC_GLOBAL void SC_LAUNCH_BOUNDS(16 * SC_FMX_RED_RG)
k_fmx_reduce(const cf32* __restrict__ partial, int n, int npc, int Q, cf32* __restrict__ C, int64_t c_sp, int64_t c_sq) {
  constexpr int RG = SC_FMX_RED_RG;
  SC_SHARED cf32 red[RG][17];
  const int tid = SC_TID, cx = tid & 15, rg = tid >> 4;
  const int col = SC_BID_X * 16 + cx;
  cf32 acc = cf_make(0.f, 0.f);
  if (col < npc) {
#pragma unroll 4
    for (int k = rg; k < n; k += RG) {
      const cf32 v = partial[(int64_t)k * npc + col];
      acc.x += v.x;
      acc.y += v.y;
    }
  }
  red[rg][cx] = acc;
  SC_SYNC();
  if (rg == 0 && col < npc) {
    cf32 t = red[0][cx];
    for (int r = 1; r < RG; ++r) {
      t.x += red[r][cx].x;
      t.y += red[r][cx].y;
    }
    const int p = col / Q, q = col - p * Q;
    C[(int64_t)p * c_sp + (int64_t)q * c_sq] = t;
  }
}
