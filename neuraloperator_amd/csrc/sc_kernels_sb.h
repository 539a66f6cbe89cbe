// sc_kernels_sb.h -- the contraction when ONE of its three extents is tiny (round 3): a small batch against a large
// weight (BASELINE configs[4]: B = 4 rows, 128 x 128 channels, 33 024 modes -> 4.33 GB of weights per launch) or,
// in the weight gradient, a short reduction (R = B) into a weight-sized result.
//
//   C[p, q, m] = sum_r opA(A[p, r, m]) * opB(B[r, q, m])          (complex, mode stride 1 everywhere)
//
// replaces tl.einsum('bixy,ioxy->boxy') and its two autograd einsums (spectral_convolution.py:21-46) for these
// shapes.  They are pure weight streaming -- 4 flop per byte, nothing for the matrix cores -- and the kernels that
// served them moved the 4.33 GB at 2.8-3.8 TB/s (profiles/r02_bench_1024_settled.json: 1.14 + 1.39 + 1.55 ms of a
// 7.07 ms step): k_modegemm reads 8 bytes per lane (a wave = one 512-byte piece per operand row) with a single
// active wave per workgroup when P <= 4 and only what the compiler's unroll-by-2 leaves in flight; the streamed
// matrix-core kernel spends 32 / P of its work on clamped duplicate rows.  Here
//   * a lane owns TWO neighbouring modes: every access is 16 bytes per lane, a wave instruction moves 1 KiB of one
//     operand row; the four waves of a workgroup take four column tiles of the same 128 modes (or 4 KiB of contiguous
//     modes of one tile: see the wave arrangements below);
//   * register tile PT x QT per lane (PT >= the whole small extent when that is P), fp32 FMAs on the vector ALUs;
//   * the reduction loop keeps ST stages of BOTH operands in flight in registers (a rotating ring written out by
//     hand: the compiler's own unrolling drains the queue at the end of every unrolled body), non-temporal loads for
//     the operand that is streamed once (the weight), ordinary loads for the small one that every q tile re-reads
//     from L2 (work items are dealt to the XCDs so that all q tiles of one mode tile share an L2, as in k_modegemm);
//   * the weight gradient (R <= 8) is one pass of 16-byte stores; with `stream_c` they are non-temporal.
// Exact fp32: each output is an r-ordered chain of fmaf -- bit-identical to k_modegemm's result.
#pragma once
#include "sc_device.h"
#include "sc_kernels_generic.h"

struct SbGemmArgs {
  int64_t P, Q, R, M;
  int64_t a_sp, a_sr, b_sr, b_sq, c_sp, c_sq;     // complex elements; mode stride 1
  int n_mt, n_pt, n_qt, per_xcd;                  // mode tiles of 128 WM, row tiles of PT, column tiles of QT
  int mt_fastest;                                 // work-item order (see the kernel)
  int nt_a, nt_b, nt_c;                           // non-temporal access to A / B (read once) and C (not read next)
};

#ifndef SC_EMU
SC_DEVICE sc_f4 sb_load(const cf32* p, const int nt) {
  const sc_f4* q = reinterpret_cast<const sc_f4*>(p);
  return nt ? __builtin_nontemporal_load(q) : *q;
}
SC_DEVICE void sb_store(cf32* p, const sc_f4 v, const int nt) {
  sc_f4* q = reinterpret_cast<sc_f4*>(p);
  if (nt) __builtin_nontemporal_store(v, q);
  else *q = v;
}
#else
inline sc_f4 sb_load(const cf32* p, const int) {
  sc_f4 v;
  std::memcpy(&v, p, 16);
  return v;
}
inline void sb_store(cf32* p, const sc_f4 v, const int) { std::memcpy(p, &v, 16); }
#endif

// acc (two modes: (x, y) and (z, w)) += a * b
template <bool CA, bool CB>
SC_DEVICE void sb_mac(sc_f4& acc, const sc_f4 a, const sc_f4 b) {
  const float a0i = CA ? -a.y : a.y, a1i = CA ? -a.w : a.w;
  const float b0i = CB ? -b.y : b.y, b1i = CB ? -b.w : b.w;
  acc.x = fmaf(a.x, b.x, acc.x);
  acc.x = fmaf(-a0i, b0i, acc.x);
  acc.y = fmaf(a.x, b0i, acc.y);
  acc.y = fmaf(a0i, b.x, acc.y);
  acc.z = fmaf(a.z, b.z, acc.z);
  acc.z = fmaf(-a1i, b1i, acc.z);
  acc.w = fmaf(a.z, b1i, acc.w);
  acc.w = fmaf(a1i, b.z, acc.w);
}

// WM x WP x WQ = the four waves of a workgroup over (mode tiles of 128, row tiles, column tiles):
//   4 x 1 x 1  one (row, column) tile, 512 contiguous modes: 4 KiB of every operand row per workgroup instruction;
//   1 x 1 x 4  four neighbouring column tiles of ONE 128-mode tile: the small operand's slice is fetched once per
//              workgroup (the other three waves hit L1) and the slice all column tiles share is 4 x smaller, so it
//              stays in the XCD's L2 (4 MB) instead of being re-read from the Infinity Cache by every column tile;
//   1 x 2 x 2  the same for the weight gradient, where BOTH operands are small and re-read by 32 tiles each.
template <int PT, int QT, int ST, int WM, int WP, int WQ, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_modegemm_sb(SbGemmArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ C) {
  static_assert(WM * WP * WQ == 4, "four waves per workgroup");
  const int tid = SC_TID, lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  const int wm = w % WM, wq = (w / WM) % WQ, wp = w / (WM * WQ);
  // work item = (mode tile, row tile group, column tile group), mode tile slowest; consecutive items go to ONE XCD
  // (block b runs on XCD b % 8): all tiles of a mode tile re-read the same slices of the small operand(s) from that L2
  const int bid = SC_BID_X;
  const int64_t item = (int64_t)(bid & 7) * g.per_xcd + (bid >> 3);
  const int n_ptg = (g.n_pt + WP - 1) / WP, n_qtg = (g.n_qt + WQ - 1) / WQ;
  const int64_t per_mt = (int64_t)n_ptg * n_qtg;
  if (item >= (int64_t)g.n_mt * per_mt) return;
  // mt_fastest (round 5, weight-sized operands): the workgroups that run side by side on an XCD take NEIGHBOURING mode
  // tiles of the SAME rows and columns, so that at any moment the chip reads a few dozen rows of the weight in
  // contiguous runs of ~100 KB instead of 1 KB pieces of thousands of rows (pages) -- the small operand is then shared
  // through the Infinity Cache rather than an XCD's L2
  const int mt = g.mt_fastest ? (int)(item % g.n_mt) : (int)(item / per_mt);
  const int rem = g.mt_fastest ? (int)(item / g.n_mt) : (int)(item - (int64_t)mt * per_mt);
  const int qt = (rem / n_ptg) * WQ + wq, pt = (rem - (rem / n_ptg) * n_ptg) * WP + wp;
  if (pt >= g.n_pt || qt >= g.n_qt) return;                         // whole wave idle (no barriers in this kernel)
  const int64_t m = (((int64_t)mt * WM + wm) * 64 + lane) * 2;      // first of this lane's two modes
  const bool active = m < g.M;                                      // M is even: a pair is inside or outside
  const int64_t mm = active ? m : g.M - 2;
  const int64_t p0 = (int64_t)pt * PT, q0 = (int64_t)qt * QT;

  const cf32* Ap[PT];
  const cf32* Bq[QT];
#pragma unroll
  for (int pp = 0; pp < PT; ++pp) Ap[pp] = A + ((p0 + pp < g.P) ? (p0 + pp) : (g.P - 1)) * g.a_sp + mm;
#pragma unroll
  for (int qq = 0; qq < QT; ++qq) Bq[qq] = B + ((q0 + qq < g.Q) ? (q0 + qq) : (g.Q - 1)) * g.b_sq + mm;

  sc_f4 acc[PT][QT];
#pragma unroll
  for (int pp = 0; pp < PT; ++pp)
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) acc[pp][qq] = sc_f4{0.f, 0.f, 0.f, 0.f};

  // ring of ST stages: stage s holds the operands of reduction step r with r % ST == s.  NO uniform branch may
  // surround a load in the steady state: behind one the compiler cannot count the loads in flight and waits with
  // vmcnt(0), i.e. for the refill it has just issued (the first version did exactly that) -- steps past the end are
  // clamped re-reads of the last one (requested, never used), whole blocks of ST steps run unguarded, the
  // R % ST remaining steps use what the ring already holds.
  sc_f4 ra[ST][PT], rb[ST][QT];
#if defined(SC_SB_ABL_NOA) || defined(SC_SB_ABL_NOB)
  for (int s_ = 0; s_ < ST; ++s_) {
    for (int pp = 0; pp < PT; ++pp) ra[s_][pp] = sc_f4{1.f, (float)lane, 2.f, 3.f};
    for (int qq = 0; qq < QT; ++qq) rb[s_][qq] = sc_f4{(float)lane, 1.f, 0.5f, 2.f};
  }
#endif
  auto request = [&](const int64_t r, sc_f4 (&a)[PT], sc_f4 (&b)[QT]) {
    const int64_t rr = r < g.R ? r : g.R - 1;
#pragma unroll
    for (int pp = 0; pp < PT; ++pp) {
#ifdef SC_SB_ABL_NOA                                       // measurement builds only: one operand's loads removed
      if (r == -12345) a[pp] = sb_load(Ap[pp] + rr * g.a_sr, g.nt_a);
#else
      a[pp] = sb_load(Ap[pp] + rr * g.a_sr, g.nt_a);
#endif
    }
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
#ifdef SC_SB_ABL_NOB
      if (r == -12345) b[qq] = sb_load(Bq[qq] + rr * g.b_sr, g.nt_b);
#else
      b[qq] = sb_load(Bq[qq] + rr * g.b_sr, g.nt_b);
#endif
    }
  };
  auto multiply = [&](const sc_f4 (&a)[PT], const sc_f4 (&b)[QT]) {
#pragma unroll
    for (int pp = 0; pp < PT; ++pp)
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) sb_mac<CA, CB>(acc[pp][qq], a[pp], b[qq]);
  };
#pragma unroll
  for (int s = 0; s < ST; ++s) request(s, ra[s], rb[s]);
  const int64_t RB = (g.R / ST) * ST;
#pragma unroll 1
  for (int64_t r0 = 0; r0 < RB; r0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      multiply(ra[s], rb[s]);
      request(r0 + s + ST, ra[s], rb[s]);                             // refill the slot just consumed
    }
  }
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (RB + s < g.R) multiply(ra[s], rb[s]);                         // the R % ST last steps (no loads behind the branch)
  if (!active) return;
#pragma unroll
  for (int pp = 0; pp < PT; ++pp) {
    if (p0 + pp >= g.P) continue;
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
      if (q0 + qq >= g.Q) continue;
      sb_store(C + (p0 + pp) * g.c_sp + (q0 + qq) * g.c_sq + m, acc[pp][qq], g.nt_c);
    }
  }
}


// ------------------------------------------------------------------------------------------
// The two contractions of a SMALL-BATCH backward pass in ONE pass over the weight (round 3, session 2):
//   gW[i, o, m]    = sum_b conj(xhat[b, i, m]) ghat[b, o, m]          (weight-sized result)
//   gxhat[b, i, m] = sum_o ghat[b, o, m] conj(W[i, o, m])             (weight-sized operand)
// BASELINE configs[4] (B = 4, 128 x 128 channels, 33 024 modes: W and gW are 4.33 GB each): as two launches of
// k_modegemm_sb the pair takes 1.20 + 1.39 ms -- per reduction step the spectrum gradient loads 4 weight + 4 ghat
// values per lane, the weight gradient 32 operand values per 16 stored ones, all of the small operands re-read from
// L2 by every tile.  Here a wave owns IT weight rows i (and 128 modes, two per lane), keeps xhat[b, i, .] and the
// gxhat accumulators in registers and walks o ONCE: per step it loads W[i, o, .] (IT values, streamed) and
// ghat[b, o, .] (BT values, shared by the four waves of the workgroup through L1 and by the workgroups of a mode
// tile through their XCD's L2), adds ghat conj(W) into the accumulators and stores gW[i, o, .] = sum_b conj(xhat)
// ghat -- 8 loads and 4 stores where the two launches issue 16 loads and 4 stores, and the weight is read while its
// gradient is written (a copy-shaped stream).  Same fmaf chains as k_modegemm_sb (o-ordered / b-ordered): both
// results are bit-identical to the two launches.
// ------------------------------------------------------------------------------------------
struct SbBwdArgs {
  int64_t B, Ci, Co, M;
  int64_t x_sb, x_si;          // xhat[b, i, m]   (complex elements; mode stride 1 everywhere)
  int64_t g_sb, g_so;          // ghat[b, o, m]
  int64_t w_si, w_so;          // W[i, o, m]
  int64_t gw_si, gw_so;        // gW[i, o, m]
  int64_t gx_sb, gx_si;        // gxhat[b, i, m]
  int n_mt, n_itg, per_xcd;    // mode tiles of 128, groups of 4 row tiles
  int mt_fastest;              // work-item order (k_modegemm_sb)
  int nt_gw;                   // non-temporal stores for gW (not read by the next kernel)
};

template <int BT, int IT, int ST>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_modegemm_sb_bwd(SbBwdArgs g, const cf32* __restrict__ xhat, const cf32* __restrict__ ghat, const cf32* __restrict__ W,
                  cf32* __restrict__ gW, cf32* __restrict__ gxhat) {
  const int tid = SC_TID, lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  const int bid = SC_BID_X;
  const int64_t item = (int64_t)(bid & 7) * g.per_xcd + (bid >> 3);
  if (item >= (int64_t)g.n_mt * g.n_itg) return;
  const int mt = g.mt_fastest ? (int)(item % g.n_mt) : (int)(item / g.n_itg);
  const int64_t i0 = ((g.mt_fastest ? item / g.n_mt : item - (int64_t)mt * g.n_itg) * 4 + w) * IT;
  if (i0 >= g.Ci) return;                                              // whole wave idle (no barriers in this kernel)
  const int64_t m = ((int64_t)mt * 64 + lane) * 2;                     // first of this lane's two modes
  const bool active = m < g.M;
  const int64_t mm = active ? m : g.M - 2;

  // rows / modes past the end are CLAMPED, loads and stores alike: such a lane recomputes -- from the same inputs --
  // and stores again exactly what the owner of the clamped element stores, so every access of the loop is
  // unconditional (behind a branch the compiler could no longer count the ring's loads in flight)
  const cf32* Wp[IT];
  cf32* GWp[IT];
  const cf32* Gp[BT];
  sc_f4 xh[BT][IT], acc[BT][IT];
#pragma unroll
  for (int ii = 0; ii < IT; ++ii) {
    const int64_t i = (i0 + ii < g.Ci) ? i0 + ii : g.Ci - 1;
    Wp[ii] = W + i * g.w_si + mm;
    GWp[ii] = gW + i * g.gw_si + mm;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      xh[b][ii] = sb_load(xhat + b * g.x_sb + i * g.x_si + mm, 0);
      acc[b][ii] = sc_f4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int b = 0; b < BT; ++b) Gp[b] = ghat + b * g.g_sb + mm;

  // ring of ST steps over o (see k_modegemm_sb: no uniform branch around a load; steps past the end re-read the last)
  sc_f4 rw[ST][IT], rg[ST][BT];
  auto request = [&](const int64_t o, sc_f4 (&wv)[IT], sc_f4 (&gv)[BT]) {
    const int64_t oo = o < g.Co ? o : g.Co - 1;
#pragma unroll
    for (int ii = 0; ii < IT; ++ii) wv[ii] = sb_load(Wp[ii] + oo * g.w_so, 1);
#pragma unroll
    for (int b = 0; b < BT; ++b) gv[b] = sb_load(Gp[b] + oo * g.g_so, 0);
  };
  auto step = [&](const int64_t o, const sc_f4 (&wv)[IT], const sc_f4 (&gv)[BT]) {
#pragma unroll
    for (int ii = 0; ii < IT; ++ii) {
      sc_f4 gw = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < BT; ++b) sb_mac<true, false>(gw, xh[b][ii], gv[b]);       // conj(xhat) ghat, b-ordered
      sb_store(GWp[ii] + o * g.gw_so, gw, g.nt_gw);
#pragma unroll
      for (int b = 0; b < BT; ++b) sb_mac<false, true>(acc[b][ii], gv[b], wv[ii]);  // ghat conj(W), o-ordered
    }
  };
#pragma unroll
  for (int s = 0; s < ST; ++s) request(s, rw[s], rg[s]);
  const int64_t OB = (g.Co / ST) * ST;
#pragma unroll 1
  for (int64_t o0 = 0; o0 < OB; o0 += ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      step(o0 + s, rw[s], rg[s]);
      request(o0 + s + ST, rw[s], rg[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (OB + s < g.Co) step(OB + s, rw[s], rg[s]);
  if (!active) return;
#pragma unroll
  for (int ii = 0; ii < IT; ++ii) {
    if (i0 + ii >= g.Ci) continue;
#pragma unroll
    for (int b = 0; b < BT; ++b) sb_store(gxhat + b * g.gx_sb + (i0 + ii) * g.gx_si + m, acc[b][ii], 0);
  }
}


// ------------------------------------------------------------------------------------------
// Contraction with a MODE-INDEPENDENT right operand (round 3): C[p, q, m] = sum_r opA(A[p, r, m]) * opB(B[r, q])
// -- the channel-factor steps of the factorized contractions (spectral_convolution.py:55-103: z = xhat U_in,
// yhat = t U_out^T and their adjoints in the backward pass; CP's factor products).  B is a small matrix (64 x 36
// complex at TFNO rank 0.1) that every mode shares, so it never has to be a per-lane operand: lanes = modes, the wave
// reads B[r, q] through the SCALAR cache (wave-uniform address -> s_load) and uses it as the scalar source of the FMAs;
// a lane holds QC complex accumulators and streams its A values once.  The four waves of a workgroup take four
// neighbouring chunks of QC columns for the same (row, mode tile): the A slice is fetched once per workgroup, the
// other waves hit L1.  The register-staged matrix-core kernel that served these calls spends its time on 64-column
// tiles for a rank of 36 and on operand staging: 41 us per call at TFNO rank 0.1 for 1.25 GFLOP and 54 MB
// (profiles/r03_tfno_kernel_stats.txt).  Exact fp32, r-ordered fmaf chains: bit-identical to k_modegemm.
// ------------------------------------------------------------------------------------------
struct BfacGemmArgs {
  int64_t P, Q, R, M;
  int64_t a_sp, a_sr, b_sr, b_sq, c_sp, c_sq;
  int n_mt, n_qg;                                 // mode tiles of 64, groups of 4 QC columns
};

template <int QC, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_modegemm_bfac(BfacGemmArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ C) {
  // the whole factor matrix, conjugated if asked, as lds[r][q] (q contiguous whatever B's own layout is): every wave
  // then reads its QC columns of row r with broadcast 16-byte LDS reads.  (First versions read B through the scalar
  // cache: 30-33 us per call for the [r][q] layout, 42-45 us for the transposed one of the backward pass -- scalar
  // loads return out of order, so every step waited for all of them; a deeper ring of A loads made it worse:
  // profiles/r03_tfno_kernel_stats.txt.)
  SC_DYN_SHARED(cf32, lds);
  const int tid = SC_TID, lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  const int Qp = (int)((g.Q + 1) & ~(int64_t)1);                    // row stride: even, so that rows stay 16-byte aligned
  for (int i = tid; i < (int)(g.R * g.Q); i += SC_BLOCK) {
    const int r = i / (int)g.Q, q = i - r * (int)g.Q;
    cf32 b = B[r * g.b_sr + q * g.b_sq];
    if (CB) b.y = -b.y;
    lds[r * Qp + q] = b;
  }
  SC_SYNC();
  // item = (row p, mode tile, column group): mode tile fastest, so that neighbouring workgroups stream neighbouring
  // 512-byte pieces of the same A rows
  const int64_t item = SC_BID_X;
  const int mt = (int)(item % g.n_mt);
  const int64_t rest = item / g.n_mt;
  const int qg = (int)(rest % g.n_qg);
  const int64_t p = rest / g.n_qg;
  const int q0 = (qg * 4 + w) * QC;                                 // wave-uniform
  if (q0 >= g.Q) return;                                            // whole wave idle (no barrier below)
  const int64_t m = (int64_t)mt * 64 + lane;
  const bool active = m < g.M;
  const cf32* Ap = A + p * g.a_sp + (active ? m : g.M - 1);
  cf32 acc[QC];
#pragma unroll
  for (int j = 0; j < QC; ++j) acc[j] = cf_make(0.f, 0.f);
  const bool whole = q0 + QC <= g.Q && (q0 & 1) == 0;               // wave-uniform: aligned 16-byte reads of a full chunk
#pragma unroll 4
  for (int64_t r = 0; r < g.R; ++r) {
    cf32 a = Ap[r * g.a_sr];
    if (CA) a.y = -a.y;
    const cf32* br = lds + r * Qp + q0;
    cf32 b[QC];
    if (whole) {
#pragma unroll
      for (int j = 0; j + 1 < QC; j += 2) sc_lds_ld128(br + j, b[j], b[j + 1]);
      if (QC & 1) b[QC - 1] = sc_lds_ld64(br + QC - 1);
    } else {
#pragma unroll
      for (int j = 0; j < QC; ++j) b[j] = br[q0 + j < g.Q ? j : (int)g.Q - 1 - q0];
    }
#pragma unroll
    for (int j = 0; j < QC; ++j) cf_mac(acc[j], a, b[j]);
  }
  if (!active) return;
#pragma unroll
  for (int j = 0; j < QC; ++j)
    if (q0 + j < g.Q) C[p * g.c_sp + (q0 + j) * g.c_sq + m] = acc[j];
}
