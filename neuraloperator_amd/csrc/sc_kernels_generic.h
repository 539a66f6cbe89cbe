// sc_kernels_generic.h -- size-agnostic kernels of the SpectralConv engine.
//
// These handle ANY grid size / dimensionality (odd sizes, 1-D .. 4-D, kept-mode counts of
// any parity) by evaluating each 1-D pruned / zero-padded DFT directly from a twiddle
// table: only the kept modes are ever computed or stored, so the full half-spectrum of the
// reference (spectral_convolution.py:443-462, 531-559) never exists.  Power-of-two grids
// take the FFT kernels in sc_kernels_fft.h instead; both produce the same layout.
//
// Common design (CDNA4): lanes run over independent lines, the transform axis is a serial
// loop per lane, and the twiddle for (n, j) is wave-uniform -> it is fetched with scalar
// loads (s_load_dwordxN from the [n][j]-major table) and fed to v_fma_f32 as an SGPR
// operand.  No cross-lane traffic; LDS is used only to turn the row-major last axis into
// lane-per-line order and back so that every global access is coalesced.
#pragma once
#include "sc_device.h"

#define SC_BLOCK 256
#define SC_WAVE 64
#define SC_LINES_PER_BLOCK 64

// ------------------------------------------------------------------------------------------
// Pass over a non-last axis:  out[o, j, i] = sum_n T[n][j] * in[o, n, i]
//   in : complex [outer, N, inner]      out: complex [outer, J, inner]
//   T  : complex [N][Jpad] (Jpad multiple of JT, zero padded)
// forward:  N = n_d, J = k_d, T[n][j] = exp(-2 pi i f_j n / n_d)       (prunes)
// inverse:  N = k_d, J = n_d, T[n][j] = exp(+2 pi i f_n j / n_d)       (zero-pads)
// lanes = flattened (o, i); grid.y = j tiles.
// ------------------------------------------------------------------------------------------
template <int JT>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_axis_pass(const cf32* __restrict__ in, cf32* __restrict__ out, const cf32* __restrict__ table,
            int64_t outer, int N, int J, int64_t inner, int Jpad) {
  const int64_t col = (int64_t)SC_BID_X * SC_BLOCK + SC_TID;
  const int j0 = SC_BID_Y * JT;
  const bool active = col < outer * inner;
  const int64_t c = active ? col : 0;
  const int64_t o = c / inner, i = c - o * inner;
  const cf32* src = in + (o * N) * inner + i;
  cf32 acc[JT];
#pragma unroll
  for (int jj = 0; jj < JT; ++jj) acc[jj] = cf_make(0.f, 0.f);
#pragma unroll 2
  for (int n = 0; n < N; ++n) {
    const cf32 v = src[(int64_t)n * inner];
    const cf32* t = table + (int64_t)n * Jpad + j0;
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) cf_mac(acc[jj], t[jj], v);
  }
  if (active) {
    cf32* dst = out + (o * J) * inner + i;
#pragma unroll
    for (int jj = 0; jj < JT; ++jj)
      if (j0 + jj < J) dst[(int64_t)(j0 + jj) * inner] = acc[jj];
  }
}

// ------------------------------------------------------------------------------------------
// Last axis, real -> complex, pruned:  out[l, j] = sum_n in[l, n] * T[n][j]
//   in: real [lines, N]    out: complex [lines, J]    T: complex [N][Jpad]
// A block owns 64 lines; its 4 waves own 4 adjacent j-tiles of JT columns; grid.y walks
// groups of 4*JT columns.  The input tile goes through LDS (coalesced load, lane-per-line
// read, row stride 65 -> conflict free); results return through LDS for coalesced stores.
// ------------------------------------------------------------------------------------------
#define SC_R2C_NC 64
template <int JT>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_last_r2c(const float* __restrict__ in, cf32* __restrict__ out, const cf32* __restrict__ table,
           int64_t lines, int N, int J, int Jpad) {
  SC_SHARED float tile[SC_LINES_PER_BLOCK][SC_R2C_NC + 1];
  SC_SHARED cf32 otile[SC_LINES_PER_BLOCK][4 * JT + 1];
  const int tid = SC_TID;
  const int lane = tid & 63, w = tid >> 6;
  const int64_t l0 = (int64_t)SC_BID_X * SC_LINES_PER_BLOCK;
  const int jg0 = SC_BID_Y * 4 * JT;
  const int j0 = jg0 + w * JT;
  cf32 acc[JT];
#pragma unroll
  for (int jj = 0; jj < JT; ++jj) acc[jj] = cf_make(0.f, 0.f);

  for (int n0 = 0; n0 < N; n0 += SC_R2C_NC) {
    for (int idx = tid; idx < SC_LINES_PER_BLOCK * SC_R2C_NC; idx += SC_BLOCK) {
      const int l = idx / SC_R2C_NC, n = idx - l * SC_R2C_NC;
      const int64_t gl = l0 + l;
      const int gn = n0 + n;
      tile[l][n] = (gl < lines && gn < N) ? in[gl * N + gn] : 0.f;
    }
    SC_SYNC();
    const int nmax = (N - n0 < SC_R2C_NC) ? (N - n0) : SC_R2C_NC;
    if (j0 < J) {
      for (int n = 0; n < nmax; ++n) {
        const float v = tile[lane][n];
        const cf32* t = table + (int64_t)(n0 + n) * Jpad + j0;
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) {
          acc[jj].x = fmaf(v, t[jj].x, acc[jj].x);
          acc[jj].y = fmaf(v, t[jj].y, acc[jj].y);
        }
      }
    }
    SC_SYNC();
  }
#pragma unroll
  for (int jj = 0; jj < JT; ++jj) otile[lane][w * JT + jj] = acc[jj];
  SC_SYNC();
  const int ncols = (J - jg0 < 4 * JT) ? (J - jg0) : 4 * JT;
  for (int idx = tid; idx < SC_LINES_PER_BLOCK * ncols; idx += SC_BLOCK) {
    const int l = idx / ncols, c = idx - l * ncols;
    if (l0 + l < lines) out[(l0 + l) * J + jg0 + c] = otile[l][c];
  }
}

// ------------------------------------------------------------------------------------------
// Last axis, complex -> real, zero-padded:
//   out[l, n] = sum_j ( in[l,j].re * T[j][n].re - in[l,j].im * T[j][n].im ) + bias[ch(l)]
//   in: complex [lines, J]   out: real [lines, N]   T: complex [J][Npad] = w_j (cos, sin)
// w_j carries the C2R column weight (1 for DC / Nyquist, 2 inside) and the norm, so the
// imaginary parts of the DC and Nyquist columns drop out exactly as in irfft
// (spectral_convolution.py:552-559).  ch(l) = (l / lines_per_image) % channels.
// ------------------------------------------------------------------------------------------
#define SC_C2R_JC 32
template <int NT>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_last_c2r(const cf32* __restrict__ in, float* __restrict__ out, const cf32* __restrict__ table,
           const float* __restrict__ bias, int64_t lines, int N, int J, int Npad,
           int64_t lines_per_image, int64_t channels) {
  SC_SHARED cf32 itile[SC_LINES_PER_BLOCK][SC_C2R_JC + 1];
  SC_SHARED float otile[SC_LINES_PER_BLOCK][4 * NT + 1];
  const int tid = SC_TID;
  const int lane = tid & 63, w = tid >> 6;
  const int64_t l0 = (int64_t)SC_BID_X * SC_LINES_PER_BLOCK;
  const int ng0 = SC_BID_Y * 4 * NT;
  const int n0 = ng0 + w * NT;
  float acc[NT];
#pragma unroll
  for (int nn = 0; nn < NT; ++nn) acc[nn] = 0.f;

  for (int j0 = 0; j0 < J; j0 += SC_C2R_JC) {
    for (int idx = tid; idx < SC_LINES_PER_BLOCK * SC_C2R_JC; idx += SC_BLOCK) {
      const int l = idx / SC_C2R_JC, j = idx - l * SC_C2R_JC;
      const int64_t gl = l0 + l;
      const int gj = j0 + j;
      itile[l][j] = (gl < lines && gj < J) ? in[gl * J + gj] : cf_make(0.f, 0.f);
    }
    SC_SYNC();
    const int jmax = (J - j0 < SC_C2R_JC) ? (J - j0) : SC_C2R_JC;
    if (n0 < N) {
      for (int j = 0; j < jmax; ++j) {
        const cf32 v = itile[lane][j];
        const cf32* t = table + (int64_t)(j0 + j) * Npad + n0;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
          acc[nn] = fmaf(v.x, t[nn].x, acc[nn]);
          acc[nn] = fmaf(-v.y, t[nn].y, acc[nn]);
        }
      }
    }
    SC_SYNC();
  }
  float badd = 0.f;
  if (bias != nullptr) {
    int64_t gl = l0 + lane;
    if (gl >= lines) gl = lines - 1;
    badd = bias[(gl / lines_per_image) % channels];
  }
#pragma unroll
  for (int nn = 0; nn < NT; ++nn) otile[lane][w * NT + nn] = acc[nn] + badd;
  SC_SYNC();
  const int ncols = (N - ng0 < 4 * NT) ? (N - ng0) : 4 * NT;
  for (int idx = tid; idx < SC_LINES_PER_BLOCK * ncols; idx += SC_BLOCK) {
    const int l = idx / ncols, c = idx - l * ncols;
    if (l0 + l < lines) out[(l0 + l) * N + ng0 + c] = otile[l][c];
  }
}

// ------------------------------------------------------------------------------------------
// Mode-batched complex GEMM:  C[p,q,m] (+)= sum_r opA(A[p,r,m]) * opB(B[r,q,m])
// lanes = modes (every operand has the mode index innermost -> all accesses coalesced, no
// LDS, no cross-lane traffic); each lane keeps a PT x QT register tile; the 4 waves of a
// block take 4 adjacent p-tiles of the same (mode tile, q tile) so B is shared through L1.
// ------------------------------------------------------------------------------------------
struct ModeGemmArgs {
  int64_t P, Q, R, M;
  int64_t a_sp, a_sr, a_sm;
  int64_t b_sr, b_sq, b_sm;
  int64_t c_sp, c_sq, c_sm;
  const int32_t* b_idx;
  const int32_t* c_idx;
  int accumulate;
  // launch geometry: 1-D grid of 8 * per_xcd blocks, work item = (mode tile, p group, q tile)
  int n_mt, n_pg, n_qt, per_xcd;
  int r_split;                  // k_modegemm_msum: the reduction index r is cut into this many chunks (more workgroups)
};

template <int PT, int QT, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_modegemm(ModeGemmArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B,
           cf32* __restrict__ C) {
  const int tid = SC_TID;
  const int lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  // XCD-aware work mapping: the dispatcher places block b on XCD b % 8 (8 private L2s).  All
  // (p group, q tile) blocks of one mode tile re-read the same xhat / W slices, so consecutive
  // work items (mode tile slowest) are given to ONE XCD as a contiguous chunk: the re-reads then
  // hit that XCD's L2 instead of going back to Infinity Cache / HBM from eight different L2s.
  const int bid = SC_BID_X;
  const int item = (bid & 7) * g.per_xcd + (bid >> 3);
  if (item >= g.n_mt * g.n_pg * g.n_qt) return;
  const int mt = item / (g.n_pg * g.n_qt);
  const int rem = item - mt * (g.n_pg * g.n_qt);
  const int qt = rem / g.n_pg, pg = rem - qt * g.n_pg;
  const int64_t m = (int64_t)mt * SC_WAVE + lane;
  const int64_t p0 = ((int64_t)pg * 4 + w) * PT;            // wave-uniform
  const int64_t q0 = (int64_t)qt * QT;                      // wave-uniform
  if (p0 >= g.P) return;  // whole wave idle (no barriers in this kernel)
  const bool active = m < g.M;
  const int64_t mm = active ? m : g.M - 1;
  // per-lane part of every address: one 32-bit element offset per operand; everything else is
  // wave-uniform and stays in SGPRs (global_load ... v_off, s[base] addressing)
  const uint32_t la = (uint32_t)(mm * g.a_sm);
  const uint32_t lb = g.b_idx ? (uint32_t)g.b_idx[mm] : (uint32_t)(mm * g.b_sm);
  const uint32_t lc = g.c_idx ? (uint32_t)g.c_idx[mm] : (uint32_t)(mm * g.c_sm);

  const cf32* Ap[PT];
  const cf32* Bq[QT];
#pragma unroll
  for (int pp = 0; pp < PT; ++pp) {
    const int64_t p = (p0 + pp < g.P) ? (p0 + pp) : (g.P - 1);
    Ap[pp] = A + p * g.a_sp;
  }
#pragma unroll
  for (int qq = 0; qq < QT; ++qq) {
    const int64_t q = (q0 + qq < g.Q) ? (q0 + qq) : (g.Q - 1);
    Bq[qq] = B + q * g.b_sq;
  }
  cf32 acc[PT][QT];
#pragma unroll
  for (int pp = 0; pp < PT; ++pp)
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) acc[pp][qq] = cf_make(0.f, 0.f);

#pragma unroll 2
  for (int64_t r = 0; r < g.R; ++r) {
    cf32 a[PT], b[QT];
    const int64_t ra = r * g.a_sr, rb = r * g.b_sr;          // uniform
#pragma unroll
    for (int pp = 0; pp < PT; ++pp) {
      a[pp] = (Ap[pp] + ra)[la];
      if (CA) a[pp].y = -a[pp].y;
    }
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
      b[qq] = (Bq[qq] + rb)[lb];
      if (CB) b[qq].y = -b[qq].y;
    }
#pragma unroll
    for (int pp = 0; pp < PT; ++pp)
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) cf_mac(acc[pp][qq], a[pp], b[qq]);
  }
  if (!active) return;
#pragma unroll
  for (int pp = 0; pp < PT; ++pp) {
    if (p0 + pp >= g.P) continue;
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
      if (q0 + qq >= g.Q) continue;
      cf32* dst = C + ((p0 + pp) * g.c_sp + (q0 + qq) * g.c_sq) + lc;
      cf32 v = acc[pp][qq];
      if (g.accumulate) {
        const cf32 old = *dst;
        v = cf_add(v, old);
      }
      *dst = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The contraction with the reference's complex-half semantics (fno_block_precision "half" / "mixed":
// einsum_utils.py:10-36 einsum_complexhalf_two_input).  The reference views both operands as real, casts them to
// float16, evaluates FOUR real einsums t[x][y] = sum_r a_x b_y (x, y in {re, im}; float16 results of fp32 sums) and
// combines re = t00 - t11, im = t10 + t01 in float16.  Same roundings here, at the same points: operands rounded
// to float16 as they are loaded, four fp32 accumulators, each rounded once, the two combinations rounded once more.
// C holds float16-representable values in fp32 storage.  (Conjugation is exact in any precision.)
// ------------------------------------------------------------------------------------------
template <bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_modegemm_f16(ModeGemmArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ C) {
  constexpr int PT = 4, QT = 4;
  const int tid = SC_TID;
  const int lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  const int item = SC_BID_X;
  if (item >= g.n_mt * g.n_pg * g.n_qt) return;
  const int mt = item / (g.n_pg * g.n_qt);
  const int rem = item - mt * (g.n_pg * g.n_qt);
  const int qt = rem / g.n_pg, pg = rem - qt * g.n_pg;
  const int64_t m = (int64_t)mt * SC_WAVE + lane;
  const int64_t p0 = ((int64_t)pg * 4 + w) * PT;
  const int64_t q0 = (int64_t)qt * QT;
  if (p0 >= g.P) return;
  const bool active = m < g.M;
  const int64_t mm = active ? m : g.M - 1;
  const int64_t la = mm * g.a_sm;
  const int64_t lb = g.b_idx ? (int64_t)g.b_idx[mm] : mm * g.b_sm;
  const int64_t lc = g.c_idx ? (int64_t)g.c_idx[mm] : mm * g.c_sm;
  float rr[PT][QT], ii[PT][QT], ri[PT][QT], ir[PT][QT];
#pragma unroll
  for (int pp = 0; pp < PT; ++pp)
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) rr[pp][qq] = ii[pp][qq] = ri[pp][qq] = ir[pp][qq] = 0.f;
  for (int64_t r = 0; r < g.R; ++r) {
    cf32 a[PT], b[QT];
#pragma unroll
    for (int pp = 0; pp < PT; ++pp) {
      const int64_t p = (p0 + pp < g.P) ? (p0 + pp) : (g.P - 1);
      const cf32 v = A[p * g.a_sp + r * g.a_sr + la];
      a[pp] = cf_make(sc_round_f16(v.x), sc_round_f16(CA ? -v.y : v.y));
    }
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
      const int64_t q = (q0 + qq < g.Q) ? (q0 + qq) : (g.Q - 1);
      const cf32 v = B[r * g.b_sr + q * g.b_sq + lb];
      b[qq] = cf_make(sc_round_f16(v.x), sc_round_f16(CB ? -v.y : v.y));
    }
#pragma unroll
    for (int pp = 0; pp < PT; ++pp)
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) {
        rr[pp][qq] = fmaf(a[pp].x, b[qq].x, rr[pp][qq]);
        ii[pp][qq] = fmaf(a[pp].y, b[qq].y, ii[pp][qq]);
        ri[pp][qq] = fmaf(a[pp].x, b[qq].y, ri[pp][qq]);
        ir[pp][qq] = fmaf(a[pp].y, b[qq].x, ir[pp][qq]);
      }
  }
  if (!active) return;
#pragma unroll
  for (int pp = 0; pp < PT; ++pp) {
    if (p0 + pp >= g.P) continue;
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) {
      if (q0 + qq >= g.Q) continue;
      cf32 v;
      v.x = sc_round_f16(sc_round_f16(rr[pp][qq]) - sc_round_f16(ii[pp][qq]));
      v.y = sc_round_f16(sc_round_f16(ir[pp][qq]) + sc_round_f16(ri[pp][qq]));
      C[(p0 + pp) * g.c_sp + (q0 + qq) * g.c_sq + lc] = v;
    }
  }
}

// out[i] = float16(in[i]) kept in fp32 storage (in == out allowed)
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_round_f16(const float* __restrict__ in, float* __restrict__ out, int64_t n, int64_t stride) {
  for (int64_t i = (int64_t)SC_BID_X * SC_BLOCK + SC_TID; i < n; i += stride) out[i] = sc_round_f16(in[i]);
}

// ------------------------------------------------------------------------------------------
// Mode-summed complex GEMM:  C[p,q] += sum_m sum_r opA(A[p,r,m]) * opB(B[r,q,m])
// the gradient of a mode-INDEPENDENT operand (Tucker / CP factor matrices, spectral_convolution.py
// :55-103): same lanes-are-modes tiles as k_modegemm; a workgroup walks every per_xcd-th mode tile
// (per_xcd = number of mode splits here) with per-lane partial sums, then ONE wave reduction and one
// atomic add per (p, q).  (One atomic per mode TILE was 729 adds onto each of 627 addresses for a
// 19 x 33 gradient over 46656 modes: 300 us of atomic serialisation.)  C must be zeroed by the caller.
// PART (session 2): no atomics -- every (mode split, r split) pair owns one slot [P][Q] of a workspace that C points
// at, written exactly once per (p, q); k_fmx_reduce then adds the slots in a fixed order, so that the result is the
// same bits on every run (float atomics land in arrival order: the small Tucker / CP factor gradients that take this
// kernel differed in the last bits from run to run, seen by the hipGraph-against-eager test).
// ------------------------------------------------------------------------------------------
template <int PT, int QT, bool CA, bool CB, bool PART = false>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_modegemm_msum(ModeGemmArgs g, const cf32* __restrict__ A, const cf32* __restrict__ B,
                cf32* __restrict__ C) {
  SC_SHARED float red[4 * 32 * 65];
  const int tid = SC_TID;
  const int lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  const int rsp = g.r_split > 1 ? g.r_split : 1;
  const int item = SC_BID_X / rsp, rs = SC_BID_X - item * rsp;
  const int64_t rch = (g.R + rsp - 1) / rsp, r_lo = rs * rch, r_hi = (r_lo + rch < g.R) ? r_lo + rch : g.R;
  const int ms = item / (g.n_pg * g.n_qt);                   // mode split: tiles ms, ms + per_xcd, ...
  const int rem = item - ms * (g.n_pg * g.n_qt);
  const int qt = rem / g.n_pg, pg = rem - qt * g.n_pg;
  const int64_t p0 = ((int64_t)pg * 4 + w) * PT;
  const int64_t q0 = (int64_t)qt * QT;
  const bool wave_on = p0 < g.P;                            // wave-uniform; idle waves still meet the syncs
  cf32 acc[PT][QT];
#pragma unroll
  for (int pp = 0; pp < PT; ++pp)
#pragma unroll
    for (int qq = 0; qq < QT; ++qq) acc[pp][qq] = cf_make(0.f, 0.f);
  for (int mt = ms; mt < g.n_mt && wave_on; mt += g.per_xcd) {
    const int64_t m = (int64_t)mt * SC_WAVE + lane;
    if (m >= g.M) continue;
    const uint32_t la = (uint32_t)(m * g.a_sm);
    const uint32_t lb = g.b_idx ? (uint32_t)g.b_idx[m] : (uint32_t)(m * g.b_sm);
#ifndef SC_MSUM_UNROLL
#define SC_MSUM_UNROLL 2                                      // steps of operand loads in flight (4: 49.6 -> 57.9 us at TFNO rank 0.1)
#endif
#define SC_PRAGMA_STR(x) _Pragma(#x)
#define SC_PRAGMA_UNROLL(n) SC_PRAGMA_STR(unroll n)       // (a macro inside `#pragma unroll` does not survive --save-temps)
SC_PRAGMA_UNROLL(SC_MSUM_UNROLL)
    for (int64_t r = r_lo; r < r_hi; ++r) {
      cf32 a[PT], b[QT];
#pragma unroll
      for (int pp = 0; pp < PT; ++pp) {
        const int64_t p = (p0 + pp < g.P) ? (p0 + pp) : (g.P - 1);
        a[pp] = (A + p * g.a_sp + r * g.a_sr)[la];
        if (CA) a[pp].y = -a[pp].y;
      }
#pragma unroll
      for (int qq = 0; qq < QT; ++qq) {
        const int64_t q = (q0 + qq < g.Q) ? (q0 + qq) : (g.Q - 1);
        b[qq] = (B + r * g.b_sr + q * g.b_sq)[lb];
        if (CB) b[qq].y = -b[qq].y;
      }
#pragma unroll
      for (int pp = 0; pp < PT; ++pp)
#pragma unroll
        for (int qq = 0; qq < QT; ++qq) cf_mac(acc[pp][qq], a[pp], b[qq]);
    }
  }
  // wave reduction, transposed: the 2 PT QT partial sums of every lane go to LDS as [value][lane] (row stride 65:
  // conflict-free both ways), lane v then adds the 64 partials of value v and issues ONE atomic.  (The first version
  // reduced one value at a time through a 6-level tree with three wave syncs per level -- 144 sync rounds for 8
  // values; with one mode tile per workgroup that cost more than the accumulation: 82 us for the 64 x 36 factor
  // gradients of TFNO rank 0.1, profiles/r02_tfno_kernel_stats.txt.)
  constexpr int NV = 2 * PT * QT;                             // floats per lane
  static_assert(NV % 32 == 0 || NV <= 32, "values are reduced 32 at a time");
  float* rw = red + w * (32 * 65);
#pragma unroll
  for (int h0 = 0; h0 < NV; h0 += 32) {
#pragma unroll
    for (int v = 0; v < 32 && h0 + v < NV; ++v) {
      const int i = (h0 + v) >> 1;                            // acc index: pp = i / QT, qq = i % QT
      const cf32 c = acc[i / QT][i % QT];
      rw[v * 65 + lane] = ((h0 + v) & 1) ? c.y : c.x;
    }
    SC_WAVE_SYNC();
    if (lane < 32 && h0 + lane < NV) {
      float sum = 0.f;
#pragma unroll 8
      for (int l = 0; l < SC_WAVE; ++l) sum += rw[lane * 65 + l];
      const int i = (h0 + lane) >> 1;
      const int pp = i / QT, qq = i % QT;
      if (wave_on && p0 + pp < g.P && q0 + qq < g.Q) {
        if (PART) {
          const int64_t slot = (int64_t)ms * rsp + rs;
          reinterpret_cast<float*>(C + (slot * g.P + p0 + pp) * g.Q + (q0 + qq))[(h0 + lane) & 1] = sum;
        } else {
          float* dst = reinterpret_cast<float*>(C + (p0 + pp) * g.c_sp + (q0 + qq) * g.c_sq) + ((h0 + lane) & 1);
#ifndef SC_EMU
          atomicAdd(dst, sum);
#else
          *dst += sum;      // emulated workgroups run one after another, waves own different p
#endif
        }
      }
    }
    SC_WAVE_SYNC();
  }
}

// gbias[c] = sum_b Re(ghat[(b * channels + c) * modes_per_image + dc]): one wave per channel, lanes run over the
// batch, tree reduction through 64 floats of LDS (the order k_bias_grad uses: same bits)
SC_DEVICE void sc_bias_grad_wave(const cf32* __restrict__ ghat, float* __restrict__ gbias, const int64_t batch,
                                 const int64_t channels, const int64_t modes_per_image, const int64_t dc,
                                 const int64_t c, const int lane, float* part) {
  float s = 0.f;
  for (int64_t b = lane; b < batch; b += SC_WAVE) s += ghat[(b * channels + c) * modes_per_image + dc].x;
  part[lane] = s;
  SC_WAVE_SYNC();
#pragma unroll
  for (int off = SC_WAVE / 2; off > 0; off >>= 1) {
    const float o = (lane < off) ? part[lane + off] : 0.f;
    SC_WAVE_SYNC();
    if (lane < off) part[lane] += o;
    SC_WAVE_SYNC();
  }
  if (lane == 0) gbias[c] = part[0];
}

SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_WAVE)
k_bias_grad(const cf32* __restrict__ ghat, float* __restrict__ gbias, int64_t batch, int64_t channels,
            int64_t modes_per_image, int64_t dc) {
  SC_SHARED float part[SC_WAVE];
  sc_bias_grad_wave(ghat, gbias, batch, channels, modes_per_image, dc, SC_BID_X, SC_TID, part);
}


// ------------------------------------------------------------------------------------------
// Block epilogue as its own pass, for the shapes whose inverse transform does not carry it in its store path
// (everything off the fused 2-D kernels): y = act(y + skip), pre-activation optionally saved (SURVEY.md 8 row f1;
// neuralop/layers/fno_block.py:392-414).  Streaming, 16 bytes per lane.
// ------------------------------------------------------------------------------------------
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_epilogue(float* __restrict__ y, const float* __restrict__ skip, float* __restrict__ preact, int act,
           int64_t n, int64_t stride) {
  for (int64_t i = (int64_t)SC_BID_X * SC_BLOCK + SC_TID; i < n; i += stride) {
    float v = y[i] + skip[i];
    if (act == 1) {
      if (preact != nullptr) preact[i] = v;
      v = sc_gelu(v);              // the SAME function as the fused store path (sc_device.h): identical bits on every route
    }
    y[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Sharded spectrum layout for the shapes whose transform kernels do not address it natively (everything off the fused
// 2-D kernels): ONE streaming permutation between the plain [image][k1][rest] block and the rank-major all-to-all
// buffer [block][image][rows][rest] (include/sc_engine.h, sc_spectrum_shards).  TO_SHARDS: plain -> sharded (rows
// past k1 of the last blocks are written as zeros), else sharded -> plain.
// ------------------------------------------------------------------------------------------
template <bool TO_SHARDS>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_spectrum_shard(const cf32* __restrict__ src, cf32* __restrict__ dst, int64_t n_images, int64_t k1, int64_t rest,
                 int64_t rows, int64_t n_blocks, int64_t block_stride, int64_t stride) {
  const int64_t per_img = rows * n_blocks * rest;           // padded rows included
  const int64_t total = n_images * per_img;
  for (int64_t e = (int64_t)SC_BID_X * SC_BLOCK + SC_TID; e < total; e += stride) {
    const int64_t img = e / per_img, rem = e - img * per_img;
    const int64_t row = rem / rest, col = rem - row * rest;
    const int64_t blk = row / rows;
    const int64_t sh = blk * block_stride + (img * rows + (row - blk * rows)) * rest + col;
    const int64_t pl = (img * k1 + row) * rest + col;
    if (TO_SHARDS) dst[sh] = row < k1 ? src[pl] : cf_make(0.f, 0.f);
    else if (row < k1) dst[pl] = src[sh];
  }
}

// ------------------------------------------------------------------------------------------
// Fused AdamW step (neuralop/training/adamw.py:155-200 without the GaLore projection): one read of
// (p, g, m, v), one write of (p, m, v).  Streaming: 7 arrays of the weight's size cross HBM once.
// CPX: elements are complex64 -- m and the update are complex, v accumulates g conj(g) = |g|^2 (its
// imaginary lane only decays, as in the reference's complex exp_avg_sq).
// ------------------------------------------------------------------------------------------
struct AdamwArgs {
  float b1, b2, one_m_b1, one_m_b2, eps, step_size, decay;   // decay = lr * weight_decay (0: none)
};

template <bool CPX>
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_BLOCK)
k_adamw(AdamwArgs a, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
        float* __restrict__ v, int64_t n_pairs, int64_t stride) {
  // one thread = two floats (one complex value, or two neighbouring real ones); stride = threads in the grid
  for (int64_t i = (int64_t)SC_BID_X * SC_BLOCK + SC_TID; i < n_pairs; i += stride) {
    const cf32 gg = reinterpret_cast<const cf32*>(g)[i];
    cf32 mm = reinterpret_cast<cf32*>(m)[i], vv = reinterpret_cast<cf32*>(v)[i], pp = reinterpret_cast<cf32*>(p)[i];
    mm.x = a.b1 * mm.x + a.one_m_b1 * gg.x;
    mm.y = a.b1 * mm.y + a.one_m_b1 * gg.y;
    float dx, dy;
    if (CPX) {
      vv.x = a.b2 * vv.x + a.one_m_b2 * (gg.x * gg.x + gg.y * gg.y);
      vv.y = a.b2 * vv.y;
      dx = dy = sqrtf(vv.x) + a.eps;
    } else {
      vv.x = a.b2 * vv.x + a.one_m_b2 * (gg.x * gg.x);
      vv.y = a.b2 * vv.y + a.one_m_b2 * (gg.y * gg.y);
      dx = sqrtf(vv.x) + a.eps;
      dy = sqrtf(vv.y) + a.eps;
    }
    pp.x -= a.step_size * (mm.x / dx);
    pp.y -= a.step_size * (mm.y / dy);
    if (a.decay > 0.f) {
      pp.x -= a.decay * pp.x;
      pp.y -= a.decay * pp.y;
    }
    reinterpret_cast<cf32*>(m)[i] = mm;
    reinterpret_cast<cf32*>(v)[i] = vv;
    reinterpret_cast<cf32*>(p)[i] = pp;
  }
}

// odd real element count: the last float on its own
SC_GLOBAL void SC_LAUNCH_BOUNDS(SC_WAVE)
k_adamw_tail(AdamwArgs a, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
             float* __restrict__ v, int64_t idx) {
  if (SC_TID != 0) return;
  const float gg = g[idx];
  const float mm = a.b1 * m[idx] + a.one_m_b1 * gg;
  const float vv = a.b2 * v[idx] + a.one_m_b2 * gg * gg;
  float pp = p[idx] - a.step_size * (mm / (sqrtf(vv) + a.eps));
  if (a.decay > 0.f) pp -= a.decay * pp;
  m[idx] = mm;
  v[idx] = vv;
  p[idx] = pp;
}
