// sc_kernels_mdft.h -- the size-agnostic transform passes on the matrix cores.
//
// The generic path evaluates every 1-D pruned / zero-padded DFT directly from a twiddle table
// (any size, odd sizes, any kept-mode count).  A direct DFT over one axis IS a GEMM between the data
// and a CONSTANT matrix, with all other indices as an enormous batch dimension -- exactly the shape
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 155 TF/s measured) is built for, and 2-5x what the
// lanes-are-lines VALU kernels of sc_kernels_generic.h reach (they stay as the fallback for the
// shapes excluded below).  No LDS: the data operand comes straight from global memory in MFMA
// operand order, the table operand is pre-arranged on the host in lane order (one coalesced 256-byte
// read per MFMA, L1/L2 resident).
//
//   k_mdft_r2c   last axis, real -> complex, pruned:        out[l][j]    = sum_n in[l][n] T[n][j]
//   k_mdft_axis  non-last axis, complex -> complex:         out[o][j][i] = sum_n T[n][j] in[o][n][i]
//   k_mdft_c2r   last axis, complex -> real, zero padded:   out[l][n]    = sum_j Re(in[l][j] T[j][n]) + bias
//
// replacing, like their VALU twins, the per-axis pieces of rfftn/ifftn/irfft restricted to the
// kept modes (spectral_convolution.py:443-449, 500-519, 531-568).
#pragma once
#include "sc_kernels_mfma.h"

// row of the 32 x 32 MFMA result held in accumulator register v of a lane in half `half`
SC_HD int mdft_row(const int v, const int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------
// last axis, real -> complex.  Result tile D[32 lines][32 output floats (2 j + c')].
//   A operand = data: lane (line, h) holds in[line][8 t + 4 h + q], q = 0..3 (one float4 load);
//   B operand = table: tab[((ct * NG + t) * 4 + q) * 64 + lane]
// requires N % 8 == 0 (16-byte aligned float4 loads); RT row tiles x CT column tiles per wave.
// ------------------------------------------------------------------------------------------
template <int RT, int CT>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_mdft_r2c(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
           int64_t lines, int N, int J, int n_ct) {
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t item = (int64_t)SC_BID_X * 4 + w;               // one wave = RT row tiles
  const int64_t l0 = item * (32 * RT);
  if (l0 >= lines) return;
  const int NG = N / 8;
#pragma unroll 1
  for (int ct0 = 0; ct0 < n_ct; ct0 += CT) {
    sc_f32x16 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;
    const float* rowp[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int64_t line = l0 + 32 * r + col;
      if (line >= lines) line = lines - 1;
      rowp[r] = in + line * N + 4 * half;
    }
    // operands of step t+1 are requested before the MFMAs of step t are issued (two register sets)
    float a0[RT][4], b0[CT][4], a1[RT][4], b1[CT][4];
    auto fetch = [&](const int t, float (&a)[RT][4], float (&b)[CT][4]) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const cf32 lo = *reinterpret_cast<const cf32*>(rowp[r] + 8 * t);
        const cf32 hi = *reinterpret_cast<const cf32*>(rowp[r] + 8 * t + 2);
        a[r][0] = lo.x; a[r][1] = lo.y; a[r][2] = hi.x; a[r][3] = hi.y;
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int ct = (ct0 + c < n_ct) ? ct0 + c : n_ct - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[c][q] = tab[(((int64_t)ct * NG + t) * 4 + q) * 64 + lane];
      }
    };
    auto multiply = [&](const float (&a)[RT][4], const float (&b)[CT][4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int c = 0; c < CT; ++c) sc_mfma_32x32x2(acc[r][c], a[r][q], b[c][q]);
    };
    fetch(0, a0, b0);
    int t = 0;
#pragma unroll 1
    for (; t + 1 < NG; t += 2) {                         // single-exit loop: the accumulators stay put
      fetch(t + 1, a1, b1);
      SC_SCHED_BARRIER();
      multiply(a0, b0);
      SC_SCHED_BARRIER();
      if (t + 2 < NG) fetch(t + 2, a0, b0);
      SC_SCHED_BARRIER();
      multiply(a1, b1);
      SC_SCHED_BARRIER();
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) SC_PIN_ACC(acc[r][c]);
    }
    if (t < NG) multiply(a0, b0);                        // odd step count
    const int64_t l0e = l0 + sc_opaque(0);               // keep the store addresses out of the loop
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int f = 32 * (ct0 + c) + col;                     // output float index inside the line
        if (ct0 + c < n_ct && f < 2 * J) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int64_t line = l0e + 32 * r + mdft_row(v, half);
            if (line < lines) out[line * 2 * J + f] = acc[r][c][v];
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------
// non-last axis, complex -> complex.  Result tile D[32 rows (2 j + c')][32 columns (o, i)].
//   A operand = table: tab[((jt * NS + s) * 2 + comp) * 64 + lane], row = lane & 31, n = 2 s + (lane >> 5)
//   B operand = data: lane (col, h) holds in[o][2 s + h][i] (one 8-byte load), .x for comp 0, .y for comp 1
// JT j-tiles (16 j each) x CT column tiles per wave.
// ------------------------------------------------------------------------------------------
template <int JT, int CT>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_mdft_axis(const cf32* __restrict__ in, cf32* __restrict__ out, const float* __restrict__ tab,
            int64_t outer, int N, int J, int64_t inner, int n_jt) {
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t ncols = outer * inner;
  const int64_t item = (int64_t)SC_BID_X * 4 + w;
  const int64_t c0 = item * (32 * CT);
  if (c0 >= ncols) return;
  const int NS = (N + 1) / 2;
  const cf32* colp[CT];
  int64_t obase[CT];
  bool cok[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    int64_t cc = c0 + 32 * c + col;
    cok[c] = cc < ncols;
    if (!cok[c]) cc = ncols - 1;
    const int64_t o = cc / inner, i = cc - o * inner;
    colp[c] = in + (o * N) * inner + i;
    obase[c] = (o * J) * inner + i;
  }
#pragma unroll 1
  for (int jt0 = 0; jt0 < n_jt; jt0 += JT) {
    sc_f32x16 acc[JT][CT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][c][v] = 0.f;
    cf32 d0[CT], d1[CT];
    float a0[JT][2], a1[JT][2];
    auto fetch = [&](const int s, cf32 (&d)[CT], float (&a)[JT][2]) {
      int n = 2 * s + half;
      if (n >= N) n = N - 1;                                  // table entry is zero there
#pragma unroll
      for (int c = 0; c < CT; ++c) d[c] = colp[c][(int64_t)n * inner];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int jt = (jt0 + j < n_jt) ? jt0 + j : n_jt - 1;
        a[j][0] = tab[(((int64_t)jt * NS + s) * 2 + 0) * 64 + lane];
        a[j][1] = tab[(((int64_t)jt * NS + s) * 2 + 1) * 64 + lane];
      }
    };
    auto multiply = [&](const cf32 (&d)[CT], const float (&a)[JT][2]) {
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          sc_mfma_32x32x2(acc[j][c], a[j][0], d[c].x);
          sc_mfma_32x32x2(acc[j][c], a[j][1], d[c].y);
        }
    };
    fetch(0, d0, a0);
    int s = 0;
#pragma unroll 1
    for (; s + 1 < NS; s += 2) {
      fetch(s + 1, d1, a1);
      SC_SCHED_BARRIER();
      multiply(d0, a0);
      SC_SCHED_BARRIER();
      if (s + 2 < NS) fetch(s + 2, d0, a0);
      SC_SCHED_BARRIER();
      multiply(d1, a1);
      SC_SCHED_BARRIER();
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int c = 0; c < CT; ++c) SC_PIN_ACC(acc[j][c]);
    }
    if (s < NS) multiply(d0, a0);
    const int64_t zo = sc_opaque(0);
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (jt0 + j < n_jt && cok[c]) {
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            const int jj = 16 * (jt0 + j) + (mdft_row(v, half) >> 1);
            if (jj < J) out[obase[c] + zo + (int64_t)jj * inner] = cf_make(acc[j][c][v], acc[j][c][v + 1]);
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------
// last axis, complex -> real, zero padded.  Result tile D[32 lines][32 outputs n].
//   A operand = data: lane (line, h) holds in[line][2 t + h] (one 8-byte load), .x / .y for comp 0 / 1
//   B operand = table: tab[((nt * JS + t) * 2 + comp) * 64 + lane]
// ------------------------------------------------------------------------------------------
template <int RT, int CT>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_mdft_c2r(const cf32* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
           const float* __restrict__ bias, int64_t lines, int N, int J, int n_nt,
           int64_t lines_per_image, int64_t channels) {
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t item = (int64_t)SC_BID_X * 4 + w;
  const int64_t l0 = item * (32 * RT);
  if (l0 >= lines) return;
  const int JS = (J + 1) / 2;
  // the host only takes this kernel with a bias when a wave's 32 RT lines lie inside one image
  // (lines_per_image % (32 RT) == 0), so the bias is one wave-uniform scalar
  const float badd = (bias != nullptr) ? bias[(l0 / lines_per_image) % channels] : 0.f;
  const cf32* rowp[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    int64_t line = l0 + 32 * r + col;
    if (line >= lines) line = lines - 1;
    rowp[r] = in + line * J;
  }
#pragma unroll 1
  for (int nt0 = 0; nt0 < n_nt; nt0 += CT) {
    sc_f32x16 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;
    cf32 d0[RT], d1[RT];
    float b0[CT][2], b1[CT][2];
    auto fetch = [&](const int t, cf32 (&d)[RT], float (&b)[CT][2]) {
      int j = 2 * t + half;
      if (j >= J) j = J - 1;                                  // table entry is zero there
#pragma unroll
      for (int r = 0; r < RT; ++r) d[r] = rowp[r][j];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int nt = (nt0 + c < n_nt) ? nt0 + c : n_nt - 1;
        b[c][0] = tab[(((int64_t)nt * JS + t) * 2 + 0) * 64 + lane];
        b[c][1] = tab[(((int64_t)nt * JS + t) * 2 + 1) * 64 + lane];
      }
    };
    auto multiply = [&](const cf32 (&d)[RT], const float (&b)[CT][2]) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          sc_mfma_32x32x2(acc[r][c], d[r].x, b[c][0]);
          sc_mfma_32x32x2(acc[r][c], d[r].y, b[c][1]);
        }
    };
    fetch(0, d0, b0);
    int t = 0;
#pragma unroll 1
    for (; t + 1 < JS; t += 2) {
      fetch(t + 1, d1, b1);
      SC_SCHED_BARRIER();
      multiply(d0, b0);
      SC_SCHED_BARRIER();
      if (t + 2 < JS) fetch(t + 2, d0, b0);
      SC_SCHED_BARRIER();
      multiply(d1, b1);
      SC_SCHED_BARRIER();
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) SC_PIN_ACC(acc[r][c]);
    }
    if (t < JS) multiply(d0, b0);
    const int64_t l0e = l0 + sc_opaque(0);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t line = l0e + 32 * r + mdft_row(v, half);
        if (line < lines) {
          float* orow = out + line * N;
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const int n = 32 * (nt0 + c) + col;
            if (nt0 + c < n_nt && n < N) orow[n] = acc[r][c][v] + badd;
          }
        }
      }
  }
}
