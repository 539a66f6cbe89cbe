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
//
// Later generations of the two last-axis passes, all in this file:
//   k_mdft_r2c_lds / k_mdft_c2r_lds   128-line tiles AND the whole table through LDS (N % 32 == 0, N <= 256, small tables),
//                                     also in "plane" form (the second-to-last axis in the same launch)
//   k_mdft_r2c_stage                  any width: 128-line tiles through LDS in 32-sample chunks, table streamed from L2
//   k_mdft_c2r_span / k_mdft_c2r_stage any width: 32-line blocks whose N-line span is staged in LDS and written as whole
//                                     aligned lines / 128-line tiles with direct stores; table streamed from L2
#pragma once
#include "sc_kernels_mfma.h"

// measurement builds only (scripts/mdft_time.py): take one resource out of a pass to see what it costs
#ifdef SC_MDFT_ABL_NOMFMA
#define MDFT_MFMA(acc, a, b) ((acc)[0] = fmaf((a), (b), (acc)[0]))
#else
#define MDFT_MFMA(acc, a, b) sc_mfma_32x32x2((acc), (a), (b))
#endif
#ifdef SC_MDFT_ABL_NOSTORE
#define MDFT_STORE_OK(val) ((val) == 12345.678f)
#else
#define MDFT_STORE_OK(val) true
#endif

// row of the 32 x 32 MFMA result held in accumulator register v of a lane in half `half`
SC_HD int mdft_row(const int v, const int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------
// last axis, real -> complex.  Result tile D[32 lines][32 output floats (2 j + c')].
//   A operand = data: lane (line, h) holds in[line][8 t + 4 h + q], q = 0..3 (one float4 load);
//   B operand = table: tab[((ct * NG + t) * 4 + q) * 64 + lane]
// RT row tiles x CT column tiles per wave.  RAGGED = false: N % 8 == 0, two 8-byte loads per lane and step.
// RAGGED = true (round 4: any N -- the reference's documented Darcy grids are 85 / 141 / 211 / 421 points wide, and until
// then every width that is not a multiple of 8 fell to the scalar VALU pass k_last_r2c: 650 us per call at 421^2 where this
// kernel needs ~190): NG = ceil(N / 8) steps, the table's rows past N are zero, and a lane reads its four points with
// 4-byte loads whose index is clamped to the line (rows are not 8-byte aligned when N is odd; nothing is read past a line).
// ------------------------------------------------------------------------------------------
// TAIL: kept-mode counts of the form 2^k + 1 (n_modes/2 + 1 with power-of-two n_modes: the usual
// case) put exactly one complex column past a 32-float tile boundary; a whole MFMA column tile for 2
// of 32 columns was 47 % of this pass's matrix work at J = 17.  That column is a plain dot product
// per line on the VALU instead (tail[(t*4 + q)*2 + h] = T[8t + 4h + q][J-1]), riding on the operand
// registers the MFMAs already hold.
template <int RT, int CT, bool TAIL, bool RAGGED = false>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_mdft_r2c(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
           const cf32* __restrict__ tail, int64_t lines, int N, int J, int n_ct) {
  SC_SHARED cf32 tsum[4][64][RT];
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t item = (int64_t)SC_BID_X * 4 + w;               // one wave = RT row tiles
  const int64_t l0 = item * (32 * RT);
  if (l0 >= lines) return;
  const int NG = (N + 7) / 8;
  cf32 tacc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) tacc[r] = cf_make(0.f, 0.f);
#pragma unroll 1
  for (int ct0 = 0; ct0 < n_ct; ct0 += CT) {
    const bool do_tail = TAIL && ct0 == 0;
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
      rowp[r] = in + line * N;
    }
    // operands of step t+1 are requested before the MFMAs of step t are issued (two register sets)
    float a0[RT][4], b0[CT][4], a1[RT][4], b1[CT][4];
    cf32 w0[4], w1[4];
    auto fetch = [&](const int t, float (&a)[RT][4], float (&b)[CT][4], cf32 (&tw)[4]) {
      const int n0 = 8 * t + 4 * half;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
#ifdef SC_MDFT_ABL_NOLOAD
        a[r][0] = (float)(n0 + r); a[r][1] = a[r][0] + 1.f; a[r][2] = a[r][0] + 2.f; a[r][3] = a[r][0] + 3.f;
#else
        if constexpr (RAGGED) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + q;
            const float v = rowp[r][n < N ? n : N - 1];           // unconditional, inside the line
            a[r][q] = n < N ? v : 0.f;
          }
        } else {
          const cf32 lo = *reinterpret_cast<const cf32*>(rowp[r] + n0);
          const cf32 hi = *reinterpret_cast<const cf32*>(rowp[r] + n0 + 2);
          a[r][0] = lo.x; a[r][1] = lo.y; a[r][2] = hi.x; a[r][3] = hi.y;
        }
#endif
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int ct = (ct0 + c < n_ct) ? ct0 + c : n_ct - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[c][q] = tab[(((int64_t)ct * NG + t) * 4 + q) * 64 + lane];
      }
      if (TAIL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tw[q] = tail[(t * 4 + q) * 2 + half];
      }
    };
    auto multiply = [&](const float (&a)[RT][4], const float (&b)[CT][4], const cf32 (&tw)[4]) {
      if (do_tail) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            tacc[r].x = fmaf(a[r][q], tw[q].x, tacc[r].x);
            tacc[r].y = fmaf(a[r][q], tw[q].y, tacc[r].y);
          }
      }
      SC_SCHED_BARRIER();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int c = 0; c < CT; ++c) MDFT_MFMA(acc[r][c], a[r][q], b[c][q]);
    };
    fetch(0, a0, b0, w0);
    int t = 0;
#pragma unroll 1
    for (; t + 1 < NG; t += 2) {                         // single-exit loop: the accumulators stay put
      fetch(t + 1, a1, b1, w1);
      SC_SCHED_BARRIER();
      multiply(a0, b0, w0);
      SC_SCHED_BARRIER();
      // unconditional (the last trip re-requests the final step): a uniform branch around the loads makes hipcc wait
      // for them at the join, i.e. every trip pays a full memory latency (seen in k_mdft_axis: vmcnt(0) per trip)
      fetch(t + 2 < NG ? t + 2 : NG - 1, a0, b0, w0);
      SC_SCHED_BARRIER();
      multiply(a1, b1, w1);
      SC_SCHED_BARRIER();
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) SC_PIN_ACC(acc[r][c]);
    }
    if (t < NG) multiply(a0, b0, w0);                    // odd step count
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
            if (line < lines && MDFT_STORE_OK(acc[r][c][v])) out[line * 2 * J + f] = acc[r][c][v];
          }
        }
      }
  }
  if (TAIL) {
    // the two halves of a line's dot product meet through LDS; lane (line, 0) stores column J - 1
#pragma unroll
    for (int r = 0; r < RT; ++r) tsum[w][lane][r] = tacc[r];
    SC_WAVE_SYNC();
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int64_t line = l0 + 32 * r + col;
        if (line < lines) {
          const cf32 o = tsum[w][lane + 32][r];
          reinterpret_cast<cf32*>(out)[line * J + (J - 1)] = cf_add(tacc[r], o);
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
// Round 4: grid.y = group of JT j-tiles (one pass over the input per block: the loop over groups re-read the input anyway,
// and a zero-padded inverse pass over few columns -- 512 images x 17 kept columns of a 421-row grid: 272 waves on 1024
// SIMDs, 27 groups each -- was latency bound: 66 us for 29 MB), and KS = true splits the N input rows over the FOUR waves
// of a block (one item per block instead of four), partial sums meeting in LDS: the forward pass of the same case runs
// 211 dependent load -> MFMA steps per wave with the chip three quarters empty (74 us for 29 MB).
// ------------------------------------------------------------------------------------------
template <int JT, int CT, bool KS = false>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (KS ? 2 : 1))      // (KS: the reduction's loads all in flight took 320 registers)
k_mdft_axis(const cf32* __restrict__ in, cf32* __restrict__ out, const float* __restrict__ tab,
            int64_t outer, int N, int J, int64_t inner, int n_jt) {
  SC_SHARED float red[KS ? 3 * JT * CT * 1024 : 1];
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t ncols = outer * inner;
  const int64_t item = KS ? (int64_t)SC_BID_X : (int64_t)SC_BID_X * 4 + w;
  const int64_t c0 = item * (32 * CT);
  if (c0 >= ncols) return;
  const int NS = (N + 1) / 2;
  const int nsq = KS ? (NS + 3) / 4 : NS;                      // this wave's steps [s_lo, s_hi)
  const int s_lo = KS ? w * nsq : 0;
  const int s_hi = (s_lo + nsq < NS) ? s_lo + nsq : NS;
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
  {
    const int jt0 = (int)SC_BID_Y * JT;
    sc_f32x16 acc[JT][CT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][c][v] = 0.f;
    // a ring of four steps: step s + 3 is requested before step s is multiplied (one step ahead left every step
    // waiting for L2 / HBM with few waves per SIMD: 39 of 74 us at 512 images x 33 columns were the loads alone).
    // Every request is unconditional -- steps past the end re-read the last one and are multiplied by zero: a uniform
    // branch around the loads makes hipcc wait for them at the join (vmcnt(0) per trip).
    cf32 dd[4][CT];
    float aa[4][JT][2];
    auto fetch = [&](const int sq, cf32 (&d)[CT], float (&a)[JT][2]) {
      const int s = sq < s_hi ? sq : s_hi - 1;
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
    auto multiply = [&](const bool live, const cf32 (&d)[CT], const float (&a)[JT][2]) {
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          MDFT_MFMA(acc[j][c], a[j][0], live ? d[c].x : 0.f);
          MDFT_MFMA(acc[j][c], a[j][1], live ? d[c].y : 0.f);
        }
    };
    if (s_lo < s_hi) {
#pragma unroll
      for (int u = 0; u < 3; ++u) fetch(s_lo + u, dd[u], aa[u]);
#pragma unroll 1
      for (int s = s_lo; s < s_hi; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          fetch(s + u + 3, dd[(u + 3) & 3], aa[(u + 3) & 3]);
          SC_SCHED_BARRIER();
          multiply(s + u < s_hi, dd[u], aa[u]);
          SC_SCHED_BARRIER();
        }
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
          for (int c = 0; c < CT; ++c) SC_PIN_ACC(acc[j][c]);
      }
    }
    // KS: tile t = j CT + c is OWNED by wave t & 3 -- the other three waves hand their partial sums of it over through
    // LDS, the owner adds them in the fixed order owner + 1, + 2, + 3 (mod 4) and stores the tile.  (Round 4 sent
    // everything to wave 0: 192 LDS reads per lane in one wave, whose scheduling wanted 320 registers and spilled at the
    // kernel's 256 -- 28 / 60 bytes of scratch, VERDICT r4 weak 7 -- while three waves idled through the stores.)
    if (KS) {
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int t = j * CT + c, own = t & 3;
          if (w != own) {                                          // uniform
            const int k = (w - own - 1) & 3;
#pragma unroll
            for (int v = 0; v < 16; ++v) red[((t * 3 + k) * 16 + v) * 64 + lane] = acc[j][c][v];
          }
        }
      SC_SYNC();
    }
    const int64_t zo = sc_opaque(0);
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int t = j * CT + c;
        if (KS && w != (t & 3)) continue;                          // uniform: not this wave's tile
        if (KS) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][c][v] += red[((t * 3 + k) * 16 + v) * 64 + lane];
        }
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
           int64_t lines_per_image, int64_t channels, int per_line_bias) {
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t item = (int64_t)SC_BID_X * 4 + w;
  const int64_t l0 = item * (32 * RT);
  if (l0 >= lines) return;
  const int JS = (J + 1) / 2;
  // a wave's 32 RT lines lie inside one image when lines_per_image % (32 RT) == 0: the bias is then one wave-uniform
  // scalar; otherwise (round 4: image heights such as 421) per_line_bias = 1 and every stored row looks its own value up
  const float badd = (bias != nullptr && !per_line_bias) ? bias[(l0 / lines_per_image) % channels] : 0.f;
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
          MDFT_MFMA(acc[r][c], d[r].x, b[c][0]);
          MDFT_MFMA(acc[r][c], d[r].y, b[c][1]);
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
      fetch(t + 2 < JS ? t + 2 : JS - 1, d0, b0);              // unconditional: see k_mdft_r2c
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
          const float bl = (per_line_bias && bias != nullptr) ? bias[(line / lines_per_image) % channels] : badd;
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const int n = 32 * (nt0 + c) + col;
            if (nt0 + c < n_nt && n < N && MDFT_STORE_OK(acc[r][c][v])) {
#ifdef SC_MDFT_PLAIN_STORE
              orow[n] = acc[r][c][v] + bl;
#else
              SC_STORE_STREAM(&orow[n], acc[r][c][v] + bl);     // 2 GB written once, read by a later kernel
#endif
            }
          }
        }
      }
  }
}

// ==========================================================================================
// LDS-staged last-axis passes (generation 2 of k_mdft_r2c / k_mdft_c2r for small tables).
//
// What the straight-from-global kernels above cost on 128^3 / 17 kept columns (2.15 GB real tensor,
// profiles/r01_mdft_ablation.txt): r2c 898 us, of which 677 us remain with the MFMAs AND the stores
// taken out -- the "lane = line" operand loads (32 different 128-byte lines per load instruction) and
// the per-wave re-read of the table from L1/L2 with a one-step prefetch are the whole bill; c2r 832 us,
// 697 us of it without MFMAs and stores.  Here a 256-thread block owns 128 consecutive lines
// (one contiguous span of memory), copies it global -> LDS with fully coalesced 16-byte loads, keeps
// the WHOLE constant table in LDS for its lifetime (several tiles), and the waves read MFMA operands
// from LDS (padded rows: conflict-free).  One 32-line row tile per wave and <= 64 accumulator
// registers: 3-4 blocks per CU hide each other's barriers and memory latency.
// ==========================================================================================
SC_HD float sc_f4_at(const sc_f4& v, const int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }

#define SC_MDFT_LB 128          // lines per block tile (4 waves x one 32-line MFMA row tile)
// c2r store patches: 8 rows x 32 floats kept as 4 lines of [row u | row u + 4], line stride 96 floats: the two
// lane halves of a write (rows u, u + 4) fill the two halves of the 64 banks, and the 16 lanes of a 16-byte
// read pass (rows 2k, 2k + 1 of one side) land 96 = 32 (mod 64) banks apart -- conflict free both ways.
// (Plain row strides: 36 -> halves 16 banks apart on the write, 40 -> neighbouring rows 8 banks short on the
// read; 23-31 % of the plane inverse's LDS cycles were bank conflicts.)
#define SC_C2R_PL 96
#define SC_C2R_PATCH_FLOATS (4 * 2 * 4 * SC_C2R_PL)

// ------------------------------------------------------------------------------------------
// last axis, real -> complex through LDS.  N % 32 == 0, N <= 256, CT column tiles (2J floats <= 32 CT,
// minus the tail column when TAIL).  The tile is consumed in chunks of 32 input samples: chunk c + 1
// sits in registers (requested right after the barrier) while chunk c is multiplied out of LDS.
//   tab  [((ct * NG + t) * 64 + lane) * 4 + q] = T[8t + 4(lane>>5) + q][32 ct + (lane & 31)]   (floats)
//   tail [n] = T[n][J - 1]                                                                     (cf32)
// dynamic LDS: the table, NG * CT * 1 KB.
// ------------------------------------------------------------------------------------------
//
// JP > 0 ("plane" form, second-to-last axis of NR = 128, 64 or 32 rows: a tile is 1, 2 or 4 whole planes):
// the tile's 128 x J result is not written out but kept in LDS (Y, over the chunk buffer), and the pruned
// transform along the NR rows of each plane follows at once as ONE real MFMA product per input row n1:
//   Z[j1][(j2, c)] = sum_n1 ( Tr[j1][n1] Y[n1][(j2, c)] + Ti[j1][n1] rot(Y)[n1][(j2, c)] ),
//   rot(Y)[(j2, 0)] = -Im Y[j2], rot(Y)[(j2, 1)] = Re Y[j2]
// i.e. tile rows = 32 kept rows j1, tile columns = the row's floats exactly as they lie in Y, k = (Re T, Im T)
// on the two lane halves (table tab1: [(jt * NR + n1) * 64 + lane] = lane < 32 ? Tr : Ti of row 32 jt + lane % 32).
// With J = 2^k + 1 the 32 CT columns are all real work and the last column is a VALU dot product again; the
// (row, re/im) x column-index tiling of k_mdft_axis spent half its MFMAs on 15 padding columns of 32.
// JP = row tiles (K1 <= 32 JP).  A plane belongs to NR / 32 waves; wave wl of them takes row tile wl % JP and
// the n1 range number wl / JP (its slice of tab1 lives in registers for the whole launch); partial sums of
// the ranges meet in LDS.
// out is then complex (planes, K1, J).  The NR x J intermediate (0.57 GB each way on 128^3) never reaches
// HBM and one launch disappears.
template <int CT, bool TAIL, int JP = 0, int NR = SC_MDFT_LB>
#ifndef SC_PLANE_OCC
#define SC_PLANE_OCC 2
#endif
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (JP ? ((JP == 2 || CT == 2) ? 1 : SC_PLANE_OCC) : (CT == 1 ? 4 : 3)))
k_mdft_r2c_lds(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
               const cf32* __restrict__ tail, int64_t lines, int N, int J, int tiles_per_block,
               const float* __restrict__ tab1, int K1) {
  constexpr int LB = SC_MDFT_LB, KC = 32, S4 = 9;            // LDS row = 36 floats = 9 float4
  constexpr int SY = 32 * CT + 8;                            // Y row stride (floats): holds 2J <= 32 CT + 2
  constexpr int DAT4 = (JP && LB * SY > LB * S4 * 4) ? LB * SY / 4 : LB * S4;
  constexpr int PL = LB / NR, WPP = 4 / PL;                  // planes per tile, waves per plane
  constexpr int JPD = JP ? JP : 1, KP = (WPP / JPD) ? (WPP / JPD) : 1, NS = NR;   // one k-step per input row
  static_assert(!JP || JPD * KP == WPP, "row tiles x n1 ranges = waves of a plane");
  constexpr int REDF = (JP && KP > 1) ? PL * (KP - 1) * JPD * CT * 1024 : 1;   // partial-sum tiles of the row pass
  SC_DYN_SHARED(sc_f4, tabL);
  SC_SHARED sc_f4 dat[DAT4];                                 // chunk buffer; plane form: then the tile result Y
  SC_SHARED sc_f4 tailL[TAIL ? 128 : 1];                     // 256 cf32
  SC_SHARED cf32 tsum[TAIL ? 128 : 1];
  SC_SHARED float red[REDF];
  SC_SHARED cf32 tred[(JP && TAIL) ? 256 : 1];                // tail column: one partial per lane
  float* Y = reinterpret_cast<float*>(dat);
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int NC = N / KC, NG = N / 8;
  const int64_t n_tiles = (lines + LB - 1) / LB;
  const int64_t tile0 = (int64_t)SC_BID_X * tiles_per_block;
  if (tile0 >= n_tiles) return;
  const int my_tiles = (int)((n_tiles - tile0 < tiles_per_block) ? n_tiles - tile0 : tiles_per_block);
  const int total = my_tiles * NC;

  {
    const sc_f4* t4 = reinterpret_cast<const sc_f4*>(tab);
    for (int i = tid; i < NG * CT * 64; i += 256) tabL[i] = t4[i];
    if (TAIL) {
      const sc_f4* s4 = reinterpret_cast<const sc_f4*>(tail);
      for (int i = tid; i < N / 2; i += 256) tailL[i] = s4[i];
    }
  }
  // plane form: this wave's (row tile, n1 range) slice of the row-pass table is the same for every plane
  // it will see -- it lives in registers (one 256-byte L2 read per MFMA with a one-step prefetch left the
  // matrix cores waiting: 883 us for the pair of passes that took 504 + 274 us apart)
  const int pl = w / WPP, wl = w % WPP;
  const int jt = wl % JPD, kp = wl / JPD;
  const int s0 = kp * (NS / KP);
  float t1r[JP ? NS / KP : 1];
  if (JP) {
    const float* tp = tab1 + ((int64_t)jt * NR + s0) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NS / KP; ++i) t1r[i] = tp[i * 64];
  }
  // loader: thread (lrow, lc4) brings 16 bytes of lines lrow + 32 m, m = 0..3, per chunk
  const int lrow = tid >> 3, lc4 = tid & 7;
  int ld_c = 0;
  int64_t ld_l0 = tile0 * LB;
  sc_f4 r[4];
  auto gload = [&]() {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      int64_t line = ld_l0 + lrow + 32 * m;
      if (line >= lines) line = lines - 1;
      r[m] = SC_LOAD_STREAM(reinterpret_cast<const sc_f4*>(in + line * N + ld_c * KC) + lc4);
    }
    if (++ld_c == NC) {
      ld_c = 0;
      ld_l0 += LB;
    }
  };
  gload();

  sc_f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
  cf32 tacc = cf_make(0.f, 0.f);
  int c = 0;
  int64_t l0 = tile0 * LB;
  const int fmax = TAIL ? 2 * J - 2 : 2 * J;
#pragma unroll 1
  for (int g = 0; g < total; ++g) {
    SC_SYNC();                                   // the previous chunk has been read (g = 0: tables are in)
#pragma unroll
    for (int m = 0; m < 4; ++m) dat[(lrow + 32 * m) * S4 + lc4] = r[m];
    SC_SYNC();
    if (g + 1 < total) gload();
    const sc_f4* arow = dat + (32 * w + col) * S4 + half;
    const sc_f4* brow = tabL + (4 * c) * 64 + lane;
    const sc_f4* trow = tailL + (32 * c + 4 * half) / 2;
    // operands of step s + 1 are read from LDS before the MFMAs of step s are issued
    sc_f4 a = arow[0], b[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b[ct] = brow[ct * NG * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sc_f4 an = a, bn[CT], tw0, tw1;
      if (s < 3) an = arow[2 * (s + 1)];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bn[ct] = (s < 3) ? brow[(ct * NG + s + 1) * 64] : b[ct];
      if (TAIL) {
        tw0 = trow[4 * s];
        tw1 = trow[4 * s + 1];
      }
      SC_SCHED_BARRIER();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) MDFT_MFMA(acc[ct], sc_f4_at(a, q), sc_f4_at(b[ct], q));
      if (TAIL) {
        tacc.x = fmaf(a.x, tw0.x, tacc.x); tacc.y = fmaf(a.x, tw0.y, tacc.y);
        tacc.x = fmaf(a.y, tw0.z, tacc.x); tacc.y = fmaf(a.y, tw0.w, tacc.y);
        tacc.x = fmaf(a.z, tw1.x, tacc.x); tacc.y = fmaf(a.z, tw1.y, tacc.y);
        tacc.x = fmaf(a.w, tw1.z, tacc.x); tacc.y = fmaf(a.w, tw1.w, tacc.y);
      }
      a = an;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) b[ct] = bn[ct];
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) SC_PIN_ACC(acc[ct]);
    if (++c == NC) {                             // tile finished: store its 32 x (2J) results per wave
      const int64_t lw = l0 + 32 * w + sc_opaque(0);
      if (JP == 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int f = 32 * ct + col;
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int64_t line = lw + mdft_row(v, half);
            if (f < fmax && line < lines && MDFT_STORE_OK(acc[ct][v])) out[line * 2 * J + f] = acc[ct][v];
            acc[ct][v] = 0.f;
          }
        }
        if (TAIL) {
          if (half == 1) tsum[32 * w + col] = tacc;
          SC_WAVE_SYNC();
          if (half == 0) {
            const int64_t line = lw + col;
            if (line < lines) reinterpret_cast<cf32*>(out)[line * J + (J - 1)] = cf_add(tacc, tsum[32 * w + col]);
          }
          tacc = cf_make(0.f, 0.f);
        }
      } else {
        // ---- the tile's result stays in LDS (over the chunk buffer, once every wave is done with it) ...
        SC_SYNC();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int f = 32 * ct + col;
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            if (f < fmax) Y[(32 * w + mdft_row(v, half)) * SY + f] = acc[ct][v];
            acc[ct][v] = 0.f;
          }
        }
        if (TAIL) {
          if (half == 1) tsum[32 * w + col] = tacc;
          SC_WAVE_SYNC();
          if (half == 0)
            reinterpret_cast<cf32*>(Y)[(32 * w + col) * (SY / 2) + (J - 1)] = cf_add(tacc, tsum[32 * w + col]);
          tacc = cf_make(0.f, 0.f);
        }
        SC_SYNC();                               // Y is complete
        // ---- ... and is transformed along the rows of its planes
        const float sgn = (half == 0) ? 1.f : ((col & 1) ? 1.f : -1.f);
        const float* yb = Y + (half == 0 ? col : (col ^ 1));       // rot(Y) for the Im T half
        const cf32* yt = reinterpret_cast<const cf32*>(Y) + (J - 1);
        sc_f32x16 z[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int v = 0; v < 16; ++v) z[ct][v] = 0.f;
        cf32 tz = cf_make(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NS / KP; ++i) {
          const int n1 = pl * NR + s0 + i;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) MDFT_MFMA(z[ct], t1r[i], sgn * yb[n1 * SY + 32 * ct]);
          if (TAIL) {                                              // (Tr + i Ti) y: this lane holds Tr or Ti
            const cf32 y = yt[n1 * (SY / 2)];
            tz.x = fmaf(t1r[i], half ? -y.y : y.x, tz.x);
            tz.y = fmaf(t1r[i], half ? y.x : y.y, tz.y);
          }
        }
        if (KP > 1) {
          if (kp > 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int v = 0; v < 16; ++v)
                red[((((pl * (KP - 1) + kp - 1) * JPD + jt) * CT + ct) * 16 + v) * 64 + lane] = z[ct][v];
          }
        }
        if (TAIL) tred[64 * w + lane] = tz;
        if (KP > 1 || TAIL) SC_SYNC();
        if (KP > 1 && kp == 0) {
#pragma unroll
          for (int k = 1; k < KP; ++k)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int v = 0; v < 16; ++v)
                z[ct][v] += red[((((pl * (KP - 1) + k - 1) * JPD + jt) * CT + ct) * 16 + v) * 64 + lane];
        }
        const int64_t plane = (l0 / LB) * PL + pl;
        if (kp == 0 && plane * NR < lines) {
          float* zo = out + (plane * K1) * 2 * J;                  // row j1 of the plane = 2J floats
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const int f = 32 * ct + col;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const int j1 = 32 * jt + mdft_row(v, half);
              if (j1 < K1 && f < fmax && MDFT_STORE_OK(z[ct][v])) zo[(int64_t)j1 * 2 * J + f] = z[ct][v];
            }
          }
          if (TAIL && half == 0) {                                 // both halves of all n1 ranges of row j1
            const int j1 = 32 * jt + col;
            cf32 t = cf_make(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < KP; ++k) {
              const int ww = pl * WPP + k * JPD + jt;
              t = cf_add(t, cf_add(tred[64 * ww + col], tred[64 * ww + 32 + col]));
            }
            if (j1 < K1) reinterpret_cast<cf32*>(zo)[(int64_t)j1 * J + (J - 1)] = t;
          }
        }
        // (the barrier that opens the next chunk frees Y; the partial sums are rewritten a whole tile later)
      }
      c = 0;
      l0 += LB;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) SC_PIN_ACC(acc[ct]);      // both paths leave the accumulators where they are
  }
}

// ------------------------------------------------------------------------------------------
// last axis, real -> complex through LDS for ANY width (round 4; k_mdft_r2c_lds above needs N % 32 == 0 and its whole
// table in LDS).  The straight-from-global kernel k_mdft_r2c reads its A operand "lane = line": one load instruction
// touches 32 different rows, every 128-byte line is wanted by 16 instructions, and the rows a CU has in flight (RT x 4 KB
// per wave) do not fit its L1 -- 262 us for the 363 MB of a 16 x 32 x 421^2 tensor (1.4 TB/s).  Here a 256-thread block
// owns 128 consecutive lines and brings them through LDS in chunks of 32 samples with loads whose 32 lanes run along a
// ROW (128 contiguous bytes per half wave, 4-byte accesses: rows of an odd width are only 4-byte aligned), each byte
// exactly once; the waves then read their MFMA operands from LDS (36-float rows: conflict free as float4).  The table
// (too big for LDS next to useful occupancy at such widths: 56 KB at 421 x 17) stays in global memory in the lane-major
// float4 layout of k_mdft_r2c_lds, one chunk of it (4 CT float4 per lane) held in registers, each slot re-requested for
// the next chunk as soon as its MFMAs have read it -- every block reads the same 50-100 KB, which L2 keeps.  Samples past N arrive as zeros (and the table rows are zero there).
//   tab  [((ct * NG + t) * 64 + lane) * 4 + q] = T[8t + 4(lane>>5) + q][32 ct + (lane & 31)],  NG = 4 ceil(N / 32)
//   tail [n] = T[n][J - 1], n < 32 ceil(N / 32) <= 1024 (kept in LDS)
// ------------------------------------------------------------------------------------------
// (Two chunks requested ahead -- 32 registers per thread, raw barriers so that the younger request is not drained --
// was tried: it needs the 168-register budget, i.e. three blocks per CU instead of four, and still spills 64 registers
// per chunk; with two blocks per CU the bytes in flight are back where they were.  Not kept.  Nor were 64-sample chunks
// (one row = 256 contiguous bytes per load instruction, half the shared lines and barriers, 35 KB of LDS, 32 staging
// registers, three blocks per CU): 118 vs 112 us at 421 x 17, 40 vs 32 us at 141, 18 vs 14 us at 85.  Nor a form that
// asks for LINES instead of samples (the 128 rows of a tile are one line-aligned span: per step every row brings its next
// whole 128-byte line with aligned 16-byte loads into a two-slot ring per row, each line once, and the waves read their
// samples at the row's own phase): bit-identical, 1.07 x instead of 1.28 x the bytes -- and 127 vs 110 us at 421 x 17,
// 42 vs 32 us at 141: sixteen 4-byte LDS reads at computed addresses per step and one parking trip per tile cost more
// than the second fetch of the shared half-lines.  profiles/r04_mdft_odd_ablation.txt (f), (g).)
#ifndef SC_MDFT_STAGE2_OCC
#define SC_MDFT_STAGE2_OCC 2         // blocks per CU of the two-tile instantiations (A-B: 3 = round 4, 15-23 spilled registers)
#endif
template <int CT, bool TAIL>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (CT == 1 ? 4 : SC_MDFT_STAGE2_OCC))
k_mdft_r2c_stage(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
                 const cf32* __restrict__ tail, int64_t lines, int N, int J, int tiles_per_block) {
  constexpr int LB = SC_MDFT_LB, KC = 32, S4 = 9;            // LDS row = 36 floats = 9 float4
  SC_SHARED sc_f4 dat[LB * S4];
  SC_SHARED sc_f4 tailL[TAIL ? 512 : 1];                     // 1024 cf32
  SC_SHARED cf32 tsum[TAIL ? 128 : 1];
  float* datf = reinterpret_cast<float*>(dat);
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int NC = (N + KC - 1) / KC, NG = 4 * NC;
  const int64_t n_tiles = (lines + LB - 1) / LB;
  const int64_t tile0 = (int64_t)SC_BID_X * tiles_per_block;
  if (tile0 >= n_tiles) return;
  const int my_tiles = (int)((n_tiles - tile0 < tiles_per_block) ? n_tiles - tile0 : tiles_per_block);
  const int total = my_tiles * NC;
  if (TAIL) {
    const sc_f4* s4 = reinterpret_cast<const sc_f4*>(tail);
    for (int i = tid; i < NC * (KC / 2); i += 256) tailL[i] = s4[i];
  }
  // loader: lane col = sample of the chunk; the two halves of a wave take rows 8 apart (36-float rows: the two
  // 32-bank runs of a write then fill the 64 banks), m walks the tile: row = 16 (m >> 1) + 8 half + 2 w + (m & 1).
  const int lr0 = 8 * half + 2 * w;
  int ld_c = 0;
  int64_t ld_l0 = tile0 * LB;
  float r[16];
  auto gload = [&]() {
    const int n = ld_c * KC + col;
    const bool nok = n < N;
    const int nn = nok ? n : N - 1;                            // the load itself is unconditional, inside the line
    // a wave-uniform tile pointer + a 32-bit byte offset per lane (saddr form: sixteen 64-bit row pointers spilled);
    // the offset is clamped to the tile's last float, so rows past the end of the tensor re-read valid memory
    const char* tile = reinterpret_cast<const char*>(in + ld_l0 * N);
    const int64_t left = lines - ld_l0;
    const uint32_t omax = (uint32_t)(((left < LB ? (int)left : LB) * N - 1) * 4);
    const uint32_t o0 = (uint32_t)((lr0 * N + nn) * 4);
    const int n4 = sc_opaque_s(4 * N);                         // (the row terms stay scalar: added where they are used)
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      uint32_t o = o0 + (uint32_t)((16 * (m >> 1) + (m & 1)) * n4);
      o = o < omax ? o : omax;
      // plain, not streaming: a row's 32-sample chunk straddles two 128-byte lines whenever rows are not line-aligned,
      // and the neighbouring chunk wants the shared line one iteration later -- from L2, if this load left it there
      // (421 x 17: 111 us plain, 124 us non-temporal; profiles/r04_mdft_odd_ablation.txt)
#ifdef SC_STAGE_NT_LOAD
      const float v = SC_LOAD_STREAM(reinterpret_cast<const float*>(tile + o));
#else
      const float v = *reinterpret_cast<const float*>(tile + o);
#endif
      r[m] = nok ? v : 0.f;
    }
    if (++ld_c == NC) {
      ld_c = 0;
      ld_l0 += LB;
    }
  };
  // table: slot s of the NEXT chunk is requested as soon as the MFMAs of step s of this chunk have read theirs
  const sc_f4* t4 = reinterpret_cast<const sc_f4*>(tab);
  sc_f4 tc[4][CT];
  auto tload = [&](const int cc, const int s) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) tc[s][ct] = t4[(ct * NG + 4 * cc + s) * 64 + lane];
  };
  gload();
#pragma unroll
  for (int s = 0; s < 4; ++s) tload(0, s);

  sc_f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
  cf32 tacc = cf_make(0.f, 0.f);
  int c = 0;
  int64_t l0 = tile0 * LB;
  const int fmax = TAIL ? 2 * J - 2 : 2 * J;
#pragma unroll 1
  for (int g = 0; g < total; ++g) {
    SC_SYNC();                                   // the previous chunk has been read (g = 0: the tail column is in)
#pragma unroll
    for (int m = 0; m < 16; ++m) datf[(16 * (m >> 1) + lr0 + (m & 1)) * (4 * S4) + col] = r[m];
    SC_SYNC();
    if (g + 1 < total) gload();
    const int cn = (c + 1 == NC) ? 0 : c + 1;
    const sc_f4* arow = dat + (32 * w + col) * S4 + half;
    const sc_f4* trow = tailL + (32 * c + 4 * half) / 2;
    sc_f4 a = arow[0];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sc_f4 an = a, tw0, tw1;
      if (s < 3) an = arow[2 * (s + 1)];
      if (TAIL) {
        tw0 = trow[4 * s];
        tw1 = trow[4 * s + 1];
      }
      SC_SCHED_BARRIER();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) MDFT_MFMA(acc[ct], sc_f4_at(a, q), sc_f4_at(tc[s][ct], q));
      if (TAIL) {
        tacc.x = fmaf(a.x, tw0.x, tacc.x); tacc.y = fmaf(a.x, tw0.y, tacc.y);
        tacc.x = fmaf(a.y, tw0.z, tacc.x); tacc.y = fmaf(a.y, tw0.w, tacc.y);
        tacc.x = fmaf(a.z, tw1.x, tacc.x); tacc.y = fmaf(a.z, tw1.y, tacc.y);
        tacc.x = fmaf(a.w, tw1.z, tacc.x); tacc.y = fmaf(a.w, tw1.w, tacc.y);
      }
      SC_SCHED_BARRIER();
      tload(cn, s);
      a = an;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) SC_PIN_ACC(acc[ct]);
    if (++c == NC) {                             // tile finished: store its 32 x (2J) results per wave
      const int64_t lw = l0 + 32 * w + sc_opaque(0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int f = 32 * ct + col;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int64_t line = lw + mdft_row(v, half);
          if (f < fmax && line < lines && MDFT_STORE_OK(acc[ct][v])) out[line * 2 * J + f] = acc[ct][v];
          acc[ct][v] = 0.f;
        }
      }
      if (TAIL) {
        if (half == 1) tsum[32 * w + col] = tacc;
        SC_WAVE_SYNC();
        if (half == 0) {
          const int64_t line = lw + col;
          if (line < lines) reinterpret_cast<cf32*>(out)[line * J + (J - 1)] = cf_add(tacc, tsum[32 * w + col]);
        }
        tacc = cf_make(0.f, 0.f);
      }
      c = 0;
      l0 += LB;
    }
  }
}

// (Measured and removed, round 4: the mirror image of k_mdft_c2r_span for this direction -- one block per 32 lines, their
// N-line span copied into LDS by LDS-DMA (16 bytes per lane, every piece requested before the first wait), the four
// waves splitting the SAMPLES and meeting in LDS.  Bit-identical in emulation and on the device, and slower everywhere:
// 145 vs 112 us at 421 x 17 (back to back and inside the layer step alike), 48 vs 33 us at 141, 23 vs 15 us at 85 --
// a block cannot overlap its one big load with its own multiplies, operands come out of LDS 4 bytes at a time, and one
// wave of four does the reduction.  profiles/r04_mdft_odd_ablation.txt (d).)

// ------------------------------------------------------------------------------------------
// last axis, complex -> real for ANY width with the table in global memory (round 4; k_mdft_c2r_lds below needs its
// whole table in LDS -- 64 KB at 421 x 17 -- and N % 4 == 0).  What k_mdft_c2r costs on a 16 x 32 x 421^2 tensor
// (363 MB written, 29 MB read; profiles/r04_mdft_odd_ablation.txt): 247 us, 153 us of it with the MFMAs AND the stores
// taken out -- the "lane = line" spectrum loads (64 different 128-byte lines per instruction, first touch from HBM, the
// same rows again for every group of 8 column tiles) and a one-step table prefetch are the bill, not the 363 MB.
// Here a 256-thread block owns 128 consecutive lines x a range of column tiles: the tile's spectrum (128 x J cf32,
// one contiguous span) comes in ONCE with coalesced 8-byte loads and is parked in LDS (rows of S floats, S / 2 odd:
// conflict-free operand reads), every lane then keeps its line's 2 JS2 operand pairs in registers for all its column
// tiles, and the loop over column tiles only streams the table: JS2 float4 per lane and tile, lane-major, each slot
// re-requested for the next tile as soon as its MFMAs have read it (all blocks read the same table: L2).
//   tab [((nt * JS2 + p) * 64 + lane) * 4 + e] : step t = 2p + (e >> 1) of column tile nt, component e & 1
//        (0 multiplies Re in[l][2t + (lane >> 5)], 1 Im), zero for 2t + (lane >> 5) >= J or 32 nt + lane % 32 >= N
// The bias is looked up per line (biasL: lines of one tile may belong to two images when heights are odd).
// Grid: x = tile, y = range of column tiles [y * nt_per, (y + 1) * nt_per): enough blocks to fill the chip several
// times over when there are few tiles.  Dynamic LDS: 128 * S floats.
// ------------------------------------------------------------------------------------------
template <int JS2>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (JS2 <= 5 ? 4 : 3))
k_mdft_c2r_stage(const cf32* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
                 const float* __restrict__ bias, int64_t lines, int N, int J, int n_nt, int S,
                 int64_t lines_per_image, int64_t channels, int nt_per) {
  constexpr int LB = SC_MDFT_LB;
  SC_DYN_SHARED(float, tileL);
  SC_SHARED float biasL[LB];
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t l0 = (int64_t)SC_BID_X * LB;
  const int nt_lo = (int)SC_BID_Y * nt_per;
  const int nt_hi = (nt_lo + nt_per < n_nt) ? nt_lo + nt_per : n_nt;
  if (l0 >= lines || nt_lo >= n_nt) return;
  const int64_t rem = lines - l0;
  const int rows = (int)(rem < LB ? rem : LB);
  {
    // element tid, tid + 256, ... of the tile's span (128 J <= 256 * 2 JS2 of them): all requested, then parked;
    // (line, j) advance by (256 / J, 256 % J)
    const int E = rows * J, dq = 256 / J, dr = 256 - dq * J;
    const cf32* src = in + l0 * J;
    cf32 pre[2 * JS2];
#pragma unroll
    for (int u = 0; u < 2 * JS2; ++u) {
      const int i = tid + 256 * u;
      pre[u] = src[i < E ? i : E - 1];                        // (plain: the other column ranges of this tile read it from L2)
    }
    int line = tid / J, j = tid - line * J;
    cf32* datc = reinterpret_cast<cf32*>(tileL);
    const int SC2 = S / 2;
#pragma unroll
    for (int u = 0; u < 2 * JS2; ++u) {
      if (tid + 256 * u < E) datc[line * SC2 + j] = pre[u];
      j += dr;
      line += dq;
      if (j >= J) {
        j -= J;
        ++line;
      }
    }
    if (tid < LB) {
      const int64_t line_g = l0 + (tid < rows ? tid : rows - 1);
      biasL[tid] = (bias != nullptr) ? bias[(line_g / lines_per_image) % channels] : 0.f;
    }
  }
  SC_SYNC();
  // this lane's operands: in[line][2t + half], t < 2 JS2 (zero past J -- the table is zero there too, but the LDS is not)
  cf32 a[2 * JS2];
  {
    const int row = 32 * w + col;
    const cf32* arow = reinterpret_cast<const cf32*>(tileL) + (row < rows ? row : rows - 1) * (S / 2);
#pragma unroll
    for (int t = 0; t < 2 * JS2; ++t) {
      const int j = 2 * t + half;
      const cf32 v = arow[j < J ? j : J - 1];
      a[t] = j < J ? v : cf_make(0.f, 0.f);
    }
  }
  const sc_f4* t4 = reinterpret_cast<const sc_f4*>(tab);
  sc_f4 tb[JS2];
#pragma unroll
  for (int p = 0; p < JS2; ++p) tb[p] = t4[(nt_lo * JS2 + p) * 64 + lane];
  const int64_t lw = l0 + 32 * w;                            // (rows of a ragged last tile: guarded at the store)
#pragma unroll 1
  for (int nt = nt_lo; nt < nt_hi; ++nt) {
    sc_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int ntn = (nt + 1 < nt_hi) ? nt + 1 : nt;
#pragma unroll
    for (int p = 0; p < JS2; ++p) {
      MDFT_MFMA(acc, a[2 * p].x, tb[p].x);
      MDFT_MFMA(acc, a[2 * p].y, tb[p].y);
      MDFT_MFMA(acc, a[2 * p + 1].x, tb[p].z);
      MDFT_MFMA(acc, a[2 * p + 1].y, tb[p].w);
      SC_SCHED_BARRIER();
      tb[p] = t4[(ntn * JS2 + p) * 64 + lane];
    }
    SC_PIN_ACC(acc);
    // stores: a wave-uniform pointer to the wave's 32 rows + a 32-bit byte offset per lane (sixteen 64-bit row
    // addresses would be hoisted out of the loop and spilled)
    const int n = 32 * nt + col;
    char* obase = reinterpret_cast<char*>(out + lw * N);
    const uint32_t so0 = (uint32_t)((4 * half * N + n) * 4);
    const int n4 = sc_opaque_s(4 * N);
    const int rleft = rows - 32 * w - 4 * half;              // rows of this wave's tile that exist, from row 4 half on
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int rv = (v & 3) + 8 * (v >> 2);                  // mdft_row(v, half) = rv + 4 half
      if (rv < rleft && n < N && MDFT_STORE_OK(acc[v])) {
        float* dst = reinterpret_cast<float*>(obase + so0 + (uint32_t)(rv * n4));
        // plain, not streaming: a store instruction covers 128 bytes of two rows that are not line-aligned; the rest
        // of each line arrives with the next column tile, and L2 merges the halves only if the first was not sent
        // on as a partial write (421 x 17: 191 us plain, 283 us non-temporal)
#ifdef SC_STAGE_NT_STORE
        SC_STORE_STREAM(dst, acc[v] + biasL[32 * w + 4 * half + rv]);
#else
        *dst = acc[v] + biasL[32 * w + 4 * half + rv];
#endif
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// last axis, complex -> real, ANY width, whole-line stores (round 4, second step).  k_mdft_c2r_stage above is bound by
// its stores: a store instruction of an MFMA result tile writes 128-byte pieces of two rows, and rows of an odd width
// are not line-aligned -- every line is written in two partial pieces one column tile apart (2.3 TB/s).  But 32
// consecutive lines of N floats are ONE span of exactly N 128-byte lines, and it starts line-aligned (32 N x 4 bytes
// per 32 lines).  Here a 256-thread block owns 32 lines: its four waves share the 32 x J spectrum rows (operands in
// registers, as above) and split the column tiles (nt = w, w + 4, ...), results + bias go to an LDS image of the span
// (row stride N, as in memory), and after one barrier the block writes the span with aligned, fully coalesced
// 16-byte streaming stores, each line whole.  Three blocks are resident per CU at N = 421 (54 KB of LDS each): while
// one drains its stores the others load and multiply.
//   tab: the layout of k_mdft_c2r_stage.  Dynamic LDS: 32 max(N, S) floats (spectrum rows, then the span).
// Host: N <= SC_C2R_SPAN_NMAX (LDS), `out` 16-byte aligned; otherwise k_mdft_c2r_stage.
// ------------------------------------------------------------------------------------------
template <int JS2>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 3)                  // three blocks per CU (LDS at N = 421): <= 168 registers
k_mdft_c2r_span(const cf32* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
                const float* __restrict__ bias, int64_t lines, int N, int J, int n_nt, int S,
                int64_t lines_per_image, int64_t channels) {
  constexpr int RB = 32;                                     // lines per block = one MFMA row tile
  SC_DYN_SHARED(float, ldsf);
  SC_SHARED float biasL[RB];
  // the spectrum rows [RB][S] and the span image [RB][N] share the buffer: the rows are dead once every lane holds its
  // operands (one more barrier), and 54 KB instead of 58 KB is three blocks per CU instead of two at N = 421 -- a
  // block's LDS is only released when its stores have drained, so the third block is what multiplies meanwhile
  float* tileL = ldsf;
  float* span = ldsf;
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int64_t l0 = (int64_t)SC_BID_X * RB;
  if (l0 >= lines) return;
  const int64_t rem = lines - l0;
  const int rows = (int)(rem < RB ? rem : RB);
  {
    // element tid, tid + 256, ... of the block's spectrum rows (32 J <= 256 * UM of them): all requested, then parked
    constexpr int UM = (4 * JS2 + 7) / 8;
    const int E = rows * J, dq = 256 / J, dr = 256 - dq * J;
    const cf32* src = in + l0 * J;
    cf32 pre[UM];
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      const int i = tid + 256 * u;
      pre[u] = src[i < E ? i : E - 1];
    }
    int line = tid / J, j = tid - line * J;
    cf32* datc = reinterpret_cast<cf32*>(tileL);
    const int SC2 = S / 2;
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      if (tid + 256 * u < E) datc[line * SC2 + j] = pre[u];
      j += dr;
      line += dq;
      if (j >= J) {
        j -= J;
        ++line;
      }
    }
    if (tid < RB) {
      const int64_t line_g = l0 + (tid < rows ? tid : rows - 1);
      biasL[tid] = (bias != nullptr) ? bias[(line_g / lines_per_image) % channels] : 0.f;
    }
  }
  SC_SYNC();
  // this lane's operands: in[line][2t + half], t < 2 JS2 (zero past J: the table is zero there too, but the LDS is not)
  cf32 a[2 * JS2];
  {
    const cf32* arow = reinterpret_cast<const cf32*>(tileL) + (col < rows ? col : rows - 1) * (S / 2);
#pragma unroll
    for (int t = 0; t < 2 * JS2; ++t) {
      const int j = 2 * t + half;
      const cf32 v = arow[j < J ? j : J - 1];
      a[t] = j < J ? v : cf_make(0.f, 0.f);
    }
  }
  // (requesting the table two column tiles ahead, the first two before the spectrum rows, was measured: 147 vs 133 us
  // at 421 x 17, 34 vs 31 us at 141 -- not kept)
  const sc_f4* t4 = reinterpret_cast<const sc_f4*>(tab);
  sc_f4 tb[JS2];
  {
    const int nt0 = w < n_nt ? w : n_nt - 1;
#pragma unroll
    for (int p = 0; p < JS2; ++p) tb[p] = t4[(nt0 * JS2 + p) * 64 + lane];
  }
  float bl[16];                                              // bias of this lane's 16 result rows
#pragma unroll
  for (int v = 0; v < 16; ++v) bl[v] = biasL[(v & 3) + 8 * (v >> 2) + 4 * half];
  SC_SYNC();                                                 // every wave has its operands: the buffer becomes the span
#pragma unroll 1
  for (int nt = w; nt < n_nt; nt += 4) {
    sc_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const int ntn = (nt + 4 < n_nt) ? nt + 4 : nt;
#pragma unroll
    for (int p = 0; p < JS2; ++p) {
      MDFT_MFMA(acc, a[2 * p].x, tb[p].x);
      MDFT_MFMA(acc, a[2 * p].y, tb[p].y);
      MDFT_MFMA(acc, a[2 * p + 1].x, tb[p].z);
      MDFT_MFMA(acc, a[2 * p + 1].y, tb[p].w);
      SC_SCHED_BARRIER();
      tb[p] = t4[(ntn * JS2 + p) * 64 + lane];
    }
    SC_PIN_ACC(acc);
    const int n = 32 * nt + col;
    float* sp = span + 4 * half * N + n;
    const int n1 = sc_opaque_s(N);                             // (row offsets added where they are used)
    if (n < N) {
#pragma unroll
      for (int v = 0; v < 16; ++v) sp[((v & 3) + 8 * (v >> 2)) * n1] = acc[v] + bl[v];
    }
  }
  SC_SYNC();
  // the span: rows x N floats from out + l0 N (line-aligned), 16 bytes per lane, each 128-byte line whole
  const int total = rows * N, n4 = total >> 2;
  float* dst = out + l0 * N;
  const sc_f4* s4 = reinterpret_cast<const sc_f4*>(span);
  sc_f4* d4 = reinterpret_cast<sc_f4*>(dst);
  for (int i = tid; i < n4; i += 256) {
    const sc_f4 o = s4[i];
    if (MDFT_STORE_OK(o.x)) SC_STORE_STREAM(d4 + i, o);
  }
  const int done = n4 << 2;                                  // ragged last block: rows x N need not be a multiple of 4
  if (tid < total - done) dst[done + tid] = span[done + tid];
}

// ------------------------------------------------------------------------------------------
// last axis, complex -> real through LDS.  One tile = 128 lines x J cf32 (contiguous in memory) copied
// to rows of S floats (S/2 odd: the 8-byte operand reads of 32 lines hit 32 different bank pairs).
//   tab [(((nt * JS + t) * 64 + lane) * 2 + comp] : comp 0 multiplies Re(in[l][2t + (lane>>5)]), comp 1 Im
// dynamic LDS: the table (n_nt * JS * 128 floats), the tile (128 * S floats), 4 x 2 store patches.
// A wave's 32 lines must share one bias value (lines_per_image % 32 == 0 when bias != nullptr).
// ------------------------------------------------------------------------------------------
//
// NR > 0 ("plane" form, second-to-last axis of NR = 128, 64 or 32 rows: a tile is 1, 2 or 4 whole planes):
// `in` is the spectrum BEFORE that axis' zero-padded inverse pass, complex (planes, K1, J).  The tile's
// planes (K1 x J each, a few KB) are copied to LDS, expanded to their NR rows by the data x table MFMA
// product of k_mdft_axis (table tabA in that kernel's layout: NR / 16 row tiles per plane, two per wave)
// straight into the tile buffer, and the last-axis pass follows as before: the NR x J intermediate never
// reaches HBM and one launch disappears.
// NPF > 0: the NEXT tile's input (its spectrum rows / its planes) is requested into NPF registers per thread
// before the current tile is multiplied, so its HBM latency hides behind the MFMAs and stores instead of
// opening every tile (the kernel sat at 36 % matrix-core busy with 46 % of its wave cycles waiting).
// Plane form row pass (CA column tiles of 32 floats, ATAIL: column J - 1 = 2^k + 1-th on the VALU) -- like the
// forward one, ONE real MFMA product per kept row j1 with tile rows = 32 output rows n1 (one tile per wave),
// tile columns = the floats of Z's row as they lie, k = (Re T, Im T) on the two lane halves; table tabA
// [(rt * K1 + j1) * 64 + lane] = lane < 32 ? Tr : Ti of T[32 rt + lane % 32][j1], this wave's slice in registers.
template <int CT, int NR = 0, int NPF = 0, int CA = 1, bool ATAIL = false>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (NR == 128 ? 2 : (NR > 0 ? (CA == 2 ? 2 : 3) : (CT == 4 ? 3 : 4))))
k_mdft_c2r_lds(const cf32* __restrict__ in, float* __restrict__ out, const float* __restrict__ tab,
               const float* __restrict__ bias, int64_t lines, int N, int J, int n_nt, int S,
               int64_t lines_per_image, int64_t channels, int tiles_per_block,
               const float* __restrict__ tabA, int K1) {
  constexpr int LB = SC_MDFT_LB;
  constexpr bool PLANE = NR > 0;
  constexpr int NRD = PLANE ? NR : LB, PL = LB / NRD, WPP = 4 / PL;   // planes per tile, waves per plane
  SC_DYN_SHARED(sc_f4, lds);
  const int tid = SC_TID, lane = tid & 63, half = lane >> 5, col = lane & 31;
  const int w = SC_UNIFORM(tid >> 6);
  const int JS = (J + 1) / 2, SC2 = S / 2;
  const int tab4 = n_nt * JS * 32;                               // table size in float4
  const cf32* tabL = reinterpret_cast<const cf32*>(lds);
  cf32* dat = reinterpret_cast<cf32*>(lds + tab4);
  float* stg = reinterpret_cast<float*>(lds + tab4) + LB * S;   // 4 waves x 2 patches (SC_C2R_PATCH_FLOATS)
  cf32* Zs = reinterpret_cast<cf32*>(stg + SC_C2R_PATCH_FLOATS);   // plane form: PL x K1 x J
  SC_SHARED cf32 tredA[(PLANE && ATAIL) ? 256 : 1];
  // plane form: the wave's slice of the row-pass table (its 32 output rows x all kept rows) stays in registers
  float tar[PLANE ? NRD / 2 : 1];
  if (PLANE) {
    const float* ta = tabA + ((int64_t)(w % WPP) * K1) * 64 + lane;
#pragma unroll
    for (int i = 0; i < NRD / 2; ++i) tar[i] = (i < K1) ? ta[i * 64] : 0.f;
  }
  const int64_t n_tiles = (lines + LB - 1) / LB;
  const int64_t tile0 = (int64_t)SC_BID_X * tiles_per_block;
  if (tile0 >= n_tiles) return;
  const int my_tiles = (int)((n_tiles - tile0 < tiles_per_block) ? n_tiles - tile0 : tiles_per_block);
  {
    const sc_f4* t4 = reinterpret_cast<const sc_f4*>(tab);
    for (int i = tid; i < tab4; i += 256) lds[i] = t4[i];
  }
  // copy coordinates of element tid, tid + 256, ... of a tile: (line, j) advance by (256 / J, 256 % J)
  const int dq = 256 / J, dr = 256 - dq * J;
  const int line_first = tid / J, j_first = tid - line_first * J;
  const int E = LB * J;
  cf32 pre[NPF ? NPF : 1];
  auto prefetch = [&](const int ti) {
    const int64_t l0 = (tile0 + ti) * LB;
    if (!PLANE) {
      const int64_t rem = lines - l0;
      const int Ev = (int)((rem < LB ? rem : LB) * J);
      const cf32* src = in + l0 * J;
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int idx = tid + 256 * u;
        pre[u] = src[idx < Ev ? idx : Ev - 1];
      }
    } else {
      const cf32* zsrc = in + (l0 / NRD) * K1 * J;
      const int64_t left = (lines - l0) / NRD;
      const int zvalid = (int)(left < PL ? left : PL) * K1 * J;
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int i = tid + 256 * u;
        pre[u] = zsrc[i < zvalid ? i : zvalid - 1];
      }
    }
  };
  if (NPF) prefetch(0);
#pragma unroll 1
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int64_t l0 = (tile0 + ti) * LB;
    const int64_t rem = lines - l0;
    const int Ev = (int)((rem < LB ? rem : LB) * J);             // valid elements (ragged last tile)
    const cf32* src = in + l0 * J;
    SC_SYNC();                                                   // previous tile consumed / table staged
    if (!PLANE && NPF) {
      int line = line_first, j = j_first;
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        if (tid + 256 * u < E) dat[line * SC2 + j] = pre[u];
        j += dr;
        line += dq;
        if (j >= J) {
          j -= J;
          ++line;
        }
      }
    } else if (!PLANE) {
      int line = line_first, j = j_first;
#pragma unroll 1
      for (int base = tid; base < E; base += 1024) {
        cf32 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + 256 * u;
          if (idx >= Ev) idx = Ev - 1;                           // rows past the end repeat finite data
          v[u] = src[idx];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (base + 256 * u < E) dat[line * SC2 + j] = v[u];
          j += dr;
          line += dq;
          if (j >= J) {
            j -= J;
            ++line;
          }
        }
      }
    } else {
      if (NPF) {
#pragma unroll
        for (int u = 0; u < NPF; ++u)
          if (tid + 256 * u < PL * K1 * J) Zs[tid + 256 * u] = pre[u];
      } else {
        const cf32* zsrc = in + (l0 / NRD) * K1 * J;
        const int64_t left = (lines - l0) / NRD;                 // planes from this tile to the end
        const int zvalid = (int)(left < PL ? left : PL) * K1 * J;
        for (int i = tid; i < PL * K1 * J; i += 256) Zs[i] = zsrc[i < zvalid ? i : zvalid - 1];
      }
      SC_SYNC();
      if (NPF && ti + 1 < my_tiles) prefetch(ti + 1);
      // rows 32 wl .. 32 wl + 31 of plane pl:  Y[n1][(j2, c)] = sum_j1 ( Tr Z[j1][(j2, c)] + Ti rot(Z)[j1][(j2, c)] )
      const int pl = w / WPP, wl = w % WPP;
      const float sgn = (half == 0) ? 1.f : ((col & 1) ? 1.f : -1.f);
      const float* zb = reinterpret_cast<const float*>(Zs) + pl * K1 * 2 * J + (half == 0 ? col : (col ^ 1));
      const cf32* zt = Zs + pl * K1 * J + (J - 1);
      sc_f32x16 ya[CA];
#pragma unroll
      for (int ca = 0; ca < CA; ++ca)
#pragma unroll
        for (int v = 0; v < 16; ++v) ya[ca][v] = 0.f;
      cf32 ty = cf_make(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < NRD / 2; ++i) {
        if (i < K1) {                                            // uniform
#pragma unroll
          for (int ca = 0; ca < CA; ++ca) MDFT_MFMA(ya[ca], tar[i], sgn * zb[i * 2 * J + 32 * ca]);
          if (ATAIL) {
            const cf32 y = zt[i * J];
            ty.x = fmaf(tar[i], half ? -y.y : y.x, ty.x);
            ty.y = fmaf(tar[i], half ? y.x : y.y, ty.y);
          }
        }
      }
      float* datf = reinterpret_cast<float*>(dat);
      const int amax = ATAIL ? 2 * J - 2 : 2 * J;
#pragma unroll
      for (int ca = 0; ca < CA; ++ca) {
        const int f = 32 * ca + col;
        if (f < amax) {
#pragma unroll
          for (int v = 0; v < 16; ++v) datf[(pl * NRD + 32 * wl + mdft_row(v, half)) * S + f] = ya[ca][v];
        }
      }
      if (ATAIL) {
        tredA[64 * w + lane] = ty;
        SC_WAVE_SYNC();
        if (half == 0)
          dat[(pl * NRD + 32 * wl + col) * SC2 + (J - 1)] = cf_add(ty, tredA[64 * w + 32 + col]);
      }
    }
    SC_SYNC();
    if (!PLANE && NPF && ti + 1 < my_tiles) prefetch(ti + 1);
    const int64_t lw = l0 + 32 * w;
    if (lw >= lines) continue;                                   // wave-uniform; barriers stay matched below
    const float badd = (bias != nullptr) ? bias[(lw / lines_per_image) % channels] : 0.f;
    const cf32* drow = dat + (32 * w + col) * SC2;
#pragma unroll 1
    for (int nt0 = 0; nt0 < n_nt; nt0 += CT) {
      sc_f32x16 acc[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
      const cf32* tp[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int nt = (nt0 + c < n_nt) ? nt0 + c : n_nt - 1;
        tp[c] = tabL + (int64_t)nt * JS * 64 + lane;
      }
#pragma unroll 2
      for (int t = 0; t < JS; ++t) {
        int j = 2 * t + half;
        if (j >= J) j = J - 1;                                   // table entry is zero there
        const cf32 d = drow[j];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const cf32 b = tp[c][t * 64];
          MDFT_MFMA(acc[c], d.x, b.x);
          MDFT_MFMA(acc[c], d.y, b.y);
        }
      }
      const int64_t lwe = lw + sc_opaque(0);
#ifdef SC_MDFT_C2R_DIRECT_STORE
      // first version: 64 dword stores per wave tile (2 x 128 bytes each) -- 500 us of store issue alone
      // on the 2.15 GB tensor, not overlapping the MFMA phase (profiles/r01_mdft_ablation.txt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t line = lwe + mdft_row(v, half);
        if (line < lines) {
          float* orow = out + line * N;
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const int n = 32 * (nt0 + c) + col;
            if (nt0 + c < n_nt && n < N && MDFT_STORE_OK(acc[c][v])) SC_STORE_STREAM(&orow[n], acc[c][v] + badd);
          }
        }
      }
#else
      // accumulator registers 4g..4g+3 of the two lane halves are 8 consecutive lines x 32 floats: through a
      // per-wave LDS patch they leave as ONE 16-byte-per-lane store (8 lines x 128 bytes) instead of four
      // dword stores -- a quarter of the store instructions.  N % 4 == 0 (host checks).
      float* patch = stg + w * (2 * 4 * SC_C2R_PL);
      const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (nt0 + c >= n_nt) continue;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float* pb = patch + ((c * 4 + gq) & 1) * (4 * SC_C2R_PL);
#pragma unroll
          for (int u = 0; u < 4; ++u) pb[u * SC_C2R_PL + 32 * half + col] = acc[c][4 * gq + u] + badd;
          SC_WAVE_SYNC();
          const sc_f4 o = *reinterpret_cast<const sc_f4*>(pb + (prow & 3) * SC_C2R_PL + 32 * (prow >> 2) + 4 * pc4);
          const int64_t line = lwe + 8 * gq + prow;
          const int n = 32 * (nt0 + c) + 4 * pc4;
          if (line < lines && n < N && MDFT_STORE_OK(o.x))
            SC_STORE_STREAM(reinterpret_cast<sc_f4*>(out + line * N + n), o);
        }
      }
      SC_WAVE_SYNC();                 // the two patches are free again before the next column group writes
#endif
    }
  }
}
