// sc_kernels_plane.h -- factorised transforms of the LAST TWO axes for 128 x 128 planes, one launch each way.
//
// FNO3d 128^3 (BASELINE configs[3]) and 2-D 128 x 128 grids: a plane is 64 KB of real data and its kept block
// (<= 32 rows x 17 columns) 4 KB, so -- unlike the large grids of sc_kernels_fft2p.h -- both axes fit one
// workgroup's LDS and the plane crosses HBM once.  The direct-DFT plane kernels of sc_kernels_mdft.h that served
// these shapes spend 0.65-0.8 ms per launch at B x C = 256 volumes (2.15 GB: 2.5-3.0 TB/s) on the matrix pipe; here
// every line is a Cooley-Tukey factorisation on the vector ALUs, as in the other FFT kernels of the engine:
//
//   rows     128 real points x 2 rows packed as one complex line, 16 lanes x 8 points (n = t + 16 j):
//              radix-8 over j in registers, twiddle w128^(t k1), exchange, 16-point DFT over t by lane k1 (< 8),
//              of which only k = k1 + 8 k2, |k| <= 16 is used; Z[k], Z[-k] -> A[k], B[k] -> tile T[row][k]
//   columns  128 complex points, 8 lanes x 16 points (n = t + 8 j): radix-16 over j, twiddle, exchange, 8-point DFT
//              over t (two k1 per lane) of which k = k1 + 16 k2, k2 in {-1, 0} are the 32 centred rows
//   forward  k_pl128_fwd   x[plane][128][128] real -> out[plane][K0][J]   (rfft2 restricted to the kept block)
//   inverse  k_pl128_inv   in[plane][K0][J]        -> y[plane][128][128] real (+ bias): the exact transpose
// For 3-D data the first axis stays a size-agnostic axis pass over the (small) plane results.
// Reference lines: spectral_convolution.py:443-449, 500-519 (forward), :531-568 (inverse).
#pragma once
#include "sc_kernels_fft2p.h"

// forward row loads: 16-byte accesses staged through the wave's exchange area (session 2; -DSC_PL_DIRECT_LOADS = the 4-byte
// loads in the lanes' own order of round 3: 446-454 us against 404-421 us at FNO3d 128^3, bit-identical results,
// profiles/r03s2_pl128_staged_loads_ab.txt)
#if !defined(SC_PL_DIRECT_LOADS) && !defined(SC_PL_STAGED_LOADS)
#define SC_PL_STAGED_LOADS 1
#endif
#ifndef SC_PL_PPW_DEFAULT
#define SC_PL_PPW_DEFAULT 1   // planes per workgroup of the forward kernel (sc_engine.cpp; environment SC_PL_PPW overrides)
#endif
#define SC_PL_N 128
#define SC_PL_RS 20          // tile row stride (complex): column-phase reads conflict-free
#define SC_PL_ES 17          // row-phase exchange stride (per k1)
#define SC_PL_JMAX 17
#define SC_PL_KMAX 32

struct PlLds {
  static constexpr int T_c = SC_PL_N * SC_PL_RS;                 // 2560 complex
  static constexpr int E_c = 16 * 8 * SC_PL_ES;                  // row exchange, 16 groups: 2176
  static constexpr int Z_c = 16 * 34;                            // row-phase Z[-16..16] per group
  static constexpr int E2_c = 17 * 148;                          // column exchange, 17 groups of 16 x 9 (+4): 2516
  static constexpr int IO_c = SC_PL_KMAX * SC_PL_JMAX;           // kept block staged for one contiguous copy
  static constexpr int off_T = 0;
  static constexpr int off_E = off_T + T_c;
  static constexpr int off_Z = off_E + (E_c > E2_c ? E_c : E2_c);
  static constexpr int off_IO = off_Z + Z_c;
  static constexpr int off_tab = off_IO + IO_c;                  // w128^m, m = 0..127 (session 2: the column phases read
                                                                 // their twiddles here instead of 14-15 global loads per lane)
  static constexpr int total_c = off_tab + SC_PL_N;              // 6292 complex = 50 KB: 3 workgroups per CU
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// ppw consecutive planes per workgroup (session 2, host: SC_PL_PPW): row round n = 4 it + r of the workgroup's planes;
// the loads of round n + depth are requested while round n is transformed -- also across the plane boundary, so only
// the first plane of a workgroup starts with nothing in flight
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 3)
k_pl128_fwd(const float* __restrict__ x, cf32* __restrict__ out, const cf32* __restrict__ tab128,
            const float* __restrict__ cs, int K0, int J, int64_t n_planes, int ppw) {
  SC_SHARED __attribute__((aligned(16))) cf32 lds[PlLds::total_c];
  cf32* T = lds + PlLds::off_T;
  cf32* OUT = lds + PlLds::off_IO;
  const int tid = SC_TID;
  const int64_t plane0 = (int64_t)SC_BID_X * ppw;
#ifndef SC_PL_TAB_GLOBAL
  cf32* tabl = lds + PlLds::off_tab;
  if (tid < SC_PL_N) tabl[tid] = tab128[tid];            // published by the barrier ahead of the row twiddles
#else
  const cf32* tabl = tab128;
#endif
  const int g = tid >> 4, t = tid & 15, L = t & 7;
  cf32* E = lds + PlLds::off_E + g * (8 * SC_PL_ES);
  cf32* Zs = lds + PlLds::off_Z + g * 34;
#ifndef SC_PL_PF_DEPTH
#define SC_PL_PF_DEPTH 1     // rounds of row loads in flight ahead of the one being transformed (A-B: 2)
#endif
  const int n_rounds = 4 * ppw;
  cf32 pfs[SC_PL_PF_DEPTH][8];
#ifdef SC_PL_STAGED_LOADS
  // Session 2 (as in sc_kernels_plane64.h): a wave owns the 8 consecutive rows of its four row pairs (4 KB per round)
  // and reads them as four 16-byte loads per lane -- one contiguous KB per instruction instead of 64-byte pieces of
  // four rows -- into its own part of the exchange buffer (row stride 136 floats: conflict-free both ways); lanes pick
  // up x[t + 16 j] from there.  The registers of the next round's loads are the same 16 as before.
  const int wv = tid >> 6, lane = tid & 63;
  float* stg = reinterpret_cast<float*>(lds + PlLds::off_E + wv * (4 * 8 * SC_PL_ES));
  sc_f4 ldq[SC_PL_PF_DEPTH][4];
  auto request = [&](const int n, sc_f4 (&q)[4]) {
#ifdef SC_PL_ABL_NOLOAD                                   // measurement build only
    if (n >= -1) return;
#endif
    if (n >= n_rounds) return;
    int64_t pl = plane0 + (n >> 2);
    pl = pl < n_planes ? pl : n_planes - 1;              // past the end: a harmless re-read
    const sc_f4* src = reinterpret_cast<const sc_f4*>(x + pl * (int64_t)(SC_PL_N * SC_PL_N) +
                                                      (2 * (4 * wv + 16 * (n & 3))) * SC_PL_N) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = SC_LOAD_STREAM(src + 64 * i);
  };
#pragma unroll
  for (int d = 0; d < SC_PL_PF_DEPTH; ++d) request(d, ldq[d]);
#else
  auto prefetch = [&](const int n, cf32 (&pf)[8]) {
    if (n >= n_rounds) return;
    int64_t pl = plane0 + (n >> 2);
    pl = pl < n_planes ? pl : n_planes - 1;
    const float* ra = x + pl * (int64_t)(SC_PL_N * SC_PL_N) + (2 * (g + 16 * (n & 3))) * SC_PL_N + t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pf[j].x = SC_LOAD_STREAM(ra + 16 * j);
      pf[j].y = SC_LOAD_STREAM(ra + SC_PL_N + 16 * j);
    }
  };
#pragma unroll
  for (int d = 0; d < SC_PL_PF_DEPTH; ++d) prefetch(d, pfs[d]);
#endif
  // the lane's row twiddles out of the LDS table (the first round's loads are in flight meanwhile; A-B
  // -DSC_PL_TW1_GLOBAL: 7 global loads per lane and plane instead of one barrier)
  cf32 tw1[8];
#if !defined(SC_PL_TAB_GLOBAL) && !defined(SC_PL_TW1_GLOBAL)
  SC_SYNC();
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) tw1[k1] = sc_lds_ld64(tabl + ((t * k1) & 127));
#else
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) tw1[k1] = tab128[(t * k1) & 127];
#endif
#pragma unroll 1
  for (int it = 0; it < ppw; ++it) {
    const int64_t plane = plane0 + it;
    if (plane >= n_planes) break;                        // uniform
    // ---------------- rows: 4 rounds of 16 packed row pairs ----------------
#if SC_PL_PF_DEPTH == 1
#pragma unroll 1
#else
#pragma unroll
#endif
    for (int r = 0; r < 4; ++r) {
      const int p = g + 16 * r;
      cf32 u[8];
      cf32 (&pf)[8] = pfs[r % SC_PL_PF_DEPTH];
#ifdef SC_PL_STAGED_LOADS
      sc_f4 (&lq)[4] = ldq[r % SC_PL_PF_DEPTH];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<sc_f4*>(stg + (2 * i + (lane >> 5)) * 136 + 4 * (lane & 31)) = lq[i];
      request(4 * it + r + SC_PL_PF_DEPTH, lq);
      SC_WAVE_SYNC();
      {
        const float* ra = stg + (2 * (g & 3)) * 136 + t;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = cf_make(ra[16 * j], ra[136 + 16 * j]);
      }
      SC_WAVE_SYNC();                                    // the staging area is this wave's exchange area
      dft8<-1>(pf, u);                                   // over j -> k1
#else
      dft8<-1>(pf, u);                                   // over j -> k1
      prefetch(4 * it + r + SC_PL_PF_DEPTH, pf);
#endif
      E[t] = u[0];
#pragma unroll
      for (int k1 = 1; k1 < 8; ++k1) E[k1 * SC_PL_ES + t] = cf_mul_cs(u[k1], tw1[k1]);
      SC_WAVE_SYNC();
      cf32 y[16], o[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) y[q] = sc_lds_ld64(E + L * SC_PL_ES + q);   // explicit widths: sc_device.h
      fft16<-1>(y, o);                                   // over t -> k2: Z[L + 8 k2]
      if (t < 8) {
        Zs[16 + t] = o[0];
        Zs[24 + t] = o[1];
        Zs[8 + t] = o[15];
        Zs[t] = o[14];
        if (t == 0) Zs[32] = o[2];
      }
      SC_WAVE_SYNC();
      {
        // A = (Z[k] + conj Z[-k]) / 2, B = -i (Z[k] - conj Z[-k]) / 2 (the 1/2 rides on the column scale)
        const cf32 zk = Zs[16 + t], zm = Zs[16 - t];
        T[(2 * p) * SC_PL_RS + t] = cf_make(zk.x + zm.x, zk.y - zm.y);
        T[(2 * p + 1) * SC_PL_RS + t] = cf_make(zk.y + zm.y, zm.x - zk.x);
        if (t == 0) {
          const cf32 zt = Zs[32], zb = Zs[0];
          T[(2 * p) * SC_PL_RS + 16] = cf_make(zt.x + zb.x, zt.y - zb.y);
          T[(2 * p + 1) * SC_PL_RS + 16] = cf_make(zt.y + zb.y, zb.x - zt.x);
        }
      }
    }
    SC_SYNC();
    // ---------------- columns: 8 lanes per kept column ----------------
    {
      const int c = tid >> 3, tc = tid & 7;
#ifdef SC_PL_ABL_NOCOL                                    // measurement build only
      const bool act = c < 0;
#else
      const bool act = c < SC_PL_JMAX;                   // waves 0, 1 and the first group of wave 2
#endif
      cf32* E2 = lds + PlLds::off_E + (act ? c : 0) * 148;
      if (act) {
        cf32 v[16], u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = sc_lds_ld64(T + (tc + 8 * j) * SC_PL_RS + c);
        fft16<-1>(v, u);                                 // over j -> k1
        E2[tc] = u[0];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) E2[k1 * 9 + tc] = cf_mul_cs(u[k1], tabl[(tc * k1) & 127]);
      }
      SC_WAVE_SYNC();
      if (act) {
        const float s = (c < J) ? cs[c] : 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k1 = tc + 8 * h;
          cf32 y[8], o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) y[q] = sc_lds_ld64(E2 + k1 * 9 + q);
          dft8<-1>(y, o);                                // over t -> k2: k = k1 + 16 k2; kept: k2 = 0 and k2 = -1
          const int rp = k1 + K0 / 2, rn = k1 - 16 + K0 / 2;
          if (c < J) {
            if (rp < K0) OUT[rp * J + c] = cf_scale(o[0], s);
            if (rn >= 0) OUT[rn * J + c] = cf_scale(o[7], s);
          }
        }
      }
    }
    SC_SYNC();                                           // also: the column exchange is free for the next plane's staging
    cf32* dst = out + plane * (int64_t)K0 * J;
    for (int i = tid; i < K0 * J; i += 256) dst[i] = OUT[i];
  }
}

// ------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 3)
k_pl128_inv(const cf32* __restrict__ in, float* __restrict__ y, const cf32* __restrict__ tab128,
            const float* __restrict__ cs, const float* __restrict__ bias, int64_t planes_per_image, int channels,
            int K0, int J) {
  SC_SHARED __attribute__((aligned(16))) cf32 lds[PlLds::total_c];
  cf32* T = lds + PlLds::off_T;
  cf32* IN = lds + PlLds::off_IO;
  const int tid = SC_TID;
  const int64_t plane = SC_BID_X;
  {
    const cf32* src = in + plane * (int64_t)K0 * J;
    for (int i = tid; i < K0 * J; i += 256) IN[i] = src[i];
  }
#ifndef SC_PL_TAB_GLOBAL
  cf32* tabl = lds + PlLds::off_tab;
  if (tid < SC_PL_N) tabl[tid] = tab128[tid];
#else
  const cf32* tabl = tab128;
#endif
  SC_SYNC();
  // ---------------- columns: kept rows -> all 128 rows of the tile ----------------
  {
    const int c = tid >> 3, t = tid & 7;
    const bool act = c < SC_PL_JMAX;
    cf32* E2 = lds + PlLds::off_E + (act ? c : 0) * 148;
    if (act) {
      const float s = (c < J) ? cs[c] : 0.f;             // norm x column weight (x 1/2 for c > 0)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k1 = t + 8 * h;
        const int rp = k1 + K0 / 2, rn = k1 - 16 + K0 / 2;
        const cf32 x0 = (c < J && rp < K0) ? cf_scale(IN[rp * J + c], s) : cf_make(0.f, 0.f);
        const cf32 xm = (c < J && rn >= 0) ? cf_scale(IN[rn * J + c], s) : cf_make(0.f, 0.f);
        // g[q] = X[k1] + w8^(-q) X[k1 - 16],  w8 = exp(+2 pi i / 8);  then the twiddle conj(w128^(q k1))
        cf32 e[8], o[8];
        e[0] = x0;
#pragma unroll
        for (int q = 1; q < 7; ++q) e[q] = cf_make(0.f, 0.f);
        e[7] = xm;
        dft8<+1>(e, o);
        E2[k1 * 9] = o[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) E2[k1 * 9 + q] = cf_mul_cs(o[q], cf_conj(tabl[(q * k1) & 127]));
      }
    }
    SC_WAVE_SYNC();
    if (act) {
      cf32 u[16], v[16];
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) u[k1] = sc_lds_ld64(E2 + k1 * 9 + t);
      fft16<+1>(u, v);                                   // over k1 -> j: row n = t + 8 j
#pragma unroll
      for (int j = 0; j < 16; ++j) T[(t + 8 * j) * SC_PL_RS + c] = v[j];
    }
  }
  SC_SYNC();
  // ---------------- rows ----------------
  {
    const int g = tid >> 4, t = tid & 15, L = t & 7;
    cf32* E = lds + PlLds::off_E + g * (8 * SC_PL_ES);
    cf32* Zs = lds + PlLds::off_Z + g * 34;
    cf32 tw2[16];
#pragma unroll
    for (int q = 1; q < 16; ++q) tw2[q] = cf_conj(tabl[(q * L) & 127]);
    const float bv = bias ? bias[(plane / planes_per_image) % channels] : 0.f;
    float* yp = y + plane * (int64_t)(SC_PL_N * SC_PL_N);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      const int p = g + 16 * r;
      {
        // Z[k] = A + i B, Z[-k] = conj A + i conj B; k = 0: (Re A, Re B)
        const cf32 A = sc_lds_ld64(T + (2 * p) * SC_PL_RS + t), B = sc_lds_ld64(T + (2 * p + 1) * SC_PL_RS + t);
        Zs[16 + t] = (t == 0) ? cf_make(A.x, B.x) : cf_make(A.x - B.y, A.y + B.x);
        if (t > 0) Zs[16 - t] = cf_make(A.x + B.y, B.x - A.y);
        if (t == 0) {
          const cf32 At = T[(2 * p) * SC_PL_RS + 16], Bt = T[(2 * p + 1) * SC_PL_RS + 16];
          Zs[32] = cf_make(At.x - Bt.y, At.y + Bt.x);
          Zs[0] = cf_make(At.x + Bt.y, Bt.x - At.y);
        }
      }
      SC_WAVE_SYNC();
      cf32 e[16], o[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) e[q] = cf_make(0.f, 0.f);
      e[0] = Zs[16 + L];
      e[1] = Zs[24 + L];
      e[15] = Zs[8 + L];
      e[14] = Zs[L];
      e[2] = (L == 0) ? Zs[32] : cf_make(0.f, 0.f);
      fft16<+1>(e, o);                                   // zero-padded 16-point stage: g[q], q = t index of the line
      if (t < 8) {
        E[t * SC_PL_ES] = o[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) E[t * SC_PL_ES + q] = cf_mul_cs(o[q], tw2[q]);
      }
      SC_WAVE_SYNC();
      cf32 u[8], z[8];
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) u[k1] = sc_lds_ld64(E + k1 * SC_PL_ES + t);
      dft8<+1>(u, z);                                    // over k1 -> j: z[j] = a[t + 16 j] + i b[t + 16 j]
#ifdef SC_PL_STAGED_STORES
      // A-B (session 2), NOT taken: the wave's 8 rows through its exchange area and out as four 16-byte stores per lane
      // -- 435-439 us against 419-425 us for the 64-byte pieces below (profiles/r03s2_pl128_staged_loads_ab.txt):
      // what helps the loads does not help the non-temporal stores
      {
        const int wv = tid >> 6, lane = tid & 63;
        float* stg = reinterpret_cast<float*>(lds + PlLds::off_E + wv * (4 * 8 * SC_PL_ES));
        SC_WAVE_SYNC();                                  // every lane has read its exchange values
        float* rs = stg + (2 * (g & 3)) * 136 + t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rs[16 * j] = z[j].x + bv;
          rs[136 + 16 * j] = z[j].y + bv;
        }
        SC_WAVE_SYNC();
        sc_f4* dst = reinterpret_cast<sc_f4*>(yp + (2 * (4 * wv + 16 * r)) * SC_PL_N) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          SC_STORE_STREAM(dst + 64 * i, *reinterpret_cast<const sc_f4*>(stg + (2 * i + (lane >> 5)) * 136 + 4 * (lane & 31)));
      }
#else
      float* ra = yp + (2 * p) * SC_PL_N + t;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#ifdef SC_PL_ABL_NOSTORE                                  // measurement build only
        if (z[j].x == 12345.678f) SC_STORE_STREAM(ra + 16 * j, z[j].x + bv);
        if (z[j].y == 12345.678f) SC_STORE_STREAM(ra + SC_PL_N + 16 * j, z[j].y + bv);
#else
        SC_STORE_STREAM(ra + 16 * j, z[j].x + bv);
        SC_STORE_STREAM(ra + SC_PL_N + 16 * j, z[j].y + bv);
#endif
      }
#endif
      SC_WAVE_SYNC();                                    // Zs / E are rewritten by the next round
    }
  }
}

// ------------------------------------------------------------------------------------------
// first axis of 3-D data: 128-point lines with an inner stride (the plane results), kept rows <= 32, centred.
//   forward  in[o][128][inner] -> out[o][K][inner];   inverse  in[o][K][inner] -> out[o][128][inner]
// Same line as the column phase above (8 lanes x 16 points, two kept k2 per k1), fed from global memory: a wave
// owns 8 neighbouring inner positions (lane = (t, c): 64 contiguous bytes per row and instruction), a workgroup 32.
// The size-agnostic matrix-core pass it replaces took 80-94 us per launch for 0.18 GB at FNO3d 128^3
// (profiles/r02_fno3d_128_plane_fft_kernel_stats.txt).
// ------------------------------------------------------------------------------------------
// Round 5: the kept-row side of the first-axis pass addresses a SHARDED spectrum natively (sh.rows > 0: the rank-major
// all-to-all buffer [block][image][rows][rest] of the mode-parallel layer, include/sc_engine.h sc_spectrum_shards) --
// until then the 3-D routes went through a staging buffer and one permutation launch per transform (k_spectrum_shard:
// 5.5 us + a kernel boundary, four times per layer step; a tenth of the ~0.3 ms per-rank step of configs[3] on 8 ranks)
SC_HD int64_t ax_row_offset(const F3Shard sh, const int64_t o, const int r, const int K, const int64_t inner) {
  if (sh.rows <= 0) return (o * K + r) * inner;
  const int blk = r / sh.rows;
  return (int64_t)blk * sh.block_stride + (o * sh.rows + (r - blk * sh.rows)) * inner;
}
template <int DIR>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 4)
k_ax128(const cf32* __restrict__ in, cf32* __restrict__ out, const cf32* __restrict__ tab128, int64_t inner, int K,
        F3Shard sh) {
  SC_SHARED __attribute__((aligned(16))) cf32 lds[4 * 8 * 148 + SC_PL_N];
  const int tid = SC_TID, w = tid >> 6, lane = tid & 63, c = lane & 7, t = lane >> 3;
  cf32* tabl = lds + 4 * 8 * 148;                        // w128^m in LDS (session 2): 15 LDS reads per lane instead of 15 global loads
  if (tid < SC_PL_N) tabl[tid] = tab128[tid];
  const int64_t o = SC_BID_Y;
  const int64_t col = ((int64_t)SC_BID_X * 4 + w) * 8 + c;
  const bool live = col < inner;
  cf32* E2 = lds + (w * 8 + c) * 148;
  if (DIR < 0) {
    const cf32* src = in + (o * SC_PL_N + t) * inner + col;
    cf32 v[16], u[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = live ? src[(int64_t)8 * j * inner] : cf_make(0.f, 0.f);
    SC_SYNC();                                           // the table (the line's loads are in flight meanwhile)
    fft16<-1>(v, u);
    E2[t] = u[0];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) E2[k1 * 9 + t] = cf_mul_cs(u[k1], sc_lds_ld64(tabl + ((t * k1) & 127)));
    SC_WAVE_SYNC();
    cf32* dst = out + col;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k1 = t + 8 * h;
      cf32 y[8], r[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = sc_lds_ld64(E2 + k1 * 9 + q);
      dft8<-1>(y, r);
      const int rp = k1 + K / 2, rn = k1 - 16 + K / 2;
      if (live && rp < K) dst[ax_row_offset(sh, o, rp, K, inner)] = r[0];
      if (live && rn >= 0) dst[ax_row_offset(sh, o, rn, K, inner)] = r[7];
    }
  } else {
    const cf32* src = in + col;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k1 = t + 8 * h;
      const int rp = k1 + K / 2, rn = k1 - 16 + K / 2;
      cf32 e[8], g[8];
      e[0] = (live && rp < K) ? src[ax_row_offset(sh, o, rp, K, inner)] : cf_make(0.f, 0.f);
#pragma unroll
      for (int q = 1; q < 7; ++q) e[q] = cf_make(0.f, 0.f);
      e[7] = (live && rn >= 0) ? src[ax_row_offset(sh, o, rn, K, inner)] : cf_make(0.f, 0.f);
      if (h == 0) SC_SYNC();                             // the table (uniform: h is the unrolled loop's index)
      dft8<+1>(e, g);
      E2[k1 * 9] = g[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) E2[k1 * 9 + q] = cf_mul_cs(g[q], cf_conj(sc_lds_ld64(tabl + ((q * k1) & 127))));
    }
    SC_WAVE_SYNC();
    cf32 u[16], v[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) u[k1] = sc_lds_ld64(E2 + k1 * 9 + t);
    fft16<+1>(u, v);
    if (live) {
      cf32* dst = out + (o * SC_PL_N + t) * inner + col;
#pragma unroll
      for (int j = 0; j < 16; ++j) dst[(int64_t)8 * j * inner] = v[j];
    }
  }
}
