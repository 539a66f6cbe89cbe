// sc_kernels_plane64.h -- factorised transforms of the LAST TWO axes for 64 x 64 planes, one launch each way.
//
// The 64-point member of the plane family of sc_kernels_plane.h (2-D 64 x 64 grids, FNO3d 64^3): a plane is 16 KB of
// real data, its kept block <= 32 rows x 17 columns.  The direct-DFT plane form of sc_kernels_mdft.h that served these
// shapes moved 1.7-2.5 TB/s (round 3, session 2: 39-41 us for 85 MB at 64 x 64, B x C = 4096).
//
//   rows     64 real points x 2 rows packed as one complex line, 8 lanes x 8 points (n = t + 8 j): radix-8 over j in
//              registers, twiddle w64^(t k1), exchange, 8-point DFT over t by lane k1 = t: Z[k1 + 8 k2], of which
//              |k| <= 16 is used; Z[k], Z[-k] -> A[k], B[k] -> tile T[row][k].  One workgroup = the 32 row pairs of
//              one plane in ONE round.
//   columns  64 complex points, 8 lanes x 8 points (n = t + 8 j): the same two stages; k = k1 + 8 k2 with
//              k2 in {0, 1, -1, -2} are the 32 centred rows
//   memory   a wave owns 16 consecutive rows (4 KB): four 16-byte accesses per lane, each instruction one contiguous
//              KB, staged through the wave's own part of the exchange buffer so that lane (pair, t) finds x[t + 8 j]
//              (row stride 68 floats: conflict-free both ways).  With 4-byte accesses in the lanes' own order a wave
//              instruction would touch eight 32-byte pieces of eight different rows.
//   forward  k_pl64_fwd   x[plane][64][64] real -> out[plane][K0][J]   (rfft2 restricted to the kept block)
//   inverse  k_pl64_inv   in[plane][K0][J]      -> y[plane][64][64] real (+ bias): the exact transpose
//   k_ax64   64-point first-axis lines of 3-D data over the plane results (as k_ax128)
// Reference lines: spectral_convolution.py:443-449, 500-519 (forward), :531-568 (inverse).
#pragma once
#include "sc_kernels_plane.h"

#define SC_P64_N 64
#define SC_P64_RS 20          // tile row stride (complex)
#define SC_P64_ES 9           // exchange stride per k1
#define SC_P64_SS 68          // staging row stride (floats)
#define SC_P64_JMAX 17
#define SC_P64_KMAX 32
#ifndef SC_P64_PPW_DEFAULT
#define SC_P64_PPW_DEFAULT 2  // planes per workgroup of the forward kernel (sc_engine.cpp; environment SC_P64_PPW overrides): 4096 planes
                              // 21.5 -> 19.7 us, 16384 planes 68 -> 65.5 us (profiles/r03s2_pl64_ppw_ab.txt)
#endif

struct P64Lds {
  static constexpr int T_c = SC_P64_N * SC_P64_RS;               // 1280 complex
  static constexpr int Ew_c = 8 * 8 * SC_P64_ES;                 // one wave's exchange: 8 groups of 8 x 9 = 576
  static constexpr int E_c = 4 * Ew_c;                           // 2304; also: row staging (16 x 68 floats = 544 complex per
                                                                 // wave), Z[-16..16] per group (8 x 34 per wave), and the
                                                                 // column exchange (17 x 72 = 1224)
  static constexpr int IO_c = SC_P64_KMAX * SC_P64_JMAX;         // kept block staged for one contiguous copy
  static constexpr int off_T = 0;
  static constexpr int off_E = off_T + T_c;
  static constexpr int off_IO = off_E + E_c;
  static constexpr int off_tab = off_IO + IO_c;                  // w64^m, m = 0..63
  static constexpr int total_c = off_tab + SC_P64_N;             // 4192 complex = 33.5 KB: 4 workgroups per CU
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// ppw consecutive planes per workgroup (host: SC_P64_PPW): the rows of plane i + 1 are requested as soon as those of
// plane i have been staged, so a workgroup has loads in flight while it transforms
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 4)
k_pl64_fwd(const float* __restrict__ x, cf32* __restrict__ out, const cf32* __restrict__ tab64,
           const float* __restrict__ cs, int K0, int J, int64_t n_planes, int ppw) {
  SC_SHARED __attribute__((aligned(16))) cf32 lds[P64Lds::total_c];
  cf32* T = lds + P64Lds::off_T;
  cf32* tabl = lds + P64Lds::off_tab;
  cf32* OUT = lds + P64Lds::off_IO;
  const int tid = SC_TID, w = tid >> 6, lane = tid & 63;
  const int64_t plane0 = (int64_t)SC_BID_X * ppw;
  const int gl = lane >> 3, t = lane & 7, p = tid >> 3;
  cf32* Ew = lds + P64Lds::off_E + w * P64Lds::Ew_c;     // this wave's exchange / staging / Z area
  float* stg = reinterpret_cast<float*>(Ew);
  sc_f4 ld[4];
  auto request = [&](const int64_t plane) {
    const int64_t pl = plane < n_planes ? plane : n_planes - 1;         // past the end: a harmless re-read
    const sc_f4* src = reinterpret_cast<const sc_f4*>(x + pl * (int64_t)(SC_P64_N * SC_P64_N) + w * (16 * SC_P64_N)) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) ld[i] = SC_LOAD_STREAM(src + 64 * i);
  };
  request(plane0);
  if (tid < SC_P64_N) tabl[tid] = tab64[tid];
  SC_SYNC();                                             // the table (the rows are in flight meanwhile)
  cf32 tw1[8];
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) tw1[k1] = sc_lds_ld64(tabl + ((t * k1) & 63));
#pragma unroll 1
  for (int it = 0; it < ppw; ++it) {
    const int64_t plane = plane0 + it;
    if (plane >= n_planes) break;                        // uniform
    // ---------------- rows: the plane's 32 packed row pairs at once ----------------
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<sc_f4*>(stg + (4 * i + (lane >> 4)) * SC_P64_SS + 4 * (lane & 15)) = ld[i];
    if (it + 1 < ppw) request(plane + 1);
    SC_WAVE_SYNC();
    cf32 a[8], u[8];
    {
      const float* ra = stg + (2 * gl) * SC_P64_SS + t;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = cf_make(ra[8 * j], ra[SC_P64_SS + 8 * j]);
    }
    SC_WAVE_SYNC();                                      // the staging area becomes the exchange
    dft8<-1>(a, u);                                      // over j -> k1
    cf32* E = Ew + gl * (8 * SC_P64_ES);
    E[t] = u[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) E[k1 * SC_P64_ES + t] = cf_mul_cs(u[k1], tw1[k1]);
    SC_WAVE_SYNC();
    {
      cf32 y[8], o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = sc_lds_ld64(E + t * SC_P64_ES + q);
      dft8<-1>(y, o);                                    // over t -> k2: Z[t + 8 k2]
      SC_WAVE_SYNC();                                    // ... and then Z[-16..16] of the wave's 8 row pairs
      cf32* Zs = Ew + gl * 34;
      Zs[16 + t] = o[0];
      Zs[24 + t] = o[1];
      Zs[8 + t] = o[7];
      Zs[t] = o[6];
      if (t == 0) Zs[32] = o[2];
      SC_WAVE_SYNC();
      // A = (Z[k] + conj Z[-k]) / 2, B = -i (Z[k] - conj Z[-k]) / 2 (the 1/2 rides on the column scale)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = t + 8 * h;
        const cf32 zk = Zs[16 + c], zm = Zs[16 - c];
        T[(2 * p) * SC_P64_RS + c] = cf_make(zk.x + zm.x, zk.y - zm.y);
        T[(2 * p + 1) * SC_P64_RS + c] = cf_make(zk.y + zm.y, zm.x - zk.x);
      }
      if (t == 0) {
        const cf32 zt = Zs[32], zb = Zs[0];
        T[(2 * p) * SC_P64_RS + 16] = cf_make(zt.x + zb.x, zt.y - zb.y);
        T[(2 * p + 1) * SC_P64_RS + 16] = cf_make(zt.y + zb.y, zb.x - zt.x);
      }
    }
    SC_SYNC();
    // ---------------- columns: 8 lanes per kept column ----------------
    {
      const int c = tid >> 3;
      const bool act = c < SC_P64_JMAX;                  // waves 0, 1 and the first group of wave 2
      cf32* E2 = lds + P64Lds::off_E + (act ? c : 0) * 72;
      if (act) {
        cf32 v[8], u2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sc_lds_ld64(T + (t + 8 * j) * SC_P64_RS + c);
        dft8<-1>(v, u2);                                 // over j -> k1
        E2[t] = u2[0];
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) E2[k1 * SC_P64_ES + t] = cf_mul_cs(u2[k1], tw1[k1]);
      }
      SC_WAVE_SYNC();
      if (act && c < J) {
        const float s = cs[c];
        cf32 y[8], o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) y[q] = sc_lds_ld64(E2 + t * SC_P64_ES + q);
        dft8<-1>(y, o);                                  // over t -> k2: k = t + 8 k2; kept: k2 = 0, 1, -1, -2
        const int r0 = t + K0 / 2;
        if (r0 < K0) OUT[r0 * J + c] = cf_scale(o[0], s);
        if (r0 + 8 < K0) OUT[(r0 + 8) * J + c] = cf_scale(o[1], s);
        if (r0 - 8 >= 0) OUT[(r0 - 8) * J + c] = cf_scale(o[7], s);
        if (r0 - 16 >= 0) OUT[(r0 - 16) * J + c] = cf_scale(o[6], s);
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
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 4)
k_pl64_inv(const cf32* __restrict__ in, float* __restrict__ y, const cf32* __restrict__ tab64,
           const float* __restrict__ cs, const float* __restrict__ bias, int64_t planes_per_image, int channels,
           int K0, int J) {
  SC_SHARED __attribute__((aligned(16))) cf32 lds[P64Lds::total_c];
  cf32* T = lds + P64Lds::off_T;
  cf32* IN = lds + P64Lds::off_IO;
  cf32* tabl = lds + P64Lds::off_tab;
  const int tid = SC_TID, w = tid >> 6, lane = tid & 63;
  const int64_t plane = SC_BID_X;
  {
    const cf32* src = in + plane * (int64_t)K0 * J;
    for (int i = tid; i < K0 * J; i += 256) IN[i] = src[i];
  }
  if (tid < SC_P64_N) tabl[tid] = tab64[tid];
  SC_SYNC();
  // ---------------- columns: kept rows -> all 64 rows of the tile ----------------
  {
    const int c = tid >> 3, t = tid & 7;
    const bool act = c < SC_P64_JMAX;
    cf32* E2 = lds + P64Lds::off_E + (act ? c : 0) * 72;
    if (act) {
      const bool live = c < J;
      const float s = live ? cs[c] : 0.f;                // norm x column weight (x 1/2 for c > 0)
      const int r0 = t + K0 / 2;
      cf32 e[8], o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) e[q] = cf_make(0.f, 0.f);
      if (live && r0 < K0) e[0] = cf_scale(IN[r0 * J + c], s);
      if (live && r0 + 8 < K0) e[1] = cf_scale(IN[(r0 + 8) * J + c], s);
      if (live && r0 - 8 >= 0) e[7] = cf_scale(IN[(r0 - 8) * J + c], s);
      if (live && r0 - 16 >= 0) e[6] = cf_scale(IN[(r0 - 16) * J + c], s);
      dft8<+1>(e, o);                                    // zero-padded stage over k2 -> q, then conj(w64^(q k1))
      E2[t * SC_P64_ES] = o[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) E2[t * SC_P64_ES + q] = cf_mul_cs(o[q], cf_conj(sc_lds_ld64(tabl + ((q * t) & 63))));
    }
    SC_WAVE_SYNC();
    if (act) {
      cf32 u[8], v[8];
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) u[k1] = sc_lds_ld64(E2 + k1 * SC_P64_ES + t);
      dft8<+1>(u, v);                                    // over k1 -> j: row n = t + 8 j
#pragma unroll
      for (int j = 0; j < 8; ++j) T[(t + 8 * j) * SC_P64_RS + c] = v[j];
    }
  }
  SC_SYNC();
  // ---------------- rows ----------------
  {
    const int gl = lane >> 3, t = lane & 7, p = tid >> 3;
    cf32* Ew = lds + P64Lds::off_E + w * P64Lds::Ew_c;
    float* stg = reinterpret_cast<float*>(Ew);
    cf32* Zs = Ew + gl * 34;
    cf32 tw2[8];
#pragma unroll
    for (int q = 1; q < 8; ++q) tw2[q] = cf_conj(sc_lds_ld64(tabl + ((q * t) & 63)));
    const float bv = bias ? bias[(plane / planes_per_image) % channels] : 0.f;
    // Z[k] = A + i B, Z[-k] = conj A + i conj B; k = 0: (Re A, Re B)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = t + 8 * h;
      const cf32 A = sc_lds_ld64(T + (2 * p) * SC_P64_RS + c), B = sc_lds_ld64(T + (2 * p + 1) * SC_P64_RS + c);
      Zs[16 + c] = (c == 0) ? cf_make(A.x, B.x) : cf_make(A.x - B.y, A.y + B.x);
      if (c > 0) Zs[16 - c] = cf_make(A.x + B.y, B.x - A.y);
    }
    if (t == 0) {
      const cf32 At = T[(2 * p) * SC_P64_RS + 16], Bt = T[(2 * p + 1) * SC_P64_RS + 16];
      Zs[32] = cf_make(At.x - Bt.y, At.y + Bt.x);
      Zs[0] = cf_make(At.x + Bt.y, Bt.x - At.y);
    }
    SC_WAVE_SYNC();
    cf32 e[8], o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = cf_make(0.f, 0.f);
    e[0] = Zs[16 + t];
    e[1] = Zs[24 + t];
    e[7] = Zs[8 + t];
    e[6] = Zs[t];
    if (t == 0) e[2] = Zs[32];
    SC_WAVE_SYNC();                                      // Z becomes the exchange
    dft8<+1>(e, o);                                      // zero-padded 8-point stage over k2
    cf32* E = Ew + gl * (8 * SC_P64_ES);
    E[t * SC_P64_ES] = o[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) E[t * SC_P64_ES + q] = cf_mul_cs(o[q], tw2[q]);
    SC_WAVE_SYNC();
    cf32 u[8], z[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) u[k1] = sc_lds_ld64(E + k1 * SC_P64_ES + t);
    dft8<+1>(u, z);                                      // over k1 -> j: z[j] = a[t + 8 j] + i b[t + 8 j]
#ifdef SC_P64_DIRECT_STORES                            // A-B, NOT taken: 32-byte pieces straight from the lanes -- 107 against
                                                       // 71 us at 16384 planes (profiles/r03s2_pl128_staged_loads_ab.txt)
    {
      float* ra = y + plane * (int64_t)(SC_P64_N * SC_P64_N) + (2 * p) * SC_P64_N + t;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        SC_STORE_STREAM(ra + 8 * j, z[j].x + bv);
        SC_STORE_STREAM(ra + SC_P64_N + 8 * j, z[j].y + bv);
      }
    }
    return;
#endif
    SC_WAVE_SYNC();                                      // the exchange becomes the staging of the wave's 16 rows
    {
      float* ra = stg + (2 * gl) * SC_P64_SS + t;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ra[8 * j] = z[j].x + bv;
        ra[SC_P64_SS + 8 * j] = z[j].y + bv;
      }
    }
    SC_WAVE_SYNC();
    sc_f4* dst = reinterpret_cast<sc_f4*>(y + plane * (int64_t)(SC_P64_N * SC_P64_N) + w * (16 * SC_P64_N)) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      SC_STORE_STREAM(dst + 64 * i,
                      *reinterpret_cast<const sc_f4*>(stg + (4 * i + (lane >> 4)) * SC_P64_SS + 4 * (lane & 15)));
  }
}

// ------------------------------------------------------------------------------------------
// first axis of 3-D data: 64-point lines with an inner stride (the plane results), kept rows <= 32, centred.
//   forward  in[o][64][inner] -> out[o][K][inner];   inverse  in[o][K][inner] -> out[o][64][inner]
// A wave owns 8 neighbouring inner positions (lane = (t, c): 64 contiguous bytes per row and instruction).
// ------------------------------------------------------------------------------------------
template <int DIR>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 4)
k_ax64(const cf32* __restrict__ in, cf32* __restrict__ out, const cf32* __restrict__ tab64, int64_t inner, int K,
       F3Shard sh) {                                     // sh: as k_ax128 (sc_kernels_plane.h, round 5)
  SC_SHARED __attribute__((aligned(16))) cf32 lds[4 * 8 * 76 + SC_P64_N];
  const int tid = SC_TID, w = tid >> 6, lane = tid & 63, c = lane & 7, t = lane >> 3;
  cf32* tabl = lds + 4 * 8 * 76;
  if (tid < SC_P64_N) tabl[tid] = tab64[tid];
  const int64_t o = SC_BID_Y;
  const int64_t col = ((int64_t)SC_BID_X * 4 + w) * 8 + c;
  const bool live = col < inner;
  cf32* E2 = lds + (w * 8 + c) * 76;
  const int r0 = t + K / 2;
  if (DIR < 0) {
    const cf32* src = in + (o * SC_P64_N + t) * inner + col;
    cf32 v[8], u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = live ? src[(int64_t)8 * j * inner] : cf_make(0.f, 0.f);
    SC_SYNC();                                           // the table (the line's loads are in flight meanwhile)
    dft8<-1>(v, u);
    E2[t] = u[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) E2[k1 * SC_P64_ES + t] = cf_mul_cs(u[k1], sc_lds_ld64(tabl + ((t * k1) & 63)));
    SC_WAVE_SYNC();
    cf32* dst = out + col;
    cf32 y[8], r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = sc_lds_ld64(E2 + t * SC_P64_ES + q);
    dft8<-1>(y, r);
    if (live && r0 < K) dst[ax_row_offset(sh, o, r0, K, inner)] = r[0];
    if (live && r0 + 8 < K) dst[ax_row_offset(sh, o, r0 + 8, K, inner)] = r[1];
    if (live && r0 - 8 >= 0) dst[ax_row_offset(sh, o, r0 - 8, K, inner)] = r[7];
    if (live && r0 - 16 >= 0) dst[ax_row_offset(sh, o, r0 - 16, K, inner)] = r[6];
  } else {
    const cf32* src = in + col;
    cf32 e[8], g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = cf_make(0.f, 0.f);
    if (live && r0 < K) e[0] = src[ax_row_offset(sh, o, r0, K, inner)];
    if (live && r0 + 8 < K) e[1] = src[ax_row_offset(sh, o, r0 + 8, K, inner)];
    if (live && r0 - 8 >= 0) e[7] = src[ax_row_offset(sh, o, r0 - 8, K, inner)];
    if (live && r0 - 16 >= 0) e[6] = src[ax_row_offset(sh, o, r0 - 16, K, inner)];
    SC_SYNC();                                           // the table
    dft8<+1>(e, g);
    E2[t * SC_P64_ES] = g[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) E2[t * SC_P64_ES + q] = cf_mul_cs(g[q], cf_conj(sc_lds_ld64(tabl + ((q * t) & 63))));
    SC_WAVE_SYNC();
    cf32 u[8], v[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) u[k1] = sc_lds_ld64(E2 + k1 * SC_P64_ES + t);
    dft8<+1>(u, v);
    if (live) {
      cf32* dst = out + (o * SC_P64_N + t) * inner + col;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[(int64_t)8 * j * inner] = v[j];
    }
  }
}
