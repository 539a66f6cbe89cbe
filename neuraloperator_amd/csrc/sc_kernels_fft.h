// sc_kernels_fft.h -- fused, pruned 2-D FFT kernels for power-of-two grids (fast path).
//
// One workgroup (256 threads) transforms one (b, c) image and never materialises anything
// but the kept modes:
//
//   forward  x[H][256] real  ->  xhat[Mx][My] complex      (Mx <= 64, My <= 33)
//   inverse  yhat[Mx][My]    ->  y[H][256] real (+ bias)
//
// replacing rfftn + fftshift + slice (spectral_convolution.py:443-449, 500-519) and
// zeros + scatter + ifftshift + ifftn + irfft + bias (:456-462, 520-568) by ONE pass over
// the real tensor each.  Structure (forward; the inverse is its transpose):
//
//   rows are processed in P = H/64 groups; group a holds rows h = P*b + a, b = 0..63
//   (decimation in time along H: Xhat[fx] = sum_a w_H^(a fx) F_a[fx mod 64]).
//   per group:  2 rounds x 16 row pairs: two real rows are packed as one complex row,
//                 256-point FFT as 16 x 16 (radix-16 in registers, one LDS transpose,
//                 radix-16 again -- only the 65 outputs |k| <= 32 survive dead-code
//                 elimination), unpacked into the group tile T_a[64][33] in LDS;
//               33 column FFTs of 64 points on T_a (4 lanes per column: radix-16 in
//                 registers, LDS exchange, radix-4), multiplied by w_H^(a fx) and
//                 accumulated into per-lane registers.
//   the 64 x 33 result leaves through LDS as one contiguous, coalesced store.
//
// LDS: 59 KB per workgroup -> 2 workgroups per CU; every global access of a lane group is a
// contiguous 64 B segment and every cache line is consumed by one wave in back-to-back
// instructions.  HBM traffic = the real tensor once + the kept spectrum once.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "sc_device.h"

#define SC_F2D_W 256
#define SC_F2D_KY 33
#define SC_F2D_KX 64
#define SC_F2D_XS 17   // k1 stride (complex) inside the row exchange buffer: conflict-free
#define SC_F2D_CS 68   // column stride (complex) inside the column exchange buffer

// ------------------------------------------------------------------------------------------
// register codelets
// ------------------------------------------------------------------------------------------
// multiply by exp(DIR * i * pi/2) : DIR = -1 -> -i (forward), +1 -> +i (inverse)
template <int DIR>
SC_HD cf32 rot90(const cf32 a) {
  return DIR < 0 ? cf_make(a.y, -a.x) : cf_make(-a.y, a.x);
}

// in-place 4-point DFT with w4 = exp(DIR * 2 pi i / 4):  a_k <- sum_j a_j w4^(jk)
template <int DIR>
SC_HD void radix4(cf32& a0, cf32& a1, cf32& a2, cf32& a3) {
  const cf32 t0 = cf_add(a0, a2), t1 = cf_sub(a0, a2);
  const cf32 t2 = cf_add(a1, a3), d = cf_sub(a1, a3);
  a0 = cf_add(t0, t2);
  a2 = cf_sub(t0, t2);
  a1 = cf_add_rot<DIR>(t1, d);                           // t1 +- (DIR i) d
  a3 = cf_sub_rot<DIR>(t1, d);
}

// a * exp(DIR * 2 pi i M / 16), M compile-time
template <int DIR, int M>
SC_HD cf32 mul_w16(const cf32 a) {
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
  constexpr float r = 0.70710678118654752440f;
  constexpr float C[16] = {1.f, c1, r, s1, 0.f, -s1, -r, -c1, -1.f, -c1, -r, -s1, 0.f, s1, r, c1};
  constexpr float S[16] = {0.f, s1, r, c1, 1.f, c1, r, s1, 0.f, -s1, -r, -c1, -1.f, -c1, -r, -s1};
  constexpr int m = M & 15;
  if (m == 0) return a;
  if (m == 4) return rot90<DIR>(a);
  if (m == 8) return cf_make(-a.x, -a.y);
  if (m == 12) return rot90<-DIR>(a);
  const float c = C[m], s = (DIR < 0) ? -S[m] : S[m];
  return cf_make(a.x * c - a.y * s, a.x * s + a.y * c);
}

// 16-point DFT, natural order in and out:  o[k] = sum_n v[n] w16^(nk),  w16 = exp(DIR 2 pi i/16)
// (4 x 4 Cooley-Tukey: n = 4 n1 + n2, k = k1 + 4 k2).  Unused outputs are eliminated by the
// compiler, which is how the pruned (|k| <= 32 of 256) second row stage gets cheaper.
template <int DIR>
SC_HD void fft16(cf32 (&v)[16], cf32 (&o)[16]) {
  // step 1: DFT over n1 for every n2  ->  v[4*k1 + n2] = G[n2][k1]
  radix4<DIR>(v[0], v[4], v[8], v[12]);
  radix4<DIR>(v[1], v[5], v[9], v[13]);
  radix4<DIR>(v[2], v[6], v[10], v[14]);
  radix4<DIR>(v[3], v[7], v[11], v[15]);
  // step 2: twiddle w16^(n2 k1)
  v[5] = mul_w16<DIR, 1>(v[5]);
  v[6] = mul_w16<DIR, 2>(v[6]);
  v[7] = mul_w16<DIR, 3>(v[7]);
  v[9] = mul_w16<DIR, 2>(v[9]);
  v[10] = mul_w16<DIR, 4>(v[10]);
  v[11] = mul_w16<DIR, 6>(v[11]);
  v[13] = mul_w16<DIR, 3>(v[13]);
  v[14] = mul_w16<DIR, 6>(v[14]);
  v[15] = mul_w16<DIR, 9>(v[15]);
  // step 3: DFT over n2 for every k1  ->  v[4*k1 + k2] = V[k1 + 4 k2]
  radix4<DIR>(v[0], v[1], v[2], v[3]);
  radix4<DIR>(v[4], v[5], v[6], v[7]);
  radix4<DIR>(v[8], v[9], v[10], v[11]);
  radix4<DIR>(v[12], v[13], v[14], v[15]);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
}

SC_HD cf32 cf_scale(const cf32 a, const float s) { return cf_make(a.x * s, a.y * s); }
// a + i*b
SC_HD cf32 cf_add_i(const cf32 a, const cf32 b) { return cf_add_rot<+1>(a, b); }
// conj(a) + i*conj(b)
SC_HD cf32 cf_conj_add_i(const cf32 a, const cf32 b) { return cf_make(a.x + b.y, b.x - a.y); }

// LDS carve (bytes) shared by both kernels
template <int H>
struct F2dLds {
  static constexpr int off_tw256 = 0;
  static constexpr int off_twH = off_tw256 + SC_F2D_W * 8;
  static constexpr int off_tw64 = off_twH + H * 8;
  static constexpr int off_xch = off_tw64 + 64 * 8;
  static constexpr int xch_bytes = 16 * 16 * SC_F2D_XS * 8;          // 34816
  static constexpr int off_T = off_xch + xch_bytes;
  static constexpr int T_bytes = 64 * SC_F2D_KY * 8;                 // 16896
  static constexpr int off_sep = off_T + T_bytes;
  static constexpr int sep_bytes = 16 * 16 * 2 * 8;                  // 4096
  static constexpr int total = off_sep + sep_bytes;
  static_assert(SC_F2D_KY * SC_F2D_CS * 8 <= xch_bytes, "column exchange must fit in xch");
  static_assert(total <= 64 * 1024, "static LDS limit");
};

SC_HD int f2d_fx(const int q) { return q < 32 ? q : q - 64; }

// ------------------------------------------------------------------------------------------
// forward: one image per workgroup
// ------------------------------------------------------------------------------------------
template <int H>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_fft2d_fwd(const float* __restrict__ x, cf32* __restrict__ xhat, const cf32* __restrict__ tabW,
            const cf32* __restrict__ tabH, const cf32* __restrict__ tab64, int Mx, int My,
            float s_dc, float s_other) {
  constexpr int P = H / 64;
  typedef F2dLds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* tw256 = reinterpret_cast<cf32*>(smem + L::off_tw256);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);
  cf32* sep = reinterpret_cast<cf32*>(smem + L::off_sep);

  const int tid = SC_TID;
  const int64_t img = SC_BID_X;
  const float* xi = x + img * (int64_t)H * SC_F2D_W;

  tw256[tid] = tabW[tid];
  for (int i = tid; i < H; i += 256) twH[i] = tabH[i];
  if (tid < 64) tw64[tid] = tab64[tid];

  const int f = tid >> 4, t = tid & 15;            // row phase: FFT slot, lane inside the FFT
  const bool ctask = tid < 4 * SC_F2D_KY;          // column phase: 4 lanes per column
  const int cc = tid >> 2, cu = tid & 3;
  cf32 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = cf_make(0.f, 0.f);
  SC_SYNC();

  // software prefetch: the 32 floats of the NEXT round are requested before the current round
  // is computed, so HBM latency overlaps the FFT work of this workgroup (and of its CU mate)
  float pa[16], pb[16];
  {
    const float* ra = xi + (int64_t)(P * (2 * f) + 0) * SC_F2D_W + t;
    const float* rb = xi + (int64_t)(P * (2 * f + 1) + 0) * SC_F2D_W + t;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      pa[j] = ra[16 * j];
      pb[j] = rb[16 * j];
    }
  }
#pragma unroll 1
  for (int a = 0; a < P; ++a) {
    // ---------------- rows of group a -> T[b][k] ----------------
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const int p = r * 16 + f;
      const int bA = 2 * p, bB = 2 * p + 1;
      cf32 v[16], o[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = cf_make(pa[j], pb[j]);
      {
        // next round: (a, 1) after (a, 0); (a + 1, 0) after (a, 1); nothing after the last
        const int rn = r ^ 1, an = a + r;
        if (an < P) {
          const int pn = rn * 16 + f;
          const float* ra = xi + (int64_t)(P * (2 * pn) + an) * SC_F2D_W + t;
          const float* rb = xi + (int64_t)(P * (2 * pn + 1) + an) * SC_F2D_W + t;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            pa[j] = ra[16 * j];
            pb[j] = rb[16 * j];
          }
        }
      }
      fft16<-1>(v, o);                              // over n1 (n = 16 n1 + t)
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cf32 y = (k1 == 0) ? o[0] : cf_mul(o[k1], tw256[t * k1]);
        xch[(f * 16 + k1) * SC_F2D_XS + t] = y;
      }
      SC_WAVE_SYNC();                               // the 16 lanes of an FFT slot share a wave
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) v[n2] = xch[(f * 16 + t) * SC_F2D_XS + n2];
      fft16<-1>(v, o);                              // o[k2] = Z[t + 16 k2]
      sep[(f * 16 + t) * 2 + 0] = o[15];            // Z[t - 16]
      sep[(f * 16 + t) * 2 + 1] = o[14];            // Z[t - 32]
      SC_WAVE_SYNC();
      const int pt = (16 - t) & 15;
      const cf32 m1 = sep[(f * 16 + pt) * 2 + 0];   // Z[-t]       (t != 0)
      const cf32 m2 = sep[(f * 16 + pt) * 2 + 1];   // Z[-t - 16]  (t != 0)
      {
        // k = t : pair (Z[k], Z[-k]);  A = (Z[k] + conj Z[-k]) / 2,  B = (Z[k] - conj Z[-k]) / 2i
        const cf32 zk = o[0];
        const cf32 zm = (t == 0) ? o[0] : m1;
        T[bA * SC_F2D_KY + t] = cf_make(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        T[bB * SC_F2D_KY + t] = cf_make(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
      }
      {
        const cf32 zk = o[1];                        // k = t + 16
        const cf32 zm = (t == 0) ? o[15] : m2;
        T[bA * SC_F2D_KY + t + 16] = cf_make(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        T[bB * SC_F2D_KY + t + 16] = cf_make(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
      }
      if (t == 0) {
        const cf32 zk = o[2];                        // k = 32, Z[-32] = own k2 = 14
        const cf32 zm = o[14];
        T[bA * SC_F2D_KY + 32] = cf_make(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        T[bB * SC_F2D_KY + 32] = cf_make(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
      }
      SC_WAVE_SYNC();                               // sep / xch are rewritten by the next round
    }
    SC_SYNC();
    // ---------------- 33 column FFTs of 64 points on T ----------------
    if (ctask) {
      cf32 v[16], o[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = T[(cu + 4 * j) * SC_F2D_KY + cc];
      fft16<-1>(v, o);                              // over j (b = cu + 4 j)
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cf32 y = (k1 == 0) ? o[0] : cf_mul(o[k1], tw64[(cu * k1) & 63]);
        xch[cc * SC_F2D_CS + k1 * 4 + cu] = y;
      }
    }
    SC_WAVE_SYNC();                                 // the 4 lanes of a column share a wave
    if (ctask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k1 = 4 * cu + i;
        cf32 g0 = xch[cc * SC_F2D_CS + k1 * 4 + 0];
        cf32 g1 = xch[cc * SC_F2D_CS + k1 * 4 + 1];
        cf32 g2 = xch[cc * SC_F2D_CS + k1 * 4 + 2];
        cf32 g3 = xch[cc * SC_F2D_CS + k1 * 4 + 3];
        radix4<-1>(g0, g1, g2, g3);                 // g_k2 = F_a[k1 + 16 k2]
        const cf32 gg[4] = {g0, g1, g2, g3};
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
          const int fx = f2d_fx(k1 + 16 * k2);
          int idx = (a * fx) % H;
          if (idx < 0) idx += H;
          cf_mac(acc[i * 4 + k2], twH[idx], gg[k2]);
        }
      }
    }
    SC_SYNC();
  }
  // ---------------- kept block -> LDS -> one contiguous store ----------------
  cf32* OUT = T;
  if (ctask && cc < My) {
    const float s = (cc == 0) ? s_dc : s_other;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        const int fx = f2d_fx(4 * cu + i + 16 * k2);
        const int row = fx + Mx / 2;
        if (row >= 0 && row < Mx) OUT[row * My + cc] = cf_scale(acc[i * 4 + k2], s);
      }
  }
  SC_SYNC();
  cf32* dst = xhat + img * (int64_t)Mx * My;
  for (int i = tid; i < Mx * My; i += 256) dst[i] = OUT[i];
}

// ------------------------------------------------------------------------------------------
// inverse: one image per workgroup
// ------------------------------------------------------------------------------------------
template <int H>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_fft2d_inv(const cf32* __restrict__ yhat, float* __restrict__ y, const float* __restrict__ bias,
            int channels, const cf32* __restrict__ tabW, const cf32* __restrict__ tabH,
            const cf32* __restrict__ tab64, int Mx, int My, float s_dc, float s_other) {
  constexpr int P = H / 64;
  typedef F2dLds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* tw256 = reinterpret_cast<cf32*>(smem + L::off_tw256);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);

  const int tid = SC_TID;
  const int64_t img = SC_BID_X;
  float* yo = y + img * (int64_t)H * SC_F2D_W;
  const cf32* src = yhat + img * (int64_t)Mx * My;

  tw256[tid] = tabW[tid];
  for (int i = tid; i < H; i += 256) twH[i] = tabH[i];
  if (tid < 64) tw64[tid] = tab64[tid];
  cf32* IN = T;
  for (int i = tid; i < Mx * My; i += 256) IN[i] = src[i];
  const float badd = (bias != nullptr) ? bias[img % channels] : 0.f;

  const int f = tid >> 4, t = tid & 15;
  const bool ctask = tid < 4 * SC_F2D_KY;
  const int cc = tid >> 2, cu = tid & 3;
  SC_SYNC();
  cf32 yh[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) yh[i] = cf_make(0.f, 0.f);
  if (ctask && cc < My) {
    const float s = (cc == 0) ? s_dc : s_other;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        const int fx = f2d_fx(4 * cu + i + 16 * k2);
        const int row = fx + Mx / 2;
        if (row >= 0 && row < Mx) yh[i * 4 + k2] = cf_scale(IN[row * My + cc], s);
      }
  }
  SC_SYNC();

#pragma unroll 1
  for (int a = 0; a < P; ++a) {
    // ---------------- 33 inverse column FFTs (64 points) -> T[b][c], rows h = P b + a ------------
    if (ctask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k1 = 4 * cu + i;
        cf32 g[4];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
          const int fx = f2d_fx(k1 + 16 * k2);
          int idx = (a * fx) % H;
          if (idx < 0) idx += H;
          g[k2] = cf_mul(yh[i * 4 + k2], cf_conj(twH[idx]));
        }
        radix4<+1>(g[0], g[1], g[2], g[3]);          // over k2 -> index u
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const cf32 val = (u == 0 || k1 == 0) ? g[u] : cf_mul(g[u], cf_conj(tw64[(u * k1) & 63]));
          xch[cc * SC_F2D_CS + u * 16 + k1] = val;
        }
      }
    }
    SC_WAVE_SYNC();                                 // the 4 lanes of a column share a wave
    if (ctask) {
      cf32 v[16], o[16];
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = xch[cc * SC_F2D_CS + cu * 16 + k1];
      fft16<+1>(v, o);                               // o[j] = U[cu + 4 j]
#pragma unroll
      for (int j = 0; j < 16; ++j) T[(cu + 4 * j) * SC_F2D_KY + cc] = o[j];
    }
    SC_SYNC();
    // ---------------- rows of group a: Hermitian-extended, zero-padded C2R, two rows packed -------
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const int p = r * 16 + f;
      const int bA = 2 * p, bB = 2 * p + 1;
      const cf32* ta = T + bA * SC_F2D_KY;
      const cf32* tb = T + bB * SC_F2D_KY;
      cf32 z[16], o[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) z[i] = cf_make(0.f, 0.f);
      // Z[k] = U~_A[k] + i U~_B[k],  U~[0] = Re U[0], U~[k] = U[k], U~[-k] = conj U[k]
      z[0] = (t == 0) ? cf_make(ta[0].x, tb[0].x) : cf_add_i(ta[t], tb[t]);
      z[1] = cf_add_i(ta[t + 16], tb[t + 16]);
      z[15] = cf_conj_add_i(ta[16 - t], tb[16 - t]);     // Z[t - 16]
      z[14] = cf_conj_add_i(ta[32 - t], tb[32 - t]);     // Z[t - 32]
      if (t == 0) z[2] = cf_add_i(ta[32], tb[32]);       // Z[32]
      fft16<+1>(z, o);                                    // over k2 -> n2
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) {
        const cf32 val = (n2 == 0) ? o[0] : cf_mul(o[n2], cf_conj(tw256[t * n2]));
        xch[(f * 16 + n2) * SC_F2D_XS + t] = val;
      }
      SC_WAVE_SYNC();                               // the 16 lanes of an FFT slot share a wave
      cf32 v[16];
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) v[k1] = xch[(f * 16 + t) * SC_F2D_XS + k1];
      fft16<+1>(v, o);                                    // o[n1] = z[t + 16 n1]
      float* ra = yo + (int64_t)(P * bA + a) * SC_F2D_W + t;
      float* rb = yo + (int64_t)(P * bB + a) * SC_F2D_W + t;
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        ra[16 * n1] = o[n1].x + badd;
        rb[16 * n1] = o[n1].y + badd;
      }
      SC_WAVE_SYNC();                               // xch is rewritten by the next round
    }
    SC_SYNC();                                      // T is rewritten by the next group
  }
}

// ------------------------------------------------------------------------------------------
// host side of the fast path
// ------------------------------------------------------------------------------------------
struct Fft2dPlan {
  int H = 0, Mx = 0, My = 0;
  float sf = 0.f, si = 0.f;
  cf32 *tabW = nullptr, *tabH = nullptr, *tab64 = nullptr;
  uint16_t* tabF = nullptr;     // bf16 operand fragments of the matrix-core row pass (sc_kernels_fft3mx.h); null = not built
  int mx_terms = 2;             // ... bf16 terms per twiddle in tabF (2: default of bf16-I/O plans, 3: SC_PLAN_MX_FFT_3TERM)
  uint16_t* tabG = nullptr;     // ... of the inverse-type kernel's row pass (k_fft2d_inv_mx); null = not built
};

static inline bool fft2d_upload(std::vector<void*>* owned, int n, cf32** out) {
  std::vector<cf32> h((size_t)n);
  const double two_pi = 6.283185307179586476925286766559;
  for (int m = 0; m < n; ++m) {
    const double th = two_pi * (double)m / (double)n;
    h[(size_t)m] = cf_make((float)std::cos(th), (float)(-std::sin(th)));
  }
  void* dev = nullptr;
  if (hipMalloc(&dev, h.size() * sizeof(cf32)) != hipSuccess) return false;
  owned->push_back(dev);
  if (hipMemcpy(dev, h.data(), h.size() * sizeof(cf32), hipMemcpyHostToDevice) != hipSuccess) return false;
  *out = (cf32*)dev;
  return true;
}

// true when the plan can take the fused kernels
static inline bool fft2d_plan_init(Fft2dPlan* fp, int nd, const int64_t* n, const int64_t* k, double sf,
                                   double si, std::vector<void*>* owned, std::string* why) {
  if (nd != 2) return false;
  const int64_t H = n[0], W = n[1];
  if (W != SC_F2D_W) return false;
  if (!(H == 64 || H == 128 || H == 256 || H == 512)) return false;
  if (k[0] > SC_F2D_KX || k[1] > SC_F2D_KY) return false;
  fp->H = (int)H;
  fp->Mx = (int)k[0];
  fp->My = (int)k[1];
  fp->sf = (float)sf;
  fp->si = (float)si;
  if (!fft2d_upload(owned, SC_F2D_W, &fp->tabW) || !fft2d_upload(owned, (int)H, &fp->tabH) ||
      !fft2d_upload(owned, 64, &fp->tab64)) {
    if (why) *why = "table upload failed";
    return false;
  }
  return true;
}

static inline size_t fft2d_workspace_bytes(const Fft2dPlan*, int64_t) { return 0; }

template <int H>
static void fft2d_launch_fwd(const Fft2dPlan* fp, const float* x, cf32* xhat, int64_t n_images, float s_dc,
                             float s_other, sc_stream_t st) {
  SC_LAUNCH((k_fft2d_fwd<H>), dim3((unsigned)n_images), dim3(256), 0, st, x, xhat, (const cf32*)fp->tabW,
            (const cf32*)fp->tabH, (const cf32*)fp->tab64, fp->Mx, fp->My, s_dc, s_other);
}

template <int H>
static void fft2d_launch_inv(const Fft2dPlan* fp, const cf32* yhat, float* y, const float* bias, int channels,
                             int64_t n_images, float s_dc, float s_other, sc_stream_t st) {
  SC_LAUNCH((k_fft2d_inv<H>), dim3((unsigned)n_images), dim3(256), 0, st, yhat, y, bias, channels,
            (const cf32*)fp->tabW, (const cf32*)fp->tabH, (const cf32*)fp->tab64, fp->Mx, fp->My, s_dc,
            s_other);
}

static inline int fft2d_forward(const Fft2dPlan* fp, int mode, const float* x, cf32* xhat, int64_t n_images,
                                void*, sc_stream_t st, std::string* err) {
  // mode 0 = scaled forward; mode 1 = adjoint of the padded C2R (interior columns x2)
  const float s_dc = (mode == 0) ? fp->sf : fp->si;
  const float s_other = (mode == 0) ? fp->sf : 2.f * fp->si;
  switch (fp->H) {
    case 64: fft2d_launch_fwd<64>(fp, x, xhat, n_images, s_dc, s_other, st); break;
    case 128: fft2d_launch_fwd<128>(fp, x, xhat, n_images, s_dc, s_other, st); break;
    case 256: fft2d_launch_fwd<256>(fp, x, xhat, n_images, s_dc, s_other, st); break;
    case 512: fft2d_launch_fwd<512>(fp, x, xhat, n_images, s_dc, s_other, st); break;
    default: *err = "sc_engine: fft2d: unsupported H"; return 1;
  }
  if (hipGetLastError() != hipSuccess) {
    *err = "sc_engine: launch of k_fft2d_fwd failed";
    return 1;
  }
  return 0;
}

static inline int fft2d_inverse(const Fft2dPlan* fp, int mode, const cf32* yhat, const float* bias,
                                int64_t channels, float* y, int64_t n_images, void*, sc_stream_t st,
                                std::string* err) {
  // mode 0 = zero-padded C2R (Hermitian extension supplies the interior x2);
  // mode 1 = adjoint of the scaled R2C: every kept column weighs 1 -> halve the interior
  const float s_dc = (mode == 0) ? fp->si : fp->sf;
  const float s_other = (mode == 0) ? fp->si : 0.5f * fp->sf;
  switch (fp->H) {
    case 64: fft2d_launch_inv<64>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st); break;
    case 128: fft2d_launch_inv<128>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st); break;
    case 256: fft2d_launch_inv<256>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st); break;
    case 512: fft2d_launch_inv<512>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st); break;
    default: *err = "sc_engine: fft2d: unsupported H"; return 1;
  }
  if (hipGetLastError() != hipSuccess) {
    *err = "sc_engine: launch of k_fft2d_inv failed";
    return 1;
  }
  return 0;
}

static inline const char* fft2d_kernel_name(int which) { return which == 0 ? "k_fft2d_fwd" : "k_fft2d_inv"; }
