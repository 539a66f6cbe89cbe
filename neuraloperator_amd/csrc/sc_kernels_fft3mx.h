// sc_kernels_fft3mx.h -- forward-type fused 2-D transform for bfloat16 real tensors with the ROW pass on the
// matrix cores (round 5, BASELINE configs[1] "bf16").
//
// With bf16 storage the fused kernel of sc_kernels_fft3.h is bound by its arithmetic (70 us with the loads removed
// against a 50 us byte floor at the metric shape), four fifths of it the 256-point row transforms.  A bf16
// input is EXACT in the matrix cores' input format, so the row pass can be a bf16 MFMA product with fp32
// accumulation that loses nothing as long as the twiddles carry enough bits:
//
//   x[4 m + r]  (row of 256 real points, r = 0..3, m = 0..63)
//   S_r[k] = sum_m x[4 m + r] w64^(m k)          four 64-point DFTs per row:  [16 rows x 64] . F64[64 x 64 columns]
//   Y[k]   = sum_r w256^(r k) S_r[k],  k = 0..32 twelve fp32 FMAs per output on the vector ALUs
//
//   * F64 = (cos, -sin) of 2 pi m k / 64 as THREE bf16 terms (hi + mid + lo = 24 bits): every product x * F_term is
//     exact in fp32, the sum is an fp32 accumulation like any FFT's -- results of fp32 round-off class;
//   * the 64 MFMA columns are four tiles of 16: Re S[even k], Im S[even k], Re S[odd k], Im S[odd k], k < 32; Im S[0] is
//     identically zero and its column carries S[32] = sum_m (-1)^m x[4 m + r] instead (real, even), so k = 32 costs no
//     tile of its own;
//   * w64^((m + 32) k) = (-1)^k w64^(m k): both halves of the m range use the SAME 32 x 64 operand (48 registers per
//     lane for the three terms) -- the even tiles add the two halves in one accumulator, the odd tiles take the second
//     half with its sign bits flipped (one v_xor per register of the fragment);
//   * the A operand of v_mfma_f32_16x16x32_bf16 wants 8 k-values per lane: lane (row i, group g) loads the 64
//     contiguous bytes 256 ks + 64 g of its row -- 32 consecutive samples -- and splits them by r with v_perm_b32
//     (16 per 64 bytes); the sum over m is order independent, so no other data movement exists between HBM and the
//     matrix core: no LDS staging, no transposes;
//   * the 16 x 16 accumulator tile of a wave IS 16 rows x 16 frequencies: Y goes to the group tile T2[row][k]
//     unpacked (no two-rows-as-one-complex trick to undo) and the column phase of k_fft2d_fwd3 runs unchanged on it.
//
// Per image: 1536 MFMAs of 16 passes (6.5 k cycles per compute unit), ~1.3 k vector instructions per wave for the
// row pass instead of ~9 k.  Two workgroups per compute unit (the operand tables, the prefetched rows and the
// accumulators take ~200 registers), persistent, the rows of the next group requested one group ahead.
// Reference lines: spectral_convolution.py:443-449 (rfftn), :500-519 (kept block).
#pragma once
#include "sc_kernels_fft3.h"

#ifndef SC_EMU
typedef __bf16 sc_mx_bf8 __attribute__((ext_vector_type(8)));  /* (measurement builds: -DSC_MX_ABL_NOMFMA) */
typedef float sc_mx_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t sc_mx_u4 __attribute__((ext_vector_type(4)));
SC_DEVICE void sc_mfma_16x16x32_bf16(sc_mx_f4& acc, const sc_mx_u4 a, const sc_mx_u4 b) {
#ifdef SC_MX_ABL_NOMFMA
  acc[0] += __uint_as_float((a.x ^ b.x) & 0x3fffffffu);
  return;
#endif
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sc_mx_bf8, a), __builtin_bit_cast(sc_mx_bf8, b), acc,
                                                0, 0, 0);
}
// {lo.half(h), hi.half(h)} as one dword: v_perm_b32
SC_DEVICE uint32_t sc_mx_pick(const uint32_t hi, const uint32_t lo, const int h) {
  return __builtin_amdgcn_perm(hi, lo, h ? 0x07060302u : 0x05040100u);
}
SC_DEVICE sc_mx_u4 sc_mx_load16_stream(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const sc_mx_u4*>(p));
}
SC_DEVICE sc_mx_u4 sc_mx_load16(const void* p) { return *reinterpret_cast<const sc_mx_u4*>(p); }
#else
struct sc_mx_f4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
struct alignas(16) sc_mx_u4 {
  uint32_t x, y, z, w;
};
namespace scemu {
inline uint16_t g_mx_a[16][64][8];
inline uint16_t g_mx_b[16][64][8];
}  // namespace scemu
// lane l supplies A[i = l & 15][k = 8 (l >> 4) + e] and B[k = 8 (l >> 4) + e][j = l & 15], e = 0..7;
// it owns D[row = 4 (l >> 4) + v][col = l & 15], v = 0..3  (cdna_hip_programming.md 3)
inline void sc_mfma_16x16x32_bf16(sc_mx_f4& acc, const sc_mx_u4 a, const sc_mx_u4 b) {
  const int w = SC_TID >> 6, l = SC_TID & 63;
  std::memcpy(scemu::g_mx_a[w][l], &a, 16);
  std::memcpy(scemu::g_mx_b[w][l], &b, 16);
  scemu::wave_barrier();
  for (int v = 0; v < 4; ++v) {
    const int row = 4 * (l >> 4) + v, col = l & 15;
    float c = acc[v];
    for (int g = 0; g < 4; ++g)
      for (int e = 0; e < 8; ++e)
        c = fmaf(sc_bits_to_f32((uint32_t)scemu::g_mx_a[w][row + 16 * g][e] << 16),
                 sc_bits_to_f32((uint32_t)scemu::g_mx_b[w][col + 16 * g][e] << 16), c);
    acc[v] = c;
  }
  scemu::wave_barrier();
}
inline uint32_t sc_mx_pick(const uint32_t hi, const uint32_t lo, const int h) {
  return h ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
}
inline sc_mx_u4 sc_mx_load16_stream(const void* p) {
  sc_mx_u4 v;
  std::memcpy(&v, p, 16);
  return v;
}
inline sc_mx_u4 sc_mx_load16(const void* p) { return sc_mx_load16_stream(p); }
#endif

#ifndef SC_MX_TERMS
#define SC_MX_TERMS 3     // bf16 terms of a twiddle (3: 24 bits; 2 would leave 2^-17 -- measured)
#endif
// how the rows reach the MFMA A layout (lane (row i, group g) wants the 64 contiguous bytes 256 ks + 64 g of row i):
//   2  full rows by coalesced 16-byte loads (a half-wave = one 512-byte row), through a per-wave LDS image whose row
//      stride of 528 bytes makes both the 16-byte writes and the 16-byte reads in MFMA order conflict-free;
//   0 / 1  straight from global memory (non-temporal / ordinary): every lane of an instruction then touches its own
//      16-byte piece, neighbouring lanes in different rows -- measured 106 us against 93 us for the vector-ALU kernel
//      (profiles/r05_mx_fft_ab.txt)
#ifndef SC_MX_LOADS
#define SC_MX_LOADS 1
#endif
#define SC_MX_SS 528
// groups of rows in flight per wave ahead of the one being transformed.  One group (8 KB per wave, 64 KB per compute
// unit) leaves the kernel waiting for memory: 86 us with real loads against NN us without (profiles/r05_mx_fft_ab.txt).
// Two need 32 more registers: the operand fragments then come from LDS (SC_MX_F_LDS), 12 KB per workgroup.
#ifndef SC_MX_PF
#define SC_MX_PF 1
#endif
// SC_MX_LAZY: the fragments of one r at a time (8 registers instead of 32) straight out of the loaded rows, the next
// group's request behind the MFMA phase instead of ahead of it -- what lets a THIRD workgroup per compute unit fit
// (168 registers).  The kernel is bound by the issue rate of its vector / LDS instructions between the MFMA phases
// (two waves per SIMD: 78 us with real loads, 74 without, 65 without MFMAs), not by memory: occupancy is the lever.
#ifndef SC_MX_LAZY
#define SC_MX_LAZY 0
#endif
#ifndef SC_MX_WGS
#define SC_MX_WGS (SC_MX_LAZY ? 3 : 2)
#endif
#ifndef SC_MX_F_LDS
#define SC_MX_F_LDS (SC_MX_PF == 2)
#endif
#define SC_MX_RS 36       // row stride (complex) of the unpacked group tile T2[64 rows][33 columns + 3 parked k = 32 columns]:
                          // the accumulator stores and the column-phase reads are both conflict-free per half-wave

template <int H>
struct F3MxLds {
  static constexpr int P = H / 64;
  static constexpr int xch_c = 33 * SC_F3_CCS > (P + 1) * SC_F3_CCS + P * 64 ? 33 * SC_F3_CCS : (P + 1) * SC_F3_CCS + P * 64;
  static constexpr int T_c = 64 * SC_MX_RS;              // 2304 complex; the output tile (<= 64 x 33) aliases it
  static constexpr int off_xch = 0;
  static constexpr int off_T = off_xch + xch_c * 8;
  static constexpr int off_twH = off_T + T_c * 8;
  static constexpr int off_tw64 = off_twH + H * 8;
  static constexpr int off_twr = off_tw64 + 64 * 8;        // w256^(r k), [r - 1][u][j], k = 2 j + u
  static constexpr int off_F = off_twr + 96 * 8;           // operand fragments [tile][term][lane] (SC_MX_F_LDS)
  static constexpr int off_stg = off_F + (SC_MX_F_LDS ? 4 * SC_MX_TERMS * 64 * 16 : 0);   // staged rows: [wave][16 rows][SC_MX_SS bytes]
  static constexpr int total = off_stg + (SC_MX_LOADS == 2 ? 4 * 16 * SC_MX_SS : 0);
  static_assert(SC_MX_PF == 1 || SC_MX_LOADS != 2, "two groups in flight: the direct load path");
  static_assert(SC_MX_WGS * total <= 160 * 1024, "workgroups per compute unit");
  static_assert(!SC_MX_LAZY || SC_MX_LOADS != 2, "lazy fragments: the direct load path");
  static_assert(P <= 4, "k = 32 of group a is parked in column 32 + a of its row (H <= 256)");
  static_assert(SC_F2D_KX * SC_F2D_KY <= T_c, "output tile aliases the group tile");
};

// the operand table of one plan: [tile 4][term][lane 64][8 bf16], MFMA B layout (lane (j, g): F[m = 8 g + e][column j]);
// tile 0 / 2: cos(2 pi m k / 64), tile 1 / 3: -sin(...), k = 2 j + (tile >> 1); tile 1, j = 0: (-1)^m  (k = 32)
static inline void fft3mx_build_table(std::vector<uint16_t>* out) {
  const double two_pi = 6.283185307179586476925286766559;
  out->assign((size_t)4 * SC_MX_TERMS * 64 * 8, 0);
  // nearest-even onto the bf16 grid (through float: a double rounding can only move a tie, and whatever a term
  // misses the next term picks up -- the remainder below is exact in double)
  auto to_bf16 = [](double v) {
    const float f = (float)v;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  auto from_bf16 = [](uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return (double)f;
  };
  for (int t = 0; t < 4; ++t)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int j = lane & 15, g = lane >> 4, m = 8 * g + e, k = 2 * j + (t >> 1);
        double v;
        if (t == 1 && j == 0) {
          v = (m & 1) ? -1.0 : 1.0;
        } else {
          const int idx = (m * k) & 63;                    // exact phase reduction
          const double th = two_pi * (double)idx / 64.0;
          v = (t & 1) ? -std::sin(th) : std::cos(th);
          if (idx % 16 == 0) v = std::round(v);            // 0, +-1 exactly
        }
        double rest = v;
        for (int term = 0; term < SC_MX_TERMS; ++term) {
          const uint16_t b = to_bf16(rest);
          (*out)[(((size_t)t * SC_MX_TERMS + term) * 64 + lane) * 8 + e] = b;
          rest -= from_bf16(b);
        }
      }
}

template <int H>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, SC_MX_WGS)
k_fft2d_fwd_mx(const sc_bf16* __restrict__ x, cf32* __restrict__ xhat, const cf32* __restrict__ tabW,
               const cf32* __restrict__ tabH, const uint16_t* __restrict__ tabF, int Mx, int My, float s_dc,
               float s_other, F3Shard sh, int64_t n_images, int gstride) {
  constexpr int P = H / 64, RS = SC_MX_RS;
  typedef F3MxLds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);

  const int tid = SC_TID;
  const int w = SC_UNIFORM(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;                 // MFMA roles: (row i = j, k group g) of A, (column j, k group g) of B
  // ---- tables: w_H and w64 to LDS, the lane's operand fragments and row twiddles to registers
  for (int q = tid; q < H; q += 256) twH[q] = tabH[q];
  if (tid < 64) tw64[tid] = tabW[(4 * (tid >> 3) * (tid & 7)) & 255];          // w64^(mu q1), [mu][q1]
#if SC_MX_F_LDS
  sc_mx_u4* Fl = reinterpret_cast<sc_mx_u4*>(smem + L::off_F);
  for (int q = tid; q < 4 * SC_MX_TERMS * 64; q += 256) Fl[q] = sc_mx_load16(tabF + (size_t)q * 8);
  Fl += lane;
#else
  sc_mx_u4 F[4][SC_MX_TERMS];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int term = 0; term < SC_MX_TERMS; ++term)
      F[t][term] = sc_mx_load16(tabF + ((size_t)(t * SC_MX_TERMS + term) * 64 + lane) * 8);
#endif
  cf32* twr = reinterpret_cast<cf32*>(smem + L::off_twr);  // (12 registers per lane as constants: the kernel spilled)
  if (tid < 96) twr[tid] = tabW[((tid / 32 + 1) * (2 * (tid & 15) + ((tid >> 4) & 1))) & 255];
  const bool lane_k0 = (j == 0);

  // ---- column phase roles (as k_fft2d_fwd3): wave w owns columns 8 w .. 8 w + 7, 8 lanes per column
  const int cl = lane >> 3, mu = lane & 7;
  auto column = [&](const cf32* src, cf32* cb, const int a, auto extra_tag, cf32* dst, cf32 (&acc)[8], const bool act) {
    constexpr bool EXTRA = decltype(extra_tag)::value != 0;
    cf32 v[8], o[8];
#pragma unroll
    for (int b1 = 0; b1 < 8; ++b1) v[b1] = SC_F3_LD64(src + (8 * b1 + mu) * RS);
    dft8<-1>(v, o);                                        // over b1 -> q1
#pragma unroll
    for (int q1 = 0; q1 < 8; ++q1) {
      const cf32 y = (q1 == 0) ? o[0] : cf_mul_pk(o[q1], tw64[mu * 8 + q1]);
      if (act) cb[q1 * 8 + mu] = y;
    }
    SC_WAVE_SYNC();
#pragma unroll
    for (int m = 0; m < 8; m += 2) SC_F3_LD128(cb + mu * 8 + m, v[m], v[m + 1]);
    dft8<-1>(v, o);                                        // over mu -> q2 : F_a[q1 + 8 q2]
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) {
      const int fx = f2d_fx(mu + 8 * q2);
      int idx = (a * fx) % H;
      if (idx < 0) idx += H;
      if (EXTRA) {
        if (act) dst[mu + 8 * q2] = cf_mul(twH[idx], o[q2]);
      } else {
        cf_mac(acc[q2], twH[idx], o[q2]);
      }
    }
    SC_WAVE_SYNC();
  };

  // ---- row requests: group a of image im = rows h = P b + a, b = 16 w + i
  constexpr int PF = SC_MX_PF;
  sc_mx_u4 xq[PF][8];
#ifdef SC_MX_ABL_NOLOAD
  for (int d = 0; d < PF; ++d)
    for (int q = 0; q < 8; ++q) xq[d][q] = sc_mx_u4{(uint32_t)(0x3f803f80u + lane), 0x3f80bf80u, 0x40003f00u, 0x3e803f80u};
#endif
#if SC_MX_LOADS == 2
  // instruction q of a wave = its rows i = 2 q, 2 q + 1 (a half-wave each: 512 contiguous bytes)
  unsigned char* stg = smem + L::off_stg + w * (16 * SC_MX_SS);
  auto request = [&](const int64_t im, const int a, sc_mx_u4 (&xr)[8]) {
    const int64_t imc = im < n_images ? im : n_images - 1;           // past the end: a harmless re-read
    const unsigned char* base = reinterpret_cast<const unsigned char*>(x + (imc * H + (P * (16 * w + (lane >> 5)) + a)) * SC_F2D_W) +
                                16 * (lane & 31);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#ifdef SC_MX_ABL_NOLOAD
      if (im == -12345) xr[q] = sc_mx_load16_stream(base + (size_t)q * (2 * P * SC_F2D_W * 2));   // measurement build only
#else
      xr[q] = sc_mx_load16_stream(base + (size_t)q * (2 * P * SC_F2D_W * 2));
#endif
    }
  };
#else
  // lane (i = j, g) takes the 64 bytes 256 ks + 64 g of its row for ks = 0, 1 (samples 128 ks + 32 g .. + 31): the
  // four lanes of a row cover 256 contiguous bytes per ks, an instruction 16-byte pieces of 32 cache lines -- ordinary
  // loads, so that the line a piece misses on serves the seven pieces that follow from L1
  auto request = [&](const int64_t im, const int a, sc_mx_u4 (&xr)[8]) {
    const int64_t imc = im < n_images ? im : n_images - 1;           // past the end: a harmless re-read
    const unsigned char* row = reinterpret_cast<const unsigned char*>(x + (imc * H + (P * (16 * w + j) + a)) * SC_F2D_W);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#ifdef SC_MX_ABL_NOLOAD
        if (im == -12345) xr[4 * ks + q] = sc_mx_load16(row + 256 * ks + 64 * g + 16 * q);       // measurement build only
#else
        xr[4 * ks + q] = SC_MX_LOADS == 0 ? sc_mx_load16_stream(row + 256 * ks + 64 * g + 16 * q)
                                           : sc_mx_load16(row + 256 * ks + 64 * g + 16 * q);
#endif
      }
  };
#endif
  // the group PF steps behind (img, a) in the workgroup's sequence of groups
  auto ahead_img = [&](const int64_t img, const int a) { return img + (int64_t)((a + PF) / P) * gstride; };
  auto ahead_a = [&](const int a) { return (a + PF) % P; };
  request(SC_BID_X, 0, xq[0]);
  if (PF == 2) request(P > 1 ? SC_BID_X : SC_BID_X + gstride, P > 1 ? 1 : 0, xq[PF - 1]);
  SC_SYNC();                                               // tables

  cf32 acc[8];
  auto group = [&](const int64_t img, const int a, sc_mx_u4 (&xr)[8]) SC_ALWAYS_INLINE_LAMBDA {
      // ---------------- rows of group a on the matrix cores ----------------
      // split the lane's 2 x 32 samples by r = n mod 4: fragment [r][ks] = x[4 m + r], m = 32 ks + 8 g + e
#if !SC_MX_LAZY
      sc_mx_u4 A[4][2];
#if SC_MX_LOADS == 2
      // registers -> the wave's LDS image -> MFMA order; the registers are free for the next group's request at once
#pragma unroll
      for (int q = 0; q < 8; ++q)
        *reinterpret_cast<sc_mx_u4*>(stg + (2 * q + (lane >> 5)) * SC_MX_SS + 16 * (lane & 31)) = xr[q];
      request(ahead_img(img, a), ahead_a(a), xr);          // (selects, no branch around the loads)
      SC_WAVE_SYNC();
#endif
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        sc_mx_u4 dq[4];
#if SC_MX_LOADS == 2
#pragma unroll
        for (int q = 0; q < 4; ++q)
          dq[q] = *reinterpret_cast<const sc_mx_u4*>(stg + j * SC_MX_SS + 256 * ks + 64 * g + 16 * q);
#else
#pragma unroll
        for (int q = 0; q < 4; ++q) dq[q] = xr[4 * ks + q];
#endif
        const uint32_t d[16] = {dq[0].x, dq[0].y, dq[0].z, dq[0].w, dq[1].x, dq[1].y, dq[1].z, dq[1].w,
                                dq[2].x, dq[2].y, dq[2].z, dq[2].w, dq[3].x, dq[3].y, dq[3].z, dq[3].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // element e sits at local sample 4 e + r = dword 2 e + (r >> 1), half r & 1
          A[r][ks].x = sc_mx_pick(d[2 + (r >> 1)], d[0 + (r >> 1)], r & 1);
          A[r][ks].y = sc_mx_pick(d[6 + (r >> 1)], d[4 + (r >> 1)], r & 1);
          A[r][ks].z = sc_mx_pick(d[10 + (r >> 1)], d[8 + (r >> 1)], r & 1);
          A[r][ks].w = sc_mx_pick(d[14 + (r >> 1)], d[12 + (r >> 1)], r & 1);
        }
      }
#if SC_MX_LOADS == 2
      SC_WAVE_SYNC();                                      // the image is rewritten at the next group
#else
      // the rows of the group PF steps on (of a later image behind the last group) while this one is transformed
      request(ahead_img(img, a), ahead_a(a), xr);          // (selects, no branch around the loads)
#endif
#endif
      float yre[2][4], yim[2][4], ere[4], eim[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc_mx_f4 c[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v) c[t][v] = 0.f;
#if SC_MX_LAZY
        // fragments of this r: element e of k step ks sits at local sample 4 e + r = dword 2 e + (r >> 1), half r & 1
        sc_mx_u4 Ar[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint32_t d[16] = {xr[4 * ks].x, xr[4 * ks].y, xr[4 * ks].z, xr[4 * ks].w,
                                  xr[4 * ks + 1].x, xr[4 * ks + 1].y, xr[4 * ks + 1].z, xr[4 * ks + 1].w,
                                  xr[4 * ks + 2].x, xr[4 * ks + 2].y, xr[4 * ks + 2].z, xr[4 * ks + 2].w,
                                  xr[4 * ks + 3].x, xr[4 * ks + 3].y, xr[4 * ks + 3].z, xr[4 * ks + 3].w};
          Ar[ks].x = sc_mx_pick(d[2 + (r >> 1)], d[0 + (r >> 1)], r & 1);
          Ar[ks].y = sc_mx_pick(d[6 + (r >> 1)], d[4 + (r >> 1)], r & 1);
          Ar[ks].z = sc_mx_pick(d[10 + (r >> 1)], d[8 + (r >> 1)], r & 1);
          Ar[ks].w = sc_mx_pick(d[14 + (r >> 1)], d[12 + (r >> 1)], r & 1);
        }
        const sc_mx_u4 A0 = Ar[0], A1 = Ar[1];
#else
        const sc_mx_u4 A0 = A[r][0], A1 = A[r][1];
#endif
        // second half of the m range: w64^(32 k) = (-1)^k -- the odd-k tiles take it negated
        sc_mx_u4 An = A1;
        An.x ^= 0x80008000u;
        An.y ^= 0x80008000u;
        An.z ^= 0x80008000u;
        An.w ^= 0x80008000u;
#pragma unroll
        for (int term = SC_MX_TERMS - 1; term >= 0; --term)          // small terms first
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < 4; ++t)                                // four accumulators in turn: no back-to-back dependence
#if SC_MX_F_LDS
              sc_mfma_16x16x32_bf16(c[t], hf == 0 ? A0 : (t < 2 ? A1 : An), Fl[(t * SC_MX_TERMS + term) * 64]);
#else
              sc_mfma_16x16x32_bf16(c[t], hf == 0 ? A0 : (t < 2 ? A1 : An), F[t][term]);
#endif
        // Y += w256^(r k) S_r (u = 0: k = 2 j, u = 1: k = 2 j + 1);  k = 32 (the Im column of k = 0): E += w8^r S_r[32]
        constexpr float h = 0.70710678118654752440f;
        const float er = (r == 0) ? 1.f : (r == 1) ? h : (r == 2) ? 0.f : -h;
        const float ei = (r == 0) ? 0.f : (r == 1) ? -h : (r == 2) ? -1.f : -h;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          cf32 tw = cf_make(1.f, 0.f);
          if (r > 0) tw = SC_F3_LD64(twr + (r - 1) * 32 + u * 16 + j);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float sre = c[2 * u][v], sim = c[2 * u + 1][v];
            if (r == 0) {
              yre[u][v] = sre;
              yim[u][v] = sim;
            } else {
              yre[u][v] = fmaf(tw.x, sre, yre[u][v]);
              yre[u][v] = fmaf(-tw.y, sim, yre[u][v]);
              yim[u][v] = fmaf(tw.x, sim, yim[u][v]);
              yim[u][v] = fmaf(tw.y, sre, yim[u][v]);
            }
            if (u == 0) {
              if (r == 0) {
                ere[v] = sim;
                eim[v] = 0.f;
              } else {
                ere[v] = fmaf(er, sim, ere[v]);
                eim[v] = fmaf(ei, sim, eim[v]);
              }
            }
          }
        }
        SC_SCHED_BARRIER();                                // one r at a time: two sets of accumulators do not fit
      }
#if SC_MX_LAZY
      request(ahead_img(img, a), ahead_a(a), xr);          // the rows are consumed: the next group's while the columns run
#endif
      SC_SYNC();                                           // the column phase of the group before has read T2
      {
        cf32* tr = T + (16 * w + 4 * g) * RS;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          tr[v * RS + 2 * j] = cf_make(yre[0][v], lane_k0 ? 0.f : yim[0][v]);      // (adjacent: one 16-byte store)
          tr[v * RS + 2 * j + 1] = cf_make(yre[1][v], yim[1][v]);
          if (lane_k0) tr[v * RS + 32 + a] = cf_make(ere[v], eim[v]);
        }
      }
      SC_SYNC();
      // ---------------- 32 column FFTs of 64 points on T2 (k = 32: behind the group loop) ----------------
#ifndef SC_MX_ABL_NOCOL
      column(T + (8 * w + cl), xch + (8 * w + cl) * SC_F3_CCS, a, sc_int<0>(), nullptr, acc, true);
#else
      acc[a & 7].x += T[tid].x;                            // measurement build only
#endif
  };
  auto finish = [&](const int64_t img) SC_ALWAYS_INLINE_LAMBDA {
    SC_SYNC();
    // ---------------- 33rd column (k = 32): one 8-lane task per group, all groups at once ----------
    cf32* part = xch + (P + 1) * SC_F3_CCS;                // [P][64] partial spectra, summed below
    if (My > 32) {
      constexpr int ABLK = (P + 3) / 4;
      const int a = ((cl % ABLK) << 2) | w;
      const int ac = a < P ? a : 0;
      column(T + 32 + ac, xch + ac * SC_F3_CCS, ac, sc_int<1>(), part + ac * 64, acc, cl < ABLK && a < P);
    }
    SC_SYNC();
    // ---------------- kept block -> LDS -> one contiguous store ----------------
    cf32* OUT = T;
    {
      const int c = 8 * w + cl;
      if (c < My) {
        const float s = (c == 0) ? s_dc : s_other;
#pragma unroll
        for (int q2 = 0; q2 < 8; ++q2) {
          const int row = f2d_fx(mu + 8 * q2) + Mx / 2;
          if (row >= 0 && row < Mx) OUT[row * My + c] = cf_scale(acc[q2], s);
        }
      }
      if (tid < 64 && 32 < My) {
        const int row = f2d_fx(tid) + Mx / 2;
        cf32 t = part[tid];
#pragma unroll
        for (int a = 1; a < P; ++a) t = cf_add(t, part[a * 64 + tid]);
        if (row >= 0 && row < Mx) OUT[row * My + 32] = cf_scale(t, s_other);
      }
    }
    SC_SYNC();
    if (sh.rows <= 0) {
      cf32* dst = xhat + img * (int64_t)Mx * My;
      for (int i = tid; i < Mx * My; i += 256) dst[i] = OUT[i];
    } else {                                               // sharded spectrum (include/sc_engine.h, sc_spectrum_shards)
      for (int i = tid; i < Mx * My; i += 256) xhat[f3_shard_index(sh, img, i, My)] = OUT[i];
    }
    // (the next image's first T2 write sits behind a workgroup barrier: OUT has been read by then)
  };
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = cf_make(0.f, 0.f);
  };
  if constexpr (PF == 1 || P % 2 == 0) {
#pragma unroll 1
    for (int64_t img = SC_BID_X; img < n_images; img += gstride) {
      clear();
#pragma unroll 1
      for (int a = 0; a < P; a += PF) {
        group(img, a, xq[0]);
        if (PF == 2) group(img, a + 1, xq[PF - 1]);
      }
      finish(img);
    }
  } else {                                                 // one group per image: the two register sets alternate over images
#pragma unroll 1
    for (int64_t img = SC_BID_X; img < n_images; img += 2 * (int64_t)gstride) {
      clear();
      group(img, 0, xq[0]);
      finish(img);
      if (img + gstride < n_images) {                      // uniform
        clear();
        group(img + gstride, 0, xq[PF - 1]);
        finish(img + gstride);
      }
    }
  }
}

// persistent: SC_MX_WGS workgroups per compute unit (the kernel's register budget)
template <int H>
static void fft3mx_launch_fwd(const Fft2dPlan* fp, const sc_bf16* x, cf32* xhat, int64_t n_images, float s_dc,
                              float s_other, sc_stream_t st, F3Shard sh) {
  int64_t grid = (int64_t)SC_MX_WGS * sc_cu_count();
  if (grid > n_images) grid = n_images;
  SC_LAUNCH((k_fft2d_fwd_mx<H>), dim3((unsigned)grid), dim3(256), 0, st, x, xhat, (const cf32*)fp->tabW,
            (const cf32*)fp->tabH, (const uint16_t*)fp->tabF, fp->Mx, fp->My, s_dc, s_other, sh, n_images, (int)grid);
}

static inline int fft3mx_forward(const Fft2dPlan* fp, int mode, const sc_bf16* x, cf32* xhat, int64_t n_images,
                                 sc_stream_t st, std::string* err, F3Shard sh = F3Shard{0, 0}) {
  const float s_dc = (mode == 0) ? fp->sf : fp->si;
  const float s_other = (mode == 0) ? fp->sf : 2.f * fp->si;
  switch (fp->H) {
    case 64: fft3mx_launch_fwd<64>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    case 128: fft3mx_launch_fwd<128>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    case 256: fft3mx_launch_fwd<256>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    default: *err = "sc_engine: fft2d (matrix-core row pass): unsupported H"; return 1;
  }
  if (hipGetLastError() != hipSuccess) {
    *err = "sc_engine: launch of k_fft2d_fwd_mx failed";
    return 1;
  }
  return 0;
}
