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
//   * the A operand of v_mfma_f32_16x16x32_bf16 wants 8 k-values per lane: lane (row i, group g) takes the 64
//     contiguous bytes 256 ks + 64 g of its row -- 32 consecutive samples -- and splits them by r with v_perm_b32
//     (16 per 64 bytes); the sum over m is order independent, so the only data movement between HBM and the matrix
//     core is the hop of whole rows through a per-wave LDS image (coalesced loads in, 16-byte pieces out);
//   * the 16 x 16 accumulator tile of a wave IS 16 rows x 16 frequencies: Y goes to the group tile T2[row][k]
//     unpacked (no two-rows-as-one-complex trick to undo) and the column phase of k_fft2d_fwd3 runs unchanged on it.
//
// Per image: 1536 MFMAs of 16 cycles (6.5 k cycles per compute unit), ~1.3 k vector instructions per wave for the
// row pass instead of ~9 k.  Two workgroups per compute unit (the operand fragments, the prefetched rows and the
// accumulators take 237 registers), persistent, the rows of the next group requested one group ahead.
// Measured (MI355X, metric shape, profiles/r05_mx_fft_ab.txt): 76-78 us against 94-98 us for k_fft2d_fwd3<256, sc_bf16>,
// 1.2e-7 against a float64 transform (the vector-ALU kernel: 1.3e-7).  At two waves per SIMD the matrix time (23.5 us),
// the vector / LDS time (~41 us) and the load wait (12.6 us) add up rather than overlap.
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
// Measured and dropped (profiles/r05_mx_fft_ab.txt): rows straight from global memory into MFMA order (non-temporal:
// 106 us, every lane's 16-byte piece its own request; ordinary loads: 76-84 us, but the step loses in its contractions
// what the transform gains -- see the row requests below), two groups of rows in flight with the operand fragments
// read from LDS (91 us), one r's fragments at a time to fit 168 registers (spills), the MFMAs of group a + 1
// interleaved with the column phase of group a (92-95 us).
#define SC_MX_RS 36       // row stride (complex) of the unpacked group tile T2[64 rows][33 columns + 3 parked k = 32 columns]:
                          // the accumulator stores and the column-phase reads are both conflict-free per half-wave
#define SC_MX_SS 144      // a wave's staged rows, four planes (r = n mod 4) of [16 rows][64 samples = 128 bytes + 16]: the
#define SC_MX_PS (16 * SC_MX_SS)   // 4-byte writes of a half-wave (one row of a plane) and the 16-byte reads of 16 rows at
                          // one offset are both conflict-free
#define SC_MX_WGS 2       // persistent workgroups per compute unit = the kernel's register budget (two waves per SIMD)

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
  static constexpr int off_stg = off_twr + 96 * 8;         // staged rows: [wave][16 rows][SC_MX_SS bytes]
  static constexpr int total = off_stg + 4 * 4 * SC_MX_PS;
  static_assert(P <= 4, "k = 32 of group a is parked in column 32 + a of its row (H <= 256)");
  static_assert(SC_F2D_KX * SC_F2D_KY <= T_c, "output tile aliases the group tile");
  static_assert(SC_MX_WGS * total <= 160 * 1024, "workgroups per compute unit");
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
  constexpr int P = H / 64, RS = SC_MX_RS, NT = SC_MX_TERMS;
  typedef F3MxLds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);
  cf32* twr = reinterpret_cast<cf32*>(smem + L::off_twr);  // (as 12 register constants per lane the kernel spilled)

  const int tid = SC_TID;
  const int w = SC_UNIFORM(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;                 // MFMA roles: (row i = j, k group g) of A, (column j, k group g) of B
  const int cl = lane >> 3, mu = lane & 7;                // column phase: wave w owns columns 8 w .. 8 w + 7, 8 lanes each
  // ---- tables: w_H, w64 and the row twiddles to LDS, the lane's operand fragments to registers
  for (int q = tid; q < H; q += 256) twH[q] = tabH[q];
  if (tid < 64) tw64[tid] = tabW[(4 * (tid >> 3) * (tid & 7)) & 255];          // w64^(mu q1), [mu][q1]
  if (tid < 96) twr[tid] = tabW[((tid / 32 + 1) * (2 * (tid & 15) + ((tid >> 4) & 1))) & 255];
  sc_mx_u4 F[4][NT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int term = 0; term < NT; ++term) F[t][term] = sc_mx_load16(tabF + ((size_t)(t * NT + term) * 64 + lane) * 8);
  const bool lane_k0 = (j == 0);

  // ---- one column task of group a (lane = (column slot, mu)), as in k_fft2d_fwd3 but on the unpacked tile: rows
  //      b = 8 b1 + mu; cb = the column's private exchange patch; F_a[q1 + 8 q2] w_H^(a fx) goes to acc (+=) or, for the
  //      deferred 33rd column, to dst[q]
  cf32 acc[8];
  auto column = [&](const cf32* src, cf32* cb, const int a, auto extra_tag, cf32* dst, const bool act) SC_ALWAYS_INLINE_LAMBDA {
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
    SC_WAVE_SYNC();                                        // the 8 lanes of a column share a wave
#pragma unroll
    for (int m = 0; m < 8; m += 2) SC_F3_LD128(cb + mu * 8 + m, v[m], v[m + 1]);   // lane mu now plays q1 = mu
    dft8<-1>(v, o);                                        // over mu -> q2 : F_a[q1 + 8 q2]
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) {
      const int idx = (a * f2d_fx(mu + 8 * q2)) & (H - 1);    // (H is a power of two)
      if (EXTRA) {
        if (act) dst[mu + 8 * q2] = cf_mul(twH[idx], o[q2]);
      } else {
        cf_mac(acc[q2], twH[idx], o[q2]);
      }
    }
    SC_WAVE_SYNC();                                        // cb is rewritten by the next task
  };

  // ---- row requests: group a of image im = rows h = P b + a, b = 16 w + i.  The MFMA A layout wants lane (i = j, g)
  //      to hold the 64 bytes 256 ks + 64 g of row i -- straight from global memory that is a 16-byte piece per lane with
  //      neighbouring lanes in different rows: as non-temporal loads 103-106 us (every piece its own request), as
  //      ordinary loads 76-84 us for the kernel but the 268 MB of x then pass through the Infinity Cache, evict the
  //      weights and cost the two contractions of a step what the transform gains (0.4178 vs 0.422 ms per bf16 step,
  //      profiles/r05_mx_fft_ab.txt).  So: whole rows by coalesced NON-TEMPORAL 16-byte loads (instruction q of a wave =
  //      its rows 2 q, 2 q + 1, a half-wave each: 512 contiguous bytes), through a per-wave LDS image, out in MFMA order.
  unsigned char* stg = smem + L::off_stg + w * (4 * SC_MX_PS);
  sc_mx_u4 xq[8];
#ifdef SC_MX_ABL_NOLOAD
  for (int q = 0; q < 8; ++q) xq[q] = sc_mx_u4{(uint32_t)(0x3f803f80u + lane), 0x3f80bf80u, 0x40003f00u, 0x3e803f80u};
#endif
  auto request = [&](const int64_t im, const int a) SC_ALWAYS_INLINE_LAMBDA {
    const int64_t imc = im < n_images ? im : n_images - 1;           // past the end: a harmless re-read
    const unsigned char* base = reinterpret_cast<const unsigned char*>(x + (imc * H + (P * (16 * w + (lane >> 5)) + a)) * SC_F2D_W) +
                                16 * (lane & 31);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#ifdef SC_MX_ABL_NOLOAD
      if (im == -12345) xq[q] = sc_mx_load16_stream(base + (size_t)q * (2 * P * SC_F2D_W * 2));   // measurement build only
#else
      xq[q] = sc_mx_load16_stream(base + (size_t)q * (2 * P * SC_F2D_W * 2));
#endif
    }
  };
  request(SC_BID_X, 0);
  SC_SYNC();                                               // tables

#pragma unroll 1
  for (int64_t img = SC_BID_X; img < n_images; img += gstride) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = cf_make(0.f, 0.f);
#pragma unroll 1
    for (int a = 0; a < P; ++a) {
      // ---------------- rows of group a on the matrix cores ----------------
      // registers -> the wave's LDS image, split by r = n mod 4 on the way: a loaded 16-byte piece = samples 8 c .. 8 c + 7
      // of its row = elements m = 2 c, 2 c + 1 of each r; (s_r, s_{r+4}) is one dword of plane r (v_perm_b32).  The
      // registers are free for the next group's request at once (the rows of the next image behind the last group).
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(stg + (2 * q + (lane >> 5)) * SC_MX_SS) + (lane & 31);
        dst[0 * (SC_MX_PS / 4)] = sc_mx_pick(xq[q].z, xq[q].x, 0);      // r = 0: low halves of dwords 0 and 2
        dst[1 * (SC_MX_PS / 4)] = sc_mx_pick(xq[q].z, xq[q].x, 1);      // r = 1: high halves
        dst[2 * (SC_MX_PS / 4)] = sc_mx_pick(xq[q].w, xq[q].y, 0);      // r = 2
        dst[3 * (SC_MX_PS / 4)] = sc_mx_pick(xq[q].w, xq[q].y, 1);      // r = 3
      }
      request(a + 1 < P ? img : img + gstride, a + 1 < P ? a + 1 : 0);     // (selects, no branch around the loads)
      SC_WAVE_SYNC();
      float yre[2][4], yim[2][4], ere[4], eim[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc_mx_f4 c[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v) c[t][v] = 0.f;
        // fragment [ks] = x[4 m + r], m = 32 ks + 8 g + e: 16 contiguous bytes of plane r, row j
        const sc_mx_u4 A0 = *reinterpret_cast<const sc_mx_u4*>(stg + r * SC_MX_PS + j * SC_MX_SS + 16 * g);
        const sc_mx_u4 A1 = *reinterpret_cast<const sc_mx_u4*>(stg + r * SC_MX_PS + j * SC_MX_SS + 64 + 16 * g);
        // second half of the m range: w64^(32 k) = (-1)^k -- the odd-k tiles take it negated
        sc_mx_u4 An = A1;
        An.x ^= 0x80008000u;
        An.y ^= 0x80008000u;
        An.z ^= 0x80008000u;
        An.w ^= 0x80008000u;
#pragma unroll
        for (int term = NT - 1; term >= 0; --term)                   // small terms first
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < 4; ++t)                                // four accumulators in turn: no back-to-back dependence
              sc_mfma_16x16x32_bf16(c[t], hf == 0 ? A0 : (t < 2 ? A1 : An), F[t][term]);
        // Y += w256^(r k) S_r (u = 0: k = 2 j, u = 1: k = 2 j + 1);  k = 32 (the Im column of k = 0): E += w8^r S_r[32]
        constexpr float h = 0.70710678118654752440f;
        const float er = (r == 0) ? 1.f : (r == 1) ? h : (r == 2) ? 0.f : -h;
        const float ei = (r == 0) ? 0.f : (r == 1) ? -h : (r == 2) ? -1.f : -h;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          cf32 tw = cf_make(1.f, 0.f);
          if (r > 0) tw = SC_F3_LD64(twr + (r > 0 ? r - 1 : 0) * 32 + u * 16 + j);
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
      SC_SYNC();                                           // the column phase of the group before has read T2 (and every
                                                           // lane of the wave its fragments: the image is free)
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
      column(T + (8 * w + cl), xch + (8 * w + cl) * SC_F3_CCS, a, sc_int<0>(), nullptr, true);
    }
    SC_SYNC();
    // ---------------- 33rd column (k = 32): one 8-lane task per group, all groups at once ----------
    cf32* part = xch + (P + 1) * SC_F3_CCS;                // [P][64] partial spectra, summed below
    if (My > 32) {
      constexpr int ABLK = (P + 3) / 4;
      const int a = ((cl % ABLK) << 2) | w;
      const int ac = a < P ? a : 0;
      column(T + 32 + ac, xch + ac * SC_F3_CCS, ac, sc_int<1>(), part + ac * 64, cl < ABLK && a < P);
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
  }
}

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
