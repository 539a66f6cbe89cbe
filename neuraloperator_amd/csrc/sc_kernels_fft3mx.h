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

// bf16 terms of a twiddle of the forward-type kernel (Fft2dPlan::mx_terms, round 6): 2 = the default of bfloat16-I/O plans
// (16 bits: the spectrum 1.3e-6 from a float64 transform of the same bf16 values -- 3000 times below the 2^-9 the bf16
// INPUT already carries per value and 8 times below the 1e-5 bar of the fp32 gradients it feeds; 72-75 us per launch at
// the metric shape, bf16 step 0.395 -> 0.368 ms, profiles/r06_mx_terms_ab.txt); 3 = SC_PLAN_MX_FFT_3TERM (24 bits: fp32
// round-off class, 1.2e-7, 81-83 us)
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
static inline void fft3mx_build_table(std::vector<uint16_t>* out, const int n_terms = 3) {
  const double two_pi = 6.283185307179586476925286766559;
  out->assign((size_t)4 * n_terms * 64 * 8, 0);
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
        for (int term = 0; term < n_terms; ++term) {
          const uint16_t b = to_bf16(rest);
          (*out)[(((size_t)t * n_terms + term) * 64 + lane) * 8 + e] = b;
          rest -= from_bf16(b);
        }
      }
}

template <int H, int NT>
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
  if (fp->mx_terms == 2)
    SC_LAUNCH((k_fft2d_fwd_mx<H, 2>), dim3((unsigned)grid), dim3(256), 0, st, x, xhat, (const cf32*)fp->tabW,
              (const cf32*)fp->tabH, (const uint16_t*)fp->tabF, fp->Mx, fp->My, s_dc, s_other, sh, n_images, (int)grid);
  else
    SC_LAUNCH((k_fft2d_fwd_mx<H, 3>), dim3((unsigned)grid), dim3(256), 0, st, x, xhat, (const cf32*)fp->tabW,
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

// ------------------------------------------------------------------------------------------
// inverse-type transform WRITING bfloat16, row pass on the matrix cores (round 5, session 2)
// ------------------------------------------------------------------------------------------
// k_fft2d_inv3<H, sc_bf16> is bound by its arithmetic as well (84-92 us by box for 303 MB at the metric shape), again
// mostly the 256-point row transforms: a zero-padded, Hermitian-extended C2R of 33 coefficients per row.  As a product:
//
//   y[n] = Re c_0 + bias + 2 sum_{k = 1..32} Re(c_k exp(2 pi i k n / 256)),   n = 4 m + r,  r = 0..3,  m = 0..63
//
//   * per r: A = the coefficients of 16 rows ([16 rows] x [K = 32]: the 16 even k as (Re, Im) pairs, or the 16 odd k),
//     B = f cos / -f sin of 2 pi k (4 m + r) / 256 ([K = 32] x [16 values of m], f = 2, k = 0: 1) -- the rotation by r
//     lives in the OPERAND: one set of fragments per r (4 x 8 KB, LDS resident, 32 x 16 bytes read per lane and 16
//     rows), so a row's coefficients are split into bf16 terms ONCE.  (First form, measured: rotation and split on the
//     vector ALUs for every r with one operand set in registers -- 83 us, bound by ~370 vector instructions per 16 rows;
//     this form: 66-68 us.  profiles/r05_mx_ifft_ab.txt.)
//   * Im c_0 never contributes and its slot carries Re(c_32 exp(i pi r / 4)) -- the one value still rotated by hand --
//     whose column is 2 (-1)^m; the bias rides on Re c_0, whose column is 1;
//   * exp(2 pi i k (m + 32) / 64) = (-1)^k: E = even-k sum, O = odd-k sum for m < 32 give y[m] = E + O and
//     y[m + 32] = E - O -- two tiles of m instead of four;
//   * the OUTPUT is bfloat16 (8 significant bits), so coefficients and table are split in TWO bf16 terms each and the
//     three products hi.hi + hi.lo + lo.hi are kept: 2^-16 relative to a row's magnitude, ~100 times below the
//     rounding of the store.  12 MFMAs per r and 16 rows; 48 per 16 rows against 96 for the forward kernel;
//   * the accumulator tile of a wave is [16 rows] x [16 m]: after the four r a lane holds 4 consecutive points
//     (4 m .. 4 m + 3) of 4 rows in 4 places (m, m + 16, m + 32, m + 48) -- sixteen 8-byte stores, 128 contiguous bytes
//     per 16 lanes (the vector-ALU kernel: 2-byte stores, 64 contiguous bytes per half-wave).
// Column phase, spectrum requests and scaling are those of k_fft2d_inv3 (its tile with a row stride of 38 complex so
// that a lane's 8 consecutive coefficients are one aligned 64-byte read, conflict-free over 16 rows); the k = 32 column
// tasks of all groups run in wave 0.  Served: H = 64 / 128 / 256, no epilogue (sc_engine.cpp).
//
// F3_NOTE_PK_MUL_LX (round 6, DESIGN 3.5 -- what made H = 64 non-repeatable in round 5).  On MI355X a packed-fp32
// instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose operand selects are op_sel:[0,1,.] -- low lane: LOW half
// of src0, HIGH half of src1; op_sel_hi does not matter -- returns 0 in the LOW result of lanes 48-63 while another
// wave of the same SIMD executes v_mfma_f32_16x16x32_bf16 (18 % of such executions in scripts/ubench_pk_forms.hip; never
// with 32x32x2 f32, 16x16x4 f32, 16x16x16 f16 / bf16, 32x32x16 bf16 or vector work in the other wave; 16x16x32 f16 does
// the same and i32 16x16x64 i8 more rarely -- the 16 x 16 shapes gfx950 added, none of them used by the engine; never for
// the other three op_sel combinations).  At P = 1 hipcc's SLP vectoriser kept the scaled spectrum entries as (im, re) register pairs and
// emitted the group-twiddle product of the column task as `v_pk_mul_f32 vD, vTw, vY op_sel:[0,1] op_sel_hi:[0,0]`: with two
// workgroups per unit the other workgroup's row pass zeroed the real part of 16 entries now and then.  Replacing exactly
// those eight instructions in the generated assembly by two v_mul_f32 each, by the commuted product (op_sel:[1,0]) or
// by natural halves with the exchange moved to the consumer's src2: 0 of 2000 / 100 / 100 launches differ
// (profiles/r06_mxi_h64_root_cause.txt).  Source-level: at P = 1 the group twiddle is 1 and the product is gone
// (`if constexpr (P == 1)` in the column task, here and in k_fft2d_inv3); tests/test_isa_pk_forms.py disassembles the
// shipped code object and fails if any kernel that executes v_mfma_f32_16x16x32_bf16 holds a packed-fp32 instruction
// with op_sel:[0,1,.].
// Reference lines: spectral_convolution.py:520-568 (zero-filled spectrum, ifftn / irfft, bias).
#define SC_MXI_URS 38     // row stride (complex) of T[64 rows][32 columns | . | 33 + a: the group's k = 32 column]
#ifndef SC_MXI_WGS
#define SC_MXI_WGS 2      // persistent workgroups per compute unit = register budget (waves per SIMD)
#endif

template <int H>
struct F3MxiLds {
  static constexpr int P = H / 64;
  static constexpr int xch_c = 33 * SC_F3_CCS;            // the column tasks' private exchange patches
  static constexpr int T_c = 64 * SC_MXI_URS;
  static constexpr int off_xch = 0;
  static constexpr int off_T = off_xch + xch_c * 8;
  static constexpr int off_twH = off_T + T_c * 8;
  static constexpr int off_tw64 = off_twH + H * 8;
  static constexpr int off_c32 = off_tw64 + 64 * 8;        // 33rd column of the spectrum (64 entries)
  static constexpr int off_G = off_c32 + 64 * 8;           // operand fragments [r 4][tile 4][term 2][lane 64][16 bytes]
  static constexpr int total = off_G + 32 * 1024;
  static_assert(P <= 4 && 33 + P <= SC_MXI_URS, "k = 32 of group a is parked in column 33 + a of its row (H <= 256)");
  static_assert(off_T % 16 == 0 && off_G % 16 == 0, "16-byte reads");
  static_assert(SC_MXI_WGS * total <= 160 * 1024, "workgroups per compute unit");
};

// operand table of the inverse row pass: [r 4][tile 4 = (parity u, m tile t)][term 2][lane 64][8 bf16], MFMA B layout:
// lane (j, g) holds rows K = 8 g + e of column m = 16 t + j; K = (k index kk = 4 g + (e >> 1), part e & 1),
// k = 2 kk + u; part 0 multiplies Re c, part 1 Im c:  f cos(2 pi k n / 256), -f sin(...), n = 4 m + r, f = 2 (k = 0: 1);
// (u = 0, kk = 0, part 1) is the k = 32 slot: 2 (-1)^m, multiplying Re(c_32 exp(i pi r / 4)) (the kernel rotates that one)
static inline void fft3mxi_build_table(std::vector<uint16_t>* out) {
  const double two_pi = 6.283185307179586476925286766559;
  out->assign((size_t)4 * 4 * 2 * 64 * 8, 0);
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
  for (int r = 0; r < 4; ++r)
    for (int tile = 0; tile < 4; ++tile)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int u = tile >> 1, t = tile & 1, j = lane & 15, g = lane >> 4;
          const int m = 16 * t + j, kk = 4 * g + (e >> 1), part = e & 1, k = 2 * kk + u;
          double v;
          if (k == 0) {
            v = part ? ((m & 1) ? -2.0 : 2.0) : 1.0;
          } else {
            const int idx = (k * (4 * m + r)) & 255;         // exact phase reduction
            const double th = two_pi * (double)idx / 256.0;
            v = part ? -2.0 * std::sin(th) : 2.0 * std::cos(th);
            if (idx % 64 == 0) v = std::round(v);
          }
          double rest = v;
          for (int term = 0; term < 2; ++term) {
            const uint16_t b = to_bf16(rest);
            (*out)[((((size_t)r * 4 + tile) * 2 + term) * 64 + lane) * 8 + e] = b;
            rest -= from_bf16(b);
          }
        }
}

#ifndef SC_EMU
typedef uint32_t sc_mx_u2 __attribute__((ext_vector_type(2)));
SC_DEVICE uint32_t sc_mx_pack_bf16(const float a, const float b) {     // {bf16(a), bf16(b)}, nearest even: v_cvt_pk_bf16_f32
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}
SC_DEVICE void sc_mx_store8_stream(void* p, const uint32_t a, const uint32_t b) {
  const sc_mx_u2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<sc_mx_u2*>(p));
}
#else
inline uint32_t sc_mx_pack_bf16(const float a, const float b) {
  return (uint32_t)sc_f32_to_bf16_bits(a) | ((uint32_t)sc_f32_to_bf16_bits(b) << 16);
}
inline void sc_mx_store8_stream(void* p, const uint32_t a, const uint32_t b) {
  const uint32_t v[2] = {a, b};
  std::memcpy(p, v, 8);
}
#endif
// (a, b) -> one dword of the leading bf16 terms and one of the remainders' (a = hi + lo to 16 bits)
SC_DEVICE void sc_mx_split2(const float a, const float b, uint32_t& hi, uint32_t& lo) {
  hi = sc_mx_pack_bf16(a, b);
  lo = sc_mx_pack_bf16(a - sc_bits_to_f32(hi << 16), b - sc_bits_to_f32(hi & 0xffff0000u));
}

template <int H>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, SC_MXI_WGS)
k_fft2d_inv_mx(const cf32* __restrict__ yhat, sc_bf16* __restrict__ y, const float* __restrict__ bias, int channels,
               const cf32* __restrict__ tabW, const cf32* __restrict__ tabH, const uint16_t* __restrict__ tabG, int Mx,
               int My, float s_dc, float s_other, F3Shard sh, int64_t n_images, int gstride) {
  constexpr int P = H / 64, URS = SC_MXI_URS;
  typedef F3MxiLds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);
  cf32* y32 = reinterpret_cast<cf32*>(smem + L::off_c32);
  sc_mx_u4* Gl = reinterpret_cast<sc_mx_u4*>(smem + L::off_G);

  const int tid = SC_TID;
  const int w = SC_UNIFORM(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 15, g = lane >> 4;                 // MFMA roles: (row j, k group g) of A, (column j, k group g) of B
  const int cl = lane >> 3, mu = lane & 7;                // column phase: wave w owns columns 8 w .. 8 w + 7, 8 lanes each

  for (int i = tid; i < H; i += 256) twH[i] = tabH[i];
  if (tid < 64) tw64[tid] = tabW[(4 * (tid >> 3) * (tid & 7)) & 255];
  for (int i = tid; i < 32 * 64; i += 256) Gl[i] = sc_mx_load16(tabG + (size_t)i * 8);

  // spectrum entries of a lane (column c = 8 w + cl, rows q = mu + 8 q2) and the parked 33rd column: straight from
  // global memory into registers, those of the next image requested behind the last group's column transforms
  // (k_fft2d_inv3)
  cf32 yh[8], y32r;
  auto request = [&](const int64_t im) SC_ALWAYS_INLINE_LAMBDA {
    const int c = 8 * w + sc_opaque(cl);
    const int mu_o = sc_opaque(mu), t_o = sc_opaque(tid);
    if (sh.rows <= 0) {
      const cf32* src = yhat + im * (int64_t)Mx * My;
#pragma unroll
      for (int q2 = 0; q2 < 8; ++q2) {
        const int row = f2d_fx(mu_o + 8 * q2) + Mx / 2;
        yh[q2] = src[(row >= 0 && row < Mx && c < My) ? row * My + c : 0];
      }
      const int row = f2d_fx(t_o & 63) + Mx / 2;
      y32r = src[(row >= 0 && row < Mx && 32 < My) ? row * My + 32 : 0];
    } else {                                             // sharded spectrum (include/sc_engine.h, sc_spectrum_shards)
#pragma unroll
      for (int q2 = 0; q2 < 8; ++q2) {
        const int row = f2d_fx(mu_o + 8 * q2) + Mx / 2;
        yh[q2] = yhat[f3_shard_index(sh, im, (row >= 0 && row < Mx && c < My) ? row * My + c : 0, My)];
      }
      const int row = f2d_fx(t_o & 63) + Mx / 2;
      y32r = yhat[f3_shard_index(sh, im, (row >= 0 && row < Mx && 32 < My) ? row * My + 32 : 0, My)];
    }
  };
  if ((int64_t)SC_BID_X < n_images) request(SC_BID_X);
  SC_SYNC();                                             // tables

#pragma unroll 1
  for (int64_t img = SC_BID_X; img < n_images; img += gstride) {
    sc_bf16* yo = y + img * (int64_t)H * SC_F2D_W;
    const float badd = (bias != nullptr) ? bias[img % channels] : 0.f;
    {
      const int c = 8 * w + sc_opaque(cl);
      const int mu_o = sc_opaque(mu), t_o = sc_opaque(tid);
      const float sc_c = (c == 0) ? s_dc : s_other;
#pragma unroll
      for (int q2 = 0; q2 < 8; ++q2) {
        const int row = f2d_fx(mu_o + 8 * q2) + Mx / 2;
        yh[q2] = (row >= 0 && row < Mx && c < My) ? cf_scale(yh[q2], sc_c) : cf_make(0.f, 0.f);
      }
      if (tid < 64) {
        const int row = f2d_fx(t_o) + Mx / 2;
        y32[tid] = (row >= 0 && row < Mx && 32 < My) ? cf_scale(y32r, s_other) : cf_make(0.f, 0.f);
      }
    }
    SC_SYNC();

    // one inverse column task of group a (k_fft2d_inv3): spectrum column -> 64 rows b of T[b][cdst]
    auto column = [&](const int cdst, cf32* cb, const int a, auto extra_tag, const bool act) SC_ALWAYS_INLINE_LAMBDA {
      constexpr bool EXTRA = decltype(extra_tag)::value != 0;
      cf32 v[8], o[8];
#pragma unroll
      for (int q2 = 0; q2 < 8; ++q2) {
        const cf32 src = EXTRA ? y32[mu + 8 * q2] : yh[q2];
        if constexpr (P == 1) {                            // a = 0: the group twiddle is 1 (and see F3_NOTE_PK_MUL_LX)
          v[q2] = src;
        } else {
          const int fx = f2d_fx(mu + 8 * q2);
          int idx = (a * fx) % H;
          if (idx < 0) idx += H;
          v[q2] = cf_mul_pk(src, cf_conj(twH[idx]));
        }
      }
      dft8<+1>(v, o);                                      // over q2 -> m (row b = m + 8 b1)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const cf32 val = (m == 0) ? o[0] : cf_mul_pk(o[m], cf_conj(SC_F3_LD64(tw64 + m * 8 + mu)));
        if (act) cb[m * 8 + mu] = val;
      }
      SC_WAVE_SYNC();
#pragma unroll
      for (int q1 = 0; q1 < 8; q1 += 2) SC_F3_LD128(cb + mu * 8 + q1, v[q1], v[q1 + 1]);
      dft8<+1>(v, o);                                      // over q1 -> b1
      if (act) {
#pragma unroll
        for (int b1 = 0; b1 < 8; ++b1) T[(mu + 8 * b1) * URS + cdst] = o[b1];
      }
      SC_WAVE_SYNC();
    };

    if (My > 32) {                                         // k = 32 of every group up front, parked in column 33 + a:
      if (w == 0) {                                        // P tasks of 8 lanes, all in wave 0 (k_fft2d_inv3 gives one to each
        const int ac = cl < P ? cl : 0;                    // wave: four instruction streams for the work of half a wave)
        column(33 + ac, xch + ac * SC_F3_CCS, ac, sc_int<1>(), cl < P);
      }
    } else if (tid < 64) {
#pragma unroll
      for (int a = 0; a < P; ++a) T[tid * URS + 33 + a] = cf_make(0.f, 0.f);
    }
    SC_SYNC();

#pragma unroll 1
    for (int a = 0; a < P; ++a) {
#ifndef SC_MXI_ABL_NOCOL                                   /* (measurement builds) */
      column(8 * w + cl, xch + (8 * w + cl) * SC_F3_CCS, a, sc_int<0>(), true);
#endif
      if (a == P - 1) request(img + gstride < n_images ? img + gstride : img);
      SC_SYNC();
      // ---------------- rows b = 16 w + j of the group on the matrix cores ----------------
      // coefficients k = 8 g + q of the lane's row, split ONCE into two bf16 terms: the rotation by r lives in the
      // operand fragments (one set per r, read from LDS: 32 x 16 bytes per lane and 16 rows)
      cf32 cr[8];
      const cf32* trow = T + (16 * w + j) * URS;
#pragma unroll
      for (int q = 0; q < 8; q += 2) SC_F3_LD128(trow + 8 * g + q, cr[q], cr[q + 1]);
      const cf32 c32 = SC_F3_LD64(trow + 33 + a);
      cr[0].x += (g == 0) ? badd : 0.f;                    // k = 0: its column of the operand is 1 for every point
      uint32_t outp[4][4][2];                              // [row v][place][dword]: 4 consecutive points as bf16
      float keep[4][4];
#ifdef SC_MXI_ABL_NOROW
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
          outp[v][pl][0] = __float_as_uint(cr[v].x) ^ pl;
          outp[v][pl][1] = __float_as_uint(cr[4 + v].y + c32.x) ^ pl;
        }
#else
      uint32_t fh[2][4], fl[2][4];                         // [parity u]: even k = cr[0], cr[2], cr[4], cr[6]; odd k = cr[1], ...
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) sc_mx_split2(cr[2 * i + u].x, cr[2 * i + u].y, fh[u][i], fl[u][i]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        {                                                  // the k = 32 slot: Re(c_32 exp(i pi r / 4)) beside Re c_0
          constexpr float hh = 0.70710678118654752440f;
          const float e32 = (r == 0) ? c32.x : (r == 1) ? hh * (c32.x - c32.y) : (r == 2) ? -c32.y : -hh * (c32.x + c32.y);
          sc_mx_split2(cr[0].x, g == 0 ? e32 : cr[0].y, fh[0][0], fl[0][0]);     // (all lanes: no branch)
        }
        const sc_mx_u4 Ah[2] = {sc_mx_u4{fh[0][0], fh[0][1], fh[0][2], fh[0][3]}, sc_mx_u4{fh[1][0], fh[1][1], fh[1][2], fh[1][3]}};
        const sc_mx_u4 Al[2] = {sc_mx_u4{fl[0][0], fl[0][1], fl[0][2], fl[0][3]}, sc_mx_u4{fl[1][0], fl[1][1], fl[1][2], fl[1][3]}};
        sc_mx_f4 acc[4];                                   // [2 u + t]
        sc_mx_u4 b0[4], b1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          b0[t] = Gl[((r * 4 + t) * 2 + 0) * 64 + lane];
          b1[t] = Gl[((r * 4 + t) * 2 + 1) * 64 + lane];
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[t][v] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) sc_mfma_16x16x32_bf16(acc[t], Al[t >> 1], b0[t]);     // small products first
#pragma unroll
        for (int t = 0; t < 4; ++t) sc_mfma_16x16x32_bf16(acc[t], Ah[t >> 1], b1[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) sc_mfma_16x16x32_bf16(acc[t], Ah[t >> 1], b0[t]);
        // y[4 (16 t + j) + r] = E + O,  y[4 (16 t + j + 32) + r] = E - O
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) {
            const int t = pl & 1;
            const float val = (pl < 2) ? acc[t][v] + acc[2 + t][v] : acc[t][v] - acc[2 + t][v];
            if ((r & 1) == 0)
              keep[v][pl] = val;
            else
              outp[v][pl][r >> 1] = sc_mx_pack_bf16(keep[v][pl], val);
          }
      }
#endif
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        sc_bf16* row = yo + (int64_t)(P * (16 * w + 4 * g + v) + a) * SC_F2D_W + 4 * j;
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#ifdef SC_MXI_ABL_NOSTORE
          if (outp[v][pl][0] == 0x12345678u && outp[v][pl][1] == 0x9abcdef0u)
#endif
          sc_mx_store8_stream(row + 64 * (pl & 1) + 128 * (pl >> 1), outp[v][pl][0], outp[v][pl][1]);
      }
      SC_SYNC();                                           // T is rewritten by the next group
    }
  }
}

template <int H>
static void fft3mxi_launch_inv(const Fft2dPlan* fp, const cf32* yhat, sc_bf16* y, const float* bias, int channels,
                               int64_t n_images, float s_dc, float s_other, sc_stream_t st, F3Shard sh) {
  int64_t grid = (int64_t)SC_MXI_WGS * sc_cu_count();
  if (grid > n_images) grid = n_images;
  SC_LAUNCH((k_fft2d_inv_mx<H>), dim3((unsigned)grid), dim3(256), 0, st, yhat, y, bias, channels,
            (const cf32*)fp->tabW, (const cf32*)fp->tabH, (const uint16_t*)fp->tabG, fp->Mx, fp->My, s_dc, s_other, sh,
            n_images, (int)grid);
}

static inline int fft3mxi_inverse(const Fft2dPlan* fp, int mode, const cf32* yhat, const float* bias, int64_t channels,
                                  sc_bf16* y, int64_t n_images, sc_stream_t st, std::string* err,
                                  F3Shard sh = F3Shard{0, 0}) {
  const float s_dc = (mode == 0) ? fp->si : fp->sf;
  const float s_other = (mode == 0) ? fp->si : 0.5f * fp->sf;
  switch (fp->H) {
    case 64: fft3mxi_launch_inv<64>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, sh); break;
    case 128: fft3mxi_launch_inv<128>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, sh); break;
    case 256: fft3mxi_launch_inv<256>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, sh); break;
    default: *err = "sc_engine: fft2d (matrix-core row pass): unsupported H"; return 1;
  }
  if (hipGetLastError() != hipSuccess) {
    *err = "sc_engine: launch of k_fft2d_inv_mx failed";
    return 1;
  }
  return 0;
}
