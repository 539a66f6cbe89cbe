// sc_kernels_fft3.h -- third generation of the fused, pruned 2-D FFT kernels (fast path).
//
// Same contract as sc_kernels_fft.h (one 256-thread workgroup per (b, c) image, W = 256,
// H in {64,128,256,512}, kept block <= 64 x 33, nothing but the kept modes ever stored), but
// re-cut for OCCUPANCY: measured on MI355X a single wave issues one VALU instruction every
// ~7 cycles and a SIMD needs >= 4 resident waves to approach its issue rate
// (profiles/r01_valu_issue_ubench.txt); generation 2 (16 points per lane, 194 VGPRs, 59 KB LDS)
// ran 2 waves per SIMD and spent 42 % of its wave-cycles parked on s_waitcnt.  Here
//
//   * a 256-point row FFT is spread over 32 lanes x 8 points (two radix-8 passes in registers,
//     one LDS transpose between them), the last, pruned radix-4 is a 4-term sum read back
//     from LDS -> ~110 VGPRs, 4 waves per SIMD;
//   * column FFTs (64 points) use 8 lanes x 8 points, so all 256 threads work in the column
//     phase (generation 2 used 132 of 256);
//   * twiddles that depend only on the lane live in registers for the whole kernel;
//   * LDS: 18 KB row/column exchange + 18 KB group tile + <= 4 KB table  -> 4 workgroups per CU;
//   * global traffic: every half-wave touches one aligned 128-byte line per instruction.
//
// Forward (R2C, pruned):   x[H][256] -> xhat[Mx][My]
//   rows h = P b + a (P = H/64 groups a, b = 0..63); row pairs (2p, 2p+1) of a group are packed
//   z = xA + i xB.  Z[k] for |k| <= 32 goes to the group tile T'[p][32 + k] WITHOUT unpacking;
//   the column phase unpacks (A = (Z[k] + conj Z[-k])/2, B = (Z[k] - conj Z[-k])/2i) while it
//   loads, runs 33 column FFTs of 64 points, multiplies by w_H^(a fx) and accumulates in
//   registers over the groups.
// Inverse (C2R, zero padded): the transpose of the above.
#pragma once
#include "sc_kernels_fft.h"

#define SC_F3_XRS 36   // row stride (complex) of the row exchange [half-wave][k1][.]: conflict-free b64
#define SC_F3_CCS 66   // column stride (complex) of the column exchange
// group tile: forward T'[32 pairs][TRS] holds Z[-32..32] (65) + per group the pair's (Z[32], Z[-32])
// stash for the deferred 33rd column; inverse T[64 rows][URS] holds 33 columns + per group the
// precomputed 33rd column.  32 * TRS == 64 * URS.
#define SC_F3_TRS(P) (66 + 2 * (P))
#define SC_F3_URS(P) (33 + (P))

template <int H>
struct F3Lds {
  static constexpr int xch_c = 8 * 8 * SC_F3_XRS;        // 2304 complex
  static constexpr int cx_c = 33 * SC_F3_CCS;            // 2178 complex, aliases xch
  static constexpr int P = H / 64;
  static constexpr int TRS = SC_F3_TRS(P), URS = SC_F3_URS(P);
  static constexpr int T_c = 32 * TRS;                   // 2368 complex at H = 256 (= 64 * URS)
  static constexpr int tw64_c = 64;                      // w64^(mu q1), [mu][q1]
  static constexpr int off_xch = 0;
  static constexpr int off_T = off_xch + xch_c * 8;
  static constexpr int off_twH = off_T + T_c * 8;
  static constexpr int off_tw64 = off_twH + H * 8;
  static constexpr int off_tw2 = off_tw64 + tw64_c * 8;   // [n4][k3] last-stage row twiddles (32)
  static constexpr int off_c32 = off_tw2 + 32 * 16;       // 33rd column spectrum of the inverse (64)
  static constexpr int total = off_c32 + 64 * 8;
  static_assert(cx_c <= xch_c, "column exchange aliases the row exchange");
  static_assert(64 * URS <= T_c, "inverse tile fits");
  static_assert((P + 1) * SC_F3_CCS + P * 64 <= xch_c, "deferred 33rd-column tasks fit in the exchange buffer");
  static_assert(SC_F2D_KX * SC_F2D_KY <= T_c, "output tile aliases the group tile");
  static_assert(total * (H <= 256 ? 4 : 3) <= 160 * 1024, "4 workgroups per CU (3 at H = 512)");
};

// ---- real-tensor I/O of the fused kernels: float32, or bfloat16 storage (SC_PLAN_IO_BF16: the spectral
//      arithmetic stays fp32, only x / y / gy / gx cross HBM as 2-byte values -- BASELINE configs[1] "bf16").
//      A lane owns points lam + 32 j of a row.  bf16: lanes 2l and 2l+1 read the SAME aligned dword (one
//      64-byte piece per half-wave and instruction) and keep their half; stores are 2-byte, 64 contiguous
//      bytes per half-wave, rounded to nearest even by v_cvt_pk_bf16_f32.
SC_DEVICE float f3_load(const float* row, const int lam, const int j) {
#ifdef SC_F3_PLAIN_IO
  return row[lam + 32 * j];
#else
  return SC_LOAD_STREAM(&row[lam + 32 * j]);             // read once: keep it out of the caches the
#endif                                                   // spectra and weights live in
}
SC_DEVICE float f3_load(const sc_bf16* row, const int lam, const int j) {
  const uint32_t u = SC_LOAD_STREAM(reinterpret_cast<const uint32_t*>(row) + (lam >> 1) + 16 * j);
  return sc_bits_to_f32((u << ((lam & 1) ? 0 : 16)) & 0xffff0000u);
}
SC_DEVICE void f3_store(float* row, const int lam, const int j, const float v) {
#ifdef SC_F3_PLAIN_IO
  row[lam + 32 * j] = v;
#else
  SC_STORE_STREAM(&row[lam + 32 * j], v);
#endif
}
SC_DEVICE void f3_store(sc_bf16* row, const int lam, const int j, const float v) {
  SC_STORE_STREAM(&row[lam + 32 * j].v, sc_f32_to_bf16_bits(v));
}
// A-B (-DSC_F3_INV_PLAIN_GROUPS=n): the first n of the P row groups leave through ordinary stores -- a plain 537 MB
// writer finishes in 87 us against 118 us for the non-temporal one, but leaves dirty lines the NEXT kernel pays for
// (profiles/r01_writeback_ubench.txt); a fraction may drain inside the slack of a compute-bound successor
#ifndef SC_F3_INV_PLAIN_GROUPS
#define SC_F3_INV_PLAIN_GROUPS 0
#endif
SC_DEVICE void f3_store_plain(float* row, const int lam, const int j, const float v) { row[lam + 32 * j] = v; }
SC_DEVICE void f3_store_plain(sc_bf16* row, const int lam, const int j, const float v) {
  row[lam + 32 * j].v = sc_f32_to_bf16_bits(v);
}

// ---- block epilogue fused into the inverse transform's store path (the FNO block computes
//      act(conv(x) + skip(x)), neuralop/layers/fno_block.py:392-414, as three more R-sized passes):
//      EPI 0: y = v;  EPI 1: y = v + skip;  EPI 2: y = gelu(v + skip) with the pre-activation optionally saved for
//      the backward pass.  gelu = torch's default form 0.5 v (1 + erf(v / sqrt 2)).  The inverse kernel is VALU /
//      LDS-issue bound, so the library erff (~40 instructions; measured: the fused inverse 113 -> 570 us, no faster
//      than the three elementwise passes it replaces) is replaced by the 5-term rational approximation of
//      Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 in erf, i.e. fp32 round-off class for the activation; 14
//      instructions with v_rcp_f32 / v_exp_f32).  preact is written only with EPI 2 (SC_ACT_NONE + preact is
//      rejected at the C-ABI, include/sc_engine.h).
// (sc_erfc_core / sc_gelu live in sc_device.h: the stand-alone epilogue pass uses them too)
template <int EPI, typename IO>
SC_DEVICE void f3_store_epi(IO* row, const float sk, IO* prow, const int lam, const int j, float v) {
  if (EPI >= 1) v += sk;
  if (EPI == 2) {
    if (prow != nullptr) f3_store(prow, lam, j, v);
    v = sc_gelu(v);
  }
  f3_store(row, lam, j, v);
}

// Sharded spectrum addressing (mode-parallel layers: the transform writes / reads the rank-major all-to-all buffer
// in place, include/sc_engine.h sc_spectrum_shards): block p = rows [p rows, (p + 1) rows) of the first kept dim of
// EVERY image, [block][image][rows][My]; rows <= 0 = the plain [image][Mx][My] layout
struct F3Shard {
  int rows;
  int64_t block_stride;      // complex elements between blocks
};
SC_HD int64_t f3_shard_index(const F3Shard sh, const int64_t img, const int i, const int My) {
  const int row = i / My, col = i - row * My;
  const int blk = row / sh.rows;
  return (int64_t)blk * sh.block_stride + (img * sh.rows + (row - blk * sh.rows)) * (int64_t)My + col;
}

// 8-point DFT, natural order in and out: b[k] = sum_n a[n] w8^(nk), w8 = exp(DIR 2 pi i / 8)
template <int DIR>
SC_HD void dft8(const cf32 (&a)[8], cf32 (&b)[8]) {
  cf32 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
  cf32 o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
  radix4<DIR>(e0, e1, e2, e3);
  radix4<DIR>(o0, o1, o2, o3);
  constexpr float r = 0.70710678118654752440f;
  // t1 = w8 o1 = (o1 + (DIR i) o1) r,  t3 = w8^3 o3 = -(o3 - (DIR i) o3) r  (same roundings as the
  // component formulas: one add, one multiply per part)
  const cf32 t1 = cf_scale(cf_add_rot<DIR>(o1, o1), r);
  const cf32 t3 = cf_scale(cf_sub_rot<DIR>(o3, o3), -r);
  b[0] = cf_add(e0, o0);
  b[4] = cf_sub(e0, o0);
  b[1] = cf_add(e1, t1);
  b[5] = cf_sub(e1, t1);
  b[2] = cf_add_rot<DIR>(e2, o2);
  b[6] = cf_sub_rot<DIR>(e2, o2);
  b[3] = cf_add(e3, t3);
  b[7] = cf_sub(e3, t3);
}

// a * i^n, n = 0..3 (lane dependent, used once at kernel start)
SC_HD cf32 cf_rot_i(const cf32 a, const int n) {
  switch (n & 3) {
    case 1: return cf_make(-a.y, a.x);
    case 2: return cf_make(-a.x, -a.y);
    case 3: return cf_make(a.y, -a.x);
    default: return a;
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// occupancy / prefetch depth of the forward kernel.  Round 3: with the last row stage in registers the kernel's compute
// (67 us) is well below its load stream (86 us alone) and the two overlap imperfectly (106.5 us at depth 1, four
// workgroups per CU); a TWO-round-deep row prefetch at three workgroups per CU (152 VGPRs) runs 103-105 us and the
// layer step 538.9 -> 529.9 us (21 interleaved rounds, profiles/r03_prefetch_depth_step_ab.txt).  Depth 2 at four
// workgroups spills (120 us), depth 4 at two / three: 117 / 179 us.  A-B: -DSC_F3_FWD_OCC=4 -DSC_F3_PF_DEPTH=1 (round 2).
#ifndef SC_F3_FWD_OCC
#define SC_F3_FWD_OCC 3
#endif
#ifndef SC_F3_PF_DEPTH
#define SC_F3_PF_DEPTH 2
#endif
#ifndef SC_F3_TW1_LEGACY
#define SC_F3_TW1_CS 1
#endif
// Round 3: the LAST row stage without LDS.  The pruned radix-4 over n4 (k = k1 + 8 k3 + 64 k4, one k4 per (k1, k3)) is
// a 4-term sum whose terms were exchanged through LDS (8 ds_write_b64 + 4 ds_read_b128 per lane and round: ~40 % of
// the row phase's LDS cycles).  Now the wave's two row pairs are laid out so that the four n4 terms of a
// (k1, k3) sit in lanes l, l ^ 16, l ^ 32, l ^ 48 (post-transpose role: k1 = lane & 7, pair = bit 3, n4 = lane >> 4)
// and the sum is a two-stage reduce-scatter with v_permlane16_swap / v_permlane32_swap: swap(t[i], t[i + 4]);
// t[i] += t[i + 4] leaves k3 = i in the lower and k3 = i + 4 in the upper lane of a pair -- uniform code, no selects.
// Each lane ends with two finished coefficients and stores them to the group tile.  A-B: -DSC_F3_EXCH2_LDS.
#ifndef SC_F3_EXCH2_LDS
#define SC_F3_SWAP 1
#endif
// explicit LDS read widths (sc_device.h: the compiler's ds_read2_b64 pairs run at half rate).  A-B: -DSC_F3_LDS_PLAIN
#ifndef SC_F3_LDS_PLAIN
#define SC_F3_LD64(p) sc_lds_ld64(p)
#define SC_F3_LD128(p, a, b) sc_lds_ld128((p), (a), (b))
#else
#define SC_F3_LD64(p) (*(p))
#define SC_F3_LD128(p, a, b) do { (a) = (p)[0]; (b) = (p)[1]; } while (0)
#endif
#define SC_F3_PS 286    // pair 1 of a wave inside the wave's exchange area: 8 * XRS - 2 (lands in pair 0's row padding;
                        // shifts its banks by two 8-byte slots: the transposed reads of both pairs are conflict-free)
template <int H, typename IO>
#ifndef SC_F3_FWD_OCC_BF16
#define SC_F3_FWD_OCC_BF16 4   // with SC_F3_PF_DEPTH_BF16 = 1: 126 registers (round 4: 3 / depth 2)
#endif
#ifndef SC_F3_PF_DEPTH_BF16
#define SC_F3_PF_DEPTH_BF16 1
#endif
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (H <= 256 ? (sizeof(IO) == 4 ? SC_F3_FWD_OCC : SC_F3_FWD_OCC_BF16) : 3))
k_fft2d_fwd3(const IO* __restrict__ x, cf32* __restrict__ xhat, const cf32* __restrict__ tabW,
             const cf32* __restrict__ tabH, int Mx, int My, float s_dc, float s_other, F3Shard sh) {
  constexpr int P = H / 64;
  // prefetch depth in row rounds: 2 for fp32 (three workgroups per unit), 1 for bf16 storage with FOUR workgroups per unit
  // (round 5: 91.8 -> 88.3 us at the metric shape, 126 registers, profiles/r05_bf16_fwd_ab.txt: with half the bytes per
  // round the kernel is bound by its arithmetic -- 70 us with the loads removed -- and a fourth workgroup overlaps more of it)
  constexpr int PFD = sizeof(IO) == 4 ? SC_F3_PF_DEPTH : SC_F3_PF_DEPTH_BF16;
  typedef F3Lds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);
  ctw4* tw2t = reinterpret_cast<ctw4*>(smem + L::off_tw2);

  const int tid = SC_TID;
  const int w = SC_UNIFORM(tid >> 6);
  const int lane = tid & 63, hs = lane >> 5, lam = lane & 31;
  const int hw = w * 2 + hs;                           // half-wave = row-pair slot of a round
  const int64_t img = SC_BID_X;
  const IO* xi = x + img * (int64_t)H * SC_F2D_W;

  // Session 2: the workgroup's tables are REQUESTED here -- every request unconditional, into registers -- and stored to
  // LDS only after the first row rounds have been requested as well (below).  Before, each table was loaded, waited
  // for and stored inside its own `if (tid < n)` block: three L2 latencies one after the other plus the lane's row
  // twiddles, all in front of the first request for image data, once per image.
  // (Untracked loads, sc_device.h: the compiler sinks an ordinary load of a __restrict__ table into the block that uses
  // it -- behind the row requests -- and then drains everything for it.)
  constexpr int NTH = (H + 255) / 256;
  cf32 rH[NTH];
#pragma unroll
  for (int q = 0; q < NTH; ++q) rH[q] = sc_gload8_untracked(tabH + (tid + 256 * q) % H);
  cf32 r64 = sc_gload8_untracked(tabW + ((4 * ((tid & 63) >> 3) * (tid & 7)) & 255));   // w64^(mu q1)
  // [k3][n4]: w32^(n4 k3), times i^n4 for k3 >= 4 (the k4 = 3 terms); slot [0][n4] = w32^(4 n4) (k = +32)
  const int t2n = tid & 3, t2k = (tid >> 2) & 7;
  cf32 r2 = sc_gload8_untracked(tabW + ((8 * t2n * (t2k == 0 ? 4 : t2k)) & 255));

  // ---- row phase roles and per-lane twiddles
#ifdef SC_F3_SWAP
  const int k1l = lane & 7, sq = (lane >> 3) & 1, n4 = lane >> 4;   // after the transpose: lane = (n4, pair, k1)
#else
  const int k1l = lam >> 2, n4 = lam & 3;             // after the transpose: lane = (k1, n4)
#endif
#if defined(SC_F3_TW1_CS)
  cf32 tw1[8];
#pragma unroll
  for (int k = 1; k < 8; ++k) tw1[k] = tabW[(lam * k) & 255];               // w256^(n2 k1) as (c, s)
#else
  ctw3 tw1[8];
#pragma unroll
  for (int k = 1; k < 8; ++k) tw1[k] = ctw3_make(tabW[(lam * k) & 255]);    // w256^(n2 k1)
#endif
  const ctw4* tw2 = tw2t + n4;                           // LDS table [k3][n4]: four adjacent 16-B slots per read
#ifdef SC_F3_SWAP
  cf32* xb = xch + w * 16 * SC_F3_XRS + hs * SC_F3_PS;   // this lane's pair as the PRODUCER of the transpose
  const cf32* xr = xch + w * 16 * SC_F3_XRS + sq * SC_F3_PS + k1l * SC_F3_XRS + n4;   // ... as its CONSUMER
#else
  cf32* xb = xch + hw * 8 * SC_F3_XRS;
#endif

  // ---- column phase roles: wave w owns columns 8 w .. 8 w + 7, 8 lanes per column
  const int cl = lane >> 3, mu = lane & 7;
  cf32 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = cf_make(0.f, 0.f);

  // one column task for group a (lane = (column slot, mu)): rows b = 8 b1 + mu, pair p = 4 b1 + (mu >> 1),
  // member mu & 1, unpacked while loading (zk -> Z_p[k], zm -> Z_p[-k], both with pair stride TRS);
  // cb = the column's private exchange patch; result F_a[q1 + 8 q2] * w_H^(a fx) goes to acc (+=)
  // or, for the deferred 33rd column, to dst[q]
  auto column = [&](const cf32* zk0, const cf32* zm0, cf32* cb, const int a, auto extra_tag, cf32* dst, const bool act) {
    constexpr bool EXTRA = decltype(extra_tag)::value != 0;
    const float sg = (mu & 1) ? -1.f : 1.f;
    cf32 v[8], o[8];
#pragma unroll
    for (int b1 = 0; b1 < 8; ++b1) {
      const cf32 zk = SC_F3_LD64(zk0 + (4 * b1 + (mu >> 1)) * L::TRS);
      const cf32 zm = SC_F3_LD64(zm0 + (4 * b1 + (mu >> 1)) * L::TRS);
      // member 0: A = (Z[k] + conj Z[-k]) / 2;  member 1: B = -i (Z[k] - conj Z[-k]) / 2
      const cf32 sres = cf_make(0.5f * (zk.x + sg * zm.x), 0.5f * (zk.y - sg * zm.y));
      v[b1] = (mu & 1) ? cf_make(sres.y, -sres.x) : sres;
    }
    dft8<-1>(v, o);                                      // over b1 -> q1
#pragma unroll
    for (int q1 = 0; q1 < 8; ++q1) {
      const cf32 y = (q1 == 0) ? o[0] : cf_mul_pk(o[q1], tw64[mu * 8 + q1]);
      if (act) cb[q1 * 8 + mu] = y;
    }
    SC_WAVE_SYNC();                                      // the 8 lanes of a column share a wave
#pragma unroll
    for (int m = 0; m < 8; m += 2) SC_F3_LD128(cb + mu * 8 + m, v[m], v[m + 1]);   // lane mu now plays q1 = mu
    dft8<-1>(v, o);                                      // over mu -> q2 : F_a[q1 + 8 q2]
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
    SC_WAVE_SYNC();                                      // cb is rewritten by the next task
  };

  // software prefetch, PFD rounds deep: the values of round t + depth are requested while round t
  // is transformed (depth register sets, rounds alternate between them).  A round is short (~1 us), about the
  // loaded HBM latency: depth 2 covers it alone at 3 workgroups per CU, depth 1 relies on the other resident
  // workgroups (4 per CU) and frees the 16 registers that make the fourth one fit
  cf32 pz[PFD][8];                            // .x = row A, .y = row B of the pair
  auto prefetch = [&](const int t, cf32 (&q)[8]) {        // t = 4 a + r
    if (t < 4 * P) {
      const int a = t >> 2, p = (t & 3) * 8 + hw;
      const IO* ra = xi + (int64_t)(P * (2 * p) + a) * SC_F2D_W;
      const IO* rb = ra + P * SC_F2D_W;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#ifdef SC_F3_ABL_NOLOAD
        q[j] = cf_make((float)(lam + j + t), (float)(lam - j));
#else
        q[j].x = f3_load(ra, lam, j);
        q[j].y = f3_load(rb, lam, j);
#endif
      }
    }
  };
  SC_COMPILER_FENCE();                                   // the table requests stay ahead of the row rounds (the compiler
                                                         // had sunk one below them and then drained everything for it)
#pragma unroll
  for (int d = 0; d < PFD; ++d) prefetch(d, pz[d]);        // depth 1, 2 or 4 (4 rounds = one row group)
  SC_COMPILER_FENCE();
  // the tables (requested ahead of the row rounds: vmcnt is in order, so waiting for them leaves the rounds in flight)
  sc_wait_vmcnt<(16 * PFD < 63 ? 16 * PFD : 63)>();
#pragma unroll
  for (int q = 0; q < NTH; ++q) sc_landed(rH[q]);
  sc_landed(r64);
  sc_landed(r2);
#pragma unroll
  for (int q = 0; q < NTH; ++q)
    if (tid + 256 * q < H) twH[tid + 256 * q] = rH[q];
  if (tid < 64) tw64[tid] = r64;
  if (tid < 32) tw2t[tid] = ctw4_make((t2k >= 4) ? cf_rot_i(r2, t2n) : r2);
  SC_SYNC();

#pragma unroll 1
  for (int a = 0; a < P; ++a) {
    // ---------------- rows of group a -> T'[p][32 + k], 4 rounds of 8 row pairs ----------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = r * 8 + hw;
      cf32 v[8], o[8];
      dft8<-1>(pz[r % PFD], o);               // over n1 (n = 32 n1 + lam) -> k1
      prefetch(4 * a + r + PFD, pz[r % PFD]);   // refill the consumed register set
#ifdef SC_F3_ABL_NOROW
      if (o[0].x == 1234.5f) T[tid] = o[1];
      continue;
#endif
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1)
#if defined(SC_F3_TW1_CS)
        xb[k1 * SC_F3_XRS + lam] = (k1 == 0) ? o[0] : cf_mul_cs(o[k1], tw1[k1]);
#else
        xb[k1 * SC_F3_XRS + lam] = (k1 == 0) ? o[0] : cf_mul_tw(o[k1], tw1[k1].c, tw1[k1].ns, tw1[k1].s);
#endif
      SC_WAVE_SYNC();
#ifndef SC_F3_SWAP
#pragma unroll
      for (int n3 = 0; n3 < 8; ++n3) v[n3] = SC_F3_LD64(xb + k1l * SC_F3_XRS + 4 * n3 + n4);
      SC_WAVE_SYNC();
      dft8<-1>(v, o);                                    // over n3 (n2 = 4 n3 + n4) -> k3
      // last stage (over n4, k = k1 + 8 k3 + 64 k4) is pruned to one k4 per (k1, k3): k4 = 0 for
      // k3 < 4 (k = 0..31), k4 = 3 for k3 >= 4 (k = -32..-1), plus k = +32 (k1 = 0, k3 = 4, k4 = 0):
      // write the twiddled terms, the consumer lane adds the four n4 contributions
      // layout of this second exchange: [k3][lane] (lane = 4 k1 + n4), so the writes are one
      // contiguous 256-byte row per instruction and the consumer's four terms are 32 contiguous bytes
#pragma unroll
      for (int k3 = 0; k3 < 8; ++k3)
        xb[32 * k3 + lam] = (k3 == 0) ? o[0] : cf_mul_tw(o[k3], tw2[4 * k3].c, tw2[4 * k3].ns, tw2[4 * k3].s);
      if (k1l == 0) xb[256 + n4] = cf_mul_tw(o[4], tw2[0].c, tw2[0].ns, tw2[0].s);   // k = +32 terms
      SC_WAVE_SYNC();
      {
        // lane j: positive k = j  (k1 = j & 7, k3 = j >> 3), negative f = j - 32 (k3 + 4)
        const cf32* pp = xb + 32 * (lam >> 3) + 4 * (lam & 7);
        const cf32 zp = cf_add(cf_add(pp[0], pp[1]), cf_add(pp[2], pp[3]));
        const cf32 zn = cf_add(cf_add(pp[128], pp[129]), cf_add(pp[130], pp[131]));
        cf32* tr = T + p * L::TRS;
        tr[32 + lam] = zp;
        tr[lam] = zn;
        if (lam == 0) {                                  // column 32 is transformed after the group loop
          const cf32* px = xb + 256;
          tr[65 + 2 * a] = cf_add(cf_add(px[0], px[1]), cf_add(px[2], px[3]));   // Z[+32]
          tr[66 + 2 * a] = zn;                                                     // Z[-32]
        }
      }
      SC_WAVE_SYNC();                                   // xb is rewritten by the next round
    }
#else
#pragma unroll
      for (int n3 = 0; n3 < 8; ++n3) v[n3] = SC_F3_LD64(xr + 4 * n3);
      SC_WAVE_SYNC();                                   // xb is rewritten by the next round
      dft8<-1>(v, o);                                    // over n3 (n2 = 4 n3 + n4) -> k3
      // last stage (over n4, k = k1 + 8 k3 + 64 k4), pruned to one k4 per (k1, k3): k4 = 0 for k3 < 4 (k = 0..31),
      // k4 = 3 for k3 >= 4 (k = -32..-1), plus k = +32 (k1 = 0, k3 = 4, k4 = 0).  Twiddle, then sum the four n4
      // lanes (l, l ^ 16, l ^ 32, l ^ 48) by a reduce-scatter in registers
      cf32 t[8];
      t[0] = o[0];
#pragma unroll
      for (int k3 = 1; k3 < 8; ++k3) t[k3] = cf_mul_tw(o[k3], tw2[4 * k3].c, tw2[4 * k3].ns, tw2[4 * k3].s);
      cf32 e = cf_mul_tw(o[4], tw2[0].c, tw2[0].ns, tw2[0].s), e2 = e;   // k = +32 term (used where k1 = 0)
#pragma unroll
      for (int i = 0; i < 4; ++i) {                      // bit 4: lower lane keeps k3 = i, upper k3 = i + 4
        sc_swap16(t[i].x, t[i + 4].x);
        sc_swap16(t[i].y, t[i + 4].y);
        t[i] = cf_add(t[i], t[i + 4]);
      }
      sc_swap16(e.x, e2.x);                              // (all-reduce: both lanes get own + partner)
      sc_swap16(e.y, e2.y);
      e = cf_add(e, e2);
      e2 = e;
#pragma unroll
      for (int i = 0; i < 2; ++i) {                      // bit 5: lower half keeps i, upper half i + 2
        sc_swap32(t[i].x, t[i + 2].x);
        sc_swap32(t[i].y, t[i + 2].y);
        t[i] = cf_add(t[i], t[i + 2]);
      }
      sc_swap32(e.x, e2.x);
      sc_swap32(e.y, e2.y);
      e = cf_add(e, e2);
      {
        // this lane holds k3 = 4 b4 + 2 b5 + {0, 1} of (pair sq, k1): k = k1 + 8 k3 -> tile column 32 + k (k3 < 4)
        // or k - 32 (k3 >= 4, i.e. frequency k - 64)
        const int b4 = n4 & 1, b5 = n4 >> 1;
        cf32* tr = T + (r * 8 + 2 * w + sq) * L::TRS;
        const int col = (b4 ? 0 : 32) + k1l + 16 * b5;
        tr[col] = t[0];
        tr[col + 8] = t[1];
        if (k1l == 0 && b5 == 0) {                       // column 32 is transformed after the group loop
          if (b4 == 0) tr[65 + 2 * a] = e;               // Z[+32]
          else tr[66 + 2 * a] = t[0];                    // Z[-32]  (k1 = 0, k3 = 4)
        }
      }
    }
#endif
    SC_SYNC();
    // ---------------- 33 column FFTs of 64 points on T' ----------------
#ifndef SC_F3_ABL_NOCOL
    column(T + 32 + (8 * w + cl), T + 32 - (8 * w + cl), xch + (8 * w + cl) * SC_F3_CCS, a, sc_int<0>(), nullptr, true);
#endif
    SC_SYNC();
  }
  // ---------------- 33rd column (k = 32): one 8-lane task per group, all groups at once ----------
  cf32* part = xch + (P + 1) * SC_F3_CCS;                // [P][64] partial spectra, summed below
#ifndef SC_F3_ABL_NOCOL32
  if (My > 32) {
    constexpr int ABLK = (P + 3) / 4;                    // lane blocks (of 8) per wave that carry a task
    const int a = ((cl % ABLK) << 2) | w;                // wave w, lane block cl -> group a
    const int ac = a < P ? a : 0;
    column(T + 65 + 2 * ac, T + 66 + 2 * ac, xch + ac * SC_F3_CCS, ac, sc_int<1>(), part + ac * 64,
           cl < ABLK && a < P);
  }
#endif
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
  } else {                                                // sharded spectrum (include/sc_engine.h, sc_spectrum_shards)
    for (int i = tid; i < Mx * My; i += 256) xhat[f3_shard_index(sh, img, i, My)] = OUT[i];
  }
}

// ------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------
#ifndef SC_F3_INV_WGS
#define SC_F3_INV_WGS 3       // persistent workgroups per compute unit
#endif
#ifndef SC_F3_INV_OCC
#define SC_F3_INV_OCC 3       // register budget: waves per SIMD the allocator leaves room for
#endif
// workgroups per compute unit of an instantiation = its register budget (waves per SIMD) = its persistent grid.
// H = 512 with the skip + GELU epilogue (EPI = 2) does not fit the 168 registers of three per unit (round 4: 1-6
// spilled registers, 8 / 20 bytes of scratch): two per unit, no scratch.
template <int H, int EPI>
constexpr int f3_inv_wgs() { return (H == 512 && EPI == 2) ? 2 : (H <= 256 && EPI == 0 ? SC_F3_INV_OCC : 3); }
template <int H, typename IO, int EPI = 0>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, (f3_inv_wgs<H, EPI>()))
k_fft2d_inv3(const cf32* __restrict__ yhat, IO* __restrict__ y, const float* __restrict__ bias,
             int channels, const cf32* __restrict__ tabW, const cf32* __restrict__ tabH, int Mx, int My,
             float s_dc, float s_other, const IO* __restrict__ skip, IO* __restrict__ preact, F3Shard sh,
             int64_t n_images, int gstride) {
  constexpr int P = H / 64;
  typedef F3Lds<H> L;
  SC_SHARED __attribute__((aligned(16))) unsigned char smem[L::total];
  cf32* xch = reinterpret_cast<cf32*>(smem + L::off_xch);
  cf32* T = reinterpret_cast<cf32*>(smem + L::off_T);
  cf32* twH = reinterpret_cast<cf32*>(smem + L::off_twH);
  cf32* tw64 = reinterpret_cast<cf32*>(smem + L::off_tw64);
  ctw4* tw2t = reinterpret_cast<ctw4*>(smem + L::off_tw2);
  cf32* y32 = reinterpret_cast<cf32*>(smem + L::off_c32);

  const int tid = SC_TID;
  const int w = SC_UNIFORM(tid >> 6);
  const int lane = tid & 63, hs = lane >> 5, lam = lane & 31;
  const int hw = w * 2 + hs;

  for (int i = tid; i < H; i += 256) twH[i] = tabH[i];
  if (tid < 64) tw64[tid] = tabW[(4 * (tid >> 3) * (tid & 7)) & 255];
  if (tid < 32) {
    // conjugates of the forward table: [k3][n4] = conj(w32^(n4 k3) i^(n4 [k3 >= 4])), [0][n4] = conj(w32^(4 n4))
    const int tn = tid & 3, tk = tid >> 2;
    const cf32 t = tabW[(8 * tn * (tk == 0 ? 4 : tk)) & 255];
    tw2t[tid] = ctw4_make(cf_conj((tk >= 4) ? cf_rot_i(t, tn) : t));
  }

  const int k1l = lam >> 2, n4 = lam & 3;
  ctw3 tw1c[8];
#pragma unroll
  for (int k = 1; k < 8; ++k) tw1c[k] = ctw3_make(cf_conj(tabW[(lam * k) & 255]));
  const ctw4* tw2c = tw2t + n4;                          // LDS table [k3][n4]
  cf32* xb = xch + hw * 8 * SC_F3_XRS;

  const int cl = lane >> 3, mu = lane & 7;
  // Round 3, second session: PERSISTENT workgroups (images b, b + gstride, ...; the host launches THREE per compute
  // unit).  A lane's spectrum entries -- column c = 8 w + cl, rows q = mu + 8 q2 (fx = q or q - 64) -- and the parked
  // 33rd column come straight from global memory into registers (the block was staged through LDS before: two
  // workgroup barriers and an exposed latency at the start of every image) and those of the NEXT image are requested
  // as soon as the last group's column transforms have consumed this image's, i.e. they land during the last four
  // row rounds.  Requests are unconditional (entries outside the kept block read element 0 and are zeroed when they
  // are scaled).  Measured (MI355X, metric shape, profiles/r03s2_fft3_persistent_ab.txt): 106.9 us one image per
  // workgroup -> 103-105 us with the register staging alone -> 99.2 us persistent at 3 workgroups per compute unit
  // (4 per unit: 112 us, 2: 110 us) against 97 us for a pure non-temporal writer of the same bytes.
  cf32 yh[8], y32r;
  auto request = [&](const int64_t im) {
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
  IO* yo = y + img * (int64_t)H * SC_F2D_W;
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
    if (tid < 64) {                                      // 33rd column parked in LDS, indexed by q
      const int row = f2d_fx(t_o) + Mx / 2;
      y32[tid] = (row >= 0 && row < Mx && 32 < My) ? cf_scale(y32r, s_other) : cf_make(0.f, 0.f);
    }
  }
  SC_SYNC();

  // one inverse column task of group a: spectrum column (yh regs or the parked 33rd column) ->
  // 64 rows, written to T[b][cdst]; cb = the task's private exchange patch
  auto column = [&](const int cdst, cf32* cb, const int a, auto extra_tag, const bool act) {
    constexpr bool EXTRA = decltype(extra_tag)::value != 0;
    cf32 v[8], o[8];
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) {
      const cf32 src = EXTRA ? y32[mu + 8 * q2] : yh[q2];
      if constexpr (P == 1) {                              // a = 0: the group twiddle is 1 (F3_NOTE_PK_MUL_LX, sc_kernels_fft3mx.h)
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
    for (int q1 = 0; q1 < 8; q1 += 2) SC_F3_LD128(cb + mu * 8 + q1, v[q1], v[q1 + 1]);   // lane mu now plays m = mu
    dft8<+1>(v, o);                                      // over q1 -> b1
    if (act) {
#pragma unroll
      for (int b1 = 0; b1 < 8; ++b1) T[(mu + 8 * b1) * L::URS + cdst] = o[b1];
    }
    SC_WAVE_SYNC();
  };

  // 33rd column (k = 32) of every group up front, one 8-lane task per group, parked in the tile's
  // spare columns 33 + a (the regular column tasks only write columns 0..31)
  if (My > 32) {
    constexpr int ABLK = (P + 3) / 4;
    const int a = ((cl % ABLK) << 2) | w;
    const int ac = a < P ? a : 0;
    column(33 + ac, xch + ac * SC_F3_CCS, ac, sc_int<1>(), cl < ABLK && a < P);
  } else if (tid < 64) {
#pragma unroll
    for (int a = 0; a < P; ++a) T[tid * L::URS + 33 + a] = cf_make(0.f, 0.f);
  }
  SC_SYNC();

#pragma unroll 1
  for (int a = 0; a < P; ++a) {
    // ---------------- 32 inverse column FFTs (64 points) -> T[b][c], rows h = P b + a ------------
    column(8 * w + cl, xch + (8 * w + cl) * SC_F3_CCS, a, sc_int<0>(), true);
    if (a == P - 1) request(img + gstride < n_images ? img + gstride : img);   // next image's entries (this one's are spent)
    SC_SYNC();
    // ---------------- rows: Hermitian-extended, zero-padded C2R of packed row pairs ---------------
    // (the four rounds unrolled: 99.2-100.2 against 100.5-100.8 us rolled; the epilogue variants keep the rolled loop --
    // with the skip values of a round in flight the unrolled body spills)
#ifdef SC_F3_INV_ROLLED_R
#pragma unroll 1
#else
#pragma unroll(EPI == 0 ? 4 : 1)
#endif
    for (int r = 0; r < 4; ++r) {
      const int p = r * 8 + hw;
      // block epilogue: the round's 16 skip values are requested here and consumed ~1 us later, after the row
      // transforms (requested next to the stores, every load waits for the store in front of it: vmcnt is in order)
      cf32 sk[8];
      if (EPI >= 1) {
        const int64_t io = img * (int64_t)H * SC_F2D_W;
        const IO* sa = skip + io + (int64_t)(P * (2 * p) + a) * SC_F2D_W;
        const IO* sb = skip + io + (int64_t)(P * (2 * p + 1) + a) * SC_F2D_W;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
          sk[n1].x = f3_load(sa, lam, n1);
          sk[n1].y = f3_load(sb, lam, n1);
        }
      }
      {
        // lane j builds Z[j] (k1 = j & 7, k3 = j >> 3) and Z[j - 32] (k3 + 4); lane 0 also Z[+32]
        const cf32* ta = T + (2 * p) * L::URS;
        const cf32* tb = ta + L::URS;
        const cf32 ua = SC_F3_LD64(ta + lam), ub = SC_F3_LD64(tb + lam);
        const int cm = (lam == 0) ? 33 + a : 32 - lam;   // column 32 lives in the group's spare column
        const cf32 va = SC_F3_LD64(ta + cm), vb = SC_F3_LD64(tb + cm);
        const cf32 zp = (lam == 0) ? cf_make(ua.x + badd, ub.x + badd) : cf_add_i(ua, ub);
        const cf32 zn = cf_conj_add_i(va, vb);
        cf32* zr = xb + (lam & 7) * SC_F3_XRS + (lam >> 3);
        zr[0] = zp;
        zr[4] = zn;
        if (lam < 8) zr[8] = (lam == 0) ? cf_add_i(va, vb) : cf_make(0.f, 0.f);
      }
      SC_WAVE_SYNC();
      cf32 v[8], o[8];
      {
        const cf32* zr = xb + k1l * SC_F3_XRS;
        cf32 xk[9];
#pragma unroll
        for (int k3 = 0; k3 < 8; k3 += 2) SC_F3_LD128(zr + k3, xk[k3], xk[k3 + 1]);
        xk[8] = SC_F3_LD64(zr + 8);
        v[0] = xk[0];
#pragma unroll
        for (int k3 = 1; k3 < 8; ++k3) v[k3] = cf_mul_tw(xk[k3], tw2c[4 * k3].c, tw2c[4 * k3].ns, tw2c[4 * k3].s);
        v[4] = cf_add(v[4], cf_mul_tw(xk[8], tw2c[0].c, tw2c[0].ns, tw2c[0].s));
      }
      SC_WAVE_SYNC();
      dft8<+1>(v, o);                                    // over k3 -> n3
#pragma unroll
      for (int n3 = 0; n3 < 8; ++n3) xb[k1l * SC_F3_XRS + 4 * n3 + n4] = o[n3];
      SC_WAVE_SYNC();
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) {
        const cf32 t = SC_F3_LD64(xb + k1 * SC_F3_XRS + lam);
        v[k1] = (k1 == 0) ? t : cf_mul_tw(t, tw1c[k1].c, tw1c[k1].ns, tw1c[k1].s);
      }
      SC_WAVE_SYNC();                                   // xb is rewritten by the next round
      dft8<+1>(v, o);                                    // over k1 -> n1 : z[32 n1 + lam]
      const int64_t oa = (int64_t)(P * (2 * p) + a) * SC_F2D_W, ob = (int64_t)(P * (2 * p + 1) + a) * SC_F2D_W;
      IO* ra = yo + oa;
      IO* rb = yo + ob;
      if (EPI == 0) {
        if (SC_F3_INV_PLAIN_GROUPS > 0 && a < SC_F3_INV_PLAIN_GROUPS) {
#pragma unroll
          for (int n1 = 0; n1 < 8; ++n1) {
            f3_store_plain(ra, lam, n1, o[n1].x);
            f3_store_plain(rb, lam, n1, o[n1].y);
          }
        } else {
#pragma unroll
          for (int n1 = 0; n1 < 8; ++n1) {
            f3_store(ra, lam, n1, o[n1].x);
            f3_store(rb, lam, n1, o[n1].y);
          }
        }
      } else {
        const int64_t io = img * (int64_t)H * SC_F2D_W;
        IO* pa = preact ? preact + io + oa : nullptr;
        IO* pb = preact ? preact + io + ob : nullptr;
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
          f3_store_epi<EPI>(ra, sk[n1].x, pa, lam, n1, o[n1].x);
          f3_store_epi<EPI>(rb, sk[n1].y, pb, lam, n1, o[n1].y);
        }
      }
    }
    SC_SYNC();                                          // T is rewritten by the next group
  }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <int H, typename IO>
static void fft3_launch_fwd(const Fft2dPlan* fp, const IO* x, cf32* xhat, int64_t n_images, float s_dc,
                            float s_other, sc_stream_t st, F3Shard sh) {
  SC_LAUNCH((k_fft2d_fwd3<H, IO>), dim3((unsigned)n_images), dim3(256), 0, st, x, xhat, (const cf32*)fp->tabW,
            (const cf32*)fp->tabH, fp->Mx, fp->My, s_dc, s_other, sh);
}

template <int H, typename IO>
static void fft3_launch_inv(const Fft2dPlan* fp, const cf32* yhat, IO* y, const float* bias, int channels,
                            int64_t n_images, float s_dc, float s_other, sc_stream_t st, int epi = 0,
                            const IO* skip = nullptr, IO* preact = nullptr, F3Shard sh = F3Shard{0, 0}) {
#define SC_F3_INV(E)                                                                                        \
  SC_LAUNCH((k_fft2d_inv3<H, IO, E>), dim3((unsigned)(grid < n_images ? grid : n_images)), dim3(256), 0, st, yhat, y,  \
            bias, channels, (const cf32*)fp->tabW, (const cf32*)fp->tabH, fp->Mx, fp->My, s_dc, s_other, skip, preact,  \
            sh, n_images, (int)(grid < n_images ? grid : n_images))
  // persistent workgroups, SC_F3_INV_WGS per compute unit (3: measured best, see the kernel; 2 for the one
  // instantiation whose register budget is two waves per SIMD)
  int64_t grid = (int64_t)SC_F3_INV_WGS * sc_cu_count();
  if (epi == 2) {
    if (f3_inv_wgs<H, 2>() < SC_F3_INV_WGS) grid = (int64_t)f3_inv_wgs<H, 2>() * sc_cu_count();
    SC_F3_INV(2);
  } else if (epi == 1) {
    SC_F3_INV(1);
  } else {
    SC_F3_INV(0);
  }
#undef SC_F3_INV
}

// x / y: float32, or bfloat16 storage when the plan carries SC_PLAN_IO_BF16 (IO = sc_bf16)
template <typename IO>
static inline int fft3_forward(const Fft2dPlan* fp, int mode, const IO* x, cf32* xhat, int64_t n_images,
                               sc_stream_t st, std::string* err, F3Shard sh = F3Shard{0, 0}) {
  const float s_dc = (mode == 0) ? fp->sf : fp->si;
  const float s_other = (mode == 0) ? fp->sf : 2.f * fp->si;
  switch (fp->H) {
    case 64: fft3_launch_fwd<64, IO>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    case 128: fft3_launch_fwd<128, IO>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    case 256: fft3_launch_fwd<256, IO>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    case 512: fft3_launch_fwd<512, IO>(fp, x, xhat, n_images, s_dc, s_other, st, sh); break;
    default: *err = "sc_engine: fft2d: unsupported H"; return 1;
  }
  if (hipGetLastError() != hipSuccess) {
    *err = "sc_engine: launch of k_fft2d_fwd3 failed";
    return 1;
  }
  return 0;
}

template <typename IO>
static inline int fft3_inverse(const Fft2dPlan* fp, int mode, const cf32* yhat, const float* bias,
                               int64_t channels, IO* y, int64_t n_images, sc_stream_t st, std::string* err,
                               int epi = 0, const IO* skip = nullptr, IO* preact = nullptr,
                               F3Shard sh = F3Shard{0, 0}) {
  const float s_dc = (mode == 0) ? fp->si : fp->sf;
  const float s_other = (mode == 0) ? fp->si : 0.5f * fp->sf;
  switch (fp->H) {
    case 64: fft3_launch_inv<64, IO>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, epi, skip, preact, sh); break;
    case 128: fft3_launch_inv<128, IO>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, epi, skip, preact, sh); break;
    case 256: fft3_launch_inv<256, IO>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, epi, skip, preact, sh); break;
    case 512: fft3_launch_inv<512, IO>(fp, yhat, y, bias, (int)channels, n_images, s_dc, s_other, st, epi, skip, preact, sh); break;
    default: *err = "sc_engine: fft2d: unsupported H"; return 1;
  }
  if (hipGetLastError() != hipSuccess) {
    *err = "sc_engine: launch of k_fft2d_inv3 failed";
    return 1;
  }
  return 0;
}
