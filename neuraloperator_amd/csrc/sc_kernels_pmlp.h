// sc_kernels_pmlp.h -- the pointwise half of an FNO block in one pass:
//
//     out = act( W2 gelu(W1 y + b1) + b2 + gate (.) skip_src )
//
// i.e. the block's ChannelMLP (two 1 x 1 convolutions with a GELU in between, channel_mlp.py:60-119), its soft-gating
// skip connection (skip_connections.py:53-130) and the block's closing non-linearity (fno_block.py:399-412), which
// the reference runs as ~10 tensor-sized elementwise / conv1d passes.  Here a wave owns a tile of 32 pixels of one
// sample for all channels and never leaves its registers between the two products:
//
//   GEMM 1  H[hid][px] = W1[hid][c] X[c][px]      exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); the B operand X[c = 2 t + k][px]
//           comes straight from global memory (per channel row 32 consecutive pixels: one 128-byte segment per half-wave)
//   GELU    on the accumulator registers
//   GEMM 2  Z[out][px] = W2[out][hid] H[hid][px]  the accumulator of GEMM 1 IS the B operand: lane (px, k) of an
//           accumulator holds rows (v & 3) + 8 (v >> 2) + 4 k, so step v contracts the row pair (row(v, 0), row(v, 1)) and
//           the A operand (W2) is pre-arranged in that order -- the hidden activations never touch LDS or memory
//   epilogue + b2 + gate[c] skip_src[c][px], activation, store (same 128-byte segments)
//
// Weights live in LDS in MFMA lane order (one conflict-free ds_read_b32 per MFMA).  Memory traffic: y and skip_src read
// once, out written once (3 tensor passes); matrix work 2 x 32 MFMAs per 32-pixel tile at 64 -> 32 -> 64 channels, far
// under the time the 24 KB of the tile need to cross HBM.
#pragma once
#include "sc_kernels_mfma.h"
#include "sc_kernels_fft3.h"      // sc_gelu

// measurement builds only (scripts/pmlp_ablate.py): take one resource out of the backward pass to see what it costs
#ifdef SC_PMLP_ABL_NOMFMA
#define PMLP_MFMA(acc, a, b) ((acc)[0] = fmaf((a), (b), (acc)[0]))
#else
#define PMLP_MFMA(acc, a, b) sc_mfma_32x32x2((acc), (a), (b))
#endif
#ifdef SC_PMLP_ABL_NOATOMIC
#define PMLP_LDS_ADD(ptr, val) do { if ((val) == 12345.678f) *(ptr) = (val); } while (0)
#else
#define PMLP_LDS_ADD(ptr, val) SC_LDS_ADD((ptr), (val))
#endif
#ifdef SC_PMLP_ABL_NOGELU
#define PMLP_GELU(x) ((x) * 0.5f)
#define PMLP_GELU_GRAD(x) ((x) * 0.25f + 0.5f)
#define PMLP_GELU_BOTH(x, g, d) do { (g) = (x) * 0.5f; (d) = (x) * 0.25f + 0.5f; } while (0)
#define PMLP_GELU_BOTH2(x0, x1, g0, g1, d0, d1) do { PMLP_GELU_BOTH(x0, g0, d0); PMLP_GELU_BOTH(x1, g1, d1); } while (0)
#else
#define PMLP_GELU(x) sc_gelu(x)
#define PMLP_GELU_GRAD(x) sc_gelu_grad(x)
#define PMLP_GELU_BOTH(x, g, d) sc_gelu_both((x), (g), (d))
#define PMLP_GELU_BOTH2(x0, x1, g0, g1, d0, d1) sc_gelu_both_pair((x0), (x1), (g0), (g1), (d0), (d1))
#endif
#ifdef SC_PMLP_ABL_NOGELU
#define PMLP_GELU2(a, b) do { (a) *= 0.5f; (b) *= 0.5f; } while (0)
#else
#define PMLP_GELU2(a, b) sc_gelu_pair((a), (b))
#endif
#ifdef SC_PMLP_ABL_NOLOAD
#define PMLP_LOAD_PLAIN(ptr) ((float)((uintptr_t)(ptr) & 1023u) * 1e-3f)
#else
#define PMLP_LOAD_PLAIN(ptr) (*(ptr))
#endif
// gelu'(x0), gelu'(x1) (the two activations of the pair form are dead code here)
#define PMLP_GELU_GRAD2(x0, x1, d0, d1) do { float g0_, g1_; PMLP_GELU_BOTH2((x0), (x1), g0_, g1_, (d0), (d1)); } while (0)
#ifdef SC_PMLP_ABL_NOLOAD
#define PMLP_LOAD(ptr) ((float)((uintptr_t)(ptr) & 1023u) * 1e-3f)
#else
#define PMLP_LOAD(ptr) SC_LOAD_STREAM(ptr)
#endif
#ifdef SC_PMLP_ABL_NOSTORE
#define PMLP_STORE(ptr, val) do { if ((val) == 12345.678f) SC_STORE_STREAM((ptr), (val)); } while (0)
#else
#define PMLP_STORE(ptr, val) SC_STORE_STREAM((ptr), (val))
#endif

// accumulator register v of a lane in half `half` holds this row of the 32 x 32 tile
SC_HD int pmlp_row(const int v, const int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

// d/dx gelu(x) = Phi(x) + x phi(x)  (round 6: one reciprocal + one exponential -- sc_gelu_both, sc_device.h -- instead of the
// erf approximation's two transcendentals plus a second exponential for the density)
SC_DEVICE float sc_gelu_grad(const float x) {
  float g, d;
  sc_gelu_both(x, g, d);
  return d;
}

struct PmlpArgs {
  const float* x;         // (batch, 32 CI, spatial)
  const float* w1;        // (32 CH, 32 CI)
  const float* b1;        // (32 CH) or null
  const float* w2;        // (32 CO, 32 CH)
  const float* b2;        // (32 CO) or null
  const float* skip;      // (batch, 32 CO, spatial) when GATE
  const float* gate;      // (32 CO) when GATE
  float* out;             // (batch, 32 CO, spatial)
  int64_t n_tiles, spatial;
  int tiles_per_sample, n_wg;
};

template <int CI, int CH, int CO, bool GATE, int ACT>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 2)
k_pmlp_fwd(PmlpArgs g) {
  constexpr int C_IN = 32 * CI, C_HID = 32 * CH, C_OUT = 32 * CO, S1 = 16 * CI;
  SC_SHARED float A1[CH * S1 * 64];
  SC_SHARED float A2[CO * CH * 16 * 64];
  SC_SHARED float B1[C_HID], B2[C_OUT], GT[C_OUT];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  for (int i = tid; i < CH * S1 * 64; i += 256) {
    const int l = i & 63, s = (i >> 6) % S1, hm = (i >> 6) / S1;
    A1[i] = g.w1[(32 * hm + (l & 31)) * C_IN + 2 * s + (l >> 5)];
  }
  for (int i = tid; i < CO * CH * 16 * 64; i += 256) {
    const int l = i & 63, v = (i >> 6) & 15, hm = ((i >> 10) % CH), om = (i >> 10) / CH;
    A2[i] = g.w2[(32 * om + (l & 31)) * C_HID + 32 * hm + pmlp_row(v, l >> 5)];
  }
  for (int i = tid; i < C_HID; i += 256) B1[i] = g.b1 ? g.b1[i] : 0.f;
  for (int i = tid; i < C_OUT; i += 256) {
    B2[i] = g.b2 ? g.b2[i] : 0.f;
    GT[i] = GATE ? g.gate[i] : 0.f;
  }
  SC_SYNC();
  // addressing: a wave-uniform base (sample, first pixel of the tile, channel row of the step) + ONE 32-bit lane offset
  // per operand layout, so the ~100 loads / stores of a tile share two address registers
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);        // B-operand rows 2 s + half
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);    // accumulator rows pmlp_row(v, half)
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * 4 + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * 4) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);        // row offsets k * sp: computed at their use, not hoisted
    const float* xs = g.x + b * C_IN * sp + px0;
    const float* ss = GATE ? g.skip + b * C_OUT * sp + px0 : nullptr;
    float* os = g.out + b * C_OUT * sp + px0;
    const int hq = sc_opaque(half);                        // bias / gate reads are loop invariant: keep them in the loop
    float xr[CI * 16];
#pragma unroll
    for (int s = 0; s < CI * 16; ++s) xr[s] = SC_LOAD_STREAM(sc_at(xs + (int64_t)(2 * s) * sp, lo_b));
    SC_SCHED_BARRIER();
    sc_f32x16 acc1[CH];
#pragma unroll
    for (int hm = 0; hm < CH; ++hm) {
#pragma unroll
      for (int v = 0; v < 16; ++v) acc1[hm][v] = 0.f;
#pragma unroll
      for (int s0 = 0; s0 < S1; s0 += 8) {                 // 8 table reads, 8 MFMAs: bounded live ranges
#pragma unroll
        for (int s = s0; s < s0 + 8; ++s) sc_mfma_32x32x2(acc1[hm], A1[(hm * S1 + s) * 64 + lane], xr[s]);
        SC_SCHED_BARRIER();
      }
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        float h0 = acc1[hm][v] + B1[32 * hm + pmlp_row(v, hq)], h1 = acc1[hm][v + 1] + B1[32 * hm + pmlp_row(v + 1, hq)];
        sc_gelu_pair(h0, h1);
        acc1[hm][v] = h0;
        acc1[hm][v + 1] = h1;
        if ((v & 3) == 2) SC_SCHED_BARRIER();              // four evaluations in flight, not sixteen
      }
    }
#pragma unroll
    for (int om = 0; om < CO; ++om) {
      float sk[16];
      if (GATE) {
#pragma unroll
        for (int v = 0; v < 16; ++v) sk[v] = SC_LOAD_STREAM(sc_at(ss + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
      }
      sc_f32x16 acc2;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc2[v] = 0.f;
#pragma unroll
      for (int hm = 0; hm < CH; ++hm)
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) sc_mfma_32x32x2(acc2, A2[((om * CH + hm) * 16 + v) * 64 + lane], acc1[hm][v]);
          SC_SCHED_BARRIER();
        }
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        float val[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = 32 * om + pmlp_row(v + u, hq);
          val[u] = acc2[v + u] + B2[r];
          if (GATE) val[u] = fmaf(GT[r], sk[v + u], val[u]);
        }
        if (ACT == 1) sc_gelu_pair(val[0], val[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) SC_STORE_STREAM(sc_at(os + (int64_t)(32 * om + pmlp_row(v + u, 0)) * sp, lo_c), val[u]);
        if ((v & 3) == 2) SC_SCHED_BARRIER();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Session 2: the WHOLE pointwise side of a default FNO block forward in one pass (fno_block.py:377-414):
//     s  = conv + (Ws x + bs)          conv = the spectral convolution's output, Ws = the 1 x 1 linear skip
//     y  = gelu(s)  (ACT 1; y = s for the last block)      -> stored (the backward pass reads it) together with s
//     out = act( W2 gelu(W1 y + b1) + b2 + gate (.) x )
// Before: k_plin_fwd wrote the skip (x in, skip out), the inverse FFT's epilogue read it back and wrote y and s, and
// k_pmlp_fwd read y and x: 8 R-sized passes.  Here the inverse FFT writes conv plainly (1) and this pass reads conv and
// x and writes s, y, out (5): the skip never exists in memory and y is not read back.  The skip product uses the same
// operand table and k order as k_plin_fwd (same bits), its accumulators -- after the GELU -- are the B operand of the
// first MLP product through the accumulator-as-operand arrangement of the second one (W1 pre-arranged in accumulator
// row order over the input channels), so y never leaves the registers on its way into the MLP.
// ------------------------------------------------------------------------------------------
struct PblockArgs {
  const float* conv;      // (batch, C, spatial)
  const float* x;         // (batch, C, spatial): the block input (linear skip + gate)
  const float* ws;        // (C, C)
  const float* bs;        // (C) or null
  const float* w1;        // (32 CH, C)
  const float* b1;
  const float* w2;        // (C, 32 CH)
  const float* b2;
  const float* gate;      // (C)
  float* y;               // (batch, C, spatial)
  float* pre;             // (batch, C, spatial) with ACT 1, else null
  int pre_is_grad;        // SC_ACT_GELU_DGRAD: `pre` receives gelu'(s) instead of s
  float* out;
  int64_t n_tiles, spatial;
  int tiles_per_sample, n_wg;
};

template <int CC, int CH, int ACT>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 2)
k_pblock_fwd(PblockArgs g) {
  constexpr int C = 32 * CC, C_HID = 32 * CH, S1 = 16 * CC;
  SC_SHARED float AS[CC * S1 * 64];                        // Ws, B operand rows (2 s, 2 s + 1)
  SC_SHARED float A1[CH * CC * 16 * 64];                   // W1, K in accumulator row order over the C input channels
  SC_SHARED float A2[CC * CH * 16 * 64];                   // W2, K in accumulator row order over the hidden channels
  SC_SHARED float BS[C], B1[C_HID], B2[C], GT[C];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  for (int i = tid; i < CC * S1 * 64; i += 256) {
    const int l = i & 63, s = (i >> 6) % S1, om = (i >> 6) / S1;
    AS[i] = g.ws[(32 * om + (l & 31)) * C + 2 * s + (l >> 5)];
  }
  for (int i = tid; i < CH * CC * 16 * 64; i += 256) {
    const int l = i & 63, v = (i >> 6) & 15, om = (i >> 10) % CC, hm = (i >> 10) / CC;
    A1[i] = g.w1[(32 * hm + (l & 31)) * C + 32 * om + pmlp_row(v, l >> 5)];
  }
  for (int i = tid; i < CC * CH * 16 * 64; i += 256) {
    const int l = i & 63, v = (i >> 6) & 15, hm = ((i >> 10) % CH), om = (i >> 10) / CH;
    A2[i] = g.w2[(32 * om + (l & 31)) * C_HID + 32 * hm + pmlp_row(v, l >> 5)];
  }
  for (int i = tid; i < C_HID; i += 256) B1[i] = g.b1 ? g.b1[i] : 0.f;
  for (int i = tid; i < C; i += 256) {
    BS[i] = g.bs ? g.bs[i] : 0.f;
    B2[i] = g.b2 ? g.b2[i] : 0.f;
    GT[i] = g.gate[i];
  }
  SC_SYNC();
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);        // B-operand rows 2 s + half
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);    // accumulator rows pmlp_row(v, half)
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * 4 + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * 4) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);
    const int64_t base = b * C * sp + px0;
    const float* xs = g.x + base;
    const float* cs = g.conv + base;
    float* ys = g.y + base;
    float* ps = ACT == 1 ? g.pre + base : nullptr;
    float* os = g.out + base;
    const int hq = sc_opaque(half);
    float xr[CC * 16];
#pragma unroll
    for (int s = 0; s < CC * 16; ++s) xr[s] = PMLP_LOAD_PLAIN(sc_at(xs + (int64_t)(2 * s) * sp, lo_b));   // ordinary loads: read again below
    SC_SCHED_BARRIER();
    // ---- linear skip + conv -> s, y (both stored); y stays in the accumulator registers
    sc_f32x16 yv[CC];
#pragma unroll
    for (int om = 0; om < CC; ++om) {
      float cv[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) cv[v] = PMLP_LOAD(sc_at(cs + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
#pragma unroll
      for (int v = 0; v < 16; ++v) yv[om][v] = 0.f;
#pragma unroll
      for (int s0 = 0; s0 < S1; s0 += 8) {
#pragma unroll
        for (int s = s0; s < s0 + 8; ++s) PMLP_MFMA(yv[om], AS[(om * S1 + s) * 64 + lane], xr[s]);
        SC_SCHED_BARRIER();
      }
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        const int64_t ro0 = (int64_t)(32 * om + pmlp_row(v, 0)) * sp, ro1 = (int64_t)(32 * om + pmlp_row(v + 1, 0)) * sp;
        float s0 = cv[v] + (yv[om][v] + BS[32 * om + pmlp_row(v, hq)]);   // conv + skip, as the epilogue adds them
        float s1 = cv[v + 1] + (yv[om][v + 1] + BS[32 * om + pmlp_row(v + 1, hq)]);
        if (ACT == 1) {
          if (g.pre_is_grad) {                             // (wave-uniform) gelu and gelu' from one evaluation
            float d0, d1;
            PMLP_GELU_BOTH2(s0, s1, s0, s1, d0, d1);
            PMLP_STORE(sc_at(ps + ro0, lo_c), d0);
            PMLP_STORE(sc_at(ps + ro1, lo_c), d1);
          } else {
            PMLP_STORE(sc_at(ps + ro0, lo_c), s0);
            PMLP_STORE(sc_at(ps + ro1, lo_c), s1);
            PMLP_GELU2(s0, s1);
          }
        }
        PMLP_STORE(sc_at(ys + ro0, lo_c), s0);
        PMLP_STORE(sc_at(ys + ro1, lo_c), s1);
        yv[om][v] = s0;
        yv[om][v + 1] = s1;
        if ((v & 3) == 2) SC_SCHED_BARRIER();
      }
    }
    // ---- hidden layer: h = gelu(W1 y + b1), y as the B operand straight from its accumulators
    sc_f32x16 acc1[CH];
#pragma unroll
    for (int hm = 0; hm < CH; ++hm) {
#pragma unroll
      for (int v = 0; v < 16; ++v) acc1[hm][v] = 0.f;
#pragma unroll
      for (int om = 0; om < CC; ++om)
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) PMLP_MFMA(acc1[hm], A1[((hm * CC + om) * 16 + v) * 64 + lane], yv[om][v]);
          SC_SCHED_BARRIER();
        }
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        float h0 = acc1[hm][v] + B1[32 * hm + pmlp_row(v, hq)], h1 = acc1[hm][v + 1] + B1[32 * hm + pmlp_row(v + 1, hq)];
        PMLP_GELU2(h0, h1);
        acc1[hm][v] = h0;
        acc1[hm][v + 1] = h1;
        if ((v & 3) == 2) SC_SCHED_BARRIER();
      }
    }
    // ---- output layer + soft-gating skip + closing activation (as k_pmlp_fwd)
#pragma unroll
    for (int om = 0; om < CC; ++om) {
      float sk[16];
#pragma unroll
      for (int v = 0; v < 16; ++v)                                      // second (last) read of x: out of L2
        sk[v] = PMLP_LOAD(sc_at(xs + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
      sc_f32x16 acc2;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc2[v] = 0.f;
#pragma unroll
      for (int hm = 0; hm < CH; ++hm)
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) PMLP_MFMA(acc2, A2[((om * CH + hm) * 16 + v) * 64 + lane], acc1[hm][v]);
          SC_SCHED_BARRIER();
        }
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        float val[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = 32 * om + pmlp_row(v + u, hq);
          val[u] = fmaf(GT[r], sk[v + u], acc2[v + u] + B2[r]);
        }
        if (ACT == 1) PMLP_GELU2(val[0], val[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) PMLP_STORE(sc_at(os + (int64_t)(32 * om + pmlp_row(v + u, 0)) * sp, lo_c), val[u]);
        if ((v & 3) == 2) SC_SCHED_BARRIER();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward: everything recomputed from x (and skip_src) inside the tile -- nothing but x is saved by the forward.
//   gz  = gout (.) act'(z_pre)                     gskip = gate (.) gz,  ggate += sum_px gz (.) skip,  gb2 += sum_px gz
//   gh  = W2^T gz,  ghp = gh (.) gelu'(h_pre)      gb1 += sum_px ghp
//   gx  = W1^T ghp
//   gW2 += gz h^T,  gW1 += ghp x^T                 contraction over the 32 pixels of the tile: the operands are transposed
//                                                  through a per-wave LDS scratch (32 x 33 floats each)
// A operands (W1, W2, W2^T, W1^T in MFMA lane order, the latter three in the accumulator row order) are prepared once
// live in LDS; the weight gradients accumulate in MFMA accumulators over all tiles of a wave, are summed over the waves
// of the workgroup in LDS (fixed order) and leave as ONE partial per workgroup; two reduction stages add the partials
// in a fixed order (bit-reproducible).
// ------------------------------------------------------------------------------------------
template <int CI, int CH, int CO>
struct PmlpDims {
  static constexpr int C_IN = 32 * CI, C_HID = 32 * CH, C_OUT = 32 * CO, S1 = 16 * CI;
  static constexpr int nA1 = CH * S1 * 64, nA2 = CO * CH * 16 * 64, nA3 = CH * CO * 16 * 64, nA4 = CI * CH * 16 * 64;
  static constexpr int oA1 = 0, oA2 = oA1 + nA1, oA3 = oA2 + nA2, oA4 = oA3 + nA3, nTab = oA4 + nA4;
  // one partial: gw2 | gw1 | gb1 | gb2 | ggate
  static constexpr int oW2 = 0, oW1 = oW2 + C_OUT * C_HID, oB1 = oW1 + C_HID * C_IN, oB2 = oB1 + C_HID, oG = oB2 + C_OUT,
                       NP = oG + C_OUT;
};

struct PmlpBwdArgs {
  const float* x;
  const float* b1;
  const float* b2;
  const float* skip;
  const float* gate;
  const float* gout;
  const float* w1;
  const float* w2;
  const float* x_pre;      // optional: x = gelu(x_pre) was produced by the previous fused pass; gx is then the gradient
                           // with respect to x_pre (the gelu backward of the Fourier layer folded into this store path)
  int x_pre_is_grad;       // SC_ACT_GELU_DGRAD: x_pre holds gelu'(pre-activation): multiply, do not evaluate
  const float* lw;         // LIN kernels (round 6): the block's linear skip W_s (C, C); `gskip` then receives
                           // W_s^T gx + gate (.) gz -- the whole gradient of the block input outside the spectral convolution
  float* gx;
  float* gskip;
  float* partial;          // [n_wg][NP]
  int64_t n_tiles, spatial;
  int tiles_per_sample, n_wg;
};

// LIN (round 6, the fused block backward): the data path of the block's 1 x 1 linear skip rides along -- the gradient of the
// Fourier layer's pre-activation (this kernel's gx with x_pre) is the B operand of one more product, W_s^T gx, accumulated
// on top of the soft-gating branch's gate (.) gz: `gskip` leaves as the complete gradient of the block input outside the
// spectral convolution (the addend of sc_layer_backward_ex) instead of crossing HBM three more times
// (k_pmlp_bwd writes it, k_plin_bwd reads it and gx and writes the sum: 10 tensor passes -> 5 + the 2 of k_plin_bwd<.., false>,
// which is left with the skip's weight gradient).
template <int CI, int CH, int CO, bool GATE, int ACT, int NW, bool LIN = false>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(64 * NW, 1)
k_pmlp_bwd(PmlpBwdArgs g) {
  typedef PmlpDims<CI, CH, CO> D;
  constexpr int S1 = D::S1, TS = 32 * 33, NT = 64 * NW;
  constexpr int nA5 = LIN ? CI * CO * 16 * 64 : 0;
  // one wave per SIMD (NW = 4) hides memory latency by requesting the next output tile's rows / the x rows of phase E a
  // phase ahead (48 registers); with two waves per SIMD (NW = 8, round 6) the other wave covers it and the registers go
  constexpr bool PF = NW <= 4;
  static_assert(!LIN || (CI == CO && GATE), "LIN: the block's shape (gated skip of the block input, C -> hidden -> C)");
  static_assert((D::nTab + nA5 + NW * (1 + CH) * TS + D::NP + D::C_HID + 2 * D::C_OUT) * 4 <= 160 * 1024, "LDS budget");
  // one workgroup of NW (8, or 4 for the larger tables) waves per CU: the four operand tables (W1, W2, W2^T, W1^T in MFMA lane order) live in LDS once
  // for all of them (read from the workspace they cost an L2 round trip per group of MFMAs: 1.7 ms per launch at the
  // metric shape, profiles/r02_block_kernel_stats_v1.txt)
  SC_SHARED float tabs[D::nTab + nA5];
  SC_SHARED float scr[NW * (1 + CH) * TS];                // per wave: T_A and one T_B per hidden tile (h)
  SC_SHARED float red[D::NP];
  SC_SHARED float B1[D::C_HID], B2[D::C_OUT], GT[D::C_OUT];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  float* TA = scr + w * (1 + CH) * TS;
  float* TH = TA + TS;                                    // h tiles, transposed: TH[hm][row][px]
  const float* A1 = tabs + D::oA1;
  const float* A2 = tabs + D::oA2;
  const float* A3 = tabs + D::oA3;
  const float* A4 = tabs + D::oA4;
  const float* A5 = tabs + D::nTab;                       // LIN: W_s^T, K in accumulator row order (k_plin_bwd's AT)
  for (int i = tid; i < nA5; i += NT) {
    const int l = i & 63, v = (i >> 6) & 15, om = (i >> 10) % CO, ci = (i >> 10) / CO;
    tabs[D::nTab + i] = g.lw[(32 * om + pmlp_row(v, l >> 5)) * D::C_IN + 32 * ci + (l & 31)];
  }
  for (int i = tid; i < D::nTab; i += NT) {
    const int l = i & 63, m = l & 31, kk = l >> 5;
    float val;
    if (i < D::oA2) {
      const int j = (i - D::oA1) >> 6, s = j % D::S1, hm = j / D::S1;
      val = g.w1[(32 * hm + m) * D::C_IN + 2 * s + kk];
    } else if (i < D::oA3) {
      const int j = (i - D::oA2) >> 6, v = j & 15, hm = (j >> 4) % CH, om = (j >> 4) / CH;
      val = g.w2[(32 * om + m) * D::C_HID + 32 * hm + pmlp_row(v, kk)];
    } else if (i < D::oA4) {
      const int j = (i - D::oA3) >> 6, v = j & 15, om = (j >> 4) % CO, hm = (j >> 4) / CO;
      val = g.w2[(32 * om + pmlp_row(v, kk)) * D::C_HID + 32 * hm + m];
    } else {
      const int j = (i - D::oA4) >> 6, v = j & 15, hm = (j >> 4) % CH, ci = (j >> 4) / CH;
      val = g.w1[(32 * hm + pmlp_row(v, kk)) * D::C_IN + 32 * ci + m];
    }
    tabs[i] = val;
  }
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);        // B-operand rows 2 s + half
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);    // accumulator rows pmlp_row(v, half)
  // weight-gradient tiles: MFMA accumulators that live across the whole tile loop (one wave per SIMD: the register file
  // has room; adding every tile's contribution into a shared LDS image with ds_add_f32 cost 0.85 of 1.8 ms,
  // profiles/r02_pmlp_ablation.txt)
  sc_f32x16 aW2[CO][CH], aW1[CH][CI];
#pragma unroll
  for (int om = 0; om < CO; ++om)
#pragma unroll
    for (int hm = 0; hm < CH; ++hm)
#pragma unroll
      for (int v = 0; v < 16; ++v) aW2[om][hm][v] = 0.f;
#pragma unroll
  for (int hm = 0; hm < CH; ++hm)
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int v = 0; v < 16; ++v) aW1[hm][ci][v] = 0.f;
  float sB2[CO], sG[CO], sB1[CH];                          // row sums (bias / gate gradients): half a row per lane
#pragma unroll
  for (int om = 0; om < CO; ++om) sB2[om] = sG[om] = 0.f;
#pragma unroll
  for (int hm = 0; hm < CH; ++hm) sB1[hm] = 0.f;
  for (int i = tid; i < D::NP; i += NT) red[i] = 0.f;
  for (int i = tid; i < D::C_HID; i += NT) B1[i] = g.b1 ? g.b1[i] : 0.f;
  for (int i = tid; i < D::C_OUT; i += NT) {
    B2[i] = g.b2 ? g.b2[i] : 0.f;
    GT[i] = GATE ? g.gate[i] : 0.f;
  }
  SC_SYNC();
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * NW + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * NW) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);        // row offsets k * sp: computed at their use, not hoisted
    const float* xs = g.x + b * D::C_IN * sp + px0;
    const float* gs = g.gout + b * D::C_OUT * sp + px0;
    const float* ss = GATE ? g.skip + b * D::C_OUT * sp + px0 : nullptr;
    float* gks = GATE ? g.gskip + b * D::C_OUT * sp + px0 : nullptr;
    float* gxs = g.gx + b * D::C_IN * sp + px0;
    // ---- A: recompute h_pre and h; h also goes to LDS transposed (operand of the W2 gradient)
    // (the operand tables are loop invariant: without an opaque lane offset per phase the compiler hoists all ~130
    // table loads out of the tile loop and the kernel lives in scratch)
    float hp[CH][16], h[CH][16], gzn[16], skn[16];
    {
      const int ln1 = sc_opaque(lane), hq1 = sc_opaque(half);
      float xr[CI * 16];
#pragma unroll
      for (int s = 0; s < CI * 16; ++s) xr[s] = PMLP_LOAD(sc_at(xs + (int64_t)(2 * s) * sp, lo_b));
      // software pipeline over the tile: the gout / skip rows of output tile om + 1 are requested before tile om is
      // worked on (tile 0 here, next to x), the x rows of phase E before phase D -- one exposed memory latency per
      // pixel tile instead of one per phase
      if (PF) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int64_t ro = (int64_t)pmlp_row(v, 0) * sp;
          gzn[v] = PMLP_LOAD(sc_at(gs + ro, lo_c));
          skn[v] = GATE ? PMLP_LOAD(sc_at(ss + ro, lo_c)) : 0.f;
        }
      }
      SC_SCHED_BARRIER();
#pragma unroll
      for (int hm = 0; hm < CH; ++hm) {
        sc_f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int s0 = 0; s0 < S1; s0 += 8) {
#pragma unroll
          for (int s = s0; s < s0 + 8; ++s) PMLP_MFMA(acc, A1[(hm * S1 + s) * 64 + ln1], xr[s]);
          SC_SCHED_BARRIER();
        }
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          // (hp holds gelu'(h_pre) from here on: both come out of one evaluation, phase C needs nothing else of h_pre)
          PMLP_GELU_BOTH2(acc[v] + B1[32 * hm + pmlp_row(v, hq1)], acc[v + 1] + B1[32 * hm + pmlp_row(v + 1, hq1)],
                          h[hm][v], h[hm][v + 1], hp[hm][v], hp[hm][v + 1]);
          TH[hm * TS + pmlp_row(v, half) * 33 + n] = h[hm][v];
          TH[hm * TS + pmlp_row(v + 1, half) * 33 + n] = h[hm][v + 1];
        }
        SC_SCHED_BARRIER();
      }
    }
    // ---- B: one output tile at a time: gz, gskip, ggate / gb2 sums, gW2 += gz h^T, gh += W2^T gz
    sc_f32x16 a2[LIN ? CO : 1];                            // LIN: gate (.) gz now, + W_s^T gx in phase D
    sc_f32x16 gh[CH];
#pragma unroll
    for (int hm = 0; hm < CH; ++hm)
#pragma unroll
      for (int v = 0; v < 16; ++v) gh[hm][v] = 0.f;
#pragma unroll
    for (int om = 0; om < CO; ++om) {
      const int ln2 = sc_opaque(lane), hq2 = sc_opaque(half);
      float gz[16], sk[16];
      if (PF) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          gz[v] = gzn[v];
          sk[v] = skn[v];
        }
        if (om + 1 < CO) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int64_t ro = (int64_t)(32 * (om + 1) + pmlp_row(v, 0)) * sp;
            gzn[v] = PMLP_LOAD(sc_at(gs + ro, lo_c));
            skn[v] = GATE ? PMLP_LOAD(sc_at(ss + ro, lo_c)) : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int64_t ro = (int64_t)(32 * om + pmlp_row(v, 0)) * sp;
          gz[v] = PMLP_LOAD(sc_at(gs + ro, lo_c));
          sk[v] = GATE ? PMLP_LOAD(sc_at(ss + ro, lo_c)) : 0.f;
        }
      }
      SC_SCHED_BARRIER();
      if (ACT == 1) {
        sc_f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int hm = 0; hm < CH; ++hm)
#pragma unroll
          for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
            for (int v = v0; v < v0 + 8; ++v)             // (two waves per SIMD: h comes back from the wave's own LDS copy)
              PMLP_MFMA(acc, A2[((om * CH + hm) * 16 + v) * 64 + ln2],
                        PF ? h[hm][v] : TH[hm * TS + pmlp_row(v, hq2) * 33 + sc_opaque(n)]);
            SC_SCHED_BARRIER();
          }
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          float z[2], d[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int r = 32 * om + pmlp_row(v + u, hq2);
            z[u] = acc[v + u] + B2[r];
            if (GATE) z[u] = fmaf(GT[r], sk[v + u], z[u]);
          }
          PMLP_GELU_GRAD2(z[0], z[1], d[0], d[1]);
          gz[v] *= d[0];
          gz[v + 1] *= d[1];
        }
      }
      if (GATE) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int r = 32 * om + pmlp_row(v, hq2);
          if (LIN) a2[LIN ? om : 0][v] = GT[r] * gz[v];
          else PMLP_STORE(sc_at(gks + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c), GT[r] * gz[v]);
          TA[pmlp_row(v, half) * 33 + n] = gz[v] * sk[v];
        }
        SC_WAVE_SYNC();
#pragma unroll
        for (int t = 0; t < 16; ++t) sG[om] += TA[n * 33 + 2 * t + half];     // row sums: lane (row, half of the pixels)
        SC_WAVE_SYNC();
      }
      SC_SCHED_BARRIER();
      // transposed gz: T[row][px]; lane (m, kk) reads T[m][2 t + kk] as the A operand of step t
#pragma unroll
      for (int v = 0; v < 16; ++v) TA[pmlp_row(v, half) * 33 + n] = gz[v];
      SC_WAVE_SYNC();
#pragma unroll
      for (int hm = 0; hm < CH; ++hm) {
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
#pragma unroll
          for (int t = t0; t < t0 + 8; ++t) {
            const float a = TA[n * 33 + 2 * t + half];
            PMLP_MFMA(aW2[om][hm], a, TH[hm * TS + n * 33 + 2 * t + half]);
            if (hm == 0) sB2[om] += a;
          }
          SC_SCHED_BARRIER();
        }
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) PMLP_MFMA(gh[hm], A3[((hm * CO + om) * 16 + v) * 64 + ln2], gz[v]);
          SC_SCHED_BARRIER();
        }
      }
      SC_WAVE_SYNC();                                      // T_A is rewritten by the next tile
    }
    // ---- C: ghp = gh * gelu'(h_pre)
    float ghp[CH][16];
#pragma unroll
    for (int hm = 0; hm < CH; ++hm)
#pragma unroll
      for (int v = 0; v < 16; ++v) ghp[hm][v] = gh[hm][v] * hp[hm][v];
    SC_SCHED_BARRIER();
    float xe[16];
    if (PF) {
      const uint32_t lo_e = (uint32_t)sc_opaque((int)lo_b);     // a second read of x (L2), not phase A's values kept alive
#pragma unroll
      for (int t = 0; t < 16; ++t) xe[t] = PMLP_LOAD_PLAIN(sc_at(xs + (int64_t)(2 * t) * sp, lo_e));
    }
    SC_SCHED_BARRIER();
    // ---- D: gx = W1^T ghp
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const int ln4 = sc_opaque(lane);
      sc_f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
      for (int hm = 0; hm < CH; ++hm)
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) PMLP_MFMA(acc, A4[((ci * CH + hm) * 16 + v) * 64 + ln4], ghp[hm][v]);
          SC_SCHED_BARRIER();
        }
      if (g.x_pre) {
        const float* ps = g.x_pre + b * D::C_IN * sp + px0;
        float pv[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) pv[v] = PMLP_LOAD(sc_at(ps + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c));
        if (g.x_pre_is_grad) {
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[v] *= pv[v];
        } else {
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            float d0, d1;
            PMLP_GELU_GRAD2(pv[v], pv[v + 1], d0, d1);
            acc[v] *= d0;
            acc[v + 1] *= d1;
            if ((v & 3) == 2) SC_SCHED_BARRIER();
          }
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) PMLP_STORE(sc_at(gxs + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c), acc[v]);
      SC_SCHED_BARRIER();
      if (LIN) {                                           // W_s^T gx: this gx tile is the B operand for every input tile
        const int ln5 = sc_opaque(lane);
#pragma unroll
        for (int cip = 0; cip < CI; ++cip)
#pragma unroll
          for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
            for (int v = v0; v < v0 + 8; ++v)
              PMLP_MFMA(a2[LIN ? cip : 0], A5[((cip * CO + ci) * 16 + v) * 64 + ln5], acc[v]);
            SC_SCHED_BARRIER();
          }
      }
    }
    if (LIN) {
#pragma unroll
      for (int cip = 0; cip < CI; ++cip) {
#pragma unroll
        for (int v = 0; v < 16; ++v) PMLP_STORE(sc_at(gks + (int64_t)(32 * cip + pmlp_row(v, 0)) * sp, lo_c), a2[LIN ? cip : 0][v]);
        SC_SCHED_BARRIER();
      }
    }
    // ---- E: gW1 += ghp x^T over the pixels: ghp transposed in T_A, the x tile (re-read: L2 resident) in the first
    //         T_H buffer as X[c][px]
#pragma unroll
    for (int hm = 0; hm < CH; ++hm) {
#pragma unroll
      for (int v = 0; v < 16; ++v) TA[pmlp_row(v, half) * 33 + n] = ghp[hm][v];
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        if (!PF) {
          const uint32_t lo_e = (uint32_t)sc_opaque((int)lo_b);
#pragma unroll
          for (int t = 0; t < 16; ++t) xe[t] = PMLP_LOAD_PLAIN(sc_at(xs + (int64_t)(32 * ci + 2 * t) * sp, lo_e));
        }
        SC_WAVE_SYNC();                                    // readers of the previous X tile (and of h, first round)
#pragma unroll
        for (int t = 0; t < 16; ++t) TH[(2 * t + half) * 33 + n] = xe[t];
        if (PF && (ci + 1 < CI || hm + 1 < CH)) {          // next x rows while this tile's products run
          const uint32_t lo_e = (uint32_t)sc_opaque((int)lo_b);
          const int cn = (ci + 1 < CI) ? ci + 1 : 0;
#pragma unroll
          for (int t = 0; t < 16; ++t) xe[t] = PMLP_LOAD_PLAIN(sc_at(xs + (int64_t)(32 * cn + 2 * t) * sp, lo_e));
        }
        SC_WAVE_SYNC();
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
#pragma unroll
          for (int t = t0; t < t0 + 8; ++t) {
            const float a = TA[n * 33 + 2 * t + half];
            PMLP_MFMA(aW1[hm][ci], a, TH[n * 33 + 2 * t + half]);
            if (ci == 0) sB1[hm] += a;
          }
          SC_SCHED_BARRIER();
        }
      }
      SC_WAVE_SYNC();
    }
  }
  // ---- one partial per workgroup: the waves add their sums into the LDS image one after the other (fixed order);
  //      the two halves of a row sum live in lanes n and n + 32
  for (int turn = 0; turn < NW; ++turn) {
    if (w == turn) {
#pragma unroll
      for (int om = 0; om < CO; ++om)
#pragma unroll
        for (int hm = 0; hm < CH; ++hm)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            red[D::oW2 + (32 * om + pmlp_row(v, half)) * D::C_HID + 32 * hm + n] += aW2[om][hm][v];
#pragma unroll
      for (int hm = 0; hm < CH; ++hm)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            red[D::oW1 + (32 * hm + pmlp_row(v, half)) * D::C_IN + 32 * ci + n] += aW1[hm][ci][v];
      for (int hh = 0; hh < 2; ++hh) {
        if (half == hh) {
#pragma unroll
          for (int om = 0; om < CO; ++om) {
            red[D::oB2 + 32 * om + n] += sB2[om];
            red[D::oG + 32 * om + n] += sG[om];
          }
#pragma unroll
          for (int hm = 0; hm < CH; ++hm) red[D::oB1 + 32 * hm + n] += sB1[hm];
        }
        SC_WAVE_SYNC();
      }
    }
    SC_SYNC();
  }
  float* dst = g.partial + (int64_t)SC_BID_X * D::NP;
  for (int i = tid; i < D::NP; i += NT) dst[i] = red[i];
}

// stage 1: out[y][i] = sum_{k = y, y + groups, ...} in[k][i]   (grid: ceil(np / 256) x groups)
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_pmlp_reduce1(const float* __restrict__ in, int n_in, int groups, int np, float* __restrict__ out) {
  const int i = SC_BID_X * 256 + SC_TID, y = SC_BID_Y;
  if (i >= np) return;
  float s = 0.f;
  for (int k = y; k < n_in; k += groups) s += in[(int64_t)k * np + i];
  out[(int64_t)y * np + i] = s;
}

// stage 2: sums[i] = sum_y in[y][i] (y ascending), scattered to the five gradient tensors (null = not wanted)
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_pmlp_reduce(const float* __restrict__ partial, int n_wg, int np, int o_w1, int o_b1, int o_b2, int o_g,
              float* __restrict__ gw2, float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gb2,
              float* __restrict__ ggate) {
  const int i = SC_BID_X * 256 + SC_TID;
  if (i >= np) return;
  float s = 0.f;
  for (int k = 0; k < n_wg; ++k) s += partial[(int64_t)k * np + i];
  if (i < o_w1) gw2[i] = s;
  else if (i < o_b1) gw1[i - o_w1] = s;
  else if (i < o_b2) { if (gb1) gb1[i - o_b1] = s; }
  else if (i < o_g) { if (gb2) gb2[i - o_b2] = s; }
  else if (ggate) ggate[i - o_g] = s;
}

// ------------------------------------------------------------------------------------------
// 1 x 1 linear map over the channels (the block's linear skip, skip_connections.py:119-169: Conv1d with kernel size 1
// on the flattened grid): out = W x (+ b), one pass; backward gx = W^T g, gW = g x^T, gb = sum g, one pass.
// Same tiles, operand layouts and reduction as the MLP pass above (GEMM 1 only / GEMM 4 + the weight-gradient
// products only).
// ------------------------------------------------------------------------------------------
struct PlinArgs {
  const float* x;          // (batch, 32 CI, spatial)
  const float* w;          // (32 CO, 32 CI)
  const float* bias;       // (32 CO) or null
  const float* gout;       // backward: (batch, 32 CO, spatial)
  const float* addend;     // backward, optional: gx = W^T g + addend (another branch's gradient of the same input)
  float* out;              // forward: (batch, 32 CO, spatial); backward: gx (batch, 32 CI, spatial)
  float* partial;          // backward: [n_wg][32 CO * 32 CI + 32 CO]
  int64_t n_tiles, spatial;
  int tiles_per_sample, n_wg;
};

template <int CI, int CO>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 2)
k_plin_fwd(PlinArgs g) {
  constexpr int C_IN = 32 * CI, C_OUT = 32 * CO, S1 = 16 * CI;
  SC_SHARED float A[CO * S1 * 64];
  SC_SHARED float Bv[C_OUT];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  for (int i = tid; i < CO * S1 * 64; i += 256) {
    const int l = i & 63, s = (i >> 6) % S1, om = (i >> 6) / S1;
    A[i] = g.w[(32 * om + (l & 31)) * C_IN + 2 * s + (l >> 5)];
  }
  for (int i = tid; i < C_OUT; i += 256) Bv[i] = g.bias ? g.bias[i] : 0.f;
  SC_SYNC();
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * 4 + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * 4) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);
    const int hq = sc_opaque(half);
    const float* xs = g.x + b * C_IN * sp + px0;
    float* os = g.out + b * C_OUT * sp + px0;
    float xr[CI * 16];
#pragma unroll
    for (int s = 0; s < CI * 16; ++s) xr[s] = SC_LOAD_STREAM(sc_at(xs + (int64_t)(2 * s) * sp, lo_b));
    SC_SCHED_BARRIER();
#pragma unroll
    for (int om = 0; om < CO; ++om) {
      sc_f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
      for (int s0 = 0; s0 < S1; s0 += 8) {
#pragma unroll
        for (int s = s0; s < s0 + 8; ++s) sc_mfma_32x32x2(acc, A[(om * S1 + s) * 64 + lane], xr[s]);
        SC_SCHED_BARRIER();
      }
#pragma unroll
      for (int v = 0; v < 16; ++v)
        SC_STORE_STREAM(sc_at(os + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c), acc[v] + Bv[32 * om + pmlp_row(v, hq)]);
      SC_SCHED_BARRIER();
    }
  }
}

template <int CI, int CO>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 2)
k_plin_bwd(PlinArgs g) {
  constexpr int C_IN = 32 * CI, C_OUT = 32 * CO, TS = 32 * 33, NW = 4, NT = 256, NPW = C_OUT * C_IN, NP = NPW + C_OUT;
  SC_SHARED float AT[CI * CO * 16 * 64];                  // W^T as the A operand, K in accumulator row order
  SC_SHARED float scr[NW * 2 * TS];
  SC_SHARED float red[NP];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  float* TA = scr + w * 2 * TS;
  float* TB = TA + TS;
  for (int i = tid; i < CI * CO * 16 * 64; i += NT) {
    const int l = i & 63, v = (i >> 6) & 15, om = (i >> 10) % CO, ci = (i >> 10) / CO;
    AT[i] = g.w[(32 * om + pmlp_row(v, l >> 5)) * C_IN + 32 * ci + (l & 31)];
  }
  for (int i = tid; i < NP; i += NT) red[i] = 0.f;
  SC_SYNC();
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);
  sc_f32x16 aW[CO][CI];
  float sB[CO];
#pragma unroll
  for (int om = 0; om < CO; ++om) {
    sB[om] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int v = 0; v < 16; ++v) aW[om][ci][v] = 0.f;
  }
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * NW + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * NW) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);
    const int ln = sc_opaque(lane);
    const float* xs = g.x + b * C_IN * sp + px0;
    const float* gs = g.gout + b * C_OUT * sp + px0;
    float* gxs = g.out + b * C_IN * sp + px0;
    const float* ads = g.addend ? g.addend + b * C_IN * sp + px0 : nullptr;
    float gz[CO][16];
#pragma unroll
    for (int om = 0; om < CO; ++om)
#pragma unroll
      for (int v = 0; v < 16; ++v) gz[om][v] = SC_LOAD_STREAM(sc_at(gs + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
    SC_SCHED_BARRIER();
    // gx = W^T g
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      sc_f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
      for (int om = 0; om < CO; ++om)
#pragma unroll
        for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
          for (int v = v0; v < v0 + 8; ++v) sc_mfma_32x32x2(acc, AT[((ci * CO + om) * 16 + v) * 64 + ln], gz[om][v]);
          SC_SCHED_BARRIER();
        }
      if (ads) {
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] += SC_LOAD_STREAM(sc_at(ads + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c));
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) SC_STORE_STREAM(sc_at(gxs + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c), acc[v]);
      SC_SCHED_BARRIER();
    }
    // gW += g x^T over the pixels of the tile (operands transposed through LDS), gb += row sums of g
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      float xe[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) xe[t] = SC_LOAD_STREAM(sc_at(xs + (int64_t)(32 * ci + 2 * t) * sp, lo_b));
      SC_WAVE_SYNC();
#pragma unroll
      for (int t = 0; t < 16; ++t) TB[(2 * t + half) * 33 + n] = xe[t];
#pragma unroll
      for (int om = 0; om < CO; ++om) {
        SC_WAVE_SYNC();
#pragma unroll
        for (int v = 0; v < 16; ++v) TA[pmlp_row(v, half) * 33 + n] = gz[om][v];
        SC_WAVE_SYNC();
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
#pragma unroll
          for (int t = t0; t < t0 + 8; ++t) {
            const float a = TA[n * 33 + 2 * t + half];
            sc_mfma_32x32x2(aW[om][ci], a, TB[n * 33 + 2 * t + half]);
            if (ci == 0) sB[om] += a;
          }
          SC_SCHED_BARRIER();
        }
      }
    }
    SC_WAVE_SYNC();
  }
  for (int turn = 0; turn < NW; ++turn) {
    if (w == turn) {
#pragma unroll
      for (int om = 0; om < CO; ++om)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
          for (int v = 0; v < 16; ++v) red[(32 * om + pmlp_row(v, half)) * C_IN + 32 * ci + n] += aW[om][ci][v];
      for (int hh = 0; hh < 2; ++hh) {
        if (half == hh) {
#pragma unroll
          for (int om = 0; om < CO; ++om) red[NPW + 32 * om + n] += sB[om];
        }
        SC_WAVE_SYNC();
      }
    }
    SC_SYNC();
  }
  float* dst = g.partial + (int64_t)SC_BID_X * NP;
  for (int i = tid; i < NP; i += NT) dst[i] = red[i];
}

// stage 2 for the linear map: gw | gb
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_plin_reduce(const float* __restrict__ partial, int n, int np, int o_b, float* __restrict__ gw, float* __restrict__ gb) {
  const int i = SC_BID_X * 256 + SC_TID;
  if (i >= np) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += partial[(int64_t)k * np + i];
  if (i < o_b) gw[i] = s;
  else if (gb) gb[i - o_b] = s;
}
