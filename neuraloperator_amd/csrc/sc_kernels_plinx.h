// sc_kernels_plinx.h -- 1 x 1 linear maps over the channels with the pointwise operations of an FNO block folded into their
// load and store paths (round 6, SURVEY.md section 8 row f1: blocks whose channel counts have no one-pass kernel --
// hidden 128, configs[4]'s width -- and rectangular maps).
//
// The one-pass kernels of sc_kernels_pmlp.h hold the whole MLP of a 32-pixel tile in registers; at 128 channels their
// operand tables and weight-gradient image no longer fit a compute unit (four 32 KB tables + 66 KB of gradients).  Here the
// same block is TWO of these passes each way (fc1, fc2) with the hidden activations crossing HBM once as their
// pre-activation -- and everything elementwise of channel_mlp.py:82-119, skip_connections.py:53-130 (soft gating) and
// fno_block.py:392-414 in the passes' own load / store paths, so that no ATen / MIOpen kernel is left between them
// (F.conv1d's fp32 backward at 128 channels lands on MIOpen's naive kernels: 338 ms per call, profiles/r06_block128_*):
//
//   forward   out = act( W xin + b + gate (.) skip ),  xin = x or gelu(x)  (XACT: x is the previous layer's pre-activation);
//             pre_out (optional) = the value before `act` (what the backward's gelu' needs)
//   backward  g   = gout (.) gelu'(pre)            (PRO)
//             gskip = gate (.) g,  ggate = sum_px g (.) skip        (GATE)
//             gx  = (W^T g + addend) (.) gelu'(xg)                  (addend, XGRAD optional; xg = x for an XACT layer: the
//                                                                    gradient then is with respect to the pre-activation)
//             gW  = g xin^T,  gb = sum_px g
//
// Tiles, operand layouts, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and the fixed-order reduction of the weight gradients
// are those of k_plin_fwd / k_plin_bwd; the options are wave-uniform runtime flags (one instantiation per channel pair).
// The weight gradient of CO output tiles needs CO * CI accumulator tiles: a launch owns at most OMN of the output tiles
// (template), the host launches ceil(CO / OMN) times, the first launch also produces gx / gskip.
// On MI355X the fp32 matrix instructions occupy the vector ALU (scripts/ubench_mfma_valu_overlap.hip: matrix + vector
// waves of one SIMD take the SUM of their times), so these kernels are bound by (matrix + vector cycles) or by their
// bytes, whichever is larger: occupancy beyond hiding the memory latency buys nothing.
#pragma once
#include "sc_kernels_pmlp.h"

#ifndef SC_PLX_XACT       // (include/sc_engine.h carries the same values for the callers)
#define SC_PLX_XACT 1     // the input is a pre-activation: xin = gelu(x)
#define SC_PLX_ACT 2      // forward: gelu on the output
#define SC_PLX_PRO 4      // backward: gout (.) gelu'(pre)
#define SC_PLX_XGRAD 8    // backward: gx (.) gelu'(xg)
#endif

struct PlinxArgs {
  const float* x;          // (batch, 32 CI, spatial)
  const float* w;          // (32 CO, 32 CI)
  const float* bias;       // forward: (32 CO) or null
  const float* skip;       // gated: (batch, 32 CO, spatial)
  const float* gate;       // gated: (32 CO)
  const float* gout;       // backward: (batch, 32 CO, spatial)
  const float* pre;        // backward, PRO: pre-activation of the forward output
  const float* xg;         // backward, XGRAD: (batch, 32 CI, spatial)
  const float* addend;     // backward, optional: added to W^T g
  float* out;              // forward output / backward gx
  float* pre_out;          // forward, optional
  float* gskip;            // backward, gated
  float* partial;          // backward: [n_wg][NP]
  int64_t n_tiles, spatial;
  int tiles_per_sample, n_wg, flags, do_gx;
};

// two workgroups per compute unit for every channel pair (the 128 -> 128 table is 64 KB): a wave's loads wait behind the
// other wave's products -- the matrix and vector work of the two do not overlap (see the header), the memory latency does
template <int CI, int CO>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 2)
k_plinx_fwd(PlinxArgs g) {
  constexpr int C_IN = 32 * CI, C_OUT = 32 * CO, S1 = 16 * CI;
  SC_SHARED float A[CO * S1 * 64];
  SC_SHARED float Bv[C_OUT], GT[C_OUT];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  for (int i = tid; i < CO * S1 * 64; i += 256) {
    const int l = i & 63, s = (i >> 6) % S1, om = (i >> 6) / S1;
    A[i] = g.w[(32 * om + (l & 31)) * C_IN + 2 * s + (l >> 5)];
  }
  for (int i = tid; i < C_OUT; i += 256) {
    Bv[i] = g.bias ? g.bias[i] : 0.f;
    GT[i] = g.gate ? g.gate[i] : 0.f;
  }
  SC_SYNC();
  const bool xact = (g.flags & SC_PLX_XACT) != 0, act = (g.flags & SC_PLX_ACT) != 0, gated = g.gate != nullptr;
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * 4 + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * 4) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);
    const int hq = sc_opaque(half);
    const float* xs = g.x + b * C_IN * sp + px0;
    const float* ss = gated ? g.skip + b * C_OUT * sp + px0 : nullptr;
    float* os = g.out + b * C_OUT * sp + px0;
    float* ps = g.pre_out ? g.pre_out + b * C_OUT * sp + px0 : nullptr;
    float xr[CI * 16];
#pragma unroll
    for (int s = 0; s < CI * 16; ++s) xr[s] = SC_LOAD_STREAM(sc_at(xs + (int64_t)(2 * s) * sp, lo_b));
    if (xact) {
#pragma unroll
      for (int s = 0; s < CI * 16; s += 2) sc_gelu_pair(xr[s], xr[s + 1]);
    }
    SC_SCHED_BARRIER();
#pragma unroll
    for (int om = 0; om < CO; ++om) {
      sc_f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
      float sk[16];
      if (gated) {
#pragma unroll
        for (int v = 0; v < 16; ++v) sk[v] = SC_LOAD_STREAM(sc_at(ss + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
      }
#pragma unroll
      for (int s0 = 0; s0 < S1; s0 += 8) {
#pragma unroll
        for (int s = s0; s < s0 + 8; ++s) sc_mfma_32x32x2(acc, A[(om * S1 + s) * 64 + lane], xr[s]);
        SC_SCHED_BARRIER();
      }
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        float val[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int r = 32 * om + pmlp_row(v + u, hq);
          val[u] = acc[v + u] + Bv[r];
          if (gated) val[u] = fmaf(GT[r], sk[v + u], val[u]);
          if (ps) SC_STORE_STREAM(sc_at(ps + (int64_t)(32 * om + pmlp_row(v + u, 0)) * sp, lo_c), val[u]);
        }
        if (act) sc_gelu_pair(val[0], val[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) SC_STORE_STREAM(sc_at(os + (int64_t)(32 * om + pmlp_row(v + u, 0)) * sp, lo_c), val[u]);
      }
      SC_SCHED_BARRIER();
    }
  }
}

// partial of one workgroup: gw rows of the launch's output tiles [OMN * 32][C_IN] | gb [OMN * 32] | ggate [OMN * 32]
template <int CI, int OMN>
struct PlinxDims {
  static constexpr int C_IN = 32 * CI, NPW = OMN * 32 * C_IN, oB = NPW, oG = NPW + OMN * 32, NP = oG + OMN * 32;
};

// OM0 = first output tile whose weight gradient this launch owns (runtime), OMN = how many (template)
// two workgroups per unit where four accumulator tiles and a 16 KB table leave room (<= 64 channels each way: the weight
// gradient of the metric block's linear skip runs through <2, 2, 2>): the second wave's loads wait behind the first's products
// LEAN = the weight / bias gradient alone with no option set (gx == NULL, flags == 0, no gate: what is left of the metric
// block's linear skip once its data path rides in k_pmlp_bwd<.., LIN>): the options fold away at compile time, the W^T
// table is not built, and two workgroups per unit fit where four accumulator tiles leave room.
#define SC_PLX_BWD_OCC(CI, CO, OMN, LEAN) (((LEAN) && (CI) * (OMN) <= 4) ? 2 : 1)
template <int CI, int CO, int OMN, bool LEAN = false>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, SC_PLX_BWD_OCC(CI, CO, OMN, LEAN))
k_plinx_bwd(PlinxArgs g, int om0) {
  typedef PlinxDims<CI, OMN> D;
  constexpr int C_IN = 32 * CI, C_OUT = 32 * CO, TS = 32 * 33, NW = 4, NT = 256;
  static_assert((CI * CO * 16 * 64 + NW * 2 * TS + D::NP + C_OUT) * 4 <= 160 * 1024, "LDS budget");
  SC_SHARED float AT[LEAN ? 64 : CI * CO * 16 * 64];      // W^T as the A operand, K in accumulator row order
  SC_SHARED float scr[NW * 2 * TS];
  SC_SHARED float red[D::NP];
  SC_SHARED float GT[C_OUT];
  const int tid = SC_TID, lane = tid & 63, n = lane & 31, half = lane >> 5;
  const int w = SC_UNIFORM(tid >> 6);
  float* TA = scr + w * 2 * TS;
  float* TB = TA + TS;
  const bool do_gx = !LEAN && g.do_gx != 0;
  if (do_gx) {
    for (int i = tid; i < CI * CO * 16 * 64; i += NT) {
      const int l = i & 63, v = (i >> 6) & 15, om = (i >> 10) % CO, ci = (i >> 10) / CO;
      AT[i] = g.w[(32 * om + pmlp_row(v, l >> 5)) * C_IN + 32 * ci + (l & 31)];
    }
  }
  for (int i = tid; i < D::NP; i += NT) red[i] = 0.f;
  for (int i = tid; i < C_OUT; i += NT) GT[i] = g.gate ? g.gate[i] : 0.f;
  SC_SYNC();
  const bool xact = !LEAN && (g.flags & SC_PLX_XACT) != 0, pro = !LEAN && (g.flags & SC_PLX_PRO) != 0;
  const bool xgrad = !LEAN && (g.flags & SC_PLX_XGRAD) != 0, gated = !LEAN && g.gate != nullptr;
  const uint32_t lo_b = 4u * (uint32_t)(n + half * g.spatial);
  const uint32_t lo_c = 4u * (uint32_t)(n + 4 * half * g.spatial);
  sc_f32x16 aW[OMN][CI];
  float sB[OMN], sG[OMN];
#pragma unroll
  for (int o = 0; o < OMN; ++o) {
    sB[o] = sG[o] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int v = 0; v < 16; ++v) aW[o][ci][v] = 0.f;
  }
#pragma unroll 1
  for (int64_t tile = (int64_t)SC_BID_X * NW + w; tile < g.n_tiles; tile += (int64_t)g.n_wg * NW) {
    const int64_t b = tile / g.tiles_per_sample;
    const int64_t px0 = (tile - b * g.tiles_per_sample) * 32;
    const int64_t sp = sc_opaque_s((int)g.spatial);
    const int ln = sc_opaque(lane), hq = sc_opaque(half);
    const float* xs = g.x + b * C_IN * sp + px0;
    const float* gs = g.gout + b * C_OUT * sp + px0;
    const float* prs = pro ? g.pre + b * C_OUT * sp + px0 : nullptr;
    const float* ss = gated ? g.skip + b * C_OUT * sp + px0 : nullptr;
    float* gks = gated ? g.gskip + b * C_OUT * sp + px0 : nullptr;
    float* gxs = g.out + b * C_IN * sp + px0;
    const float* ads = g.addend ? g.addend + b * C_IN * sp + px0 : nullptr;
    const float* xgs = xgrad ? g.xg + b * C_IN * sp + px0 : nullptr;
    // the output-channel gradient of the tiles this launch needs: all of them for gx, its own OMN for the weight gradient
    float gz[CO][16];
    // every row of g the tile needs is requested before the first is used, the rows of `pre` / `skip` one output tile ahead of
    // their use (one wave per SIMD: every exposed round trip counts -- with the loads inside the arithmetic hipcc waited for
    // each one: 64 us per tile, profiles/r06_plinx_batched_loads.txt; second pass: one exposed trip per 32-pixel tile, not one
    // per output tile -- at 64 -> 128 channels the trips were 45 % of the launch, profiles/r06_plinx_prefetch.txt)
    float prb[2][16], skb[2][16];
    auto wanted = [&](const int om) { return do_gx || (om >= om0 && om < om0 + OMN); };
    auto request = [&](const int om, float (&pr)[16], float (&sk)[16]) {
      if (pro) {
#pragma unroll
        for (int v = 0; v < 16; ++v) pr[v] = SC_LOAD_STREAM(sc_at(prs + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
      }
      if (gated) {
#pragma unroll
        for (int v = 0; v < 16; ++v) sk[v] = SC_LOAD_STREAM(sc_at(ss + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
      }
    };
#pragma unroll
    for (int om = 0; om < CO; ++om) {
      if (!wanted(om)) continue;
#pragma unroll
      for (int v = 0; v < 16; ++v) gz[om][v] = SC_LOAD_STREAM(sc_at(gs + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c));
    }
    if (wanted(0)) request(0, prb[0], skb[0]);
    SC_SCHED_BARRIER();
#pragma unroll
    for (int om = 0; om < CO; ++om) {
      const bool mine = om >= om0 && om < om0 + OMN;
      float (&pr)[16] = prb[om & 1];
      float (&sk)[16] = skb[om & 1];
      if (om + 1 < CO && wanted(om + 1)) request(om + 1, prb[(om + 1) & 1], skb[(om + 1) & 1]);
      SC_SCHED_BARRIER();
      if (!do_gx && !mine) continue;
      const bool need_sk = gated && (do_gx || mine);
      if (pro) {
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          float d0, d1;
          PMLP_GELU_GRAD2(pr[v], pr[v + 1], d0, d1);
          gz[om][v] *= d0;
          gz[om][v + 1] *= d1;
        }
      }
      if (need_sk) {
        if (do_gx) {
#pragma unroll
          for (int v = 0; v < 16; ++v)
            SC_STORE_STREAM(sc_at(gks + (int64_t)(32 * om + pmlp_row(v, 0)) * sp, lo_c), GT[32 * om + pmlp_row(v, hq)] * gz[om][v]);
        }
        if (mine) {                                        // ggate: row sums of g (.) skip through the wave's LDS patch
#pragma unroll
          for (int v = 0; v < 16; ++v) TA[pmlp_row(v, half) * 33 + n] = gz[om][v] * sk[v];
          SC_WAVE_SYNC();
          float s = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) s += TA[n * 33 + 2 * t + half];
#pragma unroll
          for (int o = 0; o < OMN; ++o) sG[o] += (om == om0 + o) ? s : 0.f;
          SC_WAVE_SYNC();
        }
      }
      SC_SCHED_BARRIER();
    }
    SC_SCHED_BARRIER();
    // the x rows of the weight-gradient phase are requested a step ahead (round 6, second pass): one wave per SIMD, so a
    // load issued where it is used costs its whole round trip -- four per tile at 128 input channels, 16 tiles per wave
    // at B = 8, 256^2: half of the launch that only owns weight-gradient tiles (profiles/r06_plinx_prefetch.txt)
    float xe[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) xe[t] = SC_LOAD_STREAM(sc_at(xs + (int64_t)(2 * t) * sp, lo_b));
    SC_SCHED_BARRIER();
    if (do_gx) {                                           // gx = (W^T g + addend) (.) gelu'(xg)
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        sc_f32x16 acc;
        float ad[16], xv[16];                              // requested ahead of the products that hide them
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        if (ads && !xgrad) {                               // (both at once: the addend is requested behind the products --
#pragma unroll                                             //  sixteen registers less at the kernel's widest point)
          for (int v = 0; v < 16; ++v) ad[v] = SC_LOAD_STREAM(sc_at(ads + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c));
        }
        if (xgrad) {
#pragma unroll
          for (int v = 0; v < 16; ++v) xv[v] = SC_LOAD_STREAM(sc_at(xgs + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c));
        }
        SC_SCHED_BARRIER();
#pragma unroll
        for (int om = 0; om < CO; ++om)
#pragma unroll
          for (int v0 = 0; v0 < 16; v0 += 8) {
#pragma unroll
            for (int v = v0; v < v0 + 8; ++v) sc_mfma_32x32x2(acc, AT[((ci * CO + om) * 16 + v) * 64 + ln], gz[om][v]);
            SC_SCHED_BARRIER();
          }
        if (ads && xgrad) {
#pragma unroll
          for (int v = 0; v < 16; ++v) ad[v] = SC_LOAD_STREAM(sc_at(ads + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c));
        }
        if (ads) {
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[v] += ad[v];
        }
        if (xgrad) {
#pragma unroll
          for (int v = 0; v < 16; v += 2) {
            float d0, d1;
            PMLP_GELU_GRAD2(xv[v], xv[v + 1], d0, d1);
            acc[v] *= d0;
            acc[v + 1] *= d1;
          }
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) SC_STORE_STREAM(sc_at(gxs + (int64_t)(32 * ci + pmlp_row(v, 0)) * sp, lo_c), acc[v]);
        SC_SCHED_BARRIER();
      }
    }
    // gW += g xin^T over the pixels of the tile (operands transposed through LDS), gb += row sums of g
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      if (xact) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) sc_gelu_pair(xe[t], xe[t + 1]);
      }
      SC_WAVE_SYNC();
#pragma unroll
      for (int t = 0; t < 16; ++t) TB[(2 * t + half) * 33 + n] = xe[t];
      if (ci + 1 < CI) {                                   // the next rows while this tile's products run
#pragma unroll
        for (int t = 0; t < 16; ++t) xe[t] = SC_LOAD_STREAM(sc_at(xs + (int64_t)(32 * (ci + 1) + 2 * t) * sp, lo_b));
        SC_SCHED_BARRIER();
      }
#pragma unroll
      for (int o = 0; o < OMN; ++o) {
        SC_WAVE_SYNC();
#pragma unroll
        for (int om = 0; om < CO; ++om) {
          if (om == om0 + o) {                             // (wave-uniform: exactly one tile matches)
#pragma unroll
            for (int v = 0; v < 16; ++v) TA[pmlp_row(v, half) * 33 + n] = gz[om][v];
          }
        }
        SC_WAVE_SYNC();
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 8) {
#pragma unroll
          for (int t = t0; t < t0 + 8; ++t) {
            const float a = TA[n * 33 + 2 * t + half];
            sc_mfma_32x32x2(aW[o][ci], a, TB[n * 33 + 2 * t + half]);
            if (ci == 0) sB[o] += a;
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
      for (int o = 0; o < OMN; ++o)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
          for (int v = 0; v < 16; ++v) red[(32 * o + pmlp_row(v, half)) * C_IN + 32 * ci + n] += aW[o][ci][v];
      for (int hh = 0; hh < 2; ++hh) {
        if (half == hh) {
#pragma unroll
          for (int o = 0; o < OMN; ++o) {
            red[D::oB + 32 * o + n] += sB[o];
            red[D::oG + 32 * o + n] += sG[o];
          }
        }
        SC_WAVE_SYNC();
      }
    }
    SC_SYNC();
  }
  float* dst = g.partial + (int64_t)SC_BID_X * D::NP;
  for (int i = tid; i < D::NP; i += NT) dst[i] = red[i];
}

// last reduction stage of one launch: rows [row0, row0 + rows) of gw, the same rows of gb / ggate (null = not wanted)
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_plinx_reduce(const float* __restrict__ partial, int n, int np, int o_b, int o_g, int c_in, int row0, float* __restrict__ gw,
               float* __restrict__ gb, float* __restrict__ ggate) {
  const int i = SC_BID_X * 256 + SC_TID;
  if (i >= np) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += partial[(int64_t)k * np + i];
  if (i < o_b) gw[(int64_t)row0 * c_in + i] = s;
  else if (i < o_g) { if (gb) gb[row0 + i - o_b] = s; }
  else if (ggate) ggate[row0 + i - o_g] = s;
}
