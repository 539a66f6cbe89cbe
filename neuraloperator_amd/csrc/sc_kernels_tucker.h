// sc_kernels_tucker.h -- the batch-independent part of the 2-D Tucker (TFNO) contraction in one launch each way:
//
//     T[f, g, x, y] = sum_{c, d} core[f, g, c, d] U_x[x, c] U_y[y, d]
//
// (_contract_tucker, spectral_convolution.py:76-103, pairwise order of SURVEY.md 8 a6: the mode factors are absorbed
// into the core first; the result is the per-mode 36 x 36 block the activations are contracted with).  As a chain of
// general mode-GEMM launches this step and its gradients were ~8 launches of 50-80 us each on tensors of 7-22 MB
// (profiles/r02_tfno_kernel_stats_msum2.txt) plus the transposing copies autograd needed between them.  Here a
// workgroup takes one (f, g) slice at a time: core slice (R_x x R_y), both factor matrices and the intermediate
// tmp[c, y] = sum_d core[c, d] U_y[y, d] live in LDS, T's slice leaves as one contiguous M_x M_y run.
// Backward: per slice s = U_x^H gT, gcore = s conj(U_y), and the factor gradients gU_x += gT tmp^H, gU_y += s^T conj(core)
// accumulate in registers over the slices of a workgroup (every thread owns fixed entries), one partial per workgroup,
// fixed-order reduction (k_pmlp_reduce1 + k_tucker_scatter).
#pragma once
#include "sc_kernels_pmlp.h"

#define SC_TK_TILE 16               // register tile of the big products: 4 interleaved groups of <= 16 (sizes <= 64)

struct TuckerModesArgs {
  const cf32* core;        // [FG][Rx][Ry]
  const cf32* ux;          // [Mx][Rx]
  const cf32* uy;          // [My][Ry]
  const cf32* gt;          // backward: [FG][Mx][My]
  cf32* t;                 // forward: [FG][Mx][My]; backward: gcore [FG][Rx][Ry]
  float* partial;          // backward: [n_wg][2 (Mx Rx + My Ry)]
  int FG, Rx, Ry, Mx, My, n_wg;
};

SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_modes_fwd(TuckerModesArgs g) {
  SC_DYN_SHARED(cf32, lds);
  cf32* ux = lds;                                  // [Mx][Rx]
  cf32* uy = ux + g.Mx * g.Rx;                     // [My][Ry]
  cf32* co = uy + g.My * g.Ry;                     // [Rx][Ry]
  cf32* tmp = co + g.Rx * g.Ry;                    // [Rx][My]
  const int tid = SC_TID;
  for (int i = tid; i < g.Mx * g.Rx; i += 256) ux[i] = g.ux[i];
  for (int i = tid; i < g.My * g.Ry; i += 256) uy[i] = g.uy[i];
  for (int fg = SC_BID_X; fg < g.FG; fg += g.n_wg) {
    SC_SYNC();                                     // tables (first round) / readers of co and tmp (later rounds)
    const cf32* cs = g.core + (int64_t)fg * g.Rx * g.Ry;
    for (int i = tid; i < g.Rx * g.Ry; i += 256) co[i] = cs[i];
    SC_SYNC();
    for (int i = tid; i < g.Rx * g.My; i += 256) {
      const int c = i / g.My, y = i - c * g.My;
      cf32 acc = cf_make(0.f, 0.f);
      for (int d = 0; d < g.Ry; ++d) cf_mac(acc, co[c * g.Ry + d], uy[y * g.Ry + d]);
      tmp[i] = acc;
    }
    SC_SYNC();
    // out[x][y] = sum_c ux[x][c] tmp[c][y]: thread (x, yg) holds the row's outputs y = yg, yg + 4, ... in registers:
    // one ux read + <= 16 tmp reads (broadcast across the x's of a wave) per 16 multiply-adds
    cf32* dst = g.t + (int64_t)fg * g.Mx * g.My;
    {
      const int x = tid >> 2, yg = tid & 3;
      if (x < g.Mx) {
        cf32 acc[SC_TK_TILE];
#pragma unroll
        for (int k = 0; k < SC_TK_TILE; ++k) acc[k] = cf_make(0.f, 0.f);
        for (int c = 0; c < g.Rx; ++c) {
          const cf32 a = ux[x * g.Rx + c];
#pragma unroll
          for (int k = 0; k < SC_TK_TILE; ++k) {
            if (4 * k >= g.My) break;                                     // uniform; lanes past the edge re-read the last column
            cf_mac(acc[k], a, tmp[c * g.My + (yg + 4 * k < g.My ? yg + 4 * k : g.My - 1)]);
          }
        }
#pragma unroll
        for (int k = 0; k < SC_TK_TILE; ++k)
          if (yg + 4 * k < g.My) dst[x * g.My + yg + 4 * k] = acc[k];
      }
    }
  }
}

// entries of gU_y a thread owns: i = tid + 256 k; gU_x: thread (x, cg) owns c = cg, cg + 4, ...
#define SC_TK_UY_PER_THREAD 4       // My Ry <= 1024

SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_modes_bwd(TuckerModesArgs g) {
  SC_DYN_SHARED(cf32, lds);
  cf32* ux = lds;                                  // [Mx][Rx]
  cf32* uy = ux + g.Mx * g.Rx;                     // [My][Ry]
  cf32* co = uy + g.My * g.Ry;                     // [Rx][Ry]
  cf32* tmp = co + g.Rx * g.Ry;                    // [Rx][My]
  cf32* gt = tmp + g.Rx * g.My;                    // [Mx][My]
  cf32* s = gt + g.Mx * g.My;                      // [Rx][My]
  const int tid = SC_TID;
  for (int i = tid; i < g.Mx * g.Rx; i += 256) ux[i] = g.ux[i];
  for (int i = tid; i < g.My * g.Ry; i += 256) uy[i] = g.uy[i];
  cf32 aux[SC_TK_TILE], auy[SC_TK_UY_PER_THREAD];
#pragma unroll
  for (int k = 0; k < SC_TK_TILE; ++k) aux[k] = cf_make(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < SC_TK_UY_PER_THREAD; ++k) auy[k] = cf_make(0.f, 0.f);
  for (int fg = SC_BID_X; fg < g.FG; fg += g.n_wg) {
    SC_SYNC();
    const cf32* cs = g.core + (int64_t)fg * g.Rx * g.Ry;
    const cf32* gs = g.gt + (int64_t)fg * g.Mx * g.My;
    for (int i = tid; i < g.Rx * g.Ry; i += 256) co[i] = cs[i];
    for (int i = tid; i < g.Mx * g.My; i += 256) gt[i] = gs[i];
    SC_SYNC();
    for (int i = tid; i < g.Rx * g.My; i += 256) {
      const int c = i / g.My, y = i - c * g.My;
      cf32 a = cf_make(0.f, 0.f);
      for (int d = 0; d < g.Ry; ++d) cf_mac(a, co[c * g.Ry + d], uy[y * g.Ry + d]);          // tmp[c][y]
      tmp[i] = a;
    }
    {                                                           // s[c][y] = sum_x conj(ux[x][c]) gt[x][y]: thread (c, yg) of Rx x ng,
      const int ng = 256 / g.Rx;                                //   ng = 256 / Rx >= 4 column groups so that all four waves work
      const int c = tid / ng, yg = tid - c * ng;
      if (c < g.Rx) {
        cf32 acc[SC_TK_TILE];
#pragma unroll
        for (int k = 0; k < SC_TK_TILE; ++k) acc[k] = cf_make(0.f, 0.f);
        for (int x = 0; x < g.Mx; ++x) {
          const cf32 a = ux[x * g.Rx + c];
#pragma unroll
          for (int k = 0; k < SC_TK_TILE; ++k) {
            if (ng * k >= g.My) break;
            const int y = yg + ng * k;
            cf_mac_conj_a(acc[k], a, gt[x * g.My + (y < g.My ? y : g.My - 1)]);
          }
        }
#pragma unroll
        for (int k = 0; k < SC_TK_TILE; ++k)
          if (yg + ng * k < g.My) s[c * g.My + yg + ng * k] = acc[k];
      }
    }
    SC_SYNC();
    cf32* gc = g.t + (int64_t)fg * g.Rx * g.Ry;
    for (int i = tid; i < g.Rx * g.Ry; i += 256) {            // gcore[c][d] = sum_y s[c][y] conj(uy[y][d])
      const int c = i / g.Ry, d = i - c * g.Ry;
      cf32 acc = cf_make(0.f, 0.f);
      for (int y = 0; y < g.My; ++y) cf_mac_conj_a(acc, uy[y * g.Ry + d], s[c * g.My + y]);
      gc[i] = acc;
    }
    {                                                           // gux[x][c] += sum_y gt[x][y] conj(tmp[c][y]), c = cg + 4 k
      const int x = tid >> 2, cg = tid & 3;
      if (x < g.Mx) {
        for (int y = 0; y < g.My; ++y) {
          const cf32 b = gt[x * g.My + y];
#pragma unroll
          for (int k = 0; k < SC_TK_TILE; ++k) {
            if (4 * k >= g.Rx) break;
            cf_mac_conj_a(aux[k], tmp[(cg + 4 * k < g.Rx ? cg + 4 * k : g.Rx - 1) * g.My + y], b);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < SC_TK_UY_PER_THREAD; ++k) {           // guy[y][d] += sum_c s[c][y] conj(core[c][d])
      const int i = tid + 256 * k;
      if (i < g.My * g.Ry) {
        const int y = i / g.Ry, d = i - y * g.Ry;
        for (int c = 0; c < g.Rx; ++c) cf_mac_conj_a(auy[k], co[c * g.Ry + d], s[c * g.My + y]);
      }
    }
  }
  float* dst = g.partial + (int64_t)SC_BID_X * 2 * (g.Mx * g.Rx + g.My * g.Ry);
  {
    const int x = tid >> 2, cg = tid & 3;
#pragma unroll
    for (int k = 0; k < SC_TK_TILE; ++k)
      if (x < g.Mx && cg + 4 * k < g.Rx) {
        const int i = x * g.Rx + cg + 4 * k;
        dst[2 * i] = aux[k].x;
        dst[2 * i + 1] = aux[k].y;
      }
  }
  dst += 2 * g.Mx * g.Rx;
#pragma unroll
  for (int k = 0; k < SC_TK_UY_PER_THREAD; ++k) {
    const int i = tid + 256 * k;
    if (i < g.My * g.Ry) {
      dst[2 * i] = auy[k].x;
      dst[2 * i + 1] = auy[k].y;
    }
  }
}

// sums[i] = sum_y stage[y][i] (y ascending) -> gux | guy (interleaved complex)
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_scatter(const float* __restrict__ stage, int n, int np, int o_uy, float* __restrict__ gux, float* __restrict__ guy) {
  const int i = SC_BID_X * 256 + SC_TID;
  if (i >= np) return;
  float acc = 0.f;
  for (int k = 0; k < n; ++k) acc += stage[(int64_t)k * np + i];
  if (i < o_uy) gux[i] = acc;
  else guy[i - o_uy] = acc;
}
