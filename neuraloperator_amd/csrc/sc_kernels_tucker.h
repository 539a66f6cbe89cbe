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

// ------------------------------------------------------------------------------------------
// Round 3: the same two launches on the matrix cores.  The VALU kernels above are LDS-instruction bound (one 8-byte
// LDS read per complex multiply-add: 58 / 130 us at ranks (36, 36, 36, 19), kept 64 x 33 -- a few us of arithmetic).
// Here every product of a slice is a set of 16 x 16 tiles of v_mfma_f32_16x16x4_f32 (exact fp32), three real
// products per complex product
//     P1 = Re A Re B,  P2 = Im A Im B,  P3 = (Re A + Im A)(Re B + Im B);   Re C = P1 - P2,  Im C = P3 - P1 - P2
// with the operands read from LDS once per tile and k step (two 8-byte reads per 3 MFMAs = 768 real multiply-adds);
// ragged extents (36, 33, 19) cost whole 16-tiles (48, 48, 32), not 64s.  Waves take the tiles of a phase round-robin;
// the factor gradients accumulate in the MFMA accumulators of fixed (wave, slot) tiles over the slices of a workgroup
// (P1 / P2 / P3 are linear in the products, so they are combined once at the end).  The next slice's core / gT rows
// are in flight (registers) while this one is multiplied.  LDS rows are padded to odd strides.
// ------------------------------------------------------------------------------------------
#ifndef SC_EMU
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
SC_DEVICE void sc_mfma_16x16x4(sc_f32x4& acc, const float a, const float b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}
#else
struct sc_f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
// lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; it owns D[row = 4 (l >> 4) + v][col = l & 15]
// (cdna_hip_programming.md 3)
inline void sc_mfma_16x16x4(sc_f32x4& acc, const float a, const float b) {
  const int w = SC_TID >> 6, l = SC_TID & 63;
  scemu::g_mfma_a[w][l] = a;
  scemu::g_mfma_b[w][l] = b;
  scemu::wave_barrier();
  for (int v = 0; v < 4; ++v) {
    const int row = 4 * (l >> 4) + v, col = l & 15;
    float c = acc[v];
    for (int k = 0; k < 4; ++k) c = fmaf(scemu::g_mfma_a[w][row + 16 * k], scemu::g_mfma_b[w][col + 16 * k], c);
    acc[v] = c;
  }
  scemu::wave_barrier();
}
#endif

struct TkAcc {
  sc_f32x4 p[3];
};
SC_DEVICE void tk_zero(TkAcc& t) {
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) t.p[u][v] = 0.f;
}

// acc += the 16 x 16 tile (i0.., j0..) of opA(A)(M x K) opB(B)(K x N): A(i, k) = A[i a_si + k a_sk], B(k, j) = B[k b_sk + j b_sj]
// (complex, LDS).  Rows / columns past the end read the last valid one (never stored); k past the end contributes zero.
template <bool CA, bool CB>
SC_DEVICE void tk_tile(const cf32* A, const int a_si, const int a_sk, const cf32* B, const int b_sk, const int b_sj,
                       const int i0, const int j0, const int M, const int N, const int K, const int lane, TkAcc& acc) {
  const int li = lane & 15, kq = lane >> 4;
  const int i = i0 + li < M ? i0 + li : M - 1, j = j0 + li < N ? j0 + li : N - 1;
  const cf32* ap = A + i * a_si + kq * a_sk;
  const cf32* bp = B + j * b_sj + kq * b_sk;
  const int kfull = K & ~3;
  const int as4 = 4 * a_sk, bs4 = 4 * b_sk;
#pragma unroll 2
  for (int k0 = 0; k0 < kfull; k0 += 4) {
    const cf32 a = sc_lds_ld64(ap), b = sc_lds_ld64(bp);
    ap += as4;
    bp += bs4;
    const float ai = CA ? -a.y : a.y, bi = CB ? -b.y : b.y;
    sc_mfma_16x16x4(acc.p[0], a.x, b.x);
    sc_mfma_16x16x4(acc.p[1], ai, bi);
    sc_mfma_16x16x4(acc.p[2], a.x + ai, b.x + bi);
  }
  if (kfull < K) {                                           // uniform: the last, partial k step
    const bool kv = kfull + kq < K;
    const int back = kv ? 0 : (kfull + kq - (K - 1));
    cf32 a = sc_lds_ld64(ap - back * a_sk);
    const cf32 b = sc_lds_ld64(bp - back * b_sk);
    if (!kv) a = cf_make(0.f, 0.f);
    const float ai = CA ? -a.y : a.y, bi = CB ? -b.y : b.y;
    sc_mfma_16x16x4(acc.p[0], a.x, b.x);
    sc_mfma_16x16x4(acc.p[1], ai, bi);
    sc_mfma_16x16x4(acc.p[2], a.x + ai, b.x + bi);
  }
}

// element v of the lane's 4 results of a tile
SC_DEVICE cf32 tk_result(const TkAcc& t, const int v) {
  const float p1 = t.p[0][v], p2 = t.p[1][v], p3 = t.p[2][v];
  return cf_make(p1 - p2, (p3 - p1) - p2);
}

// C[i c_si + j] = tile (row-major destination, LDS or global), bounds-checked
SC_DEVICE void tk_store(const TkAcc& t, cf32* C, const int c_si, const int i0, const int j0, const int M, const int N,
                        const int lane) {
  const int j = j0 + (lane & 15), ib = i0 + 4 * (lane >> 4);
  if (j < N) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (ib + v < M) C[(ib + v) * c_si + j] = tk_result(t, v);
  }
}


struct TkmLayout {                  // LDS row strides (odd) and offsets, in complex elements
  int ldx, ldy, ldc, ldt, ldg, lds;
  int o_uy, o_co, o_tmp, o_gt, o_s, total;
};
#ifndef SC_EMU
#define SC_TK_HD __host__ __device__ inline
#else
#define SC_TK_HD inline
#endif
SC_TK_HD TkmLayout tkm_layout(const int Rx, const int Ry, const int Mx, const int My, const bool bwd) {
  TkmLayout L;
  L.ldx = Rx | 1; L.ldy = Ry | 1; L.ldc = Ry | 1; L.ldt = My | 1; L.ldg = My | 1; L.lds = My | 1;
  L.o_uy = Mx * L.ldx;
  L.o_co = L.o_uy + My * L.ldy;
  L.o_tmp = L.o_co + Rx * L.ldc;
  L.o_gt = L.o_tmp + Rx * L.ldt;
  L.o_s = L.o_gt + (bwd ? Mx * L.ldg : 0);
  L.total = L.o_s + (bwd ? Rx * L.lds : 0);
  return L;
}

SC_DEVICE void tkm_load_table(const cf32* __restrict__ src, cf32* dst, const int rows, const int cols, const int ld,
                              const int tid) {
  for (int i = tid; i < rows * cols; i += 256) {
    const int r = i / cols, c = i - r * cols;
    dst[r * ld + c] = src[i];
  }
}

template <int PFC>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_modes_fwd_mx(TuckerModesArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const TkmLayout L = tkm_layout(g.Rx, g.Ry, g.Mx, g.My, false);
  cf32* ux = lds;
  cf32* uy = lds + L.o_uy;
  cf32* co = lds + L.o_co;
  cf32* tmp = lds + L.o_tmp;
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  tkm_load_table(g.ux, ux, g.Mx, g.Rx, L.ldx, tid);
  tkm_load_table(g.uy, uy, g.My, g.Ry, L.ldy, tid);
  const int n_co = g.Rx * g.Ry;
  cf32 pf[PFC];
  auto fetch = [&](const int fg) {
    const cf32* cs = g.core + (int64_t)fg * n_co;
#pragma unroll
    for (int k = 0; k < PFC; ++k)
      if (tid + 256 * k < n_co) pf[k] = cs[tid + 256 * k];
  };
  if ((int)SC_BID_X < g.FG) fetch(SC_BID_X);
  const int ti_c = (g.Rx + 15) >> 4, tj_y = (g.My + 15) >> 4, ti_x = (g.Mx + 15) >> 4;
  for (int fg = SC_BID_X; fg < g.FG; fg += g.n_wg) {
    SC_SYNC();                                     // tables (first round) / readers of co and tmp (later rounds)
#pragma unroll
    for (int k = 0; k < PFC; ++k) {
      const int i = tid + 256 * k;
      if (i < n_co) co[(i / g.Ry) * L.ldc + (i % g.Ry)] = pf[k];
    }
    SC_SYNC();
    if (fg + g.n_wg < g.FG) fetch(fg + g.n_wg);
    // tmp[c][y] = sum_d co[c][d] uy[y][d]
    for (int t = w; t < ti_c * tj_y; t += 4) {
      const int i0 = (t / tj_y) * 16, j0 = (t % tj_y) * 16;
      TkAcc a;
      tk_zero(a);
      tk_tile<false, false>(co, L.ldc, 1, uy, 1, L.ldy, i0, j0, g.Rx, g.My, g.Ry, lane, a);
      tk_store(a, tmp, L.ldt, i0, j0, g.Rx, g.My, lane);
    }
    SC_SYNC();
    // out[x][y] = sum_c ux[x][c] tmp[c][y]
    cf32* dst = g.t + (int64_t)fg * g.Mx * g.My;
    for (int t = w; t < ti_x * tj_y; t += 4) {
      const int i0 = (t / tj_y) * 16, j0 = (t % tj_y) * 16;
      TkAcc a;
      tk_zero(a);
      tk_tile<false, false>(ux, L.ldx, 1, tmp, L.ldt, 1, i0, j0, g.Mx, g.My, g.Rx, lane, a);
      tk_store(a, dst, g.My, i0, j0, g.Mx, g.My, lane);
    }
  }
}

// PFC / PFG: prefetch registers per thread for a core slice / a gT slice (ceil(Rx Ry / 256), ceil(Mx My / 256));
// SX / SY: gradient tiles per wave (ceil(tiles / 4)).  The host picks the smallest instantiation that holds the problem.
template <int PFC, int PFG, int SX, int SY>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_modes_bwd_mx(TuckerModesArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const TkmLayout L = tkm_layout(g.Rx, g.Ry, g.Mx, g.My, true);
  cf32* ux = lds;
  cf32* uy = lds + L.o_uy;
  cf32* co = lds + L.o_co;
  cf32* tmp = lds + L.o_tmp;
  cf32* gt = lds + L.o_gt;
  cf32* s = lds + L.o_s;
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  tkm_load_table(g.ux, ux, g.Mx, g.Rx, L.ldx, tid);
  tkm_load_table(g.uy, uy, g.My, g.Ry, L.ldy, tid);
  const int n_co = g.Rx * g.Ry, n_gt = g.Mx * g.My;
  cf32 pfc[PFC], pfg[PFG];
  auto fetch = [&](const int fg) {
    const cf32* cs = g.core + (int64_t)fg * n_co;
    const cf32* gs = g.gt + (int64_t)fg * n_gt;
#pragma unroll
    for (int k = 0; k < PFC; ++k)
      if (tid + 256 * k < n_co) pfc[k] = cs[tid + 256 * k];
#pragma unroll
    for (int k = 0; k < PFG; ++k)
      if (tid + 256 * k < n_gt) pfg[k] = gs[tid + 256 * k];
  };
  if ((int)SC_BID_X < g.FG) fetch(SC_BID_X);
  const int t_c = (g.Rx + 15) >> 4, t_y = (g.My + 15) >> 4, t_x = (g.Mx + 15) >> 4, t_d = (g.Ry + 15) >> 4;
  // gradient tiles of this wave: gux tile t = w + 4 slot (x block t / t_c, c block t % t_c), guy tile t = w + 4 slot
  // (y block t / t_d, d block t % t_d); the tiles of gcore (w2) and tmp (w3) start at rotated waves so that the waves
  // with one tile fewer of the one product take one more of the other
  const int w2 = (w + 2) & 3, w3 = (w + 3) & 3;
  TkAcc aux[SX], auy[SY];
#pragma unroll
  for (int k = 0; k < SX; ++k) tk_zero(aux[k]);
#pragma unroll
  for (int k = 0; k < SY; ++k) tk_zero(auy[k]);
  for (int fg = SC_BID_X; fg < g.FG; fg += g.n_wg) {
    SC_SYNC();
#pragma unroll
    for (int k = 0; k < PFC; ++k) {
      const int i = tid + 256 * k;
      if (i < n_co) co[(i / g.Ry) * L.ldc + (i % g.Ry)] = pfc[k];
    }
#pragma unroll
    for (int k = 0; k < PFG; ++k) {
      const int i = tid + 256 * k;
      if (i < n_gt) gt[(i / g.My) * L.ldg + (i % g.My)] = pfg[k];
    }
    SC_SYNC();
    if (fg + g.n_wg < g.FG) fetch(fg + g.n_wg);
    // phase A: s[c][y] = sum_x conj(ux[x][c]) gt[x][y]  and  tmp[c][y] = sum_d co[c][d] uy[y][d]
    for (int t = w; t < t_c * t_y; t += 4) {
      const int i0 = (t / t_y) * 16, j0 = (t % t_y) * 16;
      TkAcc a;
      tk_zero(a);
      tk_tile<true, false>(ux, 1, L.ldx, gt, L.ldg, 1, i0, j0, g.Rx, g.My, g.Mx, lane, a);
      tk_store(a, s, L.lds, i0, j0, g.Rx, g.My, lane);
    }
    for (int t = w3; t < t_c * t_y; t += 4) {
      const int i0 = (t / t_y) * 16, j0 = (t % t_y) * 16;
      TkAcc a;
      tk_zero(a);
      tk_tile<false, false>(co, L.ldc, 1, uy, 1, L.ldy, i0, j0, g.Rx, g.My, g.Ry, lane, a);
      tk_store(a, tmp, L.ldt, i0, j0, g.Rx, g.My, lane);
    }
    SC_SYNC();
    // phase B: gcore[c][d] = sum_y s[c][y] conj(uy[y][d])
    cf32* gc = g.t + (int64_t)fg * n_co;
    for (int t = w2; t < t_c * t_d; t += 4) {
      const int i0 = (t / t_d) * 16, j0 = (t % t_d) * 16;
      TkAcc a;
      tk_zero(a);
      tk_tile<false, true>(s, L.lds, 1, uy, L.ldy, 1, i0, j0, g.Rx, g.Ry, g.My, lane, a);
      tk_store(a, gc, g.Ry, i0, j0, g.Rx, g.Ry, lane);
    }
    // gux[x][c] += sum_y gt[x][y] conj(tmp[c][y])
#pragma unroll
    for (int k = 0; k < SX; ++k) {
      const int t = w + 4 * k;
      if (t < t_x * t_c)
        tk_tile<false, true>(gt, L.ldg, 1, tmp, 1, L.ldt, (t / t_c) * 16, (t % t_c) * 16, g.Mx, g.Rx, g.My, lane, aux[k]);
    }
    // guy[y][d] += sum_c s[c][y] conj(co[c][d])
#pragma unroll
    for (int k = 0; k < SY; ++k) {
      const int t = w + 4 * k;
      if (t < t_y * t_d)
        tk_tile<false, true>(s, 1, L.lds, co, L.ldc, 1, (t / t_d) * 16, (t % t_d) * 16, g.My, g.Ry, g.Rx, lane, auy[k]);
    }
  }
  // one partial per workgroup: [gux (Mx Rx) | guy (My Ry)] interleaved complex; every entry is owned by one lane
  cf32* dst = reinterpret_cast<cf32*>(g.partial + (int64_t)SC_BID_X * 2 * (g.Mx * g.Rx + g.My * g.Ry));
#pragma unroll
  for (int k = 0; k < SX; ++k) {
    const int t = w + 4 * k;
    if (t < t_x * t_c) tk_store(aux[k], dst, g.Rx, (t / t_c) * 16, (t % t_c) * 16, g.Mx, g.Rx, lane);
  }
  dst += g.Mx * g.Rx;
#pragma unroll
  for (int k = 0; k < SY; ++k) {
    const int t = w + 4 * k;
    if (t < t_y * t_d) tk_store(auy[k], dst, g.Ry, (t / t_d) * 16, (t % t_d) * 16, g.My, g.Ry, lane);
  }
}
