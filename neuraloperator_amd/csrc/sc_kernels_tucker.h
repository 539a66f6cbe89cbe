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
  int abl;                 // measurement only (SC_TK_ABL): 1 = no k loops, 2 = no tile stores
  uint32_t inv_rx, inv_ry, inv_my;   // ceil(2^32 / n): i / n for i < 2^16 as a multiply (matrix-core kernels)
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
// Written as inline assembly with the accumulator TIED (D = C): through the builtin the compiler routes the
// accumulators of a multi-tile k loop through one scratch tuple (four v_accvgpr_mov per MFMA, every MFMA dependent on
// the copy before it).  Wait states the compiler cannot see (cdna_hip_programming.md 5.7 item 2): `s_nop 1` ahead of
// the MFMA covers an operand the vector ALU has just written; the FIRST read of any accumulator by compiler code goes
// through sc_mfma_fence (12 states behind the last 8-pass MFMA), the others through sc_mfma_order.
SC_DEVICE void sc_mfma_16x16x4(sc_f32x4& acc, const float a, const float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
SC_DEVICE void sc_mfma_fence(sc_f32x4& acc) { asm volatile("s_nop 11" : "+v"(acc)); }
SC_DEVICE void sc_mfma_order(sc_f32x4& acc) { asm volatile("" : "+v"(acc)); }
#else
struct sc_f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
inline void sc_mfma_fence(sc_f32x4&) {}
inline void sc_mfma_order(sc_f32x4&) {}
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

// acc[t] += NT 16 x 16 tiles of opA(A)(M x K) opB(B)(K x N): A(i, k) = A[i a_si + k a_sk], B(k, j) = B[k b_sk + j b_sj]
// (complex, LDS).  SHARE_A: the tiles (i0, j0 + 16 t) of one block of rows -- the A operand of a k step is read once
// for all of them; otherwise the tiles (i0 + 16 t, j0) of one block of columns share the B operand.  Per k step:
// 1 + NT LDS reads, 3 NT independent MFMAs (NT = 3: 288 matrix-pipe cycles), a handful of vector instructions; one
// tile set-up per NT tiles (a wave that walks single tiles spends 8 vector instructions per MFMA: SQ_INSTS_VALU /
// SQ_INSTS_MFMA of the first version, profiles/r03_fmx_pmc.txt).
// The k extent of every LDS array is padded with ZEROS to a multiple of 4 (tkm_layout), so the loop has no edge
// handling; rows / columns past the end read the last valid one (never stored; whole blocks past the end are wasted
// work the callers avoid through NT).  Conjugations cost nothing inside the loop: with sa, sb = -1 for a conjugated
// operand the three products are
//     P1 = Re A Re B,  P2 = Im A Im B,  P3 = (Re A + sa Im A)(Re B + sb Im B)
// and  Re C = P1 - sa sb P2,  Im C = P3 - P1 - sa sb P2  (tk_result).
// The operands of step s + 1 are requested before the MFMAs of step s are issued -- without a branch around the
// requests (the compiler would wait for every outstanding read at the join): the last step re-reads itself.
// Round 5 (sc_kernels_tkchain.h): GA / GB = that operand lives in GLOBAL memory (a small table every workgroup reads:
// the channel factor matrices, L1 / L2 resident) and is loaded straight into the MFMA operand layout -- same loop, the
// request one k step ahead covers an L1 / L2 hit (a step is 3 NT matrix instructions = 96 NT cycles).
template <bool G>
SC_DEVICE cf32 tk_ld(const cf32* p) {
  if constexpr (G) return *p;
  else return sc_lds_ld64(p);
}
template <int NT, bool SHARE_A, bool CA, bool CB, bool GA = false, bool GB = false>
SC_DEVICE void tk_multi(const cf32* A, const int a_si, const int a_sk, const cf32* B, const int b_sk, const int b_sj,
                        const int i0, const int j0, const int M, const int N, const int K, const int lane,
                        TkAcc (&acc)[NT], const int abl = 0) {
  constexpr bool GSH = SHARE_A ? GA : GB, GMU = SHARE_A ? GB : GA;   // where the shared / the NT other operands live
  const int li = lane & 15, kq = lane >> 4;
  const int ns = (abl & 1) ? 0 : (K + 3) >> 2;
  if (ns == 0) return;
  const cf32* sp;                                            // the shared operand, the NT others
  const cf32* mp[NT];
  int s_step, m_step;
  if (SHARE_A) {
    const int i = i0 + li < M ? i0 + li : M - 1;
    sp = A + i * a_si + kq * a_sk;
    s_step = 4 * a_sk;
    m_step = 4 * b_sk;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = j0 + 16 * t + li < N ? j0 + 16 * t + li : N - 1;
      mp[t] = B + j * b_sj + kq * b_sk;
    }
  } else {
    const int j = j0 + li < N ? j0 + li : N - 1;
    sp = B + j * b_sj + kq * b_sk;
    s_step = 4 * b_sk;
    m_step = 4 * a_sk;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = i0 + 16 * t + li < M ? i0 + 16 * t + li : M - 1;
      mp[t] = A + i * a_si + kq * a_sk;
    }
  }
  sc_f32x4 p[NT][3];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 3; ++u) p[t][u] = acc[t].p[u];
  cf32 sv = tk_ld<GSH>(sp), mv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) mv[t] = tk_ld<GMU>(mp[t]);
  constexpr bool CS = SHARE_A ? CA : CB, CM = SHARE_A ? CB : CA;    // conjugation of the shared / the other operand
  for (int st = 0; st < ns; ++st) {
    const int adv = st + 1 < ns ? 1 : 0;                     // uniform select
    sp += adv * s_step;
    const cf32 sn = tk_ld<GSH>(sp);
    cf32 mn[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      mp[t] += adv * m_step;
      mn[t] = tk_ld<GMU>(mp[t]);
    }
    const float s3 = CS ? sv.x - sv.y : sv.x + sv.y;
    float m3[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) m3[t] = CM ? mv[t].x - mv[t].y : mv[t].x + mv[t].y;
    // the A operand of an MFMA is the row side
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (SHARE_A) sc_mfma_16x16x4(p[t][0], sv.x, mv[t].x);
      else sc_mfma_16x16x4(p[t][0], mv[t].x, sv.x);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (SHARE_A) sc_mfma_16x16x4(p[t][1], sv.y, mv[t].y);
      else sc_mfma_16x16x4(p[t][1], mv[t].y, sv.y);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (SHARE_A) sc_mfma_16x16x4(p[t][2], s3, m3[t]);
      else sc_mfma_16x16x4(p[t][2], m3[t], s3);
    }
    sv = sn;
#pragma unroll
    for (int t = 0; t < NT; ++t) mv[t] = mn[t];
  }
  sc_mfma_fence(p[0][0]);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (t || u) sc_mfma_order(p[t][u]);
      acc[t].p[u] = p[t][u];
    }
}

// element v of the lane's 4 results of a tile; SS = sa sb < 0 (exactly one operand conjugated)
template <bool SS>
SC_DEVICE cf32 tk_result(const TkAcc& t, const int v) {
  const float p1 = t.p[0][v], p2 = SS ? -t.p[1][v] : t.p[1][v], p3 = t.p[2][v];
  return cf_make(p1 - p2, (p3 - p1) - p2);
}

// C[i c_si + j] = tile (row-major destination, LDS or global), bounds-checked
template <bool SS>
SC_DEVICE void tk_store(const TkAcc& t, cf32* C, const int c_si, const int i0, const int j0, const int M, const int N,
                        const int lane, const int abl = 0) {
  const int j = j0 + (lane & 15), ib = i0 + 4 * (lane >> 4);
  if (j < N && !(abl & 2)) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (ib + v < M) C[(ib + v) * c_si + j] = tk_result<SS>(t, v);
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
// every array is padded to multiples of 4 rows and 4 columns (zero filled once: the k loops read the padding).
// Row strides (8-byte units; a ds_read_b64 serves lanes 0-31 / 32-63 in one cycle each when their units differ mod 32;
// a half wave is 16 rows or columns x 2 k values):
//   2 x odd   for arrays read along a row per lane (address = lane_row stride + k): conflict free; also the choice for
//             arrays read both ways (the other pattern is then 2-way);
//   16 mod 32 for arrays only read with k along the rows (address = k stride + lane_column): conflict free.
SC_TK_HD int tkm_ld_rows(const int n4) { return (n4 + 2) | 2; }                       // smallest 2 x odd > n4 ... n4 % 4 == 0
SC_TK_HD int tkm_ld_k(const int n4) { return ((n4 + 15) & ~31) + 16; }                  // smallest 16 (mod 32) >= n4
SC_TK_HD TkmLayout tkm_layout(const int Rx, const int Ry, const int Mx, const int My, const bool bwd) {
  const int rx4 = (Rx + 3) & ~3, ry4 = (Ry + 3) & ~3, mx4 = (Mx + 3) & ~3, my4 = (My + 3) & ~3;
  TkmLayout L;
  L.ldx = bwd ? tkm_ld_k(rx4) : tkm_ld_rows(rx4);    // forward: rows x, k = c; backward (s = U_x^H gT): k = x down the rows
  L.ldy = tkm_ld_rows(ry4);
  L.ldc = tkm_ld_rows(ry4);
  L.ldt = bwd ? tkm_ld_rows(my4) : tkm_ld_k(my4);    // forward: k = c down the rows; backward (gU_x): rows c, k = y
  L.ldg = tkm_ld_rows(my4);
  L.lds = tkm_ld_rows(my4);
  L.o_uy = mx4 * L.ldx;
  L.o_co = L.o_uy + my4 * L.ldy;
  L.o_tmp = L.o_co + rx4 * L.ldc;
  L.o_gt = L.o_tmp + rx4 * L.ldt;
  L.o_s = L.o_gt + (bwd ? mx4 * L.ldg : 0);
  L.total = L.o_s + (bwd ? rx4 * L.lds : 0);
  return L;
}

SC_DEVICE int tkm_div(const int i, const uint32_t inv) { return (int)(((uint64_t)(uint32_t)i * inv) >> 32); }
// a [rows][cols] table (<= 4096 entries) into its padded LDS rows in two phases, so that all global loads of the
// kernel's start are in flight together: fetch into registers, store after the zero fill
SC_DEVICE void tkm_table_fetch(const cf32* __restrict__ src, const int n, const int tid, cf32 (&v)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (tid + 256 * k < n) v[k] = src[tid + 256 * k];
}
SC_DEVICE void tkm_table_store(cf32* dst, const int n, const int cols, const uint32_t inv_cols, const int ld, const int tid,
                               const cf32 (&v)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = tid + 256 * k;
    if (i < n) {
      const int r = tkm_div(i, inv_cols);
      dst[r * ld + (i - r * cols)] = v[k];
    }
  }
}

// PFC: prefetch registers per thread for a core slice (ceil(Rx Ry / 256)); TY: 16-blocks of y a wave multiplies at once
// (>= ceil(My / 16); blocks past the end are wasted work)
template <int PFC, int TY>
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_modes_fwd_mx(TuckerModesArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const TkmLayout L = tkm_layout(g.Rx, g.Ry, g.Mx, g.My, false);
  cf32* ux = lds;
  cf32* uy = lds + L.o_uy;
  cf32* co = lds + L.o_co;
  cf32* tmp = lds + L.o_tmp;
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  const int n_co = g.Rx * g.Ry;
  cf32 pf[PFC];
  auto fetch = [&](const int fg) {
    const cf32* cs = g.core + (int64_t)fg * n_co;
#pragma unroll
    for (int k = 0; k < PFC; ++k)
      if (tid + 256 * k < n_co) pf[k] = cs[tid + 256 * k];
  };
  if ((int)SC_BID_X < g.FG) fetch(SC_BID_X);
  {
    cf32 tx[16], ty[16];
    tkm_table_fetch(g.ux, g.Mx * g.Rx, tid, tx);
    tkm_table_fetch(g.uy, g.My * g.Ry, tid, ty);
    for (int i = tid; i < L.total; i += 256) lds[i] = cf_make(0.f, 0.f);
    SC_SYNC();
    tkm_table_store(ux, g.Mx * g.Rx, g.Rx, g.inv_rx, L.ldx, tid, tx);
    tkm_table_store(uy, g.My * g.Ry, g.Ry, g.inv_ry, L.ldy, tid, ty);
  }
  int oc[PFC];                                     // LDS offsets of this thread's core elements (padded rows)
#pragma unroll
  for (int k = 0; k < PFC; ++k) {
    const int i = tid + 256 * k, r = tkm_div(i, g.inv_ry);
    oc[k] = r * L.ldc + (i - r * g.Ry);
  }
  const int t_c = (g.Rx + 15) >> 4, t_x = (g.Mx + 15) >> 4;
  for (int fg = SC_BID_X; fg < g.FG; fg += g.n_wg) {
    SC_SYNC();                                     // tables (first round) / readers of co and tmp (later rounds)
#pragma unroll
    for (int k = 0; k < PFC; ++k)
      if (tid + 256 * k < n_co) co[oc[k]] = pf[k];
    SC_SYNC();
    if (fg + g.n_wg < g.FG) fetch(fg + g.n_wg);
    // tmp[c][y] = sum_d co[c][d] uy[y][d]: wave w takes the rows c = 16 w .., all TY blocks of y
    if (w < t_c) {
      TkAcc a[TY];
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_zero(a[t]);
      tk_multi<TY, true, false, false>(co, L.ldc, 1, uy, 1, L.ldy, 16 * w, 0, g.Rx, g.My, g.Ry, lane, a, g.abl);
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_store<false>(a[t], tmp, L.ldt, 16 * w, 16 * t, g.Rx, g.My, lane, g.abl);
    }
    SC_SYNC();
    // out[x][y] = sum_c ux[x][c] tmp[c][y]: wave w takes the rows x = 16 w ..
    cf32* dst = g.t + (int64_t)fg * g.Mx * g.My;
    if (w < t_x) {
      TkAcc a[TY];
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_zero(a[t]);
      tk_multi<TY, true, false, false>(ux, L.ldx, 1, tmp, L.ldt, 1, 16 * w, 0, g.Mx, g.My, g.Rx, lane, a, g.abl);
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_store<false>(a[t], dst, g.My, 16 * w, 16 * t, g.Mx, g.My, lane, g.abl);
    }
  }
}

// PFC / PFG: prefetch registers per thread for a core slice / a gT slice (ceil(Rx Ry / 256), ceil(Mx My / 256));
// TC / TY / TD: 16-blocks of c (Rx) / y (My) / d (Ry) a wave multiplies at once (>= the number of blocks; blocks past the
// end are wasted work).  The host picks the smallest instantiation that holds the problem.
template <int PFC, int PFG, int TC, int TY, int TD>
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
  {
    cf32 tx[16], ty[16];
    tkm_table_fetch(g.ux, g.Mx * g.Rx, tid, tx);
    tkm_table_fetch(g.uy, g.My * g.Ry, tid, ty);
    for (int i = tid; i < L.total; i += 256) lds[i] = cf_make(0.f, 0.f);
    SC_SYNC();
    tkm_table_store(ux, g.Mx * g.Rx, g.Rx, g.inv_rx, L.ldx, tid, tx);
    tkm_table_store(uy, g.My * g.Ry, g.Ry, g.inv_ry, L.ldy, tid, ty);
  }
  int oc[PFC], og[PFG];                            // LDS offsets of this thread's core / gT elements (padded rows)
#pragma unroll
  for (int k = 0; k < PFC; ++k) {
    const int i = tid + 256 * k, r = tkm_div(i, g.inv_ry);
    oc[k] = r * L.ldc + (i - r * g.Ry);
  }
#pragma unroll
  for (int k = 0; k < PFG; ++k) {
    const int i = tid + 256 * k, r = tkm_div(i, g.inv_my);
    og[k] = r * L.ldg + (i - r * g.My);
  }
  const int t_c = (g.Rx + 15) >> 4, t_y = (g.My + 15) >> 4, t_x = (g.Mx + 15) >> 4;
  // gradient tiles of this wave: gux rows x = 16 w .. x all TC blocks of c; guy rows y = 16 w .. x all TD blocks of d
  TkAcc aux[TC], auy[TD];
#pragma unroll
  for (int k = 0; k < TC; ++k) tk_zero(aux[k]);
#pragma unroll
  for (int k = 0; k < TD; ++k) tk_zero(auy[k]);
  for (int fg = SC_BID_X; fg < g.FG; fg += g.n_wg) {
    SC_SYNC();
#pragma unroll
    for (int k = 0; k < PFC; ++k)
      if (tid + 256 * k < n_co) co[oc[k]] = pfc[k];
#pragma unroll
    for (int k = 0; k < PFG; ++k)
      if (tid + 256 * k < n_gt) gt[og[k]] = pfg[k];
    SC_SYNC();
    if (fg + g.n_wg < g.FG) fetch(fg + g.n_wg);
    // phase A: s[c][y] = sum_x conj(ux[x][c]) gt[x][y] (wave w: rows c = 16 w ..) and tmp[c][y] = sum_d co[c][d] uy[y][d]
    // (with fewer than four blocks of c the otherwise idle wave 3 takes all of them: 3 x 5 k steps against 16)
    if (w < t_c) {
      TkAcc a[TY];
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_zero(a[t]);
      tk_multi<TY, true, true, false>(ux, 1, L.ldx, gt, L.ldg, 1, 16 * w, 0, g.Rx, g.My, g.Mx, lane, a, g.abl);
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_store<true>(a[t], s, L.lds, 16 * w, 16 * t, g.Rx, g.My, lane, g.abl);
    }
    for (int b = (t_c < 4 ? (w == 3 ? 0 : t_c) : w); b < t_c; b += (t_c < 4 ? 1 : 4)) {
      TkAcc a[TY];
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_zero(a[t]);
      tk_multi<TY, true, false, false>(co, L.ldc, 1, uy, 1, L.ldy, 16 * b, 0, g.Rx, g.My, g.Ry, lane, a, g.abl);
#pragma unroll
      for (int t = 0; t < TY; ++t) tk_store<false>(a[t], tmp, L.ldt, 16 * b, 16 * t, g.Rx, g.My, lane, g.abl);
    }
    SC_SYNC();
    // phase B: gcore[c][d] = sum_y s[c][y] conj(uy[y][d]): block b of c on wave 3 - b (the waves with fewer rows below)
    cf32* gc = g.t + (int64_t)fg * n_co;
    {
      const int b = 3 - w;
      if (b < t_c) {
        TkAcc a[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t) tk_zero(a[t]);
        tk_multi<TD, true, false, true>(s, L.lds, 1, uy, L.ldy, 1, 16 * b, 0, g.Rx, g.Ry, g.My, lane, a, g.abl);
#pragma unroll
        for (int t = 0; t < TD; ++t) tk_store<true>(a[t], gc, g.Ry, 16 * b, 16 * t, g.Rx, g.Ry, lane, g.abl);
      }
    }
    // gux[x][c] += sum_y gt[x][y] conj(tmp[c][y])
    if (w < t_x) tk_multi<TC, true, false, true>(gt, L.ldg, 1, tmp, 1, L.ldt, 16 * w, 0, g.Mx, g.Rx, g.My, lane, aux, g.abl);
    // guy[y][d] += sum_c s[c][y] conj(co[c][d])
    if (w < t_y) tk_multi<TD, true, false, true>(s, 1, L.lds, co, L.ldc, 1, 16 * w, 0, g.My, g.Ry, g.Rx, lane, auy, g.abl);
  }
  // one partial per workgroup: [gux (Mx Rx) | guy (My Ry)] interleaved complex; every entry is owned by one lane
  cf32* dst = reinterpret_cast<cf32*>(g.partial + (int64_t)SC_BID_X * 2 * (g.Mx * g.Rx + g.My * g.Ry));
  if (w < t_x) {
#pragma unroll
    for (int k = 0; k < TC; ++k) tk_store<true>(aux[k], dst, g.Rx, 16 * w, 16 * k, g.Mx, g.Rx, lane, g.abl);
  }
  dst += g.Mx * g.Rx;
  if (w < t_y) {
#pragma unroll
    for (int k = 0; k < TD; ++k) tk_store<true>(auy[k], dst, g.Ry, 16 * w, 16 * k, g.My, g.Ry, lane, g.abl);
  }
}

// sums[i] = sum_k partial[k][i] (fixed order) over the n workgroup partials of npc complex entries -> gux | guy.
// One launch instead of the two-stage k_pmlp_reduce1 + k_tucker_scatter (10.5 + 5.7 us for 512 x 23 KB): a block takes
// 16 columns x 16 row groups, eight 8-byte loads in flight per thread.
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tucker_reduce(const cf32* __restrict__ partial, int n, int npc, int o_uy, cf32* __restrict__ gux, cf32* __restrict__ guy) {
  SC_SHARED cf32 red[16][17];
  const int tid = SC_TID, cx = tid & 15, rg = tid >> 4;
  const int col = SC_BID_X * 16 + cx;
  cf32 acc = cf_make(0.f, 0.f);
  if (col < npc) {
#pragma unroll 8
    for (int k = rg; k < n; k += 16) {
      const cf32 v = partial[(int64_t)k * npc + col];
      acc.x += v.x;
      acc.y += v.y;
    }
  }
  red[rg][cx] = acc;
  SC_SYNC();
  if (rg == 0 && col < npc) {
    cf32 t = red[0][cx];
    for (int r = 1; r < 16; ++r) {
      t.x += red[r][cx].x;
      t.y += red[r][cx].y;
    }
    if (col < o_uy) gux[col] = t;
    else guy[col - o_uy] = t;
  }
}
