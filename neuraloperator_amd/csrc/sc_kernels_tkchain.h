// sc_kernels_tkchain.h -- the activation side of the factorized Tucker contraction as ONE launch each way (round 5).
//
//   forward    z[b, f, m] = sum_i xhat[b, i, m] u_in[i, f];   t[b, g, m] = sum_f z[b, f, m] T3[f, g, m];
//              yhat[b, o, m] = sum_g t[b, g, m] u_out[o, g]
//   backward   gt = gy conj(u_out);  gu_out = sum gy conj(t);  gz = gt T3^H;  gT3 = sum_b conj(z) gt;
//              gu_in = sum conj(xhat) gz;  gxhat = gz u_in^H
// (_contract_tucker, neuralop/layers/spectral_convolution.py:76-103: the pairwise order of the einsum
// 'abcd,fghi,bf,eg,ch,di->aecd' with the mode factors absorbed into T3, and the six products of its autograd).
//
// Rounds 3-4 ran these nine products as nine launches of 24-62 us at BASELINE configs[2] (B = 32, 64 channels, ranks 36,
// 2112 kept modes: ~12 GFLOP of chain arithmetic in ~320 us, every one of them skeleton / latency bound, z, t, gt and gz
// crossing HBM between them -- profiles/r04_tfno_kernel_stats.txt).  Here a workgroup owns a TILE OF FOUR MODES for the
// whole batch and walks the chain out of LDS:
//   * wave w owns mode w of the tile: its planes X_w[b][i], z_w[b][f], t_w[b][g], T3_w are wave-private, so the three
//     products of the forward chain need no workgroup barrier between them (tk_multi: 16 x 16 x 4 exact-fp32 MFMA
//     tiles, three real products per complex product, sc_kernels_tucker.h);
//   * the spectrum stays MODE-MAJOR in HBM (xhat[b][i][m], the layout every other kernel of the engine reads): a tile is
//     staged with 16-byte loads (two modes of one (b, i) pair; the four workgroups that share a 128-byte line are
//     neighbours on ONE XCD, so the line is fetched from HBM once) and the results leave as 32-byte pieces, four modes
//     of one (b, f) pair, assembled from the four waves' planes;
//   * the per-mode operand T3 comes from a MODE-MAJOR copy T3m[m][f][g] (k_tkc_transpose: a tile's four 36 x 36 blocks
//     are one contiguous 41 KB run), the gradient leaves as gT3m[m][f][g] and is transposed back;
//   * the channel factor matrices are read from global memory straight into the MFMA operand layout (18 KB each, L1 /
//     L2 resident: tk_multi<.., GB = true>) -- in LDS they would cost 36 KB of the 147 KB a tile needs;
//   * the two factor gradients (sums over batch AND modes) are split by OUTPUT tile: wave w owns rows 16 w .. 16 w + 15
//     of gu_out / gu_in and accumulates them in MFMA registers over the four modes of every tile the (persistent)
//     workgroup walks; one partial per workgroup, fixed-order reduction (k_tkc_reduce): run-to-run identical.
// Limits (sc_tucker_chain_fused_supported; anything else takes the nine launches): batch <= 32 and a multiple of 4,
// channels <= 64, ranks <= 48, all multiples of 4, the number of modes a multiple of 4, 16-byte aligned tensors.
#pragma once
#include "sc_kernels_tucker.h"

struct TkcArgs {
  // forward: xhat, u_in, t3m, u_out -> z, t, yhat;  backward: + z, t, gy -> gxhat, gt3m, partial
  const cf32* xhat;      // [B][Ci][M]
  const cf32* u_in;      // [Ci][R1]
  const cf32* t3m;       // [M][R1][R2]
  const cf32* u_out;     // [Co][R2]
  const cf32* zin;       // backward: [B][R1][M]
  const cf32* tin;       // backward: [B][R2][M]
  const cf32* gy;        // backward: [B][Co][M]
  cf32* z;               // forward out
  cf32* t;               // forward out
  cf32* yhat;            // forward out [B][Co][M]
  cf32* gxhat;           // backward out [B][Ci][M] (may be null)
  cf32* gt3m;            // backward out [M][R1][R2]
  cf32* partial;         // backward out [n_wg][Co R2 + Ci R1]
  int B, Ci, Co, R1, R2;
  int64_t M;
  int n_tiles, n_wg;
  uint32_t inv_ci, inv_co, inv_r1, inv_r2, inv_r12;   // ceil(2^32 / n)
  int abl;                         // measurement only (SC_TK_ABL): 1 = no k loops, 2 = no tile stores, 4 = no staging loads / emits
};

struct TkcLayout {       // complex elements
  int ldi, ldo, ld1, ld2;          // row strides of planes whose rows are b: [b][i], [b][o], [b][f], [b][g]
  int ldt;                         // T3 plane row stride (forward: [g][f]; backward: [f][g])
  int PA, PB, PT;                  // plane strides: region A (X / t, gy / T3 / X / gxhat), region B (z, ...), T3
  int oA, oB, oC, total;
};
SC_TK_HD int tkc_max(const int a, const int b) { return a > b ? a : b; }
SC_TK_HD TkcLayout tkc_layout(const int B, const int Ci, const int Co, const int R1, const int R2, const bool bwd) {
  TkcLayout L;
  L.ldi = tkm_ld_rows(Ci);
  L.ldo = tkm_ld_rows(Co);
  L.ld1 = tkm_ld_rows(R1);
  L.ld2 = tkm_ld_rows(R2);
  if (!bwd) {
    // A: X planes [b][ldi], later t planes [b][ld2];  B: z planes [b][ld1] then T3 planes [g][ldt = ld1];
    // yhat planes [b][ldo] go over B + T3 once z and T3 are dead
    L.ldt = L.ld1;
    L.PA = B * tkc_max(L.ldi, L.ld2);
    L.PB = B * L.ld1;
    L.PT = R2 * L.ldt;
    L.oA = 0;
    L.oB = 4 * L.PA;
    L.oC = L.oB + 4 * L.PB;                                   // T3 planes
    L.total = L.oB + tkc_max(4 * L.PB + 4 * L.PT, 4 * B * L.ldo);
  } else {
    // A: gy planes [b][ldo] -> T3 planes [f][ldt = ld2] -> X planes [b][ldi] -> gxhat planes [b][ldi];
    // B: t planes [b][ld2] -> z planes [b][ld1] -> gz planes [b][ld1];  C: gt planes [b][ld2]
    L.ldt = L.ld2;
    L.PT = R1 * L.ldt;
    L.PA = tkc_max(B * tkc_max(L.ldi, L.ldo), L.PT);
    L.PB = B * tkc_max(L.ld1, L.ld2);
    L.oA = 0;
    L.oB = 4 * L.PA;
    L.oC = L.oB + 4 * L.PB;
    L.total = L.oC + 4 * B * L.ld2;
  }
  return L;
}

#ifndef SC_EMU
SC_DEVICE sc_f4 tkc_ld16(const cf32* p) { return *reinterpret_cast<const sc_f4*>(p); }
SC_DEVICE void tkc_st16(cf32* p, const cf32 a, const cf32 b) {
  sc_f4 v;
  v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
  *reinterpret_cast<sc_f4*>(p) = v;
}
#else
inline sc_f4 tkc_ld16(const cf32* p) {
  sc_f4 v;
  std::memcpy(&v, p, 16);
  return v;
}
inline void tkc_st16(cf32* p, const cf32 a, const cf32 b) {
  p[0] = a;
  p[1] = b;
}
#endif

// tile a persistent workgroup takes in round r: neighbouring tiles (which share 128-byte lines of the mode-major
// tensors) go to workgroups of ONE XCD (block b runs on XCD b % 8), so a line is fetched from HBM once
SC_DEVICE int tkc_tile(const int round, const int n_wg) {
  int g = SC_BID_X;
  if ((n_wg & 7) == 0) g = (g & 7) * (n_wg >> 3) + (g >> 3);
  return round * n_wg + g;
}

// stage the four mode planes of a [rows][cols][M] tensor tile: thread idx -> (pair p = idx >> 1, half h): one 16-byte
// load = modes m0 + 2 h, + 1 of pair p = (row, col); fetch and store are separate so that every load of a phase is in
// flight before the first LDS write
template <int PF>
SC_DEVICE void tkc_fetch(const cf32* __restrict__ src, const int n_pairs, const int64_t M, const int64_t m0, const int tid,
                         sc_f4 (&v)[PF]) {
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int idx = tid + 256 * k;
    if (idx < 2 * n_pairs) v[k] = tkc_ld16(src + (int64_t)(idx >> 1) * M + m0 + 2 * (idx & 1));
  }
}
template <int PF>
SC_DEVICE void tkc_plant(cf32* planes, const int plane_stride, const int ld, const int n_pairs, const int cols,
                         const uint32_t inv_cols, const int tid, const sc_f4 (&v)[PF]) {
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int idx = tid + 256 * k;
    if (idx < 2 * n_pairs) {
      const int p = idx >> 1, h = idx & 1;
      const int r = tkm_div(p, inv_cols), c = p - r * cols;
      cf32* d = planes + (2 * h) * plane_stride + r * ld + c;
      d[0] = cf_make(v[k].x, v[k].y);
      d[plane_stride] = cf_make(v[k].z, v[k].w);
    }
  }
}
// the four T3 blocks of a tile: one contiguous run of 4 R1 R2 values, 16 bytes = (f, g), (f, g + 1) [R2 even];
// TRANS: plane layout [g][f] (forward: k = f runs along a row), else [f][g]
template <int PF, bool TRANS>
SC_DEVICE void tkc_fetch_t3(const cf32* __restrict__ src, const int n16, const int tid, sc_f4 (&v)[PF]) {
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int idx = tid + 256 * k;
    if (idx < n16) v[k] = tkc_ld16(src + 2 * idx);
  }
}
template <int PF, bool TRANS>
SC_DEVICE void tkc_plant_t3(cf32* planes, const int PT, const int ldt, const int n16, const int R12, const int R2,
                            const uint32_t inv_r12, const uint32_t inv_r2, const int tid, const sc_f4 (&v)[PF]) {
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    const int idx = tid + 256 * k;
    if (idx < n16) {
      const int e = 2 * idx;
      const int m = tkm_div(e, inv_r12), r = e - m * R12;
      const int f = tkm_div(r, inv_r2), g = r - f * R2;
      cf32* d = planes + m * PT + (TRANS ? g * ldt + f : f * ldt + g);
      d[0] = cf_make(v[k].x, v[k].y);
      d[TRANS ? ldt : 1] = cf_make(v[k].z, v[k].w);
    }
  }
}
// four planes [rows][ld] -> global [rows][cols][M] at modes m0 .. m0 + 3: 32 contiguous bytes per (row, col) pair
SC_DEVICE void tkc_emit(const cf32* planes, const int plane_stride, const int ld, const int n_pairs, const int cols,
                        const uint32_t inv_cols, cf32* __restrict__ dst, const int64_t M, const int64_t m0, const int tid) {
  for (int p = tid; p < n_pairs; p += 256) {
    const int r = tkm_div(p, inv_cols), c = p - r * cols;
    const cf32* s = planes + r * ld + c;
    const cf32 v0 = s[0], v1 = s[plane_stride], v2 = s[2 * plane_stride], v3 = s[3 * plane_stride];
    cf32* d = dst + (int64_t)p * M + m0;
    tkc_st16(d, v0, v1);
    tkc_st16(d + 2, v2, v3);
  }
}

// ------------------------------------------------------------------------------------------
// forward: PFX = ceil(2 B Ci / 256), PFT = ceil(2 R1 R2 / 256) prefetch registers (16 bytes each) per thread
// ------------------------------------------------------------------------------------------
template <int PFX, int PFT>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 1)
k_tkc_fwd(TkcArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const TkcLayout L = tkc_layout(g.B, g.Ci, g.Co, g.R1, g.R2, false);
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  cf32* XA = lds + L.oA;
  cf32* ZB = lds + L.oB;
  cf32* T3 = lds + L.oC;
  cf32* YP = lds + L.oB;                            // yhat planes [b][ldo]: over z + T3 once both are dead
  const int PY = g.B * L.ldo;
  const int R12 = g.R1 * g.R2, n16 = 2 * R12;        // 16-byte pieces of a tile's four T3 blocks
  const int rt_n = (g.B + 15) >> 4;
  for (int round = 0;; ++round) {
    const int tile = tkc_tile(round, g.n_wg);
    if (tile >= g.n_tiles) break;
    const int64_t m0 = (int64_t)tile * 4;
    {
      // (an opaque copy of the thread id per phase: derived from the plain one, the ~60 address terms of the phases are
      // loop invariant, get hoisted out of the tile loop and parked in AGPRs / scratch)
      const int tid = sc_opaque(SC_TID);
      sc_f4 vx[PFX], vt[PFT];
      tkc_fetch<PFX>(g.xhat, g.B * g.Ci, g.M, m0, tid, vx);
      tkc_fetch_t3<PFT, true>(g.t3m + m0 * R12, n16, tid, vt);
      tkc_plant<PFX>(XA, L.PA, L.ldi, g.B * g.Ci, g.Ci, g.inv_ci, tid, vx);
      tkc_plant_t3<PFT, true>(T3, L.PT, L.ldt, n16, R12, g.R2, g.inv_r12, g.inv_r2, tid, vt);
    }
    SC_SYNC();
    cf32* Xw = XA + w * L.PA;
    cf32* Zw = ZB + w * L.PB;
    cf32* Tw = XA + w * L.PA;                         // t plane [b][ld2] over this wave's X plane
    // ---- z_w = X_w u_in
#pragma unroll 1
    for (int rt = 0; rt < rt_n; ++rt) {
      TkAcc a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_zero(a[k]);
      tk_multi<3, true, false, false, false, true>(Xw, L.ldi, 1, g.u_in, g.R1, 1, 16 * rt, 0, g.B, g.R1, g.Ci, lane, a, g.abl);
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_store<false>(a[k], Zw, L.ld1, 16 * rt, 16 * k, g.B, g.R1, lane, g.abl);
    }
    SC_WAVE_SYNC();
    // ---- t_w = z_w T3_w  (B(k = f, j = g) = T3[g][f])
#pragma unroll 1
    for (int rt = 0; rt < rt_n; ++rt) {
      TkAcc a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_zero(a[k]);
      tk_multi<3, true, false, false>(Zw, L.ld1, 1, T3 + w * L.PT, 1, L.ldt, 16 * rt, 0, g.B, g.R2, g.R1, lane, a, g.abl);
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_store<false>(a[k], Tw, L.ld2, 16 * rt, 16 * k, g.B, g.R2, lane, g.abl);
    }
    SC_SYNC();                                       // every wave's z and t planes are complete
    { const int tid = sc_opaque(SC_TID);
    tkc_emit(ZB, L.PB, L.ld1, g.B * g.R1, g.R1, g.inv_r1, g.z, g.M, m0, tid);
    tkc_emit(XA, L.PA, L.ld2, g.B * g.R2, g.R2, g.inv_r2, g.t, g.M, m0, tid); }
    SC_SYNC();                                       // z and T3 are dead: their region takes the yhat planes
    // ---- yhat_w = t_w u_out^T  (B(k = g, j = o) = u_out[o][g])
#pragma unroll 1
    for (int rt = 0; rt < rt_n; ++rt) {
      TkAcc a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tk_zero(a[k]);
      tk_multi<4, true, false, false, false, true>(Tw, L.ld2, 1, g.u_out, 1, g.R2, 16 * rt, 0, g.B, g.Co, g.R2, lane, a, g.abl);
#pragma unroll
      for (int k = 0; k < 4; ++k) tk_store<false>(a[k], YP + w * PY, L.ldo, 16 * rt, 16 * k, g.B, g.Co, lane, g.abl);
    }
    SC_SYNC();
    tkc_emit(YP, PY, L.ldo, g.B * g.Co, g.Co, g.inv_co, g.yhat, g.M, m0, sc_opaque(SC_TID));
    SC_SYNC();                                       // the planes are restaged by the next tile
  }
}

// ------------------------------------------------------------------------------------------
// backward: PFY = ceil(2 B max(Ci, Co) / 256), PFR = ceil(2 B max(R1, R2) / 256), PFT = ceil(2 R1 R2 / 256)
// ------------------------------------------------------------------------------------------
template <int PFY, int PFR, int PFT>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 1)
k_tkc_bwd(TkcArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const TkcLayout L = tkc_layout(g.B, g.Ci, g.Co, g.R1, g.R2, true);
  const int tid = SC_TID, lane = tid & 63, w = SC_UNIFORM(tid >> 6);
  cf32* RA = lds + L.oA;
  cf32* RB = lds + L.oB;
  cf32* RC = lds + L.oC;
  const int PC = g.B * L.ld2;
  const int R12 = g.R1 * g.R2, n16 = 2 * R12;
  const int rt_n = (g.B + 15) >> 4;
  const bool own_o = 16 * w < g.Co, own_i = 16 * w < g.Ci;    // this wave's row tile of gu_out / gu_in exists
  TkAcc auo[3], aui[3];                              // rows 16 w .. of gu_out[o][g] / gu_in[i][f], all tiles of this workgroup
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    tk_zero(auo[k]);
    tk_zero(aui[k]);
  }
  for (int round = 0;; ++round) {
    const int tile = tkc_tile(round, g.n_wg);
    if (tile >= g.n_tiles) break;
    const int64_t m0 = (int64_t)tile * 4;
    // ---- gy -> A [b][ldo], t -> B [b][ld2]
    {
      const int tid = sc_opaque(SC_TID);               // (see k_tkc_fwd)
      sc_f4 vy[PFY], vr[PFR];
      tkc_fetch<PFY>(g.gy, g.B * g.Co, g.M, m0, tid, vy);
      tkc_fetch<PFR>(g.tin, g.B * g.R2, g.M, m0, tid, vr);
      tkc_plant<PFY>(RA, L.PA, L.ldo, g.B * g.Co, g.Co, g.inv_co, tid, vy);
      tkc_plant<PFR>(RB, L.PB, L.ld2, g.B * g.R2, g.R2, g.inv_r2, tid, vr);
    }
    SC_SYNC();
    // ---- gt_w = gy_w conj(u_out)  (B(k = o, j = g) = u_out[o][g])  -> C_w [b][ld2]
#pragma unroll 1
    for (int rt = 0; rt < rt_n; ++rt) {
      TkAcc a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_zero(a[k]);
      tk_multi<3, true, false, true, false, true>(RA + w * L.PA, L.ldo, 1, g.u_out, g.R2, 1, 16 * rt, 0, g.B, g.R2, g.Co, lane, a, g.abl);
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_store<true>(a[k], RC + w * PC, L.ld2, 16 * rt, 16 * k, g.B, g.R2, lane, g.abl);
    }
    // ---- gu_out[o][g] += sum_b gy_m[b][o] conj(t_m[b][g]), rows o = 16 w .., all four modes
    if (own_o) {
#pragma unroll 1
      for (int m = 0; m < 4; ++m)
        tk_multi<3, true, false, true>(RA + m * L.PA, 1, L.ldo, RB + m * L.PB, L.ld2, 1, 16 * w, 0, g.Co, g.R2, g.B, lane, auo, g.abl);
    }
    SC_SYNC();                                       // gy and t are dead (gt planes are wave-private)
    // ---- z -> B [b][ld1], T3 -> A [f][ldt]
    {
      const int tid = sc_opaque(SC_TID);
      sc_f4 vr[PFR], vt[PFT];
      tkc_fetch<PFR>(g.zin, g.B * g.R1, g.M, m0, tid, vr);
      tkc_fetch_t3<PFT, false>(g.t3m + m0 * R12, n16, tid, vt);
      tkc_plant<PFR>(RB, L.PB, L.ld1, g.B * g.R1, g.R1, g.inv_r1, tid, vr);
      tkc_plant_t3<PFT, false>(RA, L.PA, L.ldt, n16, R12, g.R2, g.inv_r12, g.inv_r2, tid, vt);
    }
    SC_SYNC();
    // ---- gT3_w[f][g] = sum_b conj(z_w[b][f]) gt_w[b][g]  -> global gT3m[m0 + w][f][g]
    {
      cf32* dst = g.gt3m + (m0 + w) * R12;
#pragma unroll 1
      for (int ft = 0; 16 * ft < g.R1; ++ft) {
        TkAcc a[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) tk_zero(a[k]);
        tk_multi<3, true, true, false>(RB + w * L.PB, 1, L.ld1, RC + w * PC, L.ld2, 1, 16 * ft, 0, g.R1, g.R2, g.B, lane, a, g.abl);
#pragma unroll
        for (int k = 0; k < 3; ++k) tk_store<true>(a[k], dst, g.R2, 16 * ft, 16 * k, g.R1, g.R2, lane, g.abl);
      }
    }
    SC_WAVE_SYNC();
    // ---- gz_w = gt_w T3_w^H  (B(k = g, j = f) = conj T3[f][g])  -> B_w [b][ld1] (z_w is dead)
#pragma unroll 1
    for (int rt = 0; rt < rt_n; ++rt) {
      TkAcc a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_zero(a[k]);
      tk_multi<3, true, false, true>(RC + w * PC, L.ld2, 1, RA + w * L.PA, 1, L.ldt, 16 * rt, 0, g.B, g.R1, g.R2, lane, a, g.abl);
      SC_WAVE_SYNC();                                // (rt = 0: every lane's reads of z_w in the product above are done)
#pragma unroll
      for (int k = 0; k < 3; ++k) tk_store<true>(a[k], RB + w * L.PB, L.ld1, 16 * rt, 16 * k, g.B, g.R1, lane, g.abl);
    }
    SC_SYNC();                                       // T3 is dead, every gz plane is complete
    // ---- X -> A [b][ldi]
    {
      const int tid = sc_opaque(SC_TID);
      sc_f4 vy[PFY];
      tkc_fetch<PFY>(g.xhat, g.B * g.Ci, g.M, m0, tid, vy);
      tkc_plant<PFY>(RA, L.PA, L.ldi, g.B * g.Ci, g.Ci, g.inv_ci, tid, vy);
    }
    SC_SYNC();
    // ---- gu_in[i][f] += sum_b conj(X_m[b][i]) gz_m[b][f], rows i = 16 w ..
    if (own_i) {
#pragma unroll 1
      for (int m = 0; m < 4; ++m)
        tk_multi<3, true, true, false>(RA + m * L.PA, 1, L.ldi, RB + m * L.PB, L.ld1, 1, 16 * w, 0, g.Ci, g.R1, g.B, lane, aui, g.abl);
    }
    // ---- gxhat_w = gz_w u_in^H  (B(k = f, j = i) = conj u_in[i][f])  -> A_w [b][ldi] once X is dead
    if (g.gxhat) {
      SC_SYNC();                                     // every wave has read X for gu_in
#pragma unroll 1
      for (int rt = 0; rt < rt_n; ++rt) {
        TkAcc a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) tk_zero(a[k]);
        tk_multi<4, true, false, true, false, true>(RB + w * L.PB, L.ld1, 1, g.u_in, 1, g.R1, 16 * rt, 0, g.B, g.Ci, g.R1, lane, a, g.abl);
#pragma unroll
        for (int k = 0; k < 4; ++k) tk_store<true>(a[k], RA + w * L.PA, L.ldi, 16 * rt, 16 * k, g.B, g.Ci, lane, g.abl);
      }
      SC_SYNC();
      tkc_emit(RA, L.PA, L.ldi, g.B * g.Ci, g.Ci, g.inv_ci, g.gxhat, g.M, m0, sc_opaque(SC_TID));
    }
    SC_SYNC();                                       // the planes are restaged by the next tile
  }
  // ---- this workgroup's partial sums of the two factor gradients: [Co][R2] then [Ci][R1]
  cf32* po = g.partial + (int64_t)SC_BID_X * (g.Co * g.R2 + g.Ci * g.R1);
  cf32* pi = po + g.Co * g.R2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (own_o) tk_store<true>(auo[k], po, g.R2, 16 * w, 16 * k, g.Co, g.R2, lane, g.abl);
    if (own_i) tk_store<true>(aui[k], pi, g.R1, 16 * w, 16 * k, g.Ci, g.R1, lane, g.abl);
  }
}

// gu_out[j] (j < n_out) / gu_in[j - n_out] = sum_k partial[k][j], k ascending (fixed order: run-to-run identical)
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tkc_reduce(const cf32* __restrict__ partial, int n, int n_out, int n_all, cf32* __restrict__ gu_out, cf32* __restrict__ gu_in) {
  SC_SHARED cf32 red[16][17];
  const int tid = SC_TID, cx = tid & 15, rg = tid >> 4;
  const int col = SC_BID_X * 16 + cx;
  cf32 acc = cf_make(0.f, 0.f);
  if (col < n_all) {
    for (int k = rg; k < n; k += 16) {
      const cf32 v = partial[(int64_t)k * n_all + col];
      acc.x += v.x;
      acc.y += v.y;
    }
  }
  red[rg][cx] = acc;
  SC_SYNC();
  if (rg == 0 && col < n_all) {
    cf32 t = red[0][cx];
    for (int r = 1; r < 16; ++r) {
      t.x += red[r][cx].x;
      t.y += red[r][cx].y;
    }
    if (col < n_out) {
      if (gu_out) gu_out[col] = t;
    } else if (gu_in) {
      gu_in[col - n_out] = t;
    }
  }
}

// out[c][r] = in[r][c] (complex), 32 x 32 tiles: T3[f g][m] <-> T3m[m][f g]
SC_GLOBAL void SC_LAUNCH_BOUNDS(256)
k_tkc_transpose(const cf32* __restrict__ in, cf32* __restrict__ out, int64_t rows, int64_t cols, int tiles_c) {
  SC_SHARED cf32 tile[32][33];
  const int tid = SC_TID, tx = tid & 31, ty = tid >> 5;
  const int64_t bid = SC_BID_X;
  const int64_t r0 = (bid / tiles_c) * 32, c0 = (bid % tiles_c) * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = in[r * cols + c];
  }
  SC_SYNC();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 8 * k];
  }
}
