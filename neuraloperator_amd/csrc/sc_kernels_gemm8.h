// sc_kernels_gemm8.h -- second generation of the dense (i,o) x mode block GEMM on the matrix cores: operands
// STREAMED into LDS by the LDS-DMA path (global_load_lds, 16 bytes per lane) through a ring of stages.
//
//   C[p, q, m] = sum_r opA(A[p, r, m]) * opB(B[r, q, m])        (complex, m = Fourier mode, mode stride 1)
//
// replaces tl.einsum('bixy,ioxy->boxy') and its two autograd einsums (spectral_convolution.py:21-46) when the
// operands are plain contiguous-mode arrays with 16-byte aligned rows (the full kept block of the layer);
// sub-blocks through index tables, factor matrices and small / ragged channel counts stay on
// k_modegemm_mfma / k_modegemm (same results).
//
// Why a second generation (profiles/r01_mfma_gemm_ablation.txt): k_modegemm_mfma stages a K-slice with 8-byte
// register loads of 72-byte segments, commits it to LDS, then computes: its loads alone take 48 us for 104 MB.
// Here
//  * a workgroup owns 2 GS CONSECUTIVE modes (GS = granules of 16 bytes = 2 modes x (re, im) per segment): every
//    (p, r) / (r, q) element of the unit is one aligned segment of 64 (GS = 4) or 128 bytes (GS = 8, one whole cache
//    line: the form taken whenever the mode count is a multiple of 16 -- 2112 = 132 x 16 at the metric shape); a
//    wave's LDS-DMA instruction moves 64 / GS segments = 1 KiB without touching a VGPR.  Measured with 64-byte
//    segments (profiles/r02_gemm_dma_v1_ab.txt): the operand stream alone takes 36 us even from the Infinity
//    Cache, independent of the ring depth, and the MFMA time ADDS to it -- the CU's memory path is limited in
//    requests, not bytes, and a wave that cannot issue its next LDS-DMA cannot issue its MFMAs either.  Hence whole
//    lines per request and twice the waves (GS of them: two per SIMD and workgroup, two workgroups per CU);
//  * a stage = SUB pairs of r values: per pair A 2 x 32 rows + B 2 x 16 QT columns; D stages form a ring, D - 2
//    of them are in flight while one feeds the matrix cores; the only synchronisation is ONE raw s_barrier per stage
//    behind a COUNTED s_waitcnt vmcnt (loads of later stages stay in flight across it);
//  * the LDS image of a piece is lane-linear (hardware), so the bank swizzle is applied to the SOURCE address:
//    granule t of segment s lands in slot t ^ f(s), f(s) = (s / (16 / GS)) mod GS; with it the ds_read_b128
//    operand fetches (32 rows or columns, stride 16 GS bytes) are conflict free (cdna_hip_programming.md 5.4 rule 21);
//  * wave w owns mode pair (2 w, 2 w + 1) of the unit for the whole 32 x 16 QT tile: per stage ONE 16-byte read
//    gives its A' operands of both modes (re and im parts of r0 / r0+1: lane = (row, k)), QT reads give B'; the
//    complex product is the real GEMM with K doubled as in generation 1, but the two MFMAs of an r pair take
//    k = (r0, r0+1) for the real parts and again for the imaginary parts, so no lane needs a value another loaded:
//        A'[p][(r, re)] = Re A,  A'[p][(r, im)] = Im A
//        B'[(r, re)][(q, 0 / 1)] = Re B / Im B,   B'[(r, im)][(q, 0 / 1)] = -Im B / Re B
//    (conjugations are sign masks).  4 QT v_mfma_f32_32x32x2_f32 per wave and stage; exact fp32 (k-ordered fma chain).
//    The 4-wave shape uses THREE real products per complex product instead (SC_G8_USE_3M below: N = complex column,
//    12 MFMAs per stage, one B fetch per lane);
//  * C leaves through 32 KiB LDS patches ([rows][16 cols][GS slots], same swizzle) as whole segments, 16-byte stores.
// A workgroup may run several (row block, column block) tiles of its mode group one after the other (bpw): every
// launch of the metric shape thereby has 264 workgroups, all co-resident.
#pragma once
#include "sc_device.h"
#include "sc_kernels_mfma.h"

struct Gemm8Args {
  int P, Q, R;
  int n_mg, n_pb, n_qb, bpw;   // mode groups of 2 GS, row blocks of 32, column blocks of 16 QT, tiles per workgroup
  int G;                       // grid size = n_mg * ceil(n_pb * n_qb / bpw)
  int64_t a_sp, a_sr, b_sr, b_sq, c_sp, c_sq;   // complex elements; the mode stride of all three is 1
  int64_t a_sg, b_sg, c_sg;    // element offset from one group of 16 modes to the next (16 for plain arrays; anything
                               // for a group-major "tiled spectrum": [group][row][col][16 modes])
  int stream_c;                // 1: C is not read by the next kernel -> non-temporal stores
};

template <int GS, int QT, int SUB = 1>
struct Gemm8Cfg {
  static constexpr int NW = GS;                        // waves = granules of a segment (mode pairs of the unit)
  static constexpr int THREADS = 64 * NW;
  static constexpr int MODES = 2 * GS;
  static constexpr int COLS = 16 * QT;
  static constexpr int SP = 64 / GS;                   // segments per LDS-DMA piece
  static constexpr int SH = (GS == 4) ? 2 : 1;         // log2(16 / GS)
  static constexpr int A_G = 2 * 32 * GS;              // granules of a stage: A [kk][32 rows][GS]
  static constexpr int B_G = 2 * COLS * GS;            //                      B [kk][COLS][GS]
  static constexpr int SUB_G = A_G + B_G;              // granules of one r pair
  static constexpr int STAGE_G = SUB * SUB_G;          // a stage = SUB r pairs (one barrier per stage)
  static constexpr int NPA = A_G / 64, NPB = B_G / 64; // pieces per r pair
  static constexpr int PPW = SUB * (NPA + NPB) / NW;   // pieces per wave and stage
  static constexpr int RP = 2048 / (16 * GS);          // rows of an epilogue patch (2048 granules = 32 KiB)
  static constexpr int EPW = 32 / GS;                  // patch stores per wave
  // waves per SIMD the register allocator must leave room for: as many workgroups as fit the CU's 160 KiB of LDS
  // with a ring of D stages, NW / 4 waves per SIMD each
  static constexpr int occupancy(const int D) {
    const int wgs = 163840 / (D * STAGE_G * 16);
    const int w = wgs * NW / 4;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
  }
  static_assert(GS == 4 || GS == 8, "64- or 128-byte segments");
  static_assert(NPA == NW && NPB % NW == 0, "one A piece and NPB / NW B pieces per wave");
  static_assert(COLS % SP == 0, "column pieces");
};

// Three real products per complex product on the narrow (non-pipelined) shape:
//     P1 = Re A Re B,  P2 = Im A Im B,  P3 = (Re A + Im A)(Re B + Im B);   Re C = P1 - P2,  Im C = P3 - P1 - P2
// Each MFMA then covers 32 COMPLEX columns (N = column, not (column, re / im)): 3 MFMAs per r pair and mode for the
// 32 x 32 tile instead of 4, one 16-byte B fetch per lane instead of two, no re / im select.  The contraction's
// matrix-pipe time is NOT hidden behind its memory time (profiles/r02_gemm_dma_v6_ablation.txt: forward 39.9 us,
// 29.6 us without the MFMAs), so a quarter less of it shows.  Costs: 96 accumulator registers per wave instead of
// 64 (fine at 3 waves per SIMD; the 8-wave pipelined shape keeps four products) and one more rounding per output:
// the results differ from the four-product kernels in the last bits (2-3 ulp of |A||B| instead of 1-2).
#ifdef SC_G8_NO_3M                       // measurement builds only
#define SC_G8_USE_3M(GS, QT, IL) false
#else
#define SC_G8_USE_3M(GS, QT, IL) ((GS) == 4 && (QT) == 2 && !(IL))
#endif

#ifndef SC_EMU
SC_DEVICE void sc_store16(cf32* dst, const sc_f4 v, const int stream) {
  if (stream) __builtin_nontemporal_store(v, reinterpret_cast<sc_f4*>(dst));
  else *reinterpret_cast<sc_f4*>(dst) = v;
}
#else
inline void sc_store16(cf32* dst, const sc_f4 v, const int) { std::memcpy(dst, &v, 16); }
#endif

// wait until the stage needed now has landed: in the steady state exactly NEWER stages (PPW LDS-DMA instructions
// per wave each) were requested after it and stay in flight; near the end of the r loop fewer exist and the wave
// simply drains its queue
template <int NEWER, int PPW>
SC_DEVICE void g8_wait_stage(const int newer_issued) {
  static_assert(NEWER >= 0 && NEWER * PPW <= 63, "vmcnt range");
  if (newer_issued >= NEWER) sc_wait_vmcnt<NEWER * PPW>();
  else sc_wait_vmcnt<0>();
}

// IL: software-pipelined stage -- the next stage's LDS-DMA requests, operand fetches and operand preparation are
// issued BETWEEN this stage's MFMAs (a v_mfma_f32_32x32x2_f32 occupies the matrix pipe for 64 cycles, during which
// the wave may issue other instructions), operands double-buffered in registers
// the work of ONE workgroup (index gid of g.G) of a contraction; `lds` is the kernel's D-stage ring
template <int GS, int QT, int SUB, int D, bool IL, bool CA, bool CB>
SC_DEVICE void g8_workgroup(const Gemm8Args& g, const cf32* __restrict__ A, const cf32* __restrict__ B,
                            cf32* __restrict__ C, int gid, sc_f4* lds) {
  typedef Gemm8Cfg<GS, QT, SUB> K;
  static_assert(D * K::STAGE_G >= 2048, "the epilogue patch (32 KiB) lives in the ring");

  const int tid = SC_TID, lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  // ---- which mode group / which tiles: neighbouring workgroups of one XCD (block b runs on XCD b % 8) take
  //      neighbouring units, so the tiles of one mode group (which share A or B) meet in one L2
  const int nblk = g.n_pb * g.n_qb;
  const int chunks = (nblk + g.bpw - 1) / g.bpw;
  if ((g.G & 7) == 0) gid = (gid & 7) * (g.G >> 3) + (gid >> 3);
  const int mg = gid / chunks, ch = gid - mg * chunks;
  constexpr int GPG = 16 / K::MODES;                           // workgroups per group of 16 modes
  const int64_t mlo = (int64_t)(mg % GPG) * K::MODES;
  const int64_t mA = (int64_t)(mg / GPG) * g.a_sg + mlo, mB = (int64_t)(mg / GPG) * g.b_sg + mlo,
                mC = (int64_t)(mg / GPG) * g.c_sg + mlo;

  // ---- loader role: lane = (segment ls of the piece, slot ltq); it fetches granule ltq ^ f(row or column).
  //      This wave's pieces of every stage: piece w is an A piece (r0 + ra_k, rows ra_g SP ..), pieces
  //      NPA + w + NW j are B pieces (r0 + rb_k, columns rb_g SP ..)
  const int ls = lane / GS, ltq = lane % GS;
  constexpr int APR = 32 / K::SP, BPR = K::COLS / K::SP;       // pieces per r
  constexpr int NBW = K::NPB / K::NW;                          // B pieces per wave and r pair
  const int ra_k = w / APR, ra_g = w % APR;
  const int rowA = ra_g * K::SP + ls;                          // row inside the tile
  const int lgrA = ltq ^ ((rowA >> K::SH) & (GS - 1));
  // ---- MFMA role: lane = (k half kk, row i) for A', (kk, column n = 2 qq + d) for B'
  const int kk = lane >> 5, li = lane & 31, d = lane & 1, qq = li >> 1;
  const int a_g = (kk * 32 + li) * GS + (w ^ ((li >> K::SH) & (GS - 1)));
  const int b_g = K::A_G + (kk * K::COLS + qq) * GS + (w ^ ((qq >> K::SH) & (GS - 1)));   // + u * 16 * GS
  constexpr bool M3 = SC_G8_USE_3M(GS, QT, IL);
  constexpr int NB = M3 ? 1 : QT;                            // B fetches per lane and r pair
  // Four-product shape with a conjugated B (round 5): a conj(b) = conj(conj(a) b), so the loop runs the un-conjugated-B
  // code with A's conjugation toggled and the epilogue negates the Im lanes -- the same products in the same order
  // with two exact sign changes, bit-identical to the direct form.  The direct form kept a second lane-dependent sign
  // mask alive through the pipelined loop: <8, 2, 1, 4, true, *, true> spilled 4 registers at its 128 (20 bytes of
  // scratch).
  constexpr bool FLIP = CB && !M3;
  constexpr bool CAe = FLIP ? !CA : CA, CBe = FLIP ? false : CB;
  const int b_g3 = K::A_G + (kk * K::COLS + li) * GS + (w ^ ((li >> K::SH) & (GS - 1)));  // M3: lane = (kk, column li)
  const uint32_t m_re = (CBe && d) ? 0x80000000u : 0u;       // B'[(r,re)][(q,1)] = Im B  (conj: -Im B)
  const uint32_t m_im = (!CBe && !d) ? 0x80000000u : 0u;     // B'[(r,im)][(q,0)] = -Im B (conj: +Im B)
  const uint32_t dsel = d ? 0xffffffffu : 0u;
  const int NS = (g.R + 2 * SUB - 1) / (2 * SUB);            // stages of 2 SUB values of r

  const int b_end = (ch + 1) * g.bpw < nblk ? (ch + 1) * g.bpw : nblk;
#pragma unroll 1
  for (int bi = ch * g.bpw; bi < b_end; ++bi) {
    const int pb = bi / g.n_qb, qb = bi - pb * g.n_qb;
    const int p0 = pb * 32, q0 = qb * K::COLS;
    // rows / columns / r values past the end are clamped to the last one (finite duplicates: never stored, or
    // multiplied by zero below)
    int prow = p0 + rowA;
    prow = prow < g.P ? prow : g.P - 1;
    const cf32* srcA = A + (int64_t)prow * g.a_sp + mA + 2 * lgrA;
    const cf32* srcB[NBW];
    int rb_k[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int kb = w + K::NW * j;                            // B piece index
      rb_k[j] = kb / BPR;
      const int colB = (kb % BPR) * K::SP + ls;
      int qcol = q0 + colB;
      qcol = qcol < g.Q ? qcol : g.Q - 1;
      srcB[j] = B + (int64_t)qcol * g.b_sq + mB + 2 * (ltq ^ ((colB >> K::SH) & (GS - 1)));
    }
    // stages are requested in order, so the sources are running pointers (r values past the end re-read the last
    // one: only the final r pair of an odd R, multiplied by zero below, or whole r pairs past NS * 2 SUB > R)
    const int64_t stepA = 2 * g.a_sr, stepB = 2 * g.b_sr;
    const cf32* pA = srcA + (int64_t)ra_k * g.a_sr;
    const cf32* pB[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) pB[j] = srcB[j] + (int64_t)rb_k[j] * g.b_sr;
    int r_next = 0;                                       // first r of the next r pair to request
    // piece pc of a stage: r pair sub = pc / (1 + NBW); its A piece first, then its NBW B pieces
    auto issue_piece = [&](const int buf, const int pc) {
      sc_f4* sb = lds + buf * K::STAGE_G;
      const int sub = pc / (1 + NBW), e = pc % (1 + NBW);
      if (e == 0) {
        const int64_t backA = (r_next + ra_k < g.R) ? 0 : (int64_t)(r_next + ra_k - (g.R - 1)) * g.a_sr;
        SC_GLDS16(pA - backA, sb + sub * K::SUB_G + w * 64);
        pA += stepA;
      }
#pragma unroll
      for (int j = 0; j < NBW; ++j)
        if (e == 1 + j) {
          const int64_t backB = (r_next + rb_k[j] < g.R) ? 0 : (int64_t)(r_next + rb_k[j] - (g.R - 1)) * g.b_sr;
          SC_GLDS16(pB[j] - backB, sb + sub * K::SUB_G + K::A_G + (w + K::NW * j) * 64);
          pB[j] += stepB;
        }
      if (e == NBW) r_next += 2;
    };
    auto issue = [&](const int buf) {
#pragma unroll
      for (int pc = 0; pc < K::PPW; ++pc) issue_piece(buf, pc);
    };
    struct Ops {                 // MFMA operands of one stage: (Re, Im) of both modes for this lane's row / columns
      sc_f4 a;
      sc_f4 b[NB];
    };
    auto fetch = [&](const int buf, Ops (&o)[SUB]) {
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const sc_f4* sb = lds + buf * K::STAGE_G + sub * K::SUB_G;
        o[sub].a = sb[a_g];
        if constexpr (M3) o[sub].b[0] = sb[b_g3];
        else {
#pragma unroll
          for (int u = 0; u < QT; ++u) o[sub].b[u] = sb[b_g + u * 16 * GS];
        }
      }
    };

    // four products: acc[mode j][column tile u] = 32 rows x (16 columns x (re, im));
    // three products: acc[mode j][P1 / P2 / P3] = 32 rows x 32 columns
    constexpr int NACC = M3 ? 3 : QT;
    sc_f32x16 acc[2][NACC];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < NACC; ++u)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][u][v] = 0.f;

    // operands of one stage in MFMA order.  B': lane (q, d) needs Re / Im B for the (r, re) rows and -Im / Re B for
    // the (r, im) rows: a bitwise select on the lane's d and a sign mask (written as a ternary on vector components
    // the compiler builds a variable-index extract, three v_cndmask per value)
    struct Prep {
      float ar[2], ai[2], br[2][QT], bm[2][QT];
    };
    auto bsel = [&](const float hi, const float lo) { return sc_bitsel(dsel, hi, lo); };      // d ? hi : lo
    auto prep = [&](const Ops& o, const int r0, Prep& q) {
      const float keep = (r0 + kk < g.R) ? 1.f : 0.f;         // r values past the end: the duplicate contributes 0
      q.ar[0] = o.a.x * keep;
      q.ai[0] = (CAe ? -o.a.y : o.a.y) * keep;
      q.ar[1] = o.a.z * keep;
      q.ai[1] = (CAe ? -o.a.w : o.a.w) * keep;
      if constexpr (!M3) {
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          q.br[0][u] = sc_xor_sign(bsel(o.b[u].y, o.b[u].x), m_re);
          q.bm[0][u] = sc_xor_sign(bsel(o.b[u].x, o.b[u].y), m_im);
          q.br[1][u] = sc_xor_sign(bsel(o.b[u].w, o.b[u].z), m_re);
          q.bm[1][u] = sc_xor_sign(bsel(o.b[u].z, o.b[u].w), m_im);
        }
      }
    };
    auto fire = [&](const Prep& q) {
#ifdef SC_G8_ABL_NOMFMA              // measurement builds only (scripts/gemm8_ab.py): operand stream without the matrix cores
#pragma unroll
      for (int u = 0; u < QT; ++u)
        asm volatile("" ::"v"(q.br[0][u]), "v"(q.br[1][u]), "v"(q.bm[0][u]), "v"(q.bm[1][u]), "v"(q.ar[0]), "v"(q.ai[1]));
      return;
#endif
      SC_SCHED_BARRIER();
      // consecutive MFMAs write different accumulators
#pragma unroll
      for (int u = 0; u < QT; ++u) {
        sc_mfma_32x32x2(acc[0][u], q.ar[0], q.br[0][u]);
        sc_mfma_32x32x2(acc[1][u], q.ar[1], q.br[1][u]);
      }
#pragma unroll
      for (int u = 0; u < QT; ++u) {
        sc_mfma_32x32x2(acc[0][u], q.ai[0], q.bm[0][u]);
        sc_mfma_32x32x2(acc[1][u], q.ai[1], q.bm[1][u]);
      }
      SC_SCHED_BARRIER();
    };

    if constexpr (!IL) {
    // ---- prologue: D - 1 stages in flight
    sc_wait_vmcnt<0>();                 // stores of the previous tile's epilogue are counted by vmcnt too
    const int npro = NS < D - 1 ? NS : D - 1;
    for (int st = 0; st < npro; ++st) issue(st);
    // one stage: counted wait for stage st + barrier (publishes it; everybody has finished READING stage st - 1, so
    // its buffer is refilled with stage st + D - 1), then fetch / prepare / multiply r pair by r pair.  No operand
    // double-buffering in registers: the latencies exposed after the barrier are covered by the OTHER workgroups of
    // the CU (measured: one workgroup alone is latency-bound whatever it does; fewer registers = more of them)
#pragma unroll 1
    for (int st = 0; st < NS; ++st) {
      g8_wait_stage<D - 2, K::PPW>(NS - 1 - st);
      SC_WAIT_LGKM0();
      SC_BARRIER_RAW();
      if (st + D - 1 < NS) issue((st + D - 1) % D);
      Ops o[SUB];
      fetch(st % D, o);
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        if constexpr (M3) {
          const int r0 = 2 * (st * SUB + sub);
          const float keep = (r0 + kk < g.R) ? 1.f : 0.f;       // r values past the end: the duplicate contributes 0
          const sc_f4 a = o[sub].a, b = o[sub].b[0];
          const float ar0 = a.x * keep, ai0 = (CA ? -a.y : a.y) * keep, ar1 = a.z * keep, ai1 = (CA ? -a.w : a.w) * keep;
          const float bi0 = CB ? -b.y : b.y, bi1 = CB ? -b.w : b.w;
#ifdef SC_G8_ABL_NOMFMA
          asm volatile("" ::"v"(ar0 + ai0), "v"(ar1 + ai1), "v"(b.x + bi0), "v"(b.z + bi1));
#else
          SC_SCHED_BARRIER();
          sc_mfma_32x32x2(acc[0][0], ar0, b.x);
          sc_mfma_32x32x2(acc[1][0], ar1, b.z);
          sc_mfma_32x32x2(acc[0][1], ai0, bi0);
          sc_mfma_32x32x2(acc[1][1], ai1, bi1);
          sc_mfma_32x32x2(acc[0][2], ar0 + ai0, b.x + bi0);
          sc_mfma_32x32x2(acc[1][2], ar1 + ai1, b.z + bi1);
          SC_SCHED_BARRIER();
#endif
        } else {
          Prep q;
          prep(o[sub], 2 * (st * SUB + sub), q);
          fire(q);
        }
      }
    }
    } else {
    // ---- software-pipelined form: D stages requested ahead, operands of the next stage prepared during this one
    sc_wait_vmcnt<0>();
    const int npro = NS < D ? NS : D;
    for (int st = 0; st < npro; ++st) issue(st);
    g8_wait_stage<D - 1, K::PPW>(npro - 1);
    SC_BARRIER_RAW();
    Prep q0[SUB], q1[SUB];
    {
      Ops o[SUB];
      fetch(0, o);
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) prep(o[sub], 2 * sub, q0[sub]);
    }
    constexpr int NM = SUB * 4 * QT;                       // MFMAs of a stage
    constexpr int NF = SUB * (1 + QT);                     // operand fetches of a stage
    static_assert(K::PPW + NF + SUB <= NM, "one filler per MFMA");
    auto stage = [&](const int st, const Prep (&qc)[SUB], Prep (&qn)[SUB]) {
      const bool more = st + 1 < NS;
      if (more) {
        g8_wait_stage<D - 2, K::PPW>(NS - 2 - st);
        SC_WAIT_LGKM0();                // this wave's fetches of stage st are in registers
        SC_BARRIER_RAW();               // ... everybody's: buffer st % D is free, stage st + 1 is complete
      }
      const bool refill = st + D < NS;
      const sc_f4* sbn = lds + ((st + 1) % D) * K::STAGE_G;
      Ops on[SUB];
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        // MFMA m: r pair, re / im part, column tile, mode -- consecutive MFMAs write different accumulators
        const int sub = m / (4 * QT), part = (m / (2 * QT)) & 1, u = (m >> 1) % QT, j = m & 1;
        sc_mfma_32x32x2(acc[j][u], part ? qc[sub].ai[j] : qc[sub].ar[j], part ? qc[sub].bm[j][u] : qc[sub].br[j][u]);
        SC_SCHED_BARRIER();
        // filler m: one piece of work for the NEXT stages, issued while the matrix pipe is busy
        if (m < K::PPW) {
          if (refill) issue_piece(st % D, m);
        } else if (m < K::PPW + NF) {
          const int f = m - K::PPW, fs = f / (1 + QT), fe = f % (1 + QT);
          if (more) {
            if (fe == 0) on[fs].a = sbn[fs * K::SUB_G + a_g];
            else on[fs].b[fe - 1] = sbn[fs * K::SUB_G + b_g + (fe - 1) * 16 * GS];
          }
        } else if (m < K::PPW + NF + SUB) {
          const int ps = m - K::PPW - NF;
          if (more) prep(on[ps], 2 * ((st + 1) * SUB + ps), qn[ps]);
        }
        SC_SCHED_BARRIER();
      }
    };
#pragma unroll 1
    for (int st = 0; st < NS; st += 2) {
      stage(st, q0, q1);
      if (st + 1 < NS) stage(st + 1, q1, q0);
    }
    }

    // ---- C: per (16-column tile, RP rows) the waves fill a [RP rows][16 cols][GS slots] granule patch (each wave
    //      its mode pair: slot w ^ f(col)), then everybody stores whole segments with 16-byte stores: store e of
    //      wave w covers 64 / GS segments, lane = (segment ls, slot ltq) as in the loader role
    float* patch = reinterpret_cast<float*>(lds);
    SC_WAIT_LGKM0();
    SC_BARRIER_RAW();                   // nobody still reads operands from the ring
    const bool full = p0 + 32 <= g.P && q0 + K::COLS <= g.Q;
    if constexpr (M3) {
      // three products: a lane holds P1 / P2 / P3 of column li (both column tiles at once) for 16 of the 32 rows;
      // the patch is [16 rows][32 cols][GS slots] granules, lane writes (re, im) of its mode pair as one 8-byte word
      static_assert(K::RP == 32 && K::COLS == 32, "3-product epilogue: 32 x 32 tiles, 32 KiB patches");
      const int pslot3 = (li * GS + (w ^ ((li >> K::SH) & (GS - 1)))) * 4;      // + patch row * 128 GS + 2 j
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h) SC_BARRIER_RAW();        // the previous patch has been read (reads feed stores: complete)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2);              // + 4 kk: row of the 32-row tile
            if (row / 16 == h) {                                 // compile time
              const float p1 = acc[j][0][v], p2 = acc[j][1][v], p3 = acc[j][2][v];
              float* dst = patch + pslot3 + ((row % 16) + 4 * kk) * (128 * GS) + 2 * j;
              dst[0] = p1 - p2;
              dst[1] = (p3 - p1) - p2;
            }
          }
        SC_WAIT_LGKM0();
        SC_BARRIER_RAW();
        sc_f4 val[K::EPW];
#pragma unroll
        for (int e = 0; e < K::EPW; ++e) val[e] = lds[(w * K::EPW + e) * 64 + lane];
#pragma unroll
        for (int e = 0; e < K::EPW; ++e) {
          const int sbase = (w * K::EPW + e) * K::SP;            // first segment of this store (uniform)
          const int prow_ = p0 + h * 16 + (sbase >> 5);          // uniform
          const int col = (sbase & 31) + sc_opaque(ls);
          const int qc = q0 + col;
          const int gr = ltq ^ ((col >> K::SH) & (GS - 1));
#ifdef SC_G8_ABL_NOSTORE
          asm volatile("" ::"v"(val[e]));
          if (g.P < 0)
#endif
          if (full || (prow_ < g.P && qc < g.Q))
            sc_store16(C + (int64_t)prow_ * g.c_sp + (int64_t)qc * g.c_sq + mC + 2 * gr, val[e], g.stream_c);
        }
      }
    } else {
      const int pslot = ((w ^ ((qq >> K::SH) & (GS - 1))) * 4 + d) + qq * 4 * GS;   // + patch row * 64 GS + 2 j
      bool first = true;
#pragma unroll
      for (int u = 0; u < QT; ++u) {
#pragma unroll
        for (int h = 0; h < 32 / K::RP; ++h) {
          if (!first) SC_BARRIER_RAW();   // the previous patch has been read (reads feed stores: complete)
          first = false;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const int row = (v & 3) + 8 * (v >> 2);              // + 4 kk: row of the 32-row tile
              if (row / K::RP == h)                                // compile time
                patch[pslot + ((row % K::RP) + 4 * kk) * (64 * GS) + 2 * j] = acc[j][u][v];
            }
          SC_WAIT_LGKM0();
          SC_BARRIER_RAW();
          sc_f4 val[K::EPW];
#pragma unroll
          for (int e = 0; e < K::EPW; ++e) {
            val[e] = lds[(w * K::EPW + e) * 64 + lane];
            if constexpr (FLIP) {                                  // a granule = (re, im) of the unit's two modes
              val[e].y = -val[e].y;
              val[e].w = -val[e].w;
            }
          }
#pragma unroll
          for (int e = 0; e < K::EPW; ++e) {
            const int sbase = (w * K::EPW + e) * K::SP;            // first segment of this store (uniform)
            const int prow_ = p0 + h * K::RP + (sbase >> 4);       // uniform
            const int col = (sbase & 15) + sc_opaque(ls);
            const int qc = q0 + u * 16 + col;
            const int gr = ltq ^ ((col >> K::SH) & (GS - 1));
#ifdef SC_G8_ABL_NOSTORE             // measurement builds only: everything but the C stores
            asm volatile("" ::"v"(val[e]));
            if (g.P < 0)
#endif
            if (full || (prow_ < g.P && qc < g.Q))
              sc_store16(C + (int64_t)prow_ * g.c_sp + (int64_t)qc * g.c_sq + mC + 2 * gr, val[e], g.stream_c);
          }
        }
      }
    }
    SC_WAIT_LGKM0();
    SC_BARRIER_RAW();                   // the ring is refilled by the next tile
  }
}

template <int GS, int QT, int SUB, int D, bool IL, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC((Gemm8Cfg<GS, QT, SUB>::THREADS), (Gemm8Cfg<GS, QT, SUB>::occupancy(D)))
k_modegemm_dma(Gemm8Args g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ C) {
  SC_SHARED __attribute__((aligned(16))) sc_f4 lds[D * Gemm8Cfg<GS, QT, SUB>::STAGE_G];
  g8_workgroup<GS, QT, SUB, D, IL, CA, CB>(g, A, B, C, SC_BID_X, lds);
}

// ------------------------------------------------------------------------------------------
// The two contractions of the backward pass in ONE launch (sc_layer_backward):
//   job 0:  gW[i,o,m]    = sum_b conj(xhat[b,i,m]) ghat[b,o,m]       (conj A)
//   job 1:  gxhat[b,i,m] = sum_o ghat[b,o,m] conj(W[i,o,m])          (conj B)
// plus, in a few trailing workgroups, the bias gradient (the sum of ghat's zero-frequency coefficients over the
// batch).  Why: a contraction launch of the metric shape is 528 workgroups on 256 CUs -- 16 CUs carry three where
// the average is 2.06, and the launch is as slow as its busiest CU (profiles/r02_gemm_dma_v4_grid_scaling.txt:
// 512 / 528 / 1024 workgroups take 35.9 / 43.6 / 68.8 us).  Together the two jobs are 1056 workgroups: the second
// round of the one fills the tail of the other, both read ghat while it is in the Infinity Cache, and two launch
// boundaries disappear.  Octets of consecutive workgroups (one per XCD) alternate between the jobs, so inside each
// job workgroup k still runs on XCD k % 8 and g8_workgroup's XCD-aware unit order holds.
// ------------------------------------------------------------------------------------------
struct Gemm8Bias {
  const cf32* ghat;      // null: no bias role
  float* gbias;
  int64_t batch, channels, modes_per_image, dc;
};

template <int GS, int QT, int SUB, int D, bool IL>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC((Gemm8Cfg<GS, QT, SUB>::THREADS), (Gemm8Cfg<GS, QT, SUB>::occupancy(D)))
k_modegemm_dma_bwd(Gemm8Args g0, const cf32* __restrict__ A0, const cf32* __restrict__ B0, cf32* __restrict__ C0,
                   Gemm8Args g1, const cf32* __restrict__ A1, const cf32* __restrict__ B1, cf32* __restrict__ C1,
                   Gemm8Bias bias) {
  typedef Gemm8Cfg<GS, QT, SUB> K;
  SC_SHARED __attribute__((aligned(16))) sc_f4 lds[D * K::STAGE_G];
  const int b = SC_BID_X;
  const int n0 = g0.G >> 3, n1 = g1.G >> 3;                 // octets of each job (the host passes G % 8 == 0)
  const int nmin = n0 < n1 ? n0 : n1;
  const int oct = b >> 3, l8 = b & 7;
  if (oct < n0 + n1) {
    // octets alternate between the jobs while both have some left; the longer job takes the rest
    const bool alt = oct < 2 * nmin;
    const int job = alt ? (oct & 1) : (n1 > n0 ? 1 : 0);
    const int k = (alt ? (oct >> 1) : (oct - nmin)) * 8 + l8;
    if (job) g8_workgroup<GS, QT, SUB, D, IL, false, true>(g1, A1, B1, C1, k, lds);
    else g8_workgroup<GS, QT, SUB, D, IL, true, false>(g0, A0, B0, C0, k, lds);
  } else {
    // bias role: wave w of trailing workgroup t takes channel t * NW + w
    const int tid = SC_TID, lane = tid & 63, w = tid >> 6;
    const int64_t c = (int64_t)(b - 8 * (n0 + n1)) * K::NW + w;
    if (c < bias.channels)
      sc_bias_grad_wave(bias.ghat, bias.gbias, bias.batch, bias.channels, bias.modes_per_image, bias.dc, c, lane,
                        reinterpret_cast<float*>(lds) + 64 * w);
  }
}

