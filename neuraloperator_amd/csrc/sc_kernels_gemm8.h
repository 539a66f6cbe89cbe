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
// register loads, commits it to LDS, then computes -- eight exposed HBM round trips per launch with <= 60 KB in
// flight per CU: its loads alone take 48 us for 104 MB (2.2 TB/s).  Here
//  * a workgroup owns 8 CONSECUTIVE modes (2112 = 264 x 8 at the metric shape): every (p, r) / (r, q) element of
//    the unit is one 64-byte, 64-byte-aligned segment = 4 granules of 16 bytes (2 modes x (re, im)); a wave's
//    LDS-DMA instruction moves 16 segments = 1 KiB without touching a VGPR;
//  * a stage = 2 values of r: A 2 x 32 segments + B 2 x 64 segments = 12 KiB = 12 instructions (3 per wave);
//    D stages form a ring (D = 6: 72 KiB -> two workgroups per CU), D - 1 of them are in flight while one feeds
//    the matrix cores; the only synchronisation is ONE raw s_barrier per stage behind a COUNTED s_waitcnt vmcnt
//    (loads of later stages stay in flight across it);
//  * the LDS image of a piece is lane-linear (hardware), so the bank swizzle is applied to the SOURCE address:
//    granule t of segment s lands in slot t ^ ((s >> 2) & 3); with it the ds_read_b128 operand fetches (32
//    rows or columns, stride 64 bytes) are conflict free (cdna_hip_programming.md 5.4 rule 21);
//  * wave w owns mode pair (2 w, 2 w + 1) for the whole 32 x 64 tile: per stage ONE 16-byte read gives its A'
//    operands of both modes (re and im parts of r0 / r0+1: lane = (row, k)), four reads give B'; the complex
//    product is the real GEMM with K doubled as in generation 1, but the two MFMAs of an r pair take k = (r0, r0+1)
//    for the real parts and again for the imaginary parts, so no lane ever needs a value another lane loaded:
//        A'[p][(r, re)] = Re A,  A'[p][(r, im)] = Im A
//        B'[(r, re)][(q, 0 / 1)] = Re B / Im B,   B'[(r, im)][(q, 0 / 1)] = -Im B / Re B
//    (conjugations are sign masks).  16 v_mfma_f32_32x32x2_f32 per wave and stage; exact fp32 (k-ordered fma chain);
//  * C leaves through a 32 KiB LDS patch per 16-column tile in the same swizzled granule order and is stored as
//    64-byte segments with 16-byte stores.
// A workgroup may run several (row block, column block) tiles of its mode group one after the other (bpw):
// the weight-gradient launch of the metric shape (64 x 64 outputs per mode) thereby also has 264 workgroups of 128
// accumulator registers per lane, all co-resident.
#pragma once
#include "sc_device.h"
#include "sc_kernels_mfma.h"

#define SC_G8_STAGE_G 768      // granules (16 B) per stage
#define SC_G8_B_OFF 256        // first B granule of a stage (A: [kk][32 rows][4], B: [kk][64 cols][4])

struct Gemm8Args {
  int P, Q, R;
  int n_mg, n_pb, n_qb, bpw;   // mode groups of 8, row blocks of 32, column blocks of 64, tiles per workgroup
  int G;                       // grid size = n_mg * ceil(n_pb * n_qb / bpw)
  int64_t a_sp, a_sr, b_sr, b_sq, c_sp, c_sq;   // complex elements; the mode stride of all three is 1
  int stream_c;                // 1: C is not read by the next kernel -> non-temporal stores
};

struct Gemm8Ops {              // MFMA operands of one stage: (Re, Im) of both modes for this lane's row / column
  sc_f4 a;
  sc_f4 b[4];
};

#ifndef SC_EMU
SC_DEVICE void sc_store16(cf32* dst, const sc_f4 v, const int stream) {
  if (stream) __builtin_nontemporal_store(v, reinterpret_cast<sc_f4*>(dst));
  else *reinterpret_cast<sc_f4*>(dst) = v;
}
#else
inline void sc_store16(cf32* dst, const sc_f4 v, const int) { std::memcpy(dst, &v, 16); }
#endif

// `rem` stages were requested after the one needed now (3 LDS-DMA instructions per wave and stage)
template <int D>
SC_DEVICE void g8_wait_stage(const int rem) {
  static_assert(D >= 2 && D <= 8, "ring depth");
  switch (rem < D - 1 ? rem : D - 1) {
    case 0: sc_wait_vmcnt<0>(); break;
    case 1: sc_wait_vmcnt<3>(); break;
    case 2: sc_wait_vmcnt<6>(); break;
    case 3: sc_wait_vmcnt<9>(); break;
    case 4: sc_wait_vmcnt<12>(); break;
    case 5: sc_wait_vmcnt<15>(); break;
    case 6: sc_wait_vmcnt<18>(); break;
    default: sc_wait_vmcnt<21>(); break;
  }
}

template <int D, bool CA, bool CB>
SC_GLOBAL void SC_LAUNCH_BOUNDS_OCC(256, 2)
k_modegemm_s8(Gemm8Args g, const cf32* __restrict__ A, const cf32* __restrict__ B, cf32* __restrict__ C) {
  SC_SHARED __attribute__((aligned(16))) sc_f4 lds[D * SC_G8_STAGE_G];
  static_assert(D * SC_G8_STAGE_G >= 2048, "the epilogue patch (32 KiB) lives in the ring");

  const int tid = SC_TID, lane = tid & 63;
  const int w = SC_UNIFORM(tid >> 6);
  // ---- which mode group / which tiles: neighbouring workgroups of one XCD (block b runs on XCD b % 8) take
  //      neighbouring units, so the cache lines two mode groups share meet in one L2
  const int nblk = g.n_pb * g.n_qb;
  const int chunks = (nblk + g.bpw - 1) / g.bpw;
  int gid = SC_BID_X;
  if ((g.G & 7) == 0) gid = (gid & 7) * (g.G >> 3) + (gid >> 3);
  const int mg = gid / chunks, ch = gid - mg * chunks;
  const int64_t m0 = (int64_t)mg * 8;

  // ---- loader role: lane = (segment s of the piece, slot tq); it fetches granule tq ^ f(s) of its segment
  const int ls = lane >> 2, ltq = lane & 3;
  const int lgr = ltq ^ ((ls >> 2) & 3);
  // ---- MFMA role: lane = (k half kk, row i) for A', (kk, column n = 2 qq + d) for B'
  const int kk = lane >> 5, li = lane & 31, d = lane & 1, qq = li >> 1;
  const int a_g = (kk * 32 + li) * 4 + (w ^ ((li >> 2) & 3));
  const int b_g = SC_G8_B_OFF + (kk * 64 + qq) * 4 + (w ^ ((qq >> 2) & 3));
  const uint32_t m_re = (CB && d) ? 0x80000000u : 0u;        // B'[(r,re)][(q,1)] = Im B  (conj: -Im B)
  const uint32_t m_im = (!CB && !d) ? 0x80000000u : 0u;      // B'[(r,im)][(q,0)] = -Im B (conj: +Im B)
  const int NS = (g.R + 1) >> 1;

  const int b_end = (ch + 1) * g.bpw < nblk ? (ch + 1) * g.bpw : nblk;
#pragma unroll 1
  for (int bi = ch * g.bpw; bi < b_end; ++bi) {
    const int pb = bi / g.n_qb, qb = bi - pb * g.n_qb;
    const int p0 = pb * 32, q0 = qb * 64;
    // this wave's pieces of every stage: A rows p0 + 16 (w & 1) + s of r0 + (w >> 1); B columns q0 + 16 w + s of
    // r0 and r0 + 1.  Rows / columns / r values past the end are clamped to the last one (finite duplicates:
    // never stored, or multiplied by zero below)
    int prow = p0 + (w & 1) * 16 + ls, qcol = q0 + w * 16 + ls;
    prow = prow < g.P ? prow : g.P - 1;
    qcol = qcol < g.Q ? qcol : g.Q - 1;
    const cf32* srcA = A + (int64_t)prow * g.a_sp + m0 + 2 * lgr;
    const cf32* srcB = B + (int64_t)qcol * g.b_sq + m0 + 2 * lgr;
    auto issue = [&](const int st, const int buf) {
      sc_f4* sb = lds + buf * SC_G8_STAGE_G;
      int ra = 2 * st + (w >> 1), r0 = 2 * st, r1 = 2 * st + 1;
      ra = ra < g.R ? ra : g.R - 1;
      r1 = r1 < g.R ? r1 : g.R - 1;
      SC_GLDS16(srcA + (int64_t)ra * g.a_sr, sb + ((w >> 1) * 32 + (w & 1) * 16) * 4);
      SC_GLDS16(srcB + (int64_t)r0 * g.b_sr, sb + SC_G8_B_OFF + (w * 16) * 4);
      SC_GLDS16(srcB + (int64_t)r1 * g.b_sr, sb + SC_G8_B_OFF + (64 + w * 16) * 4);
    };
    auto fetch = [&](const int buf, Gemm8Ops& o) {
      const sc_f4* sb = lds + buf * SC_G8_STAGE_G;
      o.a = sb[a_g];
#pragma unroll
      for (int u = 0; u < 4; ++u) o.b[u] = sb[b_g + u * 64];
    };

    sc_f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][u][v] = 0.f;

    // operands of one stage in MFMA order.  B': lane (q, d) needs Re / Im B for the (r, re) rows and -Im / Re B for
    // the (r, im) rows: a bitwise select on the lane's d (one v_bfi_b32 each; written as a ternary on vector
    // components the compiler builds a variable-index extract, three v_cndmask per value) and a sign mask
    struct Prep {
      float ar[2], ai[2], br[2][4], bm[2][4];
    };
    const uint32_t dsel = d ? 0xffffffffu : 0u;
    auto bsel = [&](const float hi, const float lo) { return sc_bitsel(dsel, hi, lo); };      // d ? hi : lo
    auto prep = [&](const Gemm8Ops& o, const int st, Prep& q) {
      const float keep = (2 * st + kk < g.R) ? 1.f : 0.f;     // odd R: the clamped duplicate contributes 0
      q.ar[0] = o.a.x * keep;
      q.ai[0] = (CA ? -o.a.y : o.a.y) * keep;
      q.ar[1] = o.a.z * keep;
      q.ai[1] = (CA ? -o.a.w : o.a.w) * keep;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        q.br[0][u] = sc_xor_sign(bsel(o.b[u].y, o.b[u].x), m_re);
        q.bm[0][u] = sc_xor_sign(bsel(o.b[u].x, o.b[u].y), m_im);
        q.br[1][u] = sc_xor_sign(bsel(o.b[u].w, o.b[u].z), m_re);
        q.bm[1][u] = sc_xor_sign(bsel(o.b[u].z, o.b[u].w), m_im);
      }
    };
    auto fire = [&](const Prep& q) {
#ifdef SC_G8_ABL_NOMFMA              // measurement builds only (scripts/gemm8_ab.py): operand stream without the matrix cores
#pragma unroll
      for (int u = 0; u < 4; ++u)
        asm volatile("" ::"v"(q.br[0][u]), "v"(q.br[1][u]), "v"(q.bm[0][u]), "v"(q.bm[1][u]), "v"(q.ar[0]), "v"(q.ai[1]));
      return;
#endif
      SC_SCHED_BARRIER();
      // consecutive MFMAs write different accumulators
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sc_mfma_32x32x2(acc[0][u], q.ar[0], q.br[0][u]);
        sc_mfma_32x32x2(acc[1][u], q.ar[1], q.br[1][u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sc_mfma_32x32x2(acc[0][u], q.ai[0], q.bm[0][u]);
        sc_mfma_32x32x2(acc[1][u], q.ai[1], q.bm[1][u]);
      }
      SC_SCHED_BARRIER();
    };
    auto pin = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#ifdef SC_G8_PIN   // A-B only: with exactly 128 accumulator registers pinned to AGPRs the allocator spills them
          SC_PIN_ACC(acc[j][u]);
#endif
        }
    };

    // ---- prologue: fill the ring, take stage 0
    sc_wait_vmcnt<0>();                 // stores of the previous tile's epilogue are counted by vmcnt too
    const int npro = NS < D ? NS : D;
    for (int st = 0; st < npro; ++st) issue(st, st);
    g8_wait_stage<D>(npro - 1);
    SC_BARRIER_RAW();
    Gemm8Ops o0, o1;
    fetch(0, o0);
    // one stage: publish stage st + 1 (counted wait + barrier), refill the buffer stage st lived in, fetch the
    // operands of stage st + 1 while the matrix cores work on stage st
    auto step = [&](const int st, const Gemm8Ops& cur, Gemm8Ops& nxt) {
      Prep q;
      if (st + 1 < NS) {
        g8_wait_stage<D>((NS - 2 - st) < (D - 2) ? (NS - 2 - st) : (D - 2));
        SC_WAIT_LGKM0();                // this wave's reads of stage st have landed in registers
        SC_BARRIER_RAW();               // ... so have everybody's: its buffer is free, stage st + 1 is complete
        if (st + D < NS) issue(st + D, st % D);
        prep(cur, st, q);               // before the next fetch: the compiler's own wait for `cur` must not
        SC_SCHED_BARRIER();             // also wait for the reads issued below
        fetch((st + 1) % D, nxt);
      } else {
        prep(cur, st, q);
      }
      fire(q);
      pin();
    };
#pragma unroll 1
    for (int st = 0; st < NS; st += 2) {
      step(st, o0, o1);
      if (st + 1 < NS) step(st + 1, o1, o0);
    }

    // ---- C: per 16-column tile the four waves fill a [32 rows][16 cols][4 slots] granule patch (each wave its
    //      mode pair: slot w ^ f(col)), then everybody stores 64-byte segments with 16-byte stores: store e of
    //      wave w covers row 8 w + e, lane = (column ls, slot ltq) exactly as in the loader role
    float* patch = reinterpret_cast<float*>(lds);
    SC_WAIT_LGKM0();
    SC_BARRIER_RAW();                   // nobody still reads operands from the ring
    const int pslot = ((w ^ ((qq >> 2) & 3)) * 4 + d) + qq * 16 + kk * (4 * 256);   // + row(v) * 256 + 2 j
    cf32* cl = C + (int64_t)sc_opaque(ls) * g.c_sq + m0 + 2 * lgr;                     // per-lane part
    const bool full = p0 + 32 <= g.P && q0 + 64 <= g.Q;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u) SC_BARRIER_RAW();          // the previous tile's patch has been read (reads feed stores: complete)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v)
          patch[pslot + ((v & 3) + 8 * (v >> 2)) * 256 + 2 * j] = acc[j][u][v];
      SC_WAIT_LGKM0();
      SC_BARRIER_RAW();
      sc_f4 val[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = lds[(w * 8 + e) * 64 + lane];
      cf32* ct = cl + ((int64_t)(p0 + w * 8) * g.c_sp + (int64_t)(q0 + u * 16) * g.c_sq);
#ifdef SC_G8_ABL_NOSTORE             // measurement builds only: everything but the C stores
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(val[e]));
      if (g.P < 0)
#endif
      if (full) {                       // uniform: whole tile inside C, eight back-to-back stores
#pragma unroll
        for (int e = 0; e < 8; ++e) sc_store16(ct + (int64_t)e * g.c_sp, val[e], g.stream_c);
      } else {
        const bool cok = q0 + u * 16 + ls < g.Q;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (cok && p0 + w * 8 + e < g.P) sc_store16(ct + (int64_t)e * g.c_sp, val[e], g.stream_c);
      }
    }
    SC_WAIT_LGKM0();
    SC_BARRIER_RAW();                   // the ring is refilled by the next tile
  }
}
