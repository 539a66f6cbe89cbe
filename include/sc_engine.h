/*
 * sc_engine.h -- C-ABI of the MI355X-native SpectralConv engine (libsc_engine.so).
 *
 * The reference (neuraloperator v2.0.0) is 100 % Python: its hot path is a chain of ATen
 * calls inside neuralop/layers/spectral_convolution.py and there is no FFI boundary
 * upstream.  This header is the boundary a maintainer would bind
 * with ctypes from a `conv_module` drop-in; every entry point cites
 * the reference lines it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  All data pointers are DEVICE pointers
 *     owned by the caller (PyTorch's allocator); the plan owns only its twiddle tables.
 *   - real tensors: float32 (bfloat16 storage with SC_PLAN_IO_BF16), contiguous (B, C, d1..dN).
 *     complex tensors: interleaved (re, im) float32 pairs (== torch.complex64 memory).
 *   - truncated spectra are stored (B, C, k1..kN) with the mode dims in WEIGHT order:
 *     non-last dim row r  <->  signed frequency r - floor(k/2) (fftshift order, even k
 *     keeps -k/2 .. k/2-1), last dim column c <-> frequency c.
 *     (spectral_convolution.py:465-519.)
 *   - every function returns 0 on success, non-zero on error (sc_last_error() gives the
 *     message); no C++ exception crosses the ABI.
 *   - all work is enqueued on the hipStream_t passed in (void* here so that the header is
 *     usable without HIP headers); nothing synchronises the device.
 *   - a plan is immutable after creation and may be used from several streams as long as
 *     each call gets its own workspace.
 */
#ifndef SC_ENGINE_H
#define SC_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_MAX_DIMS 4

/* fft_norm of the reference module (spectral_convolution.py:303, 443, 548, 559) */
enum { SC_NORM_FORWARD = 0, SC_NORM_BACKWARD = 1, SC_NORM_ORTHO = 2 };

/* which linear map a transform call applies (all four share kernels, only tables differ) */
enum {
  SC_FWD_SCALED = 0,   /* rfftn(norm) restricted to the kept modes      (:443-449, 500-519) */
  SC_FWD_ADJ_C2R = 1,  /* adjoint of the zero-padded inverse: unscaled R2C, interior
                          last-dim columns x2  (autograd of :531-559)           */
  SC_INV_PADDED = 0,   /* zero-pad + ifftn + irfft(norm), DC/Nyquist imag ignored (:520-559)*/
  SC_INV_ADJ_R2C = 1   /* adjoint of SC_FWD_SCALED (autograd of :443)           */
};

typedef struct sc_plan sc_plan;

typedef struct {
  int32_t ndim;                  /* number of spatial dims, 1..SC_MAX_DIMS                  */
  int32_t fft_norm;              /* SC_NORM_*                                               */
  int64_t spatial[SC_MAX_DIMS];  /* d1..dN                                                  */
  int64_t kept[SC_MAX_DIMS];     /* extents of the used weight sub-block = kept modes       */
  int32_t flags;                 /* SC_PLAN_* bits                                          */
  int32_t real_col;              /* last-dim column whose imaginary part the zero-padded inverse
                                    ignores besides DC / Nyquist: the reference zeroes Im of the
                                    INPUT grid's last half-spectrum column before an irfft onto a
                                    different grid (:552-556).  0 = none (DC is always ignored)  */
  /* Optional frequency maps (host arrays, read during sc_plan_create only; NULL = default):
   * freq[d][r], r < kept[d], is the FFT index on THIS plan's grid (0 .. spatial[d]-1, negative
   * values taken mod spatial[d]) that kept row r of dim d reads / is written to, or
   * SC_FREQ_DROPPED when the row falls off the grid.  Defaults: non-last dims r - floor(k/2),
   * last dim r (real data) -- the same-grid rule of :502-517.  The host passes explicit maps
   * for the reference's resolution-changing inverse (rows stay at their INPUT-grid FFT index,
   * :524-559), its complex-data branch and the skip-path resample (resample.py:54-66); see
   * neuraloperator_amd/modes.py.  A plan with maps never takes the fused power-of-two kernels. */
  const int64_t* freq[SC_MAX_DIMS];
} sc_plan_desc;
#define SC_FREQ_DROPPED INT64_MIN

enum {
  SC_PLAN_FORCE_GENERIC = 1, /* never take the power-of-two fast kernels (debug / A-B)     */
  SC_PLAN_FFT_GEN2 = 2,      /* fast path on the generation-2 fused kernels (A-B)           */
  SC_PLAN_NO_MDFT = 4,       /* generic passes on the VALU kernels instead of the matrix cores (A-B) */
  SC_PLAN_COMPLEX = 8,       /* complex_data=True (:439-441, 536-538): x and y are complex (n_images,
                                d1..dN), every dim is a complex-to-complex pass, bias must be NULL
                                (the host adds the real bias); transforms only, no sc_layer_*      */
  SC_PLAN_NO_F2P_SMALL = 32, /* keep grids of 64 .. 640 points per axis (32 P, P in {2,3,4,5,6,8,10,12,20}) on the
                                direct-DFT passes instead of the two-pass factorised route (round 3; A-B)  */
  SC_PLAN_F2P_SMALL_ALWAYS = 64, /* take the two-pass route for supported grids below 128 x 128 points too (A-B, tests:
                                by default those stay on the one-launch direct-DFT plane passes, which are faster there) */
  SC_PLAN_NO_SPAN = 128,     /* complex -> real last-axis pass of widths off every factorised route: the 128-line chunked
                                kernel (k_mdft_c2r_stage) instead of the 32-line whole-span one (k_mdft_c2r_span; A-B, tests) */
  SC_PLAN_NO_MX_FFT = 256,   /* bfloat16 I/O: the transforms on the vector-ALU kernels k_fft2d_fwd3 / k_fft2d_inv3 instead of
                                the matrix-core row passes k_fft2d_fwd_mx / k_fft2d_inv_mx (H <= 256; round 5; A-B and tests) */
  SC_PLAN_MX_FFT_3TERM = 512, /* bfloat16 I/O: the twiddles of k_fft2d_fwd_mx as THREE bf16 terms (24 bits: the spectrum of
                                fp32 round-off class, 1.2e-7 from a float64 transform of the same values) instead of the
                                default two (16 bits: 1.3e-6 -- 3000 x below the 2^-9 the bf16 input carries per value, 8 x
                                below the 1e-5 bar of the fp32 gradients; 10 us per launch faster at the metric shape; round 6) */
  SC_PLAN_IO_BF16 = 16       /* the REAL tensors (x, y, gy, gx) are bfloat16 in memory -- the `float*`
                                arguments that carry them then point at 2-byte elements; spectra,
                                weights, bias and every arithmetic step stay float32 and y / gx are
                                rounded to nearest even on the store (BASELINE configs[1] "bf16": the
                                reference has no bf16 spectral path, torch.fft rejects the dtype, so this
                                is y = bf16(layer(fp32(x)))).  Fused 2-D kernels only: sc_plan_create
                                fails for shapes that take the size-agnostic passes (the host converts) */
};

/* ---- plan ------------------------------------------------------------------------------ */
/* replaces the per-call index math of spectral_convolution.py:429-434, 465-519 */
int sc_plan_create(sc_plan** out, const sc_plan_desc* desc);
void sc_plan_destroy(sc_plan* plan);
/* bytes of scratch a transform call over `n_images` (= batch*channels) images needs */
size_t sc_plan_workspace_bytes(const sc_plan* plan, int64_t n_images);
/* 1 if the plan runs the fused power-of-two kernels, 0 for the generic pruned-DFT passes */
int sc_plan_is_fast(const sc_plan* plan);

/* ---- transforms -------------------------------------------------------------------------- */
/* x: real (n_images, d1..dN) [complex with SC_PLAN_COMPLEX]  ->  xhat: complex (n_images, k1..kN)
 * replaces rfftn + fftshift + x[slices_x]      (spectral_convolution.py:443-449, 500-519) */
int sc_transform_forward(const sc_plan* plan, int mode, const float* x, float* xhat,
                         int64_t n_images, void* workspace, void* stream);

/* yhat: complex (n_images, k1..kN) -> y: real (n_images, d1..dN), y += bias[image % channels]
 * replaces zeros + out_fft[slices_x]= + ifftshift + ifftn + imag.zero_ + irfft + bias
 * (spectral_convolution.py:456-462, 520, 531-568).  bias may be NULL. */
int sc_transform_inverse(const sc_plan* plan, int mode, const float* yhat, const float* bias,
                         int64_t channels, float* y, int64_t n_images, void* workspace,
                         void* stream);

/* ---- mode-batched complex GEMM ---------------------------------------------------------------
 * C[p, q, m] (+)= sum_r opA(A[p, r, m]) * opB(B[r, q, m]),   m = 0..n_modes-1 on the lanes.
 * Element (complex) offsets: A: p*a_sp + r*a_sr + m*a_sm;  B: r*b_sr + q*b_sq + off_b(m);
 * C: p*c_sp + q*c_sq + off_c(m), where off(m) = idx[m] if the index table is given
 * (device int32, lets B / C be a strided sub-block of the stored weight), else m*sm.
 * a_sm / b_sm may be 0 (operand shared by all modes: Tucker/CP factor matrices).
 * replaces tl.einsum('bixy,ioxy->boxy') and its two autograd einsums
 * (spectral_convolution.py:21-46) and the pairwise steps of _contract_tucker/_contract_cp
 * (:55-103). */
enum {
  SC_GEMM_FORCE_VALU = 1,    /* never take the matrix-core kernel (debug / A-B)            */
  SC_GEMM_STREAM_C = 2,      /* C is not consumed by the next kernel: non-temporal stores    */
  SC_GEMM_PAIRED = 4,        /* P = 32: two 4-wave workgroups per CU, 5 modes each (A-B; slower from HBM) */
  SC_GEMM_WIDE = 8,          /* 9 modes per workgroup even for a small mode count (A-B / tests)          */
  SC_GEMM_NO_STREAM = 16,    /* never take the LDS-DMA streamed kernel (k_modegemm_s8): generation 1 (A-B) */
  SC_GEMM_NO_SB = 64,        /* never take the small-extent streaming kernel (k_modegemm_sb; A-B / tests)  */
  SC_GEMM_SB_WM4 = 128,      /* k_modegemm_sb: the four waves over 512 contiguous modes of one tile (A-B / tests) */
  SC_GEMM_SB_ALT_ORDER = 1 << 25, /* small-batch kernels: the OTHER work-item order (round 5; A-B / tests).  Default: k_modegemm_sb
                                walks its mode tiles slowest (the tiles of a mode tile share the small operand in one L2), the
                                one-pass backward pair k_modegemm_sb_bwd mode tiles FASTEST (neighbouring workgroups stream
                                neighbouring pieces of the same weight rows: 5.78 -> 5.54 ms per step at configs[4]) */
  SC_GEMM_NO_FMX = 1 << 24,  /* never take the matrix-core factor-matrix / mode-sum kernels (sc_kernels_fmx.h; A-B / tests) */
  SC_GEMM_F16 = 32           /* the reference's complex-half contraction (fno_block_precision "half" / "mixed",
                              * einsum_utils.py:10-36): operands rounded to float16, four real products summed in
                              * fp32 and rounded to float16, re = t00 - t11, im = t10 + t01 rounded to float16;
                              * fp32 storage throughout.  Plain C = opA(A) opB(B) launches (no accumulate) */
};
/* flags bits 8..23: cap on the number of workgroups of the matrix-core kernel (0 = auto) */
#define SC_GEMM_GRID(n) (((n) & 0xffff) << 8)

typedef struct {
  int64_t P, Q, R, n_modes;
  int64_t a_sp, a_sr, a_sm;
  int64_t b_sr, b_sq, b_sm;
  int64_t c_sp, c_sq, c_sm;
  int32_t conj_a, conj_b;
  int32_t accumulate;       /* 0: C = ..., 1: C += ...                                     */
  int32_t flags;            /* SC_GEMM_* bits                                               */
  const int32_t* b_idx;     /* optional device table [n_modes], NULL -> m*b_sm              */
  const int32_t* c_idx;     /* optional device table [n_modes], NULL -> m*c_sm              */
  /* tiled ("mode-group-major") operands, engine-private layouts only: with a_sg != 0 mode m of A lives at element
   * offset (m / 16) * a_sg + (m % 16) (+ p*a_sp + r*a_sr), i.e. the 16 modes of a group stay contiguous and the
   * groups are a_sg apart -- e.g. [group][p][r][16] with a_sg = P*R*16, a_sp = R*16, a_sr = 16 makes everything one
   * workgroup of k_modegemm_dma reads one contiguous block.  Same for b_sg / c_sg.  0 = plain (m * sm).  Needs
   * n_modes % 16 == 0, sm == 1 and no index table; only k_modegemm_dma takes it (sc_modegemm fails otherwise). */
  int64_t a_sg, b_sg, c_sg;
} sc_modegemm_desc;

int sc_modegemm(const sc_modegemm_desc* d, const float* A, const float* B, float* C,
                void* stream);
/* Two independent contractions: sc_modegemm(d0, ...) followed by sc_modegemm(d1, ...) (no output may alias an operand
 * of the other call) -- bit-identical to those two calls whenever each of them alone runs the 4-wave streamed shape or
 * the pair runs as two launches, equal up to the rounding of the last bits otherwise (a job that alone would take the
 * 8-wave four-product shape, e.g. the weight gradient at hidden 128, runs the 4-wave three-product shape here).  The pair of a layer's backward pass -- d0 the weight gradient
 * (conj_a, autograd of spectral_convolution.py:21-46 w.r.t. the weight), d1 the gradient of the spectrum (conj_b) --
 * runs as ONE launch of k_modegemm_dma_bwd when both qualify for the streamed matrix-core kernel: the second round of
 * one job fills the tail of the other.  sc_modegemm_pair_fused: 1 if a pair with 16-byte aligned operands takes that
 * launch, 0 if it runs as two.
 * Round 3, session 2: a SMALL-BATCH pair (d0->R = d1->P <= 4 rows against weight-sized d0's C / d1's B, B0 and A1 the
 * same array; BASELINE configs[4]) runs as ONE launch of k_modegemm_sb_bwd (sc_kernels_sb.h): a single pass over the
 * weight serves both gradients (the weight is read while its gradient is written), bit-identical to the two
 * k_modegemm_sb launches it replaces; sc_layer_backward takes the same launch. */
int sc_modegemm_pair(const sc_modegemm_desc* d0, const float* A0, const float* B0, float* C0,
                     const sc_modegemm_desc* d1, const float* A1, const float* B1, float* C1, void* stream);
int sc_modegemm_pair_fused(const sc_modegemm_desc* d0, const sc_modegemm_desc* d1);
/* Which launch(es) sc_modegemm_pair issues for 16-byte aligned operands with B0 == A1 (introspection for tests and
 * profiles, like sc_modegemm_path): 2 = one launch of k_modegemm_sb_bwd, 1 = one launch of k_modegemm_dma_bwd,
 * 0 = two sc_modegemm launches. */
int sc_modegemm_pair_path(const sc_modegemm_desc* d0, const sc_modegemm_desc* d1);
/* C[p, q] += sum_m sum_r opA(A[p, r, m]) * opB(B[r, q, m]) -- the gradient of a mode-independent
 * operand (Tucker / CP factor matrices, autograd of spectral_convolution.py:55-103): lanes run over
 * the modes, wave reduction, one atomic add per (p, q) and mode tile.  C (element offsets
 * p*c_sp + q*c_sq) must be zeroed by the caller; c_sm / c_idx / accumulate are ignored. */
int sc_modegemm_msum(const sc_modegemm_desc* d, const float* A, const float* B, float* C,
                     void* stream);
/* The same sum, C[p, q] OVERWRITTEN (no zeroing by the caller), on the matrix cores (round 3, sc_kernels_fmx.h:
 * k_modegemm_msum_mx stages [rows][64 modes] chunks of both operands in LDS and accumulates 16 x 16 tiles over all
 * chunks of a workgroup; one partial per workgroup in `workspace`, fixed-order reduction: run-to-run deterministic,
 * unlike the atomic adds of sc_modegemm_msum).  The matrix-core kernel takes problems where both operands have mode
 * stride 1, 8 <= P, Q <= 64 and n_modes >= 64 (sc_modegemm_msum_path == 1); every other problem runs the VALU kernel
 * of sc_modegemm_msum with one workspace slot per (mode split, r split) instead of atomics and the same fixed-order
 * reduction (path 0; session 2) -- so the result of this entry point is the same bits on every run for every shape.
 * sc_modegemm_msum_workspace_bytes returns the bytes needed (0 only for an empty problem or one beyond the launch grid). */
size_t sc_modegemm_msum_workspace_bytes(const sc_modegemm_desc* d);
int sc_modegemm_msum_ws(const sc_modegemm_desc* d, const float* A, const float* B, float* C, void* workspace,
                        size_t workspace_bytes, void* stream);
/* 1: sc_modegemm_msum_ws runs k_modegemm_msum_mx (matrix cores), 0: the slot form of the VALU kernel */
int sc_modegemm_msum_path(const sc_modegemm_desc* d);
/* 1 if this call runs on a matrix-core kernel, 0 for the VALU kernel */
int sc_modegemm_uses_matrix_cores(const sc_modegemm_desc* d);
/* which kernel a call with 16-byte aligned operands takes: 0 k_modegemm (VALU), 1 k_modegemm_mfma (register-staged
 * matrix-core kernel: sub-blocks through index tables, ragged ranks, factor operands), 2 k_modegemm_s8 (LDS-DMA
 * streamed matrix-core kernel: plain contiguous-mode operands, mode count a multiple of 8), 3 k_modegemm_sb (round 3:
 * small-extent streaming kernel on the vector ALUs -- a batch of <= 4 rows against a large weight, or a reduction of
 * <= 4 terms into a weight-sized result: plain contiguous-mode operands, even mode count, even row strides;
 * bit-identical to path 0), 4 k_modegemm_bfac (round 3: b_sm == 0, i.e. a mode-independent right operand -- the channel
 * factor matrices of Tucker / CP contractions -- read through the scalar cache; unit mode strides of A and C, Q >= 8;
 * bit-identical to path 0), 5 k_modegemm_bfac_mx (round 3: the same operand shape with 8 <= Q <= 64, 4 <= R <= 64 and
 * >= 64 modes on the matrix cores: chunks of 64 modes staged in LDS, 16 x 16 x 4 MFMA tiles, three real products per
 * complex product -- equal to path 0 up to the rounding of the last bits) */
int sc_modegemm_path(const sc_modegemm_desc* d);

/* out[i] = float16(in[i]) (round to nearest even), kept in fp32 storage; in == out allowed.  The cast points of
 * fno_block_precision "half" / "mixed" (spectral_convolution.py:436-437 x.half(), :451-454 x.chalf(), and the
 * float16 result of the inverse transform). */
int sc_round_f16(const float* in, float* out, int64_t n, void* stream);

/* gbias[c] = sum_b Re(ghat[b, c, dc]) -- the bias gradient read off the DC coefficient of
 * the already-computed SC_FWD_ADJ_C2R spectrum (autograd of :567-568). */
int sc_bias_grad(const sc_plan* plan, const float* ghat, int64_t batch, int64_t channels,
                 float* gbias, void* stream);

/* ---- batch-independent part of the 2-D Tucker contraction (TFNO) in one launch each way --------------------
 *   t[fg, x, y] = sum_{c, d} core[fg, c, d] ux[x, c] uy[y, d]
 * (_contract_tucker, neuralop/layers/spectral_convolution.py:76-103: the two mode factors absorbed into the core; fg =
 * the (in-rank, out-rank) pairs).  All arrays complex64 interleaved, contiguous: core (fg, rx, ry), ux (mx, rx),
 * uy (my, ry), t (fg, mx, my).  Backward: gcore / gux / guy overwritten, workspace sc_tucker_modes_workspace_bytes(d).
 * Limits: mx, my, rx, ry <= 64, my ry <= 1024 (sc_tucker_modes_supported; callers fall back to sc_modegemm). */
typedef struct sc_tucker_desc {
  int64_t fg, rx, ry, mx, my;
} sc_tucker_desc;
int sc_tucker_modes_supported(const sc_tucker_desc* d);
int sc_tucker_modes_forward(const sc_tucker_desc* d, const float* core, const float* ux, const float* uy, float* t,
                            void* stream);
size_t sc_tucker_modes_workspace_bytes(const sc_tucker_desc* d);
int sc_tucker_modes_backward(const sc_tucker_desc* d, const float* core, const float* ux, const float* uy,
                             const float* gt, float* gcore, float* gux, float* guy, void* workspace, void* stream);

/* ---- activation side of the factorized Tucker contraction, ONE call each way (round 3, session 2) -----------------
 *   z[b, f, m] = sum_i xhat[b, i, m] u_in[i, f];   t[b, g, m] = sum_f z[b, f, m] t3[f, g, m];
 *   yhat[b, o, m] = sum_g t[b, g, m] u_out[o, g]
 * -- the pairwise order of _contract_tucker's einsum 'abcd,fghi,bf,eg,ch,di->aecd' (neuralop/layers/
 * spectral_convolution.py:76-103; t3 = the core with the mode factors absorbed, sc_tucker_modes_forward) and the six
 * products of its autograd.  The launches are exactly those of the corresponding sc_modegemm / sc_modegemm_msum_ws
 * calls (same kernels, same bits); the point is the HOST: a factorized layer step issued call by call from Python
 * costs 0.6-0.76 ms of host time for 0.76 ms of device time (profiles/r03s2_tfno_host.txt), one call per direction
 * takes the 9 products off the interpreter.  All arrays complex64 interleaved, contiguous: xhat (B, Cin, M),
 * u_in (Cin, R1), t3 (R1, R2, M), u_out (Cout, R2), z (B, R1, M), t (B, R2, M), yhat (B, Cout, M).
 * Backward: gy (B, Cout, M) in; gxhat (B, Cin, M), gu_in (Cin, R1), gt3 (R1, R2, M), gu_out (Cout, R2) overwritten
 * (a null pointer skips that gradient); workspace: sc_tucker_chain_workspace_bytes (holds gt, gz and the partial
 * sums of the two factor gradients).
 * The backward call uses TWO streams (session 2): gt, gt3, gxhat on `stream`, the factor gradients and gz on a
 * non-blocking stream the engine owns, forked from and joined back into `stream` with events (no host
 * synchronisation; records into a hipGraph as a fork / join): everything the call issued is complete, in `stream`'s
 * order, when work issued to `stream` afterwards runs.  The launches are latency-bound and only partly dependent; side
 * by side the six take 146 instead of 193 us under the profiler, 0.735 -> 0.721 ms per TFNO step
 * (profiles/r03s2_tfno_two_streams_ab.txt, _timeline.txt).  Same kernels, same bits.  SC_NO_SIDE_STREAM=1
 * (environment) keeps every launch on `stream`. */
typedef struct sc_tucker_chain_desc {
  int64_t batch, c_in, c_out, r_in, r_out, n_modes;
} sc_tucker_chain_desc;
int sc_tucker_chain_forward(const sc_tucker_chain_desc* d, const float* xhat, const float* u_in, const float* t3,
                            const float* u_out, float* z, float* t, float* yhat, void* stream);
size_t sc_tucker_chain_workspace_bytes(const sc_tucker_chain_desc* d);
int sc_tucker_chain_backward(const sc_tucker_chain_desc* d, const float* xhat, const float* u_in, const float* t3,
                             const float* u_out, const float* z, const float* t, const float* gy, float* gxhat,
                             float* gu_in, float* gt3, float* gu_out, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ---- the same nine products as ONE kernel launch each way (round 5, csrc/sc_kernels_tkchain.h) -----------------------
 * Replaces the three einsum steps of _contract_tucker's pairwise order and their autograd
 * (neuralop/layers/spectral_convolution.py:76-103) for shapes inside sc_tucker_chain_fused_supported: batch <= 32,
 * channels <= 64, ranks <= 48, every extent and the number of modes a multiple of 4 (BASELINE configs[2]: 32 x 64 -> 64
 * channels, ranks 36, 2112 modes).  A workgroup owns four modes for the whole batch and walks x̂ -> z -> t -> ŷ
 * (backward: ĝ -> gt -> gz -> gx̂ with the factor gradients accumulated per workgroup and reduced in fixed order) out of
 * LDS; z and t are written once for the backward pass and never re-read by the forward call.
 *   t3m: a mode-major copy of t3, (n_modes, r_in, r_out) complex64 = sc_tucker_chain_t3m_bytes -- WRITTEN by the forward
 *        call (one transposing launch) and READ by the backward call: the caller keeps it between the two.
 *   backward workspace: sc_tucker_chain_backward_fused_workspace_bytes (the mode-major gradient of t3 + one partial
 *        sum of both factor gradients per workgroup).  gt3 is required; gxhat / gu_in / gu_out may be null.
 * Same arithmetic as the nine launches (exact-fp32 16 x 16 x 4 matrix tiles, three real products per complex product);
 * the factor gradients are summed in a different (fixed) order.  All tensors 16-byte aligned.
 * OPT-IN: sc_tucker_chain_fused_supported returns 0 unless SC_TKC=1 is set in the environment -- measured on MI355X
 * at configs[2] the fused kernels are as accurate as the nine launches and slower (forward 104-114 us against 68.6 us,
 * backward 182-195 us against 139 us; profiles/r05_tkchain_ab.txt), so callers keep the nine
 * launches by default. */
int sc_tucker_chain_fused_supported(const sc_tucker_chain_desc* d);
size_t sc_tucker_chain_t3m_bytes(const sc_tucker_chain_desc* d);
int sc_tucker_chain_forward_fused(const sc_tucker_chain_desc* d, const float* xhat, const float* u_in, const float* t3,
                                  const float* u_out, float* t3m, float* z, float* t, float* yhat, void* stream);
size_t sc_tucker_chain_backward_fused_workspace_bytes(const sc_tucker_chain_desc* d);
int sc_tucker_chain_backward_fused(const sc_tucker_chain_desc* d, const float* xhat, const float* u_in, const float* t3m,
                                   const float* u_out, const float* z, const float* t, const float* gy, float* gxhat,
                                   float* gu_in, float* gt3, float* gu_out, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* ---- peer-store exchange of the mode-parallel layer (round 5, csrc/sc_kernels_peer.h; OPT-IN) ---------------------------
 * The all-to-all "split modes, cat batch" of the mode-sharded layer (the shape contract of neuralop/mpu/helpers.py:81-99,
 * which the reference defines and never calls) as direct stores into the peers' memory: every rank of ONE node owns a
 * window -- fine-grained device memory, sc_peer_window_alloc -- whose 64-byte HIP IPC handle the host side hands to the
 * other ranks (any transport), which map it with sc_peer_window_open.  sc_peer_all_to_all then issues three plain kernel
 * launches on `stream`: block p of `send` ([world][block_bytes]) is stored into slot `rank` of peer p's window and this
 * rank's epoch is written to slot `rank` of every peer's flags with system-scope release; a one-workgroup launch waits until
 * all `world` flags of THIS rank carry the epoch and the third copies the window into `recv` ([world][block_bytes], block p from
 * rank p).  No host synchronisation, capturable into a hipGraph (the epoch lives in the window's header and advances
 * per call).  Every rank must call it for every exchange, in the same order; a window may serve the next exchange of
 * the same kind one layer step later (the layer's exchanges alternate directions -- csrc/sc_kernels_peer.h), the host
 * side rotates a few windows.  peer_window[p] = the mapped base of rank p's window (own entry: the local pointer);
 * world <= 8, block_bytes a multiple of 16, send / recv 16-byte aligned.  Unmeasured on more than one GPU (no multi-GPU
 * tier in the build environment); self-tested with two processes on one device.
 * sc_peer_window_control (round 6): the wait launch spins without bound by default -- a peer that is merely late must not
 * corrupt a step -- which turns a dead peer or a non-coherent mapping into a stream that never drains.  A rank may give
 * the waits on ITS window a budget (milliseconds of the device's 100 MHz wall clock; 0 = unbounded, negative = unchanged)
 * and read the error word a timed-out wait leaves behind (0 = none, 1 + p = the flag of peer p never came; reading
 * clears it).  The host side runs its set-up self-test under a 2 s budget and falls back to the collective path on error. */
typedef struct sc_peer_exchange {
  int32_t world, rank;
  int64_t block_bytes;
  void* peer_window[8];
} sc_peer_exchange;
int sc_peer_window_alloc(size_t data_bytes, void** ptr, void* handle64);
int sc_peer_window_open(const void* handle64, void** ptr);
int sc_peer_window_close(void* ptr);
int sc_peer_window_free(void* ptr);
int sc_peer_all_to_all(const sc_peer_exchange* d, const void* send, void* recv, void* stream);
int sc_peer_window_control(void* own_window, int64_t spin_budget_ms, int32_t* error_out);

/* ---- pointwise half of an FNO block in one pass ("next" row f1 of SURVEY.md section 8) -------------------
 *   out = act( W2 gelu(W1 x + b1) + b2 + gate (.) skip_src )
 * replaces ChannelMLP.forward (neuralop/layers/channel_mlp.py:82-119: two Conv1d with kernel size 1 and a GELU),
 * the soft-gating skip (neuralop/layers/skip_connections.py:53-130: per-channel weight times the block input) and
 * the closing non-linearity of FNOBlocks.forward_with_postactivation (neuralop/layers/fno_block.py:399-412).
 * x (batch, c_in, spatial), skip_src / out (batch, c_out, spatial) contiguous float32; w1 (c_hid, c_in), w2 (c_out,
 * c_hid) row-major (= Conv1d weights with the trailing 1 dropped); b1, b2, skip_src + gate optional (null).
 * Channel counts: multiples of 32 with (c_in, c_hid, c_out) / 32 in {(1,1,1), (2,1,2), (2,2,2), (4,2,4)};
 * spatial a multiple of 32 below 2^28 points per image (all sc_pointwise_* entry points: lane offsets are 32-bit byte counts). */
typedef struct sc_pmlp_desc {
  int64_t batch, c_in, c_hid, c_out, spatial;
  int32_t act;               /* SC_ACT_NONE / SC_ACT_GELU on the result */
  int32_t reserved;
} sc_pmlp_desc;
int sc_pointwise_mlp_forward(const sc_pmlp_desc* d, const float* x, const float* w1, const float* b1, const float* w2,
                             const float* b2, const float* skip_src, const float* gate, float* out, void* stream);
/* The whole pointwise side of a default block's forward in ONE pass (session 2):
 *     s = conv + (ws x + bs);   y = act(s);   out = act( W2 gelu(W1 y + b1) + b2 + gate (.) x )
 * conv = the spectral convolution's output (sc_layer_forward without an epilogue), ws / bs = the block's 1 x 1 linear
 * skip (neuralop/layers/skip_connections.py:119-169), x = the block input; y and -- with SC_ACT_GELU -- s (`pre`) are
 * STORED because the backward passes (sc_pointwise_mlp_backward_ex with x = y, x_pre = s, then
 * sc_pointwise_linear_backward and sc_layer_backward_ex) read them.  Replaces sc_pointwise_linear_forward + the block
 * epilogue of sc_layer_forward_ex + sc_pointwise_mlp_forward (8 tensor-sized passes) by 1 + 5: the skip is never
 * written and y is not read back.  d->c_in == d->c_out in {32, 64} channels with c_hid as for the MLP pass; bs, b1, b2
 * optional; `pre` is required with SC_ACT_GELU (receives s) or SC_ACT_GELU_DGRAD (receives gelu'(s), see the enum) and ignored otherwise. */
int sc_pointwise_block_forward(const sc_pmlp_desc* d, const float* conv, const float* x, const float* ws, const float* bs,
                               const float* w1, const float* b1, const float* w2, const float* b2, const float* gate,
                               float* y, float* pre, float* out, void* stream);
/* gradient of the above with respect to everything: nothing but the forward's INPUTS is needed (h and the
 * pre-activations are recomputed inside the tile).  gx (batch, c_in, spatial), gskip_src (batch, c_out, spatial; with a
 * gate), gw1 / gw2 like w1 / w2, gb1 / gb2 / ggate (null when the forward had none) are all overwritten.  workspace:
 * sc_pointwise_mlp_workspace_bytes(d) bytes (operand tables + one partial sum per workgroup; the partials are added
 * in a fixed order; inside a workgroup the waves add with LDS atomics).  Shapes: (32,32,32), (64,32,64), (64,64,64) --
 * (128,64,128) has the forward pass only. */
size_t sc_pointwise_mlp_workspace_bytes(const sc_pmlp_desc* d);
int sc_pointwise_mlp_backward(const sc_pmlp_desc* d, const float* x, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* skip_src, const float* gate, const float* gout, float* gx,
                              float* gw1, float* gb1, float* gw2, float* gb2, float* gskip_src, float* ggate,
                              void* workspace, void* stream);
/* the same with x_pre (optional, like x): x = gelu(x_pre) came out of sc_layer_forward_ex; gx is then the gradient with
 * respect to x_pre -- the GELU backward of the Fourier layer in this pass' store path instead of a pass of its own.
 * d->act == SC_ACT_GELU_DGRAD: the MLP's closing activation is GELU and x_pre holds gelu'(pre-activation) (what
 * sc_pointwise_block_forward stored under the same code). */
int sc_pointwise_mlp_backward_ex(const sc_pmlp_desc* d, const float* x, const float* x_pre, const float* w1,
                                 const float* b1, const float* w2, const float* b2, const float* skip_src,
                                 const float* gate, const float* gout, float* gx, float* gw1, float* gb1, float* gw2,
                                 float* gb2, float* gskip_src, float* ggate, void* workspace, void* stream);

/* Round 6 -- the backward of sc_pointwise_block_forward's pointwise side with the DATA path of the linear skip riding along
 * (fno_block.py:392-414 backwards: closing GELU, soft-gating skip, ChannelMLP, the Fourier layer's GELU, and the linear skip's
 * W_s^T product): one pass reads y, y_pre, x, gout and writes
 *     gz  = the gradient of the Fourier layer's pre-activation s (= the gradient of the spectral convolution's output and
 *           of the linear skip's output),
 *     gin = W_s^T gz + gate (.) g_z  = the whole gradient of the block input outside the spectral convolution (the addend
 *           of sc_layer_backward_ex),
 * and the gradients of w1 / b1 / w2 / b2 / gate -- instead of sc_pointwise_mlp_backward_ex + sc_pointwise_linear_backward
 * (10 tensor-sized reads / writes -> 5; the skip's own weight gradient is then sc_pointwise_linear_backward_ex with
 * gx = NULL: 2 more).  d->act: SC_ACT_NONE (last block: y_pre NULL), SC_ACT_GELU (y_pre = s) or SC_ACT_GELU_DGRAD
 * (y_pre = gelu'(s), as stored by the forward call under the same code).  (channels, hidden) in {(32, 32), (64, 32)}
 * (sc_pointwise_block_backward_supported); workspace sc_pointwise_mlp_workspace_bytes(d) bytes. */
int sc_pointwise_block_backward_supported(const sc_pmlp_desc* d);
int sc_pointwise_block_backward(const sc_pmlp_desc* d, const float* y, const float* y_pre, const float* x,
                                const float* ws_lin, const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* gate, const float* gout, float* gz, float* gin, float* gw1, float* gb1,
                                float* gw2, float* gb2, float* ggate, void* workspace, void* stream);

/* 1 x 1 linear map over the channels in one pass each way: out = W x (+ bias) -- the block's linear skip
 * (neuralop/layers/skip_connections.py:119-169: Flattened1dConv = Conv1d with kernel size 1 on the flattened grid).
 * x (batch, c_in, spatial), out (batch, c_out, spatial), w (c_out, c_in) row-major, bias (c_out) or null.
 * c_in = c_out in {32, 64, 128} forward, {32, 64} backward; spatial a multiple of 32.  Backward: gx, gw (and gbias
 * when not null) are overwritten; gx_addend (optional, like gx): gx = W^T gout + gx_addend -- the gradient another
 * branch of the block sends to the same input, added in the store path instead of by a separate pass; workspace
 * sc_pointwise_linear_workspace_bytes(d) bytes. */
typedef struct sc_plin_desc {
  int64_t batch, c_in, c_out, spatial;
} sc_plin_desc;
int sc_pointwise_linear_forward(const sc_plin_desc* d, const float* x, const float* w, const float* bias, float* out,
                                void* stream);
size_t sc_pointwise_linear_workspace_bytes(const sc_plin_desc* d);
int sc_pointwise_linear_backward(const sc_plin_desc* d, const float* x, const float* w, const float* gout,
                                 const float* gx_addend, float* gx, float* gw, float* gbias, void* workspace,
                                 void* stream);

/* ---- the same maps with the block's pointwise operations in their load / store paths (round 6, csrc/sc_kernels_plinx.h) ----
 * Channel counts 32 / 64 / 128 in any combination.  A ChannelMLP whose channel counts have no one-pass kernel (hidden 128,
 * configs[4]'s width: channel_mlp.py:82-119) is two of these passes each way, the hidden activations crossing memory once
 * as their pre-activation; the soft-gating skip (skip_connections.py:53-130), the GELUs (fno_block.py:392-414) and their
 * derivatives ride in the passes, so nothing elementwise is left for the caller:
 *   forward   out = act(W xin + bias + gate (.) skip);  xin = gelu(x) with SC_PLX_XACT (x = the previous pass' pre_out);
 *             act = gelu with SC_PLX_ACT; pre_out (optional) receives the value before act
 *   backward  g = gout (.) gelu'(pre) with SC_PLX_PRO;  gskip = gate (.) g, ggate = sum g (.) skip when gated;
 *             gx = (W^T g + gx_addend) (.) gelu'(xg) (addend optional; the last factor with SC_PLX_XGRAD: for an XACT pass
 *             xg = x and gx is the gradient of the pre-activation);  gw = g xin^T;  gbias = sum g (null = not wanted).
 *             gx == NULL: the weight / bias / gate gradients only.
 * The backward call issues ceil(C_out / n) kernel launches (n output tiles of weight-gradient accumulators fit the
 * register file) + fixed-order reductions; workspace sc_pointwise_linear_workspace_bytes_ex(d) bytes. */
#define SC_PLX_XACT 1
#define SC_PLX_ACT 2
#define SC_PLX_PRO 4
#define SC_PLX_XGRAD 8
typedef struct sc_plinx_desc {
  int64_t batch, c_in, c_out, spatial;
  int32_t flags;
} sc_plinx_desc;
int sc_pointwise_linear_forward_ex(const sc_plinx_desc* d, const float* x, const float* w, const float* bias,
                                   const float* skip, const float* gate, float* out, float* pre_out, void* stream);
size_t sc_pointwise_linear_workspace_bytes_ex(const sc_plinx_desc* d);
int sc_pointwise_linear_backward_ex(const sc_plinx_desc* d, const float* x, const float* w, const float* gout,
                                    const float* pre, const float* xg, const float* skip, const float* gate,
                                    const float* gx_addend, float* gx, float* gw, float* gbias, float* gskip,
                                    float* ggate, void* workspace, void* stream);

/* ---- fused AdamW step of the spectral weights ("next" row f2 of SURVEY.md section 8) -----------------
 * One pass over (param, grad, exp_avg, exp_avg_sq) instead of the ~10 elementwise launches of
 * neuralop/training/adamw.py:155-200 (non-GaLore branch), same arithmetic in the same order:
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g conj(g);  p -= step_size m / (sqrt(v) + eps);
 *   p -= lr weight_decay p        with step_size = lr sqrt(1-b2^t) / (1-b1^t) when correct_bias.
 * is_complex: the four arrays hold n complex64 values (exp_avg_sq keeps the reference's complex layout,
 * imaginary part 0); otherwise n floats.  `step` is the 1-based step count AFTER the increment. */
typedef struct {
  double lr, beta1, beta2, eps, weight_decay;   /* python floats: 1 - beta is formed in double like upstream */
  int64_t step;
  int32_t correct_bias;
  int32_t reserved;
} sc_adamw_desc;
int sc_adamw_step(const sc_adamw_desc* d, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t n, int is_complex, void* stream);

/* ---- fused dense layer (the reference module's forward / implicit backward) -------------------
 * w: complex (cin, cout, w_extent...) stored weight; the used sub-block starts at w_start[d]
 * in every mode dim and has the plan's `kept` extents (centred block of :465-489).
 * xhat_saved: complex (batch, cin, k...) written by forward, read by backward.
 * workspace: sc_layer_workspace_bytes(plan, batch, max(cin, cout)) bytes. */
typedef struct {
  int32_t batch, cin, cout, reserved;
  int64_t w_extent[SC_MAX_DIMS];
  int64_t w_start[SC_MAX_DIMS];
} sc_layer_desc;

size_t sc_layer_workspace_bytes(const sc_plan* plan, const sc_layer_desc* L);

int sc_layer_forward(const sc_plan* plan, const sc_layer_desc* L, const float* x,
                     const float* w, const float* bias, float* y, float* xhat_saved,
                     void* workspace, void* stream);

/* gw must be zero-initialised by the caller when the used sub-block is smaller than the
 * stored weight (only the sub-block is written).  gbias may be NULL. */
int sc_layer_backward(const sc_plan* plan, const sc_layer_desc* L, const float* gy,
                      const float* xhat_saved, const float* w, float* gx, float* gw,
                      float* gbias, void* workspace, void* stream);
/* sc_layer_backward with gx = (gradient through the layer) + gx_addend (optional; same shape as gx): the gradient the
 * other branches of an FNO block send to the layer's input, added in the store path of the last transform. */
int sc_layer_backward_ex(const sc_plan* plan, const sc_layer_desc* layer, const float* gy,
                         const float* xhat_saved, const float* w, float* gx, float* gw, float* gbias,
                         const float* gx_addend, void* workspace, void* stream);

/* ---- block epilogue (first "next" row f1 of SURVEY.md section 8) ---------------------------------------------------
 * The FNO block computes act(conv(x) + skip(x)) around the spectral convolution (neuralop/layers/fno_block.py:392-414:
 * x_fno + x_skip_fno, then the non-linearity) -- three more passes over tensors of the activation's size.  With an
 * epilogue the inverse transform adds `skip` (same shape and storage type as y) and applies the activation while it
 * stores:  y = act(irfft(...) + bias + skip).  preact (optional, same shape) receives the value before the
 * activation, which the backward of the activation needs.  On the fused 2-D kernels this happens in the store path of
 * the inverse transform (one extra read of the activation's size); other shapes run one streaming pass after it. */
/* GELU: torch.nn.functional.gelu's default form 0.5 v (1 + erf(v / sqrt 2)), with erf / erfc evaluated by the
 * 5-term rational approximation Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute in erf in exact arithmetic; the negative tail
 * is taken from erfc directly, so it does not cancel: relative error < 1 % down to v = -6) -- the library erff made
 * the fused inverse transform 5 x slower.  Absolute error of the activation <= 0.5 |v| 7e-7 incl. fp32 round-off
 * (rel-L2 ~1e-7 on N(0,1) data; pinned in tests/test_emu_epilogue.py for |v| <= 6).  The fused store path and the stand-alone
 * pass evaluate the SAME function (identical bits for every grid shape).
 * preact: written if and only if act == SC_ACT_GELU; preact != NULL together with SC_ACT_NONE is rejected. */
enum { SC_ACT_NONE = 0, SC_ACT_GELU = 1,
       /* round 6, sc_pointwise_block_forward / sc_pointwise_mlp_backward_ex only: as SC_ACT_GELU, but the `pre` buffer of the
          forward call receives gelu'(s) -- the DERIVATIVE of the Fourier layer's activation, which comes out of the same
          evaluation of the two transcendentals as gelu(s) itself -- and `x_pre` of the backward call holds that
          derivative: the backward multiplies by it instead of evaluating gelu' per value (on MI355X the vector work of
          these fp32 kernels runs in front of their matrix instructions, not beside them: DESIGN 3.16 b) */
       SC_ACT_GELU_DGRAD = 2 };
typedef struct {
  const float* skip;        /* device, (n_images, d1..dN); NULL = no epilogue at all          */
  float* preact;            /* device, optional                                                 */
  int32_t act;              /* SC_ACT_*                                                         */
  int32_t reserved;
} sc_epilogue;

int sc_transform_inverse_ex(const sc_plan* plan, int mode, const float* yhat, const float* bias, int64_t channels,
                            const sc_epilogue* ep, float* y, int64_t n_images, void* workspace, void* stream);
int sc_layer_forward_ex(const sc_plan* plan, const sc_layer_desc* L, const float* x, const float* w,
                        const float* bias, const sc_epilogue* ep, float* y, float* xhat_saved, void* workspace,
                        void* stream);

/* ---- sharded spectra (round 3; mode-parallel layers) -------------------------------------------
 * The reference defines the exchange of a mode-sharded layer only as a shape contract (mpu/helpers.py:81-99, the
 * unused `_transpose`: split dim0, all-to-all, concatenate dim1).  Here the transforms on either side of that
 * all-to-all write / read its rank-major buffer IN PLACE, so that no permutation copy is left around the collective:
 * the kept block of every image is cut along its FIRST mode dim into n_blocks blocks of `rows` rows
 * (n_blocks * rows >= k1; rows past k1 are zeros on the wire -- round 5: also whole blocks, e.g. k1 = 6 over 4 ranks =
 * three blocks of 2 rows and an empty one) and block p of image i lives at
 *     xhat + p * block_stride + (i * rows + r) * rest + j        (complex elements; rest = k2 * .. * kN)
 * i.e. xhat is the tensor [n_blocks][n_images][rows][rest] (block_stride >= n_images * rows * rest).  The fused 2-D
 * kernels address this layout natively; every other shape runs the plain transform through a staging copy in the
 * workspace plus ONE streaming permutation -- same results, sc_plan_workspace_bytes_sharded() bytes of workspace. */
typedef struct {
  int64_t n_blocks;      /* ranks of the model-parallel group                               */
  int64_t rows;          /* rows of the first kept dim per block: ceil(k1 / n_blocks)       */
  int64_t block_stride;  /* complex elements from one block to the next                     */
} sc_spectrum_shards;
size_t sc_plan_workspace_bytes_sharded(const sc_plan* plan, int64_t n_images);
/* as sc_transform_forward / sc_transform_inverse (same modes), spectrum in the sharded layout */
int sc_transform_forward_sharded(const sc_plan* plan, int mode, const float* x, float* xhat, int64_t n_images,
                                 const sc_spectrum_shards* shards, void* workspace, void* stream);
int sc_transform_inverse_sharded(const sc_plan* plan, int mode, const float* yhat, const float* bias,
                                 int64_t channels, float* y, int64_t n_images, const sc_spectrum_shards* shards,
                                 void* workspace, void* stream);
/* sc_bias_grad on a sharded adjoint spectrum (batch * channels images) */
int sc_bias_grad_sharded(const sc_plan* plan, const float* ghat, int64_t batch, int64_t channels,
                         const sc_spectrum_shards* shards, float* gbias, void* stream);

/* ---- misc ------------------------------------------------------------------------------------ */
const char* sc_last_error(void);
const char* sc_version(void);
/* name of the dominant kernel a transform over this plan launches (for profile matching) */
const char* sc_plan_kernel_name(const sc_plan* plan, int which);

#ifdef __cplusplus
}
#endif
#endif /* SC_ENGINE_H */
