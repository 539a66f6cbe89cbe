"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference SpectralConv hot path.

Two independent statements of the same maths (``fft_norm="forward"`` unless said
otherwise).  ``forward_torch`` also restates the separable, complex-data and
resolution-changing branches; ``forward_np64`` covers real data on an unchanged grid:

``forward_torch``   op-for-op torch restatement of
    /root/reference/neuralop/layers/spectral_convolution.py:417-570
    (rfftn -> fftshift -> centred slice -> einsum -> scatter into zeros ->
    ifftshift -> ifftn + irfft -> + bias).  Autograd supplies the backward the
    reference gets implicitly.  This is also what ``bench.py`` times as the
    ``cpu_baseline`` ("port").

``forward_np64`` / ``backward_np64``   numpy float64 "kept-rows" formulation
    (SURVEY.md section 8a "independent restatement"): no shifts, no full
    spectrum, explicit adjoint formulas.

Pinned against the verbatim reference (tests/test_oracle_vs_reference.py) and
the golden vectors under tests/golden/ generated from it.
"""
import numpy as np
import torch

_SYMS = "abcdefghijklmnopqrstuvwxyz"


# --------------------------------------------------------------------------
# mode bookkeeping  (spectral_convolution.py:400-415, 465-519)
# --------------------------------------------------------------------------
def halve_last(n_modes, complex_data=False):
    """n_modes setter rule, spectral_convolution.py:404-415."""
    n = [n_modes] if isinstance(n_modes, int) else list(n_modes)
    if not complex_data:
        n[-1] = n[-1] // 2 + 1
    return n


def weight_slices(spatial, n_modes_h, max_n_modes, complex_data=False):
    """Which sub-block of the stored weight is used and which signed frequencies
    it multiplies.  Follows spectral_convolution.py:465-519 (complex data: every dim is
    treated like a non-last dim, :475-479).

    spatial      : input spatial sizes
    n_modes_h    : the module's ``n_modes`` attribute (last entry already halved)
    max_n_modes  : the module's ``max_n_modes`` attribute (weight mode extents)

    Returns (w_slices, freqs): ``w_slices[d]`` is the python slice of weight
    mode-dim d; ``freqs[d]`` the signed frequency of every kept weight row
    (non-last dims: -k//2 ... ; last dim: 0 ... k-1).
    """
    nd = len(spatial)
    fft_size = list(spatial)
    if not complex_data:
        fft_size[-1] = fft_size[-1] // 2 + 1
    starts = [mx - min(sz, nm) for sz, nm, mx in zip(fft_size, n_modes_h, max_n_modes)]
    sl = []
    n_centred = nd if complex_data else nd - 1
    for d in range(n_centred):
        s = starts[d]
        sl.append(slice(s // 2, -s // 2) if s else slice(s, None))
    if not complex_data:
        sl.append(slice(None, -starts[-1]) if starts[-1] else slice(None))
    kept = [len(range(*s.indices(mx))) for s, mx in zip(sl, max_n_modes)]
    freqs = []
    for d in range(n_centred):
        k = kept[d]
        freqs.append(np.arange(-(k // 2), k // 2 + k % 2))
    if not complex_data:
        k = kept[-1]
        # spectral_convolution.py:514-517: last dim keeps columns [:k] (all if k >= fft_size)
        freqs.append(np.arange(min(k, fft_size[-1])))
    return sl, freqs


# --------------------------------------------------------------------------
# contractions (einsum strings of spectral_convolution.py:21-132)
# --------------------------------------------------------------------------
def contract_dense(x, w):
    """``bi...,io...->bo...``  (spectral_convolution.py:21-46)."""
    nd = x.ndim - 2
    m = _SYMS[3:3 + nd]
    return torch.einsum(f"ab{m},bc{m}->ac{m}", x, w)


def contract_dense_separable(x, w):
    """spectral_convolution.py:49-52."""
    return x * w


def reconstruct_tucker(core, factors):
    nd = len(factors)
    c = _SYMS[:nd]
    o = _SYMS[nd:2 * nd]
    res = core
    cur = c
    for d in range(nd):
        new = cur.replace(c[d], o[d])
        res = torch.einsum(f"{cur},{o[d]}{c[d]}->{new}", res, factors[d])
        cur = new
    return res


def reconstruct_cp(weights, factors):
    nd = len(factors)
    o = _SYMS[:nd]
    res = weights
    cur = "z"
    for d in range(nd):
        res = torch.einsum(f"{cur},{o[d]}z->{cur[:-1]}{o[d]}z", res, factors[d])
        cur = cur[:-1] + o[d] + "z"
    return res.sum(-1)


def reconstruct_tt(cores):
    """Dense tensor of a tensor-train with cores (r_k, s_k, r_{k+1})."""
    res = cores[0]
    for g in cores[1:]:
        res = torch.einsum("...a,abc->...bc", res, g)
    return res.squeeze(0).squeeze(-1)


def contract_tt(x, cores):
    """``abcd, r0 b r1, r1 e r2, r2 c r3, r3 d r4 -> aecd`` (spectral_convolution.py:106-132): cores over
    (in, out, modes...), contracted left to right after absorbing the in-channel core into x."""
    nd = x.ndim - 2
    m = _SYMS[7:7 + nd]
    z = torch.einsum(f"ab{m},pbq->aq{m}", x, cores[0])            # r0 = 1 summed away, q = r1
    z = torch.einsum(f"aq{m},qer->aer{m}", z, cores[1])           # out-channel core, r = r2
    for d in range(nd):                                           # mode cores: Hadamard over m[d]
        z = torch.einsum(f"aer{m},r{m[d]}s->aes{m}", z, cores[2 + d])
    return z.squeeze(2)


def contract_tucker(x, core, factors):
    """``abcd,fghi,bf,eg,ch,di->aecd`` (spectral_convolution.py:76-103) evaluated in the
    min-FLOP pairwise order of SURVEY.md section 8 row a6:
    T = U_modes . core ;  z = U_in^T x ;  m = sum_f T z ;  y = U_out m."""
    nd = x.ndim - 2
    m = _SYMS[7:7 + nd]                       # mode symbols (h, i, ...)
    r = _SYMS[12:12 + nd]                     # mode-rank symbols (m, n, ...)
    t = core                                   # (f, g, r...)
    cur = "fg" + r
    for d in range(nd):
        new = cur.replace(r[d], m[d])
        t = torch.einsum(f"{cur},{m[d]}{r[d]}->{new}", t, factors[2 + d])
        cur = new
    z = torch.einsum(f"ab{m},bf->af{m}", x, factors[0])
    mm = torch.einsum(f"af{m},fg{m}->ag{m}", z, t)
    return torch.einsum(f"ag{m},eg->ae{m}", mm, factors[1])


def contract_cp(x, weights, factors):
    """``abcd,r,br,er,cr,dr->aecd`` (spectral_convolution.py:55-73)."""
    nd = x.ndim - 2
    m = _SYMS[7:7 + nd]
    z = torch.einsum(f"ab{m},br->ar{m}", x, factors[0])          # project in-channels
    s = weights
    cur = "r"
    for d in range(nd):
        s = torch.einsum(f"{cur},{m[d]}r->{cur}{m[d]}", s, factors[2 + d])
        cur = cur + m[d]
    z = z * s                                                       # Hadamard over modes
    return torch.einsum(f"ar{m},er->ae{m}", z, factors[1])


# --------------------------------------------------------------------------
# forward, torch restatement of spectral_convolution.py:417-570
# --------------------------------------------------------------------------
def forward_torch(x, weight, bias, n_modes_h, max_n_modes=None, fft_norm="forward",
                  contract=contract_dense, enforce_hermitian_symmetry=True, separable=False,
                  output_shape=None, complex_data=False):
    """x: (B, Cin, *spatial) real (complex when ``complex_data``).  weight: dense
    (Cin, Cout, *max_n_modes) complex -- (C, *max_n_modes) when ``separable`` -- or a callable
    ``weight(mode_slices)`` returning whatever ``contract`` expects.
    n_modes_h: module ``n_modes`` (last already halved for real data).
    output_shape: spatial sizes of the result when they differ from the input's (the module's
    ``resolution_scaling_factor`` / ``output_shape``, :524-528)."""
    nd = x.ndim - 2
    spatial = list(x.shape[2:])
    if max_n_modes is None:
        max_n_modes = list(n_modes_h)
    fft_size = list(spatial)
    if not complex_data:
        fft_size[-1] = fft_size[-1] // 2 + 1
    fft_dims = list(range(-nd, 0))

    if complex_data:                                                          # :439-441
        xh = torch.fft.fftn(x, norm=fft_norm, dim=fft_dims)
        shift_dims = fft_dims
    else:
        xh = torch.fft.rfftn(x, norm=fft_norm, dim=fft_dims)                  # :443
        shift_dims = fft_dims[:-1]
    if nd > 1:
        xh = torch.fft.fftshift(xh, dim=shift_dims)                           # :446-449

    w_sl, freqs = weight_slices(spatial, n_modes_h, max_n_modes, complex_data)
    lead = (slice(None),) if separable else (slice(None), slice(None))         # :471-474
    wk = weight[lead + tuple(w_sl)] if torch.is_tensor(weight) else weight(tuple(w_sl))
    kept = [len(f) for f in freqs]
    if separable:
        cout = x.shape[1]
        contract = contract_dense_separable if contract is contract_dense else contract
    else:
        cout = wk.shape[1] if torch.is_tensor(wk) else wk.out_channels

    sl_x = [slice(None), slice(None)]
    for d in range(nd):                                                        # :502-512
        c = fft_size[d] // 2
        k = kept[d]
        sl_x.append(slice(c - k // 2, c + k // 2 + k % 2))
    # :514-517 -- the last slice is replaced by [:k] (real AND complex data: with complex data
    # the reference therefore multiplies the first k shifted columns, not the centred ones)
    sl_x[-1] = slice(None, kept[-1]) if kept[-1] < fft_size[-1] else slice(None)
    sl_x = tuple(sl_x)

    out_fft = torch.zeros([x.shape[0], cout, *fft_size], dtype=xh.dtype)       # :460-462
    out_fft[sl_x] = contract(xh[sl_x], wk)                                     # :520-522
    out_sizes = list(spatial) if output_shape is None else list(output_shape)  # :524-528
    if nd > 1:
        out_fft = torch.fft.ifftshift(out_fft, dim=fft_dims[:-1])             # :531-532 (never the last dim)
    if complex_data:                                                           # :536-538
        y = torch.fft.ifftn(out_fft, s=out_sizes, dim=fft_dims, norm=fft_norm)
    elif enforce_hermitian_symmetry:                                           # :547-559
        if nd > 1:
            out_fft = torch.fft.ifftn(out_fft, s=out_sizes[:-1], dim=fft_dims[:-1], norm=fft_norm)
        out_fft = out_fft.clone()
        out_fft[..., 0].imag.zero_()
        if out_sizes[-1] % 2 == 0:
            out_fft[..., -1].imag.zero_()
        y = torch.fft.irfft(out_fft, n=out_sizes[-1], dim=-1, norm=fft_norm)
    else:                                                                      # :564
        y = torch.fft.irfftn(out_fft, s=out_sizes, dim=fft_dims, norm=fft_norm)
    if bias is not None:
        y = y + bias                                                           # :567-568
    return y


# --------------------------------------------------------------------------
# fno_block_precision = "half" / "mixed" (spectral_convolution.py:436-459, einsum_utils.py:10-36)
# --------------------------------------------------------------------------
def contract_dense_chalf(x, w):
    """The dense contraction as ``einsum_complexhalf_two_input`` evaluates it (einsum_utils.py:10-36): both operands
    viewed as real and cast to float16, ONE real einsum with the two real/imaginary axes kept apart
    (tmp[x][y] = sum_i a_x b_y, a float16 result), re = tmp00 - tmp11, im = tmp10 + tmp01 in float16.
    Returns complex64 holding float16-representable values (complex32 has no FFT on the CPU)."""
    nd = x.ndim - 2
    m = "cdef"[:nd]
    a = torch.view_as_real(x.to(torch.complex64)).half()
    b = torch.view_as_real(w.to(torch.complex64)).half()
    tmp = torch.einsum(f"ab{m}x,bz{m}y->xyaz{m}", a, b)                          # :30-32
    res = torch.stack([tmp[0, 0] - tmp[1, 1], tmp[1, 0] + tmp[0, 1]], dim=-1)    # :33-35
    return torch.view_as_complex(res.float())


def forward_half_torch(x, weight, bias, n_modes_h, max_n_modes=None, fft_norm="forward", precision="mixed"):
    """Real data, unchanged grid, dense weight block.  The cast points of the reference -- x.half() for "half"
    (:436-437), x.chalf() before the contraction (:451-454, inside contract_dense_chalf here), the complex32
    ``out_fft`` (:455-462) whose inverse transform returns float16 -- with every value ROUNDED to float16 at that
    point and carried in fp32, and the two transforms evaluated in fp32: torch has no float16 FFT on the CPU, the
    reference's float16 FFT arithmetic is whatever the GPU backend does and cannot be pinned here."""
    assert precision in ("half", "mixed")
    if precision == "half":
        x = x.half().float()
    y = forward_torch(x, weight, None, n_modes_h, max_n_modes, fft_norm, contract=contract_dense_chalf)
    y = y.half().float()
    return y + bias if bias is not None else y.half()


# --------------------------------------------------------------------------
# numpy float64 "kept-rows" formulation
# --------------------------------------------------------------------------
def _kept_index(spatial, freqs):
    """numpy ix_ index of the kept block in the unshifted rfftn layout."""
    idx = [np.mod(f, n) for f, n in zip(freqs[:-1], spatial[:-1])]
    idx.append(freqs[-1])
    return idx


def _col_weights(spatial, freqs):
    """c_k of the C2R adjoint: 1 for DC (and Nyquist when W even), else 2."""
    w = spatial[-1]
    c = np.full(len(freqs[-1]), 2.0)
    c[freqs[-1] == 0] = 1.0
    if w % 2 == 0:
        c[freqs[-1] == w // 2] = 1.0
    return c


def _np_einsum_modes(eq_core, a, b, nd):
    m = _SYMS[6:6 + nd]
    return np.einsum(eq_core.replace("M", m), a, b)


def forward_np64(x, weight, bias, n_modes_h, max_n_modes=None):
    """Returns (y, xhat_kept).  All float64/complex128."""
    x = np.asarray(x, dtype=np.float64)
    weight = np.asarray(weight, dtype=np.complex128)
    nd = x.ndim - 2
    spatial = list(x.shape[2:])
    if max_n_modes is None:
        max_n_modes = list(n_modes_h)
    w_sl, freqs = weight_slices(spatial, n_modes_h, max_n_modes)
    wk = weight[(slice(None), slice(None)) + tuple(w_sl)]
    axes = tuple(range(2, 2 + nd))
    xh = np.fft.rfftn(x, axes=axes) / np.prod(spatial)
    idx = _kept_index(spatial, freqs)
    sel = np.ix_(np.arange(x.shape[0]), np.arange(x.shape[1]), *idx)
    xk = xh[sel]
    yk = _np_einsum_modes("abM,bcM->acM", xk, wk, nd)
    full = np.zeros((x.shape[0], wk.shape[1], *spatial[:-1], spatial[-1] // 2 + 1),
                    dtype=np.complex128)
    full[np.ix_(np.arange(x.shape[0]), np.arange(wk.shape[1]), *idx)] = yk
    # unscaled inverse; C2R ignores Im of the DC / Nyquist columns
    y = np.fft.irfftn(full, s=spatial, axes=axes) * np.prod(spatial)
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)
    return y, xk


def backward_np64(x, weight, g, n_modes_h, max_n_modes=None):
    """Explicit adjoints.  Returns (gx, gW_full, gbias);
    gW follows torch's complex convention (dL/dconj(w)), zero outside the used block."""
    x = np.asarray(x, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    weight = np.asarray(weight, dtype=np.complex128)
    nd = x.ndim - 2
    spatial = list(x.shape[2:])
    ntot = np.prod(spatial)
    if max_n_modes is None:
        max_n_modes = list(n_modes_h)
    w_sl, freqs = weight_slices(spatial, n_modes_h, max_n_modes)
    full_sl = (slice(None), slice(None)) + tuple(w_sl)
    wk = weight[full_sl]
    axes = tuple(range(2, 2 + nd))
    idx = _kept_index(spatial, freqs)
    cw = _col_weights(spatial, freqs)
    b, ci, co = x.shape[0], x.shape[1], wk.shape[1]
    xk = (np.fft.rfftn(x, axes=axes) / ntot)[np.ix_(np.arange(b), np.arange(ci), *idx)]
    gk = np.fft.rfftn(g, axes=axes)[np.ix_(np.arange(b), np.arange(co), *idx)] * cw
    gwk = _np_einsum_modes("abM,acM->bcM", np.conj(xk), gk, nd)
    gxk = _np_einsum_modes("acM,bcM->abM", gk, np.conj(wk), nd)
    gw = np.zeros_like(weight)
    gw[full_sl] = gwk
    # adjoint of the scaled R2C: (1/N) * Re sum_k g[k] e^{+i theta}; C2R doubles the
    # interior columns, so halve them first.
    full = np.zeros((b, ci, *spatial[:-1], spatial[-1] // 2 + 1), dtype=np.complex128)
    full[np.ix_(np.arange(b), np.arange(ci), *idx)] = gxk / cw
    gx = np.fft.irfftn(full, s=spatial, axes=axes)     # numpy's 1/N == the forward scaling
    gbias = g.sum(axis=(0,) + axes).reshape((co,) + (1,) * nd)
    return gx, gw, gbias


def rel_l2(a, b):
    """||a-b|| / ||b||  (LpLoss.rel, /root/reference/neuralop/losses/data_losses.py:168-203)."""
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
