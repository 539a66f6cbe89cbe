"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference SpectralConv hot path.

Two independent statements of the same maths (real-valued data, no resolution
change, ``fft_norm="forward"`` unless said otherwise):

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


def weight_slices(spatial, n_modes_h, max_n_modes):
    """Which sub-block of the stored weight is used and which signed frequencies
    it multiplies.  Follows spectral_convolution.py:465-519 (real data).

    spatial      : input spatial sizes
    n_modes_h    : the module's ``n_modes`` attribute (last entry already halved)
    max_n_modes  : the module's ``max_n_modes`` attribute (weight mode extents)

    Returns (w_slices, freqs): ``w_slices[d]`` is the python slice of weight
    mode-dim d; ``freqs[d]`` the signed frequency of every kept weight row
    (non-last dims: -k//2 ... ; last dim: 0 ... k-1).
    """
    nd = len(spatial)
    fft_size = list(spatial)
    fft_size[-1] = fft_size[-1] // 2 + 1
    starts = [mx - min(sz, nm) for sz, nm, mx in zip(fft_size, n_modes_h, max_n_modes)]
    sl = []
    for d in range(nd - 1):
        s = starts[d]
        sl.append(slice(s // 2, -s // 2) if s else slice(s, None))
    sl.append(slice(None, -starts[-1]) if starts[-1] else slice(None))
    kept = [len(range(*s.indices(mx))) for s, mx in zip(sl, max_n_modes)]
    freqs = []
    for d in range(nd - 1):
        k = kept[d]
        freqs.append(np.arange(-(k // 2), k // 2 + k % 2))
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
                  contract=contract_dense, enforce_hermitian_symmetry=True):
    """x: (B, Cin, *spatial) real.  weight: dense (Cin, Cout, *max_n_modes) complex, or
    whatever ``contract`` expects after slicing through ``weight_fn``.
    n_modes_h: module ``n_modes`` (last already halved)."""
    nd = x.ndim - 2
    spatial = list(x.shape[2:])
    if max_n_modes is None:
        max_n_modes = list(n_modes_h)
    fft_size = list(spatial)
    fft_size[-1] = fft_size[-1] // 2 + 1
    fft_dims = list(range(-nd, 0))

    xh = torch.fft.rfftn(x, norm=fft_norm, dim=fft_dims)                      # :443
    if nd > 1:
        xh = torch.fft.fftshift(xh, dim=fft_dims[:-1])                        # :446-449

    w_sl, freqs = weight_slices(spatial, n_modes_h, max_n_modes)
    wk = weight[(slice(None), slice(None)) + tuple(w_sl)] if torch.is_tensor(weight) \
        else weight(tuple(w_sl))
    kept = [len(f) for f in freqs]
    cout = wk.shape[1] if torch.is_tensor(wk) else wk.out_channels

    sl_x = [slice(None), slice(None)]
    for d in range(nd - 1):                                                    # :502-512
        c = fft_size[d] // 2
        k = kept[d]
        sl_x.append(slice(c - k // 2, c + k // 2 + k % 2))
    sl_x.append(slice(None, kept[-1]))                                         # :514-517
    sl_x = tuple(sl_x)

    out_fft = torch.zeros([x.shape[0], cout, *fft_size], dtype=xh.dtype)       # :460-462
    out_fft[sl_x] = contract(xh[sl_x], wk)                                     # :520-522
    if nd > 1:
        out_fft = torch.fft.ifftshift(out_fft, dim=fft_dims[:-1])             # :531-532
    if enforce_hermitian_symmetry:                                             # :547-559
        if nd > 1:
            out_fft = torch.fft.ifftn(out_fft, s=spatial[:-1], dim=fft_dims[:-1], norm=fft_norm)
        out_fft = out_fft.clone()
        out_fft[..., 0].imag.zero_()
        if spatial[-1] % 2 == 0:
            out_fft[..., -1].imag.zero_()
        y = torch.fft.irfft(out_fft, n=spatial[-1], dim=-1, norm=fft_norm)
    else:                                                                      # :564
        y = torch.fft.irfftn(out_fft, s=spatial, dim=fft_dims, norm=fft_norm)
    if bias is not None:
        y = y + bias                                                           # :567-568
    return y


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
