"""Mode bookkeeping of the spectral layer (host logic).

Restates the index rules of /root/reference/neuralop/layers/spectral_convolution.py
:400-415 (n_modes setter) and :465-519 (which centred sub-block of the stored weight is
used, which FFT coefficients it multiplies) as *extents and offsets* -- the engine never
builds the shifted full spectrum, it only needs, per spatial dim d:

    kept[d]     number of modes kept (= extent of the used weight sub-block)
    w_start[d]  first used row of the stored weight (centred block)

The frequencies follow from kept[d]: non-last dims row r <-> r - kept//2 (even kept keeps
-k/2 .. k/2-1), last dim column c <-> c.
"""
from typing import List, Sequence, Tuple


def halve_last_mode(n_modes, complex_data: bool = False) -> List[int]:
    """``SpectralConv.n_modes`` setter rule (:400-415)."""
    n = [n_modes] if isinstance(n_modes, int) else list(n_modes)
    if not complex_data:
        n[-1] = n[-1] // 2 + 1
    return n


def kept_block(spatial: Sequence[int], n_modes_attr: Sequence[int],
               max_n_modes_attr: Sequence[int]) -> Tuple[List[int], List[int]]:
    """(kept, w_start) for real-valued data.

    n_modes_attr      the module attribute (last entry already halved)
    max_n_modes_attr  the module attribute = mode extents of the stored weight
    """
    nd = len(spatial)
    if not (len(n_modes_attr) == len(max_n_modes_attr) == nd):
        raise ValueError(
            f"expected {nd} mode entries, got n_modes={list(n_modes_attr)} "
            f"max_n_modes={list(max_n_modes_attr)}")
    fft_size = list(spatial)
    fft_size[-1] = fft_size[-1] // 2 + 1
    kept, w_start = [], []
    for d in range(nd):
        mx = int(max_n_modes_attr[d])
        start = mx - min(fft_size[d], int(n_modes_attr[d]))          # :465-468
        if start < 0:
            raise ValueError(f"n_modes[{d}]={n_modes_attr[d]} exceeds max_n_modes[{d}]={mx}")
        if d < nd - 1:
            # python slice(start//2, -start//2): note floor division of the NEGATIVE number
            lo = start // 2
            hi = mx + ((-start) // 2) if start else mx                  # :482-485
        else:
            lo = 0
            hi = mx - start                                             # :486
        k = hi - lo
        if k < 1:
            raise ValueError(f"no modes kept along dim {d}")
        kept.append(k)
        w_start.append(lo)
    # :514-517 -- the last dim never keeps more columns than the half spectrum has
    kept[-1] = min(kept[-1], fft_size[-1])
    return kept, w_start


# ------------------------------------------------------------------------------------------
# frequency maps (sc_plan_desc.freq): which FFT index of a grid each kept row reads / is written to
# ------------------------------------------------------------------------------------------
def kept_block_complex(spatial: Sequence[int], n_modes_attr: Sequence[int],
                       max_n_modes_attr: Sequence[int]) -> Tuple[List[int], List[int]]:
    """(kept, w_start) for complex_data=True: every dim follows the centred rule (:475-479)."""
    kept, w_start = [], []
    for d, n in enumerate(spatial):
        mx = int(max_n_modes_attr[d])
        start = mx - min(int(n), int(n_modes_attr[d]))
        if start < 0:
            raise ValueError(f"n_modes[{d}]={n_modes_attr[d]} exceeds max_n_modes[{d}]={mx}")
        lo = start // 2
        hi = mx + ((-start) // 2) if start else mx
        if hi - lo < 1:
            raise ValueError(f"no modes kept along dim {d}")
        kept.append(hi - lo)
        w_start.append(lo)
    return kept, w_start


def analysis_freqs(spatial: Sequence[int], kept: Sequence[int], complex_data: bool = False):
    """FFT index on the INPUT grid each kept row multiplies; None = the engine's default map.

    Real data: rows r - k//2 (non-last dims), columns c (last dim) -- the default.
    Complex data: the reference fft-shifts every dim but then takes ``[:k]`` of the last one
    (:514-517 is applied to complex data as well), i.e. the k most NEGATIVE frequencies:
    shifted column c is FFT index (c - n//2) mod n.  Reproduced as is.  (1-d: the reference skips the
    shift altogether, :446, so column c is plain FFT index c.)"""
    if not complex_data:
        return None
    nd = len(spatial)
    freq = [None] * nd
    n, k = int(spatial[-1]), int(kept[-1])
    freq[-1] = [(c - n // 2) % n for c in range(k)] if nd > 1 else list(range(k))
    return freq


def synthesis_freqs(in_spatial: Sequence[int], out_spatial: Sequence[int], kept: Sequence[int],
                    complex_data: bool = False):
    """(freq, real_col): where each kept row lands on the OUTPUT grid of the inverse transform.

    The reference scatters the block into a spectrum of the INPUT grid's shape, un-shifts the
    non-last dims there, and only then pads / truncates every dim AT THE END to the output size
    (``ifftn(s=...)`` / ``irfft(n=...)``, :524-559).  A row therefore keeps its input-grid FFT index
    (negative frequencies do NOT move to the top of a larger grid) and is dropped when that index
    does not exist on the output grid.  Complex data: the last dim is never un-shifted (:531-532), so
    column c simply lands at index c.  Real data: the reference zeroes Im of the input grid's last
    half-spectrum column when the output width is even (:552-556) -> ``real_col``."""
    nd = len(in_spatial)
    same = [int(a) for a in in_spatial] == [int(b) for b in out_spatial]
    if same and not complex_data:
        return None, 0
    freq = []
    for d in range(nd):
        n_in, n_out, k = int(in_spatial[d]), int(out_spatial[d]), int(kept[d])
        last = d == nd - 1
        if last and not complex_data:
            cols = n_out // 2 + 1
            freq.append([c if c < cols else None for c in range(k)])
        elif last:
            freq.append([c if c < n_out else None for c in range(k)])
        else:
            idx = [(r - k // 2) % n_in for r in range(k)]
            freq.append([i if i < n_out else None for i in idx])
    real_col = 0
    if not complex_data and int(out_spatial[-1]) % 2 == 0:
        c_q = int(in_spatial[-1]) // 2
        if c_q < int(kept[-1]):
            real_col = c_q
    return freq, real_col


def resample_block(in_spatial: Sequence[int], out_spatial: Sequence[int]):
    """Spectral skip-path resample of resample.py:54-66 (3-d and up) as (kept, analysis map, synthesis
    map): per non-last dim the m = min(n_in, n_out) rows -ceil(m/2) .. floor(m/2)-1 (python's
    ``slice(-m//2, None)`` takes ceil(m/2) negative rows), last dim the first min of the two
    half-spectrum widths; each row goes to the SAME signed frequency of the new grid."""
    nd = len(in_spatial)
    kept, fa, fs = [], [], []
    for d in range(nd):
        n_in, n_out = int(in_spatial[d]), int(out_spatial[d])
        if d == nd - 1:
            m = min(n_in // 2 + 1, n_out // 2 + 1)
            kept.append(m)
            fa.append(None)
            fs.append(None)
        else:
            m = min(n_in, n_out)
            neg = m - m // 2
            signed = [r - neg for r in range(m)]
            kept.append(m)
            fa.append([f % n_in for f in signed])
            fs.append([f % n_out for f in signed])
    return kept, fa, fs
