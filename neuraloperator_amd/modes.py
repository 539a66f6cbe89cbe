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
