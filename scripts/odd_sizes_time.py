"""Sizes the factorised routes do not cover: the reference's documented Darcy grids (85 / 141 / 211 /
421 points per axis, doc/source/theory_guide/fno.rst:384-392) and a resolution-changing layer -- engine layer step
(forward + backward through the drop-in module) against the reference's own op chain on the same GPU (hipFFT + ATen
einsum via oracle.forward_torch, which is spectral_convolution.py:417-570 op for op), same shapes, fp32.
Usage: python scripts/odd_sizes_time.py > profiles/r04_odd_sizes.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv  # noqa: E402
from neuraloperator_amd.modes import halve_last_mode  # noqa: E402
from oracle import spectral_oracle as so  # noqa: E402

dev = torch.device("cuda:0")
CASES = [  # B, C, spatial, n_modes, output_shape
    (32, 32, (85, 85), (32, 32), None),
    (32, 32, (141, 141), (32, 32), None),
    (32, 32, (141, 141), (64, 64), None),
    (32, 32, (211, 211), (32, 32), None),
    (32, 32, (211, 211), (64, 64), None),
    (16, 32, (421, 421), (32, 32), None),
    (16, 32, (421, 421), (64, 64), None),
    (32, 64, (128, 128), (32, 32), (256, 256)),       # resolution-changing layer (super-resolution decoder block)
    (32, 64, (256, 256), (64, 64), (128, 128)),       # ... and the coarsening direction
    (32, 64, (256, 256), (64, 64), None),             # the metric shape for scale
    # round 5: lines above 1024 points stay on the pruned direct DFT (matrix cores) -- measured beside
    # the reference's O(N log N) chain
    (2, 16, (2048, 2048), (64, 64), None),
    (2, 16, (2048, 2048), (256, 256), None),
]


def timeit(step, n=20, reps=3):
    """best of `reps` runs of n steps after 0.3 s of the same step (clocks, allocator, first-call work); small cases are
    host-issue bound and vary by box and by what ran before -- the best run is the repeatable number"""
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:                # settle the clocks
        step()
    best = float("inf")
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


if os.environ.get("ODD_CASES"):                        # e.g. ODD_CASES=2,4,6: a subset by index
    CASES = [CASES[int(i)] for i in os.environ["ODD_CASES"].split(",")]
NO_REF = os.environ.get("ODD_NO_REF") == "1"
print(f"{'B':>3} {'C':>3} {'grid':>10} {'modes':>8} {'out':>10} | {'engine ms':>9} {'reference-chain ms':>18} {'speed-up':>8} | engine GB/s (alg) | err y")
for B, C, spatial, n_modes, out_shape in CASES:
    torch.manual_seed(0)
    conv = SpectralConv(C, C, n_modes).to(dev)
    nm = halve_last_mode(n_modes)
    x = torch.randn(B, C, *spatial, device=dev, requires_grad=True)
    osh = list(out_shape) if out_shape else list(spatial)
    g = torch.randn(B, C, *osh, device=dev)

    def eng():
        x.grad = None
        for q in conv.parameters():
            q.grad = None
        (conv(x, output_shape=out_shape) if out_shape else conv(x)).backward(g)

    w = conv.weight.tensor.detach().clone().requires_grad_(True)
    bias = conv.bias.detach().clone().requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)

    def ref():
        xr.grad = w.grad = bias.grad = None
        with torch.device(dev):
            so.forward_torch(xr, w, bias, nm, nm, output_shape=out_shape).backward(g)

    with torch.no_grad(), torch.device(dev):
        ye = conv(x, output_shape=out_shape) if out_shape else conv(x)
        yr = so.forward_torch(xr, w, bias, nm, nm, output_shape=out_shape)
        err = float((ye - yr).norm() / yr.norm())
    te, tr = timeit(eng), (float('nan') if NO_REF else timeit(ref))
    kept = [min(m, s) for m, s in zip(nm, spatial)]
    nin = B * C * spatial[0] * spatial[1] * 4
    nout = B * C * osh[0] * osh[1] * 4
    S = 8 * B * C * kept[0] * kept[1]
    Wb = 8 * C * C * kept[0] * kept[1]
    alg = 2 * nin + 2 * nout + 3 * Wb + 9 * S
    print(f"{B:3d} {C:3d} {'x'.join(map(str, spatial)):>10} {'x'.join(map(str, n_modes)):>8} "
          f"{'x'.join(map(str, osh)) if out_shape else '-':>10} | {te:9.3f} {tr:18.3f} {tr / te:8.1f} | {alg / te / 1e6:8.0f}          | {err:.1e}")
    del conv, x, g, w, bias, xr
    torch.cuda.empty_cache()
