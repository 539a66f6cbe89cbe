"""Time the pointwise MLP backward pass (sc_pointwise_mlp_backward, metric shape) for several engine builds:
ablation variants from scripts/build_variants.py (-DSC_PMLP_ABL_*).  Usage: python scripts/pmlp_ablate.py lib.so ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
dev = torch.device("cuda:0")
B, C, Hd, S = 32, 64, 32, 256 * 256
torch.manual_seed(0)
x, sk, go = (torch.randn(B, C, S, device=dev) for _ in range(3))
w1, b1, w2, b2, gt = torch.randn(Hd, C, device=dev) / 8, torch.randn(Hd, device=dev), torch.randn(C, Hd, device=dev) / 6, torch.randn(C, device=dev), torch.randn(C, device=dev)
gx, gsk = torch.empty_like(x), torch.empty_like(x)
gw1, gw2, gb1, gb2, gg = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(b1), torch.empty_like(b2), torch.empty_like(gt)
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
for path in sys.argv[1:] or [_lib.DEFAULT_LIB]:
    lib = _lib.ScEngineLib(path)
    ws = torch.empty(lib.pointwise_mlp_workspace_bytes(B, C, Hd, C, S, 1), dtype=torch.uint8, device=dev)
    fw = lambda: lib.pointwise_mlp_forward(B, C, Hd, C, S, 1, p(x), p(w1), p(b1), p(w2), p(b2), p(sk), p(gt), p(out), st)
    bw = lambda: lib.pointwise_mlp_backward(B, C, Hd, C, S, 1, p(x), p(w1), p(b1), p(w2), p(b2), p(sk), p(gt), p(go), p(gx), p(gw1),
                                            p(gb1), p(gw2), p(gb2), p(gsk), p(gg), p(ws), st)
    res = []
    for fn in (fw, bw):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print(f"{os.path.basename(path):32s} forward {res[0]:.3f} ms   backward {res[1]:.3f} ms")
