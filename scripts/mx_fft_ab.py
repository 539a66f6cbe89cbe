"""A-B of the forward-type transform on bfloat16 tensors at the metric shape: k_fft2d_fwd_mx (row pass on the matrix
cores, round 5) against k_fft2d_fwd3<256, sc_bf16> (vector ALUs; SC_PLAN_NO_MX_FFT), same process, same data.
Prints the time per launch (20 launches back to back between one pair of events, best of 5) and the rel-L2 error of
each kernel's kept block against a float64 rfft2 of the same bf16 values.
Usage: python scripts/mx_fft_ab.py [H] [B] [C]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lib = _lib.get_lib()
dev = torch.device("cuda:0")
torch.manual_seed(3)
Mx, My = min(64, H), 33
x = torch.randn(B, C, H, 256, device=dev).bfloat16()
st = torch.cuda.current_stream().cuda_stream
ref = torch.fft.rfft2(x[:2].double(), norm="forward")
ref = torch.cat([ref[..., H - Mx // 2:, :My], ref[..., :Mx // 2, :My]], dim=-2)
out = {}
for tag, fl in (("mx", _lib.SC_PLAN_IO_BF16), ("valu", _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT)):
    plan = lib.plan_create([H, 256], [Mx, My], flags=fl)
    xh = torch.zeros(B, C, Mx, My, 2, device=dev)
    f = lambda: lib.transform_forward(plan, 0, x.data_ptr(), xh.data_ptr(), B * C, 0, st)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    got = torch.view_as_complex(xh[:2].double().contiguous())
    err = float((got - ref).norm() / ref.norm())
    out[tag] = xh.clone()
    mb = (x.numel() * 2 + xh.numel() * 4) / 1e6
    print(f"{tag:5s} {lib.plan_kernel_name(plan, 0):16s} {best * 1e3:7.1f} us per launch  {mb / best / 1e3:6.2f} TB/s  rel-L2 vs float64 {err:.2e}")
    lib.plan_destroy(plan)
d = float((out["mx"] - out["valu"]).norm() / out["valu"].norm())
print(f"mx vs valu rel-L2 {d:.2e}")
