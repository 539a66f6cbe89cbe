"""A-B of the inverse-type transform writing bfloat16 tensors at the metric shape: k_fft2d_inv_mx (row pass on the matrix
cores, round 5 session 2) against k_fft2d_inv3<256, sc_bf16> (vector ALUs; SC_PLAN_NO_MX_FFT), same process, same data.
Prints the time per launch (20 launches back to back between one pair of events, best of 5), the rel-L2 error against a
float64 irfft2 of the same spectrum and the fraction of outputs on which the two kernels differ.
Usage: python scripts/mx_ifft_ab.py [H] [B] [C]"""
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
yh = torch.randn(B, C, Mx, My, 2, device=dev)
bias = torch.randn(C, device=dev)
st = torch.cuda.current_stream().cuda_stream
full = torch.zeros(2, C, H, 129, dtype=torch.complex128, device=dev)
yc = torch.view_as_complex(yh[:2].double().contiguous())
full[:, :, H - Mx // 2:, :My] = yc[:, :, :Mx // 2]
full[:, :, :Mx - Mx // 2, :My] = yc[:, :, Mx // 2:]
ref = torch.fft.irfft2(full, s=(H, 256), norm="forward") + bias.double()[None, :, None, None]
out = {}
for tag, fl in (("mx", _lib.SC_PLAN_IO_BF16), ("valu", _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT)):
    plan = lib.plan_create([H, 256], [Mx, My], flags=fl)
    y = torch.zeros(B, C, H, 256, device=dev, dtype=torch.bfloat16)
    f = lambda: lib.transform_inverse(plan, _lib.SC_INV_PADDED, yh.data_ptr(), bias.data_ptr(), C, y.data_ptr(), B * C, 0, st)
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
    err = float((y[:2].double() - ref).norm() / ref.norm())
    off = float((y[:2] != ref.float().bfloat16()).float().mean())      # vs the float64 result rounded the same way
    y2 = torch.zeros_like(y)
    lib.transform_inverse(plan, _lib.SC_INV_PADDED, yh.data_ptr(), bias.data_ptr(), C, y2.data_ptr(), B * C, 0, st)
    torch.cuda.synchronize()
    rep = bool(torch.equal(y.view(torch.int16), y2.view(torch.int16)))
    out[tag] = y.clone()
    mb = (y.numel() * 2 + yh.numel() * 4) / 1e6
    print(f"{tag:5s} {lib.plan_kernel_name(plan, 1):16s} {best * 1e3:7.1f} us per launch  {mb / best / 1e3:6.2f} TB/s  rel-L2 vs float64 {err:.2e}  off the rounded float64 result on {off:.2e} of the outputs  repeatable {rep}")
    lib.plan_destroy(plan)
d = float((out["mx"] != out["valu"]).float().mean())
print(f"outputs that differ between the two: {d:.2e} of all")
