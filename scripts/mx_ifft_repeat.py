"""Repeatability stress of k_fft2d_inv_mx: N launches at 2048 images, every result compared bit for bit with the first
(round 5, session 2: H = 64 failed this on hardware -- the kernel is not used there; H = 128 / 256: 0 of 200).
Usage: python scripts/mx_ifft_repeat.py H repeats [Mx My images]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib(); dev = torch.device("cuda:0"); torch.manual_seed(3)
H = int(sys.argv[1]); reps = int(sys.argv[2])
MX, MY, n = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (64, 33, 2048)
yh = torch.randn(n, MX, MY, 2, device=dev); bias = torch.randn(64, device=dev)
st = torch.cuda.current_stream().cuda_stream
plan = lib.plan_create([H, 256], [MX, MY], flags=_lib.SC_PLAN_IO_BF16)
y0 = torch.zeros(n, H, 256, device=dev, dtype=torch.bfloat16); y = torch.zeros_like(y0)
lib.transform_inverse(plan, 0, yh.data_ptr(), bias.data_ptr(), 64, y0.data_ptr(), n, 0, st)
bad = 0
for it in range(reps):
    lib.transform_inverse(plan, 0, yh.data_ptr(), bias.data_ptr(), 64, y.data_ptr(), n, 0, st)
    bad += int(not torch.equal(y.view(torch.int16), y0.view(torch.int16)))
print(f"H {H} kept {MX} x {MY}, {n} images ({lib.plan_kernel_name(plan, 1)}): {bad} of {reps} repeats differ from the first run")
