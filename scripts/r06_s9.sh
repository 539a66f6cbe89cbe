#!/bin/bash
# round 6, GPU call 9: soak of the shipped k_fft2d_inv_mx (1000 launches x 3 heights x 2 modes), the GPU repeat test, bf16 parity
cd /root/repo; O=gpurun_out/r06_s9; mkdir -p $O
{ for H in 64 128 256; do for m in 0 1; do python scripts/mx_ifft_repeat.py $H 1000 $m; done; done
  python scripts/mx_ifft_repeat.py 64 300 0 20 9 2048; python scripts/mx_ifft_repeat.py 64 300 0 64 33 700; python scripts/mx_ifft_repeat.py 64 300 1 12 33 513
} > $O/mx_ifft_soak.txt 2>&1
cat $O/mx_ifft_soak.txt
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "bf16" 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
