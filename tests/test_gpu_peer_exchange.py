"""GPU tier: the peer-store exchange (mpu/peer_exchange.py, csrc/sc_kernels_peer.h) with 2 and 4 ranks as separate
processes on ONE device -- windows mapped through HIP IPC, flags with system-scope release / acquire, window regrowth,
hipGraph replays with fresh epochs, and the mode-parallel layer moving its four exchanges per step through it with the
same bits as the torch path.  (More than one GPU: unmeasured, no such tier here.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_peer_exchange_processes_on_one_device(world):
    from neuraloperator_amd.mpu import comm
    port = str(comm.free_port())                          # outside the ephemeral range
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "peer_exchange_case.py"), str(r), str(world), port],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                    # exactly the processes this test started
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} of {world}: ok" in o, o[-3000:]
