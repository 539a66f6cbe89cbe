"""GPU tier: the launch paths of bench.py on a 1-GPU box.

  * `python bench.py --gpus 2` WITHOUT a launcher starts two ranks itself (SC_BENCH_SHARE_GPU=1: both on cuda:0, gloo) and
    rank 0 prints ONE line with n_gpus = 2;
  * the mode-parallel layer with one sample per rank (the per-rank step of configs[3] at 8 GPUs) takes the native-RCCL
    hipGraph step BY DEFAULT after the child-process probe, on a one-rank RCCL group."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
QUIET = ["--no-extras", "--no-pmc", "--no-cpu-baseline", "--no-gpu-reference", "--settle-ms", "0", "--stage-iters", "2"]


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                       # exactly ONE line on stdout
    return json.loads(lines[0]), out.stderr


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_gpus_two_self_launches_two_ranks():
    line, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "darcy_16_m12_c32_b4"] + QUIET,
                     {"SC_BENCH_SHARE_GPU": "1"})
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "modeshard2"
    assert line["config"]["global_batch"] == 4 and line["config"]["B_per_gpu"] == 2 and line["scaling"] == "strong"
    assert line["collectives"]["all_to_all_calls_per_step"] > 0
    assert "starting 2 ranks" in err


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_one_sample_per_rank_takes_the_graph_step_by_default():
    line, err = _run(["--parallel", "modeshard", "--workload", "fno3d_64_m16_c32_b8", "--steps", "3", "--warmup", "1"] + QUIET)
    # B = 8 on one rank: the GPU-bound case stays eager
    assert line["config"]["launch"] == "eager"
    line, err = _run(["--parallel", "modeshard", "--workload", "fno3d_128_m32_c32_b1", "--steps", "3", "--warmup", "1"] + QUIET)
    assert line["config"]["launch"].startswith("hipGraph replay"), (line["config"]["launch"], err[-1500:])
    assert "issued_by" in line["collectives"] and "ncclAllToAll" in line["collectives"]["issued_by"]
    assert line["collectives"]["all_to_all_calls_per_step"] == 4
    assert line["collectives"]["all_to_all_bytes_per_step_per_rank"] == 4 * 8 * 32 * 32 * 32 * 17
    # --no-graph: the A-B
    line, _ = _run(["--parallel", "modeshard", "--workload", "fno3d_128_m32_c32_b1", "--steps", "3", "--warmup", "1",
                    "--no-graph"] + QUIET)
    assert line["config"]["launch"] == "eager"


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_graph_probe_under_torchrun_environment():
    """The driver starts the ranks through torch.distributed.run: its TORCHELASTIC_* variables must not leak into the
    probe children (TORCHELASTIC_USE_AGENT_STORE would send their rendezvous to the parents' store and the probe would
    time out into the eager step)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                          "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--parallel",
                          "modeshard", "--workload", "fno3d_64_m16_c32_b8", "--steps", "3", "--warmup", "1"] + QUIET +
                         ["--chunk-dim", "batch"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                          "127.0.0.1", "--master-port", "29732", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--parallel",
                          "modeshard", "--workload", "fno3d_128_m32_c32_b1", "--steps", "3", "--warmup", "1"] + QUIET,
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert line["config"]["launch"].startswith("hipGraph replay"), (line["config"]["launch"], out.stderr[-2000:])
