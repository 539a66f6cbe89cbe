"""CPU tier: the timing contract of bench.py without a GPU -- W untimed + exactly K timed steps per region, the cold
region first, then the settling steps (their number derived from the cold time), then the region `value` reports."""
import importlib
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return importlib.import_module("bench")


def test_timed_steps_regions(bench):
    calls = []

    def step():
        calls.append(time.perf_counter())
        time.sleep(0.002)

    ms, cold, n = bench.timed_steps(step, steps=5, warmup=2, dist=None, dev=None, share=False, settle_ms=20.0)
    # cold region 2 + 5, settling steps, settled region 2 + 5
    assert len(calls) == 2 * (2 + 5) + n
    assert n == int(20.0 / cold) + 1 and 1 <= n <= 11                    # ~2 ms per step -> ~10 settling steps
    assert ms >= 1.9 and cold >= 1.9                                      # (no upper bounds: a loaded host sleeps longer)


def test_settle_zero_reports_the_cold_region(bench):
    count = [0]

    def step():
        count[0] += 1

    ms, cold, n = bench.timed_steps(step, steps=4, warmup=1, dist=None, dev=None, share=False, settle_ms=0)
    assert count[0] == 5 and n == 0 and ms == cold


def test_alg_bytes_metric_shape(bench):
    # 4 R + 3 Wb + 9 S = 2,666.5 MB per 32-sample step at the metric shape
    R, Wb, S, total = bench.alg_bytes(32, 64, (256, 256), [64, 33], 4)
    assert (R, Wb, S) == (32 * 64 * 256 * 256 * 4, 64 * 64 * 64 * 33 * 8, 32 * 64 * 64 * 33 * 8)
    assert total == 4 * R + 3 * Wb + 9 * S == 2666528768


def test_kernel_key_and_traffic_table_from_a_counter_database(bench, tmp_path, monkeypatch):
    """measure_step_traffic (round 4: live FETCH_SIZE / WRITE_SIZE of EVERY kernel of a workload's step) on a stand-in for
    rocprofv3: a fake executable that writes the rocpd sqlite views the real one writes.  Checks the kernel-name
    normalisation, FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KB = 1024 B, launches per step, the per-step sum and the
    dominant kernel of an extra.* entry."""
    assert bench._kernel_key("void k_fft2d_fwd3<256, float>(float const*, cf32*, int)") == "k_fft2d_fwd3<256, float>"
    assert bench._kernel_key("k_f2p_c2r_w1024(cf32 const*, float*)") == "k_f2p_c2r_w1024"
    assert bench._kernel_key("k_modegemm_dma<4, 2, 2, 3, false, (bool)1, false>(Gemm8Args)") == \
        "k_modegemm_dma<4, 2, 2, 3, false, (bool)1, false>"
    fake = tmp_path / "rocprofv3"
    fake.write_text('''#!/usr/bin/env python3
import os, sqlite3, sys
a = sys.argv
counter, d = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
os.makedirs(d, exist_ok=True)
db = sqlite3.connect(os.path.join(d, "run.db"))
db.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
db.execute("create table top_kernels (name text, total_calls int, total_duration real, average real, percentage real)")
rows = {"FETCH_SIZE": [("void k_a<1>(int)", 100.0), ("void k_a<1>(int)", 100.0), ("k_b(float*)", 10.0), ("k_b(float*)", 10.0),
                       ("k_b(float*)", 10.0), ("k_b(float*)", 10.0), ("at::native::fill", 5.0)],
        "WRITE_SIZE": [("void k_a<1>(int)", 50.0), ("void k_a<1>(int)", 50.0), ("k_b(float*)", 1.0), ("k_b(float*)", 1.0),
                       ("k_b(float*)", 1.0), ("k_b(float*)", 1.0)]}
for k, v in rows[counter]:
    db.execute("insert into counters_collection values (?, ?, ?)", (k, counter, v))
db.execute("insert into top_kernels values ('void k_a<1>(int)', 2, 400.0, 200.0, 80.0)")
db.execute("insert into top_kernels values ('k_b(float*)', 4, 80.0, 20.0, 20.0)")
db.commit()
''')
    fake.chmod(0o755)
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: str(fake) if name == "rocprofv3" else None)
    got, note = bench.measure_step_traffic((2, 4, (16, 16), (8, 8)), reps=2)
    assert got is not None, note
    ka, kb = got["kernels"]["k_a<1>"], got["kernels"]["k_b"]
    assert ka == {"launches_per_step": 1.0, "fetch_B": 100 * 1024 * 2, "write_B": 50 * 1024, "traffic_B": 250 * 1024, "ms": 0.2}
    assert kb["launches_per_step"] == 2.0 and kb["traffic_B"] == 21 * 1024 and kb["ms"] == 0.02
    assert got["step_traffic_B"] == 250 * 1024 + 2 * 21 * 1024            # torch's own kernels are not counted
    ex = bench._extra_traffic((2, 4, (16, 16), (8, 8)), "f32", "dense", 292 * 1024 // 2)
    assert ex["traffic"] == 292 * 1024 and ex["traffic_over_alg_bytes"] == 2.0 and ex["dominant_kernel"]["name"] == "k_a<1>"


def test_launch_command_is_the_contracts_line(bench):
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_gpus_n_without_a_launcher_starts_n_ranks_or_fails(bench, monkeypatch, capsys):
    """`python bench.py --gpus N` with WORLD_SIZE unset must start N ranks itself and must not
    quietly run one.  Here (no device): non-zero exit; with devices the launch line is handed to a subprocess."""
    import subprocess
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("SC_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    assert bench.self_launch(8, ["--gpus", "8"]) == 2
    assert "needs 8 visible devices" in capsys.readouterr().err
    seen = {}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    assert bench.self_launch(8, ["--gpus", "8", "--steps", "2"]) == 0
    assert "--nproc-per-node=8" in seen["cmd"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main(): a mismatch between --gpus and the launcher's WORLD_SIZE is an error, not a note
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value)
    # and --gpus N without WORLD_SIZE goes through self_launch and exits with ITS code
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: 7)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7


def test_compact_configs_is_the_last_key_and_fits_the_record_tail(bench):
    """the driver's record keeps the last 2000 characters of the line -- the `configs` object (every
    BASELINE config: ms, cold ms, samples/s, fraction of 8 TB/s, counter traffic over algorithmic bytes) must be the last key
    and fit with room to spare, for a one-GPU line and for an N-GPU line."""
    import json
    ex = lambda ms: {"ms_per_step": ms, "cold_start_ms_per_step": ms * 1.1234, "value": 61234.56, "frac_of_8TBs": 0.5123,
                     "traffic_over_alg_bytes": 1.234, "traffic": 123456789012, "workload": "x" * 40}
    out = {"ms_per_step": 0.5197, "value": 61575.31, "cold_start": {"ms_per_step": 0.5934},
           "step_roofline": {"frac_of_8TBs": 0.6413, "traffic_over_alg_bytes": 1.029},
           "config": {"workload": "fno2d_256_m64_c64_b32", "real_tensor_io": "f32", "parallelism": "single"}}
    extra = {"bf16_io": ex(0.3722), "tfno_rank01": ex(0.7259), "fno3d_single": ex(1.904), "fno2d_1024_b4": ex(5.37),
             "darcy_421": ex(0.9), "fno_block": {"fused_ms": 2.79, "reference_op_sequence_ms": 12.3},
             "sfno": {"engine_ms": 0.59, "op_sequence_ms": 0.71},
             "fno3d_rank_of_8": {"ms_per_step": 0.34, "cold_start_ms_per_step": 0.36, "launch": "hipGraph replay"}}
    c = bench.compact_configs(out, extra, 1)
    assert set(c) >= {"c1_f32", "c1_bf16", "c2_tfno", "c3_single", "c4_1024", "block", "sfno", "c3_rank_of_8"}
    assert c["c3_rank_of_8"] == {"ms": 0.34, "bound_x": 5.6}       # one GPU: the per-rank budget of 8 ranks (timing only)
    assert c["c1_f32"] == {"ms": 0.5197, "cold": 0.5934, "sps": 61575.31, "frac": 0.6413, "toa": 1.029}
    out["configs"] = c
    line = json.dumps(out)
    assert list(out)[-1] == "configs" and line.rstrip().endswith("}}")
    assert len(json.dumps(c)) <= 1500 and '"configs": ' + json.dumps(c) in line[-2000:]
    # an 8-GPU line: mode-sharded configs[3] strong-scaled beside the one-GPU replica step and their ratio
    out8 = dict(out, config={"workload": "fno2d_256_m64_c64_b32", "real_tensor_io": "f32", "parallelism": "dp8-allreduce"})
    e8 = {"dp_allreduce": ex(0.9), "fno3d_modeshard": ex(0.30), "fno3d_single": ex(1.92)}
    c8 = bench.compact_configs(out8, e8, 8)
    assert c8["n_gpus"] == 8 and c8["c3_speedup"] == 6.4 and len(json.dumps(c8)) <= 1500
