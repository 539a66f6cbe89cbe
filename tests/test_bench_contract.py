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
