"""CPU tier: selection logic of the engine's own RCCL path (neuraloperator_amd/mpu/rccl_native.py) -- no GPU, no RCCL call:
without a process group, on a gloo group, when it was not asked for and when it is forbidden the torch.distributed path
stays; the ncclUniqueId travels as 128 raw bytes (a c_char array would be cut at its first NUL byte, which is how the first
version of the binding failed in ncclCommInitRank)."""
import ctypes
import os

import pytest
import torch.distributed as dist

from neuraloperator_amd.mpu import rccl_native


@pytest.fixture(autouse=True)
def _clean():
    rccl_native._CACHE.clear()
    rccl_native.prefer_native(False)
    yield
    rccl_native._CACHE.clear()
    rccl_native.prefer_native(False)


def test_unique_id_roundtrip_keeps_nul_bytes():
    uid = rccl_native._UniqueId()
    raw = bytes([(7 * i) % 256 if i % 5 else 0 for i in range(rccl_native.NCCL_UNIQUE_ID_BYTES)])     # NULs inside
    ctypes.memmove(ctypes.byref(uid), raw, rccl_native.NCCL_UNIQUE_ID_BYTES)
    assert ctypes.string_at(ctypes.byref(uid), rccl_native.NCCL_UNIQUE_ID_BYTES) == raw
    assert ctypes.sizeof(uid) == 128


def test_not_requested_and_forbidden(monkeypatch):
    monkeypatch.delenv("SC_MPU_A2A", raising=False)
    assert rccl_native.get(None) is None and "not requested" in rccl_native.LAST_REASON
    rccl_native._CACHE.clear()
    rccl_native.prefer_native()
    assert rccl_native.get(None) is None and "no process group" in rccl_native.LAST_REASON
    rccl_native._CACHE.clear()
    monkeypatch.setenv("SC_MPU_A2A", "torch")
    assert rccl_native.get(None) is None and "SC_MPU_A2A=torch" in rccl_native.LAST_REASON


def test_gloo_group_keeps_the_torch_path(tmp_path):
    store = dist.FileStore(str(tmp_path / "store"), 1)
    dist.init_process_group("gloo", store=store, rank=0, world_size=1)
    try:
        rccl_native.prefer_native()
        assert rccl_native.get(None) is None
        assert "backend gloo" in rccl_native.LAST_REASON
        os.environ["SC_MPU_A2A"] = "native"                  # forcing it on a gloo group is an error, not a silent fallback
        rccl_native._CACHE.clear()
        with pytest.raises(rccl_native.RcclError):
            rccl_native.get(None)
    finally:
        os.environ.pop("SC_MPU_A2A", None)
        dist.destroy_process_group()


def test_cache_entry_is_tied_to_the_process_group_and_the_request(tmp_path, monkeypatch):
    """the per-group decision must not outlive the group it was taken for (destroy + init without
    comm.cleanup(), a reused id()), and a "not requested" decision is taken again once prefer_native() was called."""
    monkeypatch.delenv("SC_MPU_A2A", raising=False)
    store = dist.FileStore(str(tmp_path / "s1"), 1)
    dist.init_process_group("gloo", store=store, rank=0, world_size=1)
    try:
        assert rccl_native.get(None) is None and "not requested" in rccl_native.LAST_REASON
        ent = rccl_native._CACHE[0]
        assert ent[0]() is dist.distributed_c10d._get_default_group() and ent[1:3] == (1, 0) and ent[4] == "not requested"
        rccl_native.LAST_REASON = "stale"
        assert rccl_native.get(None) is None and rccl_native.LAST_REASON == "stale"      # a hit: nothing re-evaluated
        rccl_native.prefer_native()
        assert rccl_native.get(None) is None and "backend gloo" in rccl_native.LAST_REASON   # asked for now: decided again
    finally:
        dist.destroy_process_group()
    # the same key (0 = default group), another process group object: the entry no longer matches
    marker = object()
    ent = rccl_native._CACHE[0]
    rccl_native._CACHE[0] = (ent[0], ent[1], ent[2], marker, ent[4])

    class Fake:
        destroyed = False

        def destroy(self):
            Fake.destroyed = True
    rccl_native._CACHE[0] = (ent[0], ent[1], ent[2], Fake(), "")
    store = dist.FileStore(str(tmp_path / "s2"), 1)
    dist.init_process_group("gloo", store=store, rank=0, world_size=1)
    try:
        assert rccl_native.get(None) is None and Fake.destroyed and "backend gloo" in rccl_native.LAST_REASON
    finally:
        dist.destroy_process_group()
