"""Peer-store exchange for the mode-parallel layer (round 5, OPT-IN: ``SC_MPU_A2A=peer`` or ``prefer_peer()``).

The exchange step of BASELINE configs[3] on 8 GPUs is 4.46 MB per rank and direction, 557 KB per peer -- 3.6 us on one
xGMI link: latency, not bandwidth, and four of them sit on the critical path of a ~0.3 ms per-rank step (DESIGN.md
section 6).  Through RCCL each is a collective kernel with its own proxy / channel set-up; here it is three plain launches
of the engine (csrc/sc_kernels_peer.h, ``sc_peer_all_to_all``): every rank stores its blocks straight into the peers'
windows (fine-grained device memory mapped through HIP IPC), signals with a system-scope flag per peer, waits for its own
P flags (ONE spinning workgroup) and copies its window into the receive tensor.  Stream-ordered, no host synchronisation, records into a hipGraph.

Set-up (collective over the group, once per window size): every rank allocates ``SLOTS`` windows, the 64-byte IPC handles
travel through the torch process group (``all_gather_object``), every rank maps every peer's windows.  The slots rotate
so that a window is reused ``SLOTS`` exchanges later (two layer steps at four exchanges per step).  A start-up
self-check compares one exchange with ``torch.distributed.all_to_all_single`` bit for bit on every rank; any failure on
any rank sends ALL ranks back to the torch / RCCL path (the decision is all-reduced), with the reason in ``LAST_REASON``.

One node only (HIP IPC), at most 8 ranks.  UNMEASURED on more than one GPU: the build environment has one device; the
tests run two ranks as two processes on that device (tests/test_gpu_peer_exchange.py).
No reference counterpart: neuralop/mpu uses torch.distributed throughout (mpu/comm.py, mpu/helpers.py:81-99)."""
import os

import torch
import torch.distributed as dist

from .. import _lib

SLOTS = 8
LAST_REASON = ""
_WANT = False
_CACHE = {}


def prefer_peer(flag=True):
    """Ask for the peer-store path (before the first step of a layer: the choice is cached per group)."""
    global _WANT
    _WANT = bool(flag)


def wanted():
    return _WANT or os.environ.get("SC_MPU_A2A", "").lower() == "peer"


class PeerExchange:
    """Windows of one process group for exchanges of up to ``max_bytes`` per rank and direction."""

    def __init__(self, group, max_bytes):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > 8:
            raise RuntimeError("peer-store exchange: at most 8 ranks of one node")
        self.lib = _lib.get_lib()
        self.max_bytes = int(max_bytes)
        self.count = 0
        mine = [self.lib.peer_window_alloc(self.max_bytes) for _ in range(SLOTS)]
        self._own = [p for p, _ in mine]
        box = [None] * self.world
        dist.all_gather_object(box, [h for _, h in mine], group=group)
        self._opened = []
        self.windows = []                                  # [slot][peer] -> mapped base
        for s in range(SLOTS):
            row = []
            for p in range(self.world):
                if p == self.rank:
                    row.append(self._own[s])
                else:
                    ptr = self.lib.peer_window_open(box[p][s])
                    self._opened.append(ptr)
                    row.append(ptr)
            self.windows.append(row)
        dist.barrier(group=group)                          # every rank has mapped every window before the first store

    def all_to_all(self, send, recv, stream):
        """block p of ``send`` ([P, ...] contiguous float32) -> rank p; block p of ``recv`` <- rank p; on ``stream``"""
        if send.dtype != torch.float32 or not (send.is_contiguous() and recv.is_contiguous() and send.is_cuda):
            raise ValueError("peer-store exchange: contiguous float32 device tensors")
        nbytes = send.numel() * 4
        if send.numel() != recv.numel() or nbytes % self.world or (nbytes // self.world) % 16:
            raise ValueError("peer-store exchange: equal blocks of whole 16-byte units per rank")
        if nbytes > self.max_bytes:
            raise ValueError(f"peer-store exchange: {nbytes} bytes exceed the windows ({self.max_bytes})")
        slot = self.count % SLOTS
        self.count += 1
        self.lib.peer_all_to_all(self.world, self.rank, nbytes // self.world, self.windows[slot], send.data_ptr(),
                                 recv.data_ptr(), stream)

    def destroy(self):
        torch.cuda.synchronize()
        for p in self._opened:
            self.lib.peer_window_close(p)
        self._opened = []
        if self.group is None or dist.is_initialized():
            try:
                dist.barrier(group=self.group)             # nobody still maps a window that is about to be freed
            except Exception:
                pass
        for p in self._own:
            self.lib.peer_window_free(p)
        self._own = []


def get(group, nbytes):
    """PeerExchange of ``group`` whose windows hold ``nbytes`` per exchange, or None (torch / RCCL path); decided once per
    (group, window size), identically on every rank (collective on first use)."""
    global LAST_REASON
    if not wanted():
        return None
    key = (id(group) if group is not None else 0, dist.get_world_size(group), dist.get_rank(group))
    ent = _CACHE.get(key)
    if ent is not None and (ent is False or ent.max_bytes >= nbytes):
        return ent or None
    ok, reason, ex = 1, "", None
    try:
        if ent:
            ent.destroy()
        ex = PeerExchange(group, max(int(nbytes), 1 << 20))
        ok = 1 if _self_test(ex) else 0
        if not ok:
            reason = "self-test mismatch against torch.distributed.all_to_all_single"
    except Exception as e:                                   # allocation / IPC / launch failure: the torch path
        ok, reason = 0, f"{type(e).__name__}: {e}"
    flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        if ex is not None and not reason:
            reason = "another rank failed"
        ex = None
    LAST_REASON = reason
    _CACHE[key] = ex if ex is not None else False
    return ex


def _self_test(ex):
    P = ex.world
    g = torch.Generator(device="cuda").manual_seed(4321 + ex.rank)
    send = torch.randn(P, 1028, device="cuda", generator=g)
    want = torch.empty_like(send)
    if dist.get_backend(ex.group) == "nccl":
        dist.all_to_all_single(want, send, group=ex.group)
    else:                                                    # gloo (tests: several ranks on one device): through the host
        hs, hw = send.cpu(), torch.empty(P, 1028)
        dist.all_to_all_single(hw, hs, group=ex.group)
        want = hw.cuda()
    ok = True
    st = torch.cuda.current_stream()
    for _ in range(SLOTS + 1):                               # every slot once, the first one twice (epoch 2)
        got = torch.full_like(send, float("nan"))
        ex.all_to_all(send, got, st.cuda_stream)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(got, want))
    return ok


def active():
    return any(e for e in _CACHE.values())


def shutdown():
    for e in _CACHE.values():
        if e:
            e.destroy()
    _CACHE.clear()
