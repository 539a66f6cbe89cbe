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

Set-up order (round 6, ADVICE r5 -- the layout of rccl_native.get): (1) the rank-local steps -- world <= 8, window
allocation -- then ONE MIN all-reduce, so that a rank that cannot allocate sends everybody to the torch path before
anything can block; (2) the handle exchange, every rank opening what it can and ALL ranks taking part in the collectives
whatever failed locally, a second agreement; (3) the self-test under a 2 s spin budget (``sc_peer_window_control``: a
dead peer or a mapping that is not coherent ends as an error word the host reads, not as a stream that never drains)
and a third agreement.  A rank on the disagree path frees what it built.  Windows that were handed out are never
freed before ``shutdown()`` (their addresses may be baked into captured hipGraphs): a larger exchange gets a NEW,
larger set and the old one stays mapped; growing during a stream capture is refused.  The cache entry remembers which
ProcessGroup object, world and rank it was decided for (weak reference), like rccl_native's.

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
    """Windows of one process group for exchanges of up to ``max_bytes`` per rank and direction.  The constructor is
    rank-local (allocation only); ``connect`` is the collective part."""

    def __init__(self, group, max_bytes):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.lib = _lib.get_lib()
        self.max_bytes = int(max_bytes)
        self.count = 0
        self._own, self._handles, self._opened, self.windows = [], [], [], []
        if self.world > 8:
            raise RuntimeError("peer-store exchange: at most 8 ranks of one node")
        for _ in range(SLOTS):
            p, h = self.lib.peer_window_alloc(self.max_bytes)
            self._own.append(p)
            self._handles.append(h)

    def connect(self, local_ok=True):
        """COLLECTIVE (every rank of the group calls it, also one whose allocation failed: ``local_ok`` False): exchange
        the handles, map every peer's windows.  -> (this rank mapped everything, reason)"""
        box = [None] * self.world
        dist.all_gather_object(box, self._handles if local_ok else None, group=self.group)
        ok, reason = bool(local_ok), ""
        if ok and any(b is None for b in box):
            ok, reason = False, "a peer could not allocate its windows"
        if ok:
            try:
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
            except Exception as e:                           # hipIpcOpenMemHandle: keep going to the agreement below
                ok, reason = False, f"{type(e).__name__}: {e}"
        return ok, reason

    def set_spin_budget(self, ms):
        """budget of the wait launches on this rank's windows (0 = unbounded)"""
        for p in self._own:
            self.lib.peer_window_control(p, int(ms))

    def check(self):
        """error words of this rank's windows (after a synchronize): raises when a wait ran out of its budget"""
        errs = [self.lib.peer_window_control(p) for p in self._own]
        if any(errs):
            raise RuntimeError("peer-store exchange: a wait timed out (flag of peer %d never came)" % (max(errs) - 1))

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

    def destroy(self, collective=True):
        """collective=False: the disagree path of get() (the other ranks may be anywhere: no barrier; a peer that still
        maps one of these windows keeps its own mapping alive until it closes it)"""
        torch.cuda.synchronize()
        for p in self._opened:
            try:
                self.lib.peer_window_close(p)
            except Exception:
                pass
        self._opened = []
        if collective and (self.group is None or dist.is_initialized()):
            try:
                dist.barrier(group=self.group)             # nobody still maps a window that is about to be freed
            except Exception:
                pass
        for p in self._own:
            try:
                self.lib.peer_window_free(p)
            except Exception:
                pass
        self._own, self.windows = [], []


def _agree(ok, group):
    """MIN all-reduce of a rank-local verdict: True only when every rank says True"""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def _pg(group):
    try:
        return group if group is not None else dist.distributed_c10d._get_default_group()
    except Exception:
        return None


def get(group, nbytes):
    """PeerExchange of ``group`` whose windows hold ``nbytes`` per exchange, or None (torch / RCCL path); decided once per
    (process group, window size), identically on every rank (collective on first use and when the windows must grow)."""
    global LAST_REASON
    if not wanted():
        return None
    import weakref
    key = id(group) if group is not None else 0
    pg = _pg(group)
    ent = _CACHE.get(key)
    if ent is not None:
        ref, world, rank, cur, retired = ent
        same = pg is not None and ref() is pg and world == dist.get_world_size(group) and rank == dist.get_rank(group)
        if not same:                                         # another world behind a reused id(): nothing of it is valid
            for e in ([cur] if cur else []) + retired:
                e.destroy(collective=False)
            del _CACHE[key]
            ent = None
        elif cur is False or (cur and cur.max_bytes >= nbytes):
            return cur or None
    retired = list(ent[4]) if ent else []
    if ent and ent[3]:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("peer-store exchange: the windows would have to grow during a stream capture "
                               f"({ent[3].max_bytes} -> {nbytes} bytes); run one eager step of the largest shape first")
        retired.append(ent[3])                               # never freed before shutdown(): captured graphs may hold it
    ex, reason = None, ""
    try:                                                     # (1) rank-local
        ex = PeerExchange(group, max(int(nbytes), 1 << 20))
    except Exception as e:
        reason = f"{type(e).__name__}: {e}"
    local_ok = ex is not None and not reason
    all_ok = _agree(local_ok, group)
    if all_ok:                                               # (2) handles + mapping: every rank is here
        ok, why = ex.connect(True)
        reason = reason or why
        all_ok = _agree(ok, group)
        if all_ok:                                           # (3) self-test under a spin budget
            ok, budget_set = False, True
            try:
                ex.set_spin_budget(2000)
            except Exception as e:
                budget_set, reason = False, f"{type(e).__name__}: {e}"
            dist.barrier(group=group)                        # every rank has mapped every window before the first store
            try:
                ok = budget_set and _self_test(ex)           # (its first call is the reference collective: every rank is in it)
                if not ok:
                    reason = "self-test mismatch against torch.distributed.all_to_all_single"
                ex.check()
                ex.set_spin_budget(int(os.environ.get("SC_MPU_PEER_SPIN_MS", "0")))
            except Exception as e:
                ok, reason = False, f"{type(e).__name__}: {e}"
            all_ok = _agree(ok, group)
    if not all_ok:
        if ex is not None:
            ex.destroy(collective=False)
        if not reason:
            reason = "another rank failed"
        ex = None
    LAST_REASON = reason
    if pg is not None:
        _CACHE[key] = (weakref.ref(pg), dist.get_world_size(group), dist.get_rank(group), ex if ex is not None else False, retired)
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
        torch.cuda.synchronize()                             # bounded: the waits run under the 2 s budget set by get()
        ok = ok and bool(torch.equal(got, want))
    return ok


def active():
    return any(e[3] for e in _CACHE.values())


def shutdown():
    for e in _CACHE.values():
        for x in ([e[3]] if e[3] else []) + list(e[4]):
            x.destroy()
    _CACHE.clear()
