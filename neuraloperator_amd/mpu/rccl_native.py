"""RCCL collectives issued straight on a HIP stream (round 4).

The exchange step of the mode-parallel layer is an all-to-all of a few MB per rank.  Through ``torch.distributed`` every
call costs ~50 us of HOST time (Python -> c10d dispatch -> Work object -> watchdog bookkeeping), which makes the per-rank
step of BASELINE configs[3] at 8 GPUs host-issue bound (0.57-0.65 ms per step for 0.28 ms of kernels,
profiles/r04_modeshard_host.txt), and its watchdog thread makes the step impossible to record into a hipGraph on this
stack.  This module binds the RCCL library of the running torch build with ctypes and issues ``ncclAllToAll`` /
grouped ``ncclSend`` + ``ncclRecv`` / ``ncclAllReduce`` on a stream of the caller's choice: a few microseconds per call,
plain stream-ordered launches (events for overlap, as for any kernel), capturable.

One communicator per process group, created once: the group's rank 0 draws the ``ncclUniqueId`` and hands it to the other
ranks through the torch group itself (``broadcast_object_list``), every rank calls ``ncclCommInitRank``.  Selection:
the native path is taken when it was ASKED FOR -- ``prefer_native()`` (what ``bench.py --graph`` and
``capture_step(..., post=...)`` users call before the first step) or ``SC_MPU_A2A=native`` -- AND the group's backend is
nccl AND the communicator comes up AND a small all-to-all agrees bit for bit with
``torch.distributed.all_to_all_single`` on every rank; anything else falls back to the torch path on ALL ranks (the
decision is all-reduced), with the reason kept in ``LAST_REASON``.  It is not the default of eager steps: measured on
one rank an eager step costs the host the same either way (0.586 vs 0.591 ms, profiles/r04_modeshard_host.txt) -- the
Python / autograd / ctypes time around the calls dominates -- so the default keeps the path every earlier round ran;
the native path pays through the hipGraph it makes possible (0.59 -> 0.43 ms per step).  ``SC_MPU_A2A=torch`` forbids it.

No reference counterpart: neuralop/mpu uses torch.distributed throughout (mpu/comm.py, mpu/helpers.py:81-99)."""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
NCCL_FLOAT32 = 7
NCCL_SUM = 0

LAST_REASON = ""
_CACHE = {}
_WANT = False


def prefer_native(flag=True):
    """Ask for the native path (before the first step of a layer: the choice is cached per group)."""
    global _WANT
    _WANT = bool(flag)


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)]     # (c_char arrays read back NUL-terminated)


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path)              # the instance torch itself loaded
    vp, st = ctypes.c_void_p, ctypes.c_size_t
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclAllToAll.argtypes = [vp, vp, st, ctypes.c_int, vp, vp]
    lib.ncclSend.argtypes = [vp, st, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.ncclRecv.argtypes = [vp, st, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, st, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllToAll", "ncclSend", "ncclRecv",
              "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd"):
        getattr(lib, f).restype = ctypes.c_int
    return lib


class RcclError(RuntimeError):
    pass


class NativeComm:
    """One RCCL communicator over the ranks of a torch process group (its own, next to torch's)."""

    def __init__(self, group, lib=None):
        """Collective over `group`: every rank must call it (get() has already agreed on that, and on a loadable
        library, before anything here can block).  Rank 0 broadcasts a sentinel when it cannot draw the id, so a failure
        there is an exception on every rank instead of a hang in the broadcast."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.lib = lib if lib is not None else _load()
        uid = _UniqueId()
        payload = None
        if self.rank == 0:
            rc = self.lib.ncclGetUniqueId(ctypes.byref(uid))
            payload = ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES) if rc == 0 else \
                "ncclGetUniqueId: " + self.lib.ncclGetErrorString(rc).decode()
        box = [payload]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        if not isinstance(box[0], bytes) or len(box[0]) != NCCL_UNIQUE_ID_BYTES:
            raise RcclError(box[0] if isinstance(box[0], str) else "unique id did not arrive")
        ctypes.memmove(ctypes.byref(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        comm = ctypes.c_void_p()
        self._check(self.lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.comm = comm
        self.stream = torch.cuda.Stream()                 # where overlapped exchanges run
        self._events = [torch.cuda.Event() for _ in range(8)]   # ring reused by launch_async (no Event per call)
        self._ev_i = 0

    def _check(self, rc, what):
        if rc != 0:
            raise RcclError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()} ({rc})")

    @staticmethod
    def _f32(t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("native RCCL path: contiguous float32 device tensors")
        return t

    # ---- stream-ordered calls: `stream` = raw hipStream_t (int)
    def all_to_all(self, send, recv, stream):
        """block p of `send` ([P, ...] contiguous) goes to rank p; block p of `recv` comes from rank p"""
        self._f32(send), self._f32(recv)
        if send.numel() != recv.numel() or send.numel() % self.world:
            raise ValueError("all_to_all: equal blocks per rank")
        self._check(self.lib.ncclAllToAll(send.data_ptr(), recv.data_ptr(), send.numel() // self.world, NCCL_FLOAT32,
                                          self.comm, stream), "ncclAllToAll")

    def all_to_all_slabs(self, send_slabs, recv_slabs, stream):
        """slab p of `send_slabs` goes to rank p, slab p of `recv_slabs` comes from rank p (one grouped send / receive
        per peer: the transfers of an all-to-all straight out of / into slabs of larger tensors)"""
        self._check(self.lib.ncclGroupStart(), "ncclGroupStart")
        try:
            for p in range(self.world):
                s, r = self._f32(send_slabs[p]), self._f32(recv_slabs[p])
                self._check(self.lib.ncclSend(s.data_ptr(), s.numel(), NCCL_FLOAT32, p, self.comm, stream), "ncclSend")
                self._check(self.lib.ncclRecv(r.data_ptr(), r.numel(), NCCL_FLOAT32, p, self.comm, stream), "ncclRecv")
        finally:
            self._check(self.lib.ncclGroupEnd(), "ncclGroupEnd")

    def all_reduce_sum_(self, t, stream):
        self._f32(t)
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm,
                                           stream), "ncclAllReduce")

    def all_reduce_sum_on_current(self, t):
        """in-place sum over the ranks, ordered like a launch on the current stream (see launch_async: eager calls run
        on this communicator's own stream, calls inside a hipGraph capture on the capturing stream)"""
        if torch.cuda.is_current_stream_capturing():
            self.all_reduce_sum_(t, torch.cuda.current_stream().cuda_stream)
        else:
            self.launch_async(lambda st: self.all_reduce_sum_(t, st), (t,)).wait()

    # ---- overlapped form: the exchange runs on this communicator's own stream behind everything issued to the current
    #      stream so far; wait() makes the current stream wait for it (events only: no host synchronisation)
    #      The caller keeps `tensors` referenced until it has waited (mode_parallel's `pend` lists do): the current stream
    #      is then ordered behind the exchange before the allocator can hand the memory to anything else, so no
    #      record_stream is needed (it is not allowed inside a hipGraph capture either).
    def launch_async(self, fn, tensors):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        fn(self.stream.cuda_stream)
        ev = self._events[self._ev_i]                     # at most a few exchanges are in flight per layer (<= 4 chunks)
        self._ev_i = (self._ev_i + 1) % len(self._events)
        return _Pending(self.stream, tensors, ev)

    def destroy(self):
        # The communicator is NOT destroyed: ncclCommDestroy blocks for good once a hipGraph that recorded this
        # communicator's kernels exists in the process (RCCL 2.26.6, seen as a hang of comm.cleanup()), and a handful of
        # communicators per process are released with the process anyway.  The handle is only forgotten.
        self.comm = None


class _Pending:
    def __init__(self, stream, keep, event=None):
        self.keep = keep                                  # alive until waited for
        self.event = event if event is not None else torch.cuda.Event()
        self.event.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        self.keep = None


def _mode():
    return os.environ.get("SC_MPU_A2A", "auto").lower()


def _pg(group):
    """the ProcessGroup object behind `group` (None = the default group)"""
    if group is not None:
        return group
    try:
        return dist.distributed_c10d._get_default_group()
    except Exception:
        return None


def _agree(ok, group):
    """MIN over the group of a local 0 / 1 flag (one small all-reduce on the group's own backend)"""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def get(group):
    """NativeComm of `group`, or None (torch path): decided once per group, identically on every rank.

    The cache entry remembers WHICH process group it was decided for (a weak reference to the ProcessGroup object plus
    world size and rank): after destroy_process_group / init_process_group, or when a collected group's id() is reused,
    the entry no longer matches and the decision is taken again instead of handing out a communicator of another world
    (ADVICE r4).  A negative decision that only said "not requested" is re-evaluated once prefer_native() has been
    called.  Order of the collective part: (1) rank-local checks -- mode, request flag, library and symbols -- and ONE
    all-reduce of that flag, so a rank that cannot load librccl, or ranks that disagree about SC_MPU_A2A /
    prefer_native(), send everybody to the torch path before anything can block; (2) unique id + ncclCommInitRank;
    (3) the self-test and a second all-reduce."""
    global LAST_REASON
    key = id(group) if group is not None else 0
    pg = _pg(group)
    ent = _CACHE.get(key)
    if ent is not None:
        ref, world, rank, comm, why = ent
        same = pg is not None and ref() is pg and world == dist.get_world_size(group) and rank == dist.get_rank(group)
        if same and not (comm is None and why == "not requested" and (_WANT or _mode() == "native")):
            return comm
        if comm is not None:
            comm.destroy()
        del _CACHE[key]
    comm, reason, why = None, "", ""
    mode = _mode()
    lib, local = None, ""
    if mode == "torch":
        local = "SC_MPU_A2A=torch"
    elif mode != "native" and not _WANT:
        local, why = "not requested (prefer_native() / SC_MPU_A2A=native)", "not requested"
    if not (dist.is_available() and dist.is_initialized()):
        reason = local or "no process group"
    elif dist.get_backend(group) != "nccl" or not torch.cuda.is_available():
        reason = local or f"backend {dist.get_backend(group)}"
    else:
        if not local:
            try:
                lib = _load()
            except Exception as e:                           # library / symbol lookup: rank-local, nothing blocks yet
                local = f"{type(e).__name__}: {e}"
        if dist.get_world_size(group) == 1:
            agreed = not local
        else:
            agreed = _agree(not local, group)                # (1): also covers ranks that disagree about the request
        if not agreed:
            reason = local or "another rank does not take the native path"
        else:
            ok = True
            try:
                comm = NativeComm(group, lib)                # (2)
                ok = _self_test(comm)
                if not ok:
                    reason = "self-test mismatch against torch.distributed.all_to_all_single"
            except Exception as e:
                ok, reason = False, f"{type(e).__name__}: {e}"
            if not _agree(ok, group):                        # (3)
                if comm is not None and not reason:
                    reason = "another rank failed"
                comm = None
    LAST_REASON = reason
    if comm is None and mode == "native":                    # forced: an error, not a silent fallback
        raise RcclError("SC_MPU_A2A=native: " + reason)
    if pg is not None:
        import weakref
        _CACHE[key] = (weakref.ref(pg), dist.get_world_size(group), dist.get_rank(group), comm, why)
    return comm


def _self_test(comm):
    P = comm.world
    g = torch.Generator(device="cuda").manual_seed(1234 + comm.rank)
    send = torch.randn(P, 257, device="cuda", generator=g)
    want = torch.empty_like(send)
    dist.all_to_all_single(want, send, group=comm.group)
    got = torch.full_like(send, float("nan"))
    got2 = torch.full_like(send, float("nan"))
    torch.cuda.synchronize()
    # on the communicator's own stream: one that has run on torch's default (the legacy null) stream aborts the process
    # when it is later used inside a hipGraph capture (RCCL 2.26.6)
    with torch.cuda.stream(comm.stream):
        st = comm.stream.cuda_stream
        comm.all_to_all(send, got, st)
        comm.all_to_all_slabs([send[p] for p in range(P)], [got2[p] for p in range(P)], st)
    torch.cuda.synchronize()
    return bool(torch.equal(got, want) and torch.equal(got2, want))


def active():
    """True when some process group of this process exchanges through the native path"""
    return any(e[3] is not None for e in _CACHE.values())


def shutdown():
    """forget every communicator (mpu.comm.cleanup calls this before destroy_process_group)"""
    for e in _CACHE.values():
        if e[3] is not None:
            e[3].destroy()
    _CACHE.clear()
