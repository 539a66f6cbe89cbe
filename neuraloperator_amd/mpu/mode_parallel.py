"""Mode-parallel SpectralConv: Fourier modes sharded over the model-parallel group.

New functionality on the reference's mpu API shape (the reference has
no mode-parallel conv, its ``_transpose`` all-to-all helper, mpu/helpers.py:81-99, is dead
code).  Layout per rank p of P:

    activations   batch-sharded   x_p   (B/P, Cin, d1..dN)            (like data parallel)
    weights       mode-sharded    W_p   (Cin, Cout, rows, k2..kN)     rows [p*rows, (p+1)*rows), rows = ceil(k1/P)

    x_p --pruned rFFT--> xhat_p (B/P, Cin, k1, ..) --all-to-all(split k1, cat batch)-->
    (B, Cin, rows, ..) --contract with W_p--> (B, Cout, rows, ..)
    --all-to-all(split batch, cat k1)--> (B/P, Cout, k1, ..) --zero-padded inverse--> y_p

Each rank sees the whole batch for its modes, so gW needs NO all-reduce (a dense layer's
69 MB weight gradient would be ring-bound on xGMI); the bias is replicated and its gradient
is summed over the group.  The backward is the same pipeline mirrored (2 more all-to-alls).

One autograd Function runs the whole pipeline (``_ModeParallelFn``) so that the exchange can be scheduled by hand:
the local batch (or, for one sample per rank, the channels) is cut into a few chunks, chunk j's all-to-all is
enqueued (RCCL runs it on the process group's own stream) as soon as its transform is launched, and the
transform of chunk j+1 runs meanwhile; on the way back every chunk's inverse transform starts when ITS exchange
has landed.  NO copy is left around the collectives (round 3): the transforms write / read the rank-major
all-to-all buffers in place (engine ``*_sharded`` calls: include/sc_engine.h, sc_spectrum_shards), the receive
buffer of the way out IS the contraction's operand and the contraction's result IS the send buffer of the way back;
the R-sized real tensors are written in place (``out=`` slices).  With one sample per rank (BASELINE configs[3] on 8
GPUs) the channels take the place of the batch as the chunked dim.  Round 4: channel chunks are copy-free as well -- a
chunk's block for rank p is a contiguous slab (channels c0:c1) of the contraction's operand / result, handed to the
list form of the all-to-all (``_Exchange.exchange_slabs``), so ``comm_chunks=2`` lets the second half's transform run
under the first half's exchange on the way out and the first half's inverse under the second half's exchange on the way
back.  The DEFAULT there stays one piece per exchange: measured on one MI355X (a one-rank RCCL group,
profiles/r04_modeshard_host.txt) the per-rank step of configs[3] at 8 GPUs is bound by the HOST's issue time -- 0.57-0.65
ms per step with one piece, 0.83-0.91 ms with two (each all-to-all call costs ~50 us of host time in torch.distributed),
against 0.28 ms of kernels -- so more, smaller collectives lose.  When k1 is
not a multiple of P the mode rows are zero-padded to rows*P on the wire (SURVEY 8e "else pad").
"""
import math

import torch
import torch.distributed as dist
from torch import nn

from ..modes import halve_last_mode, kept_block
from ..spectral_conv import BaseSpectralConv
from . import comm, peer_exchange, rccl_native
from .mappings import A2A_STATS


def _bounds(ext, n):
    return [((ext * i) // n, (ext * (i + 1)) // n) for i in range(n)]


class _Exchange:
    """the two all-to-alls of one direction of the pipeline, chunk by chunk (plain tensors, no autograd)"""

    def __init__(self, group, P, rows, k1, overlap=True):
        self.group, self.P, self.rows, self.k1 = group, P, rows, k1
        # round 4: RCCL straight on a HIP stream where it was asked for and is available (rccl_native: plain stream-ordered
        # launches, capturable into a hipGraph); None = the torch.distributed path (the default of eager steps).
        # overlap=False (one piece per exchange: nothing to run beside it): the exchange still runs on the communicator's
        # own stream (never torch's default stream, see _native) and the current stream waits for it at once -- one
        # wait_stream + one recorded event from the communicator's ring per exchange.
        self.native = rccl_native.get(group)
        self.overlap = overlap

    def _native(self, fn, tensors):
        # inside a hipGraph capture the exchange stays on the capturing stream: RCCL calls on a stream that joined the
        # capture through an event abort the process on this stack (RCCL 2.26.6), on the capturing stream itself they
        # record fine (profiles/r04_modeshard_host.txt)
        if torch.cuda.is_current_stream_capturing():
            fn(torch.cuda.current_stream().cuda_stream)
            return _Works([])
        # eager: always on the communicator's own stream, never on torch's default (the legacy null) stream -- a
        # communicator that has run on the null stream aborts the process when it is later used inside a capture
        pending = self.native.launch_async(fn, tensors)
        if not self.overlap:
            pending.wait()
            return _Works([])
        return pending

    def _a2a(self, recv, send):
        A2A_STATS["calls"] += 1
        A2A_STATS["bytes"] += send.numel() * send.element_size()
        # round 5, opt-in (SC_MPU_A2A=peer / peer_exchange.prefer_peer()): direct stores into the peers' windows, three plain
        # engine launches on the current stream (mpu/peer_exchange.py); anything it cannot take falls through
        if peer_exchange.wanted() and send.is_cuda:
            px = peer_exchange.get(self.group, send.numel() * send.element_size())
            if px is not None and (send.numel() * send.element_size() // self.P) % 16 == 0:
                px.all_to_all(send, recv, torch.cuda.current_stream().cuda_stream)
                return _Works([])
        if self.native is not None:
            return self._native(lambda st: self.native.all_to_all(send, recv, st), (send, recv))
        return dist.all_to_all_single(recv, send, group=self.group, async_op=True)

    # send / recv: [P, n, C, rows, rest.., 2] float32, contiguous (block p <-> rank p of the group)
    def exchange(self, send, recv):
        return self._a2a(recv, send)

    # Round 4 -- channel chunks without copies (one sample per rank): block p of a chunk is a contiguous SLAB of a larger
    # tensor (channels c0:c1 of sample / rank p), so the exchange takes P (send slab, receive slab) pairs instead of
    # one contiguous buffer each way.  RCCL: the list form of all-to-all = one grouped send / receive per peer, the
    # same transfers all_to_all_single issues, straight out of / into the slabs.  gloo (CPU tests) has no list form:
    # P - 1 isend / irecv pairs and a local copy of this rank's own slab.
    def exchange_slabs(self, send_slabs, recv_slabs):
        A2A_STATS["calls"] += 1
        A2A_STATS["bytes"] += sum(t.numel() * t.element_size() for t in send_slabs)
        if not (all(t.is_contiguous() for t in send_slabs) and all(t.is_contiguous() for t in recv_slabs)):
            raise ValueError("exchange_slabs: every slab must be contiguous")
        if self.native is not None:
            return self._native(lambda st: self.native.all_to_all_slabs(send_slabs, recv_slabs, st),
                                list(send_slabs) + list(recv_slabs))
        if dist.get_backend(self.group) == "nccl":
            return _Works([dist.all_to_all(list(recv_slabs), list(send_slabs), group=self.group, async_op=True)])
        me = dist.get_rank(self.group)
        ops = []
        for q in range(self.P):
            if q == me:
                recv_slabs[q].copy_(send_slabs[q])
            else:
                peer = dist.get_global_rank(self.group, q) if self.group is not None else q
                ops.append(dist.P2POp(dist.isend, send_slabs[q], peer, group=self.group))
                ops.append(dist.P2POp(dist.irecv, recv_slabs[q], peer, group=self.group))
        return _Works(dist.batch_isend_irecv(ops) if ops else [])


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class _ModeParallelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        ops, P, rows = layer.ops, layer.P, layer.rows
        spatial = list(x.shape[2:])
        kept = list(layer._n_modes)
        b, ci = x.shape[:2]
        co = ci if layer.separable else weight.shape[1]
        rest = kept[1:]
        w = weight.detach().contiguous()
        by_batch = layer._by_batch(b)
        chunks = _bounds(b, min(layer._chunks(True), b)) if by_batch else _bounds(ci, min(layer._chunks(False), ci))
        ex = _Exchange(layer._group(), P, rows, kept[0], overlap=len(chunks) > 1)
        ctx.cfg = (layer, spatial, kept, b, ci, co, by_batch)
        dev = x.device

        # ---- forward transform + exchange (split modes, cat batch): the transform writes the send buffer
        #      [P, n, ci', rows, rest] directly, the receive buffer is a slice of the contraction's operand
        xd = x.detach()
        k1 = kept[0]
        if by_batch:
            xhat_all = torch.empty((P * b, ci, rows, *rest, 2), dtype=torch.float32, device=dev)
        else:
            xhat_all = torch.empty((P, ci, rows, *rest, 2), dtype=torch.float32, device=dev)
        single = (not by_batch) and len(chunks) == 1
        pend = []
        for (c0, c1) in chunks:
            send = ops.fwd_sharded(xd[c0:c1] if by_batch else xd[:, c0:c1], kept, P, rows)
            if by_batch:
                recv = xhat_all[P * c0:P * c1].view(P, c1 - c0, ci, rows, *rest, 2)
            elif single:
                recv = xhat_all.view(P, 1, ci, rows, *rest, 2)
            else:                                           # channel chunk: straight into the operand's slabs
                pend.append((ex.exchange_slabs([send[q, 0] for q in range(P)], [xhat_all[q, c0:c1] for q in range(P)]),
                             send, None, c0, c1))
                continue
            pend.append((ex.exchange(send, recv), send, recv, c0, c1))
        for work, _send, recv, c0, c1 in pend:
            work.wait()
        xhat_all = torch.view_as_complex(xhat_all)
        V = layer.emulate_world
        if V > 1:                                           # timing emulation: V x the batch on 1 / V of the rows
            xhat_all = xhat_all.reshape(V * xhat_all.shape[0], ci, rows // V, *rest)

        # ---- contraction on this rank's mode rows, whole batch
        yhat_all = (ops.contract_separable(xhat_all, w) if layer.separable else ops.contract(xhat_all, w)).contiguous()
        if V > 1:
            yhat_all = yhat_all.reshape(-1, co, rows, *rest)

        # ---- exchange back (split batch, cat modes) + zero-padded inverse reading the receive buffer in place
        y = torch.empty((b, co, *spatial), dtype=torch.float32, device=dev)
        yr = torch.view_as_real(yhat_all)
        ochunks = chunks if by_batch else _bounds(co, min(layer._chunks(False), co))
        osingle = (not by_batch) and len(ochunks) == 1
        pend = []
        for (c0, c1) in ochunks:
            if by_batch:
                send = yr[P * c0:P * c1].view(P, c1 - c0, co, rows, *rest, 2)
            elif osingle:
                send = yr.view(P, 1, co, rows, *rest, 2)
            else:                                           # channel chunk: straight out of the result's slabs
                recv = torch.empty((P, 1, c1 - c0, rows, *rest, 2), dtype=torch.float32, device=dev)
                pend.append((ex.exchange_slabs([yr[q, c0:c1] for q in range(P)], [recv[q, 0] for q in range(P)]),
                             None, recv, c0, c1))
                continue
            recv = torch.empty_like(send)
            pend.append((ex.exchange(send, recv), send, recv, c0, c1))
        bflat = None if bias is None else bias.detach().reshape(-1)
        for work, _send, recv, c0, c1 in pend:
            work.wait()
            if by_batch:
                ops.inv_sharded(recv, bflat, spatial, k1, out=y[c0:c1])
            else:
                ops.inv_sharded(recv, None if bflat is None else bflat[c0:c1], spatial, k1, out=y[:, c0:c1])
        ctx.save_for_backward(xhat_all, w)
        ctx.has_bias = bias is not None
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        layer, spatial, kept, b, ci, co, by_batch = ctx.cfg
        ops, P, rows = layer.ops, layer.P, layer.rows
        xhat_all, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        rest = kept[1:]
        dev = gy.device
        gy = gy.contiguous()
        chunks = _bounds(b, min(layer._chunks(True), b)) if by_batch else _bounds(co, min(layer._chunks(False), co))
        ex = _Exchange(layer._group(), P, rows, kept[0], overlap=len(chunks) > 1)
        k1 = kept[0]

        # ---- adjoint of the inverse transform (+ bias gradient) + exchange (split modes, cat batch)
        if by_batch:
            ghat_all = torch.empty((P * b, co, rows, *rest, 2), dtype=torch.float32, device=dev)
        else:
            ghat_all = torch.empty((P, co, rows, *rest, 2), dtype=torch.float32, device=dev)
        single = (not by_batch) and len(chunks) == 1
        want_b = need_b and ctx.has_bias
        gb_parts, pend = [], []
        for (c0, c1) in chunks:
            send, gb = ops.inv_adjoint_sharded(gy[c0:c1] if by_batch else gy[:, c0:c1], kept, P, rows, want_bias=want_b)
            gb_parts.append(gb)
            if by_batch:
                recv = ghat_all[P * c0:P * c1].view(P, c1 - c0, co, rows, *rest, 2)
            elif single:
                recv = ghat_all.view(P, 1, co, rows, *rest, 2)
            else:
                pend.append((ex.exchange_slabs([send[q, 0] for q in range(P)], [ghat_all[q, c0:c1] for q in range(P)]),
                             send, None, c0, c1))
                continue
            pend.append((ex.exchange(send, recv), send, recv, c0, c1))
        for work, _send, recv, c0, c1 in pend:
            work.wait()
        ghat_all = torch.view_as_complex(ghat_all)
        V = layer.emulate_world
        if V > 1:
            ghat_all = ghat_all.reshape(V * ghat_all.shape[0], co, rows // V, *rest)
        gbias = None
        if want_b:
            # (one chunk -- the default, and the per-rank step of configs[3] -- hands its vector over as it is: torch.stack +
            #  sum were two launches of ~5 us each in a ~0.36 ms step)
            gb_all = gb_parts[0] if len(gb_parts) == 1 else (torch.stack(gb_parts).sum(0) if by_batch else torch.cat(gb_parts))
            gbias = gb_all.reshape(ctx.bias_shape)

        # ---- the two gradient contractions on this rank's mode rows (gW is complete: no all-reduce)
        cbwd = ops.contract_separable_bwd if layer.separable else ops.contract_bwd
        gxhat_all, gw = cbwd(xhat_all, w, ghat_all, need_x, need_w)
        if V > 1 and gxhat_all is not None:
            gxhat_all = gxhat_all.reshape(-1, ci, rows, *rest)

        # ---- exchange back + adjoint of the forward transform reading the receive buffer in place
        gx = None
        if need_x:
            gx = torch.empty((b, ci, *spatial), dtype=torch.float32, device=dev)
            gr = torch.view_as_real(gxhat_all.contiguous())
            ichunks = chunks if by_batch else _bounds(ci, min(layer._chunks(False), ci))
            isingle = (not by_batch) and len(ichunks) == 1
            pend = []
            for (c0, c1) in ichunks:
                if by_batch:
                    send = gr[P * c0:P * c1].view(P, c1 - c0, ci, rows, *rest, 2)
                elif isingle:
                    send = gr.view(P, 1, ci, rows, *rest, 2)
                else:
                    recv = torch.empty((P, 1, c1 - c0, rows, *rest, 2), dtype=torch.float32, device=dev)
                    pend.append((ex.exchange_slabs([gr[q, c0:c1] for q in range(P)], [recv[q, 0] for q in range(P)]),
                                 None, recv, c0, c1))
                    continue
                recv = torch.empty_like(send)
                pend.append((ex.exchange(send, recv), send, recv, c0, c1))
            for work, _send, recv, c0, c1 in pend:
                work.wait()
                ops.fwd_adjoint_sharded(recv, spatial, k1, out=gx[c0:c1] if by_batch else gx[:, c0:c1])
        return gx, gw, gbias, None


class ModeParallelSpectralConv(BaseSpectralConv):
    """SpectralConv whose first mode dim is sharded across the model-parallel group.

    Constructor arguments follow SpectralConv.  ``ops`` (tests only) replaces the local stages (an object with the
    interface of engine.EngineRawOps); ``comm_chunks`` = pieces the exchange is pipelined in (None: 4 batch chunks,
    or one piece when a rank holds a single sample).

    The shard layout follows ``max_n_modes`` (= the construction-time ``n_modes``, or the explicit ``max_n_modes``
    argument): rank p owns rows [p rows, (p + 1) rows) of the STORED weight's first mode dim.  ``n_modes`` may be
    lowered at run time (incremental training, fno_block.py:460-464, incremental.py:183-259), the grid may be smaller
    than the modes and ``forward(x, output_shape)`` / ``resolution_scaling_factor`` may change the resolution
    (spectral_convolution.py:465-559): those calls take ``_forward_general`` -- the used centred sub-block of the
    weight stays where it is stored, the kept spectrum rows travel to the ranks that own their weight rows (embedded in
    the fixed rows-of-max wire layout, zero elsewhere), the other mode dims use the sub-block's columns -- built from the
    autograd stages of the single-GPU layer (``agops``: engine.EngineOps) and mappings.all_to_all.  It is the
    functional route (one copy on each side of each exchange); the copy-free pipelined route serves the full block on
    an unchanged grid, which is what the BASELINE configs time.  ``complex_data=True`` (round 4) takes the same general
    route with complex-to-complex transforms and the reference's centring / last-dim rules (modes.kept_block_complex,
    analysis_freqs, synthesis_freqs).

    Weights: dense (``weight``: this rank's mode rows of the (Cin, Cout, modes...) tensor; ``separable=True``: of the
    (C, modes...) tensor, spectral_convolution.py:49-52), or ``factorization`` "tucker" / "cp" / "tt"
    (spectral_convolution.py:55-132): the core / CP weights and every factor that does not carry the first mode dim are
    REPLICATED, the factor (TT: core) of the first mode dim is sharded by rows like the dense weight; a rank contracts
    with the dense block rebuilt from its shard (1 / P of the reconstruction work), the gradients of the replicated
    parameters are partial sums over this rank's modes and are summed over the group by ``reduce_replicated_grads``.
    (Round 3: CP, TT and separable joined dense and Tucker.)"""

    def __init__(self, in_channels, out_channels, n_modes, bias=True, init_std="auto",
                 fft_norm="forward", device=None, engine_flags=0, group=None, ops=None, comm_chunks=None,
                 factorization=None, rank=0.5, separable=False, max_n_modes=None, resolution_scaling_factor=None,
                 agops=None, chunk_dim=None, emulate_world=1, **unused):
        super().__init__(device=device)
        # complex_data (spectral_convolution.py:439-441, 475-479, 514-517, 536-538): complex-to-complex transforms in every
        # dim, every dim centred -- runs on the general route (round 4)
        self.complex_data = bool(unused.pop("complex_data", False))
        if chunk_dim not in (None, "batch", "channels"):
            raise ValueError("chunk_dim: None (batch chunks unless a rank holds one sample), 'batch' or 'channels'")
        self.chunk_dim = chunk_dim
        # TIMING ONLY (bench.py --emulate-world V on one device): the contraction runs on 1 / V of the mode rows for V
        # times the local batch -- the per-rank load of a V-rank group -- by reinterpreting the exchanged spectrum's bytes;
        # transforms and exchanges are the real ones of this rank, the RESULTS ARE NOT THE LAYER'S
        self.emulate_world = max(1, int(emulate_world))
        fac = (factorization or "dense").lower()
        if fac not in ("dense", "tucker", "cp", "tt"):
            raise NotImplementedError("mode-parallel layer: dense, Tucker, CP or TT weights")
        if separable and fac != "dense":
            raise NotImplementedError("mode-parallel layer: separable weights are dense")
        if separable and in_channels != out_channels:
            raise ValueError("To use separable Fourier Conv, in_channels must be equal to out_channels, "
                             f"but got in_channels={in_channels} and out_channels={out_channels}")   # :347-353
        self.factorization, self.separable = fac, bool(separable)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._n_modes = halve_last_mode(n_modes, self.complex_data)
        # spectral_convolution.py:317-321: an explicit max_n_modes is stored UN-halved; None = the (halved) n_modes
        self.max_n_modes = list(self._n_modes) if max_n_modes is None else [int(v) for v in max_n_modes]
        self.order = len(self._n_modes)
        if self.order < 2:
            raise NotImplementedError("mode sharding needs >= 2 spatial dims (dim 0 is sharded)")
        if len(self.max_n_modes) != self.order or any(n > m for n, m in zip(self._n_modes, self.max_n_modes)):
            raise ValueError(f"n_modes {self._n_modes} exceeds max_n_modes {self.max_n_modes}")
        if resolution_scaling_factor is not None:
            from ..spectral_conv import _validate_scaling_factor
            resolution_scaling_factor = _validate_scaling_factor(resolution_scaling_factor, self.order)
        self.resolution_scaling_factor = resolution_scaling_factor
        self.engine_flags = engine_flags
        self._agops = agops
        self.fft_norm = fft_norm
        self.group = group
        # pieces the exchange is pipelined in: None = 4 batch chunks when a rank holds >= 2 samples, ONE piece when it
        # holds a single sample (the step is host-issue bound there: every extra all-to-all call costs ~50 us of host
        # time, profiles/r04_modeshard_host.txt); an explicit number applies to both cases (channel chunks: copy-free
        # slabs, round 4)
        self.comm_chunks = None if comm_chunks is None else max(1, int(comm_chunks))
        self.P = comm.get_model_parallel_size() if group is None else dist.get_world_size(group)
        self.rank = comm.get_model_parallel_rank() if group is None else dist.get_rank(group)
        # rows of the first mode dim per rank; k1 not divisible by P: the last rank(s) carry zero rows
        self.rows = -(-self.max_n_modes[0] // self.P)
        if self.emulate_world > 1 and (fac != "dense" or separable or self.rows % self.emulate_world):
            raise ValueError("emulate_world: dense weights, mode rows divisible by the emulated group size")
        if init_std == "auto":
            init_std = (2 / (in_channels + out_channels)) ** 0.5
        live = min(self.rows, max(0, self.max_n_modes[0] - self.rank * self.rows))
        self.core = self.cp_weights = None
        if fac == "dense":
            lead = (in_channels,) if self.separable else (in_channels, out_channels)
            w = torch.empty(*lead, self.rows // self.emulate_world, *self.max_n_modes[1:], dtype=torch.cfloat, device=device)
            w.normal_(0, init_std)
            with torch.no_grad():
                if self.emulate_world == 1:
                    w[(slice(None),) * len(lead) + (slice(live, None),)] = 0
            self.weight = nn.Parameter(w)
            self.weight.mode_sharded = True          # exclude from data-parallel all-reduce within the group
        else:
            # factorized weights (spectral_convolution.py:55-132): the ranks follow from the FULL shape (tensorly's
            # rules, factorized.py), every factor that does not carry the first mode dim is REPLICATED, the one that
            # does (index 2) is sharded by rows like the dense weight; a rank contracts with the dense block rebuilt
            # from its shard (1 / P of the reconstruction work)
            from ..factorized import FactorList, cp_rank, tt_rank, tucker_rank
            full_shape = [in_channels, out_channels, *self.max_n_modes]
            sizes = [in_channels, out_channels, self.rows, *self.max_n_modes[1:]]
            self.weight = None
            mk = lambda *sh, std: nn.Parameter(torch.empty(*sh, dtype=torch.cfloat, device=device).normal_(0, std))
            if fac == "tucker":
                ranks = tucker_rank(full_shape, rank)
                std_f = (init_std / math.prod(math.sqrt(r) for r in ranks)) ** (1.0 / (len(full_shape) + 1))
                self.core = mk(*ranks, std=std_f)
                self.factors = FactorList([mk(n, r, std=std_f) for n, r in zip(sizes, ranks)])
            elif fac == "cp":
                r = cp_rank(full_shape, rank)
                std_f = (init_std / math.sqrt(r)) ** (1.0 / len(full_shape))
                self.cp_weights = nn.Parameter(torch.ones(r, dtype=torch.cfloat, device=device))
                self.factors = FactorList([mk(n, r, std=std_f) for n in sizes])
            else:
                ranks = tt_rank(full_shape, rank)                                  # r_0 .. r_n, r_0 = r_n = 1
                std_f = (init_std / math.prod(ranks)) ** (1.0 / len(full_shape))
                self.factors = FactorList([mk(ranks[i], n, ranks[i + 1], std=std_f) for i, n in enumerate(sizes)])
            with torch.no_grad():
                f2 = self.factors[2]
                (f2[:, live:] if fac == "tt" else f2[live:]).zero_()
            self.factors[2].mode_sharded = True
        self.bias = nn.Parameter(init_std * torch.randn(out_channels, *(1,) * self.order, device=device)) \
            if bias else None
        if ops is None:
            from ..engine import EngineRawOps
            ops = EngineRawOps(fft_norm, engine_flags)
        self.ops = ops

    def _group(self):
        return self.group if self.group is not None else comm.get_model_parallel_group()

    def _by_batch(self, b):
        """the dim the exchange is chunked over: the local batch, or -- one sample per rank -- the channels"""
        if self.chunk_dim == "channels":
            if b != 1:
                raise ValueError("chunk_dim='channels' is the one-sample-per-rank layout (local batch 1)")
            return False
        return self.chunk_dim == "batch" or b >= 2 or self.P == 1

    def _chunks(self, by_batch):
        if self.comm_chunks is not None:
            return self.comm_chunks
        return 4 if by_batch else 1

    @property
    def n_modes(self):
        return self._n_modes

    @n_modes.setter
    def n_modes(self, value):
        # spectral_convolution.py:400-415; the shard layout follows max_n_modes and does not move
        nm = halve_last_mode(value, self.complex_data)
        if len(nm) != self.order or any(n > m for n, m in zip(nm, self.max_n_modes)):
            raise ValueError(f"n_modes {nm} exceeds max_n_modes {self.max_n_modes} (the stored, sharded weight)")
        self._n_modes = nm

    def _out_shape(self, spatial, output_shape):
        if output_shape is not None:
            return [int(v) for v in output_shape]
        if self.resolution_scaling_factor is not None:
            return [round(s * r) for (s, r) in zip(spatial, self.resolution_scaling_factor)]
        return list(spatial)

    def transform(self, x, output_shape=None):
        """the skip path's resize (spectral_convolution.py:383-398): purely local, every rank resizes its batch shard"""
        spatial = list(x.shape[2:])
        out_shape = self._out_shape(spatial, output_shape)
        if out_shape == spatial:
            return x
        from ..spectral_conv import SpectralConv
        return SpectralConv._resample(self, x, out_shape)

    def forward(self, x, output_shape=None):
        spatial = list(x.shape[2:])
        if x.shape[1] != self.in_channels:
            raise ValueError(f"input has {x.shape[1]} channels, the layer expects {self.in_channels}")
        out_shape = self._out_shape(spatial, output_shape)
        if self.complex_data:
            from ..modes import kept_block_complex
            kept, w_start = kept_block_complex(spatial, self._n_modes, self.max_n_modes)
            return self._forward_general(x, spatial, out_shape, kept, w_start)
        kept, w_start = kept_block(spatial, self._n_modes, self.max_n_modes)
        if kept != list(self.max_n_modes) or out_shape != spatial:
            return self._forward_general(x, spatial, out_shape, kept, w_start)
        w = self._dense_block()
        if self.P == 1 and not dist.is_initialized():
            return _single_rank(self, x, spatial, w)
        return _ModeParallelFn.apply(x, w, self.bias, self)

    def _forward_general(self, x, spatial, out_shape, kept, w_start):
        """Runtime-reduced ``n_modes``, a grid smaller than the modes, or a different output grid (class docstring).
        kept / w_start: extents and first rows of the used centred sub-block of the STORED weight (modes.kept_block);
        the synthesis map places the kept rows on the output grid with the reference's end-padding quirk
        (modes.synthesis_freqs)."""
        from .. import modes as _modes
        from .mappings import all_to_all
        cplx = self.complex_data
        ag = self._agops
        if ag is None:
            from ..engine import EngineOps, SC_PLAN_COMPLEX
            ag = self._agops = EngineOps(self.fft_norm, self.engine_flags | (SC_PLAN_COMPLEX if cplx else 0))
        P, rows = self.P, self.rows
        fa = _modes.analysis_freqs(spatial, kept, cplx)                     # complex data: the reference's last-dim quirk
        fs, real_col = _modes.synthesis_freqs(spatial, out_shape, kept, cplx)
        xhat = ag.forward_transform(x, kept, fa)                            # (n, Cin, k1', rest')
        w = self._dense_block()                                             # this rank's rows of the stored weight
        lead = 1 if self.separable else 2
        cols = tuple(slice(s0, s0 + k) for s0, k in zip(w_start[1:], kept[1:]))
        w = w[(slice(None),) * (lead + 1) + cols]                           # sub-block columns of the other mode dims
        if P > 1:
            # kept row r multiplies stored weight row w_start[0] + r: it travels to the rank that owns that row
            n = x.shape[0]
            wire = xhat.new_zeros((n, xhat.shape[1], P * rows, *kept[1:]))
            wire[:, :, w_start[0]:w_start[0] + kept[0]] = xhat
            xloc = all_to_all(wire, 2, 0, self._group())                    # (P n, Cin, rows, rest')
            yloc = ag.contract_separable(xloc, w) if self.separable else ag.contract(xloc, w.contiguous())
            ywire = all_to_all(yloc, 0, 2, self._group())                   # (n, Cout, P rows, rest')
            yhat = ywire[:, :, w_start[0]:w_start[0] + kept[0]].contiguous()
        else:
            wr = w[(slice(None),) * lead + (slice(w_start[0], w_start[0] + kept[0]),)].contiguous()
            yhat = ag.contract_separable(xhat, wr) if self.separable else ag.contract(xhat, wr)
        if cplx:                          # a real bias added to a complex field: elementwise glue (:567-568)
            y = ag.inverse_transform(yhat, None, out_shape, fs)
            return y if self.bias is None else y + self.bias
        return ag.inverse_transform(yhat, self.bias, out_shape, fs, real_col)

    def _dense_block(self):
        """this rank's (Cin, Cout, rows, ...) block -- (C, rows, ...) when separable -- with autograd to the parameters"""
        if self.factorization == "dense":
            return self.weight
        if self.factorization == "tucker":
            return self.ops.tucker_dense(self.core, list(self.factors))
        if self.factorization == "cp":
            return self.ops.cp_dense(self.cp_weights, list(self.factors))
        return self.ops.tt_dense(list(self.factors))

    # ---- helpers for the training loop -----------------------------------------------------------
    def replicated_parameters(self):
        ps = [] if self.bias is None else [self.bias]
        if self.factorization != "dense":
            ps += [q for q in (self.core, self.cp_weights) if q is not None]
            ps += [f for i, f in enumerate(self.factors) if i != 2]
        return ps

    def reduce_replicated_grads(self):
        """Sum the gradients of the replicated parameters over the model-parallel group: the bias (every rank saw a
        different batch shard) and, for Tucker weights, the core and the unsharded factors (every rank contracted
        different modes).  The sharded weight / factor rows need nothing."""
        if self.P > 1:
            native = rccl_native.get(self._group())
            for q in self.replicated_parameters():
                if q.grad is not None:
                    g = torch.view_as_real(q.grad) if q.grad.is_complex() else q.grad
                    if native is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous():
                        native.all_reduce_sum_on_current(g)
                    else:
                        dist.all_reduce(g, group=self._group())

    def sync_replicated_parameters(self, src=0):
        """Broadcast the replicated parameters from group rank ``src`` (after a per-rank random init)."""
        if self.P > 1:
            grp = self._group()
            for q in self.replicated_parameters():
                t = torch.view_as_real(q.data) if q.is_complex() else q.data
                dist.broadcast(t, dist.get_global_rank(grp, src) if grp is not None else src, group=grp)

    def load_full_state_dict(self, state_dict, prefix=""):
        """Load the UNSHARDED parameters of a reference / single-GPU layer (state-dict keys of
        neuralop's SpectralConv and of neuraloperator_amd.SpectralConv: ``weight.tensor`` -- or a bare ``weight`` --
        for dense weights, ``weight.core`` + ``weight.factors.factor_{i}`` for Tucker, ``bias``), keeping this rank's
        mode rows of the dense weight / of the first mode dim's factor.  Complex tensors stored as real
        (..., 2) views are accepted."""
        def get(*names):
            for n in names:
                if prefix + n in state_dict:
                    t = state_dict[prefix + n]
                    if not t.is_complex() and t.shape[-1:] == (2,) and t.dtype.is_floating_point and n != "bias":
                        t = torch.view_as_complex(t.contiguous())
                    return t
            raise KeyError(f"none of {[prefix + n for n in names]} in the state dict")
        with torch.no_grad():
            if self.factorization == "dense":
                full = get("weight.tensor", "weight").to(self.weight.device)
                md = 1 if self.separable else 2                      # the first mode dim of the stored tensor
                self.weight.copy_(self._shard_rows(full, md))
            else:
                if self.core is not None:
                    self.core.copy_(get("weight.core", "core"))
                if self.cp_weights is not None:
                    self.cp_weights.copy_(get("weight.weights", "weights", "cp_weights"))
                for i, f in enumerate(self.factors):
                    full = get(f"weight.factors.factor_{i}", f"weight.factors.{i}", f"factors.factor_{i}", f"factors.{i}")
                    full = full.to(f.device)
                    f.copy_(self._shard_rows(full, 1 if self.factorization == "tt" else 0) if i == 2 else full)
            if self.bias is not None and (prefix + "bias") in state_dict:
                self.bias.copy_(state_dict[prefix + "bias"].reshape(self.bias.shape))
        return self

    def _shard_rows(self, full, dim):
        """rows [rank * rows, (rank + 1) * rows) of ``full`` along ``dim`` (zero rows past the end)"""
        k1 = full.shape[dim]
        shape = list(full.shape)
        shape[dim] = self.rows
        out = full.new_zeros(shape)
        live = min(self.rows, max(0, k1 - self.rank * self.rows))
        if live > 0:
            out.narrow(dim, 0, live).copy_(full.narrow(dim, self.rank * self.rows, live))
        return out

    @staticmethod
    def shard_tucker_factor(full_factor, rank, world):
        """Rows of the first-mode-dim factor (k1, r) that rank ``rank`` owns (zero rows past k1)."""
        k1 = full_factor.shape[0]
        rows = -(-k1 // world)
        out = full_factor.new_zeros((rows, full_factor.shape[1]))
        live = min(rows, max(0, k1 - rank * rows))
        if live > 0:
            out[:live] = full_factor[rank * rows:rank * rows + live]
        return out

    @staticmethod
    def shard_dense_weight(full_weight, rank, world):
        """Rows of a full (Cin, Cout, k1, ..) weight that rank ``rank`` owns (zero rows past k1)."""
        k1 = full_weight.shape[2]
        rows = -(-k1 // world)
        out = full_weight.new_zeros((*full_weight.shape[:2], rows, *full_weight.shape[3:]))
        live = min(rows, max(0, k1 - rank * rows))
        if live > 0:
            out[:, :, :live] = full_weight[:, :, rank * rows:rank * rows + live]
        return out


class _SingleRankFn(torch.autograd.Function):
    """P = 1 without a process group: the same stages, nothing on the wire"""

    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        ops = layer.ops
        spatial, kept = list(x.shape[2:]), list(layer._n_modes)
        w = weight.detach().contiguous()
        xhat = ops.fwd(x.detach(), kept)
        yhat = ops.contract_separable(xhat, w) if layer.separable else ops.contract(xhat, w)
        y = ops.inv(yhat, None if bias is None else bias.detach().reshape(-1), spatial)
        ctx.save_for_backward(xhat, w)
        ctx.cfg = (layer, spatial, kept, None if bias is None else tuple(bias.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        layer, spatial, kept, bshape = ctx.cfg
        xhat, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        gh, gb = layer.ops.inv_adjoint(gy.contiguous(), kept, want_bias=need_b and bshape is not None)
        cbwd = layer.ops.contract_separable_bwd if layer.separable else layer.ops.contract_bwd
        gxh, gw = cbwd(xhat, w, gh, need_x, need_w)
        gx = layer.ops.fwd_adjoint(gxh, spatial) if need_x else None
        return gx, gw, None if gb is None else gb.reshape(bshape), None


def _single_rank(layer, x, spatial, w):
    return _SingleRankFn.apply(x, w, layer.bias, layer)
