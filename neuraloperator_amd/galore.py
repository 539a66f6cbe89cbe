"""Tensor-GaLore gradient projection for the spectral weights (reference
/root/reference/neuralop/training/tensor_galore_projector.py:7-106, used by training/adamw.py:139-196).

The gradient tensor G of a spectral weight (Cin, Cout, modes...) is projected onto the leading mode-wise subspaces of a
Tucker decomposition, ``G_low = G x_0 U_0^H x_1 U_1^H ...``; AdamW keeps its moments in that low-rank space and maps
the normalised update back, ``U x_d``.  Same class name, constructor and methods as the reference.

The decomposition itself lives in tensorly (``tensorly.decomposition.tucker``, absent here and un-vendored upstream):
``tucker_hooi`` restates its published algorithm -- higher-order orthogonal iteration from an SVD initialisation,
stopping when the relative reconstruction error moves by less than ``tol`` -- and is therefore "parity unpinned" against
tensorly itself (singular vectors are only defined up to a phase anyway; the projected update depends on the
SUBSPACES only, which the tests check).  The SVDs are torch.linalg.svd (a few per ``update_proj_gap`` steps, off the
per-step path).  The per-step work -- the mode products of ``project`` / ``project_back`` on the weight-sized
gradient -- runs on the ENGINE for complex64 CUDA tensors (round 3): ``t x_d U`` is one sc_modegemm launch with the
factor as the mode-independent operand (k_modegemm_bfac: lanes = everything behind dim d), the last dim through one
transposing copy; other dtypes / CPU tensors use torch.matmul on the unfoldings (same arithmetic order per output:
an r-ordered sum)."""
import torch

from .factorized import tucker_rank


def _unfold(t, mode):
    return torch.movedim(t, mode, 0).reshape(t.shape[mode], -1)


def _engine_mode_dot(t, matrix, mode, transpose):
    """t x_mode matrix on the engine (complex64, CUDA): C[p, q, m] = sum_r A[p, r, m] opB(B)[r, q] with
    p = the dims before ``mode``, m = the dims behind it (the lanes), B = matrix^T (conj(matrix) when ``transpose``)."""
    from . import engine
    t = t.contiguous()
    shape = list(t.shape)
    n = shape[mode]
    lead = 1
    for v in shape[:mode]:
        lead *= int(v)
    tail = 1
    for v in shape[mode + 1:]:
        tail *= int(v)
    # out[.., q, ..] = sum_r M[q, r] t[.., r, ..] with M = matrix (forward) or matrix^H (transpose):
    # B[r, q] = M[q, r] = matrix[q, r]            -> matrix viewed with swapped strides, no conjugate
    #         = conj(matrix[r, q]) (transpose)     -> matrix itself, conjugated by the kernel
    mtx = matrix.to(torch.complex64)
    b = mtx if transpose else mtx.transpose(0, 1)
    q = int(b.shape[1])
    if tail > 1:
        out = engine._raw_mode_gemm(t.reshape(lead, n, tail), b, tail, False, bool(transpose))
        return out.reshape(*shape[:mode], q, *shape[mode + 1:])
    # last dim: nothing behind it to put on the lanes -- one transposing copy makes the leading dims the lanes
    tt = t.reshape(lead, n).transpose(0, 1).contiguous().reshape(1, n, lead)
    out = engine._raw_mode_gemm(tt, b, lead, False, bool(transpose))                    # [1, q, lead]
    return out.reshape(q, lead).transpose(0, 1).reshape(*shape[:mode], q)


def mode_dot(t, matrix, mode, transpose=False):
    """t x_mode matrix: contracts dim ``mode`` of t with the columns of ``matrix`` (rows, with the conjugate, when
    ``transpose``: the adjoint of the forward product)."""
    if t.is_cuda and t.dtype == torch.complex64 and torch.is_complex(matrix) and t.numel() >= 1024:
        return _engine_mode_dot(t, matrix, mode % t.ndim, transpose)
    m = matrix.conj().transpose(0, 1) if transpose else matrix
    moved = torch.movedim(t, mode, 0)
    out = (m @ moved.reshape(moved.shape[0], -1)).reshape(m.shape[0], *moved.shape[1:])
    return torch.movedim(out, 0, mode)


def multi_mode_dot(t, matrices, transpose=False, skip=None):
    for d, m in enumerate(matrices):
        if d != skip:
            t = mode_dot(t, m, d, transpose=transpose)
    return t


def tucker_hooi(tensor, rank, init="svd", n_iter_max=100, tol=1e-4):
    """Tucker decomposition by higher-order orthogonal iteration.  rank: float (fraction of the parameters to keep,
    tensorly's rule: factorized.tucker_rank), int (every mode) or one int per mode.  init: "svd" or a list of factor
    matrices (warm restart).  Returns (core, [U_d with orthonormal columns])."""
    nd = tensor.ndim
    if isinstance(rank, float):
        ranks = tucker_rank(list(tensor.shape), rank)
    elif isinstance(rank, int):
        ranks = [rank] * nd
    else:
        ranks = [int(r) for r in rank]
    ranks = [min(r, s) for r, s in zip(ranks, tensor.shape)]
    if isinstance(init, str):
        if init != "svd":
            raise ValueError(f"init={init!r}: 'svd' or a list of factors")
        factors = []
        for d in range(nd):
            u, _, _ = torch.linalg.svd(_unfold(tensor, d), full_matrices=False)
            factors.append(u[:, :ranks[d]].contiguous())
    else:
        factors = [f.to(tensor.dtype) for f in init]
    norm = torch.linalg.norm(tensor)
    prev = None
    for _ in range(n_iter_max):
        for d in range(nd):
            partial = multi_mode_dot(tensor, factors, transpose=True, skip=d)
            u, _, _ = torch.linalg.svd(_unfold(partial, d), full_matrices=False)
            factors[d] = u[:, :ranks[d]].contiguous()
        core = multi_mode_dot(tensor, factors, transpose=True)
        # ||T - core x U||^2 = ||T||^2 - ||core||^2 for orthonormal factors
        err = torch.sqrt(torch.clamp(norm ** 2 - torch.linalg.norm(core) ** 2, min=0.0)) / norm
        if prev is not None and abs(float(prev) - float(err)) < tol:
            break
        prev = err
    return core, factors


class TensorGaLoreProjector:
    def __init__(self, rank, update_proj_gap=200, scale=1.0, tucker_n_iter_max=10, warm_restart=False,
                 activation_checkpoint=False):
        self.rank = rank
        self.update_proj_gap = update_proj_gap
        self.scale = scale
        self.warm_restart = warm_restart
        self.tucker_n_iter_max = tucker_n_iter_max
        self.activation_checkpoint = activation_checkpoint        # accepted; nothing here holds activations
        self.proj_tensor = None

    def project(self, full_rank_grad, iter):
        # as upstream (:66-71): the subspace is computed when none exists yet and the step is a multiple of the gap,
        # i.e. on the first step, and then kept
        if self.proj_tensor is None and iter % self.update_proj_gap == 0:
            self.proj_tensor = self.get_projection_tensor(full_rank_grad, self.rank)
        self.proj_tensor = [f.to(full_rank_grad.device) for f in self.proj_tensor]
        return self.transform(self.proj_tensor, full_rank_grad)

    def project_back(self, low_rank_grad):
        return self.inverse_transform(self.proj_tensor, low_rank_grad) * self.scale

    def get_projection_tensor(self, weights, rank):
        t = weights.data
        if torch.is_complex(t) and t.dtype != torch.cfloat:
            t = t.cfloat()
        init = self.proj_tensor if (self.warm_restart and self.proj_tensor is not None) else "svd"
        _, factors = tucker_hooi(t, rank, init=init)
        return factors

    def transform(self, factors, x):
        return multi_mode_dot(x, factors, transpose=True)

    def inverse_transform(self, factors, x):
        return multi_mode_dot(x, factors)
