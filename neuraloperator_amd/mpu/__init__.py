"""Model-parallel utilities: process groups, autograd-aware collectives, and the
mode-parallel and the spatially decomposed spectral convolution (RCCL all-to-all over xGMI), multigrid patching."""
from . import comm, peer_exchange, rccl_native  # noqa: F401
from .mappings import (all_to_all, copy_to_model_parallel_region,  # noqa: F401
                       gather_from_model_parallel_region, reduce_from_model_parallel_region,
                       scatter_to_model_parallel_region)
from .mode_parallel import ModeParallelSpectralConv  # noqa: F401
from .spatial_parallel import SpatialParallelSpectralConv  # noqa: F401
from .patching import MultigridPatching2D, make_patches  # noqa: F401
