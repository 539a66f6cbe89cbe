"""neuraloperator_amd -- MI355X-native SpectralConv engine (drop-in ``conv_module`` for FNO/TFNO).

Only the hot path of neuralop.layers.spectral_convolution lives here:
hand-written HIP kernels behind a C-ABI (include/sc_engine.h, neuraloperator_amd/csrc) and
the Python module that mirrors the reference's plug-in interface.
"""
from .spectral_conv import BaseSpectralConv, SpectralConv  # noqa: F401
from .factorized import CPWeight, DenseWeight, SpectralWeight, TTWeight, TuckerWeight  # noqa: F401
from .optim import AdamW  # noqa: F401
from .galore import TensorGaLoreProjector  # noqa: F401
from .spherical import SHT, SphericalConv  # noqa: F401
from .graph import GraphedStep, capture_step  # noqa: F401

__version__ = "0.1.0"
