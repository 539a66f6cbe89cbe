"""TEST INFRASTRUCTURE ONLY -- import the *verbatim* reference SpectralConv.

Works only where /root/reference exists (the build container).  The reference
package's ``__init__`` files pull in wandb / zencfg / h5py which are absent, so
bare package objects are registered for ``neuralop`` and ``neuralop.layers``
(their ``__init__.py`` never runs) and the two files the hot path needs are
loaded from where they lie:

    neuralop/utils.py                          (validate_scaling_factor)
    neuralop/layers/spectral_convolution.py    (SpectralConv + _contract_*)

plus their siblings einsum_utils.py / base_spectral_conv.py / resample.py that
the file imports relatively.  No reference source is copied into this repo.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("NEURALOP_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(
        REFERENCE_ROOT, "neuralop", "layers", "spectral_convolution.py"))


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns the verbatim ``neuralop.layers.spectral_convolution`` module."""
    if not available():
        raise RuntimeError(f"reference not present under {REFERENCE_ROOT}")
    from . import tl_stub

    tl_stub.install()
    name = "neuralop.layers.spectral_convolution"
    if name in sys.modules:
        return sys.modules[name]
    root = os.path.join(REFERENCE_ROOT, "neuralop")
    if "neuralop" not in sys.modules:
        pkg = types.ModuleType("neuralop")
        pkg.__path__ = [root]
        sys.modules["neuralop"] = pkg
    if "neuralop.layers" not in sys.modules:
        lay = types.ModuleType("neuralop.layers")
        lay.__path__ = [os.path.join(root, "layers")]
        sys.modules["neuralop.layers"] = lay
    if "neuralop.utils" not in sys.modules:
        _load("neuralop.utils", os.path.join(root, "utils.py"))
    return _load(name, os.path.join(root, "layers", "spectral_convolution.py"))


def load_reference_adamw():
    """The verbatim ``neuralop.training.adamw`` module.  Its Tensor-GaLore projector imports tensorly's
    decompositions (absent); a placeholder module takes its place -- the non-GaLore branch never touches it."""
    if not available():
        raise RuntimeError(f"reference not present under {REFERENCE_ROOT}")
    name = "neuralop.training.adamw"
    if name in sys.modules:
        return sys.modules[name]
    root = os.path.join(REFERENCE_ROOT, "neuralop")
    for pkg, sub in (("neuralop", ""), ("neuralop.training", "training")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(root, sub) if sub else root]
            sys.modules[pkg] = m
    if "neuralop.training.tensor_galore_projector" not in sys.modules:
        stub = types.ModuleType("neuralop.training.tensor_galore_projector")
        stub.TensorGaLoreProjector = type("TensorGaLoreProjector", (), {})
        sys.modules["neuralop.training.tensor_galore_projector"] = stub
    return _load(name, os.path.join(root, "training", "adamw.py"))


def load_reference_fno():
    """The verbatim ``neuralop.models.fno`` module (FNO / TFNO built from the reference's own FNOBlocks, ChannelMLP,
    skip connections, embeddings ...): the real caller of the ``conv_module`` plug-in
    (neuralop/layers/fno_block.py:210-240).  Bare package objects keep the reference's ``__init__`` files (wandb,
    zencfg, h5py ...) from running; every module is imported from where it lies."""
    import importlib

    load_reference()                                   # tensorly / tltorch stubs, neuralop, neuralop.layers, utils
    root = os.path.join(REFERENCE_ROOT, "neuralop")
    if "neuralop.models" not in sys.modules:
        m = types.ModuleType("neuralop.models")
        m.__path__ = [os.path.join(root, "models")]
        sys.modules["neuralop.models"] = m
    return importlib.import_module("neuralop.models.fno")


def load_reference_patching():
    """The verbatim ``neuralop.training.patching`` (MultigridPatching2D, make_patches) with the verbatim
    ``neuralop.mpu`` modules it imports (comm, mappings, helpers)."""
    import importlib

    if not available():
        raise RuntimeError(f"reference not present under {REFERENCE_ROOT}")
    root = os.path.join(REFERENCE_ROOT, "neuralop")
    for pkg, sub in (("neuralop", ""), ("neuralop.training", "training"), ("neuralop.mpu", "mpu")):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(root, sub) if sub else root]
            sys.modules[pkg] = m
    return importlib.import_module("neuralop.training.patching")
