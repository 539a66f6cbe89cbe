"""Build measurement variants of the engine next to the product library (neuraloperator_amd/libsc_engine_<tag>.so;
*.so is git-ignored but travels to the GPU box):  python scripts/build_variants.py tag=DEF1,DEF2 tag2=DEF ..."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd.csrc import build as b  # noqa: E402


def one(spec):
    tag, _, defs = spec.partition("=")
    out = os.path.join(b.PKG, f"libsc_engine_{tag}.so")
    b.build(out=out, defines=[d for d in defs.split(",") if d], verbose=False)
    return out


with ThreadPoolExecutor(max_workers=4) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print("built", o)
