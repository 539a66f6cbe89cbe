"""Build measurement variants of libsc_engine.so (ablations / A-B switches of the MFMA DFT passes)
into scripts/abl/.  Usage: python scripts/mdft_variants.py"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuraloperator_amd.csrc import build as b

VARIANTS = {
    "directst": ["SC_MDFT_C2R_DIRECT_STORE"],
    "nostore": ["SC_MDFT_ABL_NOSTORE"],
    "nomfma": ["SC_MDFT_ABL_NOMFMA"],
    "nomfma_nostore": ["SC_MDFT_ABL_NOMFMA", "SC_MDFT_ABL_NOSTORE"],
    "plainstore": ["SC_MDFT_PLAIN_STORE"],
    "stagent": ["SC_STAGE_NT_LOAD", "SC_STAGE_NT_STORE"],   # k_mdft_*_stage: non-temporal tile loads / stores
    "nostage": ["SC_MDFT_NOSTAGE"],
}
if len(sys.argv) > 1:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[1:]}
out_dir = os.path.join(ROOT, "scripts", "abl")
os.makedirs(out_dir, exist_ok=True)
with ThreadPoolExecutor(4) as ex:
    list(ex.map(lambda kv: b.build(out=os.path.join(out_dir, f"libsc_{kv[0]}.so"), defines=kv[1], verbose=False),
                VARIANTS.items()))
print("built", sorted(os.listdir(out_dir)))
