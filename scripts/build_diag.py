"""Build the DIAGNOSTIC variant of the engine library: the same sources with -DSC_DIAG, i.e. with the A-B / measurement
switches of the scripts readable from the environment (SC_F2P_CHUNK_MB, SC_SB_*, SC_*_WGS, SC_FMX_OFF, ... -- see
`SC_DIAG_ENV` in csrc/sc_engine.cpp).  The product library (csrc/build.py) is built WITHOUT it and holds none of those names.
    python scripts/build_diag.py [extra -D defines ...]        -> neuraloperator_amd/libsc_engine_diag.so
    SC_ENGINE_LIB=neuraloperator_amd/libsc_engine_diag.so SC_F2P_CHUNK_MB=96 python scripts/f2p_time.py ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd.csrc import build as b

if __name__ == "__main__":
    out = os.path.join(b.PKG, "libsc_engine_diag.so")
    b.build(force=True, out=out, defines=("SC_DIAG",) + tuple(a[2:] if a.startswith("-D") else a for a in sys.argv[1:]))
    print(out)
