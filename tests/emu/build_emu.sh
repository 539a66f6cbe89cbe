#!/bin/sh
# TEST INFRASTRUCTURE ONLY: host-emulation build of the engine (see sc_device.h).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
g++ -O2 -std=c++17 -fPIC -shared -pthread -DSC_EMU=1 -DSC_DIAG=1 -x c++ \
    "$ROOT/neuraloperator_amd/csrc/sc_engine.cpp" "$HERE/sc_emu_runtime.cpp" \
    -o "$HERE/libsc_engine_emu.so"
echo "built $HERE/libsc_engine_emu.so"
