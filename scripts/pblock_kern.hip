// device side of scripts/ubench_pblock.cpp: ONE instantiation of k_pblock_fwd at the metric block's channel counts as a
// stand-alone code object (scripts/pblock_variants.sh: seconds per build variant instead of minutes for the library)
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "../neuraloperator_amd/csrc/sc_kernels_pmlp.h"
template __global__ void k_pblock_fwd<2, 1, 1>(PblockArgs);
