// device side of scripts/ubench_pmlp.cpp: ONE instantiation of k_pmlp_bwd at the metric block's channel counts as a
// stand-alone code object (scripts/pmlp_variants.sh: seconds per build variant instead of minutes for the library)
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "../neuraloperator_amd/csrc/sc_kernels_pmlp.h"
#ifndef PK_NW
#define PK_NW 4
#endif
#ifndef PK_LIN
#define PK_LIN false
#endif
template __global__ void k_pmlp_bwd<2, 1, 2, true, 1, PK_NW, PK_LIN>(PmlpBwdArgs);
