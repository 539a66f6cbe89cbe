// device side of scripts/ubench_plinx.cpp: instantiations of k_plinx_bwd of a 128-channel block as a stand-alone code object
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "../neuraloperator_amd/csrc/sc_kernels_plinx.h"
template __global__ void k_plinx_bwd<4, 4, 2, false>(PlinxArgs, int);
template __global__ void k_plinx_bwd<2, 4, 4, false>(PlinxArgs, int);
template __global__ void k_plinx_bwd<4, 2, 2, false>(PlinxArgs, int);
