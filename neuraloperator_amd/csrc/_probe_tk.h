#include "sc_kernels_tkchain.h"
__global__ void __launch_bounds__(256, 1) k_probe1(TkcArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const int lane = SC_TID & 63, w = SC_TID >> 6;
  TkAcc a[3];
  for (int k = 0; k < 3; ++k) tk_zero(a[k]);
  tk_multi<3, true, false, false, false, true>(lds + w * 2112, 66, 1, g.u_in, g.R1, 1, 0, 0, g.B, g.R1, g.Ci, lane, a);
  for (int k = 0; k < 3; ++k) tk_store<false>(a[k], lds + 10000 + w * 1216, 38, 0, 16 * k, g.B, g.R1, lane);
}
__global__ void __launch_bounds__(256, 1) k_probe2(TkcArgs g) {
  SC_DYN_SHARED(cf32, lds);
  const int tid = SC_TID;
  sc_f4 vx[16];
  tkc_fetch<16>(g.xhat, g.B * g.Ci, g.M, 0, tid, vx);
  tkc_plant<16>(lds, 2112, 66, g.B * g.Ci, g.Ci, g.inv_ci, tid, vx);
}
