// sc_engine.cpp -- host side of libsc_engine.so: plans, twiddle tables, kernel dispatch and
// the extern "C" entry points declared in include/sc_engine.h.
//
// Built with:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip sc_engine.cpp
// (tests/emu builds the same file with g++ -DSC_EMU; see sc_device.h).
#include <atomic>
#include "../../include/sc_engine.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "sc_device.h"
// Diagnostic / A-B switches of the measurement scripts (kernel shapes, grid sizes, routes switched off): read from the
// environment only in builds with -DSC_DIAG (scripts/build_diag.py, the emulation tier).  The product library holds
// neither the reads nor the names: `strings libsc_engine.so | grep '^SC_'` shows the documented public switches only
// (SC_NO_SIDE_STREAM, SC_PLAN_NO_MX_FFT, SC_TKC), each read once per process except SC_TKC.
#ifdef SC_DIAG
#define SC_DIAG_ENV(name) std::getenv(name)
#else
#define SC_DIAG_ENV(name) ((const char*)nullptr)
#endif
#include "sc_kernels_generic.h"
#include "sc_kernels_fft.h"
#include "sc_kernels_fft3.h"
#include "sc_kernels_fft3mx.h"
#include "sc_kernels_mfma.h"
#include "sc_kernels_gemm8.h"
#include "sc_kernels_mdft.h"
#include "sc_kernels_fft2p.h"
#include "sc_kernels_plane.h"
#include "sc_kernels_plane64.h"
#include "sc_kernels_pmlp.h"
#include "sc_kernels_plinx.h"
#include "sc_kernels_tucker.h"
#include "sc_kernels_sb.h"
#include "sc_kernels_fmx.h"
#include "sc_kernels_tkchain.h"
#include "sc_kernels_peer.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int sc_fail(const std::string& msg) {
  g_last_error = msg;
  return 1;
}

#define SC_CHECK_ARG(cond, msg) \
  do {                          \
    if (!(cond)) return sc_fail(std::string("sc_engine: ") + msg); \
  } while (0)

#define SC_CHECK_HIP(expr)                                                               \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return sc_fail(std::string("sc_engine: HIP error in " #expr ": ") + hipGetErrorString(e_)); \
  } while (0)

static int sc_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return sc_fail(std::string("sc_engine: launch of ") + what + " failed: " + hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct DeviceTable {
  cf32* ptr = nullptr;
  int rows = 0, cols_pad = 0;
};

struct IdxKey {
  int64_t ext[SC_MAX_DIMS], start[SC_MAX_DIMS];
  bool operator<(const IdxKey& o) const { return std::memcmp(this, &o, sizeof(IdxKey)) < 0; }
};

struct sc_plan {
  sc_plan_desc d;
  int nd;
  int64_t n[SC_MAX_DIMS], k[SC_MAX_DIMS];
  int64_t ntot;          // prod n
  int64_t modes;         // prod k  (kept modes per image)
  int64_t dc_index;      // flattened index of the zero frequency inside the kept block (-1: not kept)
  double sf, si;         // forward / inverse norm scales
  std::vector<int64_t> fmap[SC_MAX_DIMS];   // kept row -> FFT index on this grid (SC_FREQ_DROPPED: none)
  bool custom_map = false;
  bool cplx = false;     // SC_PLAN_COMPLEX: last dim is a complex-to-complex pass as well
  DeviceTable cx_fwd[2], cx_inv[2];         // its tables, scales folded in (modes as r2c[] / c2r[])
  int cx_fwd_jt = 0, cx_inv_jt = 0;
  // last axis
  int r2c_jt, c2r_nt;
  DeviceTable r2c[2];    // [SC_FWD_SCALED], [SC_FWD_ADJ_C2R]
  DeviceTable c2r[2];    // [SC_INV_PADDED], [SC_INV_ADJ_R2C]
  // non-last axes
  int ax_fwd_jt[SC_MAX_DIMS], ax_inv_jt[SC_MAX_DIMS];
  DeviceTable ax_fwd[SC_MAX_DIMS], ax_inv[SC_MAX_DIMS];
  // matrix-core versions of the generic passes (sc_kernels_mdft.h): tables in MFMA lane order
  bool mdft = false;
  float* m_r2c[2] = {nullptr, nullptr};
  cf32* m_r2c_tail[2] = {nullptr, nullptr};   // last kept column on the VALU when 2J % 32 == 2
  float* m_c2r[2] = {nullptr, nullptr};
  // LDS-staged generation of the two last-axis passes (small tables only, see sc_kernels_mdft.h)
  float* l_r2c[2] = {nullptr, nullptr};
  cf32* l_r2c_tail[2] = {nullptr, nullptr};
  int l_r2c_ct = 0;                            // MFMA column tiles (tail column excluded)
  // LDS-staged r2c for any width, table in global memory (k_mdft_r2c_stage; built when l_r2c is not)
  float* s_r2c[2] = {nullptr, nullptr};
  cf32* s_r2c_tail[2] = {nullptr, nullptr};
  int s_r2c_ct = 0;
  // ... and the c2r of the widths l_c2r does not take (k_mdft_c2r_stage): lane-major float4, JS2 = step pairs per tile
  float* s_c2r[2] = {nullptr, nullptr};
  int s_c2r_js2 = 0, s_c2r_s = 0;
  float* l_c2r[2] = {nullptr, nullptr};
  int l_c2r_s = 0;                             // LDS row stride of the c2r tile, floats
  float* m_pl_fwd = nullptr;                   // plane form: row-pass table of dim nd-2, forward ([jt][n1][64])
  float* m_pl_inv = nullptr;                   // ... inverse ([row tile][j1][64])
  float* m_ax_fwd[SC_MAX_DIMS] = {nullptr, nullptr, nullptr, nullptr};
  float* m_ax_inv[SC_MAX_DIMS] = {nullptr, nullptr, nullptr, nullptr};
  // fast path (power-of-two 2-D), see sc_kernels_fft.h
  Fft2dPlan fft2d;
  bool fast = false;
  // two-pass factorised route for large power-of-two 2-D grids (sc_kernels_fft2p.h)
  bool f2p = false;
  int f2p_p[2] = {0, 0};        // points per lane of a line along dim 0 / dim 1 (N = 32 P)
  int f2p_k2[2] = {0, 0};       // range of the pruned 32-point stage: kept rows / kept columns
  int f2p_ncb = 0;              // panel blocks of 8 kept columns
  cf32* f2p_tw[2] = {nullptr, nullptr};
  cf32* f2p_w1024 = nullptr;    // exp(-2 pi i m / 1024), m = 0..1023: 1024-point lines on k_f2p_c2r_w1024 / k_f2p_col_inv_w1024 (round 4)
  bool f2p_roww = false;        // rows: N1 = 1024 with the 129 kept columns k = 0..128
  bool f2p_colw = false;        // columns: N0 = 1024 with <= 256 kept rows
  float* f2p_cs_fwd[2] = {nullptr, nullptr};   // [SC_FWD_SCALED], [SC_FWD_ADJ_C2R]
  float* f2p_cs_inv[2] = {nullptr, nullptr};   // [SC_INV_PADDED], [SC_INV_ADJ_R2C]
  // factorised last-two-axes kernels for 128 x 128 planes (sc_kernels_plane.h)
  bool pl128 = false;
  bool pl64 = false;                    // 64 x 64 planes (sc_kernels_plane64.h); shares pl_tab128 (here: w64^m) and pl_cs_*
  cf32* pl_tab128 = nullptr;
  float* pl_cs_fwd[2] = {nullptr, nullptr};
  float* pl_cs_inv[2] = {nullptr, nullptr};
  // weight sub-block index tables (device), keyed by (w_extent, w_start)
  std::mutex idx_mu;
  std::map<IdxKey, int32_t*> idx_cache;
  std::vector<void*> owned;
};

static const double kTwoPi = 6.283185307179586476925286766559;

static cf32 twiddle(int64_t f, int64_t n, int64_t N, double sign, double scale) {
  int64_t prod = ((f % N) * (n % N)) % N;
  if (prod < 0) prod += N;
  const double th = kTwoPi * (double)prod / (double)N;
  return cf_make((float)(scale * std::cos(th)), (float)(sign * scale * std::sin(th)));
}

static int upload_table(sc_plan* p, const std::vector<cf32>& host, int rows, int cols_pad,
                        DeviceTable* out) {
  void* dev = nullptr;
  SC_CHECK_HIP(hipMalloc(&dev, host.size() * sizeof(cf32)));
  p->owned.push_back(dev);
  SC_CHECK_HIP(hipMemcpy(dev, host.data(), host.size() * sizeof(cf32), hipMemcpyHostToDevice));
  out->ptr = (cf32*)dev;
  out->rows = rows;
  out->cols_pad = cols_pad;
  return 0;
}

static int pick_tile(int64_t J, const int* cands, int ncand, int mult) {
  int best = cands[0];
  int64_t best_pad = -1;
  for (int c = 0; c < ncand; ++c) {
    const int64_t w = (int64_t)cands[c] * mult;
    const int64_t pad = (J + w - 1) / w * w;
    if (best_pad < 0 || pad < best_pad || (pad == best_pad && cands[c] > best)) {
      best_pad = pad;
      best = cands[c];
    }
  }
  return best;
}

static int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

static double col_weight(int64_t j, int64_t N) {
  if (j == 0) return 1.0;
  if (N % 2 == 0 && j == N / 2) return 1.0;
  return 2.0;
}

// table entry of the last (real-data) axis: kept column j at sample n.  `weighted` = the two modes that are
// (the adjoint of) the zero-padded inverse: C2R column weight, inverse scale, and real_col honoured.
static cf32 last_tw(const sc_plan* p, int64_t j, int64_t n, double sign, bool weighted) {
  const int L = p->nd - 1;
  const int64_t N = p->n[L], f = p->fmap[L][(size_t)j];
  if (f == SC_FREQ_DROPPED) return cf_make(0.f, 0.f);
  cf32 tw = twiddle(f, n, N, sign, weighted ? p->si * col_weight(f, N) : p->sf);
  if (weighted && p->d.real_col > 0 && j == p->d.real_col) tw.y = 0.f;
  return tw;
}
// table entry of a complex-to-complex axis: kept row r of dim d at sample n
static cf32 axis_tw(const sc_plan* p, int d, int64_t r, int64_t n, double sign, double scale = 1.0) {
  const int64_t f = p->fmap[d][(size_t)r];
  if (f == SC_FREQ_DROPPED) return cf_make(0.f, 0.f);
  return twiddle(f, n, p->n[d], sign, scale);
}

static int upload_floats(sc_plan* p, const std::vector<float>& host, float** out) {
  void* dev = nullptr;
  SC_CHECK_HIP(hipMalloc(&dev, host.size() * sizeof(float)));
  p->owned.push_back(dev);
  SC_CHECK_HIP(hipMemcpy(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = (float*)dev;
  return 0;
}

// A-B switches of the size-agnostic passes: COMPILE-TIME only (measurement builds, scripts/build_variants.py);
// the product library is built without any of them.
//   -DSC_MDFT_TILE4    4 accumulator tiles per wave instead of 8 (first-generation passes)
//   -DSC_MDFT_NOLDS    first-generation last-axis passes (operands straight from global memory)
//   -DSC_MDFT_NOTAIL   the 2^k + 1-th kept column as an MFMA tile instead of a VALU dot product
//   -DSC_MDFT_NOPLANE  last two axes as separate passes
//   -DSC_MDFT_NOSTAGE  widths outside k_mdft_r2c_lds's scope straight from global memory (k_mdft_r2c) instead of the
//                      LDS-staged k_mdft_r2c_stage
struct MdftSwitches {
#ifdef SC_MDFT_TILE4
  static constexpr bool tile4 = true;
#else
  static constexpr bool tile4 = false;
#endif
#ifdef SC_MDFT_NOLDS
  static constexpr bool nolds = true;
#else
  static constexpr bool nolds = false;
#endif
#ifdef SC_MDFT_NOTAIL
  static constexpr bool notail = true;
#else
  static constexpr bool notail = false;
#endif
#ifdef SC_MDFT_NOPLANE
  static constexpr bool noplane = true;
#else
  static constexpr bool noplane = false;
#endif
#ifdef SC_MDFT_NOSTAGE
  static constexpr bool nostage = true;
#else
  static constexpr bool nostage = false;
#endif
};
static constexpr MdftSwitches mdft_switches() { return MdftSwitches(); }

// tables of the matrix-core passes, in MFMA lane order (layouts: sc_kernels_mdft.h)
static int build_mdft_tables(sc_plan* p) {
  const int L = p->nd - 1;
  const int64_t N = p->n[L], J = p->k[L];
  int rc = 0;
  if (!p->cplx) {                       // any N: the last group of 8 points is zero-padded (k_mdft_r2c<.., RAGGED>)
    const int64_t NG = (N + 7) / 8, CT = (2 * J + 31) / 32;
    for (int v = 0; v < 2 && !rc; ++v) {
      std::vector<float> h((size_t)(CT * NG * 4 * 64), 0.f);
      for (int64_t ct = 0; ct < CT; ++ct)
        for (int64_t t = 0; t < NG; ++t)
          for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane) {
              const int64_t f = 32 * ct + (lane & 31), j = f >> 1, n = 8 * t + 4 * (lane >> 5) + q;
              if (j >= J || n >= N) continue;
              const cf32 tw = last_tw(p, j, n, -1.0, v != SC_FWD_SCALED);
              h[(size_t)(((ct * NG + t) * 4 + q) * 64 + lane)] = (f & 1) ? tw.y : tw.x;
            }
      rc = upload_floats(p, h, &p->m_r2c[v]);
      if (!rc && J > 1 && (2 * J) % 32 == 2) {
        std::vector<float> ht((size_t)(NG * 4 * 2 * 2), 0.f);
        for (int64_t t = 0; t < NG; ++t)
          for (int q = 0; q < 4; ++q)
            for (int hh = 0; hh < 2; ++hh) {
              const int64_t n = 8 * t + 4 * hh + q, j = J - 1;
              if (n >= N) continue;
              const cf32 tw = last_tw(p, j, n, -1.0, v != SC_FWD_SCALED);
              ht[(size_t)(((t * 4 + q) * 2 + hh) * 2 + 0)] = tw.x;
              ht[(size_t)(((t * 4 + q) * 2 + hh) * 2 + 1)] = tw.y;
            }
        float* dev = nullptr;
        rc = upload_floats(p, ht, &dev);
        p->m_r2c_tail[v] = (cf32*)dev;
      }
    }
  }
  if (!p->cplx) {
    const int64_t JS = (J + 1) / 2, NT = (N + 31) / 32;
    for (int v = 0; v < 2 && !rc; ++v) {
      std::vector<float> h((size_t)(NT * JS * 2 * 64), 0.f);
      for (int64_t nt = 0; nt < NT; ++nt)
        for (int64_t t = 0; t < JS; ++t)
          for (int comp = 0; comp < 2; ++comp)
            for (int lane = 0; lane < 64; ++lane) {
              const int64_t n = 32 * nt + (lane & 31), j = 2 * t + (lane >> 5);
              if (j >= J || n >= N) continue;
              const cf32 tw = last_tw(p, j, n, +1.0, v == SC_INV_PADDED);
              h[(size_t)(((nt * JS + t) * 2 + comp) * 64 + lane)] = comp ? -tw.y : tw.x;
            }
      rc = upload_floats(p, h, &p->m_c2r[v]);
    }
  }
  // LDS-staged r2c: whole table resident in LDS (<= 32 KB), at most 2 column tiles
  if (!rc && !p->cplx && N % 32 == 0 && N <= 256) {
    const bool tail = J > 1 && (2 * J) % 32 == 2 && 2 * J > 32;
    const int64_t NG = N / 8, CT = tail ? (2 * J - 2) / 32 : (2 * J + 31) / 32;
    if (CT <= 2 && NG * CT * 256 <= 8192) {
      p->l_r2c_ct = (int)CT;
      for (int v = 0; v < 2 && !rc; ++v) {
        std::vector<float> h((size_t)(CT * NG * 256), 0.f);
        for (int64_t ct = 0; ct < CT; ++ct)
          for (int64_t t = 0; t < NG; ++t)
            for (int lane = 0; lane < 64; ++lane)
              for (int q = 0; q < 4; ++q) {
                const int64_t f = 32 * ct + (lane & 31), j = f >> 1, n = 8 * t + 4 * (lane >> 5) + q;
                if (j >= J) continue;
                const cf32 tw = last_tw(p, j, n, -1.0, v != SC_FWD_SCALED);
                h[(size_t)(((ct * NG + t) * 64 + lane) * 4 + q)] = (f & 1) ? tw.y : tw.x;
              }
        rc = upload_floats(p, h, &p->l_r2c[v]);
        if (!rc && tail) {
          std::vector<float> ht((size_t)(2 * N), 0.f);
          for (int64_t n = 0; n < N; ++n) {
            const cf32 tw = last_tw(p, J - 1, n, -1.0, v != SC_FWD_SCALED);
            ht[(size_t)(2 * n)] = tw.x;
            ht[(size_t)(2 * n + 1)] = tw.y;
          }
          float* dev = nullptr;
          rc = upload_floats(p, ht, &dev);
          p->l_r2c_tail[v] = (cf32*)dev;
        }
      }
    }
  }
  // LDS-staged r2c of the remaining widths (any N): table in global memory, lane-major float4, NG = 4 ceil(N / 32)
  if (!rc && !p->cplx && !p->l_r2c[0] && !mdft_switches().nostage) {
    const bool tail = J > 1 && (2 * J) % 32 == 2 && 2 * J > 32 && !mdft_switches().notail;
    const int64_t NC = (N + 31) / 32, NG = 4 * NC, CT = tail ? (2 * J - 2) / 32 : (2 * J + 31) / 32;
    if (CT <= 2 && (!tail || NC * 32 <= 1024)) {
      p->s_r2c_ct = (int)CT;
      for (int v = 0; v < 2 && !rc; ++v) {
        std::vector<float> h((size_t)(CT * NG * 256), 0.f);
        for (int64_t ct = 0; ct < CT; ++ct)
          for (int64_t t = 0; t < NG; ++t)
            for (int lane = 0; lane < 64; ++lane)
              for (int q = 0; q < 4; ++q) {
                const int64_t f = 32 * ct + (lane & 31), j = f >> 1, n = 8 * t + 4 * (lane >> 5) + q;
                if (j >= J || n >= N) continue;
                const cf32 tw = last_tw(p, j, n, -1.0, v != SC_FWD_SCALED);
                h[(size_t)(((ct * NG + t) * 64 + lane) * 4 + q)] = (f & 1) ? tw.y : tw.x;
              }
        rc = upload_floats(p, h, &p->s_r2c[v]);
        if (!rc && tail) {
          std::vector<float> ht((size_t)(2 * 32 * NC), 0.f);
          for (int64_t n = 0; n < N; ++n) {
            const cf32 tw = last_tw(p, J - 1, n, -1.0, v != SC_FWD_SCALED);
            ht[(size_t)(2 * n)] = tw.x;
            ht[(size_t)(2 * n + 1)] = tw.y;
          }
          float* dev = nullptr;
          rc = upload_floats(p, ht, &dev);
          p->s_r2c_tail[v] = (cf32*)dev;
        }
      }
    }
  }
  // LDS-staged c2r: table + one 128-line tile of the spectrum within 48 KB (3+ blocks per CU)
  if (!rc && !p->cplx) {
    const int64_t JS = (J + 1) / 2, NT = (N + 31) / 32;
    const int64_t S = (J % 2) ? 2 * J : 2 * J + 2;
    if ((NT * JS * 128 + SC_MDFT_LB * S) * 4 <= 42 * 1024 && N % 4 == 0) {
      p->l_c2r_s = (int)S;
      for (int v = 0; v < 2 && !rc; ++v) {
        std::vector<float> h((size_t)(NT * JS * 128), 0.f);
        for (int64_t nt = 0; nt < NT; ++nt)
          for (int64_t t = 0; t < JS; ++t)
            for (int lane = 0; lane < 64; ++lane) {
              const int64_t n = 32 * nt + (lane & 31), j = 2 * t + (lane >> 5);
              if (j >= J || n >= N) continue;
              const cf32 tw = last_tw(p, j, n, +1.0, v == SC_INV_PADDED);
              h[(size_t)(((nt * JS + t) * 64 + lane) * 2 + 0)] = tw.x;
              h[(size_t)(((nt * JS + t) * 64 + lane) * 2 + 1)] = -tw.y;
            }
        rc = upload_floats(p, h, &p->l_c2r[v]);
      }
    }
  }
  // LDS-staged c2r of the remaining widths: table in global memory, [((nt * JS2 + p) * 64 + lane) * 4 + e]
  if (!rc && !p->cplx && !p->l_c2r[0] && !mdft_switches().nostage) {
    const int64_t JS = (J + 1) / 2, NT = (N + 31) / 32;
    const int64_t need = (JS + 1) / 2, JS2 = need <= 3 ? 3 : (need <= 5 ? 5 : (need <= 9 ? 9 : 0));
    if (JS2) {
      p->s_c2r_js2 = (int)JS2;
      p->s_c2r_s = (int)((J % 2) ? 2 * J : 2 * J + 2);
      for (int v = 0; v < 2 && !rc; ++v) {
        std::vector<float> h((size_t)(NT * JS2 * 256), 0.f);
        for (int64_t nt = 0; nt < NT; ++nt)
          for (int64_t pp = 0; pp < JS2; ++pp)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int64_t t = 2 * pp + (e >> 1), n = 32 * nt + (lane & 31), j = 2 * t + (lane >> 5);
                if (j >= J || n >= N) continue;
                const cf32 tw = last_tw(p, j, n, +1.0, v == SC_INV_PADDED);
                h[(size_t)(((nt * JS2 + pp) * 64 + lane) * 4 + e)] = (e & 1) ? -tw.y : tw.x;
              }
        rc = upload_floats(p, h, &p->s_c2r[v]);
      }
    }
  }
  for (int d = 0; d < L && !rc; ++d) {
    const int64_t Nd = p->n[d], Kd = p->k[d];
    for (int dir = 0; dir < 2 && !rc; ++dir) {
      // dir 0: forward (N = n_d inputs -> J = k_d kept rows), dir 1: inverse (N = k_d -> J = n_d)
      const int64_t Nin = dir ? Kd : Nd, Jout = dir ? Nd : Kd;
      const int64_t NS = (Nin + 1) / 2, JT = (Jout + 15) / 16;
      std::vector<float> h((size_t)(JT * NS * 2 * 64), 0.f);
      for (int64_t jt = 0; jt < JT; ++jt)
        for (int64_t sidx = 0; sidx < NS; ++sidx)
          for (int comp = 0; comp < 2; ++comp)
            for (int lane = 0; lane < 64; ++lane) {
              const int row = lane & 31;
              const int64_t j = 16 * jt + (row >> 1), n = 2 * sidx + (lane >> 5);
              if (j >= Jout || n >= Nin) continue;
              const cf32 tw = dir ? axis_tw(p, d, n, j, +1.0) : axis_tw(p, d, j, n, -1.0);
              float val;
              if (comp == 0) val = (row & 1) ? tw.y : tw.x;        // times Re(in): (Re out, Im out)
              else val = (row & 1) ? tw.x : -tw.y;                 // times Im(in)
              h[(size_t)(((jt * NS + sidx) * 2 + comp) * 64 + lane)] = val;
            }
      rc = upload_floats(p, h, dir ? &p->m_ax_inv[d] : &p->m_ax_fwd[d]);
    }
  }
  // plane form (sc_kernels_mdft.h): forward row pass of the second-to-last dim, rows j1 x (Re T | Im T) lanes
  if (!rc && L >= 1 && (p->n[L - 1] == 128 || p->n[L - 1] == 64 || p->n[L - 1] == 32) && 2 * p->k[L - 1] <= p->n[L - 1]) {
    const int d = L - 1;
    const int64_t NR = p->n[d], K1 = p->k[d], JP = (K1 + 31) / 32;
    std::vector<float> h((size_t)(JP * NR * 64), 0.f);
    for (int64_t jt = 0; jt < JP; ++jt)
      for (int64_t n1 = 0; n1 < NR; ++n1)
        for (int lane = 0; lane < 64; ++lane) {
          const int64_t j1 = 32 * jt + (lane & 31);
          if (j1 >= K1) continue;
          const cf32 tw = axis_tw(p, d, j1, n1, -1.0);
          h[(size_t)((jt * NR + n1) * 64 + lane)] = (lane >> 5) ? tw.y : tw.x;
        }
    rc = upload_floats(p, h, &p->m_pl_fwd);
    if (!rc) {
      std::vector<float> hi((size_t)((NR / 32) * K1 * 64), 0.f);
      for (int64_t rt = 0; rt < NR / 32; ++rt)
        for (int64_t j1 = 0; j1 < K1; ++j1)
          for (int lane = 0; lane < 64; ++lane) {
            const cf32 tw = axis_tw(p, d, j1, 32 * rt + (lane & 31), +1.0);
            hi[(size_t)((rt * K1 + j1) * 64 + lane)] = (lane >> 5) ? tw.y : tw.x;
          }
      rc = upload_floats(p, hi, &p->m_pl_inv);
    }
  }
  if (!rc) p->mdft = true;
  return rc;
}


// ------------------------------------------------------------------------------------------
// two-pass factorised route (sc_kernels_fft2p.h): eligibility, tables, launches
// ------------------------------------------------------------------------------------------
#ifndef SC_F2P_C2R_WGS
#define SC_F2P_C2R_WGS 2        // persistent workgroups of k_f2p_c2r per compute unit (76 KB of LDS each)
#endif
#ifndef SC_F2P_CHUNK_MB
#define SC_F2P_CHUNK_MB 192     // panel bytes in flight between the two passes (Infinity Cache: 256 MB); measured
                                // 32 / 96 / 192 / unchunked: 0.84 / 0.74 / 0.70 / 0.73 ms forward, 1.45 / 1.02 / 0.94 / 1.07 ms inverse
                                // (profiles/r02_f2p_1024_time.txt)
#endif

static int f2p_pow2_at_least(int64_t v) {
  int r = 1;
  while (r < v) r *= 2;
  return r;
}

// per-column factors of the packed-row FFT kernels (sc_kernels_fft2p.h, sc_kernels_plane.h): norm x C2R column
// weight, x 1/2 for the split of a packed pair (forward: every column; inverse: every column but k = 0)
static int fft_col_scales(sc_plan* p, float** fwd, float** inv) {
  const int L = p->nd - 1;
  const int64_t J = p->k[L];
  for (int v = 0; v < 2; ++v) {
    std::vector<float> f((size_t)J), g((size_t)J);
    for (int64_t k = 0; k < J; ++k) {
      const double wf = (v == SC_FWD_SCALED) ? p->sf : p->si * col_weight(k, p->n[L]);
      const double wi = (v == SC_INV_PADDED) ? p->si * col_weight(k, p->n[L]) : p->sf;
      f[(size_t)k] = (float)(0.5 * wf);
      g[(size_t)k] = (float)(k == 0 ? wi : 0.5 * wi);
    }
    int rc = upload_floats(p, f, &fwd[v]);
    if (!rc) rc = upload_floats(p, g, &inv[v]);
    if (rc) return rc;
  }
  return 0;
}

// points per lane of the lines the two-pass kernels are instantiated for (N = 32 P)
static bool f2p_line_ok(int64_t n) {
  if (n % 32) return false;
  switch (n / 32) {
    case 2: case 3: case 4: case 5: case 6: case 8: case 10: case 12: case 16: case 20: case 32: return true;
    default: return false;
  }
}

// large_only: the round-2 envelope (both sizes 512 / 1024), tried BEFORE the 128 x 128 plane kernels; the general
// call comes after them, so that grids with a fused one-launch route keep it
static int f2p_plan_init(sc_plan* p, bool large_only) {
  if (p->nd != 2 || p->cplx || p->custom_map || p->d.real_col) return 0;
  for (int d = 0; d < 2; ++d) {
    if (large_only ? (p->n[d] != 512 && p->n[d] != 1024) : !f2p_line_ok(p->n[d])) return 0;
  }
  // small planes: two launches + a panel per transform lose against the one-launch direct-DFT plane form of the
  // matrix-core passes (64^2, modes 32: 0.39 vs 0.23 ms/step; 192^2, modes 64: 0.70 vs 1.31 ms/step --
  // profiles/r03_f2p_widths_ab.txt).  Taken from 128 x 128 points per plane; SC_PLAN_F2P_SMALL_ALWAYS overrides (A-B, tests)
  if (!large_only && p->n[0] * p->n[1] < 128 * 128 && !(p->d.flags & SC_PLAN_F2P_SMALL_ALWAYS)) return 0;
  const int P0 = (int)(p->n[0] / 32), P1 = (int)(p->n[1] / 32);
  const int64_t K0 = p->k[0], J = p->k[1];
  if (J > p->n[1] / 2) return 0;                          // kept columns stay below the Nyquist column
  int k2r = f2p_pow2_at_least((((K0 + 1) / 2) + P0 - 1) / P0);
  int k2c = f2p_pow2_at_least(J > 1 ? (J - 1 + P1 - 1) / P1 : 1);
  if (k2r > 8 || k2c > 8) return 0;
  // lines of 512 / 1024 points are instantiated for every pruning range, the other lengths for 4 and 8 only
  if (P0 != 16 && P0 != 32 && k2r < 4) k2r = 4;
  if (P1 != 16 && P1 != 32 && k2c < 4) k2c = 4;
  p->f2p_p[0] = P0;
  p->f2p_p[1] = P1;
  p->f2p_k2[0] = k2r;
  p->f2p_k2[1] = k2c;
  p->f2p_ncb = (int)((J + SC_F2P_CB - 1) / SC_F2P_CB);
  for (int d = 0; d < 2; ++d) {
    const int P = p->f2p_p[d];
    const int64_t N = p->n[d];
    std::vector<cf32> h((size_t)P * 32);
    for (int k1 = 0; k1 < P; ++k1)
      for (int t = 0; t < 32; ++t) h[(size_t)k1 * 32 + t] = twiddle(k1, t, N, -1.0, 1.0);
    DeviceTable dt;
    int rc = upload_table(p, h, P, 32, &dt);
    if (rc) return rc;
    p->f2p_tw[d] = dt.ptr;
  }
  int rc = fft_col_scales(p, p->f2p_cs_fwd, p->f2p_cs_inv);
  if (rc) return rc;
  p->f2p_roww = p->n[1] == 1024 && J == 129;             // one wave per row pair (k_f2p_c2r_w1024): kept columns k = 0..128
  p->f2p_colw = p->n[0] == 1024 && K0 <= 256;            // 64 lanes per column line (k_f2p_col_inv_w1024)
  if (p->f2p_roww || p->f2p_colw) {
    std::vector<cf32> h(1024);
    for (int m = 0; m < 1024; ++m) h[(size_t)m] = twiddle(m, 1, 1024, -1.0, 1.0);
    DeviceTable dt;
    rc = upload_table(p, h, 1, 1024, &dt);
    if (rc) return rc;
    p->f2p_w1024 = dt.ptr;
  }
  p->f2p = true;
  return 0;
}

static int pl128_plan_init(sc_plan* p) {
  const int L = p->nd - 1;
  if (p->nd < 2 || p->cplx || p->custom_map || p->d.real_col) return 0;
  if (p->n[L] != SC_PL_N || p->n[L - 1] != SC_PL_N || p->k[L] > SC_PL_JMAX || p->k[L - 1] > SC_PL_KMAX) return 0;
  std::vector<cf32> h(128);
  for (int m = 0; m < 128; ++m) h[(size_t)m] = twiddle(m, 1, 128, -1.0, 1.0);
  DeviceTable dt;
  int rc = upload_table(p, h, 1, 128, &dt);
  if (!rc) rc = fft_col_scales(p, p->pl_cs_fwd, p->pl_cs_inv);
  if (rc) return rc;
  p->pl_tab128 = dt.ptr;
  p->pl128 = true;
  return 0;
}

static int pl64_plan_init(sc_plan* p) {
  const int L = p->nd - 1;
  if (p->nd < 2 || p->cplx || p->custom_map || p->d.real_col) return 0;
  if (p->n[L] != SC_P64_N || p->n[L - 1] != SC_P64_N || p->k[L] > SC_P64_JMAX || p->k[L - 1] > SC_P64_KMAX) return 0;
  std::vector<cf32> h(64);
  for (int m = 0; m < 64; ++m) h[(size_t)m] = twiddle(m, 1, 64, -1.0, 1.0);
  DeviceTable dt;
  int rc = upload_table(p, h, 1, 64, &dt);
  if (!rc) rc = fft_col_scales(p, p->pl_cs_fwd, p->pl_cs_inv);
  if (rc) return rc;
  p->pl_tab128 = dt.ptr;
  p->pl64 = true;
  return 0;
}

static int64_t f2p_panel_elems_per_image(const sc_plan* p) { return (int64_t)p->f2p_ncb * p->n[0] * SC_F2P_CB; }
static int64_t f2p_chunk_images(const sc_plan* p, int64_t n_images) {
  static const int mb = [] { const char* e = SC_DIAG_ENV("SC_F2P_CHUNK_MB"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : SC_F2P_CHUNK_MB; }();   // A-B
  int64_t c = ((int64_t)mb << 20) / (f2p_panel_elems_per_image(p) * (int64_t)sizeof(cf32));
  if (c < 1) c = 1;
  return c < n_images ? c : n_images;
}

template <typename F>
static bool f2p_dispatch(int P, int K2, F&& f) {
  switch (P * 100 + K2) {
    case 3201: f(sc_int<32>(), sc_int<1>()); return true;
    case 3202: f(sc_int<32>(), sc_int<2>()); return true;
    case 3204: f(sc_int<32>(), sc_int<4>()); return true;
    case 3208: f(sc_int<32>(), sc_int<8>()); return true;
    case 1601: f(sc_int<16>(), sc_int<1>()); return true;
    case 1602: f(sc_int<16>(), sc_int<2>()); return true;
    case 1604: f(sc_int<16>(), sc_int<4>()); return true;
    case 1608: f(sc_int<16>(), sc_int<8>()); return true;
#define SC_F2P_CASES(P) case P * 100 + 4: f(sc_int<P>(), sc_int<4>()); return true; \
                        case P * 100 + 8: f(sc_int<P>(), sc_int<8>()); return true;
    SC_F2P_CASES(2) SC_F2P_CASES(3) SC_F2P_CASES(4) SC_F2P_CASES(5) SC_F2P_CASES(6) SC_F2P_CASES(8)
    SC_F2P_CASES(10) SC_F2P_CASES(12) SC_F2P_CASES(20)
#undef SC_F2P_CASES
    default: return false;
  }
}

// Round 4, measured and NOT taken as the default: the two passes of a transform as a TWO-STAGE PIPELINE over half-sized
// chunks on two streams (the caller's and the engine's side stream, sc_device.h) -- the small pass of chunk c
// (k_f2p_col_*) beside the large pass of its neighbour chunk (k_f2p_r2c / k_f2p_c2r), the panel double-buffered inside
// the same workspace, dependencies by events only (fork / join).  1024^2, 512 images, same box, interleaved
// (profiles/r04_f2p_pipeline_ab.txt): forward 0.65 -> 0.71 ms, inverse 0.87 -> 0.97 ms, the adjoints 0.60 -> 0.68 and
// 0.85 -> 0.95 ms: the column pass beside the row pass takes more from it (L2 / fabric contention, half-sized launches)
// than its own time.  SC_F2P_PIPE=1 (environment) switches it on for A-B runs; the default alternates the passes on the
// caller's stream over full-sized chunks.
static bool f2p_pipe_enabled() {
  static const bool on = [] { const char* e = SC_DIAG_ENV("SC_F2P_PIPE"); return e && e[0] == '1'; }();
  return on;
}
struct F2pChunks {
  int64_t chunk;      // images per chunk
  int64_t n_chunks;
  bool pipe;          // two half-workspace panel buffers, two streams
};
static F2pChunks f2p_chunks(const sc_plan* p, int64_t n_images, const ScSide* side) {
  F2pChunks c;
  const int64_t full = f2p_chunk_images(p, n_images);        // what the workspace was sized for
  c.pipe = side != nullptr && f2p_pipe_enabled() && full >= 2 && n_images >= 16;
  c.chunk = c.pipe ? full / 2 : full;
  // equal chunks (no short tail chunk at the end of the pipeline)
  c.n_chunks = (n_images + c.chunk - 1) / c.chunk;
  if (c.pipe && c.n_chunks < 2) c.n_chunks = 2;
  c.chunk = (n_images + c.n_chunks - 1) / c.n_chunks;
  return c;
}

static bool f2p_launch_r2c(const sc_plan* p, int mode, const float* xs, cf32* panel, int64_t ni, sc_stream_t st) {
  const int N0 = (int)p->n[0], J = (int)p->k[1], NCB = p->f2p_ncb;
  return f2p_dispatch(p->f2p_p[1], p->f2p_k2[1], [&](auto P, auto K2) {
    constexpr int G = 32 / decltype(P)::value;           // row pairs per half-wave
    SC_LAUNCH((k_f2p_r2c<decltype(P)::value, decltype(K2)::value>), dim3((unsigned)((ni * N0 / 2 + 8 * G - 1) / (8 * G))),
              dim3(256), 0, st,
              xs, panel, (const cf32*)p->f2p_tw[1], (const float*)p->f2p_cs_fwd[mode], N0, J, NCB, ni * N0 / 2);
  });
}
static bool f2p_launch_col_fwd(const sc_plan* p, const cf32* panel, cf32* dst, int64_t ni, sc_stream_t st) {
  const int J = (int)p->k[1], K0 = (int)p->k[0], NCB = p->f2p_ncb;
  const int64_t n_blk = ni * NCB;
  return f2p_dispatch(p->f2p_p[0], p->f2p_k2[0], [&](auto P, auto K2) {
    constexpr int G = 32 / decltype(P)::value;           // panel blocks per workgroup
    const int pxc = (int)(((n_blk + G - 1) / G + 7) / 8);
    SC_LAUNCH((k_f2p_col_fwd<decltype(P)::value, decltype(K2)::value>), dim3((unsigned)(8 * pxc)), dim3(256), 0, st,
              panel, dst, (const cf32*)p->f2p_tw[0], NCB, J, K0, n_blk, pxc);
  });
}
static bool f2p_launch_col_inv(const sc_plan* p, const cf32* src, cf32* panel, int64_t ni, sc_stream_t st) {
  const int J = (int)p->k[1], K0 = (int)p->k[0], NCB = p->f2p_ncb;
  const int64_t n_blk = ni * NCB;
  static const bool no_colw = SC_DIAG_ENV("SC_F2P_NO_COLW") != nullptr;              // A-B: the 32-lane kernel
  if (p->f2p_colw && !no_colw && n_blk < ((int64_t)1 << 30)) {
    // columns of 1024 points: 64 lanes per line, two 512-thread workgroups per compute unit, persistent
    static const int wgs = [] { const char* e = SC_DIAG_ENV("SC_F2P_COLW_WGS"); return e ? std::atoi(e) : 2; }();
    int64_t grid = (int64_t)(wgs > 0 ? wgs : 2) * sc_cu_count();
    if (grid > n_blk) grid = n_blk;
    if (grid >= 8) grid &= ~(int64_t)7;                                              // whole XCD rounds (the kernel's block map)
    if (K0 == 256)
      SC_LAUNCH((k_f2p_col_inv_w1024<true>), dim3((unsigned)grid), dim3(512), 0, st, src, panel, (const cf32*)p->f2p_w1024,
                NCB, J, K0, (int)n_blk, (int)grid);
    else
      SC_LAUNCH((k_f2p_col_inv_w1024<false>), dim3((unsigned)grid), dim3(512), 0, st, src, panel, (const cf32*)p->f2p_w1024,
                NCB, J, K0, (int)n_blk, (int)grid);
    return true;
  }
  return f2p_dispatch(p->f2p_p[0], p->f2p_k2[0], [&](auto P, auto K2) {
    constexpr int G = 32 / decltype(P)::value;
    const int pxc = (int)(((n_blk + G - 1) / G + 7) / 8);
    SC_LAUNCH((k_f2p_col_inv<decltype(P)::value, decltype(K2)::value>), dim3((unsigned)(8 * pxc)), dim3(256), 0, st,
              src, panel, (const cf32*)p->f2p_tw[0], NCB, J, K0, n_blk, pxc);
  });
}
// skip != NULL: ys = (transform + bias) + skip (the 1024-point row kernel only: f2p_fused_add)
static bool f2p_launch_c2r(const sc_plan* p, int mode, const cf32* panel, float* ys, const float* bias, int64_t channels,
                           int64_t i0, int64_t ni, sc_stream_t st, const float* skip = nullptr) {
  const int N0 = (int)p->n[0], J = (int)p->k[1], NCB = p->f2p_ncb;
  static const bool no_w1024 = SC_DIAG_ENV("SC_F2P_NO_W1024") != nullptr;          // A-B: the half-wave kernel
  if (p->f2p_roww && !no_w1024) {
    // rows of 1024 points: one wave per packed row pair, four 4-wave workgroups per compute unit (sc_kernels_fft2p.h)
    static const int wgs = [] { const char* e = SC_DIAG_ENV("SC_F2P_W1024_WGS"); return e ? std::atoi(e) : 4; }();
    const int64_t n_pairs = ni * N0 / 2, n_items = (n_pairs + 3) / 4;
    int64_t grid = (int64_t)(wgs > 0 ? wgs : 4) * sc_cu_count();
    if (grid > n_items) grid = n_items;
    if (n_pairs < ((int64_t)1 << 30)) {
      if (skip)
        SC_LAUNCH(k_f2p_c2r_w1024<true>, dim3((unsigned)grid), dim3(256), 0, st, panel, ys, (const cf32*)p->f2p_w1024,
                  (const float*)p->f2p_cs_inv[mode], bias, (int)channels, (int)(i0 % channels), N0, NCB, (int)n_pairs,
                  (int)n_items, (int)grid, skip);
      else
        SC_LAUNCH(k_f2p_c2r_w1024<false>, dim3((unsigned)grid), dim3(256), 0, st, panel, ys, (const cf32*)p->f2p_w1024,
                  (const float*)p->f2p_cs_inv[mode], bias, (int)channels, (int)(i0 % channels), N0, NCB, (int)n_pairs,
                  (int)n_items, (int)grid, (const float*)nullptr);
      return true;
    }
  }
  if (skip) return false;                                  // (callers ask f2p_fused_add first)
  return f2p_dispatch(p->f2p_p[1], p->f2p_k2[1], [&](auto P, auto K2) {
    constexpr int G = 32 / decltype(P)::value;
    const int64_t n_items = (ni * N0 / 2 + 8 * G - 1) / (8 * G);
    int64_t grid = (int64_t)SC_F2P_C2R_WGS * sc_cu_count();   // persistent workgroups (sc_kernels_fft2p.h)
    if (grid > n_items) grid = n_items;
    SC_LAUNCH((k_f2p_c2r<decltype(P)::value, decltype(K2)::value>), dim3((unsigned)grid), dim3(256), 0, st,
              panel, ys, (const cf32*)p->f2p_tw[1], (const float*)p->f2p_cs_inv[mode], bias,
              (int)channels, (int)(i0 % channels), N0, J, NCB, ni * N0 / 2, n_items, (int)grid);
  });
}

static int f2p_forward(const sc_plan* p, int mode, const float* x, cf32* xhat, int64_t n_images, void* workspace,
                       sc_stream_t st) {
  SC_CHECK_ARG(workspace, "workspace required");
  ScSide* side = sc_side_get();
  const F2pChunks ck = f2p_chunks(p, n_images, side);
  cf32* buf[2] = {(cf32*)workspace, (cf32*)workspace + (ck.pipe ? ck.chunk * f2p_panel_elems_per_image(p) : 0)};
  const char* nok = "sc_engine: two-pass route: no kernel for this line length / kept range";
  const char* sync_fail = "sc_engine: two-pass route: event record / wait failed";
  auto ni_of = [&](int64_t c) { const int64_t i0 = c * ck.chunk; return n_images - i0 < ck.chunk ? n_images - i0 : ck.chunk; };
  if (!ck.pipe) {
    for (int64_t c = 0; c < ck.n_chunks; ++c) {
      const int64_t i0 = c * ck.chunk, ni = ni_of(c);
      if (!f2p_launch_r2c(p, mode, x + i0 * p->ntot, buf[0], ni, st) ||
          !f2p_launch_col_fwd(p, buf[0], xhat + i0 * p->modes, ni, st))
        return sc_fail(nok);
    }
    return sc_check_launch("k_f2p_r2c / k_f2p_col_fwd");
  }
  // main: r2c(0) r2c(1) r2c(2) ...        side: col_fwd(0) col_fwd(1) ...   (col_fwd(c) after r2c(c); r2c(c + 2) after
  // col_fwd(c): same panel buffer)
  ScSideJoinGuard guard(side, st);
  sc_stream_t ss = (sc_stream_t)side->stream;
  if (!f2p_launch_r2c(p, mode, x, buf[0], ni_of(0), st)) return sc_fail(nok);
  for (int64_t c = 0; c < ck.n_chunks; ++c) {
    const int64_t i0 = c * ck.chunk, ni = ni_of(c);
    if (guard.armed && !sc_side_join(side, st)) return sc_fail(sync_fail);          // main: col_fwd(<= c - 1) are done
    if (!sc_side_fork(side, st)) return sc_fail(sync_fail);                          // side: r2c(c) is done
    guard.armed = true;
    if (!f2p_launch_col_fwd(p, buf[c & 1], xhat + i0 * p->modes, ni, ss)) return sc_fail(nok);
    if (c + 1 < ck.n_chunks &&
        !f2p_launch_r2c(p, mode, x + (i0 + ck.chunk) * p->ntot, buf[(c + 1) & 1], ni_of(c + 1), st))
      return sc_fail(nok);
  }
  guard.armed = false;
  if (!sc_side_join(side, st)) return sc_fail(sync_fail);
  return sc_check_launch("k_f2p_r2c / k_f2p_col_fwd");
}

// the shapes whose row pass adds the epilogue's skip in its store path (f2p_launch_c2r: the one-wave-per-row-pair kernel)
static bool f2p_fused_add(const sc_plan* p, int64_t n_images) {
  static const bool no_w1024 = SC_DIAG_ENV("SC_F2P_NO_W1024") != nullptr;
  return p->f2p && p->f2p_roww && !no_w1024 && n_images * p->n[0] / 2 < ((int64_t)1 << 30);
}
static int f2p_inverse(const sc_plan* p, int mode, const cf32* yhat, const float* bias, int64_t channels, float* y,
                       int64_t n_images, void* workspace, sc_stream_t st, const float* skip = nullptr) {
  SC_CHECK_ARG(workspace, "workspace required");
  ScSide* side = sc_side_get();
  const F2pChunks ck = f2p_chunks(p, n_images, side);
  cf32* buf[2] = {(cf32*)workspace, (cf32*)workspace + (ck.pipe ? ck.chunk * f2p_panel_elems_per_image(p) : 0)};
  const char* nok = "sc_engine: two-pass route: no kernel for this line length / kept range";
  const char* sync_fail = "sc_engine: two-pass route: event record / wait failed";
  auto ni_of = [&](int64_t c) { const int64_t i0 = c * ck.chunk; return n_images - i0 < ck.chunk ? n_images - i0 : ck.chunk; };
  if (!ck.pipe) {
    for (int64_t c = 0; c < ck.n_chunks; ++c) {
      const int64_t i0 = c * ck.chunk, ni = ni_of(c);
      if (!f2p_launch_col_inv(p, yhat + i0 * p->modes, buf[0], ni, st) ||
          !f2p_launch_c2r(p, mode, buf[0], y + i0 * p->ntot, bias, channels, i0, ni, st, skip ? skip + i0 * p->ntot : nullptr))
        return sc_fail(nok);
    }
    return sc_check_launch("k_f2p_col_inv / k_f2p_c2r");
  }
  // side: col_inv(0) col_inv(1) ...       main: c2r(0) c2r(1) ...   (c2r(c) after col_inv(c); col_inv(c + 2) after
  // c2r(c): same panel buffer)
  ScSideJoinGuard guard(side, st);
  sc_stream_t ss = (sc_stream_t)side->stream;
  if (!sc_side_fork(side, st)) return sc_fail(sync_fail);                            // side: the spectrum is ready
  guard.armed = true;
  if (!f2p_launch_col_inv(p, yhat, buf[0], ni_of(0), ss)) return sc_fail(nok);
  for (int64_t c = 0; c < ck.n_chunks; ++c) {
    const int64_t i0 = c * ck.chunk, ni = ni_of(c);
    if (!sc_side_join(side, st)) return sc_fail(sync_fail);                          // main: col_inv(<= c) are done
    guard.armed = false;
    if (c + 1 < ck.n_chunks) {
      if (!sc_side_fork(side, st)) return sc_fail(sync_fail);                        // side: c2r(<= c - 1) are done
      guard.armed = true;
      if (!f2p_launch_col_inv(p, yhat + (i0 + ck.chunk) * p->modes, buf[(c + 1) & 1], ni_of(c + 1), ss)) return sc_fail(nok);
    }
    if (!f2p_launch_c2r(p, mode, buf[c & 1], y + i0 * p->ntot, bias, channels, i0, ni, st, skip ? skip + i0 * p->ntot : nullptr))
      return sc_fail(nok);
  }
  return sc_check_launch("k_f2p_col_inv / k_f2p_c2r");
}

extern "C" int sc_plan_create(sc_plan** out, const sc_plan_desc* desc) {
  SC_CHECK_ARG(out && desc, "null argument");
  SC_CHECK_ARG(desc->ndim >= 1 && desc->ndim <= SC_MAX_DIMS, "ndim must be 1..4");
  sc_plan* p = new sc_plan();
  p->d = *desc;
  p->nd = desc->ndim;
  p->ntot = 1;
  p->modes = 1;
  p->cplx = (desc->flags & SC_PLAN_COMPLEX) != 0;
  for (int d = 0; d < p->nd; ++d) {
    p->n[d] = desc->spatial[d];
    p->k[d] = desc->kept[d];
    if (p->n[d] < 1 || p->k[d] < 1) {
      delete p;
      return sc_fail("sc_engine: spatial sizes and kept modes must be >= 1");
    }
    const bool half = (d == p->nd - 1) && !p->cplx;       // real data: the last dim has n/2+1 columns
    const int64_t lim = half ? (p->n[d] / 2 + 1) : p->n[d];
    p->fmap[d].resize((size_t)p->k[d]);
    if (desc->freq[d]) {
      p->custom_map = true;
      for (int64_t r = 0; r < p->k[d]; ++r) {
        int64_t f = desc->freq[d][r];
        if (f != SC_FREQ_DROPPED) {
          if (f <= -p->n[d] || f >= p->n[d] || (half && (f < 0 || f >= lim))) {
            delete p;
            return sc_fail("sc_engine: frequency map entry outside the grid");
          }
          if (f < 0) f += p->n[d];
        }
        p->fmap[d][(size_t)r] = f;
      }
    } else {
      if (p->k[d] > lim) {
        delete p;
        return sc_fail("sc_engine: kept modes exceed the available spectrum");
      }
      for (int64_t r = 0; r < p->k[d]; ++r) {
        int64_t f = half ? r : r - p->k[d] / 2;
        if (f < 0) f += p->n[d];
        p->fmap[d][(size_t)r] = f;
      }
    }
    p->ntot *= p->n[d];
    p->modes *= p->k[d];
  }
  p->d.real_col = (desc->real_col > 0 && desc->real_col < p->k[p->nd - 1] && !p->cplx) ? desc->real_col : 0;
  for (int d = 0; d < SC_MAX_DIMS; ++d) p->d.freq[d] = nullptr;      // host arrays are not kept
  switch (desc->fft_norm) {
    case SC_NORM_FORWARD: p->sf = 1.0 / (double)p->ntot; p->si = 1.0; break;
    case SC_NORM_BACKWARD: p->sf = 1.0; p->si = 1.0 / (double)p->ntot; break;
    case SC_NORM_ORTHO: p->sf = p->si = 1.0 / std::sqrt((double)p->ntot); break;
    default: delete p; return sc_fail("sc_engine: unknown fft_norm");
  }
  // zero-frequency position inside the kept block (-1 when a map leaves it out: no DC-based bias gradient)
  p->dc_index = 0;
  for (int d = 0; d < p->nd; ++d) {
    int64_t r0 = -1;
    for (int64_t r = 0; r < p->k[d]; ++r)
      if (p->fmap[d][(size_t)r] == 0) {
        r0 = r;
        break;
      }
    if (r0 < 0 || p->dc_index < 0) p->dc_index = -1;
    else p->dc_index = p->dc_index * p->k[d] + r0;
  }

  const int L = p->nd - 1;
  const int64_t N = p->n[L], J = p->k[L];
  int rc = 0;
  if (p->cplx) {
    // complex data: the last dim is one more complex-to-complex pass (inner = 1) with the norm folded in
    static const int cc[] = {4, 8, 16};
    p->cx_fwd_jt = pick_tile(J, cc, 3, 1);
    p->cx_inv_jt = pick_tile(N, cc, 3, 1);
    for (int v = 0; v < 2 && !rc; ++v) {
      const int64_t Jpad = round_up(J, p->cx_fwd_jt);
      std::vector<cf32> h((size_t)N * Jpad, cf_make(0.f, 0.f));
      for (int64_t n = 0; n < N; ++n)
        for (int64_t j = 0; j < J; ++j) h[(size_t)n * Jpad + j] = axis_tw(p, L, j, n, -1.0, v == SC_FWD_SCALED ? p->sf : p->si);
      rc = upload_table(p, h, (int)N, (int)Jpad, &p->cx_fwd[v]);
    }
    for (int v = 0; v < 2 && !rc; ++v) {
      const int64_t Npad = round_up(N, p->cx_inv_jt);
      std::vector<cf32> h((size_t)J * Npad, cf_make(0.f, 0.f));
      for (int64_t j = 0; j < J; ++j)
        for (int64_t n = 0; n < N; ++n) h[(size_t)j * Npad + n] = axis_tw(p, L, j, n, +1.0, v == SC_INV_PADDED ? p->si : p->sf);
      rc = upload_table(p, h, (int)J, (int)Npad, &p->cx_inv[v]);
    }
  } else {
    static const int c1[] = {2, 3, 4, 5, 8, 9, 16, 17};
    p->r2c_jt = pick_tile(J, c1, 8, 4);
    const int64_t Jpad = round_up(J, 4 * p->r2c_jt);
    for (int v = 0; v < 2 && !rc; ++v) {
      std::vector<cf32> h((size_t)N * Jpad, cf_make(0.f, 0.f));
      for (int64_t n = 0; n < N; ++n)
        for (int64_t j = 0; j < J; ++j) {
          h[(size_t)n * Jpad + j] = last_tw(p, j, n, -1.0, v != SC_FWD_SCALED);
        }
      rc = upload_table(p, h, (int)N, (int)Jpad, &p->r2c[v]);
    }
    static const int c2[] = {4, 8, 16};
    p->c2r_nt = pick_tile(N, c2, 3, 4);
    const int64_t Npad = round_up(N, 4 * p->c2r_nt);
    for (int v = 0; v < 2 && !rc; ++v) {
      std::vector<cf32> h((size_t)J * Npad, cf_make(0.f, 0.f));
      for (int64_t j = 0; j < J; ++j)
        for (int64_t n = 0; n < N; ++n) {
          h[(size_t)j * Npad + n] = last_tw(p, j, n, +1.0, v == SC_INV_PADDED);
        }
      rc = upload_table(p, h, (int)J, (int)Npad, &p->c2r[v]);
    }
  }
  static const int c3[] = {4, 8, 16};
  for (int d = 0; d < L && !rc; ++d) {
    const int64_t Nd = p->n[d], Kd = p->k[d];
    p->ax_fwd_jt[d] = pick_tile(Kd, c3, 3, 1);
    p->ax_inv_jt[d] = pick_tile(Nd, c3, 3, 1);
    {
      const int64_t Jpad = round_up(Kd, p->ax_fwd_jt[d]);
      std::vector<cf32> h((size_t)Nd * Jpad, cf_make(0.f, 0.f));
      for (int64_t n = 0; n < Nd; ++n)
        for (int64_t j = 0; j < Kd; ++j) h[(size_t)n * Jpad + j] = axis_tw(p, d, j, n, -1.0);
      rc = upload_table(p, h, (int)Nd, (int)Jpad, &p->ax_fwd[d]);
    }
    if (!rc) {
      const int64_t Jpad = round_up(Nd, p->ax_inv_jt[d]);
      std::vector<cf32> h((size_t)Kd * Jpad, cf_make(0.f, 0.f));
      for (int64_t kk = 0; kk < Kd; ++kk)
        for (int64_t hh = 0; hh < Nd; ++hh) h[(size_t)kk * Jpad + hh] = axis_tw(p, d, kk, hh, +1.0);
      rc = upload_table(p, h, (int)Kd, (int)Jpad, &p->ax_inv[d]);
    }
  }
  if (!rc && !(desc->flags & SC_PLAN_FORCE_GENERIC) && !p->custom_map && !p->cplx) {
    std::string why;
    if (fft2d_plan_init(&p->fft2d, p->nd, p->n, p->k, p->sf, p->si, &p->owned, &why)) p->fast = true;
  }
  // bfloat16 I/O on the fused kernels: the operand fragments of the matrix-core row pass (12 KB, sc_kernels_fft3mx.h)
  if (!rc && p->fast && (desc->flags & SC_PLAN_IO_BF16) && !(desc->flags & (SC_PLAN_FFT_GEN2 | SC_PLAN_NO_MX_FFT)) &&
      p->fft2d.H <= 256 && !getenv("SC_PLAN_NO_MX_FFT")) {
    for (int which = 0; which < 2 && !rc; ++which) {       // forward-type rows in, inverse-type rows out
      std::vector<uint16_t> h;
      p->fft2d.mx_terms = (desc->flags & SC_PLAN_MX_FFT_3TERM) ? 3 : 2;
      if (which == 0) fft3mx_build_table(&h, p->fft2d.mx_terms); else fft3mxi_build_table(&h);
      void* dev = nullptr;
      if (hipMalloc(&dev, h.size() * sizeof(uint16_t)) != hipSuccess) {
        rc = sc_fail("sc_engine: table allocation failed");
      } else {
        p->owned.push_back(dev);
        if (hipMemcpy(dev, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess)
          rc = sc_fail("sc_engine: table upload failed");
        else
          (which == 0 ? p->fft2d.tabF : p->fft2d.tabG) = (uint16_t*)dev;
      }
    }
  }
  if (!rc && !p->fast && !(desc->flags & SC_PLAN_FORCE_GENERIC) && !(desc->flags & SC_PLAN_IO_BF16))
    rc = f2p_plan_init(p, true);
  if (!rc && !p->fast && !p->f2p && !(desc->flags & (SC_PLAN_FORCE_GENERIC | SC_PLAN_IO_BF16 | SC_PLAN_NO_MDFT)))
    rc = pl128_plan_init(p);
  if (!rc && !p->fast && !p->f2p && !p->pl128 &&
      !(desc->flags & (SC_PLAN_FORCE_GENERIC | SC_PLAN_IO_BF16 | SC_PLAN_NO_MDFT | SC_PLAN_F2P_SMALL_ALWAYS)) &&
      !SC_DIAG_ENV("SC_PLAN_NO_PL64"))
    rc = pl64_plan_init(p);
  if (!rc && !p->fast && !p->f2p && !p->pl128 && !p->pl64 &&
      !(desc->flags & (SC_PLAN_FORCE_GENERIC | SC_PLAN_IO_BF16 | SC_PLAN_NO_MDFT | SC_PLAN_NO_F2P_SMALL)))
    rc = f2p_plan_init(p, false);
  if (!rc && (desc->flags & SC_PLAN_IO_BF16) && (!p->fast || (desc->flags & SC_PLAN_FFT_GEN2)))
    rc = sc_fail("sc_engine: SC_PLAN_IO_BF16 is implemented on the fused 2-D kernels (generation 3) only: "
                 "width 256, height 64..512, kept block <= 64 x 33, no frequency maps");
  if (!rc && !p->fast && !p->f2p && !(desc->flags & SC_PLAN_NO_MDFT)) rc = build_mdft_tables(p);
  if (rc) {
    sc_plan_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}

extern "C" void sc_plan_destroy(sc_plan* p) {
  if (!p) return;

  for (void* q : p->owned) (void)hipFree(q);
  for (auto& kv : p->idx_cache) (void)hipFree(kv.second);
  delete p;
}

extern "C" int sc_plan_is_fast(const sc_plan* p) { return p && p->fast ? 1 : 0; }

// intermediate sizes (complex elements) of the generic pass chain
static void generic_ws_sizes(const sc_plan* p, int64_t n_images, int64_t* s1, int64_t* s2) {
  const int L = p->nd - 1;
  int64_t lines = n_images;
  for (int d = 0; d < L; ++d) lines *= p->n[d];
  *s1 = (p->nd >= 2) ? lines * p->k[L] : 0;
  if (p->nd >= 3) {
    int64_t o = n_images;
    for (int d = 0; d < L - 1; ++d) o *= p->n[d];
    *s2 = o * p->k[L - 1] * p->k[L];
  } else {
    *s2 = 0;
  }
}

extern "C" size_t sc_plan_workspace_bytes(const sc_plan* p, int64_t n_images) {
  if (!p) return 0;
  if (p->fast) return fft2d_workspace_bytes(&p->fft2d, n_images);
  if (p->f2p) return (size_t)(f2p_chunk_images(p, n_images) * f2p_panel_elems_per_image(p)) * sizeof(cf32) + 256;
  int64_t s1, s2;
  generic_ws_sizes(p, n_images, &s1, &s2);
  return (size_t)(s1 + s2) * sizeof(cf32) + 256;
}

// ------------------------------------------------------------------------------------------
// generic pass launchers
// ------------------------------------------------------------------------------------------
template <int JT>
static void launch_r2c(const float* in, cf32* out, const DeviceTable& t, int64_t lines, int N, int J,
                       sc_stream_t st) {
  dim3 grid((unsigned)((lines + SC_LINES_PER_BLOCK - 1) / SC_LINES_PER_BLOCK),
            (unsigned)((J + 4 * JT - 1) / (4 * JT)));
  SC_LAUNCH((k_last_r2c<JT>), grid, dim3(SC_BLOCK), 0, st, in, out, (const cf32*)t.ptr, lines, N, J,
            t.cols_pad);
}

// 16-byte aligned tensor storage (the wide loads / stores of the matrix-core and plane passes)
static bool sc_io_aligned(const void* io) { return !(((uintptr_t)io) & 15); }

template <int RT, int CT, bool TAIL>
static void launch_mdft_r2c(const float* in, cf32* out, const float* tab, const cf32* tail, int64_t lines, int N,
                            int J, int n_ct, sc_stream_t st) {
  const int64_t items = (lines + 32 * RT - 1) / (32 * RT);
  const dim3 grid((unsigned)((items + 3) / 4));
  if (N % 8 == 0 && sc_io_aligned(in))     // 8-byte loads
    SC_LAUNCH((k_mdft_r2c<RT, CT, TAIL, false>), grid, dim3(256), 0, st, in, (float*)out, tab, tail, lines, N, J, n_ct);
  else                                     // any width / any 4-byte aligned view: clamped 4-byte loads
    SC_LAUNCH((k_mdft_r2c<RT, CT, TAIL, true>), grid, dim3(256), 0, st, in, (float*)out, tab, tail, lines, N, J, n_ct);
}

template <bool TAIL>
static void dispatch_mdft_r2c(const float* in, cf32* out, const float* tab, const cf32* tail, int64_t lines, int N,
                              int J, int n_ct, sc_stream_t st) {
  // 8 accumulator tiles per wave by default; SC_MDFT_TILE=4 selects 4-tile waves (2-3 waves per
  // SIMD) for A-B: measured equal on 128^3 (5.38 vs 5.43 ms/step) and slower on 1024^2 (33.1 vs 30.5)
  const bool big = !mdft_switches().tile4;
  if (n_ct == 1) {                       // J = 17 with the tail column on the VALU: no second column tile to pad
    if (big) launch_mdft_r2c<4, 1, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
    else launch_mdft_r2c<2, 1, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
  } else if (big) {
    if (n_ct <= 2) launch_mdft_r2c<4, 2, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
    else if (n_ct <= 4) launch_mdft_r2c<2, 4, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
    else launch_mdft_r2c<1, 8, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
  } else {
    if (n_ct <= 2) launch_mdft_r2c<2, 2, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
    else launch_mdft_r2c<1, 4, TAIL>(in, out, tab, tail, lines, N, J, n_ct, st);
  }
}

// tiles per block of the LDS-staged passes: amortise the table copy once the grid covers the chip 8 x over
static int mdft_lds_tiles_per_block(int64_t lines) {
  const int64_t n_tiles = (lines + SC_MDFT_LB - 1) / SC_MDFT_LB;
  int64_t tpb = n_tiles / 2048;
  if (tpb < 1) tpb = 1;
  if (tpb > 16) tpb = 16;
  return (int)tpb;
}

template <int CT, bool TAIL, int JP = 0, int NR = SC_MDFT_LB>
static void launch_mdft_r2c_lds(const float* in, cf32* out, const float* tab, const cf32* tail, int64_t lines, int N,
                                int J, sc_stream_t st, const float* tab1 = nullptr, int K1 = 0) {
  const int tpb = mdft_lds_tiles_per_block(lines);
  const int64_t n_tiles = (lines + SC_MDFT_LB - 1) / SC_MDFT_LB;
  SC_LAUNCH((k_mdft_r2c_lds<CT, TAIL, JP, NR>), dim3((unsigned)((n_tiles + tpb - 1) / tpb)), dim3(256),
            (size_t)(N / 8) * CT * 1024, st, in, (float*)out, tab, tail, lines, N, J, tpb, tab1, K1);
}

template <int CT, bool TAIL>
static void launch_mdft_r2c_stage(const float* in, cf32* out, const float* tab, const cf32* tail, int64_t lines, int N,
                                  int J, sc_stream_t st) {
  const int tpb = mdft_lds_tiles_per_block(lines);
  const int64_t n_tiles = (lines + SC_MDFT_LB - 1) / SC_MDFT_LB;
  SC_LAUNCH((k_mdft_r2c_stage<CT, TAIL>), dim3((unsigned)((n_tiles + tpb - 1) / tpb)), dim3(256), 0, st, in,
            (float*)out, tab, tail, lines, N, J, tpb);
}

#define SC_C2R_NPF 12          // registers per thread holding the next tile's input (k_mdft_c2r_lds)
// ---- "plane" form: the last TWO axes in one launch when the second-to-last has 128, 64 or 32 rows --------
static bool plane_rows_ok(int64_t nr) { return nr == 128 || nr == 64 || nr == 32; }
// io = the real tensor the pass reads (forward) / writes (inverse); nullptr = "aligned".  The plane kernels and the
// matrix-core DFT passes move its rows with 16-byte accesses.  A contiguous view with an odd storage offset
// (flat[1:1 + n].view(...), a gradient handed over out of a bucketed buffer) takes the straight-from-global matrix-core
// passes with 4-byte accesses for the real-side pass instead (k_mdft_r2c<.., RAGGED> / k_mdft_c2r; the VALU passes
// k_last_r2c / k_last_c2r under SC_PLAN_NO_MDFT) -- the call does not fail.
static bool plane_fwd_ok(const sc_plan* p, int mode, const void* io = nullptr) {
  if (!sc_io_aligned(io)) return false;
  if (p->pl128 || p->pl64) return true;
  const int L = p->nd - 1;
  return p->nd >= 2 && p->mdft && !p->cplx && p->l_r2c[mode] && p->m_pl_fwd && plane_rows_ok(p->n[L - 1]) &&
         2 * p->k[L - 1] <= p->n[L - 1] && !mdft_switches().nolds && !mdft_switches().noplane;
}
static bool plane_inv_ok(const sc_plan* p, int mode, const void* io = nullptr) {
  if (!sc_io_aligned(io)) return false;
  if (p->pl128 || p->pl64) return true;
  const int L = p->nd - 1;
  if (!(p->nd >= 2 && p->mdft && !p->cplx && p->l_c2r[mode] && p->m_pl_inv && plane_rows_ok(p->n[L - 1]) &&
        2 * p->k[L - 1] <= p->n[L - 1] && !mdft_switches().nolds && !mdft_switches().noplane))
    return false;
  const int64_t N = p->n[L], J = p->k[L], n_nt = (N + 31) / 32, JS = (J + 1) / 2;
  const int64_t pl = SC_MDFT_LB / p->n[L - 1];
  const int64_t bytes = (n_nt * JS * 128 + SC_MDFT_LB * p->l_c2r_s + SC_C2R_PATCH_FLOATS) * 4 + pl * p->k[L - 1] * J * 8;
  const bool atail = J > 1 && (2 * J) % 32 == 2 && 2 * J > 32;
  const int64_t ca = ((atail ? 2 * J - 2 : 2 * J) + 31) / 32;
  return bytes <= 60 * 1024 && pl * p->k[L - 1] * J <= 256 * SC_C2R_NPF && ca <= 2;
}

template <int CT, bool TAIL>
static void dispatch_plane_fwd(const sc_plan* p, int mode, const float* in, cf32* out, int64_t lines, sc_stream_t st) {
  const int L = p->nd - 1;
  const int N = (int)p->n[L], J = (int)p->k[L], K1 = (int)p->k[L - 1];
  const float* tab = p->l_r2c[mode];
  const cf32* tail = p->l_r2c_tail[mode];
  const float* t1 = p->m_pl_fwd;
  const int64_t nr = p->n[L - 1];                      // K1 <= nr / 2 (plane_fwd_ok)
  if (nr == 128) {
    if (K1 <= 32) launch_mdft_r2c_lds<CT, TAIL, 1, 128>(in, out, tab, tail, lines, N, J, st, t1, K1);
    else launch_mdft_r2c_lds<CT, TAIL, 2, 128>(in, out, tab, tail, lines, N, J, st, t1, K1);
  } else if (nr == 64) {
    launch_mdft_r2c_lds<CT, TAIL, 1, 64>(in, out, tab, tail, lines, N, J, st, t1, K1);
  } else {
    launch_mdft_r2c_lds<CT, TAIL, 1, 32>(in, out, tab, tail, lines, N, J, st, t1, K1);
  }
}

// x (planes x 128 x N real) -> (planes x K1 x J complex): last axis + second-to-last axis
static int run_plane_fwd(const sc_plan* p, int mode, const float* in, cf32* out, int64_t lines, sc_stream_t st) {
  if (!sc_io_aligned(in)) return sc_fail("sc_engine: the plane kernels need 16-byte aligned planes (plane_fwd_ok)");
  if (p->pl128) {
    const int L = p->nd - 1;
    // planes per workgroup: the next plane's first rows are in flight while this one is finished (A-B: SC_PL_PPW)
    static const int ppw_env = [] { const char* e = SC_DIAG_ENV("SC_PL_PPW"); return e ? std::atoi(e) : 0; }();
    const int64_t n_planes = lines / SC_PL_N;
    int ppw = ppw_env > 0 ? ppw_env : SC_PL_PPW_DEFAULT;
    while (ppw > 1 && n_planes / ppw < (int64_t)6 * sc_cu_count()) --ppw;     // keep two full rounds of workgroups
    SC_LAUNCH(k_pl128_fwd, dim3((unsigned)((n_planes + ppw - 1) / ppw)), dim3(256), 0, st, in, out,
              (const cf32*)p->pl_tab128, (const float*)p->pl_cs_fwd[mode], (int)p->k[L - 1], (int)p->k[L], n_planes, ppw);
    return sc_check_launch("k_pl128_fwd");
  }
  if (p->pl64) {
    const int L = p->nd - 1;
    // planes per workgroup: the next plane's rows are in flight while this one is transformed (A-B: SC_P64_PPW)
    static const int ppw_env = [] { const char* e = SC_DIAG_ENV("SC_P64_PPW"); return e ? std::atoi(e) : 0; }();
    const int64_t n_planes = lines / SC_P64_N;
    int ppw = ppw_env > 0 ? ppw_env : SC_P64_PPW_DEFAULT;
    while (ppw > 1 && n_planes / ppw < (int64_t)8 * sc_cu_count()) --ppw;     // keep two full rounds of workgroups
    SC_LAUNCH(k_pl64_fwd, dim3((unsigned)((n_planes + ppw - 1) / ppw)), dim3(256), 0, st, in, out,
              (const cf32*)p->pl_tab128, (const float*)p->pl_cs_fwd[mode], (int)p->k[L - 1], (int)p->k[L], n_planes, ppw);
    return sc_check_launch("k_pl64_fwd");
  }
  const bool tail = p->l_r2c_tail[mode] != nullptr;
  if (tail) {
    if (p->l_r2c_ct == 1) dispatch_plane_fwd<1, true>(p, mode, in, out, lines, st);
    else dispatch_plane_fwd<2, true>(p, mode, in, out, lines, st);
  } else {
    if (p->l_r2c_ct == 1) dispatch_plane_fwd<1, false>(p, mode, in, out, lines, st);
    else dispatch_plane_fwd<2, false>(p, mode, in, out, lines, st);
  }
  return sc_check_launch("k_mdft_r2c_lds<plane>");
}

static int run_r2c(const sc_plan* p, int mode, const float* in, cf32* out, int64_t lines, sc_stream_t st) {
  const int L = p->nd - 1;
  const int N = (int)p->n[L], J = (int)p->k[L];
  const bool al = sc_io_aligned(in);                   // 16-byte loads in the matrix-core passes
  if (al && p->mdft && p->l_r2c[mode] && lines < ((int64_t)1 << 36) && !mdft_switches().nolds) {
    const float* tab = p->l_r2c[mode];
    const cf32* tail = p->l_r2c_tail[mode];
    if (tail) {
      if (p->l_r2c_ct == 1) launch_mdft_r2c_lds<1, true>(in, out, tab, tail, lines, N, J, st);
      else launch_mdft_r2c_lds<2, true>(in, out, tab, tail, lines, N, J, st);
    } else {
      if (p->l_r2c_ct == 1) launch_mdft_r2c_lds<1, false>(in, out, tab, tail, lines, N, J, st);
      else launch_mdft_r2c_lds<2, false>(in, out, tab, tail, lines, N, J, st);
    }
    return sc_check_launch("k_mdft_r2c_lds");
  }
  if (p->mdft && p->s_r2c[mode] && lines < ((int64_t)1 << 36)) {       // 4-byte loads: any view
    const float* tab = p->s_r2c[mode];
    const cf32* tail = p->s_r2c_tail[mode];
    if (tail) {
      if (p->s_r2c_ct == 1) launch_mdft_r2c_stage<1, true>(in, out, tab, tail, lines, N, J, st);
      else launch_mdft_r2c_stage<2, true>(in, out, tab, tail, lines, N, J, st);
    } else {
      if (p->s_r2c_ct == 1) launch_mdft_r2c_stage<1, false>(in, out, tab, tail, lines, N, J, st);
      else launch_mdft_r2c_stage<2, false>(in, out, tab, tail, lines, N, J, st);
    }
    return sc_check_launch("k_mdft_r2c_stage");
  }
  if (p->mdft && p->m_r2c[mode] && lines < ((int64_t)1 << 36)) {
    if (p->m_r2c_tail[mode] && !mdft_switches().notail && 2 * J > 32)
      dispatch_mdft_r2c<true>(in, out, p->m_r2c[mode], p->m_r2c_tail[mode], lines, N, J, (2 * J - 2) / 32, st);
    else
      dispatch_mdft_r2c<false>(in, out, p->m_r2c[mode], nullptr, lines, N, J, (2 * J + 31) / 32, st);
    return sc_check_launch("k_mdft_r2c");
  }
  const DeviceTable& t = p->r2c[mode];
  switch (p->r2c_jt) {
    case 2: launch_r2c<2>(in, out, t, lines, N, J, st); break;
    case 3: launch_r2c<3>(in, out, t, lines, N, J, st); break;
    case 4: launch_r2c<4>(in, out, t, lines, N, J, st); break;
    case 5: launch_r2c<5>(in, out, t, lines, N, J, st); break;
    case 8: launch_r2c<8>(in, out, t, lines, N, J, st); break;
    case 9: launch_r2c<9>(in, out, t, lines, N, J, st); break;
    case 16: launch_r2c<16>(in, out, t, lines, N, J, st); break;
    case 17: launch_r2c<17>(in, out, t, lines, N, J, st); break;
    default: return sc_fail("sc_engine: bad r2c tile");
  }
  return sc_check_launch("k_last_r2c");
}

template <int NT>
static void launch_c2r(const cf32* in, float* out, const DeviceTable& t, const float* bias, int64_t lines,
                       int N, int J, int64_t lpi, int64_t channels, sc_stream_t st) {
  dim3 grid((unsigned)((lines + SC_LINES_PER_BLOCK - 1) / SC_LINES_PER_BLOCK),
            (unsigned)((N + 4 * NT - 1) / (4 * NT)));
  SC_LAUNCH((k_last_c2r<NT>), grid, dim3(SC_BLOCK), 0, st, in, out, (const cf32*)t.ptr, bias, lines, N, J,
            t.cols_pad, lpi, channels);
}

template <int RT, int CT>
static void launch_mdft_c2r(const cf32* in, float* out, const float* tab, const float* bias, int64_t lines, int N,
                            int J, int n_nt, int64_t lpi, int64_t channels, sc_stream_t st) {
  const int64_t items = (lines + 32 * RT - 1) / (32 * RT);
  // the bias is one scalar per wave when a wave's 32 RT lines cannot straddle two images, a per-row lookup otherwise
  const int per_line = (bias != nullptr && lpi % (32 * RT) != 0) ? 1 : 0;
  SC_LAUNCH((k_mdft_c2r<RT, CT>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, in, out, tab, bias, lines, N,
            J, n_nt, lpi, channels, per_line);
}

template <int CT, int NR, int NPF, int CA = 1, bool ATAIL = false>
static void launch_mdft_c2r_lds_npf(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias,
                                    int64_t lines, int N, int J, int64_t lpi, int64_t channels, sc_stream_t st);

template <int CT, int NR = 0>
static void launch_mdft_c2r_lds(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias,
                                int64_t lines, int N, int J, int64_t lpi, int64_t channels, sc_stream_t st) {
  // next-tile prefetch when a thread's share of a tile's input fits SC_C2R_NPF registers
  const int L = p->nd - 1;
  const int64_t per_tile = NR > 0 ? (int64_t)(SC_MDFT_LB / (NR > 0 ? NR : SC_MDFT_LB)) * p->k[L - 1 >= 0 ? L - 1 : 0] * J
                                  : (int64_t)SC_MDFT_LB * J;
  if (NR > 0) {                    // plane form (plane_inv_ok: the planes of a tile fit the prefetch registers)
    const bool atail = J > 1 && (2 * J) % 32 == 2 && 2 * J > 32;
    const int ca = ((atail ? 2 * J - 2 : 2 * J) + 31) / 32;
    if (atail && ca == 1) launch_mdft_c2r_lds_npf<CT, NR, (NR ? SC_C2R_NPF : 0), 1, true>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else if (atail) launch_mdft_c2r_lds_npf<CT, NR, (NR ? SC_C2R_NPF : 0), 2, true>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else if (ca == 1) launch_mdft_c2r_lds_npf<CT, NR, (NR ? SC_C2R_NPF : 0), 1, false>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else launch_mdft_c2r_lds_npf<CT, NR, (NR ? SC_C2R_NPF : 0), 2, false>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
  } else if (per_tile <= 256 * SC_C2R_NPF) {
    launch_mdft_c2r_lds_npf<CT, NR, SC_C2R_NPF>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
  } else {
    launch_mdft_c2r_lds_npf<CT, NR, 0>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
  }
}

template <int CT, int NR, int NPF, int CA, bool ATAIL>
static void launch_mdft_c2r_lds_npf(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias,
                                    int64_t lines, int N, int J, int64_t lpi, int64_t channels, sc_stream_t st) {
  constexpr bool PLANE = NR > 0;
  const int tpb = mdft_lds_tiles_per_block(lines);
  const int64_t n_tiles = (lines + SC_MDFT_LB - 1) / SC_MDFT_LB;
  const int n_nt = (N + 31) / 32, JS = (J + 1) / 2;
  const int L = p->nd - 1;
  const int K1 = PLANE ? (int)p->k[L - 1] : 0;
  const size_t lds = ((size_t)n_nt * JS * 128 + (size_t)SC_MDFT_LB * p->l_c2r_s + SC_C2R_PATCH_FLOATS) * sizeof(float) +
                     (size_t)(PLANE ? SC_MDFT_LB / NR : 0) * K1 * J * sizeof(cf32);
  SC_LAUNCH((k_mdft_c2r_lds<CT, NR, NPF, CA, ATAIL>), dim3((unsigned)((n_tiles + tpb - 1) / tpb)), dim3(256), lds, st, in, out,
            (const float*)p->l_c2r[mode], bias, lines, N, J, n_nt, p->l_c2r_s, lpi, channels, tpb,
            PLANE ? (const float*)p->m_pl_inv : (const float*)nullptr, K1);
}

template <int JS2>
static void launch_mdft_c2r_stage(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias,
                                  int64_t lines, int N, int J, int64_t lpi, int64_t channels, sc_stream_t st) {
  const int64_t n_tiles = (lines + SC_MDFT_LB - 1) / SC_MDFT_LB;
  const int n_nt = (N + 31) / 32;
  // ranges of column tiles per 128-line tile: ~6 blocks per resident slot (4 per CU), so that the last round of
  // blocks is a small share of the launch; every range re-reads the tile's spectrum (8 % of the bytes, from L2)
  int64_t splits = ((int64_t)24 * sc_cu_count() + n_tiles - 1) / n_tiles;
  static const int split_env = [] {                          // A-B only (scripts/mdft_time.py)
    const char* e = SC_DIAG_ENV("SC_C2R_STAGE_SPLITS");
    return e ? atoi(e) : 0;
  }();
  if (split_env > 0) splits = split_env;
  if (splits < 1) splits = 1;
  if (splits > n_nt) splits = n_nt;
  const int nt_per = (int)((n_nt + splits - 1) / splits);
  const unsigned gy = (unsigned)((n_nt + nt_per - 1) / nt_per);
  SC_LAUNCH((k_mdft_c2r_stage<JS2>), dim3((unsigned)n_tiles, gy), dim3(256), (size_t)SC_MDFT_LB * p->s_c2r_s * sizeof(float),
            st, in, out, (const float*)p->s_c2r[mode], bias, lines, N, J, n_nt, p->s_c2r_s, lpi, channels, nt_per);
}

// whole-line form of the same pass: one block per 32 lines, their N-line span staged in LDS (k_mdft_c2r_span)
static size_t c2r_span_lds(const sc_plan* p, int N) {         // spectrum rows and span image share the buffer
  return (size_t)32 * (N > p->s_c2r_s ? N : p->s_c2r_s) * sizeof(float);
}
static bool c2r_span_ok(const sc_plan* p, int N, const float* out, int64_t lines) {
  static const bool off = SC_DIAG_ENV("SC_C2R_NOSPAN") != nullptr;             // A-B against k_mdft_c2r_stage
  return !off && !(p->d.flags & SC_PLAN_NO_SPAN) && sc_io_aligned(out) && c2r_span_lds(p, N) <= (size_t)80 * 1024 &&
         (lines + 31) / 32 < ((int64_t)1 << 31);
}
template <int JS2>
static int launch_mdft_c2r_span(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias,
                                int64_t lines, int N, int J, int64_t lpi, int64_t channels, sc_stream_t st) {
  const size_t lds = c2r_span_lds(p, N);
#ifndef SC_EMU
  // once per instantiation and device: the 80 KB ceiling.  The kernel also holds 128 B of static LDS (biasL), so the
  // ceiling is raised as soon as static + dynamic reaches the 64 KB default (N = 512, J = 9..36: dynamic = 65536 exactly;
  // ADVICE r4).  The flags are atomics: plans of two host threads may take their first launch at the same time.
  if (lds + 128 >= 64 * 1024) {
    static std::atomic<bool> raised[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised[dev].load(std::memory_order_acquire)) {
      SC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mdft_c2r_span<JS2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      raised[dev].store(true, std::memory_order_release);
    }
  }
#endif
  SC_LAUNCH((k_mdft_c2r_span<JS2>), dim3((unsigned)((lines + 31) / 32)), dim3(256), lds, st, in, out,
            (const float*)p->s_c2r[mode], bias, lines, N, J, (N + 31) / 32, p->s_c2r_s, lpi, channels);
  return sc_check_launch("k_mdft_c2r_span");
}

// (planes x K1 x J complex) -> y (planes x 128 x N real): second-to-last axis + last axis (+ bias)
static int run_plane_inv(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias, int64_t lines,
                         int64_t lpi, int64_t channels, sc_stream_t st) {
  const int L = p->nd - 1;
  if (!sc_io_aligned(out)) return sc_fail("sc_engine: the plane kernels need 16-byte aligned planes (plane_inv_ok)");
  if (p->pl128) {
    SC_LAUNCH(k_pl128_inv, dim3((unsigned)(lines / SC_PL_N)), dim3(256), 0, st, in, out, (const cf32*)p->pl_tab128,
              (const float*)p->pl_cs_inv[mode], bias, lpi / SC_PL_N, (int)channels, (int)p->k[L - 1], (int)p->k[L]);
    return sc_check_launch("k_pl128_inv");
  }
  if (p->pl64) {
    SC_LAUNCH(k_pl64_inv, dim3((unsigned)(lines / SC_P64_N)), dim3(256), 0, st, in, out, (const cf32*)p->pl_tab128,
              (const float*)p->pl_cs_inv[mode], bias, lpi / SC_P64_N, (int)channels, (int)p->k[L - 1], (int)p->k[L]);
    return sc_check_launch("k_pl64_inv");
  }
  const int N = (int)p->n[L], J = (int)p->k[L];
  const int n_nt = (N + 31) / 32;
  const int64_t nr = p->n[L - 1];
#define SC_PLANE_INV(NRV)                                                                                     \
  do {                                                                                                        \
    if (n_nt >= 4) launch_mdft_c2r_lds<4, NRV>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);       \
    else if (n_nt >= 2) launch_mdft_c2r_lds<2, NRV>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);  \
    else launch_mdft_c2r_lds<1, NRV>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);                 \
  } while (0)
  if (nr == 128) SC_PLANE_INV(128);
  else if (nr == 64) SC_PLANE_INV(64);
  else SC_PLANE_INV(32);
#undef SC_PLANE_INV
  return sc_check_launch("k_mdft_c2r_lds<plane>");
}

static int run_c2r(const sc_plan* p, int mode, const cf32* in, float* out, const float* bias, int64_t lines,
                   int64_t lpi, int64_t channels, sc_stream_t st) {
  const int L = p->nd - 1;
  const int N = (int)p->n[L], J = (int)p->k[L];
  const bool al = sc_io_aligned(out);                  // 16-byte stores in the matrix-core passes
  if (al && p->mdft && p->l_c2r[mode] && lines < ((int64_t)1 << 36) && (bias == nullptr || lpi % 32 == 0) &&
      !mdft_switches().nolds) {
    const int n_nt = (N + 31) / 32;
    if (n_nt >= 4) launch_mdft_c2r_lds<4>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else if (n_nt >= 2) launch_mdft_c2r_lds<2>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else launch_mdft_c2r_lds<1>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    return sc_check_launch("k_mdft_c2r_lds");
  }
  if (p->mdft && p->s_c2r[mode] && lines < ((int64_t)1 << 36) && (lines + SC_MDFT_LB - 1) / SC_MDFT_LB < ((int64_t)1 << 31)) {
    if (c2r_span_ok(p, N, out, lines)) {
      if (p->s_c2r_js2 == 3) return launch_mdft_c2r_span<3>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
      if (p->s_c2r_js2 == 5) return launch_mdft_c2r_span<5>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
      return launch_mdft_c2r_span<9>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    }
    if (p->s_c2r_js2 == 3) launch_mdft_c2r_stage<3>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else if (p->s_c2r_js2 == 5) launch_mdft_c2r_stage<5>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    else launch_mdft_c2r_stage<9>(p, mode, in, out, bias, lines, N, J, lpi, channels, st);
    return sc_check_launch("k_mdft_c2r_stage");
  }
  if (p->mdft && p->m_c2r[mode] && lines < ((int64_t)1 << 36)) {       // 4-byte stores: any view
    const int n_nt = (N + 31) / 32;
    const bool big = !mdft_switches().tile4;
    const int rt = big ? (n_nt <= 2 ? 4 : (n_nt <= 4 ? 2 : 1)) : (n_nt <= 2 ? 2 : 1);
    {
      if (big) {
        if (rt == 4) launch_mdft_c2r<4, 2>(in, out, p->m_c2r[mode], bias, lines, N, J, n_nt, lpi, channels, st);
        else if (rt == 2) launch_mdft_c2r<2, 4>(in, out, p->m_c2r[mode], bias, lines, N, J, n_nt, lpi, channels, st);
        else launch_mdft_c2r<1, 8>(in, out, p->m_c2r[mode], bias, lines, N, J, n_nt, lpi, channels, st);
      } else {
        if (rt == 2) launch_mdft_c2r<2, 2>(in, out, p->m_c2r[mode], bias, lines, N, J, n_nt, lpi, channels, st);
        else launch_mdft_c2r<1, 4>(in, out, p->m_c2r[mode], bias, lines, N, J, n_nt, lpi, channels, st);
      }
      return sc_check_launch("k_mdft_c2r");
    }
  }
  const DeviceTable& t = p->c2r[mode];
  switch (p->c2r_nt) {
    case 4: launch_c2r<4>(in, out, t, bias, lines, N, J, lpi, channels, st); break;
    case 8: launch_c2r<8>(in, out, t, bias, lines, N, J, lpi, channels, st); break;
    case 16: launch_c2r<16>(in, out, t, bias, lines, N, J, lpi, channels, st); break;
    default: return sc_fail("sc_engine: bad c2r tile");
  }
  return sc_check_launch("k_last_c2r");
}

template <int JT>
static void launch_axis(const cf32* in, cf32* out, const DeviceTable& t, int64_t outer, int N, int J,
                        int64_t inner, sc_stream_t st) {
  const int64_t cols = outer * inner;
  dim3 grid((unsigned)((cols + SC_BLOCK - 1) / SC_BLOCK), (unsigned)((J + JT - 1) / JT));
  SC_LAUNCH((k_axis_pass<JT>), grid, dim3(SC_BLOCK), 0, st, in, out, (const cf32*)t.ptr, outer, N, J, inner,
            t.cols_pad);
}

template <int JT, int CT>
static void launch_mdft_axis(const cf32* in, cf32* out, const float* tab, int64_t outer, int N, int J, int64_t inner,
                             int n_jt, sc_stream_t st) {
  const int64_t items = (outer * inner + 32 * CT - 1) / (32 * CT);
  const unsigned gy = (unsigned)((n_jt + JT - 1) / JT);        // one group of JT j-tiles per block
  // few items and a long reduction: the four waves of a block share one item and split its N rows (KS)
  if constexpr (JT * CT <= 4) {                                // (48 KB of partial sums at most)
    if (N >= 64 && items * gy < (int64_t)8 * sc_cu_count()) {
      SC_LAUNCH((k_mdft_axis<JT, CT, true>), dim3((unsigned)items, gy), dim3(256), 0, st, in, out, tab, outer, N, J,
                inner, n_jt);
      return;
    }
  }
  SC_LAUNCH((k_mdft_axis<JT, CT, false>), dim3((unsigned)((items + 3) / 4), gy), dim3(256), 0, st, in, out, tab,
            outer, N, J, inner, n_jt);
}

// 128-point first-axis lines over the plane results (sc_kernels_plane.h)
static bool ax128_ok(const sc_plan* p, int d) {
  return p->pl128 && d < p->nd - 2 && p->n[d] == SC_PL_N && p->k[d] <= SC_PL_KMAX;
}
static bool ax64_ok(const sc_plan* p, int d) {
  return p->pl64 && d < p->nd - 2 && p->n[d] == SC_P64_N && p->k[d] <= SC_P64_KMAX;
}
// sh.rows > 0: the kept-row side (forward: the result, inverse: the operand) is a sharded spectrum (first axis only)
static int run_ax128(const sc_plan* p, int dir, const cf32* in, cf32* out, int64_t outer, int K, int64_t inner,
                     sc_stream_t st, F3Shard sh = F3Shard{0, 0}) {
  const dim3 grid((unsigned)((inner + 31) / 32), (unsigned)outer);
  if (p->pl64) {
    if (dir < 0) SC_LAUNCH((k_ax64<-1>), grid, dim3(256), 0, st, in, out, (const cf32*)p->pl_tab128, inner, K, sh);
    else SC_LAUNCH((k_ax64<+1>), grid, dim3(256), 0, st, in, out, (const cf32*)p->pl_tab128, inner, K, sh);
    return sc_check_launch("k_ax64");
  }
  if (dir < 0) SC_LAUNCH((k_ax128<-1>), grid, dim3(256), 0, st, in, out, (const cf32*)p->pl_tab128, inner, K, sh);
  else SC_LAUNCH((k_ax128<+1>), grid, dim3(256), 0, st, in, out, (const cf32*)p->pl_tab128, inner, K, sh);
  return sc_check_launch("k_ax128");
}
// plans whose FIRST-axis pass is k_ax128 / k_ax64 address a sharded spectrum natively there (round 5)
static bool ax_native_shards(const sc_plan* p, int64_t n_images) {
  return !p->fast && !p->cplx && p->nd >= 3 && (ax128_ok(p, 0) || ax64_ok(p, 0)) && n_images <= 65535;
}

static int run_axis_mdft(const float* tab, const cf32* in, cf32* out, int64_t outer, int N, int J, int64_t inner,
                         sc_stream_t st) {
  const int n_jt = (J + 15) / 16;
  // column tiles per wave: 4 when that still leaves >= 4 waves per SIMD's worth of waves (4096), fewer for
  // small passes -- the 128^3 first-axis pass (139 k columns) ran 1 wave/SIMD with every table and data
  // load latency exposed: 180 us for 0.18 GB
  const int64_t cols = outer * inner;
  const int ct_max = cols >= (int64_t)4096 * 128 ? 4 : (cols >= (int64_t)4096 * 64 ? 2 : 1);
  if (mdft_switches().tile4) {
    if (n_jt <= 2) launch_mdft_axis<2, 2>(in, out, tab, outer, N, J, inner, n_jt, st);
    else launch_mdft_axis<4, 1>(in, out, tab, outer, N, J, inner, n_jt, st);
  } else if (ct_max == 1) {
    if (n_jt <= 2) launch_mdft_axis<2, 1>(in, out, tab, outer, N, J, inner, n_jt, st);
    else launch_mdft_axis<4, 1>(in, out, tab, outer, N, J, inner, n_jt, st);
  } else if (ct_max == 2) {
    if (n_jt <= 2) launch_mdft_axis<2, 2>(in, out, tab, outer, N, J, inner, n_jt, st);
    else launch_mdft_axis<4, 2>(in, out, tab, outer, N, J, inner, n_jt, st);
  } else {
    if (n_jt <= 2) launch_mdft_axis<2, 4>(in, out, tab, outer, N, J, inner, n_jt, st);
    else if (n_jt <= 4) launch_mdft_axis<4, 2>(in, out, tab, outer, N, J, inner, n_jt, st);
    else launch_mdft_axis<8, 1>(in, out, tab, outer, N, J, inner, n_jt, st);
  }
  return sc_check_launch("k_mdft_axis");
}

static int run_axis(int jt, const cf32* in, cf32* out, const DeviceTable& t, int64_t outer, int N, int J,
                    int64_t inner, sc_stream_t st) {
  switch (jt) {
    case 4: launch_axis<4>(in, out, t, outer, N, J, inner, st); break;
    case 8: launch_axis<8>(in, out, t, outer, N, J, inner, st); break;
    case 16: launch_axis<16>(in, out, t, outer, N, J, inner, st); break;
    default: return sc_fail("sc_engine: bad axis tile");
  }
  return sc_check_launch("k_axis_pass");
}

// ------------------------------------------------------------------------------------------
// transforms
// ------------------------------------------------------------------------------------------
static int transform_forward_impl(const sc_plan* p, int mode, const float* x, float* xhat, int64_t n_images,
                                  void* workspace, void* stream, F3Shard sh);

extern "C" int sc_transform_forward(const sc_plan* p, int mode, const float* x, float* xhat,
                                    int64_t n_images, void* workspace, void* stream) {
  return transform_forward_impl(p, mode, x, xhat, n_images, workspace, stream, F3Shard{0, 0});
}

// sh.rows > 0 (sharded spectrum): only plans whose kernels address it natively arrive here with it (native_shards)
static int transform_forward_impl(const sc_plan* p, int mode, const float* x, float* xhat, int64_t n_images,
                                  void* workspace, void* stream, F3Shard sh) {
  SC_CHECK_ARG(p, "null argument");
  SC_CHECK_ARG(mode == SC_FWD_SCALED || mode == SC_FWD_ADJ_C2R, "bad forward mode");
  if (n_images <= 0) return 0;
  SC_CHECK_ARG(x && xhat, "null argument");
  sc_stream_t st = (sc_stream_t)stream;
  if (p->fast) {
    if (p->d.flags & SC_PLAN_FFT_GEN2)
      return fft2d_forward(&p->fft2d, mode, x, (cf32*)xhat, n_images, workspace, st, &g_last_error);
    if (p->d.flags & SC_PLAN_IO_BF16) {
      if (p->fft2d.tabF && !((uintptr_t)x & 15))           // round 5: the row pass on the matrix cores (16-byte row loads)
        return fft3mx_forward(&p->fft2d, mode, (const sc_bf16*)x, (cf32*)xhat, n_images, st, &g_last_error, sh);
      return fft3_forward(&p->fft2d, mode, (const sc_bf16*)x, (cf32*)xhat, n_images, st, &g_last_error, sh);
    }
    return fft3_forward(&p->fft2d, mode, x, (cf32*)xhat, n_images, st, &g_last_error, sh);
  }
  if (p->f2p) return f2p_forward(p, mode, x, (cf32*)xhat, n_images, workspace, st);
  const int L = p->nd - 1;
  int64_t lines = n_images;
  for (int d = 0; d < L; ++d) lines *= p->n[d];
  // last dim first: real -> complex, or (complex data) one more axis pass with inner = 1
  auto last_pass = [&](cf32* dst) {
    if (p->cplx)
      return run_axis(p->cx_fwd_jt, (const cf32*)x, dst, p->cx_fwd[mode], lines, (int)p->n[L], (int)p->k[L], 1, st);
    return run_r2c(p, mode, x, dst, lines, st);
  };
  if (p->nd == 1) return last_pass((cf32*)xhat);
  if (p->nd == 2 && plane_fwd_ok(p, mode, x)) return run_plane_fwd(p, mode, x, (cf32*)xhat, lines, st);
  SC_CHECK_ARG(workspace, "workspace required");
  int64_t s1, s2;
  generic_ws_sizes(p, n_images, &s1, &s2);
  cf32* bufA = (cf32*)workspace;
  cf32* bufB = bufA + s1;
  int rc;
  cf32* cur = bufA;
  int64_t inner = p->k[L];
  int d_first = L - 1;
  if (plane_fwd_ok(p, mode, x)) {            // last two axes in one launch; its result takes the place of
    cur = bufB;                              // the second-to-last axis pass' (bufB)
    rc = run_plane_fwd(p, mode, x, cur, lines, st);
    inner *= p->k[L - 1];
    d_first = L - 2;
  } else {
    rc = last_pass(bufA);
  }
  if (rc) return rc;
  for (int d = d_first; d >= 0; --d) {
    int64_t outer = n_images;
    for (int e = 0; e < d; ++e) outer *= p->n[e];
    cf32* dst = (d == 0) ? (cf32*)xhat : (cur == bufA ? bufB : bufA);
    if ((ax128_ok(p, d) || ax64_ok(p, d)) && outer <= 65535)
      rc = run_ax128(p, -1, cur, dst, outer, (int)p->k[d], inner, st, d == 0 ? sh : F3Shard{0, 0});
    else if (p->mdft && p->m_ax_fwd[d])
      rc = run_axis_mdft(p->m_ax_fwd[d], cur, dst, outer, (int)p->n[d], (int)p->k[d], inner, st);
    else
      rc = run_axis(p->ax_fwd_jt[d], cur, dst, p->ax_fwd[d], outer, (int)p->n[d], (int)p->k[d], inner, st);
    if (rc) return rc;
    cur = dst;
    inner *= p->k[d];
  }
  return 0;
}

static int run_epilogue_pass(const sc_plan* p, const sc_epilogue* ep, float* y, int64_t n_images, sc_stream_t st) {
  const int64_t n = n_images * p->ntot;
  int64_t blocks = (n + SC_BLOCK - 1) / SC_BLOCK;
  if (blocks > 16384) blocks = 16384;
  SC_LAUNCH(k_epilogue, dim3((unsigned)blocks), dim3(SC_BLOCK), 0, st, y, ep->skip, ep->preact, (int)ep->act, n,
            blocks * SC_BLOCK);
  return sc_check_launch("k_epilogue");
}

extern "C" int sc_transform_inverse(const sc_plan* p, int mode, const float* yhat, const float* bias,
                                    int64_t channels, float* y, int64_t n_images, void* workspace,
                                    void* stream) {
  return sc_transform_inverse_ex(p, mode, yhat, bias, channels, nullptr, y, n_images, workspace, stream);
}

static int transform_inverse_impl(const sc_plan* p, int mode, const float* yhat, const float* bias,
                                  int64_t channels, const sc_epilogue* ep, float* y, int64_t n_images,
                                  void* workspace, void* stream, F3Shard sh);

extern "C" int sc_transform_inverse_ex(const sc_plan* p, int mode, const float* yhat, const float* bias,
                                       int64_t channels, const sc_epilogue* ep, float* y, int64_t n_images,
                                       void* workspace, void* stream) {
  return transform_inverse_impl(p, mode, yhat, bias, channels, ep, y, n_images, workspace, stream, F3Shard{0, 0});
}

// sh.rows > 0 (sharded spectrum): only plans whose kernels address it natively arrive here with it (native_shards)
static int transform_inverse_impl(const sc_plan* p, int mode, const float* yhat, const float* bias,
                                  int64_t channels, const sc_epilogue* ep, float* y, int64_t n_images,
                                  void* workspace, void* stream, F3Shard sh) {
  SC_CHECK_ARG(p, "null argument");
  if (ep && !ep->skip) ep = nullptr;
  if (ep) {
    SC_CHECK_ARG(ep->act == SC_ACT_NONE || ep->act == SC_ACT_GELU, "unknown activation");
    SC_CHECK_ARG(!p->cplx, "complex-data plans take no epilogue");
    SC_CHECK_ARG(ep->act == SC_ACT_GELU || ep->preact == nullptr,
                 "preact is the input of the activation: it is written only with SC_ACT_GELU (pass NULL with SC_ACT_NONE)");
    SC_CHECK_ARG(mode == SC_INV_PADDED || ep->act == SC_ACT_NONE,
                 "an activation in the epilogue belongs to the forward inverse transform (SC_INV_PADDED)");
  }
  const int epi = ep ? (ep->act == SC_ACT_GELU ? 2 : 1) : 0;
  SC_CHECK_ARG(mode == SC_INV_PADDED || mode == SC_INV_ADJ_R2C, "bad inverse mode");
  if (n_images <= 0) return 0;
  SC_CHECK_ARG(yhat && y, "null argument");
  if (channels <= 0) channels = 1;
  sc_stream_t st = (sc_stream_t)stream;
  if (p->fast) {
    if (p->d.flags & SC_PLAN_FFT_GEN2) {
      int rc2 = fft2d_inverse(&p->fft2d, mode, (const cf32*)yhat, bias, channels, y, n_images, workspace, st,
                              &g_last_error);
      return (rc2 || !ep) ? rc2 : run_epilogue_pass(p, ep, y, n_images, st);
    }
    if ((p->d.flags & SC_PLAN_IO_BF16) && p->fft2d.tabG && epi == 0 && !((uintptr_t)y & 7))   // 8-byte row stores
      return fft3mxi_inverse(&p->fft2d, mode, (const cf32*)yhat, bias, channels, (sc_bf16*)y, n_images, st,
                             &g_last_error, sh);
    if (p->d.flags & SC_PLAN_IO_BF16)
      return fft3_inverse(&p->fft2d, mode, (const cf32*)yhat, bias, channels, (sc_bf16*)y, n_images, st,
                          &g_last_error, epi, ep ? (const sc_bf16*)ep->skip : nullptr,
                          ep ? (sc_bf16*)ep->preact : nullptr, sh);
    return fft3_inverse(&p->fft2d, mode, (const cf32*)yhat, bias, channels, y, n_images, st, &g_last_error, epi,
                        ep ? ep->skip : nullptr, ep ? ep->preact : nullptr, sh);
  }
  if (ep && epi == 1 && f2p_fused_add(p, n_images))   // 1024-point rows: the addend rides in the row pass's stores
    return f2p_inverse(p, mode, (const cf32*)yhat, bias, channels, y, n_images, workspace, st, ep->skip);
  if (ep) {                                  // size-agnostic passes: the plain transform, then one streaming pass
    int rc2 = sc_transform_inverse_ex(p, mode, yhat, bias, channels, nullptr, y, n_images, workspace, stream);
    return rc2 ? rc2 : run_epilogue_pass(p, ep, y, n_images, st);
  }
  if (p->f2p) return f2p_inverse(p, mode, (const cf32*)yhat, bias, channels, y, n_images, workspace, st);
  const int L = p->nd - 1;
  int64_t lpi = 1;
  for (int d = 0; d < L; ++d) lpi *= p->n[d];
  const int64_t lines = n_images * lpi;
  SC_CHECK_ARG(!(p->cplx && bias), "complex-data plans take no bias (the host adds it)");
  auto last_pass = [&](const cf32* src) {
    if (p->cplx)
      return run_axis(p->cx_inv_jt, src, (cf32*)y, p->cx_inv[mode], lines, (int)p->k[L], (int)p->n[L], 1, st);
    return run_c2r(p, mode, src, y, bias, lines, lpi, channels, st);
  };
  if (p->nd == 1) return last_pass((const cf32*)yhat);
  if (p->nd == 2 && plane_inv_ok(p, mode, y))
    return run_plane_inv(p, mode, (const cf32*)yhat, y, bias, lines, lpi, channels, st);
  SC_CHECK_ARG(workspace, "workspace required");
  int64_t s1, s2;
  generic_ws_sizes(p, n_images, &s1, &s2);
  cf32* bufA = (cf32*)workspace;  // s1: the last (largest) intermediate lives here
  cf32* bufB = bufA + s1;
  // choose buffers so that the final intermediate lands in bufA
  const cf32* cur = (const cf32*)yhat;
  int64_t outer = n_images;
  const bool plane = plane_inv_ok(p, mode, y);
  const int d_end = plane ? L - 1 : L;
  for (int d = 0; d < d_end; ++d) {
    int64_t inner = 1;
    for (int e = d + 1; e <= L; ++e) inner *= p->k[e];
    const int remaining = L - 1 - d;  // passes after this one
    cf32* dst = (remaining % 2 == 0) ? bufA : bufB;
    int rc;
    if ((ax128_ok(p, d) || ax64_ok(p, d)) && outer <= 65535)
      rc = run_ax128(p, +1, cur, dst, outer, (int)p->k[d], inner, st, d == 0 ? sh : F3Shard{0, 0});
    else if (p->mdft && p->m_ax_inv[d])
      rc = run_axis_mdft(p->m_ax_inv[d], cur, dst, outer, (int)p->k[d], (int)p->n[d], inner, st);
    else
      rc = run_axis(p->ax_inv_jt[d], cur, dst, p->ax_inv[d], outer, (int)p->k[d], (int)p->n[d], inner, st);
    if (rc) return rc;
    cur = dst;
    outer *= p->n[d];
  }
  if (plane) return run_plane_inv(p, mode, cur, y, bias, lines, lpi, channels, st);
  return last_pass(cur);
}

// ------------------------------------------------------------------------------------------
// mode-batched GEMM
// ------------------------------------------------------------------------------------------
template <int PT, int QT, bool CA, bool CB>
static void launch_modegemm(const ModeGemmArgs& g0, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  ModeGemmArgs g = g0;
  g.n_mt = (int)((g.M + SC_WAVE - 1) / SC_WAVE);
  g.n_pg = (int)((g.P + 4 * PT - 1) / (4 * PT));
  g.n_qt = (int)((g.Q + QT - 1) / QT);
  // (the four waves of a workgroup over four neighbouring mode tiles instead of p groups when P <= PT -- 2 KB
  // contiguous per operand row for the weight-streaming launches at B = 4 -- changed nothing:
  // profiles/r02_valu_contraction_wave_modes_ab.txt)
  const int64_t total = (int64_t)g.n_mt * g.n_pg * g.n_qt;
  g.per_xcd = (int)((total + 7) / 8);
  dim3 grid((unsigned)(8 * g.per_xcd));
  SC_LAUNCH((k_modegemm<PT, QT, CA, CB>), grid, dim3(SC_BLOCK), 0, st, g, A, B, C);
}

template <int PT, int QT>
static int dispatch_modegemm_conj(const ModeGemmArgs& g, int ca, int cb, const cf32* A, const cf32* B, cf32* C,
                                  sc_stream_t st) {
  if (!ca && !cb) launch_modegemm<PT, QT, false, false>(g, A, B, C, st);
  else if (ca && !cb) launch_modegemm<PT, QT, true, false>(g, A, B, C, st);
  else if (!ca && cb) launch_modegemm<PT, QT, false, true>(g, A, B, C, st);
  else launch_modegemm<PT, QT, true, true>(g, A, B, C, st);
  return sc_check_launch("k_modegemm");
}

// ---- small-extent streaming path (sc_kernels_sb.h): a batch of <= SB_MAX rows against a large weight, or a
//      reduction of <= SB_MAX terms into a weight-sized result (BASELINE configs[4], B = 4)
static int sb_max_extent() {
  // default 4: the regime where both older kernels are known to be slow.  SC_SB_MAX=n (environment,
  // read once) moves the bound for A-B runs: 0 switches the path off, 8 also takes FNO3d's B = 8 launches
  static const int v = [] {
    const char* e = SC_DIAG_ENV("SC_SB_MAX");
    const int n = e ? std::atoi(e) : 4;
    return n < 0 ? 0 : (n > 8 ? 8 : n);
  }();
  return v;
}
// work-item order of the small-batch kernels (sc_kernels_sb.h, SbGemmArgs::mt_fastest).  Defaults (round 5, measured at
// configs[4], profiles/r05_sb_order_ab.txt): k_modegemm_sb mode tiles slowest, the one-pass pair k_modegemm_sb_bwd mode
// tiles fastest.  SC_GEMM_SB_ALT_ORDER on a descriptor and SC_SB_ALT_ORDER (environment, read once: bit 0 = single
// launches, bit 1 = the pair) each flip it.
static int sb_alt_order_env() {
  static const int v = [] { const char* e = SC_DIAG_ENV("SC_SB_ALT_ORDER"); return e ? std::atoi(e) : 0; }();
  return v;
}
static bool sb_gemm_eligible(const sc_modegemm_desc* d, const void* A, const void* B, const void* C) {
  if (d->flags & (SC_GEMM_F16 | SC_GEMM_NO_SB)) return false;
  if (d->accumulate || d->b_idx || d->c_idx || d->a_sg || d->b_sg || d->c_sg) return false;
  if (d->a_sm != 1 || d->b_sm != 1 || d->c_sm != 1) return false;
  if ((d->n_modes & 1) || d->n_modes < 2) return false;
  if ((d->a_sp | d->a_sr | d->b_sr | d->b_sq | d->c_sp | d->c_sq) & 1) return false;   // 16-byte aligned rows
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return false;
  const int64_t small = d->P < d->R ? d->P : d->R;
  return small <= sb_max_extent();
}

template <int PT, int QT, int ST, int WM, int WP, int WQ>
static int run_sb_gemm_t(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  SbGemmArgs g;
  g.P = d->P; g.Q = d->Q; g.R = d->R; g.M = d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.c_sp = d->c_sp; g.c_sq = d->c_sq;
  g.n_mt = (int)((d->n_modes + 128 * WM - 1) / (128 * WM));
  g.n_pt = (int)((d->P + PT - 1) / PT);
  g.n_qt = (int)((d->Q + QT - 1) / QT);
  const int64_t total = (int64_t)g.n_mt * ((g.n_pt + WP - 1) / WP) * ((g.n_qt + WQ - 1) / WQ);
  if (total >= ((int64_t)1 << 30)) return -1;
  g.per_xcd = (int)((total + 7) / 8);
  g.mt_fastest = (0 ^ (sb_alt_order_env() & 1) ^ ((d->flags & SC_GEMM_SB_ALT_ORDER) ? 1 : 0)) & 1;
  // an operand that exactly one tile reads crosses the chip once: keep it out of the caches the shared one lives in
  g.nt_a = g.n_qt == 1;
  g.nt_b = g.n_pt == 1;
  static const bool plain_c = SC_DIAG_ENV("SC_SB_PLAIN_C") != nullptr;               // A-B
  g.nt_c = (d->flags & SC_GEMM_STREAM_C) && !plain_c ? 1 : 0;
  const dim3 grid((unsigned)(8 * g.per_xcd));
#define SC_SB_LAUNCH(CA, CB) SC_LAUNCH((k_modegemm_sb<PT, QT, ST, WM, WP, WQ, CA, CB>), grid, dim3(SC_BLOCK), 0, st, g, A, B, C)
  if (!d->conj_a && !d->conj_b) SC_SB_LAUNCH(false, false);
  else if (d->conj_a && !d->conj_b) SC_SB_LAUNCH(true, false);
  else if (!d->conj_a && d->conj_b) SC_SB_LAUNCH(false, true);
  else SC_SB_LAUNCH(true, true);
#undef SC_SB_LAUNCH
  return sc_check_launch("k_modegemm_sb");
}

static int run_sb_gemm(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  // small batch: the register tile holds every row (4 x 4, or 8 x 2 for 5..8 rows), three reduction steps in
  // flight; short reduction (weight gradient): 4 x 4 outputs per lane, two steps in flight.  Wave arrangement
  // (sc_kernels_sb.h): the four waves over four column tiles (2 x 2 tiles for the weight gradient) of one 128-mode
  // tile; SC_GEMM_SB_WM4 / SC_SB_WM=4 (flag / environment, A-B) = four neighbouring 128-mode tiles of one tile instead
  static const bool wm4_env = [] { const char* e = SC_DIAG_ENV("SC_SB_WM"); return e && std::atoi(e) == 4; }();
  const bool wm4 = wm4_env || (d->flags & SC_GEMM_SB_WM4);
  if (d->P <= 4)
    return wm4 ? run_sb_gemm_t<4, 4, 3, 4, 1, 1>(d, A, B, C, st) : run_sb_gemm_t<4, 4, 3, 1, 1, 4>(d, A, B, C, st);
  if (d->P <= 8 && d->P <= d->R)
    return wm4 ? run_sb_gemm_t<8, 2, 3, 4, 1, 1>(d, A, B, C, st) : run_sb_gemm_t<8, 2, 3, 1, 1, 4>(d, A, B, C, st);
  return wm4 ? run_sb_gemm_t<4, 4, 2, 4, 1, 1>(d, A, B, C, st) : run_sb_gemm_t<4, 4, 2, 1, 2, 2>(d, A, B, C, st);
}

// ---- the two contractions of a small-batch backward pass in one pass over the weight (sc_kernels_sb.h,
//      k_modegemm_sb_bwd): d0 = weight gradient (conj A: xhat^H ghat), d1 = gradient of the spectrum (conj B: ghat W^H),
//      both operands named ghat the SAME array.  Returns -1 when the pair does not qualify.
template <int BT>
static int run_sb_bwd_t(const SbBwdArgs& g, const cf32* xhat, const cf32* ghat, const cf32* W, cf32* gW, cf32* gxhat,
                        sc_stream_t st) {
  SC_LAUNCH((k_modegemm_sb_bwd<BT, 4, 2>), dim3((unsigned)(8 * g.per_xcd)), dim3(SC_BLOCK), 0, st, g, xhat, ghat, W, gW,
            gxhat);
  return sc_check_launch("k_modegemm_sb_bwd");
}

static bool sb_bwd_eligible(const sc_modegemm_desc* d0, const void* A0, const void* B0, const void* C0,
                            const sc_modegemm_desc* d1, const void* A1, const void* B1, const void* C1) {
  static const bool off = SC_DIAG_ENV("SC_SB_NO_PAIR") != nullptr;                     // A-B
  if (off) return false;
  if (!sb_gemm_eligible(d0, A0, B0, C0) || !sb_gemm_eligible(d1, A1, B1, C1)) return false;
  if (!(d0->conj_a && !d0->conj_b && !d1->conj_a && d1->conj_b)) return false;
  if (B0 != A1 || d0->b_sr != d1->a_sp || d0->b_sq != d1->a_sr) return false;          // one ghat[b, o, m]
  if (d0->n_modes != d1->n_modes || d0->R != d1->P || d0->P != d1->Q || d0->Q != d1->R) return false;
  if (d0->R < 1 || d0->R > 4 || d0->R > sb_max_extent()) return false;                // the batch lives in registers
  // the weight-sized arrays must dominate: otherwise the separate launches (more, smaller work items) fill the chip better
  return d0->P * d0->Q >= 64 * d0->R;
}

static int run_sb_bwd(const sc_modegemm_desc* d0, const cf32* A0, const cf32* B0, cf32* C0,
                      const sc_modegemm_desc* d1, const cf32* A1, const cf32* B1, cf32* C1, sc_stream_t st) {
  if (!sb_bwd_eligible(d0, A0, B0, C0, d1, A1, B1, C1)) return -1;
  SbBwdArgs g;
  g.B = d0->R; g.Ci = d0->P; g.Co = d0->Q; g.M = d0->n_modes;
  g.x_si = d0->a_sp; g.x_sb = d0->a_sr;
  g.g_sb = d0->b_sr; g.g_so = d0->b_sq;
  g.gw_si = d0->c_sp; g.gw_so = d0->c_sq;
  g.w_so = d1->b_sr; g.w_si = d1->b_sq;
  g.gx_sb = d1->c_sp; g.gx_si = d1->c_sq;
  g.n_mt = (int)((g.M + 127) / 128);
  g.n_itg = (int)(((g.Ci + 3) / 4 + 3) / 4);
  const int64_t total = (int64_t)g.n_mt * g.n_itg;
  if (total >= ((int64_t)1 << 30)) return -1;
  g.per_xcd = (int)((total + 7) / 8);
  g.mt_fastest = (1 ^ ((sb_alt_order_env() >> 1) & 1) ^ (((d0->flags | d1->flags) & SC_GEMM_SB_ALT_ORDER) ? 1 : 0)) & 1;
  static const bool plain_c = SC_DIAG_ENV("SC_SB_PLAIN_C") != nullptr;                 // A-B
  g.nt_gw = (d0->flags & SC_GEMM_STREAM_C) && !plain_c ? 1 : 0;
  switch (g.B) {
    case 1: return run_sb_bwd_t<1>(g, A0, B0, B1, C0, C1, st);
    case 2: return run_sb_bwd_t<2>(g, A0, B0, B1, C0, C1, st);
    case 3: return run_sb_bwd_t<3>(g, A0, B0, B1, C0, C1, st);
    default: return run_sb_bwd_t<4>(g, A0, B0, B1, C0, C1, st);
  }
}

// ---- mode-independent right operand (sc_kernels_sb.h, k_modegemm_bfac): factor matrices through the scalar cache
static bool bfac_gemm_eligible(const sc_modegemm_desc* d) {
  if (d->flags & (SC_GEMM_F16 | SC_GEMM_NO_SB)) return false;
  if (d->accumulate || d->b_idx || d->c_idx || d->a_sg || d->b_sg || d->c_sg) return false;
  if (d->b_sm != 0 || d->a_sm != 1 || d->c_sm != 1) return false;
  return d->Q >= 8 && d->R >= 4 && d->n_modes >= 64 && d->R * ((d->Q + 1) & ~(int64_t)1) <= 8192;   // B in <= 64 KiB of LDS
}

template <int QC>
static int run_bfac_gemm_t(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  BfacGemmArgs g;
  g.P = d->P; g.Q = d->Q; g.R = d->R; g.M = d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.c_sp = d->c_sp; g.c_sq = d->c_sq;
  g.n_mt = (int)((d->n_modes + 63) / 64);
  g.n_qg = (int)((d->Q + 4 * QC - 1) / (4 * QC));
  const int64_t total = (int64_t)g.n_mt * g.n_qg * d->P;
  if (total >= ((int64_t)1 << 31)) return -1;
  const dim3 grid((unsigned)total);
  const size_t shmem = (size_t)(d->R * ((d->Q + 1) & ~(int64_t)1)) * sizeof(cf32);
#define SC_BF_LAUNCH(CA, CB) SC_LAUNCH((k_modegemm_bfac<QC, CA, CB>), grid, dim3(SC_BLOCK), shmem, st, g, A, B, C)
  if (!d->conj_a && !d->conj_b) SC_BF_LAUNCH(false, false);
  else if (d->conj_a && !d->conj_b) SC_BF_LAUNCH(true, false);
  else if (!d->conj_a && d->conj_b) SC_BF_LAUNCH(false, true);
  else SC_BF_LAUNCH(true, true);
#undef SC_BF_LAUNCH
  return sc_check_launch("k_modegemm_bfac");
}

static int run_bfac_gemm(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  // columns per wave: 9 when that divides the work into whole waves better (ranks such as 36, 18, 27), else 8
  const int64_t w8 = (d->Q + 7) / 8, w9 = (d->Q + 8) / 9;
  return (w9 * 9 - d->Q < w8 * 8 - d->Q) ? run_bfac_gemm_t<9>(d, A, B, C, st) : run_bfac_gemm_t<8>(d, A, B, C, st);
}

#ifndef SC_EMU
#define SC_FMX_ATTR(kern, lds)                                                                                    \
  if ((lds) > 64 * 1024)                                                                                         \
  SC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)))
#else
#define SC_FMX_ATTR(kern, lds) (void)0
#endif
static int tucker_abl() {
  static const int v = [] { const char* e = SC_DIAG_ENV("SC_TK_ABL"); return e ? std::atoi(e) : 0; }();
  return v;
}
// ---- factor-matrix products and mode-summed contractions on the matrix cores (sc_kernels_fmx.h) ---------------
static bool fmx_off() {
  static const bool off = SC_DIAG_ENV("SC_FMX_OFF") != nullptr;                        // A-B against the VALU kernels
  return off;
}
// workgroups for n chunks with `cap` co-resident: every workgroup the same number of rounds
static int fmx_wgs(int64_t chunks, int64_t cap) {
  const int64_t rounds = (chunks + cap - 1) / cap;
  int64_t wgs = (chunks + rounds - 1) / rounds;
  // Session 2: a launch is as slow as its busiest compute unit, so the workgroup count is rounded DOWN to a multiple
  // of the unit count when that costs the busiest workgroup at most one more chunk: TFNO rank 0.1 has 1056 chunks
  // (32 rows x 33 mode blocks): 528 workgroups of 2 chunks put three workgroups = 6 chunks on 16 units (average 4.1),
  // 512 workgroups (32 of them with 3 chunks) put 5 on the busiest.  SC_FMX_WGS_EXACT=1 (environment, A-B): the old rule
  static const bool exact = SC_DIAG_ENV("SC_FMX_WGS_EXACT") != nullptr;
  const int64_t cus = sc_cu_count();
  const int64_t m = (wgs / cus) * cus;
  if (!exact && m >= cus && m < wgs && (chunks + m - 1) / m <= rounds + 1) wgs = m;
  return (int)wgs;
}
static bool fmx_bfac_eligible(const sc_modegemm_desc* d) {
  if (fmx_off() || (d->flags & (SC_GEMM_F16 | SC_GEMM_NO_FMX | SC_GEMM_FORCE_VALU))) return false;
  if (d->accumulate || d->b_idx || d->c_idx || d->a_sg || d->b_sg || d->c_sg) return false;
  if (d->b_sm != 0 || d->a_sm != 1 || d->c_sm != 1) return false;
  if (d->Q < 8 || d->Q > 64 || d->R < 4 || d->R > 64 || d->n_modes < 64) return false;
  return d->P * ((d->n_modes + 63) / 64) < ((int64_t)1 << 30);
}
template <int PF, int TQ>
static int run_fmx_bfac_t(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  FmxArgs g;
  g.P = d->P; g.Q = d->Q; g.R = d->R; g.M = d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.c_sp = d->c_sp; g.c_sq = d->c_sq;
  g.n_mb = (int)((d->n_modes + 63) / 64);
  g.n_chunks = (int)(d->P * g.n_mb);
  const int q4 = (int)((d->Q + 3) & ~(int64_t)3), r4 = (int)((d->R + 3) & ~(int64_t)3);
  g.ldb = tkm_ld_rows(r4);
  g.abl = tucker_abl();
  g.inv_q = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)d->Q - 1) / (uint64_t)d->Q);
  const size_t lds = (size_t)(q4 * g.ldb + r4 * SC_FMX_LDK) * sizeof(cf32);
  g.n_wg = fmx_wgs(g.n_chunks, sc_cu_count() * (int64_t)(160 * 1024 / lds > 4 ? 4 : 160 * 1024 / lds));
#define SC_FX_LAUNCH(CA, CB)                                                                                     \
  do {                                                                                                           \
    auto kern = k_modegemm_bfac_mx<PF, TQ, CA, CB>;                                                                  \
    SC_FMX_ATTR(kern, lds);                                                                                      \
    SC_LAUNCH(kern, dim3((unsigned)g.n_wg), dim3(256), lds, st, g, A, B, C);                                     \
  } while (0)
  if (!d->conj_a && !d->conj_b) SC_FX_LAUNCH(false, false);
  else if (d->conj_a && !d->conj_b) SC_FX_LAUNCH(true, false);
  else if (!d->conj_a && d->conj_b) SC_FX_LAUNCH(false, true);
  else SC_FX_LAUNCH(true, true);
#undef SC_FX_LAUNCH
  return sc_check_launch("k_modegemm_bfac_mx");
}
static int run_fmx_bfac(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  if (d->R <= 36) return d->Q <= 48 ? run_fmx_bfac_t<9, 3>(d, A, B, C, st) : run_fmx_bfac_t<9, 4>(d, A, B, C, st);
  return d->Q <= 48 ? run_fmx_bfac_t<16, 3>(d, A, B, C, st) : run_fmx_bfac_t<16, 4>(d, A, B, C, st);
}

static bool fmx_msum_eligible(const sc_modegemm_desc* d) {
  if (fmx_off() || (d->flags & (SC_GEMM_F16 | SC_GEMM_NO_FMX | SC_GEMM_FORCE_VALU))) return false;
  if (d->b_idx || d->c_idx || d->a_sg || d->b_sg || d->c_sg) return false;
  if (d->a_sm != 1 || d->b_sm != 1) return false;
  if (d->P < 8 || d->P > 64 || d->Q < 8 || d->Q > 64 || d->n_modes < 64 || d->R < 1) return false;
  return d->R * ((d->n_modes + 63) / 64) < ((int64_t)1 << 30);
}
static void fmx_msum_args(const sc_modegemm_desc* d, FmxArgs& g, size_t& lds) {
  g.P = d->P; g.Q = d->Q; g.R = d->R; g.M = d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.c_sp = d->c_sp; g.c_sq = d->c_sq;
  g.n_mb = (int)((d->n_modes + 63) / 64);
  g.n_chunks = (int)(d->R * g.n_mb);
  g.ldb = 0;
  g.inv_q = 0;
  g.abl = tucker_abl();
  const int p4 = (int)((d->P + 3) & ~(int64_t)3), q4 = (int)((d->Q + 3) & ~(int64_t)3);
  lds = (size_t)((p4 + q4) * SC_FMX_LDR) * sizeof(cf32);
  const int64_t per_cu = 160 * 1024 / lds > 3 ? 3 : 160 * 1024 / lds;
  g.n_wg = fmx_wgs(g.n_chunks, sc_cu_count() * per_cu);
}
template <int PFA, int PFB, int SLOTS>
static int run_fmx_msum_t(const sc_modegemm_desc* d, const FmxArgs& g, size_t lds, const cf32* A, const cf32* B,
                          cf32* partial, sc_stream_t st) {
#define SC_FX_LAUNCH(CA, CB)                                                                                     \
  do {                                                                                                           \
    auto kern = k_modegemm_msum_mx<PFA, PFB, SLOTS, CA, CB>;                                                     \
    SC_FMX_ATTR(kern, lds);                                                                                      \
    SC_LAUNCH(kern, dim3((unsigned)g.n_wg), dim3(256), lds, st, g, A, B, partial);                               \
  } while (0)
  if (!d->conj_a && !d->conj_b) SC_FX_LAUNCH(false, false);
  else if (d->conj_a && !d->conj_b) SC_FX_LAUNCH(true, false);
  else if (!d->conj_a && d->conj_b) SC_FX_LAUNCH(false, true);
  else SC_FX_LAUNCH(true, true);
#undef SC_FX_LAUNCH
  return sc_check_launch("k_modegemm_msum_mx");
}

// ---- matrix-core path (sc_kernels_mfma.h): channel counts that fill 32 x 32 MFMA tiles ----------
static bool mfma_gemm_eligible(const sc_modegemm_desc* d) {
  // one workgroup tile is 32 or 64 rows x 64 columns; ragged problems (Tucker / TT ranks such as 36) take it
  // when they fill at least ~half of a tile, smaller ones stay on the lanes-are-modes VALU kernel
  if (d->accumulate) return false;
  if (d->Q < 24 || d->Q > 64) return false;
  if (d->P < 24 || d->P > 64) return false;
  if (d->R < 8) return false;
  // a single 8-deep stage only pays on a (nearly) full tile: P = Q = 32, R = 8 over 17 k modes was 148 us
  // here against 85 us on the VALU kernel
  const int64_t rows = d->P <= 32 ? 32 : 64;
  if (d->R < 16 && 4 * d->P * d->Q < 3 * rows * 64) return false;
  if (d->n_modes >= ((int64_t)1 << 31) / 16) return false;
  return true;
}

// two shapes of the matrix-core kernel:
//   wide   (default): 9 modes per workgroup, 8 waves, one workgroup per CU;
//   paired (P = 32, SC_GEMM_PAIRED, A-B only): 5 modes per workgroup, 4 waves, two workgroups per CU.
//     Measured: same speed with warm caches (53.9 vs 53.8 us), SLOWER from HBM (78.6 vs 63.9 us):
//     40-byte segments cost more DRAM/fabric efficiency than the second workgroup's latency hiding buys.
#define SC_MG_NM_WIDE 9
#define SC_MG_NM_PAIRED 5
//   few    (n_modes <= SC_MG_FEW_MAX, e.g. a 64 x 64 grid keeping 32 x 17): 4 modes per workgroup.  With 9
//     slots a small problem either leaves most CUs idle or, spread over all of them, multiplies stale
//     slots (the MFMAs of a workgroup always cover all its NM slots): 72 us for 53 MB at 544 modes.
#define SC_MG_NM_FEW 4
#ifndef SC_MG_FEW_MAX
#define SC_MG_FEW_MAX 1152
#endif
template <int PT, int NM, int NWV, bool CA, bool CB>
static void launch_mfma_gemm(const MfmaGemmArgs& g, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  SC_LAUNCH((k_modegemm_mfma<PT, 4, NM, CA, CB, NWV>), dim3((unsigned)g.G),
            dim3((MfmaGemmCfg<PT, 4, NM, NWV>::THREADS)), 0, st, g, A, B, C);
}

template <int PT, int NM, int NWV>
static void dispatch_mfma_gemm(const MfmaGemmArgs& g, int ca, int cb, const cf32* A, const cf32* B, cf32* C,
                               sc_stream_t st) {
  if (!ca && !cb) launch_mfma_gemm<PT, NM, NWV, false, false>(g, A, B, C, st);
  else if (ca && !cb) launch_mfma_gemm<PT, NM, NWV, true, false>(g, A, B, C, st);
  else if (!ca && cb) launch_mfma_gemm<PT, NM, NWV, false, true>(g, A, B, C, st);
  else launch_mfma_gemm<PT, NM, NWV, true, true>(g, A, B, C, st);
}

static int run_mfma_gemm(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  MfmaGemmArgs g;
  g.P = (int)d->P; g.Q = (int)d->Q; g.R = (int)d->R; g.M = (int)d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.a_sm = d->a_sm;
  g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.b_sm = d->b_sm;
  g.c_sp = d->c_sp; g.c_sq = d->c_sq; g.c_sm = d->c_sm;
  g.b_idx = d->b_idx; g.c_idx = d->c_idx;
  g.stream_c = (d->flags & SC_GEMM_STREAM_C) ? 1 : 0;
  // contiguous mode ranges of <= NM modes, split evenly over (workgroups per CU) x 256 CUs
  const bool paired = d->P <= 32 && (d->flags & SC_GEMM_PAIRED);
  const int64_t M = d->n_modes;
  const bool few = !paired && M <= SC_MG_FEW_MAX && !(d->flags & SC_GEMM_WIDE);
  const int64_t nmx = paired ? SC_MG_NM_PAIRED : (few ? SC_MG_NM_FEW : SC_MG_NM_WIDE);
  int64_t G = (M + nmx - 1) / nmx;
  const int64_t slots = paired ? 512 : 256;
  if (few) G = G < 8 ? G : (G + 7) / 8 * 8;        // keep the ranges full: ceil(M / 4) workgroups
  else if (G < slots) G = M < slots ? M : slots;
  else G = (G + 7) / 8 * 8;
  if (G > M) G = M;
  const int64_t cap = (d->flags >> 8) & 0xffff;               // SC_GEMM_GRID(n): tests / tuning
  if (cap > 0 && cap < G && cap * nmx >= M) G = cap;
  g.G = (int)G;
  if (paired) dispatch_mfma_gemm<1, SC_MG_NM_PAIRED, 4>(g, d->conj_a, d->conj_b, A, B, C, st);
  else if (few && d->P <= 32) dispatch_mfma_gemm<1, SC_MG_NM_FEW, 8>(g, d->conj_a, d->conj_b, A, B, C, st);
  else if (few) dispatch_mfma_gemm<2, SC_MG_NM_FEW, 8>(g, d->conj_a, d->conj_b, A, B, C, st);
  else if (d->P <= 32) dispatch_mfma_gemm<1, SC_MG_NM_WIDE, 8>(g, d->conj_a, d->conj_b, A, B, C, st);
  else dispatch_mfma_gemm<2, SC_MG_NM_WIDE, 8>(g, d->conj_a, d->conj_b, A, B, C, st);
  return sc_check_launch("k_modegemm_mfma");
}

// ---- streamed matrix-core path (sc_kernels_gemm8.h): plain contiguous-mode operands ------------------------------
// Two shapes of the kernel are built into the library (profiles/r02_gemm_dma_v3_shapes_ab.txt):
//   narrow   8 modes x 32 x 32 tiles, 4 waves, 2 r pairs per stage, plain stage loop: 528 workgroups for the forward
//            / gX contraction of the metric shape (45 us; 16 modes x 8 waves: 54 us)
//   wide    16 modes x 32 x 32 tiles, 8 waves, 128-byte segments, software-pipelined stage: calls with >= 8 tiles per
//            mode group (hidden 128: weight gradient 173 against 224 us, and the store-dominated weight gradient of
//            the 1024^2 config 1.38 against 2.1 ms).  The metric shape's weight gradient (4 tiles) measures 46.6
//            against 51.2 us stand-alone but 63.9 against 53.5 us INSIDE a step (profiles/r02_gpu6_kernel_stats.txt:
//            its operands come from HBM there) and stays on the narrow shape
// Measured (profiles/r02_gemm_dma_diag_grid_layout.txt): ONE workgroup needs ~33 us for its stages whatever the
// operand layout, segment size, ring depth or instruction order -- so a launch is as fast as its busiest CU.
#if defined(SC_G8_SHAPE) && SC_G8_SHAPE == 1       // measurement builds: one r pair per stage, deeper ring
#define SC_G8_NARROW 4, 2, 1, 5, false             // 40 KiB of LDS: 4 workgroups per CU
#define SC_G8_NARROW_RESIDENT 1024
#elif defined(SC_G8_SHAPE) && SC_G8_SHAPE == 2
#define SC_G8_NARROW 4, 2, 1, 4, false             // 32 KiB: 5 per CU
#define SC_G8_NARROW_RESIDENT 1280
#else
#define SC_G8_NARROW 4, 2, 2, 3, false
#define SC_G8_NARROW_RESIDENT 768
#endif
#define SC_G8_WIDE 8, 2, 1, 4, true
// workgroups the chip holds at once: narrow 3 per CU (48 KiB of LDS each), wide 2 per CU (64 KiB)
#define SC_G8_RESIDENT(wide) ((wide) ? 512 : SC_G8_NARROW_RESIDENT)
static bool gemm8_eligible(const sc_modegemm_desc* d, const void* A, const void* B, const void* C) {
  if (d->flags & (SC_GEMM_FORCE_VALU | SC_GEMM_NO_STREAM | SC_GEMM_F16)) return false;
  if (d->accumulate || d->b_idx || d->c_idx) return false;
  if (d->a_sm != 1 || d->b_sm != 1 || d->c_sm != 1) return false;
  if (d->n_modes % 8 != 0 || d->n_modes >= ((int64_t)1 << 31)) return false;
  if ((d->a_sg || d->b_sg || d->c_sg) && d->n_modes % 16 != 0) return false;        // tiled operands: groups of 16
  if ((d->a_sg | d->b_sg | d->c_sg) & 1) return false;
  // 16-byte granules: every row / column of every operand must start on an even complex element
  if ((d->a_sp | d->a_sr | d->b_sr | d->b_sq | d->c_sp | d->c_sq) & 1) return false;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return false;
  // tiles are 32 rows x 32 columns: take problems that fill them to >= 1/2 (SC_G8_FILL4 quarters; round 2: 3/4).  The
  // ragged 36 x 36 per-mode products of the Tucker chain at configs[2] (two or four tiles, 56 % / 32 % filled) take
  // 24 / 25 / 39 us here against 57 / 57 / 65 us on the register-staged k_modegemm_mfma
  // (profiles/r03_tfno_kernel_stats_g8fill2.txt): the operand stream, not the matrix pipe, is what a tile costs
  const int64_t cols = 32;
  const int64_t Pp = (d->P + 31) / 32 * 32, Qp = (d->Q + cols - 1) / cols * cols;
  static const int64_t fill4 = [] { const char* e = SC_DIAG_ENV("SC_G8_FILL4"); return e ? (int64_t)std::atoi(e) : (int64_t)2; }();
  bool rows_ok = 4 * d->P >= fill4 * Pp;
  // a small batch against a weight read ACROSS its rows (the gradient of the spectrum: B[r, q] = W[q, r], q stride >
  // r stride): the lanes-are-modes VALU kernel gathers 512-byte pieces of W there (FNO3d 128^3, B = 8: 115 us), the
  // streamed kernel does not care (55 us) although 32 / P of its matrix work is spent on clamped duplicate rows.  Not
  // for the forward product (VALU 45 us, streamed 53 us) and not below 8 rows (B = 4 at 1024^2 / hidden 128: 1.09 ->
  // 1.50 ms): profiles/r02_gemm_small_batch_ab.txt
  if (d->P >= 8 && d->P <= 32 && d->b_sq > d->b_sr) rows_ok = true;
  if (!rows_ok || 4 * d->Q < fill4 * Qp) return false;
  if (d->R < 4) return false;
  if (Pp / 32 * (Qp / cols) * (d->n_modes / 8) >= ((int64_t)1 << 30)) return false;
  return true;
}

template <int GS, int QT, int SUB, int D, bool IL, bool CA, bool CB>
static void launch_gemm8(const Gemm8Args& g, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  SC_LAUNCH((k_modegemm_dma<GS, QT, SUB, D, IL, CA, CB>), dim3((unsigned)g.G), dim3((Gemm8Cfg<GS, QT, SUB>::THREADS)),
            0, st, g, A, B, C);
}

template <int GS, int QT, int SUB, int D, bool IL>
static void dispatch_gemm8(const Gemm8Args& g, int ca, int cb, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  if (!ca && !cb) launch_gemm8<GS, QT, SUB, D, IL, false, false>(g, A, B, C, st);
  else if (ca && !cb) launch_gemm8<GS, QT, SUB, D, IL, true, false>(g, A, B, C, st);
  else if (!ca && cb) launch_gemm8<GS, QT, SUB, D, IL, false, true>(g, A, B, C, st);
  else launch_gemm8<GS, QT, SUB, D, IL, true, true>(g, A, B, C, st);
}

// launch geometry of one contraction; `resident` = workgroups the launch may count on being co-resident
static bool gemm8_args(const sc_modegemm_desc* d, Gemm8Args& g, int64_t resident_narrow, int64_t resident_wide,
                       int force_shape /* -1: choose, 0: narrow, 1: wide */) {
  const int64_t cols = 32;
  const int64_t tiles = ((d->P + 31) / 32) * ((d->Q + cols - 1) / cols);
  bool wide;
  if (force_shape >= 0) wide = force_shape == 1;
  else {
#if defined(SC_G8_FORCE_NARROW)          // measurement builds only
    wide = false;
#elif defined(SC_G8_FORCE_WIDE)
    wide = d->n_modes % 16 == 0;
#else
    wide = d->n_modes % 16 == 0 && tiles >= 8;
#endif
  }
  const int64_t modes = wide ? 16 : 8;
  const int64_t resident = wide ? resident_wide : resident_narrow;
  g.P = (int)d->P; g.Q = (int)d->Q; g.R = (int)d->R;
  g.n_mg = (int)(d->n_modes / modes);
  g.n_pb = (int)((d->P + 31) / 32);
  g.n_qb = (int)((d->Q + cols - 1) / cols);
  g.a_sp = d->a_sp; g.a_sr = d->a_sr;
  g.b_sr = d->b_sr; g.b_sq = d->b_sq;
  g.c_sp = d->c_sp; g.c_sq = d->c_sq;
  g.a_sg = d->a_sg ? d->a_sg : 16;                            // groups of 16 modes (plain arrays: 16 apart)
  g.b_sg = d->b_sg ? d->b_sg : 16;
  g.c_sg = d->c_sg ? d->c_sg : 16;
  g.stream_c = (d->flags & SC_GEMM_STREAM_C) ? 1 : 0;
  // tiles per workgroup: a launch a little larger than what the chip holds at once runs its tiles back to back
  // inside fewer workgroups instead of queueing a short second round
  const int64_t nblk = (int64_t)g.n_pb * g.n_qb;
  int64_t bpw = 1;
  if (g.n_mg <= resident && g.n_mg * nblk > resident) {
    bpw = nblk;
    for (int64_t b = 1; b <= nblk; ++b)
      if (g.n_mg * ((nblk + b - 1) / b) <= resident) {
        bpw = b;
        break;
      }
  }
  const int64_t cap = (d->flags >> 8) & 0xffff;               // SC_GEMM_GRID(n) doubles as "tiles per workgroup" (tests)
  if (cap > 0 && cap <= nblk) bpw = cap;
  g.bpw = (int)bpw;
  g.G = (int)(g.n_mg * ((nblk + bpw - 1) / bpw));
  return wide;
}

static int run_gemm8(const sc_modegemm_desc* d, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  Gemm8Args g;
  const bool wide = gemm8_args(d, g, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), -1);
  if (wide) dispatch_gemm8<SC_G8_WIDE>(g, d->conj_a, d->conj_b, A, B, C, st);
  else dispatch_gemm8<SC_G8_NARROW>(g, d->conj_a, d->conj_b, A, B, C, st);
  return sc_check_launch("k_modegemm_dma");
}

// The two contractions of a backward pass (and the bias gradient) as ONE launch of k_modegemm_dma_bwd
// (sc_kernels_gemm8.h): d0 = weight gradient (conj A), d1 = gradient of the spectrum (conj B).  Returns -1 when the
// pair does not qualify (the caller then launches them one after the other).
#ifndef SC_G8_PAIR_BPW                   // measurement builds: 0 = as chosen per job, 1 = all tiles of a mode group in
#define SC_G8_PAIR_BPW 0                 // one workgroup (both jobs), 2 = that for the weight gradient only
#endif
static int run_gemm8_bwd(const sc_modegemm_desc* d0, const cf32* A0, const cf32* B0, cf32* C0,
                         const sc_modegemm_desc* d1, const cf32* A1, const cf32* B1, cf32* C1,
                         const Gemm8Bias& bias, sc_stream_t st) {
#ifdef SC_G8_NO_PAIR                     // measurement builds only: the round-1 sequence of launches
  return -1;
#endif
  if (!gemm8_eligible(d0, A0, B0, C0) || !gemm8_eligible(d1, A1, B1, C1)) return -1;
  if (!(d0->conj_a && !d0->conj_b && !d1->conj_a && d1->conj_b)) return -1;
  if (d0->n_modes != d1->n_modes) return -1;
  // the narrow shape (3 workgroups per CU) for both jobs: the wide one wins on a weight gradient alone but loses
  // inside a step, and one kernel has one shape
  Gemm8Args g0, g1;
#ifdef SC_G8_PAIR_NARROW_ONLY            // measurement builds: pair only what would run the narrow shape anyway
  if (gemm8_args(d0, g0, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), -1) ||
      gemm8_args(d1, g1, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), -1)) return -1;
#endif
  gemm8_args(d0, g0, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), 0);
  gemm8_args(d1, g1, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), 0);
#if SC_G8_PAIR_BPW >= 1
  g0.bpw = g0.n_pb * g0.n_qb; g0.G = g0.n_mg;
#endif
#if SC_G8_PAIR_BPW == 1
  g1.bpw = g1.n_pb * g1.n_qb; g1.G = g1.n_mg;
#endif
  if ((g0.G & 7) || (g1.G & 7)) return -1;                    // octets of workgroups alternate between the jobs
  typedef Gemm8Cfg<4, 2, 2> K;
  const int64_t nb = bias.ghat ? (bias.channels + K::NW - 1) / K::NW : 0;
  if ((int64_t)g0.G + g1.G + nb >= ((int64_t)1 << 30)) return -1;
  SC_LAUNCH((k_modegemm_dma_bwd<SC_G8_NARROW>), dim3((unsigned)(g0.G + g1.G + nb)), dim3(K::THREADS), 0, st,
            g0, A0, B0, C0, g1, A1, B1, C1, bias);
  return sc_check_launch("k_modegemm_dma_bwd");
}

extern "C" int sc_modegemm(const sc_modegemm_desc* d, const float* A, const float* B, float* C,
                           void* stream) {
  SC_CHECK_ARG(d && A && B && C, "null argument");
  SC_CHECK_ARG(d->P >= 0 && d->Q >= 0 && d->R >= 0 && d->n_modes >= 0, "negative extent");
  if (d->P == 0 || d->Q == 0 || d->n_modes == 0) return 0;
  SC_CHECK_ARG(d->R > 0, "R must be > 0");
  ModeGemmArgs g;
  g.P = d->P; g.Q = d->Q; g.R = d->R; g.M = d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.a_sm = d->a_sm;
  g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.b_sm = d->b_sm;
  g.c_sp = d->c_sp; g.c_sq = d->c_sq; g.c_sm = d->c_sm;
  g.b_idx = d->b_idx; g.c_idx = d->c_idx;
  g.accumulate = d->accumulate;
  SC_CHECK_ARG(((g.M + 63) / 64) * ((g.P + 15) / 16) * ((g.Q + 3) / 4) < ((int64_t)1 << 30),
               "problem too large for one launch grid");
  sc_stream_t st = (sc_stream_t)stream;
  const cf32* a = (const cf32*)A;
  const cf32* b = (const cf32*)B;
  cf32* c = (cf32*)C;
  if (d->flags & SC_GEMM_F16) {
    SC_CHECK_ARG(!d->accumulate && !(d->a_sg || d->b_sg || d->c_sg), "SC_GEMM_F16: plain C = A B launches only");
    g.n_mt = (int)((g.M + SC_WAVE - 1) / SC_WAVE);
    g.n_pg = (int)((g.P + 15) / 16);
    g.n_qt = (int)((g.Q + 3) / 4);
    g.per_xcd = 0;
    const dim3 grid((unsigned)((int64_t)g.n_mt * g.n_pg * g.n_qt));
    if (d->conj_a && d->conj_b) SC_LAUNCH((k_modegemm_f16<true, true>), grid, dim3(SC_BLOCK), 0, st, g, a, b, c);
    else if (d->conj_a) SC_LAUNCH((k_modegemm_f16<true, false>), grid, dim3(SC_BLOCK), 0, st, g, a, b, c);
    else if (d->conj_b) SC_LAUNCH((k_modegemm_f16<false, true>), grid, dim3(SC_BLOCK), 0, st, g, a, b, c);
    else SC_LAUNCH((k_modegemm_f16<false, false>), grid, dim3(SC_BLOCK), 0, st, g, a, b, c);
    return sc_check_launch("k_modegemm_f16");
  }
  if (fmx_bfac_eligible(d)) return run_fmx_bfac(d, a, b, c, st);
  if (bfac_gemm_eligible(d)) {
    const int rc = run_bfac_gemm(d, a, b, c, st);
    if (rc >= 0) return rc;
  }
  if (sb_gemm_eligible(d, A, B, C)) {
    const int rc = run_sb_gemm(d, a, b, c, st);
    if (rc >= 0) return rc;
  }
  if (gemm8_eligible(d, A, B, C)) return run_gemm8(d, a, b, c, st);
  SC_CHECK_ARG(!(d->a_sg || d->b_sg || d->c_sg),
               "tiled operands (a_sg / b_sg / c_sg) need the streamed matrix-core kernel: n_modes % 16 == 0, unit mode "
               "strides, no index tables, 16-byte aligned rows, near-full 32 x 32 tiles");
  if (!(d->flags & SC_GEMM_FORCE_VALU) && mfma_gemm_eligible(d))
    return run_mfma_gemm(d, a, b, c, st);
  if (g.Q > 4) return dispatch_modegemm_conj<4, 8>(g, d->conj_a, d->conj_b, a, b, c, st);
  return dispatch_modegemm_conj<4, 4>(g, d->conj_a, d->conj_b, a, b, c, st);
}

extern "C" int sc_modegemm_pair(const sc_modegemm_desc* d0, const float* A0, const float* B0, float* C0,
                                const sc_modegemm_desc* d1, const float* A1, const float* B1, float* C1,
                                void* stream) {
  SC_CHECK_ARG(d0 && d1 && A0 && B0 && C0 && A1 && B1 && C1, "null argument");
  if (d0->P > 0 && d0->Q > 0 && d0->R > 0 && d0->n_modes > 0 && d1->P > 0 && d1->Q > 0 && d1->R > 0) {
    Gemm8Bias nobias;
    std::memset(&nobias, 0, sizeof(nobias));
    int rc = run_sb_bwd(d0, (const cf32*)A0, (const cf32*)B0, (cf32*)C0, d1, (const cf32*)A1, (const cf32*)B1, (cf32*)C1,
                        (sc_stream_t)stream);
    if (rc >= 0) return rc;
    rc = run_gemm8_bwd(d0, (const cf32*)A0, (const cf32*)B0, (cf32*)C0, d1, (const cf32*)A1,
                       (const cf32*)B1, (cf32*)C1, nobias, (sc_stream_t)stream);
    if (rc >= 0) return rc;
  }
  const int rc = sc_modegemm(d0, A0, B0, C0, stream);
  return rc ? rc : sc_modegemm(d1, A1, B1, C1, stream);
}

// which launch(es) a pair with 16-byte aligned operands (B0 and A1 the same array) takes: 2 = ONE pass over the weight
// (k_modegemm_sb_bwd), 1 = one launch of k_modegemm_dma_bwd, 0 = two launches
extern "C" int sc_modegemm_pair_path(const sc_modegemm_desc* d0, const sc_modegemm_desc* d1) {
  if (!d0 || !d1) return 0;
  static const float* const al = reinterpret_cast<const float*>(uintptr_t(256));   // alignment probe only
  if (sb_bwd_eligible(d0, al, al, al, d1, al, al, al)) return 2;
  return sc_modegemm_pair_fused(d0, d1) ? 1 : 0;
}

extern "C" int sc_modegemm_pair_fused(const sc_modegemm_desc* d0, const sc_modegemm_desc* d1) {
  if (!d0 || !d1) return 0;
#ifdef SC_G8_NO_PAIR
  return 0;
#else
  static const float* const al = reinterpret_cast<const float*>(uintptr_t(256));   // alignment probe only
  if (!gemm8_eligible(d0, al, al, al) || !gemm8_eligible(d1, al, al, al)) return 0;
  if (!(d0->conj_a && !d0->conj_b && !d1->conj_a && d1->conj_b) || d0->n_modes != d1->n_modes) return 0;
  Gemm8Args g0, g1;
  gemm8_args(d0, g0, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), 0);
  gemm8_args(d1, g1, SC_G8_RESIDENT(false), SC_G8_RESIDENT(true), 0);
#if SC_G8_PAIR_BPW >= 1
  g0.G = g0.n_mg;
#endif
#if SC_G8_PAIR_BPW == 1
  g1.G = g1.n_mg;
#endif
  return !((g0.G & 7) || (g1.G & 7));
#endif
}

// launch geometry of k_modegemm_msum; returns the number of (mode split, r split) slots
static int64_t msum_geometry(ModeGemmArgs& g, bool* wide_out) {
  g.n_mt = (int)((g.M + SC_WAVE - 1) / SC_WAVE);
  // 4 x 8 outputs per wave (12 operand loads per 32 products) when the problem still yields enough workgroups,
  // 2 x 4 for small outputs
  const bool wide = g.P >= 16 && g.Q >= 8;
  const int PT = wide ? 4 : 2, QT = wide ? 8 : 4;
  g.n_pg = (int)((g.P + 4 * PT - 1) / (4 * PT));
  g.n_qt = (int)((g.Q + QT - 1) / QT);
  // mode splits: enough workgroups to fill the chip (~4096), as few partial sums per output as that allows
  int64_t splits = 4096 / ((int64_t)g.n_pg * g.n_qt);
  if (splits < 1) splits = 1;
  if (splits > g.n_mt) splits = g.n_mt;
  g.per_xcd = (int)splits;
  // when the mode tiles alone do not fill the chip (TFNO rank 0.1: 33 tiles x 20 output tiles = 660 workgroups, 2.8 waves
  // per SIMD, 45 % of the wave cycles issue-stalled: profiles/r03_tfno_pmc.txt) the reduction index is cut as well;
  // SC_MSUM_RSPLIT (environment, A-B) overrides
  static const int rsplit_env = [] { const char* e = SC_DIAG_ENV("SC_MSUM_RSPLIT"); return e ? std::atoi(e) : 0; }();
  int64_t rsplit = rsplit_env > 0 ? rsplit_env : 1;
  if (rsplit > g.R) rsplit = g.R;
  g.r_split = (int)rsplit;
  *wide_out = wide;
  return splits * rsplit;
}

template <bool CA, bool CB, bool PART = false>
static void launch_msum(const ModeGemmArgs& g0, const cf32* A, const cf32* B, cf32* C, sc_stream_t st) {
  ModeGemmArgs g = g0;
  bool wide;
  const int64_t total = msum_geometry(g, &wide) * g.n_pg * g.n_qt;
  if (wide) SC_LAUNCH((k_modegemm_msum<4, 8, CA, CB, PART>), dim3((unsigned)total), dim3(SC_BLOCK), 0, st, g, A, B, C);
  else SC_LAUNCH((k_modegemm_msum<2, 4, CA, CB, PART>), dim3((unsigned)total), dim3(SC_BLOCK), 0, st, g, A, B, C);
}

static void msum_args(const sc_modegemm_desc* d, ModeGemmArgs& g) {
  g.P = d->P; g.Q = d->Q; g.R = d->R; g.M = d->n_modes;
  g.a_sp = d->a_sp; g.a_sr = d->a_sr; g.a_sm = d->a_sm;
  g.b_sr = d->b_sr; g.b_sq = d->b_sq; g.b_sm = d->b_sm;
  g.c_sp = d->c_sp; g.c_sq = d->c_sq; g.c_sm = 0;
  g.b_idx = d->b_idx; g.c_idx = nullptr;
  g.accumulate = 1;
}

/* C[p,q] += sum_m sum_r opA(A[p,r,m]) opB(B[r,q,m]); C (strides c_sp, c_sq) zeroed by the caller */
extern "C" int sc_modegemm_msum(const sc_modegemm_desc* d, const float* A, const float* B, float* C,
                                void* stream) {
  SC_CHECK_ARG(d && A && B && C, "null argument");
  SC_CHECK_ARG(d->P >= 0 && d->Q >= 0 && d->R >= 0 && d->n_modes >= 0, "negative extent");
  if (d->P == 0 || d->Q == 0 || d->n_modes == 0 || d->R == 0) return 0;
  ModeGemmArgs g;
  msum_args(d, g);
  SC_CHECK_ARG(((g.P + 7) / 8) * ((g.Q + 3) / 4) < ((int64_t)1 << 30) && (g.M + 63) / 64 < ((int64_t)1 << 31),
               "problem too large for one launch grid");
  sc_stream_t st = (sc_stream_t)stream;
  const cf32* a = (const cf32*)A;
  const cf32* b = (const cf32*)B;
  cf32* c = (cf32*)C;
  if (!d->conj_a && !d->conj_b) launch_msum<false, false>(g, a, b, c, st);
  else if (d->conj_a && !d->conj_b) launch_msum<true, false>(g, a, b, c, st);
  else if (!d->conj_a && d->conj_b) launch_msum<false, true>(g, a, b, c, st);
  else launch_msum<true, true>(g, a, b, c, st);
  return sc_check_launch("k_modegemm_msum");
}

// C[p, q] = sum over modes and r (OVERWRITTEN, not accumulated) with a caller-provided workspace: the matrix-core
// kernel of sc_kernels_fmx.h where the problem qualifies, else k_modegemm_msum<PART>; one partial per workgroup /
// slot and a fixed-order reduction either way (bit-reproducible, unlike the atomics of sc_modegemm_msum)
static bool msum_slots_ok(const sc_modegemm_desc* d) {
  return ((d->P + 7) / 8) * ((d->Q + 3) / 4) < ((int64_t)1 << 30) && (d->n_modes + 63) / 64 < ((int64_t)1 << 31) &&
         d->P * d->Q < ((int64_t)1 << 31);
}
extern "C" size_t sc_modegemm_msum_workspace_bytes(const sc_modegemm_desc* d) {
  if (!d || d->P <= 0 || d->Q <= 0 || d->n_modes <= 0 || d->R <= 0) return 0;
  if (!fmx_msum_eligible(d)) {
    if (!msum_slots_ok(d)) return 0;
    ModeGemmArgs mg;
    msum_args(d, mg);
    bool wide;
    return (size_t)msum_geometry(mg, &wide) * (size_t)(d->P * d->Q) * sizeof(cf32) + 256;
  }
  FmxArgs g;
  size_t lds;
  fmx_msum_args(d, g, lds);
  return (size_t)g.n_wg * (size_t)(d->P * d->Q) * sizeof(cf32) + 256;
}

extern "C" int sc_modegemm_msum_path(const sc_modegemm_desc* d) {
  return d && d->P > 0 && d->Q > 0 && d->n_modes > 0 && d->R > 0 && fmx_msum_eligible(d) ? 1 : 0;
}

extern "C" int sc_modegemm_msum_ws(const sc_modegemm_desc* d, const float* A, const float* B, float* C, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  SC_CHECK_ARG(d && A && B && C && workspace, "null argument");
  SC_CHECK_ARG(d->P > 0 && d->Q > 0 && d->R > 0 && d->n_modes > 0, "empty extent");
  SC_CHECK_ARG(fmx_msum_eligible(d) || msum_slots_ok(d),
               "sc_modegemm_msum_ws: the problem does not qualify (sc_modegemm_msum_workspace_bytes == 0)");
  SC_CHECK_ARG(workspace_bytes >= sc_modegemm_msum_workspace_bytes(d), "workspace too small");
  sc_stream_t st = (sc_stream_t)stream;
  cf32* partial = (cf32*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const cf32* a = (const cf32*)A;
  const cf32* b = (const cf32*)B;
  if (!fmx_msum_eligible(d)) {
    ModeGemmArgs mg;
    msum_args(d, mg);
    bool wide;
    const int64_t slots = msum_geometry(mg, &wide);
    if (!d->conj_a && !d->conj_b) launch_msum<false, false, true>(mg, a, b, partial, st);
    else if (d->conj_a && !d->conj_b) launch_msum<true, false, true>(mg, a, b, partial, st);
    else if (!d->conj_a && d->conj_b) launch_msum<false, true, true>(mg, a, b, partial, st);
    else launch_msum<true, true, true>(mg, a, b, partial, st);
    int rc = sc_check_launch("k_modegemm_msum<slots>");
    if (rc) return rc;
    const int npc = (int)(d->P * d->Q);
    SC_LAUNCH(k_fmx_reduce, dim3((unsigned)((npc + 15) / 16)), dim3(16 * SC_FMX_RED_RG), 0, st, (const cf32*)partial, (int)slots, npc,
              (int)d->Q, (cf32*)C, d->c_sp, d->c_sq);
    return sc_check_launch("k_fmx_reduce");
  }
  FmxArgs g;
  size_t lds;
  fmx_msum_args(d, g, lds);
  const int64_t pa = (d->P + 3) / 4, pb = (d->Q + 3) / 4;
  int rc;
  if (pa <= 16 && pb <= 9 && d->Q <= 48) rc = run_fmx_msum_t<16, 9, 3>(d, g, lds, a, b, partial, st);
  else if (pa <= 9) rc = run_fmx_msum_t<9, 16, 4>(d, g, lds, a, b, partial, st);
  else rc = run_fmx_msum_t<16, 16, 4>(d, g, lds, a, b, partial, st);
  if (rc) return rc;
  const int npc = (int)(d->P * d->Q);
  SC_LAUNCH(k_fmx_reduce, dim3((unsigned)((npc + 15) / 16)), dim3(16 * SC_FMX_RED_RG), 0, st, (const cf32*)partial, g.n_wg, npc, (int)d->Q,
            (cf32*)C, d->c_sp, d->c_sq);
  return sc_check_launch("k_fmx_reduce");
}

// ------------------------------------------------------------------------------------------
// activation side of the factorized Tucker contraction: the nine products of a layer step from TWO host calls
// (include/sc_engine.h; the launches are those of the corresponding sc_modegemm / sc_modegemm_msum_ws calls)
// ------------------------------------------------------------------------------------------
static sc_modegemm_desc tkc_desc(int64_t P, int64_t Q, int64_t R, int64_t M, int64_t a_sp, int64_t a_sr, int64_t a_sm,
                                 int64_t b_sr, int64_t b_sq, int64_t b_sm, int64_t c_sp, int64_t c_sq, int64_t c_sm,
                                 int conj_a, int conj_b) {
  sc_modegemm_desc d;
  std::memset(&d, 0, sizeof(d));
  d.P = P; d.Q = Q; d.R = R; d.n_modes = M;
  d.a_sp = a_sp; d.a_sr = a_sr; d.a_sm = a_sm;
  d.b_sr = b_sr; d.b_sq = b_sq; d.b_sm = b_sm;
  d.c_sp = c_sp; d.c_sq = c_sq; d.c_sm = c_sm;
  d.conj_a = conj_a; d.conj_b = conj_b;
  return d;
}
// the two factor gradients: C[p, q] = sum_{m, r} conj(A[p, r, m]) B[r, q, m]
static sc_modegemm_desc tkc_gu_in_desc(const sc_tucker_chain_desc* c) {      // gu_in[i, f] <- xhat^H gz
  const int64_t M = c->n_modes;
  return tkc_desc(c->c_in, c->r_in, c->batch, M, M, c->c_in * M, 1, c->r_in * M, M, 1, c->r_in, 1, 0, 1, 0);
}
// gu_out[o, g] = sum_{b, m} gy[b, o, m] conj(t[b, g, m]) -- A = gy (P = c_out rows), B = t conjugated (Q = r_out columns).
// Round 4: until then the roles were the other way round (A = t conjugated, P = r_out = 36 rows, the result stored
// transposed): the accumulating kernel gives each of its four waves a 16-row block of P, so 36 rows kept three waves busy
// with four column tiles each -- 52 us against 26 us for the mirror-image gu_in launch (profiles/r04_tfno_kernel_stats.txt).
static sc_modegemm_desc tkc_gu_out_desc(const sc_tucker_chain_desc* c) {
  const int64_t M = c->n_modes;
  return tkc_desc(c->c_out, c->r_out, c->batch, M, M, c->c_out * M, 1, c->r_out * M, M, 1, c->r_out, 1, 0, 0, 1);
}
static size_t tkc_align(size_t v) { return (v + 255) & ~(size_t)255; }

static bool tkc_valid(const sc_tucker_chain_desc* d) {
  return d && d->batch > 0 && d->c_in > 0 && d->c_out > 0 && d->r_in > 0 && d->r_out > 0 && d->n_modes > 0;
}

extern "C" int sc_tucker_chain_forward(const sc_tucker_chain_desc* c, const float* xhat, const float* u_in,
                                       const float* t3, const float* u_out, float* z, float* t, float* yhat,
                                       void* stream) {
  SC_CHECK_ARG(tkc_valid(c), "sc_tucker_chain: null or empty descriptor");
  SC_CHECK_ARG(xhat && u_in && t3 && u_out && z && t && yhat, "null argument");
  const int64_t B = c->batch, Ci = c->c_in, Co = c->c_out, R1 = c->r_in, R2 = c->r_out, M = c->n_modes;
  // z = xhat u_in (mode-independent right operand)
  sc_modegemm_desc d = tkc_desc(B, R1, Ci, M, Ci * M, M, 1, R1, 1, 0, R1 * M, M, 1, 0, 0);
  int rc = sc_modegemm(&d, xhat, u_in, z, stream);
  if (rc) return rc;
  // t = z t3 (per-mode R1 x R2 products)
  d = tkc_desc(B, R2, R1, M, R1 * M, M, 1, R2 * M, M, 1, R2 * M, M, 1, 0, 0);
  rc = sc_modegemm(&d, z, t3, t, stream);
  if (rc) return rc;
  // yhat = t u_out^T
  d = tkc_desc(B, Co, R2, M, R2 * M, M, 1, 1, R2, 0, Co * M, M, 1, 0, 0);
  return sc_modegemm(&d, t, u_out, yhat, stream);
}

extern "C" size_t sc_tucker_chain_workspace_bytes(const sc_tucker_chain_desc* c) {
  if (!tkc_valid(c)) return 0;
  const sc_modegemm_desc di = tkc_gu_in_desc(c), dout = tkc_gu_out_desc(c);
  const size_t wi = sc_modegemm_msum_workspace_bytes(&di), wo = sc_modegemm_msum_workspace_bytes(&dout);
  return tkc_align((size_t)c->batch * c->r_out * c->n_modes * sizeof(cf32)) +
         tkc_align((size_t)c->batch * c->r_in * c->n_modes * sizeof(cf32)) + tkc_align(wi > wo ? wi : wo) + 512;
}

// one factor gradient: the matrix-core kernel with its fixed-order reduction where it qualifies, else zero fill +
// the atomic-add kernel
static int tkc_factor_grad(const sc_modegemm_desc* d, const float* A, const float* B, float* C, int64_t c_elems,
                           void* ws, size_t ws_bytes, void* stream) {
  const size_t need = sc_modegemm_msum_workspace_bytes(d);
  if (need && need <= ws_bytes) return sc_modegemm_msum_ws(d, A, B, C, ws, ws_bytes, stream);
  if (hipMemsetAsync(C, 0, (size_t)c_elems * sizeof(cf32), (sc_stream_t)stream) != hipSuccess)
    return sc_fail("sc_tucker_chain_backward: hipMemsetAsync failed");
  return sc_modegemm_msum(d, A, B, C, stream);
}

extern "C" int sc_tucker_chain_backward(const sc_tucker_chain_desc* c, const float* xhat, const float* u_in,
                                        const float* t3, const float* u_out, const float* z, const float* t,
                                        const float* gy, float* gxhat, float* gu_in, float* gt3, float* gu_out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  SC_CHECK_ARG(tkc_valid(c), "sc_tucker_chain: null or empty descriptor");
  SC_CHECK_ARG(xhat && u_in && t3 && u_out && z && t && gy && workspace, "null argument");
  SC_CHECK_ARG(workspace_bytes >= sc_tucker_chain_workspace_bytes(c), "workspace too small");
  const int64_t B = c->batch, Ci = c->c_in, Co = c->c_out, R1 = c->r_in, R2 = c->r_out, M = c->n_modes;
  unsigned char* w0 = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* gt = (float*)w0;
  float* gz = (float*)(w0 + tkc_align((size_t)B * R2 * M * sizeof(cf32)));
  unsigned char* wred = (unsigned char*)gz + tkc_align((size_t)B * R1 * M * sizeof(cf32));
  const size_t wred_bytes = workspace_bytes - (size_t)(wred - (unsigned char*)workspace);
  int rc;
  // Two streams (session 2): the six launches are small, latency-bound and only partly dependent, so the factor
  // gradients and gz run on the engine's side stream beside gt / gt3 / gxhat:
  //     main:  gt ----------> gt3 ---------> [gz ready] gxhat
  //     side:  gu_out -> [gt ready] gz ----> gu_in                      (the two reductions share `wred`: same stream)
  // Without a side stream (emulation, SC_NO_SIDE_STREAM=1) the same launches are issued in this order on `stream`.
  sc_stream_t main = (sc_stream_t)stream;
  ScSide* side = sc_side_get();
  void* ss = side ? (void*)side->stream : stream;
  const char* sync_fail = "sc_tucker_chain_backward: event record / wait failed";
  ScSideJoinGuard guard(side, main);            // every exit below -- error returns included -- joins the side stream
  if (side && !sc_side_fork(side, main)) return sc_fail(sync_fail);                 // the inputs are ready
  guard.armed = side != nullptr;
  // yhat = t u_out^T:  gt = gy conj(u_out);  gu_out[o, g] = sum_{b, m} conj(t[b, g, m]) gy[b, o, m]
  sc_modegemm_desc d = tkc_desc(B, R2, Co, M, Co * M, M, 1, R2, 1, 0, R2 * M, M, 1, 0, 1);
  if ((rc = sc_modegemm(&d, gy, u_out, gt, stream))) return rc;
  if (gu_out) {
    d = tkc_gu_out_desc(c);
    if ((rc = tkc_factor_grad(&d, gy, t, gu_out, Co * R2, wred, wred_bytes, ss))) return rc;
  }
  if (side && !sc_side_fork(side, main)) return sc_fail(sync_fail);                 // gt is ready
  // t = z t3:  gz = gt t3^H;  gt3[f, g, m] = sum_b conj(z[b, f, m]) gt[b, g, m]
  d = tkc_desc(B, R1, R2, M, R2 * M, M, 1, M, R2 * M, 1, R1 * M, M, 1, 0, 1);
  if ((rc = sc_modegemm(&d, gt, t3, gz, ss))) return rc;
  if (gt3) {
    d = tkc_desc(R1, R2, B, M, M, R1 * M, 1, R2 * M, M, 1, R2 * M, M, 1, 1, 0);
    if ((rc = sc_modegemm(&d, z, gt, gt3, stream))) return rc;
  }
  if (side && !sc_side_join(side, main)) return sc_fail(sync_fail);                 // gz (and gu_out) for main, after gt3
  if (gu_in) {
    d = tkc_gu_in_desc(c);                                                         // gu_in[i, f] = sum conj(xhat) gz
    if ((rc = tkc_factor_grad(&d, xhat, gz, gu_in, Ci * R1, wred, wred_bytes, ss))) return rc;
  }
  // z = xhat u_in:  gxhat = gz u_in^H
  if (gxhat) {
    d = tkc_desc(B, Ci, R1, M, R1 * M, M, 1, 1, R1, 0, Ci * M, M, 1, 0, 1);
    if ((rc = sc_modegemm(&d, gz, u_in, gxhat, stream))) return rc;
  }
  guard.armed = false;
  if (side && !sc_side_join(side, main)) return sc_fail(sync_fail);                 // gu_in
  return 0;
}

// ------------------------------------------------------------------------------------------
// Round 5: the same nine products as ONE launch each way (sc_kernels_tkchain.h): a workgroup owns four modes for the
// whole batch and walks the chain out of LDS; T3 travels as a mode-major copy T3m[m][f][g] the forward call writes and
// the backward call reads.
// ------------------------------------------------------------------------------------------
static uint32_t tkc_inv(int64_t n) { return (uint32_t)((((uint64_t)1 << 32) + (uint64_t)n - 1) / (uint64_t)n); }
static bool tkc_aligned(const void* p) { return ((uintptr_t)p & 15) == 0; }
extern "C" int sc_tucker_chain_fused_supported(const sc_tucker_chain_desc* c) {
  // OPT-IN (SC_TKC=1): measured on MI355X at configs[2] the one-launch-each-way kernels are as accurate as the nine
  // launches and SLOWER -- forward 104-114 us against 68.6 us, backward 182-195 us against 139 us
  // (profiles/r05_tkchain_ab.txt: 55 / 80 us of that is the staging / emit skeleton of 32-byte pieces with one
  // workgroup per compute unit, and 528 four-mode tiles on 256 units are three rounds where 2.06 would do).
  const char* e = std::getenv("SC_TKC");                  // read per call: tests switch it inside one process
  if (!(e && std::atoi(e) != 0) || !tkc_valid(c)) return 0;
  if (c->batch > 32 || c->c_in > 64 || c->c_out > 64 || c->r_in > 48 || c->r_out > 48) return 0;
  if ((c->batch | c->c_in | c->c_out | c->r_in | c->r_out | c->n_modes) & 3) return 0;
  if (c->n_modes / 4 >= ((int64_t)1 << 30)) return 0;
  const size_t lf = (size_t)tkc_layout((int)c->batch, (int)c->c_in, (int)c->c_out, (int)c->r_in, (int)c->r_out, false).total,
               lb = (size_t)tkc_layout((int)c->batch, (int)c->c_in, (int)c->c_out, (int)c->r_in, (int)c->r_out, true).total;
  return (lf > lb ? lf : lb) * sizeof(cf32) <= (size_t)160 * 1024 ? 1 : 0;
}
static void tkc_args(const sc_tucker_chain_desc* c, TkcArgs& g) {
  std::memset(&g, 0, sizeof(g));
  g.B = (int)c->batch; g.Ci = (int)c->c_in; g.Co = (int)c->c_out; g.R1 = (int)c->r_in; g.R2 = (int)c->r_out;
  g.M = c->n_modes;
  g.n_tiles = (int)(c->n_modes / 4);
  // persistent: one workgroup per compute unit (a tile needs ~147 KB of LDS); SC_TKC_WGS (environment, A-B)
  static const int env = [] { const char* e = SC_DIAG_ENV("SC_TKC_WGS"); return e ? std::atoi(e) : 0; }();
  const int cap = env > 0 ? env : sc_cu_count();
  g.n_wg = g.n_tiles < cap ? g.n_tiles : cap;
  g.inv_ci = tkc_inv(g.Ci); g.inv_co = tkc_inv(g.Co); g.inv_r1 = tkc_inv(g.R1); g.inv_r2 = tkc_inv(g.R2);
  g.inv_r12 = tkc_inv((int64_t)g.R1 * g.R2);
  g.abl = tucker_abl();
}
static int tkc_transpose(const cf32* in, cf32* out, int64_t rows, int64_t cols, sc_stream_t st) {
  const int64_t tr = (rows + 31) / 32, tc = (cols + 31) / 32;
  if (tr * tc >= ((int64_t)1 << 31)) return sc_fail("sc_tucker_chain: transpose grid too large");
  SC_LAUNCH(k_tkc_transpose, dim3((unsigned)(tr * tc)), dim3(256), 0, st, in, out, rows, cols, (int)tc);
  return sc_check_launch("k_tkc_transpose");
}
extern "C" size_t sc_tucker_chain_t3m_bytes(const sc_tucker_chain_desc* c) {
  return tkc_valid(c) ? (size_t)c->n_modes * c->r_in * c->r_out * sizeof(cf32) : 0;
}
extern "C" int sc_tucker_chain_forward_fused(const sc_tucker_chain_desc* c, const float* xhat, const float* u_in,
                                             const float* t3, const float* u_out, float* t3m, float* z, float* t,
                                             float* yhat, void* stream) {
  SC_CHECK_ARG(tkc_valid(c), "sc_tucker_chain: null or empty descriptor");
  SC_CHECK_ARG(xhat && u_in && t3 && u_out && t3m && z && t && yhat, "null argument");
  SC_CHECK_ARG(sc_tucker_chain_fused_supported(c), "sc_tucker_chain_forward_fused: shape outside the fused kernels' limits");
  SC_CHECK_ARG(tkc_aligned(xhat) && tkc_aligned(t3m) && tkc_aligned(z) && tkc_aligned(t) && tkc_aligned(yhat),
               "sc_tucker_chain_forward_fused: 16-byte aligned tensors");
  sc_stream_t st = (sc_stream_t)stream;
  int rc = tkc_transpose((const cf32*)t3, (cf32*)t3m, c->r_in * c->r_out, c->n_modes, st);
  if (rc) return rc;
  TkcArgs g;
  tkc_args(c, g);
  g.xhat = (const cf32*)xhat; g.u_in = (const cf32*)u_in; g.t3m = (const cf32*)t3m; g.u_out = (const cf32*)u_out;
  g.z = (cf32*)z; g.t = (cf32*)t; g.yhat = (cf32*)yhat;
  const size_t lds = (size_t)tkc_layout(g.B, g.Ci, g.Co, g.R1, g.R2, false).total * sizeof(cf32);
  SC_FMX_ATTR((k_tkc_fwd<16, 18>), lds);
  SC_LAUNCH((k_tkc_fwd<16, 18>), dim3((unsigned)g.n_wg), dim3(256), lds, st, g);
  return sc_check_launch("k_tkc_fwd");
}
extern "C" size_t sc_tucker_chain_backward_fused_workspace_bytes(const sc_tucker_chain_desc* c) {
  if (!tkc_valid(c)) return 0;
  TkcArgs g;
  tkc_args(c, g);
  return tkc_align(sc_tucker_chain_t3m_bytes(c)) +
         tkc_align((size_t)g.n_wg * ((size_t)c->c_out * c->r_out + (size_t)c->c_in * c->r_in) * sizeof(cf32)) + 512;
}
extern "C" int sc_tucker_chain_backward_fused(const sc_tucker_chain_desc* c, const float* xhat, const float* u_in,
                                              const float* t3m, const float* u_out, const float* z, const float* t,
                                              const float* gy, float* gxhat, float* gu_in, float* gt3, float* gu_out,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  SC_CHECK_ARG(tkc_valid(c), "sc_tucker_chain: null or empty descriptor");
  SC_CHECK_ARG(xhat && u_in && t3m && u_out && z && t && gy && gt3 && workspace, "null argument");
  SC_CHECK_ARG(sc_tucker_chain_fused_supported(c), "sc_tucker_chain_backward_fused: shape outside the fused kernels' limits");
  SC_CHECK_ARG(workspace_bytes >= sc_tucker_chain_backward_fused_workspace_bytes(c), "workspace too small");
  SC_CHECK_ARG(tkc_aligned(xhat) && tkc_aligned(t3m) && tkc_aligned(z) && tkc_aligned(t) && tkc_aligned(gy) &&
               (!gxhat || tkc_aligned(gxhat)), "sc_tucker_chain_backward_fused: 16-byte aligned tensors");
  sc_stream_t st = (sc_stream_t)stream;
  unsigned char* w0 = (unsigned char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  TkcArgs g;
  tkc_args(c, g);
  g.xhat = (const cf32*)xhat; g.u_in = (const cf32*)u_in; g.t3m = (const cf32*)t3m; g.u_out = (const cf32*)u_out;
  g.zin = (const cf32*)z; g.tin = (const cf32*)t; g.gy = (const cf32*)gy; g.gxhat = (cf32*)gxhat;
  g.gt3m = (cf32*)w0;
  g.partial = (cf32*)(w0 + tkc_align(sc_tucker_chain_t3m_bytes(c)));
  const size_t lds = (size_t)tkc_layout(g.B, g.Ci, g.Co, g.R1, g.R2, true).total * sizeof(cf32);
  SC_FMX_ATTR((k_tkc_bwd<16, 12, 18>), lds);
  SC_LAUNCH((k_tkc_bwd<16, 12, 18>), dim3((unsigned)g.n_wg), dim3(256), lds, st, g);
  int rc = sc_check_launch("k_tkc_bwd");
  if (rc) return rc;
  if ((rc = tkc_transpose((const cf32*)g.gt3m, (cf32*)gt3, c->n_modes, c->r_in * c->r_out, st))) return rc;
  if (gu_in || gu_out) {
    const int n_out = g.Co * g.R2, n_all = n_out + g.Ci * g.R1;
    SC_LAUNCH(k_tkc_reduce, dim3((unsigned)((n_all + 15) / 16)), dim3(256), 0, st, (const cf32*)g.partial, g.n_wg, n_out,
              n_all, (cf32*)gu_out, (cf32*)gu_in);
    rc = sc_check_launch("k_tkc_reduce");
  }
  return rc;
}

// ------------------------------------------------------------------------------------------
// Round 5: peer-store exchange (sc_kernels_peer.h).  A window = 512 header bytes (flags [8] at 0, this rank's epoch at
// 256, its workgroup ticket at 264, the wait kernel's error word at 272 and spin budget at 280: round 6) + the data; fine-grained device memory so that stores from a peer GPU and the flag
// loads of the owner are coherent inside running kernels; shared through HIP IPC handles.
// ------------------------------------------------------------------------------------------
#define SC_PEER_HEADER 512
extern "C" int sc_peer_window_alloc(size_t data_bytes, void** ptr, void* handle64) {
  SC_CHECK_ARG(ptr && handle64 && data_bytes > 0, "null argument");
  const size_t bytes = SC_PEER_HEADER + ((data_bytes + 255) & ~(size_t)255);
  void* p = nullptr;
#ifndef SC_EMU
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
  SC_CHECK_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  SC_CHECK_HIP(hipMemset(p, 0, bytes));
  SC_CHECK_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  SC_CHECK_HIP(hipIpcGetMemHandle(&h, p));
  std::memcpy(handle64, &h, 64);
#else
  p = std::calloc(bytes, 1);
  std::memset(handle64, 0, 64);
  std::memcpy(handle64, &p, sizeof(p));               // emulation: one process, the handle is the pointer
#endif
  *ptr = p;
  return 0;
}
extern "C" int sc_peer_window_open(const void* handle64, void** ptr) {
  SC_CHECK_ARG(ptr && handle64, "null argument");
#ifndef SC_EMU
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, 64);
  SC_CHECK_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
#else
  std::memcpy(ptr, handle64, sizeof(*ptr));
#endif
  return 0;
}
extern "C" int sc_peer_window_close(void* ptr) {
#ifndef SC_EMU
  if (ptr) SC_CHECK_HIP(hipIpcCloseMemHandle(ptr));
#endif
  return 0;
}
extern "C" int sc_peer_window_free(void* ptr) {
#ifndef SC_EMU
  if (ptr) SC_CHECK_HIP(hipFree(ptr));
#else
  std::free(ptr);
#endif
  return 0;
}
// spin budget of k_peer_wait on THIS rank's window (milliseconds; 0 = unbounded, < 0 = leave as it is) and the error word it
// leaves behind when the budget runs out (0 = none, 1 + p = the flag of peer p never came); reading clears it.
// Synchronous with respect to the device (the header is fine-grained memory read through a blocking copy).
extern "C" int sc_peer_window_control(void* own_window, int64_t spin_budget_ms, int32_t* error_out) {
  SC_CHECK_ARG(own_window, "null argument");
  unsigned char* mine = (unsigned char*)own_window;
  if (spin_budget_ms >= 0) {
    const unsigned long long ticks = (unsigned long long)spin_budget_ms * 100000ull;      // 100 MHz wall clock
#ifndef SC_EMU
    SC_CHECK_HIP(hipMemcpy(mine + 280, &ticks, 8, hipMemcpyHostToDevice));
#else
    std::memcpy(mine + 280, &ticks, 8);
#endif
  }
  if (error_out) {
    unsigned int e = 0, zero = 0;
#ifndef SC_EMU
    SC_CHECK_HIP(hipMemcpy(&e, mine + 272, 4, hipMemcpyDeviceToHost));
    if (e) SC_CHECK_HIP(hipMemcpy(mine + 272, &zero, 4, hipMemcpyHostToDevice));
#else
    std::memcpy(&e, mine + 272, 4);
    std::memcpy(mine + 272, &zero, 4);
#endif
    *error_out = (int32_t)e;
  }
  return 0;
}
extern "C" int sc_peer_all_to_all(const sc_peer_exchange* d, const void* send, void* recv, void* stream) {
  SC_CHECK_ARG(d && send && recv, "null argument");
  SC_CHECK_ARG(d->world >= 1 && d->world <= 8 && d->rank >= 0 && d->rank < d->world, "sc_peer_all_to_all: 1..8 ranks of one node");
  SC_CHECK_ARG(d->block_bytes > 0 && d->block_bytes % 16 == 0, "sc_peer_all_to_all: blocks of whole 16-byte units");
  SC_CHECK_ARG((((uintptr_t)send | (uintptr_t)recv) & 15) == 0, "sc_peer_all_to_all: 16-byte aligned buffers");
  PeerArgs g;
  std::memset((void*)&g, 0, sizeof(g));
  for (int p = 0; p < d->world; ++p) {
    SC_CHECK_ARG(d->peer_window[p], "sc_peer_all_to_all: a peer window is not mapped");
    unsigned char* base = (unsigned char*)d->peer_window[p];
    g.peer_flag[p] = (unsigned long long*)base;
    g.peer_win[p] = (sc_f4*)(base + SC_PEER_HEADER);
  }
  unsigned char* mine = (unsigned char*)d->peer_window[d->rank];
  g.my_flag = (unsigned long long*)mine;
  g.epoch = (unsigned long long*)(mine + 256);
  g.ticket = (unsigned int*)(mine + 264);
  g.error = (unsigned int*)(mine + 272);
  g.spin_budget = (const unsigned long long*)(mine + 280);
  g.my_win = (const sc_f4*)(mine + SC_PEER_HEADER);
  g.send = (const sc_f4*)send;
  g.recv = (sc_f4*)recv;
  g.block16 = d->block_bytes / 16;
  g.P = d->world;
  g.rank = d->rank;
  // 557 KB per peer at configs[3] on 8 ranks: 17 workgroups of 256 lanes per peer, 8 rounds of 4 KB each; at most ~512
  // workgroups in all (few ranks, large blocks: more per peer)
  const int64_t want = (g.block16 + 2047) / 2048, cap = 512 / g.P;
  g.wg_per_peer = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  sc_stream_t st = (sc_stream_t)stream;
  SC_LAUNCH(k_peer_put, dim3((unsigned)(g.P * g.wg_per_peer)), dim3(256), 0, st, g);
  int rc = sc_check_launch("k_peer_put");
  if (rc) return rc;
  SC_LAUNCH(k_peer_wait, dim3(1), dim3(64), 0, st, g);
  if ((rc = sc_check_launch("k_peer_wait"))) return rc;
  const int64_t want_c = ((int64_t)g.P * g.block16 + 1023) / 1024;          // four 16-byte units per lane and workgroup pass
  const int n_wg = (int)(want_c < 1 ? 1 : (want_c > 512 ? 512 : want_c));
  SC_LAUNCH(k_peer_copy, dim3((unsigned)n_wg), dim3(256), 0, st, g, n_wg);
  return sc_check_launch("k_peer_copy");
}

// ------------------------------------------------------------------------------------------
// pointwise MLP of an FNO block (sc_kernels_pmlp.h)
// ------------------------------------------------------------------------------------------
template <int CI, int CH, int CO>
static void launch_pmlp_fwd(const PmlpArgs& g, bool gate, int act, sc_stream_t st) {
  const dim3 grid((unsigned)g.n_wg), block(256);
  if (gate && act) SC_LAUNCH((k_pmlp_fwd<CI, CH, CO, true, 1>), grid, block, 0, st, g);
  else if (gate) SC_LAUNCH((k_pmlp_fwd<CI, CH, CO, true, 0>), grid, block, 0, st, g);
  else if (act) SC_LAUNCH((k_pmlp_fwd<CI, CH, CO, false, 1>), grid, block, 0, st, g);
  else SC_LAUNCH((k_pmlp_fwd<CI, CH, CO, false, 0>), grid, block, 0, st, g);
}

static int pmlp_shape_id(const sc_pmlp_desc* d) {
  if (d->c_in % 32 || d->c_hid % 32 || d->c_out % 32) return 0;
  return (int)(d->c_in / 32) * 100 + (int)(d->c_hid / 32) * 10 + (int)(d->c_out / 32);
}

// the pointwise kernels address a lane as (wave-uniform row base) + a 32-bit BYTE offset of up to 4 * (31 + 4 * spatial)
// (sc_at, sc_device.h)
#define SC_PW_MAX_SPATIAL ((int64_t)1 << 28)
extern "C" int sc_pointwise_mlp_forward(const sc_pmlp_desc* d, const float* x, const float* w1, const float* b1,
                                        const float* w2, const float* b2, const float* skip_src, const float* gate,
                                        float* out, void* stream) {
  SC_CHECK_ARG(d, "null argument");
  if (d->batch <= 0 || d->spatial <= 0) return 0;
  SC_CHECK_ARG(x && w1 && w2 && out, "null argument");
  SC_CHECK_ARG(d->act == SC_ACT_NONE || d->act == SC_ACT_GELU, "unknown activation");
  SC_CHECK_ARG((skip_src == nullptr) == (gate == nullptr), "skip_src and gate come together");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise MLP: the spatial size must be a multiple of 32 below 2^28 points");
  PmlpArgs g;
  g.x = x; g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.skip = skip_src; g.gate = gate; g.out = out;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  const int64_t wgs = (g.n_tiles + 3) / 4;
  // persistent: the weight tables are built once per workgroup (SC_PMLP_FWD_WGS = workgroup count, A-B)
  static const int wgs_env = [] { const char* e = SC_DIAG_ENV("SC_PMLP_FWD_WGS"); return e ? std::atoi(e) : 0; }();
  const int64_t resident = wgs_env > 0 ? wgs_env : 2048;     // 2048 / 768 / 512 workgroups: 0.381 / 0.407 / 0.437 ms (session 2)
  g.n_wg = (int)(wgs < resident ? wgs : resident);
  sc_stream_t st = (sc_stream_t)stream;
  switch (pmlp_shape_id(d)) {
    case 111: launch_pmlp_fwd<1, 1, 1>(g, gate != nullptr, d->act, st); break;
    case 212: launch_pmlp_fwd<2, 1, 2>(g, gate != nullptr, d->act, st); break;
    case 222: launch_pmlp_fwd<2, 2, 2>(g, gate != nullptr, d->act, st); break;
    case 424: launch_pmlp_fwd<4, 2, 4>(g, gate != nullptr, d->act, st); break;
    default:
      return sc_fail("sc_engine: pointwise MLP: channel counts (c_in, c_hid, c_out) must be one of (32,32,32), "
                     "(64,32,64), (64,64,64), (128,64,128)");
  }
  return sc_check_launch("k_pmlp_fwd");
}

#define SC_PMLP_RED_GROUPS 16
// waves per workgroup of the backward kernel: one per SIMD (its register file holds the weight-gradient accumulators)
template <int CC, int CH>
static void launch_pblock_fwd(const PblockArgs& g, int act, sc_stream_t st) {
  const dim3 grid((unsigned)g.n_wg), block(256);
  if (act) SC_LAUNCH((k_pblock_fwd<CC, CH, 1>), grid, block, 0, st, g);
  else SC_LAUNCH((k_pblock_fwd<CC, CH, 0>), grid, block, 0, st, g);
}

extern "C" int sc_pointwise_block_forward(const sc_pmlp_desc* d, const float* conv, const float* x, const float* ws,
                                          const float* bs, const float* w1, const float* b1, const float* w2,
                                          const float* b2, const float* gate, float* y, float* pre, float* out,
                                          void* stream) {
  SC_CHECK_ARG(d, "null argument");
  if (d->batch <= 0 || d->spatial <= 0) return 0;
  SC_CHECK_ARG(conv && x && ws && w1 && w2 && gate && y && out, "null argument");
  SC_CHECK_ARG(d->act == SC_ACT_NONE || d->act == SC_ACT_GELU || d->act == SC_ACT_GELU_DGRAD, "unknown activation");
  SC_CHECK_ARG(d->act == SC_ACT_NONE || pre, "the pre-activation buffer is required with SC_ACT_GELU / SC_ACT_GELU_DGRAD");
  SC_CHECK_ARG(d->c_in == d->c_out, "pointwise block pass: c_in == c_out (the linear skip maps the block's channels onto themselves)");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise block pass: the spatial size must be a multiple of 32 below 2^28 points");
  PblockArgs g;
  g.conv = conv; g.x = x; g.ws = ws; g.bs = bs; g.w1 = w1; g.b1 = b1; g.w2 = w2; g.b2 = b2; g.gate = gate;
  g.y = y; g.pre = d->act != SC_ACT_NONE ? pre : nullptr; g.out = out;
  g.pre_is_grad = d->act == SC_ACT_GELU_DGRAD ? 1 : 0;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  const int64_t wgs = (g.n_tiles + 3) / 4;
  static const int wgs_env = [] { const char* e = SC_DIAG_ENV("SC_PBLOCK_FWD_WGS"); return e ? std::atoi(e) : 0; }();
  const int64_t cap = wgs_env > 0 ? wgs_env : 2048;          // persistent: the weight tables are built once per workgroup
  g.n_wg = (int)(wgs < cap ? wgs : cap);
  sc_stream_t st = (sc_stream_t)stream;
  switch (pmlp_shape_id(d)) {
    case 111: launch_pblock_fwd<1, 1>(g, d->act != SC_ACT_NONE, st); break;
    case 212: launch_pblock_fwd<2, 1>(g, d->act != SC_ACT_NONE, st); break;
    case 222: launch_pblock_fwd<2, 2>(g, d->act != SC_ACT_NONE, st); break;
    default:
      return sc_fail("sc_engine: pointwise block pass: (c, c_hid, c) must be (32,32,32), (64,32,64) or (64,64,64)");
  }
  return sc_check_launch("k_pblock_fwd");
}

static int pmlp_bwd_waves(const sc_pmlp_desc*) { return 4; }
static int pmlp_bwd_wgs(const sc_pmlp_desc* d) {
  const int nw = pmlp_bwd_waves(d);
  const int64_t wgs = (d->batch * (d->spatial / 32) + nw - 1) / nw;
  return (int)(wgs < 256 ? wgs : 256);                      // one workgroup per CU
}

template <int CI, int CH, int CO>
static size_t pmlp_ws_floats(int n_wg) {
  typedef PmlpDims<CI, CH, CO> D;
  return (size_t)(n_wg + SC_PMLP_RED_GROUPS) * D::NP;       // one partial per workgroup + the first reduction stage
}

extern "C" size_t sc_pointwise_mlp_workspace_bytes(const sc_pmlp_desc* d) {
  if (!d || d->batch <= 0 || d->spatial <= 0) return 0;
  const int n_wg = pmlp_bwd_wgs(d);
  size_t f = 0;
  switch (pmlp_shape_id(d)) {
    case 111: f = pmlp_ws_floats<1, 1, 1>(n_wg); break;
    case 212: f = pmlp_ws_floats<2, 1, 2>(n_wg); break;
    case 222: f = pmlp_ws_floats<2, 2, 2>(n_wg); break;
    default: return 0;                                      // (128, 64, 128): forward kernel only
  }
  return f * sizeof(float) + 256;
}

template <int CI, int CH, int CO, int NW>
static void launch_pmlp_bwd(PmlpBwdArgs g, float* ws, bool gate, int act, float* gw1, float* gb1, float* gw2, float* gb2,
                            float* ggate, sc_stream_t st) {
  typedef PmlpDims<CI, CH, CO> D;
  g.partial = ws;
  float* stage = ws + (size_t)g.n_wg * D::NP;
  const dim3 grid((unsigned)g.n_wg), block(64 * NW);
  if (gate && act) SC_LAUNCH((k_pmlp_bwd<CI, CH, CO, true, 1, NW>), grid, block, 0, st, g);
  else if (gate) SC_LAUNCH((k_pmlp_bwd<CI, CH, CO, true, 0, NW>), grid, block, 0, st, g);
  else if (act) SC_LAUNCH((k_pmlp_bwd<CI, CH, CO, false, 1, NW>), grid, block, 0, st, g);
  else SC_LAUNCH((k_pmlp_bwd<CI, CH, CO, false, 0, NW>), grid, block, 0, st, g);
  const unsigned nb = (unsigned)((D::NP + 255) / 256);
  const int groups = g.n_wg < SC_PMLP_RED_GROUPS ? g.n_wg : SC_PMLP_RED_GROUPS;
  SC_LAUNCH(k_pmlp_reduce1, dim3(nb, (unsigned)groups), dim3(256), 0, st, (const float*)g.partial, g.n_wg, groups, (int)D::NP,
            stage);
  SC_LAUNCH(k_pmlp_reduce, dim3(nb), dim3(256), 0, st, (const float*)stage, groups, (int)D::NP, (int)D::oW1, (int)D::oB1,
            (int)D::oB2, (int)D::oG, gw2, gw1, gb1, gb2, gate ? ggate : (float*)nullptr);
}

extern "C" int sc_pointwise_mlp_backward(const sc_pmlp_desc* d, const float* x, const float* w1, const float* b1,
                                         const float* w2, const float* b2, const float* skip_src, const float* gate,
                                         const float* gout, float* gx, float* gw1, float* gb1, float* gw2, float* gb2,
                                         float* gskip_src, float* ggate, void* workspace, void* stream) {
  return sc_pointwise_mlp_backward_ex(d, x, nullptr, w1, b1, w2, b2, skip_src, gate, gout, gx, gw1, gb1, gw2, gb2, gskip_src,
                                      ggate, workspace, stream);
}

extern "C" int sc_pointwise_mlp_backward_ex(const sc_pmlp_desc* d, const float* x, const float* x_pre, const float* w1,
                                            const float* b1, const float* w2, const float* b2, const float* skip_src,
                                            const float* gate, const float* gout, float* gx, float* gw1, float* gb1,
                                            float* gw2, float* gb2, float* gskip_src, float* ggate, void* workspace,
                                            void* stream) {
  SC_CHECK_ARG(d, "null argument");
  SC_CHECK_ARG(d->batch > 0 && d->spatial > 0, "pointwise MLP backward: empty input");
  SC_CHECK_ARG(x && w1 && w2 && gout && gx && gw1 && gw2 && workspace, "null argument");
  SC_CHECK_ARG(d->act == SC_ACT_NONE || d->act == SC_ACT_GELU || d->act == SC_ACT_GELU_DGRAD, "unknown activation");
  SC_CHECK_ARG(d->act != SC_ACT_GELU_DGRAD || x_pre, "SC_ACT_GELU_DGRAD: x_pre (the stored derivative) is required");
  SC_CHECK_ARG((skip_src == nullptr) == (gate == nullptr), "skip_src and gate come together");
  SC_CHECK_ARG(!gate || (gskip_src && ggate), "a gated forward needs gskip_src and ggate");
  SC_CHECK_ARG((b1 != nullptr || gb1 == nullptr) && (b2 != nullptr || gb2 == nullptr), "bias gradient without a bias");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise MLP: the spatial size must be a multiple of 32 below 2^28 points");
  PmlpBwdArgs g;
  g.x = x; g.b1 = b1; g.b2 = b2; g.skip = skip_src; g.gate = gate; g.gout = gout; g.gx = gx; g.gskip = gskip_src;
  g.w1 = w1; g.w2 = w2; g.x_pre = x_pre; g.partial = nullptr; g.lw = nullptr;
  g.x_pre_is_grad = d->act == SC_ACT_GELU_DGRAD ? 1 : 0;
  const int act = d->act == SC_ACT_NONE ? 0 : 1;           // the MLP's own closing activation
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  g.n_wg = pmlp_bwd_wgs(d);
  sc_stream_t st = (sc_stream_t)stream;
  float* ws = (float*)workspace;
  const bool gt = gate != nullptr;
  switch (pmlp_shape_id(d)) {
    case 111: launch_pmlp_bwd<1, 1, 1, 4>(g, ws, gt, act, gw1, gb1, gw2, gb2, ggate, st); break;
    case 212: launch_pmlp_bwd<2, 1, 2, 4>(g, ws, gt, act, gw1, gb1, gw2, gb2, ggate, st); break;
    case 222: launch_pmlp_bwd<2, 2, 2, 4>(g, ws, gt, act, gw1, gb1, gw2, gb2, ggate, st); break;
    default:
      return sc_fail("sc_engine: pointwise MLP backward: channel counts (c_in, c_hid, c_out) must be one of (32,32,32), "
                     "(64,32,64), (64,64,64); (128,64,128) has the forward pass only (operand tables + gradient image "
                     "exceed a CU's LDS)");
  }
  return sc_check_launch("k_pmlp_bwd");
}

// ---- round 6: the MLP pass of a block's backward with the DATA path of the linear skip riding along (k_pmlp_bwd<.., LIN>):
//      gz = the gradient of the Fourier layer's pre-activation, gin = W_s^T gz + gate (.) g_z -- the whole gradient of the
//      block input outside the spectral convolution, i.e. the addend of sc_layer_backward_ex.  What is left of the linear
//      skip's backward is its weight gradient: sc_pointwise_linear_backward_ex(gx = NULL).
template <int CI, int CH, int CO>
static void launch_pblock_bwd(PmlpBwdArgs g, float* ws, int act, float* gw1, float* gb1, float* gw2, float* gb2, float* ggate,
                              sc_stream_t st) {
  typedef PmlpDims<CI, CH, CO> D;
  g.partial = ws;
  float* stage = ws + (size_t)g.n_wg * D::NP;
  const dim3 grid((unsigned)g.n_wg), block(256);
  if (act) SC_LAUNCH((k_pmlp_bwd<CI, CH, CO, true, 1, 4, true>), grid, block, 0, st, g);
  else SC_LAUNCH((k_pmlp_bwd<CI, CH, CO, true, 0, 4, true>), grid, block, 0, st, g);
  const unsigned nb = (unsigned)((D::NP + 255) / 256);
  const int groups = g.n_wg < SC_PMLP_RED_GROUPS ? g.n_wg : SC_PMLP_RED_GROUPS;
  SC_LAUNCH(k_pmlp_reduce1, dim3(nb, (unsigned)groups), dim3(256), 0, st, (const float*)g.partial, g.n_wg, groups, (int)D::NP,
            stage);
  SC_LAUNCH(k_pmlp_reduce, dim3(nb), dim3(256), 0, st, (const float*)stage, groups, (int)D::NP, (int)D::oW1, (int)D::oB1,
            (int)D::oB2, (int)D::oG, gw2, gw1, gb1, gb2, ggate);
}

extern "C" int sc_pointwise_block_backward_supported(const sc_pmlp_desc* d) {
  if (!d || d->c_in != d->c_out || d->spatial % 32 || d->spatial >= SC_PW_MAX_SPATIAL) return 0;
  const int id = pmlp_shape_id(d);
  return (id == 111 || id == 212) ? 1 : 0;                 // (64, 64, 64): tables + scratch + gradient image exceed 160 KB of LDS
}

extern "C" int sc_pointwise_block_backward(const sc_pmlp_desc* d, const float* y, const float* y_pre, const float* x,
                                           const float* ws_lin, const float* w1, const float* b1, const float* w2,
                                           const float* b2, const float* gate, const float* gout, float* gz, float* gin,
                                           float* gw1, float* gb1, float* gw2, float* gb2, float* ggate, void* workspace,
                                           void* stream) {
  SC_CHECK_ARG(d, "null argument");
  SC_CHECK_ARG(d->batch > 0 && d->spatial > 0, "pointwise block backward: empty input");
  SC_CHECK_ARG(y && x && ws_lin && w1 && w2 && gate && gout && gz && gin && gw1 && gw2 && ggate && workspace, "null argument");
  SC_CHECK_ARG(d->act == SC_ACT_NONE || d->act == SC_ACT_GELU || d->act == SC_ACT_GELU_DGRAD, "unknown activation");
  SC_CHECK_ARG((d->act == SC_ACT_NONE) == (y_pre == nullptr), "y_pre comes with SC_ACT_GELU / SC_ACT_GELU_DGRAD (the block's y = gelu(y_pre))");
  SC_CHECK_ARG((b1 != nullptr || gb1 == nullptr) && (b2 != nullptr || gb2 == nullptr), "bias gradient without a bias");
  SC_CHECK_ARG(sc_pointwise_block_backward_supported(d), "pointwise block backward: (channels, hidden) must be (32, 32) or (64, 32)");
  PmlpBwdArgs g;
  std::memset((void*)&g, 0, sizeof(g));
  g.x = y; g.b1 = b1; g.b2 = b2; g.skip = x; g.gate = gate; g.gout = gout; g.gx = gz; g.gskip = gin;
  g.w1 = w1; g.w2 = w2; g.x_pre = y_pre; g.lw = ws_lin;
  g.x_pre_is_grad = d->act == SC_ACT_GELU_DGRAD ? 1 : 0;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  g.n_wg = pmlp_bwd_wgs(d);
  const int act = d->act == SC_ACT_NONE ? 0 : 1;
  sc_stream_t st = (sc_stream_t)stream;
  if (pmlp_shape_id(d) == 111) launch_pblock_bwd<1, 1, 1>(g, (float*)workspace, act, gw1, gb1, gw2, gb2, ggate, st);
  else launch_pblock_bwd<2, 1, 2>(g, (float*)workspace, act, gw1, gb1, gw2, gb2, ggate, st);
  return sc_check_launch("k_pmlp_bwd<LIN>");
}

// ---- 1 x 1 linear map (the block's linear skip)
static int plin_shape_id(const sc_plin_desc* d) {
  if (d->c_in % 32 || d->c_out % 32) return 0;
  return (int)(d->c_in / 32) * 10 + (int)(d->c_out / 32);
}
static int plin_bwd_wgs(const sc_plin_desc* d) {
  static const int env = [] { const char* e = SC_DIAG_ENV("SC_PLIN_BWD_WGS"); return e ? std::atoi(e) : 0; }();   // A-B
  const int64_t cap = env > 0 ? env : 512;
  const int64_t wgs = (d->batch * (d->spatial / 32) + 3) / 4;
  return (int)(wgs < cap ? wgs : cap);
}

extern "C" int sc_pointwise_linear_forward(const sc_plin_desc* d, const float* x, const float* w, const float* bias,
                                           float* out, void* stream) {
  SC_CHECK_ARG(d, "null argument");
  if (d->batch <= 0 || d->spatial <= 0) return 0;
  SC_CHECK_ARG(x && w && out, "null argument");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise linear map: the spatial size must be a multiple of 32 below 2^28 points");
  PlinArgs g;
  g.x = x; g.w = w; g.bias = bias; g.gout = nullptr; g.addend = nullptr; g.out = out; g.partial = nullptr;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  const int64_t wgs = (g.n_tiles + 3) / 4;
  // persistent: exactly as many workgroups as the chip holds at once -- two per compute unit (the kernel's launch bound;
  // the weight table in LDS is 4 / 16 / 64 KB) -- instead of 2048: 0.274 -> 0.22 ms at the metric shape (session 2,
  // profiles/r03s2_plin_time.txt; a next-tile prefetch at three or four per unit measured slower: 0.237 / 0.266 ms)
  const int64_t lds = (int64_t)(d->c_in / 32) * 16 * (d->c_out / 32) * 64 * 4 + d->c_out * 4;
  int64_t per_cu = 160 * 1024 / lds;
  per_cu = per_cu > 2 ? 2 : (per_cu < 1 ? 1 : per_cu);
  static const int plin_wgs_env = [] { const char* e = SC_DIAG_ENV("SC_PLIN_FWD_WGS"); return e ? std::atoi(e) : 0; }();
  const int64_t resident = plin_wgs_env > 0 ? plin_wgs_env : per_cu * sc_cu_count();
  g.n_wg = (int)(wgs < resident ? wgs : resident);
  sc_stream_t st = (sc_stream_t)stream;
  const dim3 grid((unsigned)g.n_wg), block(256);
  switch (plin_shape_id(d)) {
    case 11: SC_LAUNCH((k_plin_fwd<1, 1>), grid, block, 0, st, g); break;
    case 22: SC_LAUNCH((k_plin_fwd<2, 2>), grid, block, 0, st, g); break;
    case 44: SC_LAUNCH((k_plin_fwd<4, 4>), grid, block, 0, st, g); break;
    default: return sc_fail("sc_engine: pointwise linear map: c_in = c_out must be 32, 64 or 128");
  }
  return sc_check_launch("k_plin_fwd");
}

extern "C" size_t sc_pointwise_linear_workspace_bytes(const sc_plin_desc* d) {
  if (!d || d->batch <= 0 || d->spatial <= 0) return 0;
  const int id = plin_shape_id(d);
  if (id != 11 && id != 22) return 0;
  const size_t np = (size_t)d->c_out * d->c_in + d->c_out;
  return (size_t)(plin_bwd_wgs(d) + SC_PMLP_RED_GROUPS) * np * sizeof(float) + 256;
}

template <int CI, int CO>
static void launch_plin_bwd(PlinArgs g, float* ws, float* gw, float* gb, sc_stream_t st) {
  constexpr int NPW = 32 * CO * 32 * CI, NP = NPW + 32 * CO;
  g.partial = ws;
  float* stage = ws + (size_t)g.n_wg * NP;
  SC_LAUNCH((k_plin_bwd<CI, CO>), dim3((unsigned)g.n_wg), dim3(256), 0, st, g);
  const unsigned nb = (unsigned)((NP + 255) / 256);
  const int groups = g.n_wg < SC_PMLP_RED_GROUPS ? g.n_wg : SC_PMLP_RED_GROUPS;
  SC_LAUNCH(k_pmlp_reduce1, dim3(nb, (unsigned)groups), dim3(256), 0, st, (const float*)g.partial, g.n_wg, groups, (int)NP, stage);
  SC_LAUNCH(k_plin_reduce, dim3(nb), dim3(256), 0, st, (const float*)stage, groups, (int)NP, (int)NPW, gw, gb);
}

extern "C" int sc_pointwise_linear_backward(const sc_plin_desc* d, const float* x, const float* w, const float* gout,
                                            const float* gx_addend, float* gx, float* gw, float* gbias, void* workspace,
                                            void* stream) {
  SC_CHECK_ARG(d, "null argument");
  SC_CHECK_ARG(d->batch > 0 && d->spatial > 0, "pointwise linear map backward: empty input");
  SC_CHECK_ARG(x && w && gout && gx && gw && workspace, "null argument");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise linear map: the spatial size must be a multiple of 32 below 2^28 points");
  PlinArgs g;
  g.x = x; g.w = w; g.bias = nullptr; g.gout = gout; g.addend = gx_addend; g.out = gx; g.partial = nullptr;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  g.n_wg = plin_bwd_wgs(d);
  sc_stream_t st = (sc_stream_t)stream;
  switch (plin_shape_id(d)) {
    case 11: launch_plin_bwd<1, 1>(g, (float*)workspace, gw, gbias, st); break;
    case 22: launch_plin_bwd<2, 2>(g, (float*)workspace, gw, gbias, st); break;
    default: return sc_fail("sc_engine: pointwise linear map backward: c_in = c_out must be 32 or 64");
  }
  return sc_check_launch("k_plin_bwd");
}

// ---- 1 x 1 linear maps with the block's pointwise operations in their load / store paths (sc_kernels_plinx.h, round 6):
//      any channel counts in {32, 64, 128}, the two-pass form of a ChannelMLP whose channel counts have no one-pass kernel
static int plinx_ok(const sc_plinx_desc* d) {
  auto okc = [](int64_t c) { return c == 32 || c == 64 || c == 128; };
  return d && okc(d->c_in) && okc(d->c_out);
}
static int plinx_omn(int ci, int co) { const int cap = 8 / ci; return co < cap ? co : cap; }   // accumulator tiles <= 8
static int plinx_bwd_wgs(const sc_plinx_desc* d, bool lean = false) {
  const int ci = (int)(d->c_in / 32), co = (int)(d->c_out / 32);
  const int64_t wgs = (d->batch * (d->spatial / 32) + 3) / 4;
  const int64_t cap = (int64_t)SC_PLX_BWD_OCC(ci, co, plinx_omn(ci, co), lean) * sc_cu_count();     // the kernel's launch bound
  return (int)(wgs < cap ? wgs : cap);
}

template <int CI, int CO>
static void launch_plinx_fwd(const PlinxArgs& g, sc_stream_t st) {
  SC_LAUNCH((k_plinx_fwd<CI, CO>), dim3((unsigned)g.n_wg), dim3(256), 0, st, g);
}

extern "C" int sc_pointwise_linear_forward_ex(const sc_plinx_desc* d, const float* x, const float* w, const float* bias,
                                              const float* skip, const float* gate, float* out, float* pre_out,
                                              void* stream) {
  SC_CHECK_ARG(d, "null argument");
  if (d->batch <= 0 || d->spatial <= 0) return 0;
  SC_CHECK_ARG(x && w && out, "null argument");
  SC_CHECK_ARG(plinx_ok(d), "pointwise linear map: channel counts must be 32, 64 or 128");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise linear map: the spatial size must be a multiple of 32 below 2^28 points");
  SC_CHECK_ARG((skip == nullptr) == (gate == nullptr), "skip and gate come together");
  SC_CHECK_ARG((d->flags & ~(SC_PLX_XACT | SC_PLX_ACT)) == 0, "forward flags: SC_PLX_XACT | SC_PLX_ACT");
  PlinxArgs g;
  std::memset((void*)&g, 0, sizeof(g));
  g.x = x; g.w = w; g.bias = bias; g.skip = skip; g.gate = gate; g.out = out; g.pre_out = pre_out; g.flags = d->flags;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  const int ci = (int)(d->c_in / 32), co = (int)(d->c_out / 32);
  const int64_t wgs = (g.n_tiles + 3) / 4, resident = (int64_t)2 * sc_cu_count();
  g.n_wg = (int)(wgs < resident ? wgs : resident);
  sc_stream_t st = (sc_stream_t)stream;
  switch (ci * 10 + co) {
    case 11: launch_plinx_fwd<1, 1>(g, st); break;
    case 12: launch_plinx_fwd<1, 2>(g, st); break;
    case 14: launch_plinx_fwd<1, 4>(g, st); break;
    case 21: launch_plinx_fwd<2, 1>(g, st); break;
    case 22: launch_plinx_fwd<2, 2>(g, st); break;
    case 24: launch_plinx_fwd<2, 4>(g, st); break;
    case 41: launch_plinx_fwd<4, 1>(g, st); break;
    case 42: launch_plinx_fwd<4, 2>(g, st); break;
    default: launch_plinx_fwd<4, 4>(g, st); break;
  }
  return sc_check_launch("k_plinx_fwd");
}

extern "C" size_t sc_pointwise_linear_workspace_bytes_ex(const sc_plinx_desc* d) {
  if (!plinx_ok(d) || d->batch <= 0 || d->spatial <= 0) return 0;
  const int omn = plinx_omn((int)(d->c_in / 32), (int)(d->c_out / 32));
  const size_t np = (size_t)omn * 32 * d->c_in + 2 * (size_t)omn * 32;
  return (size_t)(plinx_bwd_wgs(d, true) + SC_PMLP_RED_GROUPS) * np * sizeof(float) + 256;    // (the larger of the two grids)
}

template <int CI, int CO, int OMN>
static void launch_plinx_bwd(PlinxArgs g, float* ws, bool want_gx, float* gw, float* gb, float* ggate, sc_stream_t st,
                             bool lean) {
  typedef PlinxDims<CI, OMN> D;
  g.partial = ws;
  float* stage = ws + (size_t)g.n_wg * D::NP;
  const unsigned nb = (unsigned)((D::NP + 255) / 256);
  const int groups = g.n_wg < SC_PMLP_RED_GROUPS ? g.n_wg : SC_PMLP_RED_GROUPS;
  for (int om0 = 0; om0 < CO; om0 += OMN) {                 // (stream-ordered: the launches share the partial buffer)
    g.do_gx = (want_gx && om0 == 0) ? 1 : 0;
    if (lean) SC_LAUNCH((k_plinx_bwd<CI, CO, OMN, true>), dim3((unsigned)g.n_wg), dim3(256), 0, st, g, om0);
    else SC_LAUNCH((k_plinx_bwd<CI, CO, OMN>), dim3((unsigned)g.n_wg), dim3(256), 0, st, g, om0);
    SC_LAUNCH(k_pmlp_reduce1, dim3(nb, (unsigned)groups), dim3(256), 0, st, (const float*)g.partial, g.n_wg, groups, (int)D::NP,
              stage);
    SC_LAUNCH(k_plinx_reduce, dim3(nb), dim3(256), 0, st, (const float*)stage, groups, (int)D::NP, (int)D::oB, (int)D::oG,
              (int)D::C_IN, 32 * om0, gw, gb, ggate);
  }
}

extern "C" int sc_pointwise_linear_backward_ex(const sc_plinx_desc* d, const float* x, const float* w, const float* gout,
                                               const float* pre, const float* xg, const float* skip, const float* gate,
                                               const float* gx_addend, float* gx, float* gw, float* gbias, float* gskip,
                                               float* ggate, void* workspace, void* stream) {
  SC_CHECK_ARG(d, "null argument");
  SC_CHECK_ARG(d->batch > 0 && d->spatial > 0, "pointwise linear map backward: empty input");
  SC_CHECK_ARG(x && w && gout && gw && workspace, "null argument");
  SC_CHECK_ARG(plinx_ok(d), "pointwise linear map: channel counts must be 32, 64 or 128");
  SC_CHECK_ARG(d->spatial % 32 == 0 && d->spatial < SC_PW_MAX_SPATIAL, "pointwise linear map: the spatial size must be a multiple of 32 below 2^28 points");
  SC_CHECK_ARG((skip == nullptr) == (gate == nullptr), "skip and gate come together");
  SC_CHECK_ARG(!gate || ggate, "a gated forward needs ggate");
  SC_CHECK_ARG(!gate || !gx || gskip, "a gated forward needs gskip beside gx");
  SC_CHECK_ARG((d->flags & ~(SC_PLX_XACT | SC_PLX_PRO | SC_PLX_XGRAD)) == 0, "backward flags: SC_PLX_XACT | SC_PLX_PRO | SC_PLX_XGRAD");
  SC_CHECK_ARG(!(d->flags & SC_PLX_PRO) || pre, "SC_PLX_PRO needs the pre-activation");
  SC_CHECK_ARG(!(d->flags & SC_PLX_XGRAD) || xg, "SC_PLX_XGRAD needs xg");
  SC_CHECK_ARG(gx || !gx_addend, "an addend without gx");
  PlinxArgs g;
  std::memset((void*)&g, 0, sizeof(g));
  g.x = x; g.w = w; g.gout = gout; g.pre = pre; g.xg = xg; g.skip = skip; g.gate = gate; g.addend = gx_addend;
  g.out = gx; g.gskip = gskip; g.flags = d->flags;
  g.spatial = d->spatial;
  g.tiles_per_sample = (int)(d->spatial / 32);
  g.n_tiles = d->batch * g.tiles_per_sample;
  const bool want_gx = gx != nullptr;
  const bool lean = !want_gx && d->flags == 0 && !gate;    // the weight / bias gradient alone: k_plinx_bwd<.., LEAN>
  g.n_wg = plinx_bwd_wgs(d, lean);
  sc_stream_t st = (sc_stream_t)stream;
  float* ws = (float*)workspace;
  switch ((int)(d->c_in / 32) * 10 + (int)(d->c_out / 32)) {
    case 11: launch_plinx_bwd<1, 1, 1>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 12: launch_plinx_bwd<1, 2, 2>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 14: launch_plinx_bwd<1, 4, 4>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 21: launch_plinx_bwd<2, 1, 1>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 22: launch_plinx_bwd<2, 2, 2>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 24: launch_plinx_bwd<2, 4, 4>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 41: launch_plinx_bwd<4, 1, 1>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    case 42: launch_plinx_bwd<4, 2, 2>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
    default: launch_plinx_bwd<4, 4, 2>(g, ws, want_gx, gw, gbias, ggate, st, lean); break;
  }
  return sc_check_launch("k_plinx_bwd");
}

// ---- Tucker mode factors (sc_kernels_tucker.h)
// matrix-core form (round 3) unless SC_TK_VALU is set (environment, A-B against the round-2 VALU kernels)
// (also VALU where the padded, conflict-free LDS layout of the matrix-core kernels does not fit: every extent near 64)
static bool tucker_use_mx(const sc_tucker_desc* d) {
  static const bool valu = SC_DIAG_ENV("SC_TK_VALU") != nullptr;
  return !valu && (size_t)tkm_layout((int)d->rx, (int)d->ry, (int)d->mx, (int)d->my, true).total * sizeof(cf32) <= 150 * 1024;
}
static size_t tucker_lds_bytes(const sc_tucker_desc* d, bool bwd) {
  if (tucker_use_mx(d)) return (size_t)tkm_layout((int)d->rx, (int)d->ry, (int)d->mx, (int)d->my, bwd).total * sizeof(cf32);
  size_t c = (size_t)d->mx * d->rx + (size_t)d->my * d->ry + (size_t)d->rx * d->ry + (size_t)d->rx * d->my;
  if (bwd) c += (size_t)d->mx * d->my + (size_t)d->rx * d->my;
  return c * sizeof(cf32);
}
static void tucker_invs(TuckerModesArgs& g) {
  const auto inv = [](int n) { return (uint32_t)((((uint64_t)1 << 32) + (uint64_t)n - 1) / (uint64_t)n); };
  g.inv_rx = inv(g.Rx); g.inv_ry = inv(g.Ry); g.inv_my = inv(g.My);
}
static int tucker_wgs(const sc_tucker_desc* d) {
  // workgroups of the two mode-factor kernels (each walks slices wg, wg + n, ...); SC_TK_WGS (environment, A-B)
  static const int env = [] { const char* e = SC_DIAG_ENV("SC_TK_WGS"); return e ? std::atoi(e) : 0; }();
  const int64_t cap = env > 0 ? env : 512;
  return (int)(d->fg < cap ? d->fg : cap);
}

extern "C" int sc_tucker_modes_supported(const sc_tucker_desc* d) {
  if (!d || d->fg <= 0 || d->rx <= 0 || d->ry <= 0 || d->mx <= 0 || d->my <= 0) return 0;
  if (d->mx > 64 || d->my > 64 || d->rx > 64 || d->ry > 64 || d->my * d->ry > 256 * SC_TK_UY_PER_THREAD) return 0;
  return tucker_lds_bytes(d, true) <= 150 * 1024 ? 1 : 0;
}

template <typename K>
static int tucker_launch(K kernel, const TuckerModesArgs& g, size_t lds, sc_stream_t st, const char* what) {
#ifndef SC_EMU
  if (lds > 64 * 1024)
    SC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  SC_LAUNCH(kernel, dim3((unsigned)g.n_wg), dim3(256), lds, st, g);
  return sc_check_launch(what);
}

extern "C" int sc_tucker_modes_forward(const sc_tucker_desc* d, const float* core, const float* ux, const float* uy,
                                       float* t, void* stream) {
  SC_CHECK_ARG(d && core && ux && uy && t, "null argument");
  SC_CHECK_ARG(sc_tucker_modes_supported(d), "Tucker mode factors: sizes outside the kernel's limits");
  TuckerModesArgs g;
  g.core = (const cf32*)core; g.ux = (const cf32*)ux; g.uy = (const cf32*)uy; g.gt = nullptr; g.t = (cf32*)t; g.partial = nullptr;
  g.FG = (int)d->fg; g.Rx = (int)d->rx; g.Ry = (int)d->ry; g.Mx = (int)d->mx; g.My = (int)d->my; g.n_wg = tucker_wgs(d); g.abl = tucker_abl();
  tucker_invs(g);
  if (tucker_use_mx(d)) {
    const size_t lds = tucker_lds_bytes(d, false);
    sc_stream_t st = (sc_stream_t)stream;
    const bool few = d->rx * d->ry <= 4 * 256, y3 = d->my <= 48;
    if (few && y3) return tucker_launch(k_tucker_modes_fwd_mx<4, 3>, g, lds, st, "k_tucker_modes_fwd_mx");
    if (few) return tucker_launch(k_tucker_modes_fwd_mx<4, 4>, g, lds, st, "k_tucker_modes_fwd_mx");
    if (y3) return tucker_launch(k_tucker_modes_fwd_mx<16, 3>, g, lds, st, "k_tucker_modes_fwd_mx");
    return tucker_launch(k_tucker_modes_fwd_mx<16, 4>, g, lds, st, "k_tucker_modes_fwd_mx");
  }
  return tucker_launch(k_tucker_modes_fwd, g, tucker_lds_bytes(d, false), (sc_stream_t)stream, "k_tucker_modes_fwd");
}

extern "C" size_t sc_tucker_modes_workspace_bytes(const sc_tucker_desc* d) {
  if (!sc_tucker_modes_supported(d)) return 0;
  const size_t np = 2 * ((size_t)d->mx * d->rx + (size_t)d->my * d->ry);
  return (size_t)(tucker_wgs(d) + SC_PMLP_RED_GROUPS) * np * sizeof(float) + 256;
}

extern "C" int sc_tucker_modes_backward(const sc_tucker_desc* d, const float* core, const float* ux, const float* uy,
                                        const float* gt, float* gcore, float* gux, float* guy, void* workspace,
                                        void* stream) {
  SC_CHECK_ARG(d && core && ux && uy && gt && gcore && gux && guy && workspace, "null argument");
  SC_CHECK_ARG(sc_tucker_modes_supported(d), "Tucker mode factors: sizes outside the kernel's limits");
  TuckerModesArgs g;
  g.core = (const cf32*)core; g.ux = (const cf32*)ux; g.uy = (const cf32*)uy; g.gt = (const cf32*)gt; g.t = (cf32*)gcore;
  g.partial = (float*)workspace;
  g.FG = (int)d->fg; g.Rx = (int)d->rx; g.Ry = (int)d->ry; g.Mx = (int)d->mx; g.My = (int)d->my; g.n_wg = tucker_wgs(d); g.abl = tucker_abl();
  tucker_invs(g);
  sc_stream_t st = (sc_stream_t)stream;
  int rc;
  if (tucker_use_mx(d)) {
    // the smallest instantiation that holds the problem: (3, 9, 3, 3, 2) is ranks (36, ., 36, 19) on 64 x 33 kept modes
    const size_t lds = tucker_lds_bytes(d, true);
    if (d->rx * d->ry <= 3 * 256 && d->mx * d->my <= 9 * 256 && d->rx <= 48 && d->my <= 48 && d->ry <= 32)
      rc = tucker_launch(k_tucker_modes_bwd_mx<3, 9, 3, 3, 2>, g, lds, st, "k_tucker_modes_bwd_mx");
    else
      rc = tucker_launch(k_tucker_modes_bwd_mx<16, 16, 4, 4, 4>, g, lds, st, "k_tucker_modes_bwd_mx");
  } else
    rc = tucker_launch(k_tucker_modes_bwd, g, tucker_lds_bytes(d, true), st, "k_tucker_modes_bwd");
  if (rc) return rc;
  const int np = 2 * (g.Mx * g.Rx + g.My * g.Ry);
  if (tucker_use_mx(d)) {
    SC_LAUNCH(k_tucker_reduce, dim3((unsigned)((np / 2 + 15) / 16)), dim3(256), 0, st, (const cf32*)g.partial, g.n_wg, np / 2,
              g.Mx * g.Rx, (cf32*)gux, (cf32*)guy);
    return sc_check_launch("k_tucker_reduce");
  }
  float* stage = g.partial + (size_t)g.n_wg * np;
  const unsigned nb = (unsigned)((np + 255) / 256);
  const int groups = g.n_wg < SC_PMLP_RED_GROUPS ? g.n_wg : SC_PMLP_RED_GROUPS;
  SC_LAUNCH(k_pmlp_reduce1, dim3(nb, (unsigned)groups), dim3(256), 0, st, (const float*)g.partial, g.n_wg, groups, np, stage);
  SC_LAUNCH(k_tucker_scatter, dim3(nb), dim3(256), 0, st, (const float*)stage, groups, np, 2 * g.Mx * g.Rx, gux, guy);
  return sc_check_launch("k_tucker_scatter");
}

extern "C" int sc_round_f16(const float* in, float* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  SC_CHECK_ARG(in && out, "null argument");
  int64_t blocks = (n + SC_BLOCK - 1) / SC_BLOCK;
  if (blocks > 16384) blocks = 16384;
  SC_LAUNCH(k_round_f16, dim3((unsigned)blocks), dim3(SC_BLOCK), 0, (sc_stream_t)stream, in, out, n, blocks * SC_BLOCK);
  return sc_check_launch("k_round_f16");
}

extern "C" int sc_modegemm_path(const sc_modegemm_desc* d) {
  if (!d) return 0;
  if (fmx_bfac_eligible(d)) return 5;
  if (bfac_gemm_eligible(d)) return 4;
  if (sb_gemm_eligible(d, nullptr, nullptr, nullptr)) return 3;
  if (gemm8_eligible(d, nullptr, nullptr, nullptr)) return 2;
  return !(d->flags & SC_GEMM_FORCE_VALU) && mfma_gemm_eligible(d) ? 1 : 0;
}

extern "C" int sc_modegemm_uses_matrix_cores(const sc_modegemm_desc* d) {
  const int path = sc_modegemm_path(d);
  return path == 1 || path == 2;
}

// ------------------------------------------------------------------------------------------
// Sharded spectrum layout (mode-parallel layers): the truncated spectrum is written / read as the
// rank-major all-to-all buffer [block][image][rows][rest] in place of the plain [image][k1][rest] block
// ------------------------------------------------------------------------------------------
static bool native_shards(const sc_plan* p) { return p->fast && !(p->d.flags & SC_PLAN_FFT_GEN2); }
static int64_t shard_rest(const sc_plan* p) {
  int64_t r = 1;
  for (int d = 1; d < p->nd; ++d) r *= p->k[d];
  return r;
}
static size_t shard_ws_base(const sc_plan* p, int64_t n_images) {
  return (sc_plan_workspace_bytes(p, n_images) + 255) / 256 * 256;
}
static int check_shards(const sc_plan* p, const sc_spectrum_shards* sh, int64_t n_images) {
  SC_CHECK_ARG(p && sh, "null argument");
  SC_CHECK_ARG(!p->cplx, "sharded spectra: real-data plans");
  SC_CHECK_ARG(sh->n_blocks >= 1 && sh->rows >= 1 && sh->n_blocks * sh->rows >= p->k[0],
               "shards must cover the first kept dim: n_blocks * rows >= k1");
  // (whole blocks past the first kept dim are legal since round 5: k1 = 6 over 4 ranks is three blocks of 2 rows and an
  // empty one -- the rank that owns it holds zero rows of the weight and sees zeros on the wire)
  SC_CHECK_ARG(sh->block_stride >= n_images * sh->rows * shard_rest(p), "block_stride smaller than a block");
  SC_CHECK_ARG(sh->rows < ((int64_t)1 << 30), "rows too large");
  return 0;
}

extern "C" size_t sc_plan_workspace_bytes_sharded(const sc_plan* p, int64_t n_images) {
  if (!p) return 0;
  if (native_shards(p)) return sc_plan_workspace_bytes(p, n_images);
  return shard_ws_base(p, n_images) + (size_t)(n_images * p->modes) * sizeof(cf32);
}

template <bool TO_SHARDS>
static int run_spectrum_shard(const sc_plan* p, const sc_spectrum_shards* sh, const cf32* src, cf32* dst,
                              int64_t n_images, sc_stream_t st) {
  const int64_t total = n_images * sh->rows * sh->n_blocks * shard_rest(p);
  int64_t blocks = (total + SC_BLOCK - 1) / SC_BLOCK;
  if (blocks > 8192) blocks = 8192;
  SC_LAUNCH((k_spectrum_shard<TO_SHARDS>), dim3((unsigned)blocks), dim3(SC_BLOCK), 0, st, src, dst, n_images,
            (int64_t)p->k[0], shard_rest(p), (int64_t)sh->rows, (int64_t)sh->n_blocks, (int64_t)sh->block_stride,
            blocks * SC_BLOCK);
  return sc_check_launch("k_spectrum_shard");
}

extern "C" int sc_transform_forward_sharded(const sc_plan* p, int mode, const float* x, float* xhat,
                                            int64_t n_images, const sc_spectrum_shards* sh, void* workspace,
                                            void* stream) {
  if (n_images <= 0) return 0;
  if (int rc = check_shards(p, sh, n_images)) return rc;
  sc_stream_t st = (sc_stream_t)stream;
  if (native_shards(p) || ax_native_shards(p, n_images)) {
    // rows past k1 (k1 not a multiple of the block size, or fewer rows than blocks) are zeros on the wire: clear every
    // block that has some -- the one k1 ends in and the empty ones behind it
    for (int64_t b = p->k[0] / sh->rows; b < sh->n_blocks; ++b) {
      cf32* blk = (cf32*)xhat + b * sh->block_stride;
      if (hipMemsetAsync(blk, 0, (size_t)(n_images * sh->rows * shard_rest(p)) * sizeof(cf32), st) != hipSuccess)
        return sc_fail("sc_engine: hipMemsetAsync failed");
    }
    return transform_forward_impl(p, mode, x, xhat, n_images, workspace, stream, F3Shard{(int)sh->rows, sh->block_stride});
  }
  SC_CHECK_ARG(workspace, "workspace required (sc_plan_workspace_bytes_sharded)");
  cf32* stage = (cf32*)((unsigned char*)workspace + shard_ws_base(p, n_images));
  if (int rc = sc_transform_forward(p, mode, x, (float*)stage, n_images, workspace, stream)) return rc;
  return run_spectrum_shard<true>(p, sh, stage, (cf32*)xhat, n_images, st);
}

extern "C" int sc_transform_inverse_sharded(const sc_plan* p, int mode, const float* yhat, const float* bias,
                                            int64_t channels, float* y, int64_t n_images,
                                            const sc_spectrum_shards* sh, void* workspace, void* stream) {
  if (n_images <= 0) return 0;
  if (int rc = check_shards(p, sh, n_images)) return rc;
  if (native_shards(p) || ax_native_shards(p, n_images))
    return transform_inverse_impl(p, mode, yhat, bias, channels, nullptr, y, n_images, workspace, stream,
                                  F3Shard{(int)sh->rows, sh->block_stride});
  SC_CHECK_ARG(workspace, "workspace required (sc_plan_workspace_bytes_sharded)");
  cf32* stage = (cf32*)((unsigned char*)workspace + shard_ws_base(p, n_images));
  if (int rc = run_spectrum_shard<false>(p, sh, (const cf32*)yhat, stage, n_images, (sc_stream_t)stream)) return rc;
  return sc_transform_inverse(p, mode, (const float*)stage, bias, channels, y, n_images, workspace, stream);
}

// bias gradient off a SHARDED adjoint spectrum (the DC coefficient of every image sits in one block)
extern "C" int sc_bias_grad_sharded(const sc_plan* p, const float* ghat, int64_t batch, int64_t channels,
                                    const sc_spectrum_shards* sh, float* gbias, void* stream) {
  SC_CHECK_ARG(p && ghat && gbias, "null argument");
  SC_CHECK_ARG(p->dc_index >= 0, "the plan's frequency maps keep no zero-frequency coefficient");
  if (int rc = check_shards(p, sh, batch * channels)) return rc;
  if (channels <= 0) return 0;
  const int64_t rest = shard_rest(p);
  const int64_t row = p->dc_index / rest, col = p->dc_index - row * rest, blk = row / sh->rows;
  const cf32* base = (const cf32*)ghat + blk * sh->block_stride;
  SC_LAUNCH(k_bias_grad, dim3((unsigned)channels), dim3(SC_WAVE), 0, (sc_stream_t)stream, base, gbias, batch,
            channels, sh->rows * rest, (row - blk * sh->rows) * rest + col);
  return sc_check_launch("k_bias_grad");
}

extern "C" int sc_bias_grad(const sc_plan* p, const float* ghat, int64_t batch, int64_t channels,
                            float* gbias, void* stream) {
  SC_CHECK_ARG(p && ghat && gbias, "null argument");
  SC_CHECK_ARG(p->dc_index >= 0, "the plan's frequency maps keep no zero-frequency coefficient");
  if (channels <= 0) return 0;
  SC_LAUNCH(k_bias_grad, dim3((unsigned)channels), dim3(SC_WAVE), 0, (sc_stream_t)stream, (const cf32*)ghat,
            gbias, batch, channels, p->modes, p->dc_index);
  return sc_check_launch("k_bias_grad");
}

// ------------------------------------------------------------------------------------------
// fused AdamW step
// ------------------------------------------------------------------------------------------
extern "C" int sc_adamw_step(const sc_adamw_desc* d, float* param, const float* grad, float* exp_avg,
                             float* exp_avg_sq, int64_t n, int is_complex, void* stream) {
  SC_CHECK_ARG(d, "null argument");
  if (n <= 0) return 0;
  SC_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "null argument");
  SC_CHECK_ARG(d->step >= 1, "step counts from 1");
  AdamwArgs a;
  a.b1 = (float)d->beta1;
  a.b2 = (float)d->beta2;
  a.one_m_b1 = (float)(1.0 - d->beta1);
  a.one_m_b2 = (float)(1.0 - d->beta2);
  a.eps = (float)d->eps;
  double step_size = d->lr;
  if (d->correct_bias) {                      // adamw.py:176-179, in double like the python arithmetic
    const double bc1 = 1.0 - std::pow(d->beta1, (double)d->step);
    const double bc2 = 1.0 - std::pow(d->beta2, (double)d->step);
    step_size = step_size * std::sqrt(bc2) / bc1;
  }
  a.step_size = (float)step_size;
  a.decay = d->weight_decay > 0.0 ? (float)(d->lr * d->weight_decay) : 0.f;
  sc_stream_t st = (sc_stream_t)stream;
  const int64_t pairs = is_complex ? n : n / 2;
  if (pairs > 0) {
    int64_t blocks = (pairs + SC_BLOCK - 1) / SC_BLOCK;
    if (blocks > 8192) blocks = 8192;
    const int64_t stride = blocks * SC_BLOCK;
    if (is_complex) SC_LAUNCH((k_adamw<true>), dim3((unsigned)blocks), dim3(SC_BLOCK), 0, st, a, param, grad, exp_avg, exp_avg_sq, pairs, stride);
    else SC_LAUNCH((k_adamw<false>), dim3((unsigned)blocks), dim3(SC_BLOCK), 0, st, a, param, grad, exp_avg, exp_avg_sq, pairs, stride);
  }
  if (!is_complex && (n & 1))
    SC_LAUNCH(k_adamw_tail, dim3(1), dim3(SC_WAVE), 0, st, a, param, grad, exp_avg, exp_avg_sq, n - 1);
  return sc_check_launch("k_adamw");
}

// ------------------------------------------------------------------------------------------
// fused dense layer
// ------------------------------------------------------------------------------------------
static bool layer_full_block(const sc_plan* p, const sc_layer_desc* L) {
  for (int d = 0; d < p->nd; ++d)
    if (L->w_start[d] != 0 || L->w_extent[d] != p->k[d]) return false;
  return true;
}

static int layer_index_table(const sc_plan* cp, const sc_layer_desc* L, const int32_t** out) {
  sc_plan* p = const_cast<sc_plan*>(cp);
  if (layer_full_block(p, L)) {
    *out = nullptr;
    return 0;
  }
  IdxKey key;
  std::memset(&key, 0, sizeof(key));
  int64_t wtot = 1;
  for (int d = 0; d < p->nd; ++d) {
    key.ext[d] = L->w_extent[d];
    key.start[d] = L->w_start[d];
    SC_CHECK_ARG(L->w_start[d] >= 0 && L->w_start[d] + p->k[d] <= L->w_extent[d],
                 "weight sub-block outside the stored weight");
    wtot *= L->w_extent[d];
  }
  SC_CHECK_ARG(wtot < (int64_t)1 << 31, "weight slab too large for int32 index table");
  std::lock_guard<std::mutex> lock(p->idx_mu);
  auto it = p->idx_cache.find(key);
  if (it != p->idx_cache.end()) {
    *out = it->second;
    return 0;
  }
  std::vector<int32_t> h((size_t)p->modes);
  std::vector<int64_t> r(p->nd, 0);
  for (int64_t m = 0; m < p->modes; ++m) {
    int64_t off = 0;
    for (int d = 0; d < p->nd; ++d) off = off * L->w_extent[d] + (L->w_start[d] + r[d]);
    h[(size_t)m] = (int32_t)off;
    for (int d = p->nd - 1; d >= 0; --d) {
      if (++r[d] < p->k[d]) break;
      r[d] = 0;
    }
  }
  void* dev = nullptr;
  SC_CHECK_HIP(hipMalloc(&dev, h.size() * sizeof(int32_t)));
  SC_CHECK_HIP(hipMemcpy(dev, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  p->idx_cache[key] = (int32_t*)dev;
  *out = (int32_t*)dev;
  return 0;
}

extern "C" size_t sc_layer_workspace_bytes(const sc_plan* p, const sc_layer_desc* L) {
  if (!p || !L) return 0;
  const int64_t cmax = L->cin > L->cout ? L->cin : L->cout;
  const size_t tws = round_up((int64_t)sc_plan_workspace_bytes(p, (int64_t)L->batch * cmax), 256);
  const size_t spec = (size_t)L->batch * (size_t)(L->cin + L->cout) * (size_t)p->modes * sizeof(cf32);
  return tws + round_up((int64_t)spec, 256) + 512;
}

static int64_t weight_slab(const sc_plan* p, const sc_layer_desc* L) {
  int64_t w = 1;
  for (int d = 0; d < p->nd; ++d) w *= L->w_extent[d];
  return w;
}

extern "C" int sc_layer_forward(const sc_plan* p, const sc_layer_desc* L, const float* x, const float* w,
                                const float* bias, float* y, float* xhat_saved, void* workspace,
                                void* stream) {
  return sc_layer_forward_ex(p, L, x, w, bias, nullptr, y, xhat_saved, workspace, stream);
}

extern "C" int sc_layer_forward_ex(const sc_plan* p, const sc_layer_desc* L, const float* x, const float* w,
                                   const float* bias, const sc_epilogue* ep, float* y, float* xhat_saved,
                                   void* workspace, void* stream) {
  SC_CHECK_ARG(p && L, "null argument");
  SC_CHECK_ARG(!p->cplx, "complex-data plans: call the transform / contraction stages (no fused layer)");
  const int64_t B = L->batch, Ci = L->cin, Co = L->cout, Mk = p->modes;
  if (B == 0) return 0;                      // empty batch: nothing to do (pointers may be null)
  SC_CHECK_ARG(x && w && y && xhat_saved && workspace, "null argument");
  const int32_t* idx = nullptr;
  int rc = layer_index_table(p, L, &idx);
  if (rc) return rc;
  const int64_t Wm = weight_slab(p, L);
  const int64_t cmax = Ci > Co ? Ci : Co;
  char* ws = (char*)workspace;
  const size_t tws = round_up((int64_t)sc_plan_workspace_bytes(p, B * cmax), 256);
  float* yhat = (float*)(ws + tws);

  rc = sc_transform_forward(p, SC_FWD_SCALED, x, xhat_saved, B * Ci, ws, stream);
  if (rc) return rc;
  sc_modegemm_desc g;
  std::memset(&g, 0, sizeof(g));
  g.P = B; g.Q = Co; g.R = Ci; g.n_modes = Mk;
  g.a_sp = Ci * Mk; g.a_sr = Mk; g.a_sm = 1;
  g.b_sr = Co * Wm; g.b_sq = Wm; g.b_sm = 1; g.b_idx = idx;
  g.c_sp = Co * Mk; g.c_sq = Mk; g.c_sm = 1;
  g.flags = (p->d.flags & SC_PLAN_FORCE_GENERIC) ? SC_GEMM_FORCE_VALU : 0;
  rc = sc_modegemm(&g, xhat_saved, w, yhat, stream);
  if (rc) return rc;
  return sc_transform_inverse_ex(p, SC_INV_PADDED, yhat, bias, Co, ep, y, B * Co, ws, stream);
}

extern "C" int sc_layer_backward(const sc_plan* p, const sc_layer_desc* L, const float* gy,
                                 const float* xhat_saved, const float* w, float* gx, float* gw,
                                 float* gbias, void* workspace, void* stream) {
  return sc_layer_backward_ex(p, L, gy, xhat_saved, w, gx, gw, gbias, nullptr, workspace, stream);
}

extern "C" int sc_layer_backward_ex(const sc_plan* p, const sc_layer_desc* L, const float* gy,
                                    const float* xhat_saved, const float* w, float* gx, float* gw,
                                    float* gbias, const float* gx_addend, void* workspace, void* stream) {
  SC_CHECK_ARG(p && L, "null argument");
  SC_CHECK_ARG(!p->cplx, "complex-data plans: call the transform / contraction stages (no fused layer)");
  const int64_t B = L->batch, Ci = L->cin, Co = L->cout, Mk = p->modes;
  if (B == 0) {                              // empty batch: gradients of the parameters are zero
    if (gw) SC_CHECK_HIP(hipMemsetAsync(gw, 0, (size_t)Ci * Co * weight_slab(p, L) * sizeof(cf32), (sc_stream_t)stream));
    if (gbias) SC_CHECK_HIP(hipMemsetAsync(gbias, 0, (size_t)Co * sizeof(float), (sc_stream_t)stream));
    return 0;
  }
  SC_CHECK_ARG(gy && xhat_saved && w && workspace, "null argument");
  const int32_t* idx = nullptr;
  int rc = layer_index_table(p, L, &idx);
  if (rc) return rc;
  const int64_t Wm = weight_slab(p, L);
  const int64_t cmax = Ci > Co ? Ci : Co;
  char* ws = (char*)workspace;
  const size_t tws = round_up((int64_t)sc_plan_workspace_bytes(p, B * cmax), 256);
  float* ghat = (float*)(ws + tws);
  float* gxhat = ghat + 2 * B * Co * Mk;

  rc = sc_transform_forward(p, SC_FWD_ADJ_C2R, gy, ghat, B * Co, ws, stream);
  if (rc) return rc;
  // (running {gbias, gW} on a side stream beside {gXhat -> gx} was tried: 165 -> 145 us for the isolated pair,
  // nothing measurable in the step -- every kernel here fills the chip; profiles/r01_stream_overlap.txt.  What
  // does pay is ONE launch for {gbias, gW, gXhat}: k_modegemm_dma_bwd)
  const int32_t gflags = (p->d.flags & SC_PLAN_FORCE_GENERIC) ? SC_GEMM_FORCE_VALU : 0;
  sc_modegemm_desc dw, dx;
  // gW[i,o,m] = sum_b conj(xhat[b,i,m]) * ghat[b,o,m]
  std::memset(&dw, 0, sizeof(dw));
  dw.P = Ci; dw.Q = Co; dw.R = B; dw.n_modes = Mk;
  dw.a_sp = Mk; dw.a_sr = Ci * Mk; dw.a_sm = 1; dw.conj_a = 1;
  dw.b_sr = Co * Mk; dw.b_sq = Mk; dw.b_sm = 1;
  dw.c_sp = Co * Wm; dw.c_sq = Wm; dw.c_sm = 1; dw.c_idx = idx;
  dw.flags = gflags ? gflags : SC_GEMM_STREAM_C;
  // gxhat[b,i,m] = sum_o ghat[b,o,m] * conj(W[i,o,m])
  std::memset(&dx, 0, sizeof(dx));
  dx.P = B; dx.Q = Ci; dx.R = Co; dx.n_modes = Mk;
  dx.a_sp = Co * Mk; dx.a_sr = Mk; dx.a_sm = 1;
  dx.b_sr = Wm; dx.b_sq = Co * Wm; dx.b_sm = 1; dx.b_idx = idx; dx.conj_b = 1;
  dx.c_sp = Ci * Mk; dx.c_sq = Mk; dx.c_sm = 1;
  dx.flags = gflags;
  bool paired = false;
  if (gw && gx && !idx) {
    // small batch against a large weight (BASELINE configs[4]): ONE pass over W for both gradients
    rc = run_sb_bwd(&dw, (const cf32*)xhat_saved, (const cf32*)ghat, (cf32*)gw, &dx, (const cf32*)ghat, (const cf32*)w,
                    (cf32*)gxhat, (sc_stream_t)stream);
    if (rc > 0) return rc;
    if (rc == 0) {
      paired = true;
      if (gbias) {
        rc = sc_bias_grad(p, ghat, B, Co, gbias, stream);
        if (rc) return rc;
      }
    }
  }
  if (!paired && gw && gx && (!gbias || p->dc_index >= 0)) {
    Gemm8Bias gb;
    gb.ghat = gbias ? (const cf32*)ghat : nullptr;
    gb.gbias = gbias; gb.batch = B; gb.channels = Co; gb.modes_per_image = Mk; gb.dc = p->dc_index;
    rc = run_gemm8_bwd(&dw, (const cf32*)xhat_saved, (const cf32*)ghat, (cf32*)gw,
                       &dx, (const cf32*)ghat, (const cf32*)w, (cf32*)gxhat, gb, (sc_stream_t)stream);
    if (rc > 0) return rc;
    paired = rc == 0;
  }
  if (!paired) {
    if (gbias) {
      rc = sc_bias_grad(p, ghat, B, Co, gbias, stream);
      if (rc) return rc;
    }
    if (gw) {
      rc = sc_modegemm(&dw, xhat_saved, ghat, gw, stream);
      if (rc) return rc;
    }
    if (gx) {
      rc = sc_modegemm(&dx, ghat, w, gxhat, stream);
      if (rc) return rc;
    }
  }
  if (gx) {
    sc_epilogue ep;
    ep.skip = gx_addend; ep.preact = nullptr; ep.act = SC_ACT_NONE; ep.reserved = 0;
    rc = sc_transform_inverse_ex(p, SC_INV_ADJ_R2C, gxhat, nullptr, Ci, gx_addend ? &ep : nullptr, gx, B * Ci, ws, stream);
    if (rc) return rc;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
extern "C" const char* sc_last_error(void) { return g_last_error.c_str(); }

extern "C" const char* sc_version(void) {
#ifdef SC_EMU
  return "sc_engine 0.1 (host emulation build -- tests only)";
#else
  return "sc_engine 0.1 (gfx950)";
#endif
}

extern "C" const char* sc_plan_kernel_name(const sc_plan* p, int which) {
  if (!p) return "";
  if (p->fast) {
    if (p->d.flags & SC_PLAN_FFT_GEN2) return fft2d_kernel_name(which);
    return which == 0 ? (p->fft2d.tabF ? "k_fft2d_fwd_mx" : "k_fft2d_fwd3") : (p->fft2d.tabG ? "k_fft2d_inv_mx" : "k_fft2d_inv3");
  }
  if (p->cplx) return "k_axis_pass";
  if (p->f2p) return which == 0 ? "k_f2p_r2c" : "k_f2p_c2r";
  if (p->pl128) return which == 0 ? "k_pl128_fwd" : "k_pl128_inv";
  if (p->pl64) return which == 0 ? "k_pl64_fwd" : "k_pl64_inv";
  if (p->mdft) {
    if (which == 0 && plane_fwd_ok(p, 0)) return "k_mdft_r2c_lds<plane>";
    if (which == 1 && plane_inv_ok(p, 0)) return "k_mdft_c2r_lds<plane>";
    const int Nl = (int)p->n[p->nd - 1];                      // (16-byte aligned tensors assumed, lines = 1)
    if (which == 0)
      return p->l_r2c[0] ? "k_mdft_r2c_lds"
                         : (p->s_r2c[0] ? "k_mdft_r2c_stage" : (p->m_r2c[0] ? "k_mdft_r2c" : "k_last_r2c"));
    return p->l_c2r[0] ? "k_mdft_c2r_lds"
                       : (p->s_c2r[0] ? (c2r_span_ok(p, Nl, nullptr, 1) ? "k_mdft_c2r_span" : "k_mdft_c2r_stage") : "k_mdft_c2r");
  }
  return which == 0 ? "k_last_r2c" : "k_last_c2r";
}
