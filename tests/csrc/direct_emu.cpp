// direct_emu.cpp - TEST INFRASTRUCTURE ONLY.  Runs qmps_amd/csrc/qmps_direct_core.h (the mathematics of the
// D = 4 direct-solve kernel) on the CPU with the four lanes of a DPP quad emulated in lock-step, so that the
// CPU test-suite can compare the very source the GPU kernel is built from with the oracle.  Nothing in the
// product loads this file; the product path has no CPU fallback.
//
// Build: g++ -O2 -std=c++17 -shared -fPIC -I qmps_amd/csrc tests/csrc/direct_emu.cpp -o tests/csrc/libdirect_emu.so
#include <cmath>
#include <cstdint>

#include "qmps_direct_core.h"

namespace {

struct Q4 {
  double v[4];
};
struct P4 {
  bool v[4];
};
inline Q4 operator+(Q4 a, Q4 b) { return {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]}}; }
inline Q4 operator-(Q4 a, Q4 b) { return {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]}}; }
inline Q4 operator*(Q4 a, Q4 b) { return {{a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]}}; }
inline Q4 operator-(Q4 a) { return {{-a.v[0], -a.v[1], -a.v[2], -a.v[3]}}; }

struct HostOps {
  using V = Q4;
  using P = P4;
  const double* A;   // [2][4][4] complex, interleaved
  P q_eq(int i) const { return {{i == 0, i == 1, i == 2, i == 3}}; }
  P q_gt(int i) const { return {{0 > i, 1 > i, 2 > i, 3 > i}}; }
  template <int L>
  static V bcast(V a) {
    return {{a.v[L], a.v[L], a.v[L], a.v[L]}};
  }
  template <int L, int K>
  void row_bcast(const V (&row)[16], V (&out)[16]) const {
    for (int j = K; j < 16; ++j) out[j] = bcast<L>(row[j]);
  }
  static V qsum(V a) {
    // same association as the DPP butterfly: (x + partner[1,0,3,2]) + pair[2,3,0,1]
    const double s01 = a.v[0] + a.v[1], s23 = a.v[2] + a.v[3];
    const double t = s01 + s23;
    return {{t, t, t, t}};
  }
  static V qmax(V a) {
    const double t = std::fmax(std::fmax(a.v[0], a.v[1]), std::fmax(a.v[2], a.v[3]));
    return {{t, t, t, t}};
  }
  static V vabs(V a) { return {{std::fabs(a.v[0]), std::fabs(a.v[1]), std::fabs(a.v[2]), std::fabs(a.v[3])}}; }
  static V vmax(V a, V b) { return {{std::fmax(a.v[0], b.v[0]), std::fmax(a.v[1], b.v[1]), std::fmax(a.v[2], b.v[2]), std::fmax(a.v[3], b.v[3])}}; }
  static V sel(P p, V a, V b) { return {{p.v[0] ? a.v[0] : b.v[0], p.v[1] ? a.v[1] : b.v[1], p.v[2] ? a.v[2] : b.v[2], p.v[3] ? a.v[3] : b.v[3]}}; }
  static V splat(double x) { return {{x, x, x, x}}; }
  static V fma(V a, V b, V c) { return {{std::fma(a.v[0], b.v[0], c.v[0]), std::fma(a.v[1], b.v[1], c.v[1]), std::fma(a.v[2], b.v[2], c.v[2]), std::fma(a.v[3], b.v[3], c.v[3])}}; }
  static V rcp(V a) { return {{1.0 / a.v[0], 1.0 / a.v[1], 1.0 / a.v[2], 1.0 / a.v[3]}}; }
  static P lt(V a, V b) { return {{a.v[0] < b.v[0], a.v[1] < b.v[1], a.v[2] < b.v[2], a.v[3] < b.v[3]}}; }
  static P gt0(V a) { return {{a.v[0] > 0, a.v[1] > 0, a.v[2] > 0, a.v[3] > 0}}; }
  static P p_and(P a, P b) { return {{a.v[0] && b.v[0], a.v[1] && b.v[1], a.v[2] && b.v[2], a.v[3] && b.v[3]}}; }
  static P p_not(P a) { return {{!a.v[0], !a.v[1], !a.v[2], !a.v[3]}}; }
  static bool any(P a) { return a.v[0] || a.v[1] || a.v[2] || a.v[3]; }
  void own(int s, int j, V& re, V& im) const {
    for (int q = 0; q < 4; ++q) {
      re.v[q] = A[2 * ((s * 4 + q) * 4 + j)];
      im.v[q] = A[2 * ((s * 4 + q) * 4 + j) + 1];
    }
  }
  void uni(int s, int i, int j, V& re, V& im) const {
    re = splat(A[2 * ((s * 4 + i) * 4 + j)]);
    im = splat(A[2 * ((s * 4 + i) * 4 + j) + 1]);
  }
};

}  // namespace

// One batch through the same sequence of core calls as energy_direct_d4_kernel.
// A [B][2][4][4] c128, h [nt][4][4] c128; outputs E [B][nt], r [B][4][4] c128, rho [B][4][4] c128, iters, status [B].
extern "C" int direct_emu_d4(long B, const double* A, const double* h, int nt, int max_iter, double tol, double* E,
                             double* r_out, double* rho_out, int32_t* iters, int32_t* status, double* resid, double* E_lean) {
  using Core = qmps::DirectD4<HostOps>;
  using V = Q4;
  for (long b = 0; b < B; ++b) {
    HostOps o{A + b * 64};
    V Rc[4][16], x[4], y[4], us[16];
    Core::build(o, Rc);
    V pivmax;
    Core::solve(o, Rc, x, pivmax);
    Core::normalise(o, x);
    Core::gather(x, us);
    const V d2 = Core::power_step(o, x, us, y);
    const double tol2 = tol * tol;
    P4 ok = HostOps::p_and(HostOps::lt(d2, HostOps::splat(tol2)), HostOps::lt(pivmax, HostOps::splat(Core::kMaxInversePivot)));
    V steps = HostOps::splat(1.0);
    int st = 0;
    if (resid) resid[b] = std::sqrt(d2.v[0]);
    if (!ok.v[0]) {
      V sq = HostOps::splat(0.0);
      Core::build(o, Rc);
      const P4 left = Core::squaring(o, Rc, HostOps::p_not(ok), max_iter - 1, tol2, x, sq);
      steps = HostOps::splat(1.0) + sq;
      st = left.v[0] ? 1 : 0;
      if (st == 1 && steps.v[0] + 1.0 <= (double)max_iter) {
        // the budget ended the chain between two powers of two: the plain method's own test on the last iterate (as the kernel, round 5)
        V yb[4];
        Core::gather(x, us);
        const V d2b = Core::power_step(o, x, us, yb);
        if (d2b.v[0] < tol2) {
          for (int l = 0; l < 4; ++l) x[l] = yb[l];
          steps = steps + HostOps::splat(1.0);
          st = 0;
        }
      }
    }
    V pre[4][4], pim[4][4];
    Core::gather(x, us);
    const P4 pd = Core::density(o, us, pre, pim);
    if (st == 0 && !pd.v[0]) st = 2;
    for (int t = 0; t < nt; ++t) E[b * nt + t] = HostOps::qsum(Core::energy(h + 32 * t, pre, pim)).v[0];
    // the density-matrix-free route of the energy-only kernel, from the same environment
    if (E_lean) {
      V bre[4][4], bim[4][4];
      Core::b_rows(o, bre, bim);
      for (int t = 0; t < nt; ++t) E_lean[b * nt + t] = HostOps::qsum(Core::energy_lean(bre, bim, us, h + 32 * t)).v[0];
    }
    iters[b] = (int32_t)steps.v[0];
    status[b] = st;
    if (r_out)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const int lo = i < j ? i : j, hi = i < j ? j : i;
          const double re = us[4 * lo + hi].v[0], im = i == j ? 0.0 : us[4 * hi + lo].v[0];
          r_out[2 * (b * 16 + i * 4 + j)] = re;
          r_out[2 * (b * 16 + i * 4 + j) + 1] = i <= j ? im : -im;
        }
    if (rho_out)
      for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) {
          const int lo = t < s ? t : s, hi = t < s ? s : t;
          const double re = HostOps::qsum(pre[lo][hi]).v[0], im = t == s ? 0.0 : HostOps::qsum(pim[lo][hi]).v[0];
          rho_out[2 * (b * 16 + t * 4 + s)] = re;
          rho_out[2 * (b * 16 + t * 4 + s) + 1] = t <= s ? im : -im;
        }
  }
  return 0;
}
