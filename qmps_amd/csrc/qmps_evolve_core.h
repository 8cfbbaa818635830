// qmps_evolve_core.h - the device-resident BFGS time evolution of ONE trajectory (gfx950 only): the optimiser loop shared by
// evolve_bfgs_d2_kernel (a wave per trajectory, candidates = lanes: qmps_evolve_d2.hip) and evolve_bfgs_d4_kernel (a workgroup per
// trajectory, candidates = waves: qmps_evolve_d4.hip).
// Reference: qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 (`minimize(obj, params, (A_, WW))` per time step: scipy BFGS
// with finite-difference gradients).  The iteration is tools.batched_bfgs / qmps_evolve_bfgs for one trajectory, decision for
// decision: speculative full step with its gradient, Armijo ladder, first-accepted / best rung, rank-two update with the curvature
// guard, steepest-descent restart; the host driver's floating-point expressions of the optimiser algebra are reproduced with
// explicitly rounded operations (no FMA contraction).
//
// The caller supplies
//   evaluate(coef, n_ladder)  one evaluation pass: candidates 0 .. 2P of z = x + coef d (0 the point itself, 1 + k / 1 + P + k = +- h e_k)
//                             and, behind them, n_ladder backtracking points x + alphas[r + 1] d; leaves -sqrt|eta| in L.F[c] and
//                             status == OK in L.OK[c] for every candidate c, visible to all threads (it ends with sync())
//   build_reference()         the step's reference tensor from L.X (ends with sync())
//   sync()                    barrier over the threads that share the trajectory
//   vl                        this thread's vector index (0 .. P - 1: it owns element vl of x, g, d, s and row vl of H) or -1
//   writer                    true in ONE thread: it stores the per-step records
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"

namespace qmps {

constexpr int kEvolvePMax = 16;     // parameters per trajectory

struct BfgsLds {
  double *X, *G, *D, *S, *Gn, *Hy;
  double (*H)[kEvolvePMax + 1];
  double* F;
  int* OK;
  double* AR = nullptr;     // nullable [2]: (f, slope) of the iteration, published before a ladder pass of its own (an evaluator that
                            // stops at the first rung passing the Armijo test needs them: qmps_evolve_d16.hip)
};

namespace evolve_detail {
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }
}  // namespace evolve_detail

template <class Eval, class RefBuild, class Sync>
__device__ __forceinline__ void bfgs_time_evolution(const EvolveD2Args& p, int64_t t, int vl, bool writer, const BfgsLds& L, Eval evaluate,
                                                    RefBuild build_reference, Sync sync, bool ladder_in_pass) {
  using namespace evolve_detail;
  const int P = p.P, NA = p.NA, G = NA - 1, G1 = 2 * P + 1;
  const double NaN = __builtin_nan("");
  // objective and gradient of the last pass: f, and g into `gout` (LDS)
  auto read_fg = [&](double& f, double* gout) {
    f = L.OK[0] ? L.F[0] : NaN;
    if (vl >= 0) gout[vl] = (L.OK[1 + vl] && L.OK[1 + P + vl]) ? (L.F[1 + vl] - L.F[1 + P + vl]) / (2.0 * p.h) : NaN;
    sync();
  };
  auto gmax_at_least = [&](const double* gt, double bound) {      // np.abs(g).max() >= bound; false with any NaN
    double m = 0.0;
    for (int k = 0; k < P; ++k) {
      const double v = gt[k];
      if (v != v) return false;
      const double a = fabs(v);
      m = a > m ? a : m;
    }
    return m >= bound;
  };
  auto set_identity = [&]() {
    if (vl >= 0)
      for (int b = 0; b < P; ++b) L.H[vl][b] = vl == b ? 1.0 : 0.0;
    sync();
  };
  if (vl >= 0) {
    L.X[vl] = p.params[t * P + vl];
    L.D[vl] = 0.0;
    if (p.carry_in && p.hinv != nullptr)
      for (int b = 0; b < P; ++b) L.H[vl][b] = p.hinv[(t * P + vl) * P + b];
  }
  sync();
  if (!(p.carry_in && p.hinv != nullptr)) set_identity();

  for (int step = 0; step < p.n_steps; ++step) {
    build_reference();
    if (!(p.carry && (step > 0 || p.carry_in))) set_identity();
    double f;
    evaluate(0.0, 0);
    read_fg(f, L.G);
    if (writer) p.f_hist[((int64_t)step * 2 + 0) * p.T + t] = f;
    bool active = gmax_at_least(L.G, p.gtol);
    int nit = 0;
    while (nit < p.maxiter && active) {
      // ---- direction d = -H g (row vl by its owner), slope = g . d; not a descent direction: restart from steepest descent
      if (vl >= 0) {
        double acc = 0.0;
        for (int b = 0; b < P; ++b) acc = add_rn(acc, mul_rn(L.H[vl][b], L.G[b]));
        L.D[vl] = -acc;
      }
      sync();
      double sl = 0.0;
      for (int a = 0; a < P; ++a) sl = add_rn(sl, mul_rn(L.G[a], L.D[a]));
      if (!(sl < 0.0)) {
        set_identity();
        if (vl >= 0) L.D[vl] = -L.G[vl];
        sync();
        sl = 0.0;
        for (int a = 0; a < P; ++a) sl = sub_rn(sl, mul_rn(L.G[a], L.G[a]));
      }
      // ---- the full step with its gradient; the rest of the ladder in the same pass where the candidates fit (its values are used
      // only on rejection), else in a pass of its own when the full step is rejected
      evaluate(p.alphas[0], ladder_in_pass ? G : 0);
      double fs;
      read_fg(fs, L.Gn);
      const double F0 = (fs == fs && fabs(fs) != INFINITY) ? fs : INFINITY;
      const bool need = !(F0 <= add_rn(f, mul_rn(mul_rn(p.c1, p.alphas[0]), sl)));
      if (need && !ladder_in_pass && G > 0) {
        // (the gradient of the full step is parked in Gn; a ladder pass overwrites F / OK of the candidates, not Gn)
        if (L.AR != nullptr) {
          if (writer) {
            L.AR[0] = f;
            L.AR[1] = sl;
          }
          sync();
        }
        evaluate(NaN, G);          // NaN: no gradient candidates in this pass, the n_ladder points sit at candidates G1 ..
      }
      int first = -1, best = 0;
      double Fbest = F0, Ffirst = 0.0;
      for (int r = 0; r < NA; ++r) {
        double Fr = F0;
        if (r > 0) {
          const double v = (need && L.OK[G1 + r - 1]) ? L.F[G1 + r - 1] : NaN;
          Fr = (v == v && fabs(v) != INFINITY) ? v : INFINITY;
        }
        if (first < 0 && Fr <= add_rn(f, mul_rn(mul_rn(p.c1, p.alphas[r]), sl))) { first = r; Ffirst = Fr; }
        if (Fr < Fbest) { Fbest = Fr; best = r; }
      }
      if (first < 0) { first = best; Ffirst = Fbest; }
      const bool moved = Ffirst < f;
      const double a_step = moved ? p.alphas[first] : 0.0;
      sync();                       // (every thread has read F / OK of this pass before anything overwrites them)
      if (vl >= 0) L.S[vl] = mul_rn(a_step, L.D[vl]);
      sync();
      double fn = fs;
      if (need && moved) {
        // the accepted point is a shorter rung: its objective and gradient (x + s = x + alpha_first d)
        evaluate(a_step, 0);
        read_fg(fn, L.Gn);
      }
      if (moved) {
        // ---- rank-two update of the inverse Hessian (curvature guard as scipy), then accept
        double sy = 0.0, ss = 0.0, yy = 0.0;
        for (int k = 0; k < P; ++k) {
          const double y = sub_rn(L.Gn[k], L.G[k]);
          sy = add_rn(sy, mul_rn(L.S[k], y));
          ss = add_rn(ss, mul_rn(L.S[k], L.S[k]));
          yy = add_rn(yy, mul_rn(y, y));
        }
        if (sy > 1e-12 * sqrt(mul_rn(ss, yy)) && sy > 0.0) {
          const double rho = 1.0 / sy;
          if (vl >= 0) {
            double acc = 0.0;
            for (int b = 0; b < P; ++b) acc = add_rn(acc, mul_rn(L.H[vl][b], sub_rn(L.Gn[b], L.G[b])));
            L.Hy[vl] = acc;
          }
          sync();
          double yHy = 0.0;
          for (int a = 0; a < P; ++a) yHy = add_rn(yHy, mul_rn(sub_rn(L.Gn[a], L.G[a]), L.Hy[a]));
          const double coef = mul_rn(rho, add_rn(1.0, mul_rn(rho, yHy)));
          if (vl >= 0) {
            const int a = vl;
            for (int b = 0; b < P; ++b)
              L.H[a][b] = add_rn(sub_rn(L.H[a][b], add_rn(mul_rn(mul_rn(rho, L.S[a]), L.Hy[b]), mul_rn(mul_rn(rho, L.S[b]), L.Hy[a]))), mul_rn(mul_rn(coef, L.S[a]), L.S[b]));
          }
        }
        sync();                     // (every thread has read G / Gn / S of this iteration)
        f = fn;
        if (vl >= 0) {
          L.G[vl] = L.Gn[vl];
          L.X[vl] = add_rn(L.X[vl], L.S[vl]);
        }
        sync();
      }
      active = moved && gmax_at_least(L.G, p.gtol);
      ++nit;
      sync();
    }
    if (writer) {
      p.f_hist[((int64_t)step * 2 + 1) * p.T + t] = f;
      p.nit[(int64_t)step * p.T + t] = nit;
    }
    if (p.params_hist != nullptr && vl >= 0) p.params_hist[((int64_t)step * p.T + t) * P + vl] = L.X[vl];
  }
  if (vl >= 0) {
    p.params[t * P + vl] = L.X[vl];
    if (p.hinv != nullptr)
      for (int b = 0; b < P; ++b) p.hinv[(t * P + vl) * P + b] = L.H[vl][b];
  }
}

}  // namespace qmps
