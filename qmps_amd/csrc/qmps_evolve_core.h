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

// The optimiser algebra below used to walk its vectors in loops of runtime length P - every iteration a dependent LDS read (~120 cycles for
// a wave alone on its SIMD, nothing to hide it behind): ~8 such loops per BFGS iteration were 6.4 us of a 15 us pass at D = 2 (round 6,
// phase timers of a tuning build).  Now the vectors are ZERO-PADDED to kEvolvePMax (see the start of bfgs_time_evolution) and read EIGHT
// elements at a time - independent loads issued together, no guards: a padded term adds an exact zero - while the sums keep their order and
// their explicitly rounded operations: the same bits as before.  (Sixteen at a time, and arrays held across phases, cost the D = 2 kernel 60
// registers and its second wave per SIMD: 20 % at 4 096 trajectories.)
constexpr int kEvolveChunk = 8;
__device__ __forceinline__ void load8(const double* src, double (&v)[kEvolveChunk]) {
#pragma unroll
  for (int k = 0; k < kEvolveChunk; ++k) v[k] = src[k];
}
// acc + a[0] b[0] + a[1] b[1] + ... in this order, each product and each sum rounded
__device__ __forceinline__ double dot_rn(const double* a, const double* b, double acc) {
#pragma unroll
  for (int c = 0; c < kEvolvePMax; c += kEvolveChunk) {
    double av[kEvolveChunk], bv[kEvolveChunk];
    load8(a + c, av);
    load8(b + c, bv);
#pragma unroll
    for (int k = 0; k < kEvolveChunk; ++k) acc = __dadd_rn(acc, __dmul_rn(av[k], bv[k]));
    __builtin_amdgcn_sched_barrier(0);      // (chunk by chunk: see above)
  }
  return acc;
}

template <class Eval, class RefBuild, class Sync>
__device__ __forceinline__ void bfgs_time_evolution(const EvolveD2Args& p, int64_t t, int vl, bool writer, const BfgsLds& L, Eval evaluate,
                                                    RefBuild build_reference, Sync sync, bool ladder_in_pass) {
  using namespace evolve_detail;
  constexpr int PM = kEvolvePMax, CH = kEvolveChunk;
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
    bool nan = false;
#pragma unroll
    for (int c = 0; c < PM; c += CH) {
      double gv[CH];
      load8(gt + c, gv);
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const double v = gv[k];
        nan = nan || (v != v);
        const double a = fabs(v);
        m = a > m ? a : m;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return !nan && m >= bound;
  };
  auto set_identity = [&]() {
    if (vl >= 0)
      for (int b = 0; b < P; ++b) L.H[vl][b] = vl == b ? 1.0 : 0.0;
    sync();
  };
  // zero padding of the vectors and of the rows of H up to kEvolvePMax (nothing below ever writes beyond P)
  if (vl >= 0) {
    for (int b = 0; b < PM; ++b) L.H[vl][b] = 0.0;
    if (vl == 0)
      for (int k = P; k < PM; ++k) L.X[k] = L.G[k] = L.D[k] = L.S[k] = L.Gn[k] = L.Hy[k] = 0.0;
  }
  sync();
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
      if (vl >= 0) L.D[vl] = -dot_rn(L.H[vl], L.G, 0.0);
      sync();
      double sl = dot_rn(L.G, L.D, 0.0);
      if (!(sl < 0.0)) {
        set_identity();
        if (vl >= 0) L.D[vl] = -L.G[vl];
        sync();
        sl = 0.0;
#pragma unroll
        for (int c = 0; c < PM; c += CH) {
          double gv[CH];
          load8(L.G + c, gv);
#pragma unroll
          for (int k = 0; k < CH; ++k) sl = sub_rn(sl, mul_rn(gv[k], gv[k]));
        }
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
      if (need) {                   // (an accepted full step - the usual case - needs no look at the ladder: first = best = 0)
        for (int r = 0; r < NA; ++r) {
          double Fr = F0;
          if (r > 0) {
            const double v = L.OK[G1 + r - 1] ? L.F[G1 + r - 1] : NaN;
            Fr = (v == v && fabs(v) != INFINITY) ? v : INFINITY;
          }
          if (first < 0 && Fr <= add_rn(f, mul_rn(mul_rn(p.c1, p.alphas[r]), sl))) { first = r; Ffirst = Fr; }
          if (Fr < Fbest) { Fbest = Fr; best = r; }
        }
        if (first < 0) { first = best; Ffirst = Fbest; }
      } else {
        first = 0;
        Ffirst = F0;
      }
      const bool moved = Ffirst < f;
      const double a_step = moved ? p.alphas[first] : 0.0;
      if (vl >= 0) L.S[vl] = mul_rn(a_step, L.D[vl]);
      sync();                       // (S visible; and every thread has read F / OK of this pass before the next pass overwrites them)
      double fn = fs;
      if (need && moved) {
        // the accepted point is a shorter rung: its objective and gradient (x + s = x + alpha_first d)
        evaluate(a_step, 0);
        read_fg(fn, L.Gn);
      }
      if (moved) {
        // ---- rank-two update of the inverse Hessian (curvature guard as scipy), then accept.  Two barriers: before the first every thread has
        // s.y, s.s, y.y (y = Gn - G formed on the fly) and the owners H y; behind it the owners accept (G <- Gn, X <- X + S) and publish H y
        // and y (parked in D, which is dead until the next direction); behind the second come y.Hy, the rows of H and the convergence test
        // on the new G.  (Round 5 had five barriers here and read its vectors element by element.)
        double sy = 0.0, ss = 0.0, yy = 0.0, hy_own = 0.0;
        const double* hrow = L.H[vl >= 0 ? vl : 0];      // (threads without a row walk row 0 and store nothing: no branch inside the chunks)
#pragma nounroll      // (the two chunks as a LOOP: unrolled, the register allocation of the whole kernel went from 218 to 256 + 17)
        for (int c = 0; c < PM; c += CH) {
          double sv[CH], gn[CH], gv[CH], hv[CH];
          load8(L.S + c, sv);
          load8(L.Gn + c, gn);
          load8(L.G + c, gv);
          load8(hrow + c, hv);
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            const double y = sub_rn(gn[k], gv[k]);
            sy = add_rn(sy, mul_rn(sv[k], y));
            ss = add_rn(ss, mul_rn(sv[k], sv[k]));
            yy = add_rn(yy, mul_rn(y, y));
            hy_own = add_rn(hy_own, mul_rn(hv[k], y));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        const bool curv = sy > 1e-12 * sqrt(mul_rn(ss, yy)) && sy > 0.0;
        double xnew = 0.0, gnew = 0.0, ynew = 0.0, sa = 0.0;
        if (vl >= 0) {
          sa = L.S[vl];
          xnew = add_rn(L.X[vl], sa);
          gnew = L.Gn[vl];
          ynew = sub_rn(gnew, L.G[vl]);
        }
        sync();                     // (every thread has read what it needs of S, G, Gn of this iteration)
        f = fn;
        if (vl >= 0) {
          L.G[vl] = gnew;
          L.X[vl] = xnew;
          L.D[vl] = ynew;
          if (curv) L.Hy[vl] = hy_own;
        }
        sync();
        if (curv) {
          const double rho = 1.0 / sy;
          const double yHy = dot_rn(L.D, L.Hy, 0.0);
          const double coef = mul_rn(rho, add_rn(1.0, mul_rn(rho, yHy)));
          if (vl >= 0) {
            const double hya = hy_own;      // (Hy[vl]: this thread's own)
#pragma nounroll
            for (int c = 0; c < PM; c += CH) {
              double hv[CH], hy[CH], sv[CH];
              load8(L.H[vl] + c, hv);
              load8(L.Hy + c, hy);
              load8(L.S + c, sv);
#pragma unroll
              for (int k = 0; k < CH; ++k)      // (no guard: a padded column has hv = hy = sv = 0 and stays 0)
                L.H[vl][c + k] = add_rn(sub_rn(hv[k], add_rn(mul_rn(mul_rn(rho, sa), hy[k]), mul_rn(mul_rn(rho, sv[k]), hya))), mul_rn(mul_rn(coef, sa), sv[k]));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
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
