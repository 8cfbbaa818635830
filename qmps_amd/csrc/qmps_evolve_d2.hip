// qmps_evolve_d2.hip - the WHOLE time evolution of the reference's own case (bond dimension D = 2) in one launch (gfx950 only).
//
// Reference: qmps/new_time_evolve.py:276-292 (`minimize(obj, params, (A_, WW))` per time step, scipy BFGS with finite-difference
// gradients, ShallowFullStateTensor(2, .) with 15 angles, :186-187), scripts/loschmidt.py:367-375 (ShallowCNOTStateTensor(2, .)).
// The lock-step drivers (tools.batched_bfgs, qmps_evolve_bfgs) pay a host round trip per BFGS iteration and make every trajectory
// wait for the slowest one; at D = 2 the GPU then idles (profiles/r03h_evolve_d2_t256.json: 0.03 ms of kernels inside 5 ms).  Here
// ONE WAVE owns a trajectory for the whole run - every time step, every BFGS iteration, without returning to the host:
//   lanes = candidates of an evaluation pass: lane 0 the iterate z, lanes 1 .. P  z + h e_k, lanes P + 1 .. 2P  z - h e_k (the
//   central-difference columns), lanes 2P + 1 .. 2P + G the backtracking points x + alpha_r d - each lane simulates the ansatz
//   circuit of ITS parameter vector (two columns of the 4 x 4 unitary, qmps_circuit.h), and solves ITS mixed transfer map
//   (4 x 4 complex, squared until converged: qmps_overlap_d2.h - the code of overlap_lane_kernel);
//   x, g, d, s, the inverse Hessian H (P x P) and the reference tensor live in the wave's LDS; lane a owns row a of H.
// The iteration is tools.batched_bfgs / qmps_evolve_bfgs for ONE trajectory, decision for decision (speculative full step with
// its gradient, Armijo ladder, first-accepted / best rung, rank-two update with the curvature guard, steepest-descent restart),
// the host driver's floating-point expressions of the optimiser algebra reproduced with explicitly rounded operations (no FMA
// contraction); the evaluations differ from the host path's in the last bit (the compiler contracts the same source differently
// in a different kernel), so the two drivers agree to rounding level per iteration, not bit for bit over a whole minimisation.
// A double-precision sincos costs more than a two-qubit layer: the P angles of a pass's base point are computed once and shared
// through LDS, a central-difference lane adds ONE sincos (its shifted angle).  Trajectories are independent: nobody waits.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_circuit.h"
#include "qmps_overlap_d2.h"

namespace qmps {

namespace {

constexpr int PMAX = 16;     // parameters per trajectory (ShallowFull: 15; ShallowCNOT at D = 2: 2 per layer)

__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }

}  // namespace


template <int KIND>
__global__ __launch_bounds__(64) void evolve_bfgs_d2_kernel(EvolveD2Args p) {
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x, P = p.P, NA = p.NA, G = NA - 1;
  __shared__ double sX[PMAX], sG[PMAX], sD[PMAX], sS[PMAX], sGn[PMAX], sHy[PMAX], sH[PMAX][PMAX + 1], sF[64];
  __shared__ int sOK[64];
  __shared__ double2 sA[8], sCS[PMAX], sLCS[kEvolveMaxAlphas * PMAX];
  __shared__ double sZ[PMAX];
  const double2* W = (const double2*)p.WW;
  const double NaN = __builtin_nan("");
  // ---- one evaluation pass.  Lane roles: < G1 = 2P + 1: central-difference columns of z = x + coef d; [G1, G1 + n_ladder): x + alpha_{r+1} d
  // returns nothing: sF / sOK hold -sqrt|eta| and status == OK of every lane
  double nfev = 0.0, nrounds = 0.0;       // (nrounds: this LANE's squarings, summed over the wave at the end)
  int nfail = 0;
  auto evaluate = [&](double coef, int n_ladder) {
    const int G1 = 2 * P + 1;
    const bool grad_lane = lane < G1, ladder_lane = !grad_lane && lane < G1 + n_ladder;
    // ---- cos / sin of every (scaled) angle ONCE per pass, shared through LDS: the P angles of the base point z = x + coef d (the
    // central-difference columns differ from it in one angle each) by lanes 0 .. P - 1, the n_ladder P angles of the backtracking
    // points spread over the wave - a lane then computes ONE sincos of its own (its shifted angle) instead of 2 P
    if (lane < P) {
      const double z = coef != 0.0 ? add_rn(sX[lane], mul_rn(coef, sD[lane])) : sX[lane];
      sZ[lane] = z;
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(lane) * z, &sn, &cs_);
      sCS[lane] = make_double2(cs_, sn);
    }
    for (int idx = lane; idx < n_ladder * P; idx += 64) {
      const int r = idx / P, l = idx - r * P;
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(l) * add_rn(sX[l], mul_rn(p.alphas[r + 1], sD[l])), &sn, &cs_);
      sLCS[r * PMAX + l] = make_double2(cs_, sn);
    }
    __builtin_amdgcn_wave_barrier();
    if (grad_lane || ladder_lane) {
      const int isel = (grad_lane && lane > 0) ? (lane - 1) % P : -1;
      double2 own = make_double2(1.0, 0.0);
      if (isel >= 0) {
        double sn, cs_;
        sincos(ansatz_param_scale<KIND>(isel) * add_rn(sZ[isel], lane <= P ? p.h : -p.h), &sn, &cs_);
        own = make_double2(cs_, sn);
      }
      const double2* lcs = sLCS + (ladder_lane ? lane - G1 : 0) * PMAX;
      double bre[8], bim[8];
      if (p.probe & 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { bre[k] = sA[k].x + 1e-3 * own.x * (k == 3); bim[k] = sA[k].y; }
      } else
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        Reg<2> r;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          r.re[x] = (x == j) ? 1.0 : 0.0;
          r.im[x] = 0.0;
        }
        ansatz_circuit_cs<2, KIND>(r, [&](int l) { return ladder_lane ? lcs[l] : (l == isel ? own : sCS[l]); }, P);
#pragma unroll
        for (int x = 0; x < 4; ++x) {          // B[s][i][j] = amplitude[2 i + s] of column j
          bre[((x & 1) * 2 + (x >> 1)) * 2 + j] = r.re[x];
          bim[((x & 1) * 2 + (x >> 1)) * 2 + j] = r.im[x];
        }
      }
      OverlapLaneResult o;
      if (p.probe & 1) { o.eta_r = bre[0] + bre[5]; o.eta_i = bim[3]; o.rounds = 0; o.status = QMPS_ST_OK; }
      else overlap_lane_solve([&](int k) { return sA[k]; }, [&](int k) { return make_double2(bre[k], bim[k]); }, W, p.max_rounds, p.tol, o);
      sF[lane] = -__builtin_sqrt(__builtin_sqrt(o.eta_r * o.eta_r + o.eta_i * o.eta_i));
      sOK[lane] = o.status == QMPS_ST_OK ? 1 : 0;
      nrounds += (double)o.rounds;
    }
    __builtin_amdgcn_wave_barrier();
    nfev += (double)(G1 + n_ladder);
    for (int l = 0; l < G1 + n_ladder; ++l) nfail += sOK[l] ? 0 : 1;
  };
  // objective and gradient of the last pass (gradient lanes): f, and g into `gout` (LDS)
  auto read_fg = [&](double& f, double* gout) {
    f = sOK[0] ? sF[0] : NaN;
    if (lane < P) gout[lane] = (sOK[1 + lane] && sOK[1 + P + lane]) ? (sF[1 + lane] - sF[1 + P + lane]) / (2.0 * p.h) : NaN;
    __builtin_amdgcn_wave_barrier();
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
    if (lane < P)
      for (int b = 0; b < P; ++b) sH[lane][b] = lane == b ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
  };

  if (lane < P) {
    sX[lane] = p.params[t * P + lane];
    sD[lane] = 0.0;
  }
  if (p.carry_in && p.hinv != nullptr) {
    if (lane < P)
      for (int b = 0; b < P; ++b) sH[lane][b] = p.hinv[(t * P + lane) * P + b];
  }
  __builtin_amdgcn_wave_barrier();
  if (!(p.carry_in && p.hinv != nullptr)) set_identity();

  for (int step = 0; step < p.n_steps; ++step) {
    // ---- the step's reference tensor A = tensor(x): lanes 0, 1 simulate the two columns
    if (lane < P) {
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(lane) * sX[lane], &sn, &cs_);
      sCS[lane] = make_double2(cs_, sn);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 2) {
      Reg<2> r;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        r.re[x] = (x == lane) ? 1.0 : 0.0;
        r.im[x] = 0.0;
      }
      ansatz_circuit_cs<2, KIND>(r, [&](int l) { return sCS[l]; }, P);
#pragma unroll
      for (int x = 0; x < 4; ++x) sA[((x & 1) * 2 + (x >> 1)) * 2 + lane] = make_double2(r.re[x], r.im[x]);
    }
    __builtin_amdgcn_wave_barrier();
    if (!(p.carry && (step > 0 || p.carry_in))) set_identity();
    double f;
    evaluate(0.0, 0);
    read_fg(f, sG);
    if (lane == 0) p.f_hist[((int64_t)step * 2 + 0) * p.T + t] = f;
    bool active = gmax_at_least(sG, p.gtol);
    int nit = 0;
    while (nit < p.maxiter && active) {
      // ---- direction d = -H g (lane a: row a), slope = g . d; not a descent direction: restart from steepest descent
      if (lane < P) {
        double acc = 0.0;
        for (int b = 0; b < P; ++b) acc = add_rn(acc, mul_rn(sH[lane][b], sG[b]));
        sD[lane] = -acc;
      }
      __builtin_amdgcn_wave_barrier();
      double sl = 0.0;
      for (int a = 0; a < P; ++a) sl = add_rn(sl, mul_rn(sG[a], sD[a]));
      if (!(sl < 0.0)) {
        set_identity();
        if (lane < P) sD[lane] = -sG[lane];
        __builtin_amdgcn_wave_barrier();
        sl = 0.0;
        for (int a = 0; a < P; ++a) sl = sub_rn(sl, mul_rn(sG[a], sG[a]));
      }
      // ---- the full step with its gradient AND the rest of the ladder in one pass (the ladder values are used only on rejection)
      evaluate(p.alphas[0], G);
      double fs;
      read_fg(fs, sGn);
      const int G1 = 2 * P + 1;
      double F0 = (fs == fs && fabs(fs) != INFINITY) ? fs : INFINITY;
      const bool need = !(F0 <= add_rn(f, mul_rn(mul_rn(p.c1, p.alphas[0]), sl)));
      int first = -1, best = 0;
      double Fbest = F0, Ffirst = 0.0;
      for (int r = 0; r < NA; ++r) {
        double Fr = F0;
        if (r > 0) {
          const double v = (need && sOK[G1 + r - 1]) ? sF[G1 + r - 1] : NaN;
          Fr = (v == v && fabs(v) != INFINITY) ? v : INFINITY;
        }
        if (first < 0 && Fr <= add_rn(f, mul_rn(mul_rn(p.c1, p.alphas[r]), sl))) { first = r; Ffirst = Fr; }
        if (Fr < Fbest) { Fbest = Fr; best = r; }
      }
      if (first < 0) { first = best; Ffirst = Fbest; }
      const bool moved = Ffirst < f;
      const double a_step = moved ? p.alphas[first] : 0.0;
      if (lane < P) sS[lane] = mul_rn(a_step, sD[lane]);
      __builtin_amdgcn_wave_barrier();
      double fn = fs;
      if (need && moved) {
        // the accepted point is a shorter rung: its objective and gradient (x + s = x + alpha_first d)
        evaluate(a_step, 0);
        read_fg(fn, sGn);
      }
      if (moved) {
        // ---- rank-two update of the inverse Hessian (curvature guard as scipy), then accept
        double sy = 0.0, ss = 0.0, yy = 0.0;
        for (int k = 0; k < P; ++k) {
          const double y = sub_rn(sGn[k], sG[k]);
          sy = add_rn(sy, mul_rn(sS[k], y));
          ss = add_rn(ss, mul_rn(sS[k], sS[k]));
          yy = add_rn(yy, mul_rn(y, y));
        }
        if (sy > 1e-12 * sqrt(mul_rn(ss, yy)) && sy > 0.0) {
          const double rho = 1.0 / sy;
          if (lane < P) {
            double acc = 0.0;
            for (int b = 0; b < P; ++b) acc = add_rn(acc, mul_rn(sH[lane][b], sub_rn(sGn[b], sG[b])));
            sHy[lane] = acc;
          }
          __builtin_amdgcn_wave_barrier();
          double yHy = 0.0;
          for (int a = 0; a < P; ++a) yHy = add_rn(yHy, mul_rn(sub_rn(sGn[a], sG[a]), sHy[a]));
          const double coef = mul_rn(rho, add_rn(1.0, mul_rn(rho, yHy)));
          if (lane < P) {
            const int a = lane;
            for (int b = 0; b < P; ++b)
              sH[a][b] = add_rn(sub_rn(sH[a][b], add_rn(mul_rn(mul_rn(rho, sS[a]), sHy[b]), mul_rn(mul_rn(rho, sS[b]), sHy[a]))), mul_rn(mul_rn(coef, sS[a]), sS[b]));
          }
          __builtin_amdgcn_wave_barrier();
        }
        f = fn;
        if (lane < P) {
          sG[lane] = sGn[lane];
          sX[lane] = add_rn(sX[lane], sS[lane]);
        }
        __builtin_amdgcn_wave_barrier();
      }
      active = moved && gmax_at_least(sG, p.gtol);
      ++nit;
    }
    if (lane == 0) {
      p.f_hist[((int64_t)step * 2 + 1) * p.T + t] = f;
      p.nit[(int64_t)step * p.T + t] = nit;
    }
    if (p.params_hist != nullptr && lane < P) p.params_hist[((int64_t)step * p.T + t) * P + lane] = sX[lane];
  }
  if (lane < P) {
    p.params[t * P + lane] = sX[lane];
    if (p.hinv != nullptr)
      for (int b = 0; b < P; ++b) p.hinv[(t * P + lane) * P + b] = sH[lane][b];
  }
  nrounds = wave_sum(nrounds);
  if (lane == 0) {
    if (p.nfev != nullptr) p.nfev[t] = nfev;
    if (p.rounds != nullptr) p.rounds[t] = nrounds;
    if (p.fail != nullptr) p.fail[t] = nfail;
  }
}

hipError_t launch_evolve_bfgs_d2(int kind, const EvolveD2Args& a, hipStream_t st) {
  if (a.T <= 0) return hipSuccess;
  if (a.P < 1 || a.P > PMAX || a.NA < 1 || a.NA > kEvolveMaxAlphas || 2 * a.P + 1 + (a.NA - 1) > 64) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.T), block(64);
  switch (kind) {
    case 0: hipLaunchKernelGGL(evolve_bfgs_d2_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(evolve_bfgs_d2_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(evolve_bfgs_d2_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(evolve_bfgs_d2_kernel<3>, grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL(evolve_bfgs_d2_kernel<6>, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace qmps
