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
#include "qmps_evolve_core.h"

namespace qmps {

namespace {

constexpr int PMAX = kEvolvePMax;     // parameters per trajectory (ShallowFull: 15; ShallowCNOT at D = 2: 2 per layer)
using evolve_detail::add_rn;
using evolve_detail::mul_rn;

}  // namespace


template <int KIND>
__global__ __launch_bounds__(64) void evolve_bfgs_d2_kernel(EvolveD2Args p) {
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x, P = p.P;
  __shared__ double sX[PMAX], sG[PMAX], sD[PMAX], sS[PMAX], sGn[PMAX], sHy[PMAX], sH[PMAX][PMAX + 1], sF[64];
  __shared__ int sOK[64];
  __shared__ double2 sA[8], sCS[PMAX], sLCS[kEvolveMaxAlphas * PMAX];
  __shared__ double sZ[PMAX];
  const double2* W = (const double2*)p.WW;
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
  BfgsLds L;
  L.X = sX; L.G = sG; L.D = sD; L.S = sS; L.Gn = sGn; L.Hy = sHy; L.H = sH; L.F = sF; L.OK = sOK;
  auto build_reference = [&]() {
    // the step's reference tensor A = tensor(x): lanes 0, 1 simulate the two columns
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
  };
  bfgs_time_evolution(p, t, lane < P ? lane : -1, lane == 0, L, evaluate, build_reference, [] { __builtin_amdgcn_wave_barrier(); }, true);
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
