// qmps_evolve_d16.hip - the whole BFGS time evolution at bond dimension D = 16 in one launch (gfx950 only): ONE WORKGROUP of eight waves
// owns a trajectory and runs every BFGS iteration of every time step on its own compute unit, at its own pace.
//
// Reference: qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 (`minimize(obj, params, (A_, WW))` per time step; config 4 of
// BASELINE.json).  The lock-step driver (qmps_evolve_bfgs, qmps_evolve_lockstep.hip) evaluates all trajectories' gradients as ONE batch:
// a batch lasts as long as its SLOWEST pair of eigen-solves (max ~50 power steps against a mean of ~15 at 256 trajectories), and the
// six kernels of an iteration are ~40 us of launches and tails around it - 0.55-0.60 ms per time step.  Here, as at D = 2 and 4
// (qmps_evolve_d2.hip, qmps_evolve_d4.hip), the optimiser never leaves the device and a trajectory waits for nobody:
//   * tensors: the five-qubit ShallowCNOT circuit distributed over the lanes of a wave (qmps_circuit_wave.h), two columns per pass,
//     into LDS (rows padded to 17);
//   * the RIGHT and the LEFT fixed point of the iterate's mixed transfer map by the power method on the matrix cores
//     (v_mfma_f64_16x16x4, three-product complex form): waves 0-3 the map, waves 4-7 the adjoint map, wave w of a team owns the
//     physical index s = w exactly as in overlap_mfma_d16x4_body (qmps_overlap.hip); each team runs at its own pace on a barrier of
//     its own (an arrival counter in LDS: team_barrier) and waits for the other at the workgroup barrier behind the solves;
//   * G_s = y^+ C_s r from the right team's registers (C_s is still there in the A-layout, r in the accumulator layout IS the
//     B-layout): two products per wave;
//   * the 2 P central-difference neighbours and the iterate itself by the two-sided quotient eta' = <y, T'(r)> / <y, r>
//     (qmps_overlap_grad.hip): a wave builds a neighbour's tensor in its own LDS tile, forms merge(B', B') on the matrix cores and
//     contracts it with G - nothing of an evaluation ever touches HBM;
//   * a rejected full step's backtracking points: two at a time (one per team, both the map itself), started from the rejected
//     step's fixed point, stopping at the first rung that passes the Armijo test (the rungs behind it cannot change the decision);
//   * the optimiser loop itself: qmps_evolve_core.h, shared with D = 2 and 4.
// 146 KB of LDS and up to 256 registers per wave: one workgroup per compute unit, 256 trajectories fill the chip.
// MEASURED SLOWER than the lock-step driver at 32 ... 2 048 trajectories (profiles/EXPERIMENTS.md, round 5): the launch lasts as long as
// the trajectory whose map has the smallest gap, and that trajectory is better served by the lock-step (a compute unit per solve, its
// neighbours built elsewhere).  An option (qmps_evolve_bfgs_device at D = 16), not the default.
// Tolerances as in the lock-step driver: the two solves of a gradient stop at max(tol, 1e-8) (the objective comes from the two-sided
// quotient, error ~ residual^2), with `adaptive` at clamp(1e-3 max|g|, max(tol, 1e-8), 1e-6); backtracking points at tol.
// A solve that exhausts its cap leaves status 1: the objective is NaN, the optimiser treats the point as rejected, `fail` counts it
// (counters_out[1] of qmps_evolve_bfgs_device; the Krylov fall-back for such maps lives in the lock-step path, qmps_evolve_bfgs).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_overlap_d4.h"      // cmma16, cmma16_3m
#include "qmps_circuit_wave.h"
#include "qmps_evolve_core.h"

namespace qmps {

namespace {
constexpr int kW16 = 8;                      // waves per workgroup: two teams of four
constexpr int kLdp = 17;                     // padded row of a tensor / transpose tile in LDS
constexpr int kTens = 2 * 16 * kLdp;         // a tensor [s][i][j]
constexpr int kScrT = kW16 * 16 * kLdp;      // the waves' private transpose tiles
constexpr int kScrX = kScrT + 2 * kW16 * 256;    // + two sets of one exchange matrix per wave
constexpr int kScr = kScrX > kW16 * kTens ? kScrX : kW16 * kTens;      // (between two solves: a neighbour tile per wave)

struct Solve16 {
  v4f64 xr, xi;            // the fixed point in the accumulator layout, unit Frobenius norm (alike in the team's four waves)
  double cre[4], cim[4];   // C_w in the A-layout (the adjoint team: C_w^+)
  double eta_r, eta_i;
  int iters, status;
};

// Barrier over the four waves of a team: *cnt counts the arrivals of the current solve (monotone, reset between solves under a
// workgroup barrier), step k waits for 4 k.  LDS operations of a wave execute in order, so a wave that sees the count sees the
// tiles published before it; all eight waves of the workgroup are resident, so spinning cannot starve anyone.
__device__ __forceinline__ void team_barrier(int* cnt, int target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Both teams of the workgroup solve one fixed point each (called by all 512 threads): team = wave >> 2 iterates the map of (sA, Bt)
// - Bt this team's candidate tensor - or, with adj, its adjoint, from the start vector xs (row-major [16][16] in LDS; null or zero:
// the identity) to a residual of sqrt(tol2) or max_rounds steps.  enabled = false: the team sets up and skips the iteration.
// The two workgroup barriers of the set-up are the only ones: the caller joins the teams again behind the call.
// The loop is overlap_mfma_d16x4_body's (qmps_overlap.hip) without deflation steps and without the Krylov hand-over.
__device__ __forceinline__ void solve_teams_d16(const double2* sA, const double2* Bt, bool adj, bool enabled, const double2* W, const double2* xs, double tol2,
                                                int max_rounds, double2 (*sT_all)[16 * kLdp], double2 (*sX_all)[256], int* sArrive, Solve16& o) {
  constexpr int D = 16, LD = kLdp;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, team = wave >> 2, tw = wave & 3;
  double2* sT = sT_all[wave];
  auto to_a_layout = [&](const v4f64& re, const v4f64& im, double (&are)[4], double (&aim)[4]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) sT[(4 * q + g) * LD + c] = make_double2(re[q], im[q]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double2 t = sT[c * LD + 4 * kk + g];
      are[kk] = t.x;
      aim[kk] = t.y;
    }
  };
  const int t1 = tw >> 1, t2 = tw & 1;          // this wave's pair (s = 2 t1 + t2 = tw)
  double bre[4], bimn[4];                       // conj(Bm_w) in A-layout == Bm_w^+ in B-layout (adjoint: Bm_w in B-layout)
  {
    double pa[4], pai[4], pb[4], pbi[4];
    v4f64 qa, qai, qb, qbi;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double2 va = sA[(t1 * D + c) * LD + 4 * kk + g], vb = Bt[(t1 * D + c) * LD + 4 * kk + g];
      pa[kk] = va.x; pai[kk] = va.y;
      pb[kk] = vb.x; pbi[kk] = vb.y;
      const double2 wa = sA[(t2 * D + 4 * kk + g) * LD + c], wb = Bt[(t2 * D + 4 * kk + g) * LD + c];
      qa[kk] = wa.x; qai[kk] = wa.y;
      qb[kk] = wb.x; qbi[kk] = wb.y;
    }
    v4f64 zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
    cmma16(pa, pai, qa, qai, zr, zi);           // AA_w = A_t1 A_t2
#pragma unroll
    for (int q = 0; q < 4; ++q) sX_all[wave][q * 64 + lane] = make_double2(zr[q], zi[q]);
    v4f64 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
    cmma16(pb, pbi, qb, qbi, yr, yi);           // Bm_w = B_t1 B_t2
    if (adj) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bre[kk] = yr[kk];
        bimn[kk] = yi[kk];
      }
    } else {
      double tr[4], ti[4];
      to_a_layout(yr, yi, tr, ti);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bre[kk] = tr[kk];
        bimn[kk] = -ti[kk];
      }
    }
    if (threadIdx.x < 2) sArrive[threadIdx.x] = 0;
    __syncthreads();
    v4f64 sr = {0, 0, 0, 0}, si = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double2 w = W[tw * 4 + t];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 v = sX_all[team * 4 + t][q * 64 + lane];
        sr[q] += w.x * v.x - w.y * v.y;
        si[q] += w.x * v.y + w.y * v.x;
      }
    }
    if (adj) {                                  // C_w^+ in A-layout = conj of C_w in the accumulator layout
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        o.cre[kk] = sr[kk];
        o.cim[kk] = -si[kk];
      }
    } else {
      to_a_layout(sr, si, o.cre, o.cim);
    }
    __syncthreads();                            // the exchange buffers are free again; the arrival counters are reset
  }
  v4f64 xr, xi;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double2 x0 = overlap_cold_start(4 * q + g, c, 16);      // (generic: see overlap_cold_start)
    xr[q] = x0.x;
    xi[q] = x0.y;
  }
  if (xs != nullptr) {
    v4f64 wr, wi;
    double n2 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 t = xs[(4 * q + g) * D + c];
      wr[q] = t.x; wi[q] = t.y;
      n2 = dfma(t.x, t.x, dfma(t.y, t.y, n2));
    }
    n2 = lane0(wave_sum(n2));
    if (n2 > 1e-200 && n2 < 1e200) {
      const double inv = 1.0 / __builtin_sqrt(n2);
      xr = wr * inv;
      xi = wi * inv;
    }
  }
  double eta_r = 0.0, eta_i = 0.0;
  int iters = 0, status = QMPS_ST_NOT_CONVERGED;
  const v4f64 qre = {bre[0], bre[1], bre[2], bre[3]};
  const v4f64 qim = {bimn[0], bimn[1], bimn[2], bimn[3]};
  // Each team runs its own power method at its own pace: the step's barrier is the TEAM's (an arrival counter in LDS, see
  // team_barrier), not the workgroup's.  With one s_barrier for both teams the two waves of a SIMD formed their partial maps at the
  // same time and exchanged, summed and tested at the same time - 3.2 us per step of the pair against the 2.1 us the matrix pipe
  // needs (24 v_mfma_f64_16x16x4 per wave and step); two teams half a step apart on workgroup barriers were no better (3.5 us: a
  // lone wave cannot keep the pipe busy).  Free-running, a team's LDS and vector work falls into the other's matrix work, as it
  // does between the two workgroups of a trajectory in the lock-step launch.
  for (int k = 1; enabled; ++k) {
    // the partial maps are exchanged through one of TWO sets of buffers, alternately: ONE barrier per step
    double2 (*sXk)[256] = sX_all + kW16 * (k & 1);
    {
      double xar[4], xai[4];
      to_a_layout(xr, xi, xar, xai);
      v4f64 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0}, pr = {0, 0, 0, 0}, pi = {0, 0, 0, 0};
      cmma16_3m(xar, xai, qre, qim, yr, yi);           // Y_w = x Bm_w^+
      cmma16_3m(o.cre, o.cim, yr, yi, pr, pi);         // C_w Y_w
#pragma unroll
      for (int q = 0; q < 4; ++q) sXk[wave][q * 64 + lane] = make_double2(pr[q], pi[q]);
    }
    team_barrier(sArrive + team, 4 * k);
    v4f64 nr = {0, 0, 0, 0}, ni = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w) {                 // the same order in every wave of the team: bit-identical sums
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 t = sXk[team * 4 + w][q * 64 + lane];
        nr[q] += t.x;
        ni[q] += t.y;
      }
    }
    iters = k;
    const bool last = k >= max_rounds;
    if (!((k & 1) == 0 || last)) {                // no test on odd steps
      xr = nr;
      xi = ni;
      continue;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a0 = dfma(xr[q], nr[q], a0);
      a0 = dfma(xi[q], ni[q], a0);
      a1 = dfma(xr[q], ni[q], a1);
      a1 = dfma(-xi[q], nr[q], a1);
      a2 = dfma(nr[q], nr[q], a2);
      a2 = dfma(ni[q], ni[q], a2);
      a3 = dfma(xr[q], xr[q], a3);
      a3 = dfma(xi[q], xi[q], a3);
    }
    const double xx = wave_sum(a3), ixx = xx > 0.0 ? 1.0 / xx : 0.0;
    eta_r = wave_sum(a0) * ixx;
    eta_i = wave_sum(a1) * ixx;
    const double nn = wave_sum(a2);
    double rs = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double dr = nr[q] - (eta_r * xr[q] - eta_i * xi[q]), di = ni[q] - (eta_r * xi[q] + eta_i * xr[q]);
      rs = dfma(dr, dr, rs);
      rs = dfma(di, di, rs);
    }
    const double res2 = lane0(wave_sum(rs)) * ixx;
    if (res2 < tol2) {
      status = QMPS_ST_OK;
      const double inv = xx > 0.0 ? 1.0 / __builtin_sqrt(xx) : 0.0;
      xr *= inv;
      xi *= inv;
      break;
    }
    const double inv = nn > 0.0 ? 1.0 / __builtin_sqrt(nn) : 0.0;
    xr = nr * inv;
    xi = ni * inv;
    if (last) break;
  }
  o.xr = xr;
  o.xi = xi;
  o.eta_r = lane0(eta_r);
  o.eta_i = lane0(eta_i);
  o.iters = iters;
  o.status = status;
}
}  // namespace

template <int KIND>
__global__ __launch_bounds__(64 * kW16) void evolve_bfgs_d16_kernel(EvolveD2Args p) {
  constexpr int PMAX = kEvolvePMax, D = 16, LD = kLdp;
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15, team = wave >> 2, tw = wave & 3;
  const int P = p.P, G1 = 2 * P + 1;
  __shared__ double sX[PMAX], sG[PMAX], sD[PMAX], sS[PMAX], sGn[PMAX], sHy[PMAX], sH[PMAX][PMAX + 1], sF[64], sAR[2];
  __shared__ int sOK[64], sArrive[2], sStat[2], sFirst;
  __shared__ double sCnt[4], sG0max;
  __shared__ double2 sYR;
  __shared__ __attribute__((aligned(16))) double2 sA[kTens], sB[kTens], sR[256], sY[256], sG4[4][256], sScr[kScr];
  double2 (*sT_all)[16 * LD] = (double2 (*)[16 * LD])sScr;
  double2 (*sX_all)[256] = (double2 (*)[256])(sScr + kScrT);
  double2* sNb = sScr + wave * kTens;          // this wave's neighbour tile (the solver's scratch, idle between two solves)
  double2* sB2 = &sG4[0][0];                   // the second team's backtracking point (G is idle in a ladder pass)
  const double2* W = (const double2*)p.WW;
  const double tol_ladder2 = p.tol * p.tol;
  const double grad_tol = p.grad_tol > 0.0 ? p.grad_tol : p.tol;
  const double tol_hi = grad_tol > 1e-6 ? grad_tol : 1e-6;
  const int grad_rounds = p.max_rounds > 100000 ? p.max_rounds : 100000;
  if (tid < 4) sCnt[tid] = 0.0;
  if (tid == 0) { sG0max = 0.0; sFirst = 1; }
  for (int i = tid; i < 256; i += blockDim.x) {      // no fixed points yet: the first solves start from the identity
    sR[i] = make_double2(0.0, 0.0);
    sY[i] = make_double2(0.0, 0.0);
  }
  // columns 2 pass, 2 pass + 1 of the ansatz unitary whose half-angle cosines / sines lane l of the calling wave holds for angle l
  auto build_pass = [&](double2* out, double cn, double sn, int pass) {
    const int a = lane & 31, j = 2 * pass + (lane >> 5);
    double re, im;
    shallow_cnot_wave_column_d16<KIND>(cn, sn, P, j, re, im);
    out[((a & 1) * D + (a >> 1)) * LD + j] = make_double2(re, im);      // A[s][i][j] = amplitude[2 i + s] of column j
  };
  // parameter l of candidate `cand` of the pass around z = x + coef d (0: z; 1 + k / 1 + P + k: z +- h e_k; G1 + r: x + alphas[r + 1] d)
  auto angle_of = [&](int cand, double coef, int l) {
    const bool grad = cand < G1;
    const double a = grad ? coef : p.alphas[cand - G1 + 1];
    double v = a != 0.0 ? evolve_detail::add_rn(sX[l], evolve_detail::mul_rn(a, sD[l])) : sX[l];
    if (grad && cand > 0 && (cand - 1) % P == l) v = evolve_detail::add_rn(v, cand <= P ? p.h : -p.h);
    return v;
  };
  auto half_sincos = [&](int cand, double coef, double& cn, double& sn) {
    cn = 1.0;
    sn = 0.0;
    if (lane < P) sincos(0.5 * angle_of(cand, coef, lane), &sn, &cn);
  };
  // f = -sqrt|eta'| of the tensor Bt by the two-sided quotient (one wave): eta' = sum_s tr(Bm'_s^+ G_s) / <y, r>
  auto probe = [&](int cand, const double2* Bt) {
    double par[2][4], pai[2][4];
    v4f64 qbr[2], qbi[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const double2 va = Bt[(s * D + c) * LD + 4 * kk + g], vb = Bt[(s * D + 4 * kk + g) * LD + c];
        par[s][kk] = va.x;
        pai[s][kk] = va.y;
        qbr[s][kk] = vb.x;
        qbi[s][kk] = vb.y;
      }
    double nr = 0.0, ni = 0.0;
#pragma unroll
    for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        v4f64 mr = {0, 0, 0, 0}, mi = {0, 0, 0, 0};
        cmma16_3m(par[s1], pai[s1], qbr[s2], qbi[s2], mr, mi);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double2 gv = sG4[2 * s1 + s2][(4 * q + g) * D + c];
          nr = dfma(mr[q], gv.x, dfma(mi[q], gv.y, nr));       // conj(bm) g
          ni = dfma(mr[q], gv.y, dfma(-mi[q], gv.x, ni));
        }
      }
    nr = wave_sum(nr);
    ni = wave_sum(ni);
    if (lane == 0) {
      const double2 d = sYR;
      const double den = d.x * d.x + d.y * d.y;
      const double er = (nr * d.x + ni * d.y) / den, ei = (ni * d.x - nr * d.y) / den;
      sF[cand] = -__builtin_sqrt(__builtin_sqrt(er * er + ei * ei));
      sOK[cand] = (sStat[0] == QMPS_ST_OK && sStat[1] == QMPS_ST_OK && den > 1e-280) ? 1 : 0;
    }
  };
  // ---- one evaluation pass (see qmps_evolve_core.h).  coef finite: objective and central-difference gradient at z = x + coef d;
  // coef NaN: the n_ladder backtracking points (candidates G1 ..), until one passes the Armijo test of (sAR[0], sAR[1]) = (f, slope)
  double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool profiling = p.prof != nullptr && tid == 0;
  auto tick = [&]() { return profiling ? (long long)wall_clock64() : 0LL; };
  auto evaluate = [&](double coef, int n_ladder) {
    const bool with_grad = coef == coef;
    const long long k0 = tick();
    if (with_grad) {
      // the tolerance of this gradient's two solves
      double tolg = grad_tol;
      if (p.adaptive) {
        double m = 0.0;
        bool isn = false;
        if (sFirst) {
          m = sG0max;
        } else {
          for (int k = 0; k < P; ++k) {
            const double v = sG[k];
            if (v != v) isn = true;
            const double a = fabs(v);
            m = a > m ? a : m;
          }
        }
        const double tr = 1e-3 * m;
        tolg = isn ? grad_tol : (tr < grad_tol ? grad_tol : (tr > tol_hi ? tol_hi : tr));
      }
      // the iterate's tensor: a pass per wave
      double cn, sn;
      half_sincos(0, coef, cn, sn);
      build_pass(sB, cn, sn, wave);
      __syncthreads();
      const long long k1 = tick();
      Solve16 so;
      solve_teams_d16(sA, sB, team == 1, true, W, team == 0 ? sR : sY, tolg * tolg, grad_rounds, sT_all, sX_all, sArrive, so);
      const long long k2 = tick();
      if (tw == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) (team == 0 ? sR : sY)[(4 * q + g) * D + c] = make_double2(so.xr[q], so.xi[q]);
        if (lane == 0) {
          sStat[team] = so.status;
          atomicAdd(&sCnt[1], (double)so.iters);
          if (so.status != QMPS_ST_OK) atomicAdd(&sCnt[2], 1.0);
        }
      }
      __syncthreads();
      // the first neighbour of every wave (the circuit needs neither fixed point), the right team's waves after their G_w
      if (team == 0) {
        v4f64 ur = {0, 0, 0, 0}, ui = {0, 0, 0, 0}, gr = {0, 0, 0, 0}, gi = {0, 0, 0, 0};
        cmma16_3m(so.cre, so.cim, so.xr, so.xi, ur, ui);          // Z_w = C_w r  (r in the accumulator layout IS the B-layout)
        double yr_[4], yin[4], a = 0.0, b = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double2 yv = sY[(4 * kk + g) * D + c];           // (y^+)[c][4 kk + g] = conj(y[4 kk + g][c])
          yr_[kk] = yv.x;
          yin[kk] = -yv.y;
          a = dfma(yv.x, so.xr[kk], dfma(yv.y, so.xi[kk], a));   // conj(y) r
          b = dfma(yv.x, so.xi[kk], dfma(-yv.y, so.xr[kk], b));
        }
        cmma16_3m(yr_, yin, ur, ui, gr, gi);                      // G_w = y^+ Z_w
#pragma unroll
        for (int q = 0; q < 4; ++q) sG4[tw][(4 * q + g) * D + c] = make_double2(gr[q], gi[q]);
        if (tw == 0) {
          a = wave_sum(a);
          b = wave_sum(b);
          if (lane == 0) sYR = make_double2(a, b);
        }
      }
      const int n_items = 2 * P;
      if (wave < n_items) {
        half_sincos(1 + wave, coef, cn, sn);
#pragma unroll 1
        for (int pass = 0; pass < 8; ++pass) build_pass(sNb, cn, sn, pass);
      }
      __syncthreads();
      const long long k3 = tick();
#pragma unroll 1
      for (int it = wave; it < n_items; it += kW16) {
        if (it != wave) {
          __builtin_amdgcn_wave_barrier();
          half_sincos(1 + it, coef, cn, sn);
#pragma unroll 1
          for (int pass = 0; pass < 8; ++pass) build_pass(sNb, cn, sn, pass);
        }
        __builtin_amdgcn_wave_barrier();
        probe(1 + it, sNb);
      }
      if (wave == kW16 - 1) probe(0, sB);      // the iterate itself: the objective by the same quotient
      __syncthreads();
      if (tid == 0) {
        sCnt[0] += (double)G1;
        if (sFirst) {
          double m = 0.0;
          bool isn = false;
          for (int k = 0; k < P; ++k) {
            const double v = (sOK[1 + k] && sOK[1 + P + k]) ? (sF[1 + k] - sF[1 + P + k]) / (2.0 * p.h) : __builtin_nan("");
            if (v != v) isn = true;
            const double a = fabs(v);
            m = a > m ? a : m;
          }
          sG0max = isn ? 0.0 : m;
          sFirst = 0;
        }
      }
      __syncthreads();
      if (profiling) {
        const long long k4 = tick();
        prof[1] += (double)(k1 - k0); prof[2] += (double)(k2 - k1); prof[3] += (double)(k3 - k2); prof[4] += (double)(k4 - k3);
        prof[6] += 1.0; prof[7] += (double)so.iters;
      }
      return;
    }
    // ---- the backtracking ladder: two rungs at a time
    if (tid < n_ladder) sOK[G1 + tid] = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n_ladder; r0 += 2) {
      const int r = r0 + team;
      const bool on = r < n_ladder;
      double2* Bt = team == 0 ? sB : sB2;
      if (on) {
        double cn, sn;
        half_sincos(G1 + r, coef, cn, sn);
        build_pass(Bt, cn, sn, 2 * tw);
        build_pass(Bt, cn, sn, 2 * tw + 1);
      }
      __syncthreads();
      Solve16 so;
      solve_teams_d16(sA, Bt, false, on, W, sR, tol_ladder2, p.max_rounds, sT_all, sX_all, sArrive, so);
      if (on && tw == 0 && lane == 0) {
        sF[G1 + r] = -__builtin_sqrt(__builtin_sqrt(so.eta_r * so.eta_r + so.eta_i * so.eta_i));
        sOK[G1 + r] = so.status == QMPS_ST_OK ? 1 : 0;
        atomicAdd(&sCnt[0], 1.0);
        atomicAdd(&sCnt[1], (double)so.iters);
        if (so.status != QMPS_ST_OK) atomicAdd(&sCnt[2], 1.0);
      }
      __syncthreads();
      bool hit = false;
      for (int q = r0; q < r0 + 2 && q < n_ladder; ++q) {
        const double v = sOK[G1 + q] ? sF[G1 + q] : __builtin_nan("");
        const double Fr = (v == v && fabs(v) != INFINITY) ? v : INFINITY;
        if (Fr <= evolve_detail::add_rn(sAR[0], evolve_detail::mul_rn(evolve_detail::mul_rn(p.c1, p.alphas[q + 1]), sAR[1]))) hit = true;
      }
      if (hit) break;      // (uniform: every thread reads the same words)
    }
    __syncthreads();
    if (profiling) prof[5] += (double)(tick() - k0);
  };
  auto build_reference = [&]() {
    double cn = 1.0, sn = 0.0;
    if (lane < P) sincos(0.5 * sX[lane], &sn, &cn);
    build_pass(sA, cn, sn, wave);
    if (tid == 0) sFirst = 1;
    __syncthreads();
  };
  BfgsLds L;
  L.X = sX; L.G = sG; L.D = sD; L.S = sS; L.Gn = sGn; L.Hy = sHy; L.H = sH; L.F = sF; L.OK = sOK; L.AR = sAR;
  __syncthreads();
  const long long kstart = tick();
  bfgs_time_evolution(p, t, (wave == 0 && lane < P) ? lane : -1, tid == 0, L, evaluate, build_reference, [] { __syncthreads(); }, false);
  __syncthreads();
  if (profiling) {
    prof[0] = (double)(tick() - kstart);
    for (int q = 0; q < 8; ++q) p.prof[t * 8 + q] = prof[q];
  }
  if (tid == 0) {
    if (p.nfev != nullptr) p.nfev[t] = sCnt[0];
    if (p.rounds != nullptr) p.rounds[t] = sCnt[1];
    if (p.fail != nullptr) p.fail[t] = (int32_t)sCnt[2];
  }
}

hipError_t launch_evolve_bfgs_d16(int kind, const EvolveD2Args& a, hipStream_t st) {
  if (a.T <= 0) return hipSuccess;
  if (a.P < 1 || a.P > kEvolvePMax || a.NA < 1 || a.NA > kEvolveMaxAlphas || 2 * a.P + a.NA > 64) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.T), block(64 * kW16);
  switch (kind) {
    case 0: hipLaunchKernelGGL(evolve_bfgs_d16_kernel<0>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(evolve_bfgs_d16_kernel<3>, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace qmps
