// qmps_evolve_d4.hip - the whole BFGS time evolution at bond dimension D = 4 in one launch (gfx950 only): ONE WORKGROUP owns a trajectory,
// its WAVES are the candidates of an evaluation pass.
//
// Reference: qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 (`minimize(obj, params, (A_, WW))` per time step).  The
// host-loop driver (qmps_evolve_bfgs) spends 15 BFGS iterations x ~0.1 ms of round trip per time step around ~0.05 ms of kernels
// at D = 4 (profiles/r03h_evolve_d4_t256.json: kernel share of wall 0.3).  Here, as in qmps_evolve_d2.hip, the optimiser never
// leaves the device: wave w of the workgroup builds the tensor of candidate w (wave 0 the point itself, 1 .. P  +h e_k,
// P + 1 .. 2P  -h e_k, then the backtracking points) - four lanes simulate the four columns of the 8 x 8 ansatz unitary - and
// eigen-solves its mixed transfer map, one complex 16 x 16 tile SQUARED on the matrix cores until it is rank one
// (qmps_overlap_d4.h: the code of overlap_square_d4_kernel); the 2 P + 1 + (n_alphas - 1) solves of a pass run side by side
// and cost the latency of one.  The optimiser loop itself (qmps_evolve_core.h) is the D = 2 kernel's, with workgroup barriers.
// The point itself and the backtracking points are eigen-solved; the 2 P central-difference neighbours are evaluated to second
// order in h from the point's right and left fixed points, as in the host driver (qmps_overlap_gradient) - eigen-solving all nine
// candidates of a pass cost 34 us (nine solves share the CU's four matrix pipes), one solve + eight contractions ~14.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_device.h"
#include "qmps_circuit.h"
#include "qmps_overlap_d4.h"
#include "qmps_evolve_core.h"

namespace qmps {

namespace {
constexpr int NWMAX = 8;         // waves per workgroup (two per SIMD: 256 registers each - the squaring, the circuit and the optimiser loop need ~170)
}

// the eigen-solve of one neighbour of a TIED point, out of line (the rare path of `evaluate` below: a second inlined copy of the squaring rounds would
// sit in the register budget of the common one)
struct TiedNeighbour { double eta_r, eta_i; int rounds, status; };
__device__ __attribute__((noinline)) TiedNeighbour solve_tied_neighbour(const double2* Ap, const double2* Bp, const double2* W, double2* sT, int max_rounds, double tol2) {
  TiedNeighbour o;
  v4f64 mr, mi;
  overlap_square_d4_item(Ap, Bp, W, sT, max_rounds, tol2, o.eta_r, o.eta_i, o.rounds, o.status, mr, mi);
  return o;
}

template <int KIND>
__global__ __launch_bounds__(64 * NWMAX) void evolve_bfgs_d4_kernel(EvolveD2Args p) {
  constexpr int PMAX = kEvolvePMax;
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, P = p.P, G1 = 2 * P + 1;
  __shared__ double sX[PMAX], sG[PMAX], sD[PMAX], sS[PMAX], sGn[PMAX], sHy[PMAX], sH[PMAX][PMAX + 1], sF[64];
  __shared__ int sOK[64];
  __shared__ double2 sA[32], sB[NWMAX][32], sT[NWMAX][kSquareD4Scratch];
  __shared__ double sCnt[4];
  __shared__ double2 sR[16], sY[16], sGs[64], sYR, sBm[2 * kEvolvePMax][64], sCSw[NWMAX][kEvolvePMax];
  const double2* W = (const double2*)p.WW;
  const double tol2 = p.tol * p.tol;
  if (tid < 4) sCnt[tid] = 0.0;
  // the four columns of the ansatz unitary for the parameter vector par(l), by lanes 0 .. 3 of the calling wave, as the tensor [2][4][4]
  // (round 6: the P sincos of the vector by P lanes of the wave, shared through LDS - the four column lanes used to compute all of them one after
  // the other, ~0.3 us each on the critical path of wave 0's solve)
  auto build_tensor = [&](double2* out, auto par) {
    __builtin_amdgcn_wave_barrier();
    if (lane < P) {
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(lane) * par(lane), &sn, &cs_);
      sCSw[wave][lane] = make_double2(cs_, sn);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 4) {
      Reg<3> r;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        r.re[x] = (x == lane) ? 1.0 : 0.0;
        r.im[x] = 0.0;
      }
      ansatz_circuit_cs<3, KIND>(r, [&](int l) { return sCSw[wave][l]; }, P);
#pragma unroll
      for (int x = 0; x < 8; ++x) out[((x & 1) * 4 + (x >> 1)) * 4 + lane] = make_double2(r.re[x], r.im[x]);      // A[s][i][j] = amplitude[2 i + s] of column j
    }
    __builtin_amdgcn_wave_barrier();
  };
  // one evaluation pass.  coef finite: candidates 0 .. 2P are the central-difference columns of z = x + coef d, then n_ladder
  // backtracking points; coef NaN: the n_ladder backtracking points only (they keep their candidate numbers G1 ..).
  // The point itself (wave 0) and the backtracking points are EIGEN-SOLVED (squaring on the matrix cores); the 2 P neighbours of
  // the point - they differ from it by h = 1e-6 - are evaluated to second order in h from its right and left fixed points (r, y: the
  // largest column and the conjugate of the largest row of the squared map),
  //     eta' = <y, T'(r)> / <y, r> = sum_s tr(Bm'_s^+ G_s) / <y, r> ,   G_s = y^+ C_s r   (qmps_overlap_grad.hip: the host driver's gradient),
  // one small contraction per neighbour instead of a solve: nine solves on four matrix pipes were what a pass cost (34 us).
#ifdef QMPS_D2_PHASES
  long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
  const long long k0 = wall_clock64();
#endif
  auto evaluate = [&](double coef, int n_ladder) {
    const bool with_grad = coef == coef;
#ifdef QMPS_D2_PHASES
    const long long t0 = wall_clock64();
    long long t1 = t0, t2 = t0, t3 = t0;
#endif
    const int NW = (int)(blockDim.x >> 6);
    // parameter vector of candidate `cand` (0: z = x + coef d; 1 + k / 1 + P + k: z +- h e_k; G1 + r: x + alphas[r + 1] d)
    auto par_of = [&](int cand) {
      const bool grad = cand < G1;
      const double a = grad ? coef : p.alphas[cand - G1 + 1];
      const int isel = (grad && cand > 0) ? (cand - 1) % P : -1;
      const double hs = cand <= P ? p.h : -p.h;
      return [=](int l) {
        double v = a != 0.0 ? evolve_detail::add_rn(sX[l], evolve_detail::mul_rn(a, sD[l])) : sX[l];
        if (l == isel) v = evolve_detail::add_rn(v, hs);
        return v;
      };
    };
    // Bm'_s = B'_s1 B'_s2 of the tensor in sB[wave]: lane = (s, i, j)
    auto merged_entry = [&]() {
      const int s_ = lane >> 4, i = (lane >> 2) & 3, j = lane & 3, s1 = s_ >> 1, s2 = s_ & 1;
      double2 bm = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < 4; ++k) cfma(sB[wave][(s1 * 4 + i) * 4 + k], sB[wave][(s2 * 4 + k) * 4 + j], bm);
      return bm;
    };
    // f of neighbour `cand` from its Bm' entry: eta' = sum conj(Bm') G / <y, r>
    auto probe = [&](int cand, double2 bm) {
      const double2 gg = sGs[lane], d = sYR;
      const double nr = wave_sum(dfma(bm.x, gg.x, bm.y * gg.y)), ni = wave_sum(dfma(bm.x, gg.y, -bm.y * gg.x));
      if (lane == 0) {
        const double den = d.x * d.x + d.y * d.y;
        const double er = (nr * d.x + ni * d.y) / den, ei = (ni * d.x - nr * d.y) / den;
        sF[cand] = -__builtin_sqrt(__builtin_sqrt(er * er + ei * ei));
        sOK[cand] = (sOK[0] == 1 && den > 1e-280) ? 1 : 0;      // (a TIED point has an objective but no fixed points to expand round: no gradient)
      }
    };
    const int solve_cand = with_grad ? (wave == 0 ? 0 : -1) : (wave < n_ladder ? G1 + wave : -1);
    if (solve_cand >= 0) {
      build_tensor(sB[wave], par_of(solve_cand));
#ifdef QMPS_D2_PHASES
      t1 = wall_clock64();
#endif
      double eta_r, eta_i;
      int rounds, status;
      v4f64 mr, mi;
      double2* sT0 = sT[wave];
      overlap_square_d4_item(sA, sB[wave], W, sT0, p.max_rounds, tol2, eta_r, eta_i, rounds, status, mr, mi);
#ifdef QMPS_D2_PHASES
      t2 = wall_clock64();
#endif
      if (lane == 0) {
        sF[solve_cand] = -__builtin_sqrt(__builtin_sqrt(eta_r * eta_r + eta_i * eta_i));
        // 1: eigen-solved; 2: dominant eigenvalues tied in modulus (QMPS_ST_TIED: the objective is their common modulus - usable, as at D = 2 -
        // but the power holds a mixture: the probes below give the neighbours no value - they are eigen-solved one by one at the end of the pass); 0: no answer
        sOK[solve_cand] = status == QMPS_ST_OK ? 1 : (overlap_usable(status) ? 2 : 0);
        atomicAdd(&sCnt[1], (double)rounds);
        if (!overlap_usable(status)) atomicAdd(&sCnt[2], 1.0);
      }
      if (with_grad) {
        // ---- r = largest column of M, y = conjugate of its largest row (M -> u v^+ ; any scale: eta' is a quotient)
        const int g = lane >> 4, c = lane & 15;
        double cn = 0.0, rn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double m2 = dfma(mr[q], mr[q], mi[q] * mi[q]);
          cn += m2;
          rn[q] = row16_sum(m2);
        }
        cn = group4_sum(cn);
        __builtin_amdgcn_wave_barrier();
        if (g == 0) sT0[c] = make_double2(cn, 0.0);
        if (c == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sT0[16 + 4 * q + g] = make_double2(rn[q], 0.0);
        }
        __builtin_amdgcn_wave_barrier();
        int bc = 0, br = 0;
        double bcn = -1.0, brn = -1.0;
        for (int k = 0; k < 16; ++k) {
          const double vc = sT0[k].x, vr = sT0[16 + k].x;
          if (vc > bcn) { bcn = vc; bc = k; }
          if (vr > brn) { brn = vr; br = k; }
        }
        __builtin_amdgcn_wave_barrier();
        if (c == bc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sR[4 * q + g] = make_double2(mr[q], mi[q]);
        }
        if (g == (br & 3)) {
          double vr = 0.0, vi = 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == (br >> 2)) { vr = mr[q]; vi = mi[q]; }
          sY[c] = make_double2(vr, -vi);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- Z_s = C_s r, G_s = y^+ Z_s  (lane = (s, i, j); C_s from the copy the solve kept), <y, r>
        const int s_ = lane >> 4, i = (lane >> 2) & 3, j = lane & 3;
        double2 z = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 4; ++k) cfma(sT0[kSquareD4Keep + 16 * s_ + 4 * i + k], sR[4 * k + j], z);
        sT0[64 + lane] = z;
        __builtin_amdgcn_wave_barrier();
        double2 gg = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double2 yk = sY[4 * k + i], zk = sT0[64 + 16 * s_ + 4 * k + j];
          gg.x = dfma(yk.x, zk.x, dfma(yk.y, zk.y, gg.x));          // conj(y) z
          gg.y = dfma(yk.x, zk.y, dfma(-yk.y, zk.x, gg.y));
        }
        sGs[lane] = gg;
        double ar = 0.0, ai = 0.0;
        if (lane < 16) {
          const double2 yv = sY[lane], rv = sR[lane];
          ar = yv.x * rv.x + yv.y * rv.y;
          ai = yv.x * rv.y - yv.y * rv.x;
        }
        ar = wave_sum(ar);
        ai = wave_sum(ai);
        if (lane == 0) sYR = make_double2(ar, ai);
      }
    }
    // the neighbours, spread over waves 1 .. NW - 1: tensors and merged entries BESIDE the solve of wave 0, the contractions behind it
    if (with_grad && wave > 0) {
      for (int nb = wave - 1; nb < 2 * P; nb += NW - 1) {
        __builtin_amdgcn_wave_barrier();
        build_tensor(sB[wave], par_of(1 + nb));
        sBm[nb][lane] = merged_entry();
      }
    }
#ifdef QMPS_D2_PHASES
    t3 = wall_clock64();
#endif
    __syncthreads();
#ifdef QMPS_D2_PHASES
    const long long t4 = wall_clock64();
#endif
    if (with_grad && wave > 0)
      for (int nb = wave - 1; nb < 2 * P; nb += NW - 1) probe(1 + nb, sBm[nb][lane]);
    __syncthreads();
    if (with_grad && sOK[0] == 2) {
      // the point is TIED (a non-injective state on the special grid: 1, 1, -1, -1): no fixed points to expand round, the probes above left the
      // neighbours without a value.  Every neighbour is eigen-solved instead - a tie's common modulus is an objective like any other, which is
      // what the reference's finite differences see (ARPACK returns one member of the tie) and what the D = 2 kernel does for every candidate:
      // the trajectory leaves the point instead of resting there.  (Uniform over the workgroup: sOK[0] is behind a barrier.)
      for (int nb = wave; nb < 2 * P; nb += NW) {
        build_tensor(sB[wave], par_of(1 + nb));
        const TiedNeighbour o = solve_tied_neighbour(sA, sB[wave], W, sT[wave], p.max_rounds, tol2);
        if (lane == 0) {
          sF[1 + nb] = -__builtin_sqrt(__builtin_sqrt(o.eta_r * o.eta_r + o.eta_i * o.eta_i));
          sOK[1 + nb] = overlap_usable(o.status) ? 1 : 0;
          atomicAdd(&sCnt[1], (double)o.rounds);
          if (!overlap_usable(o.status)) atomicAdd(&sCnt[2], 1.0);
        }
      }
      __syncthreads();
    }
#ifdef QMPS_D2_PHASES
    { const long long t5 = wall_clock64(); ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += 1; }
#endif
    if (tid == 0) sCnt[0] += (double)((with_grad ? G1 : 0) + n_ladder);
  };
  auto build_reference = [&]() {
    if (wave == 0) build_tensor(sA, [&](int l) { return sX[l]; });
    __syncthreads();
  };
  BfgsLds L;
  L.X = sX; L.G = sG; L.D = sD; L.S = sS; L.Gn = sGn; L.Hy = sHy; L.H = sH; L.F = sF; L.OK = sOK;
  __syncthreads();
  const bool ladder_in_pass = false;       // (a backtracking point costs a solve: only when the full step is rejected)
  bfgs_time_evolution(p, t, (wave == 0 && lane < P) ? lane : -1, tid == 0, L, evaluate, build_reference, [] { __syncthreads(); }, ladder_in_pass);
  __syncthreads();
#ifdef QMPS_D2_PHASES
  if (tid == 0 && p.prof != nullptr) {      // total | tensor | solve | extract | wait for the neighbours | probes | algebra (the rest) ... passes in slot 6, squarings in 7
    ph[6] = wall_clock64() - k0;
    p.prof[t * 8 + 0] = (double)ph[6]; p.prof[t * 8 + 1] = (double)ph[0]; p.prof[t * 8 + 2] = (double)ph[1]; p.prof[t * 8 + 3] = (double)ph[2];
    p.prof[t * 8 + 4] = (double)ph[3]; p.prof[t * 8 + 5] = (double)(ph[6] - ph[0] - ph[1] - ph[2] - ph[3] - ph[4]); p.prof[t * 8 + 6] = (double)ph[5]; p.prof[t * 8 + 7] = (double)ph[4] * 0.01;
  }
#endif
  if (tid == 0) {
    if (p.nfev != nullptr) p.nfev[t] = sCnt[0];
    if (p.rounds != nullptr) p.rounds[t] = sCnt[1];
    if (p.fail != nullptr) p.fail[t] = (int32_t)sCnt[2];
  }
}

hipError_t launch_evolve_bfgs_d4(int kind, const EvolveD2Args& a, hipStream_t st) {
  if (a.T <= 0) return hipSuccess;
  const int G = a.NA - 1;
  if (a.P < 1 || a.P > kEvolvePMax || a.NA < 1 || a.NA > kEvolveMaxAlphas || G > NWMAX || 2 * a.P + 1 + G > 64) return hipErrorInvalidValue;
  // eight waves: wave 0 solves the point (its neighbours are contractions, spread over all waves), up to eight backtracking points are solved side by side
  const dim3 grid((unsigned)a.T), block(64 * NWMAX);
  switch (kind) {
    case 0: hipLaunchKernelGGL(evolve_bfgs_d4_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(evolve_bfgs_d4_kernel<1>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(evolve_bfgs_d4_kernel<3>, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace qmps
