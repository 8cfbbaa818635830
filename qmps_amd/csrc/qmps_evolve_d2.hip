// qmps_evolve_d2.hip - the WHOLE time evolution of the reference's own case (bond dimension D = 2) in one launch (gfx950 only).
//
// Reference: qmps/new_time_evolve.py:276-292 (`minimize(obj, params, (A_, WW))` per time step, scipy BFGS with finite-difference
// gradients, ShallowFullStateTensor(2, .) with 15 angles, :186-187), scripts/loschmidt.py:367-375 (ShallowCNOTStateTensor(2, .)).
// The lock-step drivers (tools.batched_bfgs, qmps_evolve_bfgs) pay a host round trip per BFGS iteration and make every trajectory
// wait for the slowest one; at D = 2 the GPU then idles (profiles/r03h_evolve_d2_t256.json: 0.03 ms of kernels inside 5 ms).  Here
// ONE WORKGROUP (one to three waves) owns a trajectory for the whole run - every time step, every BFGS iteration, without returning to the host:
//   QUADS of lanes = candidates of an evaluation pass: candidate 0 the iterate z, 1 .. P  z + h e_k, P + 1 .. 2P  z - h e_k (the
//   central-difference columns), 2P + 1 .. 2P + G the backtracking points x + alpha_r d - two lanes of the quad simulate the two
//   columns of the candidate's 4 x 4 ansatz unitary (qmps_circuit.h), then the four lanes eigen-solve its mixed transfer map, a row
//   of the 4 x 4 complex matrix each (squared until converged: the algorithm of overlap_lane_kernel / qmps_overlap_d2.h);
//   x, g, d, s, the inverse Hessian H (P x P; thread a owns row a) and the reference tensor live in the workgroup's LDS
//   (one to three waves per trajectory).  First version: ONE lane per candidate - 25 of the 31 us of a pass were that lane's solve.
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


// ---- the eigen-solve of one candidate by a QUAD of lanes (lane q owns row q = (i, i') of the 4 x 4 complex matrix
//   E[(i,i'),(j,j')] = sum_s C_s[i][j] conj(Bm_s[i'][j']),  Bm = merge(B, B)
// of the mixed transfer map; qmps_overlap_d2.h is the same algorithm in ONE lane - 25 of the 31 us of an evaluation pass were that lane's
// dependent chain of ~6 squarings, here a squaring is 16 complex multiply-adds per lane and the rows travel by quad broadcasts).
//   amp_r / amp_i: lanes 0, 1 of the quad hold the four amplitudes of column 0 / 1 of the candidate's unitary (B[s][i][k] = amplitude 2 i + s of column k)
//   sC: C_s[i][j] of the trajectory's reference tensor, [s][i][j] in LDS.  Results uniform over the quad.
__device__ __forceinline__ void overlap_quad_build(const double2* sC, const double (&amp_r)[4], const double (&amp_i)[4], int q, double (&er)[4], double (&ei)[4]) {
  const int i = q >> 1, ip = q & 1;
  // the candidate's tensor by quad broadcasts: B[s][i][k] = amplitude 2 i + s of column k (lane k).  (No array of broadcast values:
  // a select between two elements of a private array becomes a dynamically indexed load - scratch.)
  auto Bre = [&](int s, int i_, int k) { return k == 0 ? quad_bcast<0>(amp_r[2 * i_ + s]) : quad_bcast<1>(amp_r[2 * i_ + s]); };
  auto Bim = [&](int s, int i_, int k) { return k == 0 ? quad_bcast<0>(amp_i[2 * i_ + s]) : quad_bcast<1>(amp_i[2 * i_ + s]); };
  // row a = (i, i') of E
#pragma unroll
  for (int c = 0; c < 4; ++c) er[c] = ei[c] = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int s1 = s >> 1, s2 = s & 1;
    // Bm_s[i'][j'] = sum_k B[s1][i'][k] B[s2][k][j']
    double mr_[2], mi_[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      double xr = 0.0, xi = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const double u1r = Bre(s1, 1, k), u0r = Bre(s1, 0, k), u1i = Bim(s1, 1, k), u0i = Bim(s1, 0, k);
        const double ur = ip ? u1r : u0r, ui = ip ? u1i : u0i;
        const double kr = Bre(s2, k, jp), ki = Bim(s2, k, jp);
        xr = dfma(ur, kr, dfma(-ui, ki, xr));
        xi = dfma(ur, ki, dfma(ui, kr, xi));
      }
      mr_[jp] = xr;
      mi_[jp] = xi;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double2 cv = sC[(s * 2 + i) * 2 + j];
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {       // += C conj(Bm)
        er[2 * j + jp] = dfma(cv.x, mr_[jp], dfma(cv.y, mi_[jp], er[2 * j + jp]));
        ei[2 * j + jp] = dfma(cv.y, mr_[jp], dfma(-cv.x, mi_[jp], ei[2 * j + jp]));
      }
    }
  }
}

// the dominant eigenvalue of the quad's 4 x 4 map (row q in er / ei) by SQUARING (see above; QMPS_EVOLVE_D2_SQUARING selects it since round 6)
__device__ __forceinline__ void overlap_quad_solve(const double (&er)[4], const double (&ei)[4], int q, int max_rounds, double tol,
                                                   double& eta_r, double& eta_i, int& rounds, int& status) {
  double mr[4], mi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { mr[c] = er[c]; mi[c] = ei[c]; }
  const double tol2 = tol * tol;
  eta_r = eta_i = 0.0;
  status = QMPS_ST_NOT_CONVERGED;
  rounds = 0;
  bool done = false, collapsed = false;
  // row a of M M = sum_k M[a][k] row_k(M), and its squared Frobenius norm
  auto square = [&](const double (&xr)[4], const double (&xi)[4], double (&nr)[4], double (&ni)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) nr[c] = ni[c] = 0.0;
#define QMPS_ROW(K)                                                                                            \
    {                                                                                                          \
      const double ar = xr[K], ai = xi[K];                                                                     \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
        const double kr = quad_bcast<K>(xr[c]), ki = quad_bcast<K>(xi[c]);                                     \
        nr[c] = dfma(ar, kr, dfma(-ai, ki, nr[c]));                                                            \
        ni[c] = dfma(ar, ki, dfma(ai, kr, ni[c]));                                                             \
      }                                                                                                        \
    }
    QMPS_ROW(0) QMPS_ROW(1) QMPS_ROW(2) QMPS_ROW(3)
#undef QMPS_ROW
    double f = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) f = dfma(nr[c], nr[c], dfma(ni[c], ni[c], f));
    return quad_sum(f);
  };
  constexpr int kFirstTest = 3;
  double m2 = 0.0;      // ||M||_F^2: computed for the first power, 1 afterwards
#pragma unroll
  for (int c = 0; c < 4; ++c) m2 = dfma(mr[c], mr[c], dfma(mi[c], mi[c], m2));
  m2 = quad_sum(m2);
  for (int m = 0; m <= max_rounds; ++m) {
    double nr[4], ni[4], f2 = 0.0;
    if (!done) {
      // the square first: the next power, and the test for a COLLAPSED power (a nilpotent map: rounding noise) it carries
      f2 = square(mr, mi, nr, ni);
      rounds = m;
      if (f2 < 1e-28 * m2 * m2) {
        if (m <= 8) { eta_r = 0.0; eta_i = 0.0; status = QMPS_ST_OK; }
        collapsed = true;
        done = true;
      } else if (m >= kFirstTest || m == max_rounds) {
        // (rounds 0 .. kFirstTest - 1 skip the eigen-residual test - 135 of a round's 420 instructions: a map of the time-evolution
        // objective needs 5 - 10 squarings at tol 1e-12; one that would have passed earlier squares on to round kFirstTest, harmlessly)
        // dominant right vector = largest column of the current power; v[a] in lane a
        int bc = 0;
        double best = -1.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double n2 = quad_sum(dfma(mr[c], mr[c], mi[c] * mi[c]));
          if (n2 > best) { best = n2; bc = c; }
        }
        double vr = 0.0, vi = 0.0;          // (a sum of selects against zero: a select between array elements becomes a dynamic index - scratch)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          vr += bc == c ? mr[c] : 0.0;
          vi += bc == c ? mi[c] : 0.0;
        }
        // w = E v, eta = <v, w>/<v, v>, residual
        double wr = 0.0, wi = 0.0;
        {
          const double v0r = quad_bcast<0>(vr), v0i = quad_bcast<0>(vi), v1r = quad_bcast<1>(vr), v1i = quad_bcast<1>(vi);
          const double v2r = quad_bcast<2>(vr), v2i = quad_bcast<2>(vi), v3r = quad_bcast<3>(vr), v3i = quad_bcast<3>(vi);
          wr = dfma(er[0], v0r, dfma(-ei[0], v0i, wr)); wi = dfma(er[0], v0i, dfma(ei[0], v0r, wi));
          wr = dfma(er[1], v1r, dfma(-ei[1], v1i, wr)); wi = dfma(er[1], v1i, dfma(ei[1], v1r, wi));
          wr = dfma(er[2], v2r, dfma(-ei[2], v2i, wr)); wi = dfma(er[2], v2i, dfma(ei[2], v2r, wi));
          wr = dfma(er[3], v3r, dfma(-ei[3], v3i, wr)); wi = dfma(er[3], v3i, dfma(ei[3], v3r, wi));
        }
        const double num_r = quad_sum(dfma(vr, wr, vi * wi)), num_i = quad_sum(dfma(vr, wi, -vi * wr)), vv = quad_sum(dfma(vr, vr, vi * vi));
        // (v_rcp_f64 + one Newton step instead of two IEEE divisions: ~25 dependent instructions off the critical path of every round; 1e-16 relative)
        const double ivv = fast_rcp(vv);
        eta_r = num_r * ivv;
        eta_i = num_i * ivv;
        const double dr = wr - (eta_r * vr - eta_i * vi), di = wi - (eta_r * vi + eta_i * vr);
        const double res = quad_sum(dfma(dr, dr, di * di));
        // ... and, only looked at once the column passes: is the power RANK ONE (at symmetric points a column of an early power is an exact
        // eigenvector of a LESSER eigenvalue: overlap_lane_solve, qmps_overlap_d2.h)?
        if (res < tol2 * vv) {
          double dgr = 0.0, dgi = 0.0;      // tr(M): the diagonal element of row q is column q
#pragma unroll
          for (int c = 0; c < 4; ++c) { dgr += q == c ? mr[c] : 0.0; dgi += q == c ? mi[c] : 0.0; }
          const double trr = quad_sum(dgr), tri = quad_sum(dgi);
          double r1 = 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double xr = nr[c] - (trr * mr[c] - tri * mi[c]), xi = ni[c] - (trr * mi[c] + tri * mr[c]);
            r1 = dfma(xr, xr, dfma(xi, xi, r1));
          }
          if (quad_sum(r1) < 1e-20 * f2) { status = QMPS_ST_OK; done = true; }
        }
      }
      if (m == max_rounds) done = true;
    }
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;        // (the quads of a wave leave together)
    if (!done) {
      // Frobenius-normalise the square - to rounding: the scale only keeps the powers O(1), v_rsq_f64 + one Newton step (1e-16) replaces a
      // square root and a division (~30 dependent instructions per round)
      const double inv = fast_rsqrt(f2);
#pragma unroll
      for (int c = 0; c < 4; ++c) { mr[c] = nr[c] * inv; mi[c] = ni[c] * inv; }
      m2 = 1.0;
    }
  }
  if (status != QMPS_ST_OK && !collapsed && rounds >= 30) {
    // tied dominant eigenvalues (overlap_lane_solve): their common modulus from the norms of the squared powers, ||E^(2^m)||^(1/2^m) - the rare
    // path, so the logarithms live here, in a second pass over the squarings, and not in the loop above
    double xr[4], xi[4], nr[4], ni[4], n2 = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { xr[c] = er[c]; xi[c] = ei[c]; n2 = dfma(er[c], er[c], dfma(ei[c], ei[c], n2)); }
    n2 = quad_sum(n2);
    double log_rho = 0.5 * log(n2);
    const double inv0 = 1.0 / __builtin_sqrt(n2);
#pragma unroll
    for (int c = 0; c < 4; ++c) { xr[c] *= inv0; xi[c] *= inv0; }
    for (int m = 0; m < 44; ++m) {
      const double f2 = square(xr, xi, nr, ni);
      if (!(f2 > 1e-280)) break;
      log_rho += ldexp(0.5 * log(f2), -(m + 1));
      const double inv = 1.0 / __builtin_sqrt(f2);
#pragma unroll
      for (int c = 0; c < 4; ++c) { xr[c] = nr[c] * inv; xi[c] = ni[c] * inv; }
    }
    eta_r = exp(log_rho);
    eta_i = 0.0;
    status = QMPS_ST_OK;        // (internal to this kernel, where status 0 means "the objective is usable": the lane solver reports QMPS_ST_TIED)
  }
}

// ---- the same eigenvalue WITHOUT iterating on the map: the largest root of its characteristic polynomial (round 6).
// The squaring solve needs log2(28 / gap) rounds - 6 for a typical map of the time-evolution objective, 10 - 11 for the trajectory with the smallest
// spectral gap, and the slowest trajectory IS a launch of this kernel (phase timers: 8.1 of its 13.1 us per pass).  A 4 x 4 map has a quartic:
//   p_k = tr E^k, k = 1 .. 4, from ONE product (E^2 by quad broadcasts; tr E^3 = sum E^2[q][c] E[c][q], tr E^4 = sum E^2[q][c] E^2[c][q]: the transposed
//   elements through the quad's LDS scratch), Newton's identities -> e_1 .. e_4, P(z) = z^4 - e_1 z^3 + e_2 z^2 - e_3 z + e_4;
//   its four roots by the Aberth - Ehrlich iteration, ONE ROOT PER LANE of the quad (the other three by quad rotations): cubic convergence on simple
//   roots, 4 - 8 iterations on the maps of the time-evolution objective whatever the gap (starting points: see below);
//   eta = the root of largest modulus.
// What the objective -sqrt|eta| needs and nothing more: no eigenvector (the lane solver of the C-ABI's D = 2 overlap entry points keeps the squaring:
// it owes its callers r_out).  Tied moduli (a complex-conjugate pair, a ring) are distinct simple roots - no special path.  Conditioning: the
// coefficients carry ~1e-16 |eta|^k, a simple dominant root moves by 1e-16 |eta| / prod_j |1 - eta_j / eta| (2e-15 at a gap of 0.05) - but an m-fold
// dominant root by eps^(1/m) of its modulus (1e-8, 5e-6, 1e-4) and the iteration converges only linearly on it, and a nilpotent map's P = z^4 + noise
// has roots of 1e-4 ||E||.  Those maps - points of the special grid (multiples of pi / 4: product states, permutation-like tensors), never met
// by a generic trajectory - and every largest root the quartic knows badly (kappa = prod_j |1 - eta_j / eta| < 1e-6) are NOT answered here: `fallback`
// comes back set (that, or a largest root below 1e-3 ||E||_F) and the caller runs the squaring solve above on the quad, whose Gelfand route and collapse test give them to 1e-11.
// E is scaled to unit Frobenius norm first.  sT: 32 double2 of LDS per quad.  rounds = Aberth iterations.  warm (nullable, uniform over the workgroup):
// four eigenvalues of a NEIGHBOURING map to start from (3 iterations instead of 5 - 8); root: this lane's eigenvalue at the end (NaN in lane q's .x when two
// of the four coincide: no starting points for anybody).
__device__ __forceinline__ void overlap_quad_charpoly(const double (&er)[4], const double (&ei)[4], int q, double2* sT, const double2* warm, double& eta_r, double& eta_i,
                                                      int& rounds, int& status, bool& fallback, double2& root) {
  status = QMPS_ST_OK;
  fallback = false;
  rounds = 0;
  eta_r = eta_i = 0.0;
  double m2 = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) m2 = dfma(er[c], er[c], dfma(ei[c], ei[c], m2));
  m2 = quad_sum(m2);
  const bool zero = !(m2 > 1e-300);                 // (uniform over the quad; a NaN map falls through and ends as NaN)
  const double sc = zero ? 0.0 : fast_rsqrt(m2);
  double xr[4], xi[4], nr[4], ni[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { xr[c] = er[c] * sc; xi[c] = ei[c] * sc; }
  // row q of X X (X = E / ||E||_F)
#pragma unroll
  for (int c = 0; c < 4; ++c) nr[c] = ni[c] = 0.0;
#define QMPS_ROW(K)                                                                                            \
  {                                                                                                            \
    const double ar = xr[K], ai = xi[K];                                                                       \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                            \
      const double kr = quad_bcast<K>(xr[c]), ki = quad_bcast<K>(xi[c]);                                       \
      nr[c] = dfma(ar, kr, dfma(-ai, ki, nr[c]));                                                              \
      ni[c] = dfma(ar, ki, dfma(ai, kr, ni[c]));                                                               \
    }                                                                                                          \
  }
  QMPS_ROW(0) QMPS_ROW(1) QMPS_ROW(2) QMPS_ROW(3)
#undef QMPS_ROW
  // columns through the quad's scratch (wave-private: LDS is in order per wave)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    sT[q * 4 + c] = make_double2(xr[c], xi[c]);
    sT[16 + q * 4 + c] = make_double2(nr[c], ni[c]);
  }
  __builtin_amdgcn_wave_barrier();
  double p1r, p1i, p2r, p2i, p3r = 0.0, p3i = 0.0, p4r = 0.0, p4i = 0.0;
  {
    const double2 d1 = sT[q * 5], d2 = sT[16 + q * 5];        // the diagonal elements of row q
    p1r = quad_sum(d1.x); p1i = quad_sum(d1.y);
    p2r = quad_sum(d2.x); p2i = quad_sum(d2.y);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double2 a = sT[c * 4 + q], b = sT[16 + c * 4 + q];   // X[c][q], X^2[c][q]
      p3r = dfma(nr[c], a.x, dfma(-ni[c], a.y, p3r)); p3i = dfma(nr[c], a.y, dfma(ni[c], a.x, p3i));
      p4r = dfma(nr[c], b.x, dfma(-ni[c], b.y, p4r)); p4i = dfma(nr[c], b.y, dfma(ni[c], b.x, p4i));
    }
    p3r = quad_sum(p3r); p3i = quad_sum(p3i); p4r = quad_sum(p4r); p4i = quad_sum(p4i);
  }
  __builtin_amdgcn_wave_barrier();
  // Newton's identities: e1 = p1, 2 e2 = e1 p1 - p2, 3 e3 = e2 p1 - e1 p2 + p3, 4 e4 = e3 p1 - e2 p2 + e1 p3 - p4
  auto cmul_r = [](double ar, double ai, double br, double bi) { return dfma(ar, br, -ai * bi); };
  auto cmul_i = [](double ar, double ai, double br, double bi) { return dfma(ar, bi, ai * br); };
  const double e1r = p1r, e1i = p1i;
  const double e2r = 0.5 * (cmul_r(e1r, e1i, p1r, p1i) - p2r), e2i = 0.5 * (cmul_i(e1r, e1i, p1r, p1i) - p2i);
  const double e3r = (cmul_r(e2r, e2i, p1r, p1i) - cmul_r(e1r, e1i, p2r, p2i) + p3r) * (1.0 / 3.0);
  const double e3i = (cmul_i(e2r, e2i, p1r, p1i) - cmul_i(e1r, e1i, p2r, p2i) + p3i) * (1.0 / 3.0);
  const double e4r = 0.25 * (cmul_r(e3r, e3i, p1r, p1i) - cmul_r(e2r, e2i, p2r, p2i) + cmul_r(e1r, e1i, p3r, p3i) - p4r);
  const double e4i = 0.25 * (cmul_i(e3r, e3i, p1r, p1i) - cmul_i(e2r, e2i, p2r, p2i) + cmul_i(e1r, e1i, p3r, p3i) - p4i);
  // P(z) and P'(z) by Horner
  auto poly = [&](double zr, double zi, double& Pr, double& Pi, double& dr, double& di) {
    double ar = zr - e1r, ai = zi - e1i;                                   // z - e1
    double br = 4.0 * zr - 3.0 * e1r, bi = 4.0 * zi - 3.0 * e1i;           // 4 z - 3 e1
    double tr = cmul_r(ar, ai, zr, zi) + e2r, ti = cmul_i(ar, ai, zr, zi) + e2i;
    double ur = cmul_r(br, bi, zr, zi) + 2.0 * e2r, ui = cmul_i(br, bi, zr, zi) + 2.0 * e2i;
    ar = cmul_r(tr, ti, zr, zi) - e3r; ai = cmul_i(tr, ti, zr, zi) - e3i;
    dr = cmul_r(ur, ui, zr, zi) - e3r; di = cmul_i(ur, ui, zr, zi) - e3i;
    Pr = cmul_r(ar, ai, zr, zi) + e4r; Pi = cmul_i(ar, ai, zr, zi) + e4i;
  };
  // ---- starting points.  Lane 0: the power-sum quotient p4 / p3 (the dominant root up to O((eta_2 / eta_1)^3) when there is one); lanes 1 - 3:
  // round the centroid c3 of the cubic Q that is left when that estimate is divided out, at the scale |Q(c3)|^(1/3).  When p3 is too small for the
  // quotient to mean anything (a spectrum symmetric under z -> -z ...): all four round c = e1 / 4 at the scale |P(c)|^(1/4).  Two traps of the
  // textbook circle, both met on structured maps: (i) a centroid that IS a root (spectrum {l, -l, 0, 0}: P(c) = 0) puts every point ON it, where
  // coincident points move together for ever - the scale falls back on the next coefficients of the polynomial shifted to the centroid;
  // (ii) points on a regular polygon keep the polygon's symmetry when the polynomial has it (z^4 + a z^2 + b from a square: 30 iterations of a
  // rotating square until rounding breaks it) - the points sit at unequal radii and unequal angles.
  // (`warm`, uniform over the workgroup: the four eigenvalues the caller's LAST solve of a neighbouring map ended with - the candidates of a BFGS
  // pass sit a step away from the point of the pass before: 2 - 4 iterations from there where the starting points below need 5 - 8, and the
  // slowest quad of a wave is what a pass waits for)
  double zr, zi;
  if (warm != nullptr) {
    const double2 w = warm[q];
    zr = w.x * sc;
    zi = w.y * sc;
  } else {
    const double n3 = dfma(p3r, p3r, p3i * p3i), n4 = dfma(p4r, p4r, p4i * p4i);
    const bool informed = n3 > 1e-12 && n4 < 4.0 * n3;              // (|p4 / p3| <= 2 in units of ||E||_F: a quotient inside the spectrum's disc)
    const double i3 = informed ? fast_rcp(n3) : 0.0;
    const double ar = dfma(p4r, p3r, p4i * p3i) * i3, ai = dfma(p4i, p3r, -p4r * p3i) * i3;      // p4 / p3
    // Q(z) = z^3 + b2 z^2 + b1 z + b0 = P(z) / (z - a), approximately
    const double b2r = ar - e1r, b2i = ai - e1i;
    const double b1r = e2r + cmul_r(ar, ai, b2r, b2i), b1i = e2i + cmul_i(ar, ai, b2r, b2i);
    const double b0r = -e3r + cmul_r(ar, ai, b1r, b1i), b0i = -e3i + cmul_i(ar, ai, b1r, b1i);
    const double c3r = -b2r * (1.0 / 3.0), c3i = -b2i * (1.0 / 3.0);
    double Qr, Qi;                                                    // Q(c3 + w) = w^3 + A3 w + Q(c3)
    {
      const double tr = c3r + b2r, ti = c3i + b2i;
      const double ur = cmul_r(tr, ti, c3r, c3i) + b1r, ui = cmul_i(tr, ti, c3r, c3i) + b1i;
      Qr = cmul_r(ur, ui, c3r, c3i) + b0r;
      Qi = cmul_i(ur, ui, c3r, c3i) + b0i;
    }
    const double c4r = 0.25 * e1r, c4i = 0.25 * e1i;
    double Pr, Pi, dr, di;                                            // P(c4 + w) = w^4 + A4 w^2 + P'(c4) w + P(c4)
    poly(c4r, c4i, Pr, Pi, dr, di);
    // the scale of the starting points: |Q(c3)|^(1/3) or |P(c4)|^(1/4) - a GUESS: single-precision log2 / exp2 (two instructions; the double-precision
    // cbrt and the chain of three square roots were ~150 of a solve's ~1 100), x = the SQUARED modulus; below 1e-38 it reads as zero: trap (i)
    auto root_of = [](double x, float half_exponent) { return (double)__builtin_amdgcn_exp2f(half_exponent * __builtin_amdgcn_logf((float)x)); };
    double rad3 = 0.0, rad4 = 0.0;
    {
      const double rad = root_of(informed ? dfma(Qr, Qr, Qi * Qi) : dfma(Pr, Pr, Pi * Pi), informed ? (1.0f / 6.0f) : 0.125f);
      rad3 = rad4 = rad;
    }
    if (rad3 < 1e-3) {                                                // (trap (i): rare, uniform over the quad)
      double tr = 6.0 * c4r - 3.0 * e1r, ti = 6.0 * c4i - 3.0 * e1i;
      const double A4r = cmul_r(tr, ti, c4r, c4i) + e2r, A4i = cmul_i(tr, ti, c4r, c4i) + e2i;
      tr = 3.0 * c3r + 2.0 * b2r; ti = 3.0 * c3i + 2.0 * b2i;
      const double A3r = cmul_r(tr, ti, c3r, c3i) + b1r, A3i = cmul_i(tr, ti, c3r, c3i) + b1i;
      rad4 = fmax(rad4, fmax(root_of(dfma(A4r, A4r, A4i * A4i), 0.25f), root_of(dfma(dr, dr, di * di), 1.0f / 6.0f)));
      rad3 = fmax(rad3, root_of(dfma(A3r, A3r, A3i * A3i), 0.25f));
    }
    // four points: radii (1, 0.8, 1.25, 0.9) at the angles (0.7, 2.1, 4.0, 5.3); three points (lanes 1 - 3): radii (1, 0.8, 1.2) at (0.7, 2.6, 4.9)
    const double u4x = q == 0 ? 0.7648421872844885 : (q == 1 ? -0.40387688367988606 : (q == 2 ? -0.8170545260795149 : 0.4989369025612447));
    const double u4y = q == 0 ? 0.644217687237691 : (q == 1 ? 0.6905674933190991 : (q == 2 ? -0.9460031191349103 : -0.7490406980015112));
    const double u3x = q == 1 ? 0.7648421872844885 : (q == 2 ? -0.6855110026951579 : 0.22381484330709092);
    const double u3y = q == 1 ? 0.644217687237691 : (q == 2 ? 0.41240109745717135 : -1.178943135149199);
    if (informed) {
      zr = q == 0 ? ar : dfma(rad3, u3x, c3r);
      zi = q == 0 ? ai : dfma(rad3, u3y, c3i);
    } else {
      zr = dfma(rad4, u4x, c4r);
      zi = dfma(rad4, u4y, c4i);
    }
  }
  // ---- Aberth - Ehrlich.  Only the root of LARGEST modulus has to be converged, the others only far enough to be ranked below it: the iteration
  // of a quad is over when its largest root has settled (step < 1e-14 of its modulus) and every other root either has settled too (tied moduli) or
  // HAS LOCALISED a root below the largest one: a step under 1e-2 of the largest modulus (cubic convergence: the root is within that step) and,
  // with four times that step, inside the largest one's circle.  (Small roots settle slowly in RELATIVE terms - a root at zero never - and are
  // not wanted; but a point that is merely PASSING at small modulus with a large step may be on its way to the dominant root nobody has found
  // yet, and is not "below" anything.)
  bool fin = false;
  double prev_s2 = 1e300;
  for (int it = 0; it < 40; ++it) {
    double Pr, Pi, dr, di;
    poly(zr, zi, Pr, Pi, dr, di);
    const double dn = dfma(dr, dr, di * di);
    const double idn = dn > 0.0 ? fast_rcp(dn) : 0.0;
    const double wr = dfma(Pr, dr, Pi * di) * idn, wi = dfma(Pi, dr, -Pr * di) * idn;      // P / P'
    double sr = 0.0, si = 0.0;                                                             // sum_j 1 / (z - z_j) over the other three lanes
#define QMPS_OTHER(CTRL)                                                                                       \
    {                                                                                                          \
      const double xr_ = zr - quad_perm<CTRL>(zr), xi_ = zi - quad_perm<CTRL>(zi);                            \
      const double n_ = dfma(xr_, xr_, xi_ * xi_);                                                             \
      const double in_ = n_ > 0.0 ? fast_rcp(n_) : 0.0;                                                       \
      sr = dfma(xr_, in_, sr);                                                                                 \
      si = dfma(-xi_, in_, si);                                                                                \
    }
    QMPS_OTHER(0x39) QMPS_OTHER(0x4E) QMPS_OTHER(0x93)        // quad_perm [1,2,3,0], [2,3,0,1], [3,0,1,2]
#undef QMPS_OTHER
    const double gr = 1.0 - cmul_r(wr, wi, sr, si), gi = -cmul_i(wr, wi, sr, si);
    const double gn = dfma(gr, gr, gi * gi);
    const double ign = gn > 0.0 ? fast_rcp(gn) : 0.0;
    const double stepr = dfma(wr, gr, wi * gi) * ign, stepi = dfma(wi, gr, -wr * gi) * ign;
    if (!fin) {
      zr -= stepr;
      zi -= stepi;
      rounds = it + 1;
    }
    const double s2 = dfma(stepr, stepr, stepi * stepi), z2 = dfma(zr, zr, zi * zi);
    // the quad's largest modulus; this lane is fine when it has settled, or when it is safely below the largest
    const double zmax = fmax(fmax(z2, quad_perm<0xB1>(z2)), fmax(quad_perm<0x4E>(z2), quad_perm<0x1B>(z2)));
    // settled: a step below 1e-14 of the modulus - or, for a root whose neighbours leave P / P' no such accuracy (a cluster: the noise of the step is
    // ~1e-16 / gap), a step that has stopped shrinking (super-linear convergence ends in the noise floor) below 1e-10 of the modulus
    const bool settled = !(s2 > 1e-28 * z2 + 1e-300) || (s2 < 1e-20 * z2 && s2 > 0.04 * prev_s2);
    prev_s2 = s2;
    // (|z| + 4 |step|)^2 < zmax without the two square roots - ~50 instructions of an iteration's ~170:  8 |z||step| < zmax - z2 - 16 s2, squared
    const double room = zmax - z2 - 16.0 * s2;
    const bool below = room > 0.0 && 64.0 * z2 * s2 < room * room && s2 < 1e-4 * zmax;
    const bool ok = settled || below;
    const double okall = quad_sum(ok ? 0.0 : 1.0);
    fin = fin || okall == 0.0;
    if (it >= 9 && !fin) {
      // (simple, well separated roots are done by now) the largest root is one the quartic knows badly - a multiple root or a near-cluster:
      // linear convergence, eps / kappa accuracy (see the end of this function; |P'(z)| = kappa |z|^3 at a root): stop, the caller's squaring
      // solve answers it
      const bool top = z2 == zmax;
      const double dtop = quad_sum(top ? dn : 0.0), tn = quad_sum(top ? 1.0 : 0.0);
      if (dtop < 1e-12 * zmax * zmax * zmax * tn) { fin = true; fallback = true; }
    }
    if (__builtin_amdgcn_ballot_w64(!fin) == 0) break;        // (every quad of the wave is done)
  }
  // (the cap is no exit: a quad that has not finished after 40 iterations - none did in the emulation's 30 000 maps or the campaigns, cold or from
  // the last pass's eigenvalues - has no largest root to report: the squaring solve answers it)
  if (!fin) fallback = true;
  // the root of largest modulus, in every lane of the quad (the first of equal ones)
  double br_ = zr, bi_ = zi, bm = dfma(zr, zr, zi * zi);
  int kx = 0;      // the lane that holds it
  {
    const double m0 = quad_bcast<0>(bm), r0 = quad_bcast<0>(br_), i0 = quad_bcast<0>(bi_);
    double Mx = m0, Rx = r0, Ix = i0;
    const double m1 = quad_bcast<1>(bm), r1 = quad_bcast<1>(br_), i1 = quad_bcast<1>(bi_);
    if (m1 > Mx) { Mx = m1; Rx = r1; Ix = i1; kx = 1; }
    const double m2_ = quad_bcast<2>(bm), r2 = quad_bcast<2>(br_), i2 = quad_bcast<2>(bi_);
    if (m2_ > Mx) { Mx = m2_; Rx = r2; Ix = i2; kx = 2; }
    const double m3 = quad_bcast<3>(bm), r3 = quad_bcast<3>(br_), i3 = quad_bcast<3>(bi_);
    if (m3 > Mx) { Mx = m3; Rx = r3; Ix = i3; kx = 3; }
    br_ = Rx; bi_ = Ix; bm = Mx;
  }
  // How well does the quartic know its largest root?  A simple root moves by eps / kappa, kappa = |P'(z)| / |z|^3 = prod_j |1 - z_j / z|, when the
  // coefficients move by eps: 1e-15 at the gaps of a generic trajectory - but four eigenvalues within 1e-3 of each other (a trajectory one BFGS step
  // off the special grid: 0.93408, 0.93311, 0.93311, 0.93213) have kappa = 2e-9 and came back 1.4e-7 off, none of them "within 1e-3 of the largest".
  // kappa < 1e-6 (a multiple root: 0; what passes is good to 1e-10), or a largest root below 1e-3 ||E||_F (a nilpotent map's noise): the squaring
  // solve's cases.  (1e-4 handed back the whole early evolution of ONE of 256 generic trajectories - near-cluster after near-cluster - and the
  // slowest trajectory is the launch: +10 % on the first ten time steps.)
  {
    double prod2 = 1.0, near2 = 1e300;
#define QMPS_OTHER(CTRL)                                                                                       \
    {                                                                                                          \
      const double xr_ = zr - quad_perm<CTRL>(zr), xi_ = zi - quad_perm<CTRL>(zi);                            \
      const double d2_ = dfma(xr_, xr_, xi_ * xi_);                                                            \
      prod2 *= d2_;                                                                                            \
      near2 = fmin(near2, d2_);                                                                                \
    }
    QMPS_OTHER(0x39) QMPS_OTHER(0x4E) QMPS_OTHER(0x93)
#undef QMPS_OTHER
    // the value of the lane that holds the largest root
    const bool pick = q == kx;
    const double kap2 = quad_sum(pick ? prod2 : 0.0), nr2 = quad_sum(pick ? near2 : 0.0);
    // (and a root within 1e-3 of it: the copies of a noise-split MULTIPLE root sit eps^(1/m) apart - their kappa is itself noise, 1.2e-6 on a
    // double root of modulus 6e-4 ||E||)
    if (kap2 < 1e-12 * bm * bm * bm || nr2 < 1e-6 * bm || bm < 1e-6) fallback = true;
    // (two of the four computed roots coincide - to 1e-10 of the largest: coincident starting points move together for ever, so these four are
    // no starting points for a later solve: NaN tells the caller)
    if (quad_sum(near2 > 1e-20 * bm ? 0.0 : 1.0) != 0.0) zr = __builtin_nan("");      // (also a zero map: four zeros)
  }
  const double back = zero ? 0.0 : __builtin_sqrt(m2);        // undo the scaling
  eta_r = br_ * back;
  eta_i = bi_ * back;
  root = make_double2(zr * back, zi * back);                   // this lane's eigenvalue: a starting point for the caller's next solve
  if (zero) fallback = false;                                  // (E = 0: eta = 0, nothing to square)
  // (the cap of 40 iterations is never the exit of a simple largest root; NaN - a NaN map - is not an answer)
  if (!(eta_r == eta_r && eta_i == eta_i)) { status = QMPS_ST_NOT_CONVERGED; fallback = false; }
}

// One trajectory per workgroup; FOUR LANES per candidate of an evaluation pass (16 candidates per wave; 2P + 1 + 7 candidates:
// three waves for ShallowFull's 15 angles, two for eight angles)
template <int KIND>
__global__ __launch_bounds__(256) void evolve_bfgs_d2_kernel(EvolveD2Args p) {
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, P = p.P, nthreads = (int)blockDim.x;
  __shared__ double sX[PMAX], sG[PMAX], sD[PMAX], sS[PMAX], sGn[PMAX], sHy[PMAX], sH[PMAX][PMAX + 1], sF[64];
  __shared__ int sOK[64];
  __shared__ double2 sA[8], sC[16], sCS[PMAX], sLCS[kEvolveMaxAlphas * PMAX];
  __shared__ double sZ[PMAX], sCnt[4];
  __shared__ double2 sT[64 * 32];      // 512 B per quad: the transposed elements of the characteristic-polynomial solve
  __shared__ double2 sRoots[2][4];     // the four eigenvalues of the last pass's point (double-buffered: quads read one while the point's quad writes the other)
  __shared__ int sRootsOK[2];
  int rbuf = 0;
  if (tid < 2) sRootsOK[tid] = 0;
  const double2* W = (const double2*)p.WW;
  if (tid < 4) sCnt[tid] = 0.0;
  // ---- one evaluation pass.  Candidate c = tid / 4: c < G1 = 2P + 1 the central-difference columns of z = x + coef d; [G1, G1 + n_ladder): x + alpha_{r+1} d
#ifdef QMPS_D2_PHASES
  long long ph[6] = {0, 0, 0, 0, 0, 0};      // thread 0: sincos + barrier | circuit | matrix build + solve | closing barrier | passes | total
  const long long k0 = wall_clock64();
#define QMPS_TICK(var) const long long var = wall_clock64()
#else
#define QMPS_TICK(var)
#endif
  auto evaluate = [&](double coef, int n_ladder) {
    const int G1 = 2 * P + 1, cand = tid >> 2, q = tid & 3;
    QMPS_TICK(t0);
    const bool grad = cand < G1, ladder = !grad && cand < G1 + n_ladder;
    // cos / sin of every (scaled) angle ONCE per pass, shared through LDS (a double-precision sincos costs more than a two-qubit layer):
    // the P angles of the base point by threads 0 .. P - 1, the n_ladder P angles of the backtracking points spread over the workgroup
    if (tid < P) {
      const double z = coef != 0.0 ? add_rn(sX[tid], mul_rn(coef, sD[tid])) : sX[tid];
      sZ[tid] = z;
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(tid) * z, &sn, &cs_);
      sCS[tid] = make_double2(cs_, sn);
    }
    for (int idx = tid; idx < n_ladder * P; idx += nthreads) {
      const int r = idx / P, l = idx - r * P;
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(l) * add_rn(sX[l], mul_rn(p.alphas[r + 1], sD[l])), &sn, &cs_);
      sLCS[r * PMAX + l] = make_double2(cs_, sn);
    }
    __syncthreads();
    QMPS_TICK(t1);
#ifdef QMPS_D2_PHASES
    long long t2 = t1, t3 = t1;
#endif
    if (grad || ladder) {
      const int isel = (grad && cand > 0) ? (cand - 1) % P : -1;
      double amp_r[4] = {0, 0, 0, 0}, amp_i[4] = {0, 0, 0, 0};
      if (q < 2) {
        // lanes 0, 1 of the quad simulate columns 0, 1 of the candidate's unitary
        double2 own = make_double2(1.0, 0.0);
        if (isel >= 0) {
          double sn, cs_;
          sincos(ansatz_param_scale<KIND>(isel) * add_rn(sZ[isel], cand <= P ? p.h : -p.h), &sn, &cs_);
          own = make_double2(cs_, sn);
        }
        const double2* lcs = sLCS + (ladder ? cand - G1 : 0) * PMAX;
        Reg<2> r;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          r.re[x] = (x == q) ? 1.0 : 0.0;
          r.im[x] = 0.0;
        }
        ansatz_circuit_cs<2, KIND>(r, [&](int l) {
          double2 v = ladder ? lcs[l] : sCS[l];        // (value selects: a select between `own` and an LDS element would put `own` in scratch)
          if (l == isel) { v.x = own.x; v.y = own.y; }
          return v;
        }, P);
#pragma unroll
        for (int x = 0; x < 4; ++x) { amp_r[x] = r.re[x]; amp_i[x] = r.im[x]; }
      }
      double er, ei;
      int rounds, status;
#ifdef QMPS_D2_PHASES
      t2 = wall_clock64();
#endif
      {
        double rowr[4], rowi[4];
        overlap_quad_build(sC, amp_r, amp_i, q, rowr, rowi);
        // the characteristic-polynomial solve; the squaring solve for the maps it hands back (multiple or collapsed largest roots: points of the
        // special grid) - and for every map under QMPS_EVOLVE_D2_SQUARING.  ONE call site of the squaring: the body is inlined once.
        bool squaring = (p.probe & 4) != 0;
        int rounds_cp = 0;
        if (!squaring) {
          double2 root;
          overlap_quad_charpoly(rowr, rowi, q, sT + (tid >> 2) * 32, sRootsOK[rbuf] ? sRoots[rbuf] : nullptr, er, ei, rounds_cp, status, squaring, root);
          rounds = rounds_cp;
          if (cand == 0) {      // the point of the pass: its eigenvalues start the next pass's solves - unless the quartic handed it back
            sRoots[rbuf ^ 1][q] = root;
            if (q == 0) sRootsOK[rbuf ^ 1] = (!squaring && status == QMPS_ST_OK && root.x == root.x) ? 1 : 0;
          }
        }
        else if (cand == 0 && q == 0) sRootsOK[rbuf ^ 1] = 0;
        if (squaring) {                                   // (uniform over a quad)
          overlap_quad_solve(rowr, rowi, q, p.max_rounds, p.tol, er, ei, rounds, status);
          rounds += rounds_cp;
        }
      }
#ifdef QMPS_D2_PHASES
      t3 = wall_clock64();
#endif
      if (q == 0) {
        sF[cand] = -__builtin_sqrt(__builtin_sqrt(er * er + ei * ei));
        sOK[cand] = status == QMPS_ST_OK ? 1 : 0;
        atomicAdd(&sCnt[1], (double)rounds);
        if (status != QMPS_ST_OK) atomicAdd(&sCnt[2], 1.0);
      }
    }
    __syncthreads();
    if (coef == coef) rbuf ^= 1;      // (a pass with a point: its eigenvalues are the next pass's starting points)
#ifdef QMPS_D2_PHASES
    { const long long t4 = wall_clock64(); ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += 1; }
#endif
    if (tid == 0) sCnt[0] += (double)(G1 + n_ladder);
  };
  BfgsLds L;
  L.X = sX; L.G = sG; L.D = sD; L.S = sS; L.Gn = sGn; L.Hy = sHy; L.H = sH; L.F = sF; L.OK = sOK;
  auto build_reference = [&]() {
    // the step's reference tensor A = tensor(x) (threads 0, 1: its two columns), then C_s = sum_t WW[s][t] A_t1 A_t2 (thread = (s, i, j))
    if (tid < P) {
      double sn, cs_;
      sincos(ansatz_param_scale<KIND>(tid) * sX[tid], &sn, &cs_);
      sCS[tid] = make_double2(cs_, sn);
    }
    __syncthreads();
    if (tid < 2) {
      Reg<2> r;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        r.re[x] = (x == tid) ? 1.0 : 0.0;
        r.im[x] = 0.0;
      }
      ansatz_circuit_cs<2, KIND>(r, [&](int l) { return sCS[l]; }, P);
#pragma unroll
      for (int x = 0; x < 4; ++x) sA[((x & 1) * 2 + (x >> 1)) * 2 + tid] = make_double2(r.re[x], r.im[x]);
    }
    __syncthreads();
    if (tid < 16) {
      const int s = tid >> 2, i = (tid >> 1) & 1, j = tid & 1;
      double cr = 0.0, ci = 0.0;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int t1 = tt >> 1, t2 = tt & 1;
        double ar = 0.0, ai = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double2 x = sA[(t1 * 2 + i) * 2 + k], y = sA[(t2 * 2 + k) * 2 + j];
          ar += x.x * y.x - x.y * y.y;
          ai += x.x * y.y + x.y * y.x;
        }
        const double2 w = W[s * 4 + tt];
        cr += w.x * ar - w.y * ai;
        ci += w.x * ai + w.y * ar;
      }
      sC[tid] = make_double2(cr, ci);
    }
    __syncthreads();
  };
  __syncthreads();
  bfgs_time_evolution(p, t, tid < P ? tid : -1, tid == 0, L, evaluate, build_reference, [] { __syncthreads(); }, true);
  __syncthreads();
#ifdef QMPS_D2_PHASES
  if (tid == 0 && p.prof != nullptr) {
    ph[5] = wall_clock64() - k0;
    p.prof[t * 8 + 0] = (double)ph[5]; p.prof[t * 8 + 1] = (double)ph[0]; p.prof[t * 8 + 2] = (double)ph[2]; p.prof[t * 8 + 3] = (double)ph[1];
    p.prof[t * 8 + 4] = (double)ph[3]; p.prof[t * 8 + 5] = (double)(ph[5] - ph[0] - ph[1] - ph[2] - ph[3]); p.prof[t * 8 + 6] = (double)ph[4]; p.prof[t * 8 + 7] = sCnt[1];
  }
#endif
  if (tid == 0) {
    if (p.nfev != nullptr) p.nfev[t] = sCnt[0];
    if (p.rounds != nullptr) p.rounds[t] = sCnt[1];
    if (p.fail != nullptr) p.fail[t] = (int32_t)sCnt[2];
  }
}

hipError_t launch_evolve_bfgs_d2(int kind, const EvolveD2Args& a, hipStream_t st) {
  if (a.T <= 0) return hipSuccess;
  const int cands = 2 * a.P + 1 + (a.NA - 1);
  if (a.P < 1 || a.P > PMAX || a.NA < 1 || a.NA > kEvolveMaxAlphas || cands > 64) return hipErrorInvalidValue;
  const dim3 grid((unsigned)a.T), block(64 * ((cands + 15) / 16));        // four lanes per candidate
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
