// qmps_overlap_d2.h - the D = 2 time-evolution overlap solve of ONE lane (gfx950 only), shared by overlap_lane_kernel
// (qmps_overlap.hip) and the device-resident BFGS time evolution (qmps_evolve_d2.hip).
// Reference: qmps/new_time_evolve.py:193-221, scripts/loschmidt.py:209-239, qmps/time_evolve_tools.py:20-23.
//   T(x) = sum_{s=0..3} C_s x Bm_s^+ ,  C = WW . merge(A, A) ,  Bm = merge(B, B)
// eta = dominant eigenvalue of T by power iteration applied 2^m steps at a time: the 4 x 4 complex matrix of T is squared
// (Frobenius-normalised each round), its dominant right vector v read off the largest column, eta = <v, T v>/<v, v>, stop when
// ||T v - eta v|| < tol ||v||.  Ap(k) / Bp(k): element k of the tensors [2][2][2] (reference / candidate), any storage.
#pragma once
#include <hip/hip_runtime.h>

#include "qmps_kernels.h"

namespace qmps {

struct OverlapLaneResult {
  double eta_r, eta_i;
  int rounds, status;
  double vr[4], vi[4];      // the dominant right vector (not normalised)
};

template <class GA, class GB>
__device__ __forceinline__ void overlap_lane_solve(GA Ap, GB Bp, const double2* W, int max_rounds, double tol, OverlapLaneResult& out) {
  // two-site products: AA[t1 t2] = A_t1 A_t2, BB likewise (2 x 2 complex each)
  double aar[4][2][2], aai[4][2][2], bbr[4][2][2], bbi[4][2][2];
#pragma unroll
  for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          double ar = 0, ai = 0, br = 0, bi = 0;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const double2 x = Ap((t1 * 2 + i) * 2 + k), y = Ap((t2 * 2 + k) * 2 + j);
            ar += x.x * y.x - x.y * y.y;
            ai += x.x * y.y + x.y * y.x;
            const double2 u = Bp((t1 * 2 + i) * 2 + k), v = Bp((t2 * 2 + k) * 2 + j);
            br += u.x * v.x - u.y * v.y;
            bi += u.x * v.y + u.y * v.x;
          }
          aar[2 * t1 + t2][i][j] = ar; aai[2 * t1 + t2][i][j] = ai;
          bbr[2 * t1 + t2][i][j] = br; bbi[2 * t1 + t2][i][j] = bi;
        }
  // C_s = sum_t WW[s][t] AA_t
  double cr[4][2][2], ci[4][2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        double xr = 0, xi = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double2 w = W[s * 4 + t];
          xr += w.x * aar[t][i][j] - w.y * aai[t][i][j];
          xi += w.x * aai[t][i][j] + w.y * aar[t][i][j];
        }
        cr[s][i][j] = xr; ci[s][i][j] = xi;
      }
  // E[(i,i'),(j,j')] = sum_s C_s[i][j] conj(Bm_s[i'][j'])
  double er[4][4], ei[4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ip = 0; ip < 2; ++ip)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          double xr = 0, xi = 0;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            xr += cr[s][i][j] * bbr[s][ip][jp] + ci[s][i][j] * bbi[s][ip][jp];
            xi += ci[s][i][j] * bbr[s][ip][jp] - cr[s][i][j] * bbi[s][ip][jp];
          }
          er[2 * i + ip][2 * j + jp] = xr; ei[2 * i + ip][2 * j + jp] = xi;
        }
  double mr[4][4], mi[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) { mr[a][c] = er[a][c]; mi[a][c] = ei[a][c]; }
  double eta_r = 0.0, eta_i = 0.0, vr[4] = {1, 0, 0, 0}, vi[4] = {0, 0, 0, 0};
  int status = QMPS_ST_NOT_CONVERGED, rounds = 0;
  const double tol2 = tol * tol;
  bool collapsed = false;
  double m2 = 0.0;      // ||M||_F^2 of the current power: computed for the first one, 1 afterwards
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) m2 += mr[a][c] * mr[a][c] + mi[a][c] * mi[a][c];
  for (int m = 0; m <= max_rounds; ++m) {
    // dominant right vector = largest column of the current power
    double best = -1.0;
    int bc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double n2 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) n2 += mr[a][c] * mr[a][c] + mi[a][c] * mi[a][c];
      if (n2 > best) { best = n2; bc = c; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      vr[a] = bc == 0 ? mr[a][0] : (bc == 1 ? mr[a][1] : (bc == 2 ? mr[a][2] : mr[a][3]));
      vi[a] = bc == 0 ? mi[a][0] : (bc == 1 ? mi[a][1] : (bc == 2 ? mi[a][2] : mi[a][3]));
    }
    // eta = <v, E v>/<v, v>, residual ||E v - eta v||^2 / ||v||^2
    double wr[4], wi[4], num_r = 0, num_i = 0, vv = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double xr = 0, xi = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xr += er[a][c] * vr[c] - ei[a][c] * vi[c];
        xi += er[a][c] * vi[c] + ei[a][c] * vr[c];
      }
      wr[a] = xr; wi[a] = xi;
      num_r += vr[a] * xr + vi[a] * xi;
      num_i += vr[a] * xi - vi[a] * xr;
      vv += vr[a] * vr[a] + vi[a] * vi[a];
    }
    eta_r = num_r / vv; eta_i = num_i / vv;
    double res = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double dr = wr[a] - (eta_r * vr[a] - eta_i * vi[a]), di = wi[a] - (eta_r * vi[a] + eta_i * vr[a]);
      res += dr * dr + di * di;
    }
    rounds = m;
    // the square: the next power - and the two tests the eigen-residual of a column cannot make.  Has the power COLLAPSED to rounding noise
    // (||M M|| < 1e-14 ||M||^2: within the first rounds a nilpotent map - reference and candidate orthogonal, every eigenvalue zero; later a
    // defective dominant eigenvalue losing its digits: no answer)?  And - looked at only once the column passes - is the power RANK ONE,
    // ||M M - tr(M) M|| << ||M M||?  At symmetric points of the ansatz a column of an early power can be an EXACT eigenvector of a
    // sub-dominant eigenvalue (|eta_2/eta_1| = 0.9994: accepted after 5 squarings with the wrong eigenvalue; stress_overlap.py, round 5).
    double qr[4][4], qi[4][4], f2 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double xr = 0, xi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xr += mr[a][k] * mr[k][c] - mi[a][k] * mi[k][c];
          xi += mr[a][k] * mi[k][c] + mi[a][k] * mr[k][c];
        }
        qr[a][c] = xr; qi[a][c] = xi;
        f2 += xr * xr + xi * xi;
      }
    if (f2 < 1e-28 * m2 * m2) {
      if (m <= 8) { eta_r = 0.0; eta_i = 0.0; status = QMPS_ST_OK; }
      collapsed = true;
      break;
    }
    if (res < tol2 * vv) {
      double tr_r = 0.0, tr_i = 0.0, r1 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) { tr_r += mr[a][a]; tr_i += mi[a][a]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double dr = qr[a][c] - (tr_r * mr[a][c] - tr_i * mi[a][c]), di = qi[a][c] - (tr_r * mi[a][c] + tr_i * mr[a][c]);
          r1 += dr * dr + di * di;
        }
      if (r1 < 1e-20 * f2) { status = QMPS_ST_OK; break; }
    }
    if (m == max_rounds) break;
    const double inv = 1.0 / __builtin_sqrt(f2);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { mr[a][c] = qr[a][c] * inv; mi[a][c] = qi[a][c] * inv; }
    m2 = 1.0;
  }
  if (status != QMPS_ST_OK && !collapsed && rounds >= 30) {
    // Thirty and more squarings without a rank-one power: the dominant eigenvalues are TIED in modulus (a complex-conjugate pair, a ring -
    // generic on symmetric manifolds of the ansatz: beta = -gamma of the depth-1 ShallowCNOT gate, product states).  No unique fixed point - but
    // their common modulus is what the reference's objective -sqrt|eta| measures with whichever member ARPACK returns, and ||E^(2^m)||^(1/2^m)
    // (a Gelfand bound, from the norms of the squared powers) has it to 2^-m ln(condition) ~ 1e-11: eta = |eta| (real), QMPS_ST_TIED (ABI 6.2: status 0;
    // round 5: BFGS trajectories that walk into such a manifold used to die of NaN).  The rare path: the logarithms live in this second pass
    // over the squarings.  The vector handed out is the largest column of the last power - a mixture, not an eigenvector.
    double n2 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { mr[a][c] = er[a][c]; mi[a][c] = ei[a][c]; n2 += er[a][c] * er[a][c] + ei[a][c] * ei[a][c]; }
    double log_rho = 0.5 * log(n2);
    {
      const double inv0 = 1.0 / __builtin_sqrt(n2);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { mr[a][c] *= inv0; mi[a][c] *= inv0; }
    }
    for (int m = 0; m < 44; ++m) {
      double qr[4][4], qi[4][4], f2 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double xr = 0, xi = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            xr += mr[a][k] * mr[k][c] - mi[a][k] * mi[k][c];
            xi += mr[a][k] * mi[k][c] + mi[a][k] * mr[k][c];
          }
          qr[a][c] = xr; qi[a][c] = xi;
          f2 += xr * xr + xi * xi;
        }
      if (!(f2 > 1e-280)) break;
      log_rho += ldexp(0.5 * log(f2), -(m + 1));
      const double inv = 1.0 / __builtin_sqrt(f2);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { mr[a][c] = qr[a][c] * inv; mi[a][c] = qi[a][c] * inv; }
    }
    eta_r = exp(log_rho);
    eta_i = 0.0;
    status = QMPS_ST_TIED;      // (ABI 6.4: its own status - usable as an objective, but r_out is no fixed point; it was status 0 in ABI 6.2 / 6.3)
  }
  out.eta_r = eta_r;
  out.eta_i = eta_i;
  out.rounds = rounds;
  out.status = status;
#pragma unroll
  for (int a = 0; a < 4; ++a) { out.vr[a] = vr[a]; out.vi[a] = vi[a]; }
}

}  // namespace qmps
