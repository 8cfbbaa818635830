// qmps_direct_core.h - the per-evaluation mathematics of the DIRECT environment solve at D = 4, written once for
// two back-ends:
//   device : V = double - one lane of a DPP quad, four lanes per evaluation (qmps_direct.hip);
//   host   : V = four doubles in lock-step - TEST INFRASTRUCTURE ONLY (tests/csrc/direct_emu.cpp): it lets the CPU
//            test-suite check this very source against the oracle.  The product never runs it.
//
// What is computed (reference: the exact eigen-solve behind get_env_exact, qmps/tools.py:176-182, then
// State + psi^+ (1 x h x 1) psi, qmps/represent.py:258-262, qmps/ground_state.py:159-167):
//   the right environment r is the fixed point of T(r) = sum_s A_s r A_s^+.  For a left isometry T preserves the
//   trace, so in real coordinates u of the Hermitian matrix r the fixed point solves the REAL 16 x 16 system
//       (R - 1 + e_15 t^T) u = e_15        (t = trace functional; R = matrix of T)
//   by Gauss-Jordan elimination without pivoting (measured on 65536 Haar tensors: residual ||T(r) - r||_F
//   < 1.2e-15), ACCEPTED only if one power step moves it by less than tol in Frobenius norm - the same criterion as
//   the iterative solvers, so status and tolerance semantics are unchanged; otherwise (not an isometry, degenerate
//   transfer spectrum, an unlucky pivot) the power method 2^m steps at a time takes over from r_0 = 1/D.
//
// Coordinates (plain, not orthonormal), index a = 4 i + i':
//   u[(i,i)] = r_ii ;  u[(i,i')] = Re r_ii' (i < i') ;  u[(i,i')] = Im r_i'i (i > i')
//   ||r - r'||_F^2 = sum_diag du^2 + 2 sum_offdiag du^2.
// Lane q of the quad owns the four coordinates a = (q, i'), i' = 0..3, i.e. rows 4 q .. 4 q + 3 of every matrix.
//
// Ops policy O (an object; holds the lane id and the tensor):
//   types   O::V (value), O::P (lane predicate)
//   lanes   o.q_eq(i), o.q_gt(i)                       predicates on the lane's row index q
//   quad    O::template bcast<L>(v), O::qsum(v)        value of lane L / sum over the quad, in every lane
//           o.template row_bcast<L, K>(row, out)       out[K..15] = row[K..15] of lane L, in every lane
//   select  O::sel(p, a, b);  O::rcp(v);  O::fma(a, b, c);  O::lt(a, b) -> P;  O::gt0(a) -> P
//           O::p_and(p, q), O::p_not(p), O::any(p) -> bool
//   tensor  o.own(s, j, re, im)   A_s[q][j]  (the lane's own row)
//           o.uni(s, i, j, re, im) A_s[i][j]  (the same value in every lane of the quad)
#pragma once

#include <utility>

namespace qmps {

#if defined(__HIPCC__)
#define QMPS_CORE_FN __device__ inline __attribute__((always_inline))
// instruction-scheduling fence: the compiler may not move anything across it.  The fully unrolled phases below are
// long straight-line blocks; without fences the scheduler hoists the operand reads / DPP moves of later steps and
// the register demand explodes (470 VGPRs without them).
#define QMPS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define QMPS_CORE_FN inline __attribute__((always_inline))
#define QMPS_SCHED_FENCE() ((void)0)
#endif

template <int... I, class F>
QMPS_CORE_FN void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
QMPS_CORE_FN void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <class O>
struct DirectD4 {
  using V = typename O::V;
  using P = typename O::P;

  // the lane's own diagonal coordinate (q, q) out of its four values
  static QMPS_CORE_FN V own_diag(const O& o, const V (&x)[4]) {
    return O::sel(o.q_eq(0), x[0], O::sel(o.q_eq(1), x[1], O::sel(o.q_eq(2), x[2], x[3])));
  }

  // ---- 1. the lane's four rows of the real transfer matrix: Rc[i'][4 j + j'] = coordinate (q, i') of T(H_(j,j')) ----
  // With P(j,j') = sum_s (gamma A_s[q][j]) conj(A_s[i'][j']), gamma = 1 (q <= i': a real part) or i (q > i': minus an
  // imaginary part):  column (j,j): Re P(j,j);  column (lo,hi): Re (P(lo,hi) + P(hi,lo));
  // column (hi,lo): -Im (P(lo,hi) - P(hi,lo)).
  static QMPS_CORE_FN void build(const O& o, V (&Rc)[4][16]) {
    V ar[2][4], ai[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) o.own(s, j, ar[s][j], ai[s][j]);
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      const P rot = o.q_gt(ip);
      V tr[2][4], ti[2][4], br[2][4], bi[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (no lane has q > 3: the last partner index needs no select)
          tr[s][j] = ip == 3 ? ar[s][j] : O::sel(rot, -ai[s][j], ar[s][j]);
          ti[s][j] = ip == 3 ? ai[s][j] : O::sel(rot, ar[s][j], ai[s][j]);
          o.uni(s, ip, j, br[s][j], bi[s][j]);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        V v = tr[0][j] * br[0][j];
        v = O::fma(ti[0][j], bi[0][j], v);
        v = O::fma(tr[1][j], br[1][j], v);
        v = O::fma(ti[1][j], bi[1][j], v);
        Rc[ip][5 * j] = v;
      }
#pragma unroll
      for (int lo = 0; lo < 4; ++lo)
#pragma unroll
        for (int hi = lo + 1; hi < 4; ++hi) {
          V re = tr[0][lo] * br[0][hi], im = tr[0][lo] * bi[0][hi];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (s > 0) {
              re = O::fma(tr[s][lo], br[s][hi], re);
              im = O::fma(tr[s][lo], bi[s][hi], im);
            }
            re = O::fma(ti[s][lo], bi[s][hi], re);
            re = O::fma(tr[s][hi], br[s][lo], re);
            re = O::fma(ti[s][hi], bi[s][lo], re);
            im = O::fma(-ti[s][lo], br[s][hi], im);
            im = O::fma(ti[s][hi], br[s][lo], im);
            im = O::fma(-tr[s][hi], bi[s][lo], im);
          }
          Rc[ip][4 * lo + hi] = re;
          Rc[ip][4 * hi + lo] = im;
        }
    }
  }

  // ---- 2. (R - 1 + e_15 t^T) u = e_15 by Gauss-Jordan elimination, rows distributed over the quad ----
  // Step k: the lane that owns row k (lane k / 4, register k % 4) broadcasts what is left of it; every lane
  // eliminates column k from its four rows (the pivot row itself is skipped by a zero multiplier).
  // M: the lane's rows of R on entry; destroyed.
  // pivmax: the largest |1 / pivot| of the elimination (the same in every lane of the quad) - see kMaxInversePivot
  static QMPS_CORE_FN void solve(const O& o, V (&M)[4][16], V (&x)[4], V& pivmax) {
    V y[4], dinv[4];
    const V one = O::splat(1.0), zero = O::splat(0.0);
    V w[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) w[l] = O::sel(o.q_eq(l), one, zero);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int l = 0; l < 4; ++l) M[r][4 * l + r] = M[r][4 * l + r] - w[l];   // - identity: column 4 q + r
      dinv[r] = zero;
    }
    // + the trace functional on the LAST pivot row (lane 3, register 3), right-hand side e_15: the singular direction
    // of R - 1 is then resolved by the final pivot (measured on the 65536 Haar tensors of the benchmark: largest
    // residual ||T(r) - r||_F 1.1e-15, against 3.9e-13 with the functional on row 0).  The right-hand side stays e_15
    // until that last pivot, so it is not carried through the elimination: at k = 15 row r receives -f_r.
#pragma unroll
    for (int j = 0; j < 4; ++j) M[3][5 * j] = M[3][5 * j] + w[3];
    static_for<16>([&](auto K) {
      constexpr int k = decltype(K)::value, pl = k >> 2, pr = k & 3;
      V prow[16];
      o.template row_bcast<pl, k>(M[pr], prow);      // entries k .. 15 of the pivot row, in every lane of the quad
      const V pinv = O::rcp(prow[k]);
      const P mine = o.q_eq(pl);
      dinv[pr] = O::sel(mine, pinv, dinv[pr]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        V f = M[r][k] * pinv;
        if (r == pr) f = O::sel(mine, zero, f);
#pragma unroll
        for (int j = k + 1; j < 16; ++j) M[r][j] = O::fma(-f, prow[j], M[r][j]);
        if (k == 15) y[r] = -f;
      }
    });
    y[3] = O::sel(o.q_eq(3), one, y[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = y[r] * dinv[r];
    pivmax = O::qmax(O::vmax(O::vmax(O::vabs(dinv[0]), O::vabs(dinv[1])), O::vmax(O::vabs(dinv[2]), O::vabs(dinv[3]))));
  }

  // A transfer map whose fixed point is NOT unique (degenerate dominant eigenvalue: product states, special angles of the
  // ansatz) makes the system singular to rounding; the elimination then returns SOME fixed point - it passes the acceptance
  // step like any other - which one being decided by rounding errors.  Such evaluations (a pivot below 1e-10) are handed to
  // the power method instead, which returns the projection of r_0 = 1/D onto the fixed-point space or reports that it does
  // not converge, exactly as the iterative solvers do.
  static constexpr double kMaxInversePivot = 1e10;

  // all sixteen coordinates in every lane: xs[4 l + r] = x[r] of lane l
  static QMPS_CORE_FN void gather(const V (&x)[4], V (&xs)[16]) {
    static_for<4>([&](auto L) {
      constexpr int l = decltype(L)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) xs[4 * l + r] = O::template bcast<l>(x[r]);
    });
  }

  // scale to trace 1 (the trace lives on the lanes' diagonal coordinates)
  static QMPS_CORE_FN V normalise(const O& o, V (&x)[4]) {
    const V tr = O::qsum(own_diag(o, x));
    const V inv = O::rcp(tr);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = x[r] * inv;
    return inv;
  }

  // ||x - y||_F^2 of the two Hermitian matrices (the same in every lane of the quad)
  static QMPS_CORE_FN V dist2(const O& o, const V (&x)[4], const V (&y)[4]) {
    V d = O::splat(0.0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const V dd = y[r] - x[r];
      const V wt = O::sel(o.q_eq(r), O::splat(1.0), O::splat(2.0));
      d = O::fma(wt * dd, dd, d);
    }
    return O::qsum(d);
  }

  // ---- 3. one power step y = T(x)/tr straight from the tensor, and the squared Frobenius distance to x ----
  // us: the gathered coordinates of x.  Lane q forms row q of sum_s A_s r A_s^+ (X_s = A_s[q][:] r, then
  // X_s A_s^+) and keeps the real or imaginary part its coordinates call for.
  static QMPS_CORE_FN V power_step(const O& o, const V (&x)[4], const V (&us)[16], V (&y)[4]) {
    V ar[2][4], ai[2][4], xr[2][4], xi[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) o.own(s, j, ar[s][j], ai[s][j]);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        V cr = O::splat(0.0), ci = O::splat(0.0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const V rr = us[k <= l ? 4 * k + l : 4 * l + k];          // Re r[k][l]
          cr = O::fma(ar[s][k], rr, cr);
          ci = O::fma(ai[s][k], rr, ci);
          if (k != l) {
            const V ri = k < l ? us[4 * l + k] : -us[4 * k + l];    // Im r[k][l]
            cr = O::fma(-ai[s][k], ri, cr);
            ci = O::fma(ar[s][k], ri, ci);
          }
        }
        xr[s][l] = cr;
        xi[s][l] = ci;
      }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      // r'[q][l] = sum_s sum_k X_s[k] conj(A_s[l][k])
      V cr = O::splat(0.0), ci = O::splat(0.0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          V br, bi;
          o.uni(s, l, k, br, bi);
          cr = O::fma(xr[s][k], br, cr);
          cr = O::fma(xi[s][k], bi, cr);
          ci = O::fma(xi[s][k], br, ci);
          ci = O::fma(-xr[s][k], bi, ci);
        }
      y[l] = l == 3 ? cr : O::sel(o.q_gt(l), -ci, cr);     // u[(q,l)] = Re r'[q][l] (q <= l),  Im r'[l][q] = -Im r'[q][l] (q > l)
    }
    normalise(o, y);
    return dist2(o, x, y);
  }

  // ---- 4. fall-back: the power method 2^m steps at a time from r_0 = 1/4, Rc <- Rc Rc in the quad layout ----
  // Reference formulation, run by the CPU emulation.  The device kernel runs the SAME recurrence one evaluation at a
  // time on the matrix cores (qmps_direct.hip: squaring_fallback; this form holds two 16 x 16 matrices per quad in
  // registers, which would cost the common path its register budget); tests/test_direct_gpu.py compares the two.
  static QMPS_CORE_FN void square(V (&Rc)[4][16]) {
    V out[4][16];
    static_for<16>([&](auto C) {
      constexpr int c = decltype(C)::value, cl = c >> 2, cr = c & 3;
      V rowc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) rowc[j] = O::template bcast<cl>(Rc[cr][j]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) out[r][j] = c == 0 ? Rc[r][c] * rowc[j] : O::fma(Rc[r][c], rowc[j], out[r][j]);
      QMPS_SCHED_FENCE();
    });
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) Rc[r][j] = out[r][j];
  }

  // z_m = T^(2^m) r_0 / tr, stop at ||z_m - z_(m-1)||_F < tol;  steps = 2^m (as a value: it is per evaluation).
  // `active` lanes take part; converged ones are frozen.  Returns the still-unconverged predicate.
  static QMPS_CORE_FN P squaring(const O& o, V (&Rc)[4][16], P active, int max_iter, double tol2, V (&x)[4], V& steps) {
    V prev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prev[r] = O::sel(o.q_eq(r), O::splat(0.25), O::splat(0.0));
    int m = 0;
    while (O::any(active) && m < 29 && (2 << m) <= max_iter) {
      square(Rc);
      ++m;
      V z[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = O::splat(0.25) * ((Rc[r][0] + Rc[r][5]) + (Rc[r][10] + Rc[r][15]));
      const V inv = normalise(o, z);
      // 1/tr(T^(2^m) r_0) ~ 1/lambda^(2^m): keeps the squared matrix at O(1) for tensors that are not isometries
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) Rc[r][j] = Rc[r][j] * inv;
      const P conv = O::lt(dist2(o, prev, z), O::splat(tol2));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[r] = O::sel(active, z[r], x[r]);
        prev[r] = z[r];
      }
      steps = O::sel(active, O::splat((double)(1 << m)), steps);
      active = O::p_and(active, O::p_not(conv));
    }
    return active;
  }

  // row q of B_tau = A_t1 A_t2, tau = 2 t1 + t2
  static QMPS_CORE_FN void b_rows(const O& o, V (&bre)[4][4], V (&bim)[4][4]) {
    V ar[2][4], ai[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) o.own(s, j, ar[s][j], ai[s][j]);
    // one row j of A_t2 at a time (8 values live instead of the whole matrix): B[(t1 t2)][k] += A_t1[q][j] A_t2[j][k]
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        V ur[4], ui[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o.uni(t2, j, k, ur[k], ui[k]);
#pragma unroll
        for (int t1 = 0; t1 < 2; ++t1)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            V cr, ci;
            if (j == 0) {
              cr = ar[t1][0] * ur[k];
              ci = ar[t1][0] * ui[k];
            } else {
              cr = O::fma(ar[t1][j], ur[k], bre[2 * t1 + t2][k]);
              ci = O::fma(ar[t1][j], ui[k], bim[2 * t1 + t2][k]);
            }
            bre[2 * t1 + t2][k] = O::fma(-ai[t1][j], ui[k], cr);
            bim[2 * t1 + t2][k] = O::fma(ai[t1][j], ur[k], ci);
          }
      }
  }

  // ---- 5b. one energy without the density matrix (the energy-only contraction chain at high occupancy): with
  // C_sigma = sum_tau h[sigma,tau] B_tau (row q) and Z_sigma = C_sigma r,
  //   E = Re sum_sigma sum_l Z_sigma[l] conj(B_sigma[l])     ( = Re sum h[sigma,tau] tr(B_tau r B_sigma^+) )
  // - only the four B rows (b_rows, computed once for all terms) and r stay live (~150 registers against ~230 for the
  // rho route).  Returns the lane's share.
  static QMPS_CORE_FN V energy_lean(const V (&bre)[4][4], const V (&bim)[4][4], const V (&us)[16], const double* h) {
    V e = O::splat(0.0);
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
      V cr[4], ci[4];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        cr[l] = O::splat(0.0);
        ci[l] = O::splat(0.0);
      }
#pragma unroll
      for (int tau = 0; tau < 4; ++tau) {
        const V hr = O::splat(h[2 * (4 * sg + tau)]), hi = O::splat(h[2 * (4 * sg + tau) + 1]);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          cr[l] = O::fma(hr, bre[tau][l], cr[l]);
          cr[l] = O::fma(-hi, bim[tau][l], cr[l]);
          ci[l] = O::fma(hr, bim[tau][l], ci[l]);
          ci[l] = O::fma(hi, bre[tau][l], ci[l]);
        }
      }
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        // Z[l] = sum_k C[k] r[k][l];  r[k][l] for k <= l from (us[4k+l], us[4l+k]), the conjugate otherwise
        V zr = O::splat(0.0), zi = O::splat(0.0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const V rr = us[k <= l ? 4 * k + l : 4 * l + k];
          zr = O::fma(cr[k], rr, zr);
          zi = O::fma(ci[k], rr, zi);
          if (k != l) {
            const V ri = k < l ? us[4 * l + k] : -us[4 * k + l];
            zr = O::fma(-ci[k], ri, zr);
            zi = O::fma(cr[k], ri, zi);
          }
        }
        e = O::fma(zr, bre[sg][l], e);
        e = O::fma(zi, bim[sg][l], e);
      }
    }
    return e;
  }

  // ---- 5. positive-definiteness of r (LDL^H pivots > 0: the criterion of cholesky(r), qmps/tools.py:182) and the
  //         lane's share of the two-site density matrix rho[tau][sigma] = tr(B_tau r B_sigma^+), tau <= sigma ----
  // us: all sixteen coordinates of r (trace 1).  Lane q contributes row q of B_tau = A_t1 A_t2 (tau = 2 t1 + t2).
  static QMPS_CORE_FN P density(const O& o, const V (&us)[16], V (&pre)[4][4], V (&pim)[4][4]) {
    // r[k][l], k <= l
    V rre[4][4], rim[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int l = k; l < 4; ++l) {
        rre[k][l] = us[4 * k + l];
        rim[k][l] = k == l ? O::splat(0.0) : us[4 * l + k];
      }
    // LDL^H (replicated in the four lanes)
    P pd = O::gt0(rre[0][0]);
    {
      V lre[4][4], lim[4][4], d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        V dj = rre[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) dj = O::fma(-(lre[j][k] * lre[j][k] + lim[j][k] * lim[j][k]), d[k], dj);
        d[j] = dj;
        pd = O::p_and(pd, O::gt0(dj));
        const V inv = O::rcp(O::sel(O::gt0(dj), dj, O::splat(1.0)));
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
          V cr = rre[j][i], ci = -rim[j][i];   // r[i][j] = conj(r[j][i])
#pragma unroll
          for (int k = 0; k < j; ++k) {
            // L[i][k] conj(L[j][k]) d_k
            const V pr = lre[i][k] * lre[j][k] + lim[i][k] * lim[j][k];
            const V pi = lim[i][k] * lre[j][k] - lre[i][k] * lim[j][k];
            cr = O::fma(-pr, d[k], cr);
            ci = O::fma(-pi, d[k], ci);
          }
          lre[i][j] = cr * inv;
          lim[i][j] = ci * inv;
        }
      }
    }
    V bre[4][4], bim[4][4];
    b_rows(o, bre, bim);
    // Y_tau = (row q of B_tau) r
    V yre[4][4], yim[4][4];
#pragma unroll
    for (int tau = 0; tau < 4; ++tau)
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        V cr = O::splat(0.0), ci = O::splat(0.0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // r[k][l]: k <= l stored; k > l the conjugate of r[l][k]
          const V rr = k <= l ? rre[k][l] : rre[l][k];
          cr = O::fma(bre[tau][k], rr, cr);
          ci = O::fma(bim[tau][k], rr, ci);
          if (k != l) {
            const V ri = k < l ? rim[k][l] : -rim[l][k];
            cr = O::fma(-bim[tau][k], ri, cr);
            ci = O::fma(bre[tau][k], ri, ci);
          }
        }
        yre[tau][l] = cr;
        yim[tau][l] = ci;
      }
    // rho[tau][sigma] += sum_l Y_tau[l] conj(B_sigma[l])
#pragma unroll
    for (int tau = 0; tau < 4; ++tau)
#pragma unroll
      for (int sg = tau; sg < 4; ++sg) {
        V cr = yre[tau][0] * bre[sg][0], ci = O::splat(0.0);
        cr = O::fma(yim[tau][0], bim[sg][0], cr);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          if (l > 0) {
            cr = O::fma(yre[tau][l], bre[sg][l], cr);
            cr = O::fma(yim[tau][l], bim[sg][l], cr);
          }
          if (sg != tau) {
            ci = O::fma(yim[tau][l], bre[sg][l], ci);
            ci = O::fma(-yre[tau][l], bim[sg][l], ci);
          }
        }
        pre[tau][sg] = cr;
        pim[tau][sg] = ci;
      }
    return pd;
  }

  // E = Re sum_{s,t} h[s][t] rho[t][s] from the upper triangle of rho; h: 16 complex numbers (re, im interleaved),
  // the same for every lane
  static QMPS_CORE_FN V energy(const double* h, const V (&pre)[4][4], const V (&pim)[4][4]) {
    V e = O::splat(0.0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double hr = h[2 * (4 * s + t)], hi = h[2 * (4 * s + t) + 1];
        const V rr = t <= s ? pre[t][s] : pre[s][t];
        e = O::fma(O::splat(hr), rr, e);
        if (t != s) {
          const V ri = t < s ? pim[t][s] : -pim[s][t];
          e = O::fma(O::splat(-hi), ri, e);
        }
      }
    return e;
  }
};

}  // namespace qmps
