// qmps_roto_rule.h - the scalar minimisers behind the double-frequency rotosolve update (qmps/tools.py:447-452).  Plain C++ with
// no wave-level operation: hipcc builds it for gfx950 (the kernels include it through qmps_roto_math.h) and g++ builds the very
// same source for tests/csrc/libroto_emu.so, so the CPU test-suite checks THIS file, call by call, against the answers scipy gave
// the reference (tests/golden/refshim_golden.npz: refshim_roto_fits).  The product never runs the host build.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define QMPS_HD __host__ __device__ inline
#else
#define QMPS_HD inline
#endif

namespace qmps {

QMPS_HD double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Global minimiser on [-pi, pi) of the fit of the double-frequency rotosolve (qmps/tools.py:447-451; the reference hands
// it to scipy's minimize_scalar),  f(x) = P sin(2x + u) + Q sin(x + v) = a sin 2x + b cos 2x + c sin x + d cos x
// with (a, b) = P (cos u, sin u), (c, d) = Q (cos v, sin v) - i.e. the fitted coefficients themselves, no hypot / atan2 /
// sincos round trip.  32-point grid (sines and cosines of k pi/16: compile-time constants), then the root of f' inside the
// bracket around the best grid point: 6 bisections + up to 5 guarded Newton steps, ONE sincos per evaluation (sin 2x and
// cos 2x by the double-angle formulas), for BOTH basins when the fit has two; the lower refined value wins.  A flat fit returns 0.  (First version: 32 + ~80 transcendental calls per fit, the
// update kernel took 21 us - a third of a parameter update at D = 4.)
QMPS_HD constexpr double sin_k_pi_16(int k) {
  constexpr double S[9] = {0.0, 0.19509032201612825, 0.3826834323650898, 0.5555702330196022, 0.7071067811865476,
                           0.8314696123025452, 0.9238795325112867, 0.9807852804032304, 1.0};
  k &= 31;
  return k <= 8 ? S[k] : (k <= 16 ? S[16 - k] : (k <= 24 ? -S[k - 16] : -S[32 - k]));
}
QMPS_HD double double_sinusoid_argmin(double a, double b, double c, double d) {
  if (!(fabs(a) + fabs(b) + fabs(c) + fabs(d) > 0.0)) return 0.0;
  constexpr int NG = 32;
  constexpr double H = 6.283185307179586 / NG;
  // The fit has at most two minima per period.  Grid values rank the two basins only to O(H^2 f''): when the minima are nearly
  // degenerate the lower GRID point may sit in the basin of the higher minimum (3 of the 486 fits of the reference-run fixtures:
  // f differs by 4e-4 between the basins) - so both basins are refined and the refined values compared.
  double fg_prev2 = 0.0, fg_prev = 0.0;
  double best0 = 1e300, best1 = 1e300;      // the two lowest LOCAL minima of the grid (cyclic), and their grid indices
  int g0 = -1, g1 = -1;
  auto grid = [&](int g) {
    // x_g = -pi + g pi/16:  sin x_g = -sin(g pi/16),  cos x_g = -cos(g pi/16);  2 x_g = g pi/8 (mod 2 pi)
    const double sx = -sin_k_pi_16(g), cx = -sin_k_pi_16(g + 8), s2 = sin_k_pi_16(2 * g), c2 = sin_k_pi_16(2 * g + 8);
    return rfma(a, s2, rfma(b, c2, rfma(c, sx, d * cx)));
  };
  auto offer = [&](int g, double f) {
    if (f < best0) { best1 = best0; g1 = g0; best0 = f; g0 = g; }
    else if (f < best1) { best1 = f; g1 = g; }
  };
#pragma unroll
  for (int g = 0; g < NG + 2; ++g) {         // g = NG, NG + 1 wrap around: every grid point is tested as the middle of a triple
    const double f = grid(g);
    if (g >= 2 && fg_prev <= fg_prev2 && fg_prev < f) offer((g - 1) & (NG - 1), fg_prev);
    fg_prev2 = fg_prev;
    fg_prev = f;
  }
  if (g0 < 0) return 0.0;                    // (a constant grid: flat to rounding)
  auto eval = [&](double x, double& f, double& df, double& d2f) {
    double sx, cx;
    sincos(x, &sx, &cx);
    const double s2 = 2.0 * sx * cx, c2 = rfma(cx, cx, -sx * sx);
    f = rfma(a, s2, rfma(b, c2, rfma(c, sx, d * cx)));
    df = rfma(2.0 * a, c2, rfma(-2.0 * b, s2, rfma(c, cx, -d * sx)));
    d2f = -rfma(4.0 * a, s2, rfma(4.0 * b, c2, rfma(c, sx, d * cx)));
  };
  // root of f' inside the bracket around a grid minimum: 6 bisections + up to 5 guarded Newton steps; returns the refined value
  auto refine = [&](int gb, double fgrid, double& xout) {
    const double xb = -3.141592653589793 + gb * H;
    double lo = xb - H, hi = xb + H, x = xb, f, df, d2f, dlo, dhi;
    xout = xb;
    eval(lo, f, dlo, d2f);
    eval(hi, f, dhi, d2f);
    if (!(dlo < 0.0 && dhi > 0.0)) return fgrid;
    for (int it = 0; it < 6; ++it) {
      const double mid = 0.5 * (lo + hi);
      eval(mid, f, df, d2f);
      if (df > 0.0) hi = mid; else lo = mid;
    }
    x = 0.5 * (lo + hi);
    for (int it = 0; it < 5; ++it) {
      eval(x, f, df, d2f);
      if (!(d2f > 0.0)) break;
      const double xn = x - df / d2f;
      if (!(xn > lo - 1e-3 && xn < hi + 1e-3) || xn == x) break;
      x = xn;
    }
    eval(x, f, df, d2f);
    if (!(f <= fgrid)) return fgrid;
    xout = x;
    return f;
  };
  double x0, x1 = 0.0;
  const double f0 = refine(g0, best0, x0);
  if (g1 >= 0 && refine(g1, best1, x1) < f0) return x1;
  return x0;
}

// The reference's own update of the double-frequency rotosolve (qmps/tools.py:447-452, qmps/rotosolve.py:233-239):
//   theta = minimize_scalar(f, bounds=[-pi, pi]).x,   f(x) = P sin(2x + u) + Q sin(x + v) = a sin 2x + b cos 2x + c sin x + d cos x.
// With `bounds` and no `method` scipy runs its 'bounded' method - Brent's fmin of Forsythe, Malcolm & Moler (1977) ch. 8:
// golden-section steps, a parabolic step through the three best points whenever its vertex lies inside the bracket and the
// step is below half the one before last; scipy's defaults xatol = 1e-5, at most 500 evaluations, first point
// a + (3 - sqrt 5)/2 (b - a).  It returns a LOCAL minimiser (in 63 of the 486 calls the reference made while the fixtures
// tests/golden/refshim_golden.npz were generated, not the global one), and the optimiser's trajectory depends on which - so
// this, not the global argmin above, is the DEFAULT rule of the device drivers (qmps_set_roto_rule).  Same decision
// sequence as oracle/qmps_oracle.py:fminbound, which is pinned call by call to scipy's recorded answers; ~11 evaluations
// on average (at most 29 in those runs), ONE sincos each.  The result lies in (-pi, pi): the reference's arctan2 wrap of it
// is the identity.
QMPS_HD double double_sinusoid_fminbound(double a, double b, double c, double d) {
  // scipy computes every product and difference below in separate IEEE operations.  hipcc's default -ffp-contract=fast fuses ACROSS
  // statements - q - r with q, r the two nearly equal products of the parabola becomes one fma - and the search then takes other
  // decisions: measured on gfx950 against the reference's 486 recorded calls, 200 differ (a few end in the other basin) with
  // contraction, none without (profiles/experiments/r05/rule_variants.hip).  So: no contraction in this function.
#pragma clang fp contract(off)
  auto f = [&](double x) {
    double sx, cx;
    sincos(x, &sx, &cx);
    const double s2 = 2.0 * sx * cx, c2 = rfma(cx, cx, -sx * sx);
    return rfma(a, s2, rfma(b, c2, rfma(c, sx, d * cx)));
  };
  constexpr double SQRT_EPS = 1.4832396974191326e-08;      // sqrt(2.2e-16)
  constexpr double GM = 0.3819660112501051;                // (3 - sqrt 5) / 2
  constexpr double XATOL3 = 1e-5 / 3.0;
  double lo = -3.141592653589793, hi = 3.141592653589793;
  double fulc = lo + GM * (hi - lo), nfc = fulc, xf = fulc, rat = 0.0, e = 0.0;
  double fx = f(xf), ffulc = fx, fnfc = fx;
  double xm = 0.5 * (lo + hi), tol1 = SQRT_EPS * fabs(xf) + XATOL3, tol2 = 2.0 * tol1;
  int num = 1;
  while (fabs(xf - xm) > (tol2 - 0.5 * (hi - lo))) {
    bool golden = true;
    if (fabs(e) > tol1) {
      golden = false;
      double r = (xf - nfc) * (fx - ffulc);
      double q = (xf - fulc) * (fx - fnfc);
      double p = (xf - fulc) * q - (xf - nfc) * r;
      q = 2.0 * (q - r);
      if (q > 0.0) p = -p;
      q = fabs(q);
      r = e;
      e = rat;
      if (fabs(p) < fabs(0.5 * q * r) && p > q * (lo - xf) && p < q * (hi - xf)) {
        rat = p / q;
        const double x = xf + rat;
        if ((x - lo) < tol2 || (hi - x) < tol2) rat = (xm >= xf) ? tol1 : -tol1;
      } else {
        golden = true;
      }
    }
    if (golden) {
      e = (xf >= xm) ? lo - xf : hi - xf;
      rat = GM * e;
    }
    const double x = xf + (rat >= 0.0 ? 1.0 : -1.0) * fmax(fabs(rat), tol1);
    const double fu = f(x);
    ++num;
    if (fu <= fx) {
      if (x >= xf) lo = xf; else hi = xf;
      fulc = nfc; ffulc = fnfc;
      nfc = xf; fnfc = fx;
      xf = x; fx = fu;
    } else {
      if (x < xf) lo = x; else hi = x;
      if (fu <= fnfc || nfc == xf) {
        fulc = nfc; ffulc = fnfc;
        nfc = x; fnfc = fu;
      } else if (fu <= ffulc || fulc == xf || fulc == nfc) {
        fulc = x; ffulc = fu;
      }
    }
    xm = 0.5 * (lo + hi);
    tol1 = SQRT_EPS * fabs(xf) + XATOL3;
    tol2 = 2.0 * tol1;
    if (num >= 500) break;
  }
  return xf;
}

// rule: QMPS_ROTO_REFERENCE (0) = the reference's bounded Brent search; QMPS_ROTO_GLOBAL_ARGMIN (1) = the global minimiser of the
// fit, wrapped into [-pi, pi] (a departure from tools.py:451 - never worse on the fitted curve, different trajectory).
QMPS_HD double double_sinusoid_step(double a, double b, double c, double d, int rule) {
  if (rule == 0) return double_sinusoid_fminbound(a, b, c, d);
  const double theta = double_sinusoid_argmin(a, b, c, d);
  return theta < -3.141592653589793 ? theta + 6.283185307179586 : (theta > 3.141592653589793 ? theta - 6.283185307179586 : theta);
}

}  // namespace qmps
