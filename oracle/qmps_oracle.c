/* TEST INFRASTRUCTURE ONLY - plain-C restatement ("oracle") of the qmps two-site-energy path.
 *
 * Not product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * this library, as the checker / the timed CPU baseline ("port").  The product path
 * (qmps_amd + libqmps_hip.so) never links or calls it.
 *
 * What it restates (reference = fergusfinn/qmps):
 *   - unitary_to_tensor            qmps/tools.py:151-154      (A[s,i,j] = U[2i+s, j], j < D)
 *   - right environment            qmps/tools.py:176-182 -> xmps TransferMatrix(A).eigs()
 *                                  (external, un-pinned): dominant right eigen-matrix of
 *                                  r -> sum_s A_s r A_s^+, found here by the normalised power
 *                                  iteration of `krylov` (Power Method.ipynb cells 5-6; quantum
 *                                  statement: PowerCircuit qmps/represent.py:235-248)
 *   - Cholesky positive-definiteness check   qmps/tools.py:182, qmps/ground_state.py:153-157
 *   - two-site energy              qmps/ground_state.py:159-167 in closed form
 *                                  E = Re sum h[s,t] tr(B_t r B_s^+)/tr r, B_{2 s1+s2}=A_s1 A_s2
 *                                  (merge: qmps/time_evolve_tools.py:20-23)
 * Parity pinning: see the header of oracle/qmps_oracle.py (numpy twin, checked against the
 * reference's golden vectors); tests/test_oracle.py requires this C file == the numpy twin.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DMAX 16
typedef struct { double re, im; } cplx;

static inline cplx cmul(cplx a, cplx b) { cplx c = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return c; }
static inline cplx cmulc(cplx a, cplx b) { /* a * conj(b) */ cplx c = {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; return c; }
static inline cplx cadd(cplx a, cplx b) { cplx c = {a.re + b.re, a.im + b.im}; return c; }

/* C = X * Y (D x D) */
static void matmul(int D, const cplx* X, const cplx* Y, cplx* C) {
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      cplx acc = {0, 0};
      for (int k = 0; k < D; ++k) acc = cadd(acc, cmul(X[i * D + k], Y[k * D + j]));
      C[i * D + j] = acc;
    }
}
/* C = X * Y^+ */
static void matmul_h(int D, const cplx* X, const cplx* Y, cplx* C) {
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) {
      cplx acc = {0, 0};
      for (int k = 0; k < D; ++k) acc = cadd(acc, cmulc(X[i * D + k], Y[j * D + k]));
      C[i * D + j] = acc;
    }
}

/* one evaluation; A = [2][D][D] complex, h = [nt][4][4] complex, r0 nullable [D][D] */
/* hermitise + trace-normalise rn in place */
static void herm_normalise(int D, cplx* rn) {
  double tr = 0;
  for (int i = 0; i < D; ++i) tr += rn[i * D + i].re;
  for (int i = 0; i < D; ++i)
    for (int j = i; j < D; ++j) {
      cplx a = rn[i * D + j], b = rn[j * D + i];
      cplx m = {0.5 * (a.re + b.re) / tr, 0.5 * (a.im - b.im) / tr};
      if (i == j) m.im = 0;
      rn[i * D + j] = m;
      rn[j * D + i].re = m.re; rn[j * D + i].im = -m.im;
    }
}

/* Repeated-squaring tail (D <= 4): P_m = T^(2^m) as a D^2 x D^2 matrix, r_m = herm(P_m r_C)/tr,
 * stop when ||r_m - r_{m-1}||_F^2 < tol^2; returns the equivalent number of power steps.
 * period > 0 (the D = 4 kernel): after the `skip` squarings the power method continues with P_m itself,
 * r <- herm(P_m r)/tr (one product = 2^m power steps, same stopping rule), and P_m is squared once more
 * after every `period` unconverged products; e0_start: r_C = |0><0|, the chain starts at herm(P_skip r_C)/tr. */
static int squaring_tail(int D, const cplx* A, cplx* r, int done, int max_iter, double tol2, int skip, int period,
                         int e0_start, int* st) {
  const int n = D * D;
  cplx P[16 * 16], Q[16 * 16];   /* D <= 4 here: n = D^2 <= 16 (was 2 MB of thread-local storage per OpenMP thread) */
  cplx rC[256], rn[256];
  for (int i = 0; i < D; ++i) for (int ip = 0; ip < D; ++ip)
    for (int j = 0; j < D; ++j) for (int jp = 0; jp < D; ++jp) {
      cplx acc = {0, 0};
      for (int s = 0; s < 2; ++s) acc = cadd(acc, cmulc(A[s * n + i * D + j], A[s * n + ip * D + jp]));
      P[(i * D + ip) * n + (j * D + jp)] = acc;
    }
  memcpy(rC, r, sizeof(cplx) * n);
  int m = 0, it = done;
  /* untracked squarings: the comparison chain starts at r_skip = herm(P_skip r_C)/tr */
  while (m < skip && done + (1 << (m + 1)) <= max_iter) {
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) {
        cplx acc = {0, 0};
        for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], P[k * n + b]));
        Q[a * n + b] = acc;
      }
    memcpy(P, Q, sizeof(cplx) * n * n);
    ++m;
  }
  if (period > 0) {
    if (m > 0 && e0_start) {
      for (int a = 0; a < n; ++a) {
        cplx acc = {0, 0};
        for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], rC[k]));
        r[a] = acc;
      }
      herm_normalise(D, r);
      it = done + (1 << m);
    }
    int count = 0;
    while ((long)it + (1L << m) <= max_iter) {
      for (int a = 0; a < n; ++a) {
        cplx acc = {0, 0};
        for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], r[k]));
        rn[a] = acc;
      }
      double lam = 0;
      for (int i = 0; i < D; ++i) lam += rn[i * D + i].re;
      herm_normalise(D, rn);
      it += 1 << m;
      double d2 = 0;
      for (int e = 0; e < n; ++e) {
        double dr = rn[e].re - r[e].re, di = rn[e].im - r[e].im;
        d2 += dr * dr + di * di;
      }
      memcpy(r, rn, sizeof(cplx) * n);
      if (d2 < tol2) { *st = 0; return it; }
      if (++count == period && m < 29 && (long)it + (2L << m) <= max_iter) {
        const double sc = 1.0 / (lam * lam);
        for (int a = 0; a < n; ++a)
          for (int b = 0; b < n; ++b) {
            cplx acc = {0, 0};
            for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], P[k * n + b]));
            Q[a * n + b].re = acc.re * sc; Q[a * n + b].im = acc.im * sc;
          }
        memcpy(P, Q, sizeof(cplx) * n * n);
        ++m;
        count = 0;
      }
    }
    *st = 1;
    return it;
  }
  if (m > 0) {
    for (int a = 0; a < n; ++a) {
      cplx acc = {0, 0};
      for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], rC[k]));
      r[a] = acc;
    }
    herm_normalise(D, r);
    it = done + (1 << m);
  }
  while (done + (1 << (m + 1)) <= max_iter && m < 29) {
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) {
        cplx acc = {0, 0};
        for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], P[k * n + b]));
        Q[a * n + b] = acc;
      }
    memcpy(P, Q, sizeof(cplx) * n * n);
    ++m;
    for (int a = 0; a < n; ++a) {
      cplx acc = {0, 0};
      for (int k = 0; k < n; ++k) acc = cadd(acc, cmul(P[a * n + k], rC[k]));
      rn[a] = acc;
    }
    herm_normalise(D, rn);
    double d2 = 0;
    for (int e = 0; e < n; ++e) {
      double dr = rn[e].re - r[e].re, di = rn[e].im - r[e].im;
      d2 += dr * dr + di * di;
    }
    memcpy(r, rn, sizeof(cplx) * n);
    it = done + (1 << m);
    if (d2 < tol2) { *st = 0; return it; }
  }
  /* the budget ran out between two powers of two (max_iter = 10 000: the last comparison is r_8192 against r_4096): the plain method's own
   * test on the last iterate, one application of T itself - accepted with one more iteration (round 5, as squaring_tail_d2 of the kernel) */
  if ((long)it + 1 <= max_iter) {
    cplx X[256], T[256];
    for (int e = 0; e < n; ++e) { rn[e].re = 0; rn[e].im = 0; }
    for (int s = 0; s < 2; ++s) {
      matmul(D, A + s * n, r, X);
      matmul_h(D, X, A + s * n, T);
      for (int e = 0; e < n; ++e) rn[e] = cadd(rn[e], T[e]);
    }
    herm_normalise(D, rn);
    double d2 = 0;
    for (int e = 0; e < n; ++e) {
      double dr = rn[e].re - r[e].re, di = rn[e].im - r[e].im;
      d2 += dr * dr + di * di;
    }
    if (d2 < tol2) {
      memcpy(r, rn, sizeof(cplx) * n);
      *st = 0;
      return it + 1;
    }
  }
  *st = 1;
  return it;
}

static void eval_one(int D, const cplx* A, const cplx* h, int nt, const cplx* r0, int max_iter, double tol,
                     int handoff, int skip, int period, double* E, int* iters, int* status, cplx* r_out, cplx* rho_out) {
  cplx r[DMAX * DMAX], rn[DMAX * DMAX], X[DMAX * DMAX], T[DMAX * DMAX];
  const int n = D * D;
  if (r0) { memcpy(r, r0, sizeof(cplx) * n); herm_normalise(D, r); }
  else
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        /* default start 1/D; squaring from the start (handoff == 0) starts from |0><0| */
        r[i * D + j].re = (i == j) ? (handoff == 0 ? (i == 0 ? 1.0 : 0.0) : 1.0 / D) : 0.0;
        r[i * D + j].im = 0.0;
      }
  int it = 0, st = 1;
  const double tol2 = tol * tol;
  const int plain = (handoff >= 0 && handoff < max_iter) ? handoff : max_iter;
  for (int k = 1; k <= plain; ++k) {
    /* rn = sum_s A_s r A_s^+ */
    for (int e = 0; e < n; ++e) { rn[e].re = 0; rn[e].im = 0; }
    for (int s = 0; s < 2; ++s) {
      matmul(D, A + s * n, r, X);
      matmul_h(D, X, A + s * n, T);
      for (int e = 0; e < n; ++e) rn[e] = cadd(rn[e], T[e]);
    }
    herm_normalise(D, rn);
    double d2 = 0;
    for (int e = 0; e < n; ++e) {
      double dr = rn[e].re - r[e].re, di = rn[e].im - r[e].im;
      d2 += dr * dr + di * di;
    }
    memcpy(r, rn, sizeof(cplx) * n);
    it = k;
    if (d2 < tol2) { st = 0; break; }
  }
  if (st == 1 && handoff >= 0 && plain < max_iter && D <= 4) it = squaring_tail(D, A, r, plain, max_iter, tol2, skip, period, handoff == 0 && !r0, &st);
  /* Cholesky positive-definiteness check (LAPACK zpotrf criterion: pivot <= 0 or NaN fails) */
  if (st == 0) {
    cplx L[DMAX * DMAX];
    memset(L, 0, sizeof(L));
    for (int j = 0; j < D && st == 0; ++j) {
      double d = r[j * D + j].re;
      for (int k = 0; k < j; ++k) d -= L[j * D + k].re * L[j * D + k].re + L[j * D + k].im * L[j * D + k].im;
      if (!(d > 0.0)) { st = 2; break; }
      double ljj = sqrt(d);
      L[j * D + j].re = ljj;
      for (int i = j + 1; i < D; ++i) {
        cplx acc = r[i * D + j];
        for (int k = 0; k < j; ++k) { cplx p = cmulc(L[i * D + k], L[j * D + k]); acc.re -= p.re; acc.im -= p.im; }
        L[i * D + j].re = acc.re / ljj; L[i * D + j].im = acc.im / ljj;
      }
    }
  }
  /* rho[t][s] = tr(B_t r B_s^+)/tr r,  B_{2 s1 + s2} = A_s1 A_s2 */
  cplx Bm[4][DMAX * DMAX], Br[4][DMAX * DMAX];
  for (int s1 = 0; s1 < 2; ++s1)
    for (int s2 = 0; s2 < 2; ++s2) {
      matmul(D, A + s1 * n, A + s2 * n, Bm[2 * s1 + s2]);
      matmul(D, Bm[2 * s1 + s2], r, Br[2 * s1 + s2]);
    }
  double tr = 0;
  for (int i = 0; i < D; ++i) tr += r[i * D + i].re;
  cplx rho[16];
  for (int t = 0; t < 4; ++t)
    for (int s = 0; s < 4; ++s) {
      cplx acc = {0, 0};
      for (int e = 0; e < n; ++e) acc = cadd(acc, cmulc(Br[t][e], Bm[s][e]));
      rho[t * 4 + s].re = acc.re / tr; rho[t * 4 + s].im = acc.im / tr;
    }
  for (int q = 0; q < nt; ++q) {
    double e = 0;
    for (int s = 0; s < 4; ++s)
      for (int t = 0; t < 4; ++t) {
        cplx hv = h[q * 16 + s * 4 + t], rv = rho[t * 4 + s];
        e += hv.re * rv.re - hv.im * rv.im;
      }
    E[q] = e;
  }
  *iters = it; *status = st;
  if (r_out) memcpy(r_out, r, sizeof(cplx) * n);
  if (rho_out) memcpy(rho_out, rho, sizeof(cplx) * 16);
}

/* Batched entry point.  A: [B][2][D][D] complex128 (numpy C order); h: [nt][4][4]; r0 nullable
 * [B][D][D]; E: [B][nt]; iters,status: [B]; r_out nullable [B][D][D]; rho_out nullable [B][4][4].
 * threads <= 0 -> 1.  handoff >= 0: plain power steps before the repeated-squaring tail (0 = squaring
 * from the start); handoff < 0: plain power iteration only; skip, period: see squaring_tail.
 * Returns 0, or -1 on bad arguments. */
int qmps_oracle_energy_batch(int D, long B, const double* A, const double* h, int nt, const double* r0,
                             int max_iter, double tol, double* E, int* iters, int* status, double* r_out,
                             double* rho_out, int threads, int handoff, int skip, int period) {
  if (D < 1 || D > DMAX || B < 0 || nt < 1 || !A || !h || !E || !iters || !status) return -1;
  const long n = (long)D * D;
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
#endif
  for (long b = 0; b < B; ++b)
    eval_one(D, (const cplx*)A + b * 2 * n, (const cplx*)h, nt, r0 ? (const cplx*)r0 + b * n : NULL, max_iter, tol,
             handoff, skip, period, E + b * nt, iters + b, status + b, r_out ? (cplx*)r_out + b * n : NULL,
             rho_out ? (cplx*)rho_out + b * 16 : NULL);
  (void)threads;
  return 0;
}

/* unitary_to_tensor, batched: U [B][2D][2D] -> A [B][2][D][D] */
int qmps_oracle_unitary_to_tensor(int D, long B, const double* U, double* A) {
  if (D < 1 || !U || !A) return -1;
  const cplx* u = (const cplx*)U;
  cplx* a = (cplx*)A;
  const long N = 2L * D;
  for (long b = 0; b < B; ++b)
    for (int s = 0; s < 2; ++s)
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) a[((b * 2 + s) * D + i) * D + j] = u[(b * N + (2 * i + s)) * N + j];
  return 0;
}

int qmps_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
