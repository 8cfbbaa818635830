// qmps_brickwall.hip - the brick-wall ("new_tdvp") classical contractions (new_tdvp/ClassicalTDVPStripped.py) and the
// variational-environment objective (qmps/ground_state.py:170-228): one evaluation per lane, state vectors in registers.
// Split out of qmps_kernels.hip in round 3.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_knobs.h"
#include "qmps_device.h"
#include "qmps_circuit.h"

namespace qmps {

// ------------------------------------------------------------------------------------------
// Kernel 3e: brick-wall ("new_tdvp") classical contractions (SURVEY 8(a)-11 / (f)-4;
// new_tdvp/ClassicalTDVPStripped.py), one evaluation per lane, state vectors in registers.
//   psi_l = (1 x U1^(l-1) x 1)(U2^l)|0..0> on 2 l qubits (bwMPS.state :179-191)
//   expectation values  <psi_l| 1 x O x 1 |psi_l>,  l = 2 (O 4x4, :511-544) and l = 3 (O 16x16, :464-496)
//   environment matrices of RightEnvironment / LeftEnvironment.exact_environment_circuit (:399-422, :316-338)
//     and their dominant eigenpair with the reference's rule eta[np.argmax(eta)] (largest REAL part)
//   ManifoldOverlap.circuit (:239-275)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_u4(const double2* p, double (&gr)[16], double (&gi)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double2 v = p[k];
    gr[k] = v.x;
    gi[k] = v.y;
  }
}
__device__ __forceinline__ void load_u4_dagger(const double2* p, double (&gr)[16], double (&gi)[16]) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double2 v = p[c * 4 + a];
      gr[a * 4 + c] = v.x;
      gi[a * 4 + c] = -v.y;
    }
}

template <int L>
__global__ __launch_bounds__(64) void bw_expval_kernel(BwArgs p) {
  constexpr int NQ = 2 * L;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
  Reg<NQ> r;
#pragma unroll
  for (int x = 0; x < Reg<NQ>::N; ++x) { r.re[x] = (x == 0) ? 1.0 : 0.0; r.im[x] = 0.0; }
  double gr[16], gi[16];
  load_u4((const double2*)p.U2 + b * 16, gr, gi);
#pragma unroll
  for (int k = 0; k < L; ++k) r.u4(2 * k, 2 * k + 1, gr, gi);
  load_u4((const double2*)p.U1 + b * 16, gr, gi);
#pragma unroll
  for (int k = 0; k < L - 1; ++k) r.u4(2 * k + 1, 2 * k + 2, gr, gi);
  // <psi| 1 x O x 1 |psi>: the operator acts on qubits 1 .. NQ-2 (index bits NQ-2 .. 1)
  constexpr int NO = 1 << (NQ - 2);
  const double2* O = (const double2*)p.O + (p.o_shared ? 0 : b * (int64_t)NO * NO);
  double er = 0.0, ei = 0.0;
#pragma unroll
  for (int xm = 0; xm < NO; ++xm)
#pragma unroll
    for (int ym = 0; ym < NO; ++ym) {
      const double2 o = O[xm * NO + ym];
      // sum over the outer bits of conj(psi[hi, xm, lo]) psi[hi, ym, lo]
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
          const int x = (hi << (NQ - 1)) | (xm << 1) | lo, y = (hi << (NQ - 1)) | (ym << 1) | lo;
          sr = dfma(r.re[x], r.re[y], sr);
          sr = dfma(r.im[x], r.im[y], sr);
          si = dfma(r.re[x], r.im[y], si);
          si = dfma(-r.im[x], r.re[y], si);
        }
      er = dfma(o.x, sr, er);
      er = dfma(-o.y, si, er);
      ei = dfma(o.x, si, ei);
      ei = dfma(o.y, sr, ei);
    }
  ((double2*)p.out)[b] = make_double2(er, ei);
}

// exp((1 - i eps) M) by Taylor series (||M|| <= ~1: transfer matrices of unitaries), then repeated squaring:
// the dominant-modulus eigenvector of exp(cM) is the eigenvector of M with the largest real part (ties broken
// towards the larger imaginary part by the -i eps tilt) - the reference's eta[np.argmax(eta)].
__device__ __forceinline__ void mat4_mul(const double (&ar)[4][4], const double (&ai)[4][4], const double (&br)[4][4],
                                         const double (&bi)[4][4], double (&cr)[4][4], double (&ci)[4][4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double xr = 0.0, xi = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xr += ar[a][k] * br[k][c] - ai[a][k] * bi[k][c];
        xi += ar[a][k] * bi[k][c] + ai[a][k] * br[k][c];
      }
      cr[a][c] = xr;
      ci[a][c] = xi;
    }
}

__global__ __launch_bounds__(64) void bw_env_kernel(BwArgs p) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
  // phi_i = U1 U2 |i,0,0> (right: U2 on (b,c), U1 on (a,b), open wire a = qubit 0)
  //         (left : U2 on (a,b), U1 on (b,c), open wire c = qubit 2)
  // chi_i' = (U2' U1')^+ |i',0,0>  resp. mirrored;  Mmat[(i,i'),(j,j')] = sum_{rest} conj(chi_i'[..j'..]) phi_i[..j..]
  const bool left = p.side != 0;
  Reg<3> phi[2], chi[2];
  double gr[16], gi[16];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int start = left ? i : (i << 2);
      phi[i].re[x] = (x == start) ? 1.0 : 0.0; phi[i].im[x] = 0.0;
      chi[i].re[x] = (x == start) ? 1.0 : 0.0; chi[i].im[x] = 0.0;
    }
  }
  load_u4((const double2*)p.U2 + b * 16, gr, gi);
#pragma unroll
  for (int i = 0; i < 2; ++i) { if (left) phi[i].u4(0, 1, gr, gi); else phi[i].u4(1, 2, gr, gi); }
  load_u4((const double2*)p.U1 + b * 16, gr, gi);
#pragma unroll
  for (int i = 0; i < 2; ++i) { if (left) phi[i].u4(1, 2, gr, gi); else phi[i].u4(0, 1, gr, gi); }
  load_u4_dagger((const double2*)p.U2p + b * 16, gr, gi);
#pragma unroll
  for (int i = 0; i < 2; ++i) { if (left) chi[i].u4(0, 1, gr, gi); else chi[i].u4(1, 2, gr, gi); }
  load_u4_dagger((const double2*)p.U1p + b * 16, gr, gi);
#pragma unroll
  for (int i = 0; i < 2; ++i) { if (left) chi[i].u4(1, 2, gr, gi); else chi[i].u4(0, 1, gr, gi); }
  // open wire carrying (j, j'): right -> qubit 2 (bit 0); left -> qubit 0 (bit 2)
  double mr[4][4], mi[4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ip = 0; ip < 2; ++ip)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          double xr = 0.0, xi = 0.0;
#pragma unroll
          for (int rest = 0; rest < 4; ++rest) {
            const int xphi = left ? ((j << 2) | rest) : ((rest << 1) | j);
            const int xchi = left ? ((jp << 2) | rest) : ((rest << 1) | jp);
            xr += chi[ip].re[xchi] * phi[i].re[xphi] + chi[ip].im[xchi] * phi[i].im[xphi];
            xi += chi[ip].re[xchi] * phi[i].im[xphi] - chi[ip].im[xchi] * phi[i].re[xphi];
          }
          mr[2 * i + ip][2 * j + jp] = xr;
          mi[2 * i + ip][2 * j + jp] = xi;
        }
  if (p.mat_out != nullptr) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) ((double2*)p.mat_out)[b * 16 + a * 4 + c] = make_double2(mr[a][c], mi[a][c]);
  }
  // P = exp((1 - i eps) M), 20-term Taylor series evaluated by Horner
  const double eps = 1e-6;
  double cr[4][4], ci[4][4], pr[4][4], pi[4][4], tr_[4][4], ti_[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      cr[a][c] = mr[a][c] + eps * mi[a][c];       // (1 - i eps)(mr + i mi)
      ci[a][c] = mi[a][c] - eps * mr[a][c];
      pr[a][c] = (a == c) ? 1.0 : 0.0;
      pi[a][c] = 0.0;
    }
  for (int k = 20; k >= 1; --k) {
    mat4_mul(cr, ci, pr, pi, tr_, ti_);
    const double inv = 1.0 / k;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        pr[a][c] = tr_[a][c] * inv + ((a == c) ? 1.0 : 0.0);
        pi[a][c] = ti_[a][c] * inv;
      }
  }
  // (unit Frobenius norm from the start: log_rho accumulates log ||P_0^(2^m)|| / 2^m, a Gelfand bound on log rho(exp(cM)) = max Re(c lambda))
  double log_rho = 0.0;
  {
    double n2 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) n2 += pr[a][c] * pr[a][c] + pi[a][c] * pi[a][c];
    const double inv = 1.0 / __builtin_sqrt(n2);
    log_rho = 0.5 * log(n2);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { pr[a][c] *= inv; pi[a][c] *= inv; }
  }
  double eta_r = 0.0, eta_i = 0.0, vr[4] = {1, 0, 0, 0}, vi[4] = {0, 0, 0, 0};
  int status = QMPS_ST_NOT_CONVERGED, rank1_rounds = 0;
  const double tol2 = p.tol * p.tol;
  for (int m = 0; m <= p.max_rounds; ++m) {
    double best = -1.0;
    int bc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double n2 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) n2 += pr[a][c] * pr[a][c] + pi[a][c] * pi[a][c];
      if (n2 > best) { best = n2; bc = c; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      vr[a] = bc == 0 ? pr[a][0] : (bc == 1 ? pr[a][1] : (bc == 2 ? pr[a][2] : pr[a][3]));
      vi[a] = bc == 0 ? pi[a][0] : (bc == 1 ? pi[a][1] : (bc == 2 ? pi[a][2] : pi[a][3]));
    }
    double wr[4], wi[4], num_r = 0, num_i = 0, vv = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double xr = 0, xi = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xr += mr[a][c] * vr[c] - mi[a][c] * vi[c];
        xi += mr[a][c] * vi[c] + mi[a][c] * vr[c];
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
    // The eigen-residual of the largest column is not enough: at special unitaries (1, SWAP, CNOT, products) a column of exp(cM) itself is an
    // EXACT eigenvector of a lesser eigenvalue - a kernel vector of M, residual 0 at round 0 - and was accepted in place of the eigenvalue with
    // the largest real part (round 5, profiles/experiments/r05/stress_brickwall.py: 4 of 9 360).  The power must also have become RANK ONE,
    // ||P P - tr(P) P|| << ||P P||.  Conversely an ill-conditioned eigenvector (two eigenvalues at 0 next to the leading one, cond ~ 1e3) stalls
    // at a residual of ~1e-13: once the power has been rank one for three rounds a residual below min(1e-10, 1e3 tol) is what there is (status 0; numpy's
    // eig - the reference's route - is no more accurate on those matrices).
    mat4_mul(pr, pi, pr, pi, tr_, ti_);
    double f2 = 0.0, r1 = 0.0, trr = 0.0, tri = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) { trr += pr[a][a]; tri += pi[a][a]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f2 += tr_[a][c] * tr_[a][c] + ti_[a][c] * ti_[a][c];
        const double dr = tr_[a][c] - (trr * pr[a][c] - tri * pi[a][c]), di = ti_[a][c] - (trr * pi[a][c] + tri * pr[a][c]);
        r1 += dr * dr + di * di;
      }
    rank1_rounds = r1 < 1e-22 * f2 ? rank1_rounds + 1 : 0;      // (1e-24 until the fourth stress campaign: rounding leaves r1 / f2 ~ (eps cond)^2, 1.3e-25 at cond 1 800 - borderline)
    // A DEGENERATE leading eigenvalue (several eigenvectors, one eigenvalue: 1 x 1, SWAP ...) never gives a rank-one power, and any of its
    // eigenvectors is an answer (numpy returns one of them): from round 30 on the column is also accepted if ITS eigenvalue has the largest
    // real part there is - the growth rate of the power, log rho(exp(cM)) = max Re(c lambda), to 2^-30 ln(condition) ~ 3e-8.
    const bool leading = m >= 30 && eta_r + eps * eta_i >= log_rho - 3e-8;
    // (the relaxed acceptance follows the caller's tol - residual < min(1e-10, 1e3 tol), documented with qmps_bw_env: it was a fixed 1e-10 whatever
    // tol asked for - advisor, round 5)
    const double relaxed2 = fmin(1e-20, 1e6 * tol2);
    if ((res < tol2 * vv && (r1 < 1e-20 * f2 || leading)) || (rank1_rounds >= 3 && res < relaxed2 * vv)) { status = QMPS_ST_OK; break; }
    if (m == p.max_rounds) break;
    log_rho += ldexp(0.5 * log(f2), -(m + 1));
    const double inv = f2 > 0.0 ? 1.0 / __builtin_sqrt(f2) : 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { pr[a][c] = tr_[a][c] * inv; pi[a][c] = ti_[a][c] * inv; }
  }
  // unit 2-norm, phase: largest-magnitude entry real positive
  double n2 = 0.0, bigr = 1.0, bigi = 0.0, bigm = -1.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const double m2 = vr[a] * vr[a] + vi[a] * vi[a];
    n2 += m2;
    if (m2 > bigm) { bigm = m2; bigr = vr[a]; bigi = vi[a]; }
  }
  const double sc = 1.0 / (__builtin_sqrt(n2) * __builtin_sqrt(bigm));
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const double xr = (vr[a] * bigr + vi[a] * bigi) * sc, xi = (vi[a] * bigr - vr[a] * bigi) * sc;
    ((double2*)p.vec_out)[b * 4 + a] = make_double2(xr, xi);
  }
  ((double2*)p.out)[b] = make_double2(eta_r, eta_i);
  p.status[b] = status;
}

// ManifoldOverlap.circuit (new_tdvp/ClassicalTDVPStripped.py:239-275) without a 6-qubit state vector (the literal
// simulation of round 1 held 64 amplitudes per lane and spilled 882 registers).  With a = U2[:, 0] (the pair state
// U2|00>), b = U2'[0, :] (the bra <00|U2') and big-endian two-bit indices,
//   ket(y0; y12; y34; y5) = sum_z U1[y12, z1 z2] U1[y34, z3 z4] a[y0 z1] a[z2 z3] a[z4 y5]
//   bra(x0; x12; x34; x5) = sum_w b[x0 w1] b[w2 w3] b[w4 x5] U1'[w1 w2, x12] U1'[w3 w4, x34]
//   out = sum Ml[x0, y0] Mr[x5, y5] bra(x) W[x12 x34, y12 y34] ket(y)
// Both vectors have rank 2 across the middle bond: ket = sum_c KL[y0][y12][c] KR[y5][y34][c] (the middle pair a[z2 z3] folded
// into KL), bra likewise with Ml, Mr folded into BL, BR.  So out = sum_{y0, y5} <B_{y0 y5}| W |K_{y0 y5}> : four 16 x 16
// sandwiches, the 16-vectors rebuilt from their 4 x 2 factors on the fly.
__global__ __launch_bounds__(64) void bw_manifold_kernel(BwArgs p) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
  const double2* U1 = (const double2*)p.U1 + b * 16;
  const double2* U2 = (const double2*)p.U2 + b * 16;
  const double2* U1p = (const double2*)p.U1p + b * 16;
  const double2* U2p = (const double2*)p.U2p + b * 16;
  const double2* Ml = (const double2*)p.Ml + (p.m_shared ? 0 : b * 4);
  const double2* Mr = (const double2*)p.Mr + (p.m_shared ? 0 : b * 4);
  const double2* W = (const double2*)p.O + (p.o_shared ? 0 : b * 256);
  auto cm = [](double2 x, double2 y) { return make_double2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x); };
  auto acc = [](double2& t, double2 x, double2 y) {
    t.x = dfma(x.x, y.x, t.x);
    t.x = dfma(-x.y, y.y, t.x);
    t.y = dfma(x.x, y.y, t.y);
    t.y = dfma(x.y, y.x, t.y);
  };
  double2 a[4], bb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = U2[k * 4];        // column 0 of U2
    bb[k] = U2p[k];          // row 0 of U2'
  }
  // ket factors: KL[y0][y12][z3] = sum_{z1 z2} U1[y12, z1 z2] a[y0 z1] a[z2 z3];  KR[y5][y34][z3] = sum_{z4} U1[y34, z3 z4] a[z4 y5]
  double2 KL[2][4][2], KR[2][4][2];
  {
    double2 u1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) u1[k] = U1[k];
#pragma unroll
    for (int y0 = 0; y0 < 2; ++y0)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        double2 t[2];     // sum_{z1} U1[y, z1 z2] a[y0 z1], z2 = 0, 1
#pragma unroll
        for (int z2 = 0; z2 < 2; ++z2) {
          t[z2] = make_double2(0.0, 0.0);
#pragma unroll
          for (int z1 = 0; z1 < 2; ++z1) acc(t[z2], u1[y * 4 + 2 * z1 + z2], a[2 * y0 + z1]);
        }
#pragma unroll
        for (int z3 = 0; z3 < 2; ++z3) {
          double2 v = make_double2(0.0, 0.0);
#pragma unroll
          for (int z2 = 0; z2 < 2; ++z2) acc(v, t[z2], a[2 * z2 + z3]);
          KL[y0][y][z3] = v;
        }
      }
#pragma unroll
    for (int y5 = 0; y5 < 2; ++y5)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int z3 = 0; z3 < 2; ++z3) {
          double2 v = make_double2(0.0, 0.0);
#pragma unroll
          for (int z4 = 0; z4 < 2; ++z4) acc(v, u1[y * 4 + 2 * z3 + z4], a[2 * z4 + y5]);
          KR[y5][y][z3] = v;
        }
  }
  // bra factors with the boundary matrices folded in:
  //   BL[y0][x12][w3] = sum_{x0} Ml[x0, y0] sum_{w1 w2} b[x0 w1] U1'[w1 w2, x12] b[w2 w3]
  //   BR[y5][x34][w3] = sum_{x5} Mr[x5, y5] sum_{w4} U1'[w3 w4, x34] b[w4 x5]
  double2 BL[2][4][2], BR[2][4][2];
  {
    double2 u1p[16], ml[4], mr[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) u1p[k] = U1p[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ml[k] = Ml[k]; mr[k] = Mr[k]; }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      double2 raw[2][2];   // [x0][w3]
#pragma unroll
      for (int x0 = 0; x0 < 2; ++x0) {
        double2 t[2];      // sum_{w1} b[x0 w1] U1'[w1 w2, x], w2 = 0, 1
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          t[w2] = make_double2(0.0, 0.0);
#pragma unroll
          for (int w1 = 0; w1 < 2; ++w1) acc(t[w2], bb[2 * x0 + w1], u1p[(2 * w1 + w2) * 4 + x]);
        }
#pragma unroll
        for (int w3 = 0; w3 < 2; ++w3) {
          double2 v = make_double2(0.0, 0.0);
#pragma unroll
          for (int w2 = 0; w2 < 2; ++w2) acc(v, t[w2], bb[2 * w2 + w3]);
          raw[x0][w3] = v;
        }
      }
#pragma unroll
      for (int y0 = 0; y0 < 2; ++y0)
#pragma unroll
        for (int w3 = 0; w3 < 2; ++w3) {
          double2 v = cm(ml[0 * 2 + y0], raw[0][w3]);
          acc(v, ml[1 * 2 + y0], raw[1][w3]);
          BL[y0][x][w3] = v;
        }
      double2 rawr[2][2];  // [x5][w3]
#pragma unroll
      for (int x5 = 0; x5 < 2; ++x5)
#pragma unroll
        for (int w3 = 0; w3 < 2; ++w3) {
          double2 v = make_double2(0.0, 0.0);
#pragma unroll
          for (int w4 = 0; w4 < 2; ++w4) acc(v, u1p[(2 * w3 + w4) * 4 + x], bb[2 * w4 + x5]);
          rawr[x5][w3] = v;
        }
#pragma unroll
      for (int y5 = 0; y5 < 2; ++y5)
#pragma unroll
        for (int w3 = 0; w3 < 2; ++w3) {
          double2 v = cm(mr[0 * 2 + y5], rawr[0][w3]);
          acc(v, mr[1 * 2 + y5], rawr[1][w3]);
          BR[y5][x][w3] = v;
        }
    }
  }
  // out = sum_{y0, y5} sum_{x, y} B_{y0 y5}(x) W[x, y] K_{y0 y5}(y),  x = 4 x12 + x34,  y = 4 y12 + y34
  double2 out = make_double2(0.0, 0.0);
#pragma unroll
  for (int y0 = 0; y0 < 2; ++y0)
#pragma unroll
    for (int y5 = 0; y5 < 2; ++y5) {
      double2 K[16];
#pragma unroll
      for (int yl = 0; yl < 4; ++yl)
#pragma unroll
        for (int yr = 0; yr < 4; ++yr) {
          double2 v = cm(KL[y0][yl][0], KR[y5][yr][0]);
          acc(v, KL[y0][yl][1], KR[y5][yr][1]);
          K[4 * yl + yr] = v;
        }
#pragma unroll
      for (int x = 0; x < 16; ++x) {       // (unrolled: a rolled loop would index BL / BR dynamically and push them to scratch)
        double2 sx = make_double2(0.0, 0.0);
#pragma unroll
        for (int y = 0; y < 16; ++y) acc(sx, W[x * 16 + y], K[y]);
        const int xl = x >> 2, xr = x & 3;
        double2 bx = cm(BL[y0][xl][0], BR[y5][xr][0]);
        acc(bx, BL[y0][xl][1], BR[y5][xr][1]);
        acc(out, bx, sx);
        __builtin_amdgcn_sched_barrier(0);   // row by row: keeps the 1024 loads of W from being hoisted into the register file
      }
    }
  ((double2*)p.out)[b] = out;
}

hipError_t launch_bw(int what, const BwArgs& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  const dim3 grid((unsigned)((a.B + 63) / 64)), block(64);
  switch (what) {
    case 0: hipLaunchKernelGGL(bw_expval_kernel<2>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(bw_expval_kernel<3>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(bw_env_kernel, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(bw_manifold_kernel, grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Kernel 3f: variational-environment objective (qmps/ground_state.py:170-228), D = 2, one
// evaluation per lane.  30 parameters: p2 = params[:15] -> U, p1 = params[15:] -> V (both
// ShallowFullStateTensor).  Four circuits, simulated literally:
//   energy     (4 qubits): V(2,3) U(1,2) U(0,1);                  <1 x H x 1>
//   v_purity   (4 qubits): V(0,1) V(2,3) SWAP(0,1);               <SWAP(1,2)>
//   u_purity   (6 qubits): V(1,2) U(0,1) V(4,5) U(3,4) SWAP(0,1) SWAP(1,2);   <SWAP(2,3)>
//   uv_purity  (5 qubits): V(3,4) U(2,3) V(0,1) SWAP(0,1);        <SWAP(1,2)>
//   f = energy + k (u_purity + v_purity - 2 uv_purity)
// The three purity circuits act on PRODUCT states - psi_V = V|00> on a pair, phi = U(0,1) V(1,2)|000> on a triple - and for
// |alpha> x |beta> the expectation of a SWAP between a qubit of alpha and a qubit of beta is tr(rho_alpha rho_beta).  The
// SWAPs inside each circuit only move qubit 0 of the factor next to the measured cut, so (round 2; the literal 5- and
// 6-qubit simulation of round 1 spilled 1390 registers per lane)
//   v_purity = tr(rho_V^2),  uv_purity = tr(rho_V rho_phi),  u_purity = tr(rho_phi^2),   rho_X = one-qubit state of qubit 0 of X
// Parity: oracle.opt_environment_objective simulates the four circuits literally.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void opt_env_lane_kernel(const double* __restrict__ params, const double2* __restrict__ h,
                                                          double k, double* __restrict__ f, double* __restrict__ parts,
                                                          int64_t B) {
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double cu[15], su[15], cv[15], sv[15];      // cos / sin of the half angles: U = params[:15], V = params[15:]
  for (int i = 0; i < 15; ++i) {               // (a rolled loop: one copy of the sincos expansion)
    sincos(0.5 * params[b * 30 + i], &su[i], &cu[i]);
    sincos(0.5 * params[b * 30 + 15 + i], &sv[i], &cv[i]);
  }
  double energy;
  {
    Reg<4> r;
    r.reset();
    r.shallow_full_cs(2, 3, cv, sv);
    r.shallow_full_cs(1, 2, cu, su);
    r.shallow_full_cs(0, 1, cu, su);
    // <psi| 1 x H x 1 |psi>, H on qubits 1,2 = index bits 2,1
    double e = 0.0;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
      for (int lo = 0; lo < 2; ++lo)
#pragma unroll
        for (int xm = 0; xm < 4; ++xm)
#pragma unroll
          for (int ym = 0; ym < 4; ++ym) {
            const double2 o = h[xm * 4 + ym];
            const int x = (hi << 3) | (xm << 1) | lo, y = (hi << 3) | (ym << 1) | lo;
            // Re( conj(psi[x]) o psi[y] )
            const double yr = o.x * r.re[y] - o.y * r.im[y], yi = o.x * r.im[y] + o.y * r.re[y];
            e += r.re[x] * yr + r.im[x] * yi;
          }
    energy = e;
  }
  double vr[2][2], vi[2][2], fr[2][2], fi[2][2];
  {
    Reg<2> r;
    r.reset();
    r.shallow_full_cs(0, 1, cv, sv);
    r.rdm_q0(vr, vi);
  }
  {
    Reg<3> r;
    r.reset();
    r.shallow_full_cs(1, 2, cv, sv);
    r.shallow_full_cs(0, 1, cu, su);
    r.rdm_q0(fr, fi);
  }
  // tr(X Y) = sum_ab X[a][b] Y[b][a]  (real for Hermitian X, Y)
  auto trprod = [](const double (&xr)[2][2], const double (&xi)[2][2], const double (&yr)[2][2], const double (&yi)[2][2]) {
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) t += xr[a][c] * yr[c][a] - xi[a][c] * yi[c][a];
    return t;
  };
  const double v_purity = trprod(vr, vi, vr, vi), u_purity = trprod(fr, fi, fr, fi), uv_purity = trprod(vr, vi, fr, fi);
  f[b] = energy + k * (u_purity + v_purity - 2.0 * uv_purity);
  if (parts != nullptr) {
    parts[b * 4 + 0] = energy;
    parts[b * 4 + 1] = u_purity;
    parts[b * 4 + 2] = v_purity;
    parts[b * 4 + 3] = uv_purity;
  }
}

hipError_t launch_opt_env(const double* params, const void* h, double k, double* f, double* parts, int64_t B,
                          hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(opt_env_lane_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, params, (const double2*)h, k, f,
                     parts, B);
  return hipGetLastError();
}

}  // namespace qmps
