// qmps_circuit.h - small-register circuit simulator and the ansatz gate lists (gfx950 only), shared by the kernel
// translation units: ansatz_tensor_kernel / the rotosolve kernels (qmps_kernels.hip) and the fused D = 4 kernel, which
// builds the state tensor in front of the environment solve (qmps_direct.hip).  qmps/represent.py:268-404.
#pragma once
#include <hip/hip_runtime.h>

#include "qmps_device.h"

namespace qmps {

template <int NQ>
struct Reg {
  static constexpr int N = 1 << NQ;
  double re[N], im[N];
  __device__ __forceinline__ static constexpr int mask(int q) { return 1 << (NQ - 1 - q); }
  // general single-qubit gate [[a, b], [c, d]] (complex) on qubit q
  __device__ __forceinline__ void u2(int q, double ar, double ai, double br, double bi, double cr, double ci, double dr,
                                     double di) {
    const int m = mask(q);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if (!(x & m)) {
        const double pr = re[x], pi = im[x], qr = re[x | m], qi = im[x | m];
        re[x] = ar * pr - ai * pi + br * qr - bi * qi;
        im[x] = ar * pi + ai * pr + br * qi + bi * qr;
        re[x | m] = cr * pr - ci * pi + dr * qr - di * qi;
        im[x | m] = cr * pi + ci * pr + dr * qi + di * qr;
      }
  }
  // rz / rx with the half-angle cosine and sine given (one sincos per angle, whatever the number of qubits it acts on);
  // only the non-zero entries of the gate are multiplied
  __device__ __forceinline__ void rz_cs(int q, double c, double s) {   // diag(c - i s, c + i s)
    const int m = mask(q);
#pragma unroll
    for (int x = 0; x < N; ++x) {
      const double pr = re[x], pi = im[x];
      if (x & m) {
        re[x] = dfma(c, pr, -s * pi);
        im[x] = dfma(c, pi, s * pr);
      } else {
        re[x] = dfma(c, pr, s * pi);
        im[x] = dfma(c, pi, -s * pr);
      }
    }
  }
  // rz with the same angle on EVERY qubit = one diagonal: amplitude x picks up (c - i s)^(NQ - 2 popcount(x)).  The powers
  // cost a few complex products, then ONE complex multiplication per amplitude instead of NQ (D = 16, five qubits: 128 instead
  // of 640 instructions per layer - a fifth of the ansatz kernel, which is a chain of ~6 500 instructions of one lane).
  __device__ __forceinline__ void rz_all_cs(double c, double s) {
    double zr[NQ + 1], zi[NQ + 1];          // z^m, z = c - i s, m = 0 .. NQ
    zr[0] = 1.0; zi[0] = 0.0;
#pragma unroll
    for (int m = 1; m <= NQ; ++m) {
      zr[m] = dfma(zr[m - 1], c, zi[m - 1] * s);
      zi[m] = dfma(zi[m - 1], c, -zr[m - 1] * s);
    }
#pragma unroll
    for (int x = 0; x < N; ++x) {
      const int e = NQ - 2 * __builtin_popcount((unsigned)x);       // compile-time after unrolling
      const double wr = zr[e < 0 ? -e : e], wi = e < 0 ? -zi[-e] : zi[e];
      const double pr = re[x], pi = im[x];
      re[x] = dfma(wr, pr, -wi * pi);
      im[x] = dfma(wr, pi, wi * pr);
    }
  }
  __device__ __forceinline__ void rx_cs(int q, double c, double s) {   // [[c, -i s], [-i s, c]]
    const int m = mask(q);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if (!(x & m)) {
        const double pr = re[x], pi = im[x], qr = re[x | m], qi = im[x | m];
        re[x] = dfma(c, pr, s * qi);
        im[x] = dfma(c, pi, -s * qr);
        re[x | m] = dfma(c, qr, s * pi);
        im[x | m] = dfma(c, qi, -s * pr);
      }
  }
  __device__ __forceinline__ void had_fast(int q) {
    const double h = 0.70710678118654752;
    const int m = mask(q);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if (!(x & m)) {
        const double pr = re[x], pi = im[x], qr = re[x | m], qi = im[x | m];
        re[x] = h * (pr + qr);
        im[x] = h * (pi + qi);
        re[x | m] = h * (pr - qr);
        im[x | m] = h * (pi - qi);
      }
  }
  __device__ __forceinline__ void rz(int q, double t) {
    double s, c;
    sincos(0.5 * t, &s, &c);
    u2(q, c, -s, 0, 0, 0, 0, c, s);
  }
  __device__ __forceinline__ void rx(int q, double t) {
    double s, c;
    sincos(0.5 * t, &s, &c);
    u2(q, c, 0, 0, -s, 0, -s, c, 0);
  }
  __device__ __forceinline__ void ry(int q, double t) {
    double s, c;
    sincos(0.5 * t, &s, &c);
    u2(q, c, 0, -s, 0, s, 0, c, 0);
  }
  __device__ __forceinline__ void had(int q) {
    const double h = 0.70710678118654752;
    u2(q, h, 0, h, 0, h, 0, -h, 0);
  }
  __device__ __forceinline__ void xpow(int q, double t) {   // cirq.X**t = e^{i pi t/2} (cos I - i sin X)
    double s, c;
    sincos(1.5707963267948966 * t, &s, &c);
    xpow_cs(q, c, s);
  }
  __device__ __forceinline__ void xpow_cs(int q, double c, double s) {   // (c, s) = cos / sin of pi t / 2
    const double ps = s, pc = c;                             // global phase e^{i pi t / 2} = (c + i s)
    // (pc + i ps) * [[c, -i s], [-i s, c]]
    const double dr = pc * c, di = ps * c;                   // diagonal
    const double orr = ps * s, oi = -pc * s;                 // off-diagonal: (pc + i ps)(-i s) = ps s - i pc s
    u2(q, dr, di, orr, oi, orr, oi, dr, di);
  }
  __device__ __forceinline__ void zzpow(int q1, int q2, double t) {   // diag(1, e, e, 1), e = e^{i pi t}
    double s, c;
    sincos(3.141592653589793 * t, &s, &c);
    zzpow_cs(q1, q2, c, s);
  }
  __device__ __forceinline__ void zzpow_cs(int q1, int q2, double c, double s) {   // (c, s) = cos / sin of pi t
    const int m1 = mask(q1), m2 = mask(q2);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if (((x & m1) != 0) != ((x & m2) != 0)) {
        const double pr = re[x], pi = im[x];
        re[x] = c * pr - s * pi;
        im[x] = c * pi + s * pr;
      }
  }
  // general single-qubit matrix m[2][2] (complex, row-major: re/im arrays)
  __device__ __forceinline__ void u2m(int q, const double (&mr_)[4], const double (&mi_)[4]) {
    u2(q, mr_[0], mi_[0], mr_[1], mi_[1], mr_[2], mi_[2], mr_[3], mi_[3]);
  }
  // general two-qubit matrix g[4][4] (complex) on qubits (q1, q2), q1 = more significant bit of the gate index
  __device__ __forceinline__ void u4(int q1, int q2, const double (&gr)[16], const double (&gi)[16]) {
    const int m1 = mask(q1), m2 = mask(q2);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if (!(x & m1) && !(x & m2)) {
        const int idx[4] = {x, x | m2, x | m1, x | m1 | m2};
        double pr[4], pi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pr[k] = re[idx[k]]; pi[k] = im[idx[k]]; }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          double xr = 0.0, xi = 0.0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            xr = dfma(gr[a * 4 + k], pr[k], xr);
            xr = dfma(-gi[a * 4 + k], pi[k], xr);
            xi = dfma(gr[a * 4 + k], pi[k], xi);
            xi = dfma(gi[a * 4 + k], pr[k], xi);
          }
          re[idx[a]] = xr;
          im[idx[a]] = xi;
        }
      }
  }
  __device__ __forceinline__ void swapq(int q1, int q2) {
    const int m1 = mask(q1), m2 = mask(q2);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if ((x & m1) && !(x & m2)) {
        const int y = (x & ~m1) | m2;
        const double pr = re[x], pi = im[x];
        re[x] = re[y]; im[x] = im[y];
        re[y] = pr; im[y] = pi;
      }
  }
  // <psi| SWAP(q1,q2) |psi> (real: SWAP is Hermitian)
  __device__ __forceinline__ double swap_expectation(int q1, int q2) const {
    const int m1 = mask(q1), m2 = mask(q2);
    double e = 0.0;
#pragma unroll
    for (int x = 0; x < N; ++x) {
      const bool b1 = (x & m1) != 0, b2 = (x & m2) != 0;
      const int y = (b1 == b2) ? x : (x ^ m1 ^ m2);
      e = dfma(re[x], re[y], e);
      e = dfma(im[x], im[y], e);
    }
    return e;
  }
  // reduced density matrix of qubit 0 (the most significant index bit): rho[a][b] = sum_rest psi[a, rest] conj(psi[b, rest])
  __device__ __forceinline__ void rdm_q0(double (&rr)[2][2], double (&ri)[2][2]) const {
    constexpr int H = N / 2;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        double xr = 0.0, xi = 0.0;
#pragma unroll
        for (int t = 0; t < H; ++t) {
          xr = dfma(re[a * H + t], re[b * H + t], xr);
          xr = dfma(im[a * H + t], im[b * H + t], xr);
          xi = dfma(im[a * H + t], re[b * H + t], xi);
          xi = dfma(-re[a * H + t], im[b * H + t], xi);
        }
        rr[a][b] = xr;
        ri[a][b] = xi;
      }
  }
  // ShallowFullStateTensor(2, v) (qmps/represent.py:393-401) on qubits (a, b)
  __device__ __forceinline__ void shallow_full(int a, int b, const double* v) {
    rz(a, v[0]); rx(a, v[1]); rz(a, v[2]);
    rz(b, v[3]); rx(b, v[4]); rz(b, v[5]);
    cnot(a, b);
    ry(a, v[6]);
    cnot(b, a);
    ry(a, v[7]); rz(b, v[8]);
    cnot(a, b);
    rz(a, v[9]); rx(a, v[10]); rz(a, v[11]);
    rz(b, v[12]); rx(b, v[13]); rz(b, v[14]);
  }
  // the same gate list with the half-angle cosines / sines already computed (one sincos per angle, whatever the number
  // of circuits the gate appears in)
  __device__ __forceinline__ void shallow_full_cs(int a, int b, const double* c, const double* s) {
    u2(a, c[0], -s[0], 0, 0, 0, 0, c[0], s[0]); u2(a, c[1], 0, 0, -s[1], 0, -s[1], c[1], 0); u2(a, c[2], -s[2], 0, 0, 0, 0, c[2], s[2]);
    u2(b, c[3], -s[3], 0, 0, 0, 0, c[3], s[3]); u2(b, c[4], 0, 0, -s[4], 0, -s[4], c[4], 0); u2(b, c[5], -s[5], 0, 0, 0, 0, c[5], s[5]);
    cnot(a, b);
    u2(a, c[6], 0, -s[6], 0, s[6], 0, c[6], 0);
    cnot(b, a);
    u2(a, c[7], 0, -s[7], 0, s[7], 0, c[7], 0); u2(b, c[8], -s[8], 0, 0, 0, 0, c[8], s[8]);
    cnot(a, b);
    u2(a, c[9], -s[9], 0, 0, 0, 0, c[9], s[9]); u2(a, c[10], 0, 0, -s[10], 0, -s[10], c[10], 0); u2(a, c[11], -s[11], 0, 0, 0, 0, c[11], s[11]);
    u2(b, c[12], -s[12], 0, 0, 0, 0, c[12], s[12]); u2(b, c[13], 0, 0, -s[13], 0, -s[13], c[13], 0); u2(b, c[14], -s[14], 0, 0, 0, 0, c[14], s[14]);
  }
  // cirq.XX**t / cirq.YY**t = (1 + e)/2 + (1 - e)/2 PP, e = e^{i pi t}: eigenvalue 1 on PP = +1, e on PP = -1.
  // XX flips both bits; YY flips both bits with sign -1 when they are equal (Y|0> = i|1>, Y|1> = -i|0>).
  __device__ __forceinline__ void pppow(int q1, int q2, double t, bool yy) {
    double sn, cn;
    sincos(3.141592653589793 * t, &sn, &cn);
    pppow_cs(q1, q2, cn, sn, yy);
  }
  __device__ __forceinline__ void pppow_cs(int q1, int q2, double cn, double sn, bool yy) {   // (cn, sn) = cos / sin of pi t
    const double ar = 0.5 * (1.0 + cn), ai = 0.5 * sn, br = 0.5 * (1.0 - cn), bi = -0.5 * sn;
    const int m1 = mask(q1), m2 = mask(q2);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if (!(x & m1)) {                 // pairs (x, x ^ m1 ^ m2), visited once: the representative has bit q1 = 0
        const int y = x ^ m1 ^ m2;
        const double sg = (yy && !(x & m2)) ? -1.0 : 1.0;      // bits of x equal (0, 0) <-> (1, 1): YY gives -1
        const double pr = re[x], pi = im[x], qr = re[y], qi = im[y];
        re[x] = ar * pr - ai * pi + sg * (br * qr - bi * qi);
        im[x] = ar * pi + ai * pr + sg * (br * qi + bi * qr);
        re[y] = ar * qr - ai * qi + sg * (br * pr - bi * pi);
        im[y] = ar * qi + ai * qr + sg * (br * pi + bi * pr);
      }
  }
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int x = 0; x < N; ++x) { re[x] = (x == 0) ? 1.0 : 0.0; im[x] = 0.0; }
  }
  __device__ __forceinline__ void cnot(int ctrl, int tgt) {
    const int mc = mask(ctrl), mt = mask(tgt);
#pragma unroll
    for (int x = 0; x < N; ++x)
      if ((x & mc) && !(x & mt)) {
        const double pr = re[x], pi = im[x];
        re[x] = re[x | mt]; im[x] = im[x | mt];
        re[x | mt] = pr; im[x | mt] = pi;
      }
  }
};

// One layer of a layered ansatz, the cosines / sines of its angles given: KIND 0 / 3 half angles (rz, rx[, rz]),
// KIND 1 (pi/2 beta, pi gamma).  ansatz_angle_scale<KIND>(position in the layer) is the factor in front of the parameter.
template <int KIND>
__device__ __forceinline__ constexpr double ansatz_angle_scale(int pos) {
  return KIND == 1 ? (pos == 0 ? 1.5707963267948966 : 3.141592653589793) : 0.5;
}
template <int NQ, int KIND>
__device__ __forceinline__ void ansatz_layer_cs(Reg<NQ>& r, const double* c, const double* s) {
  if (KIND == 0 || KIND == 3) {
    r.rz_all_cs(c[0], s[0]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) r.rx_cs(q, c[1], s[1]);
    if (KIND == 3) r.rz_all_cs(c[2], s[2]);
    r.had_fast(0);
#pragma unroll
    for (int q = NQ - 2; q >= 0; --q) r.cnot(q, q + 1);
  } else if (KIND == 1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) r.xpow_cs(q, c[0], s[0]);
#pragma unroll
    for (int q = 0; q + 1 < NQ; ++q) r.zzpow_cs(q, q + 1, c[1], s[1]);
  }
}

// The ansatz circuits on a register file of NQ qubits; `par(l)` returns angle l (HBM, LDS, shifted ... the caller's choice).
template <int NQ, int KIND, class Par>
__device__ __forceinline__ void ansatz_circuit(Reg<NQ>& r, Par par, int n_params) {
  if (KIND == 0 || KIND == 1 || KIND == 3) {
    constexpr int per = (KIND == 3) ? 3 : 2;
    for (int l = 0; l + per <= n_params; l += per) {
      double c[per], s[per];
#pragma unroll
      for (int k = 0; k < per; ++k) sincos(ansatz_angle_scale<KIND>(k) * par(l + k), &s[k], &c[k]);
      ansatz_layer_cs<NQ, KIND>(r, c, s);
    }
  } else if (KIND == 4) {
    // ShallowCNOTStateTensor_nonuniform (represent.py:312-332): per layer 2 NQ angles - rz(p[i]) and rx(p[i + NQ]) on qubit i,
    // CNOT ladder from the bottom (no Hadamard)
    for (int l = 0; l + 2 * NQ <= n_params; l += 2 * NQ) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) r.rz(q, par(l + q));
#pragma unroll
      for (int q = 0; q < NQ; ++q) r.rx(q, par(l + NQ + q));
#pragma unroll
      for (int q = NQ - 2; q >= 0; --q) r.cnot(q, q + 1);
    }
  } else if (KIND == 5) {
    // ExactAfter4 (represent.py:356-380): per layer (a, b, c, d, e, f): rz(a) q0, rz(d) q1, rx(b) q0, rx(e) q1, rz(c) q0, rz(f) q1,
    // CNOT ladder from the bottom, then SWAP(q[i], q[i + 1]) for i = 0 .. NQ - 2 and SWAP(q[NQ - 1], q[0])
    for (int l = 0; l + 6 <= n_params; l += 6) {
      r.rz(0, par(l)); r.rz(1, par(l + 3));
      r.rx(0, par(l + 1)); r.rx(1, par(l + 4));
      r.rz(0, par(l + 2)); r.rz(1, par(l + 5));
#pragma unroll
      for (int q = NQ - 2; q >= 0; --q) r.cnot(q, q + 1);
#pragma unroll
      for (int q = 0; q < NQ; ++q) r.swapq(q, q + 1 < NQ ? q + 1 : 0);
    }
  } else if (KIND == 6) {
    // StateGate (represent.py:406-423), two qubits: rx(a) q0, rx(b) q1, rz(c) q0, rz(d) q1, XX**e, YY**f
    if constexpr (NQ == 2) {
      r.rx(0, par(0)); r.rx(1, par(1));
      r.rz(0, par(2)); r.rz(1, par(3));
      r.pppow(0, 1, par(4), false);
      r.pppow(0, 1, par(5), true);
    }
  } else if (KIND == 2) {
    if constexpr (NQ == 2) {
      r.rz(0, par(0)); r.rx(0, par(1)); r.rz(0, par(2));
      r.rz(1, par(3)); r.rx(1, par(4)); r.rz(1, par(5));
      r.cnot(0, 1);
      r.ry(0, par(6));
      r.cnot(1, 0);
      r.ry(0, par(7)); r.rz(1, par(8));
      r.cnot(0, 1);
      r.rz(0, par(9)); r.rx(0, par(10)); r.rz(0, par(11));
      r.rz(1, par(12)); r.rx(1, par(13)); r.rz(1, par(14));
    }
  }
}

// The factor in front of parameter l inside its gate's sincos, for EVERY ansatz kind with a fixed number of qubits per angle
// (0, 3: half angles; 1: pi/2 beta, pi gamma; 2: half angles; 6: half angles of rx, rx, rz, rz, then pi e, pi f)
template <int KIND>
__device__ __forceinline__ double ansatz_param_scale(int l) {
  if (KIND == 1) return (l & 1) ? 3.141592653589793 : 1.5707963267948966;
  if (KIND == 6) return l < 4 ? 0.5 : 3.141592653589793;
  return 0.5;
}
// The same circuits with cos / sin of the scaled angles supplied by `cs(l)` -> (cos, sin): callers that simulate many circuits
// whose angles mostly coincide (the central-difference columns of one iterate) compute each sincos ONCE and share it
// (kinds 0, 1, 2, 3, 6; a double-precision sincos costs more than a whole two-qubit layer).
template <int NQ, int KIND, class CS>
__device__ __forceinline__ void ansatz_circuit_cs(Reg<NQ>& r, CS cs, int n_params) {
  if (KIND == 0 || KIND == 1 || KIND == 3) {
    constexpr int per = (KIND == 3) ? 3 : 2;
    for (int l = 0; l + per <= n_params; l += per) {
      double c[per], s[per];
#pragma unroll
      for (int k = 0; k < per; ++k) {
        const double2 t = cs(l + k);
        c[k] = t.x;
        s[k] = t.y;
      }
      ansatz_layer_cs<NQ, KIND>(r, c, s);
    }
  } else if (KIND == 2) {
    if constexpr (NQ == 2) {
      double c[15], s[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) {
        const double2 t = cs(k);
        c[k] = t.x;
        s[k] = t.y;
      }
      r.shallow_full_cs(0, 1, c, s);
    }
  } else if (KIND == 6) {
    if constexpr (NQ == 2) {
      double2 t = cs(0); r.rx_cs(0, t.x, t.y);
      t = cs(1); r.rx_cs(1, t.x, t.y);
      t = cs(2); r.rz_cs(0, t.x, t.y);
      t = cs(3); r.rz_cs(1, t.x, t.y);
      t = cs(4); r.pppow_cs(0, 1, t.x, t.y, false);
      t = cs(5); r.pppow_cs(0, 1, t.x, t.y, true);
    }
  }
}

// nsh = 3: shifts {0, +pi/2, -pi/2} (qmps/rotosolve.py:175);  nsh = 6: {0, pi, +-pi/2, +-pi/4} (qmps/tools.py:434-438)
__device__ __forceinline__ double roto_shift_value(int nsh, int k) {
  if (nsh == 3) return k == 0 ? 0.0 : (k == 1 ? 1.5707963267948966 : -1.5707963267948966);
  switch (k) {
    case 0: return 0.0;
    case 1: return 3.141592653589793;
    case 2: return 1.5707963267948966;
    case 3: return -1.5707963267948966;
    case 4: return 0.7853981633974483;
    default: return -0.7853981633974483;
  }
}

}  // namespace qmps
