// qmps_circuit_wave.h - the ShallowCNOT circuits of D = 16 (five qubits) DISTRIBUTED over the lanes of a wave (gfx950 only):
// lane 32 jj + a holds amplitude a (five bits, qubit 0 = the most significant) of one of two columns.  rz on every qubit is one phase
// per lane (powers of c - i s by popcount), rx on qubit q a butterfly with lane a xor bit (ds_swizzle, groups of 32), the Hadamard
// another, the CNOT ladder ONE gather (ds_bpermute) - ~60 instructions per layer (qmps/represent.py:288-310, 334-354).
// Shared by ansatz_tensor_wave_d16_kernel (qmps_ansatz.hip) and the neighbour-building probe kernel (qmps_overlap_grad.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "qmps_device.h"

namespace qmps {

template <int PATTERN>
__device__ __forceinline__ double swz32(double v) {      // value of the lane (own index xor mask), within groups of 32 lanes
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN);
  return __hiloint2double(hi, lo);
}
template <int PATTERN>
__device__ __forceinline__ void rx_lanes(double& re, double& im, double c, double s) {      // a' = c a - i s (partner's a)
  const double pr = swz32<PATTERN>(re), pi = swz32<PATTERN>(im);
  const double nr = dfma(c, re, s * pi), ni = dfma(c, im, -s * pr);
  re = nr;
  im = ni;
}

// Column j (this lane's: j depends on the lane's half of the wave) of the five-qubit circuit: amplitude a = lane & 31 comes back in
// (re, im).  cn / sn: cos / sin of HALF the angle l held by lane l of the wave (all 64 lanes call this together).
// KIND 0: ShallowCNOTStateTensor (2 angles per layer), 3: ShallowCNOTStateTensor3 (3 angles per layer).
template <int KIND>
__device__ __forceinline__ void shallow_cnot_wave_column_d16(double cn, double sn, int n_params, int j, double& re, double& im) {
  constexpr int per = KIND == 3 ? 3 : 2;
  const int lane = threadIdx.x & 63, a = lane & 31;
  // |0>|j>: basis state x = j (qubit 0 = 0)
  re = a == j ? 1.0 : 0.0;
  im = 0.0;
  const int pop = __builtin_popcount((unsigned)a);
  auto rz_all = [&](double c, double s) {
    // prod_q rz(theta) = diag(z^(5 - 2 popcount)), z = c - i s (phi = theta / 2): z^1, z^3, z^5 and their conjugates
    const double z2r = dfma(c, c, -s * s), z2i = -2.0 * s * c;
    const double z3r = dfma(z2r, c, z2i * s), z3i = dfma(z2i, c, -z2r * s);
    const double z5r = dfma(z3r, z2r, -z3i * z2i), z5i = dfma(z3r, z2i, z3i * z2r);
    const int m = 5 - 2 * pop;
    const double pr = (m == 1 || m == -1) ? c : ((m == 3 || m == -3) ? z3r : z5r);
    const double pa = (m == 1 || m == -1) ? -s : ((m == 3 || m == -3) ? z3i : z5i);
    const double pi = m > 0 ? pa : -pa;
    const double nr = dfma(re, pr, -im * pi), ni = dfma(re, pi, im * pr);
    re = nr;
    im = ni;
  };
  for (int l0 = 0; l0 + per <= n_params; l0 += per) {
    rz_all(__shfl(cn, l0, 64), __shfl(sn, l0, 64));
    {
      const double c = __shfl(cn, l0 + 1, 64), s = __shfl(sn, l0 + 1, 64);
      rx_lanes<0x041F>(re, im, c, s);      // qubit 4 <-> lane bit 0
      rx_lanes<0x081F>(re, im, c, s);      // qubit 3
      rx_lanes<0x101F>(re, im, c, s);      // qubit 2
      rx_lanes<0x201F>(re, im, c, s);      // qubit 1
      rx_lanes<0x401F>(re, im, c, s);      // qubit 0 <-> lane bit 4
    }
    if (KIND == 3) rz_all(__shfl(cn, l0 + 2, 64), __shfl(sn, l0 + 2, 64));
    {
      // H on qubit 0 (lane bit 4): (own + partner)/sqrt 2 on the 0 side, (partner - own)/sqrt 2 on the 1 side
      const double h = 0.70710678118654752, sg = (a & 16) ? -h : h;
      const double pr = swz32<0x401F>(re), pi = swz32<0x401F>(im);
      re = dfma(sg, re, h * pr);
      im = dfma(sg, im, h * pi);
    }
    {
      // CNOT(q3, q4), CNOT(q2, q3), CNOT(q1, q2), CNOT(q0, q1): amplitude (b0 .. b4) moves to (b0, b1^b0, b2^b1, b3^b2, b4^b3);
      // destination lane d gathers from the source whose image it is (prefix xor of its bits)
      const int d0 = (a >> 4) & 1, b1 = ((a >> 3) & 1) ^ d0, b2 = ((a >> 2) & 1) ^ b1, b3 = ((a >> 1) & 1) ^ b2, b4 = (a & 1) ^ b3;
      const int src = (lane & 32) | (d0 << 4) | (b1 << 3) | (b2 << 2) | (b3 << 1) | b4;
      re = __shfl(re, src, 64);
      im = __shfl(im, src, 64);
    }
  }
}

}  // namespace qmps
