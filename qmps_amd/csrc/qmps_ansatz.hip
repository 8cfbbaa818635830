// qmps_ansatz.hip - parameters -> state tensor on the device (qmps/represent.py:268-423), unitary_to_tensor (qmps/tools.py:151-154)
// and the update / record kernels of the step-by-step rotosolve drivers (qmps/rotosolve.py:154-181, qmps/tools.py:422-457).
// Split out of qmps_kernels.hip in round 3.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "qmps_kernels.h"
#include "qmps_knobs.h"
#include "qmps_device.h"
#include "qmps_roto_math.h"
#include "qmps_circuit.h"
#include "qmps_circuit_wave.h"

namespace qmps {

// ------------------------------------------------------------------------------------------
// Kernel 3: unitary_to_tensor (qmps/tools.py:151-154):  A[b][s][i][j] = U[b][2 i + s][j], j < D
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unitary_to_tensor_kernel(const double2* __restrict__ U,
                                                                double2* __restrict__ A, int D, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int n = D * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / (2 * n);
    const int e = (int)(t % (2 * n));
    const int s = e / n, i = (e / D) % D, j = e % D;
    A[t] = U[b * (4 * n) + (int64_t)(2 * i + s) * (2 * D) + j];
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 3b: ansatz parameters -> state tensor on the device (qmps/represent.py:268-404).
// Thread (b, j) simulates the ansatz circuit on the basis state |0>|j> of the n + 1 = log2(2 D) qubit
// register (big-endian: qubit 0 is the most significant bit) - i.e. column j < D of the unitary, which
// is all unitary_to_tensor keeps (qmps/tools.py:151-154): A[s][i][j] = U[2 i + s][j].  The 2 D
// amplitudes live in registers; CNOTs are register renames.  HBM input drops from 64 D^2 bytes (U) to
// 8 P bytes (parameters) per evaluation.
//   kind 0: ShallowCNOTStateTensor   per (beta, gamma): rz(beta) all, rx(gamma) all, H(q0), CNOT ladder
//   kind 1: ShallowQAOAStateTensor   per (beta, gamma): X**beta all, ZZ**gamma neighbours
//   kind 2: ShallowFullStateTensor   15 angles, two qubits (D = 2)
//   kind 3: ShallowCNOTStateTensor3  per (beta, gamma, omega): rz, rx, rz all, H(q0), CNOT ladder
//   kind 4: ShallowCNOTStateTensor_nonuniform  per layer 2 (n + 1) angles: rz(p[i]), rx(p[i + n + 1]) on qubit i, CNOT ladder
//   kind 5: ExactAfter4              per layer 6 angles on qubits 0, 1, CNOT ladder, cyclic SWAPs
//   kind 6: StateGate                6 angles, two qubits (D = 2): rx, rx, rz, rz, XX**e, YY**f
// ------------------------------------------------------------------------------------------
// (Reg<NQ>, ansatz_circuit, roto_shift_value: qmps_circuit.h)
template <int D, int KIND>
// nsh > 0: rotosolve shift batches without a separate shift-build kernel - evaluation b = nsh r + k is restart r (parameter
// row r) with shift k added to parameter *i_ptr
// fd_h != 0: central-difference batches - nsh = 2 P evaluations per row, evaluation nsh r + k = row r with +fd_h (k < P) or
// -fd_h (k >= P) added to parameter k mod P
__global__ __launch_bounds__(64) void ansatz_tensor_kernel(const double* __restrict__ params, int n_params,
                                                           double2* __restrict__ A, int64_t B, int nsh,
                                                           const int* __restrict__ i_ptr, double fd_h, const unsigned char* __restrict__ active) {
  constexpr int NQ = (D == 2 ? 2 : D == 4 ? 3 : D == 8 ? 4 : 5);
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t b = t / D;
  const int j = (int)(t % D);
  if (b >= B) return;
  const int64_t row = nsh > 0 ? b / nsh : b;
  if (active != nullptr && active[row] == 0) return;       // (central-difference batches of the evolve drivers: a skipped trajectory's neighbours are never read)
  const int shift_k = nsh > 0 ? (int)(b - row * nsh) : 0;
  const bool fd = fd_h != 0.0;
  const int isel = nsh > 0 ? (fd ? shift_k % n_params : *i_ptr) : -1;
  const double fd_shift = shift_k < n_params ? fd_h : -fd_h;
  const double* pp = params + row * n_params;
  Reg<NQ> r;
#pragma unroll
  for (int x = 0; x < Reg<NQ>::N; ++x) {
    r.re[x] = (x == j) ? 1.0 : 0.0;
    r.im[x] = 0.0;
  }
  ansatz_circuit<NQ, KIND>(r, [&](int l) {
    double v = pp[l];
    if (l == isel) v += fd ? fd_shift : roto_shift_value(nsh, shift_k);
    return v;
  }, n_params);
  // A[b][s][i][j] = amplitude[2 i + s]
  double2* out = A + b * (2 * D * D);
#pragma unroll
  for (int x = 0; x < Reg<NQ>::N; ++x) out[((x & 1) * D + (x >> 1)) * D + j] = make_double2(r.re[x], r.im[x]);
}

// ------------------------------------------------------------------------------------------
// Kernel 3b': the same at D = 16 with the circuit DISTRIBUTED over the lanes (round 4).  One lane per column walks through ~6 500
// dependent instructions (29 us per launch, whatever the batch: it sits in front of every gradient batch of the config-4 time
// evolution); here a wave owns TWO columns, lane 32 jj + a holds amplitude a (five bits, qubit 0 = the most significant) of column
// 2 w + jj: rz on every qubit is one phase per lane (powers of c - i s by popcount), rx on qubit q a butterfly with lane a xor bit
// (ds_swizzle, groups of 32), the Hadamard another, the CNOT ladder ONE gather (ds_bpermute) - ~60 instructions per layer.
// kinds 0 (ShallowCNOT) and 3 (ShallowCNOT3); shifts / central differences as ansatz_tensor_kernel.
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void ansatz_tensor_wave_d16_kernel(const double* __restrict__ params, int n_params, double2* __restrict__ A, int64_t B,
                                                                     int nsh, const int* __restrict__ i_ptr, double fd_h, const unsigned char* __restrict__ active) {
  const int lane = threadIdx.x & 63, a = lane & 31;
  const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // wave = (tensor, pair of columns)
  const int64_t b = wv >> 3;
  if (b >= B) return;
  const int j = (int)(wv & 7) * 2 + (lane >> 5);
  const int64_t row = nsh > 0 ? b / nsh : b;
  if (active != nullptr && active[row] == 0) return;
  const int shift_k = nsh > 0 ? (int)(b - row * nsh) : 0;
  const bool fd = fd_h != 0.0;
  const int isel = nsh > 0 ? (fd ? shift_k % n_params : *i_ptr) : -1;
  const double fd_shift = shift_k < n_params ? fd_h : -fd_h;
  // one sincos per angle and wave: lane l takes angle l (n_params <= 64), broadcast by readlane
  double cn = 1.0, sn = 0.0;
  if (lane < n_params) {
    double v = params[row * n_params + lane];
    if (lane == isel) v += fd ? fd_shift : roto_shift_value(nsh, shift_k);
    sincos(0.5 * v, &sn, &cn);
  }
  double re, im;
  shallow_cnot_wave_column_d16<KIND>(cn, sn, n_params, j, re, im);
  // A[b][s][i][j] = amplitude[2 i + s]
  A[b * 512 + ((a & 1) * 16 + (a >> 1)) * 16 + j] = make_double2(re, im);
}

template <int D>
static hipError_t launch_ansatz_d(int kind, const double* params, int n_params, void* A, int64_t B, int nsh, const int* i_ptr, hipStream_t st, double fd_h = 0.0,
                                  const unsigned char* active = nullptr) {
  const int64_t threads = B * D;
  const dim3 grid((unsigned)((threads + 63) / 64)), block(64);
  if constexpr (D == 16) {
    // ShallowCNOT families: the circuit distributed over the lanes of a wave (two columns per wave)
    // (the distributed form is ~10 x shorter in latency but costs ~2 x the issue slots - eight waves per tensor: plain batches - the
    // iterates and references of the evolve drivers, one tensor per trajectory, whose build sits in front of the solves - always,
    // whatever their size, so that a trajectory's tensor does not depend on how many others share its launch (the two kernels round
    // differently); shifted batches (central-difference neighbours, rotosolve: 2 P or 3 tensors per row) only when small)
    if ((kind == 0 || kind == 3) && n_params <= 64 && (nsh == 0 || B <= 512)) {
      const dim3 g2((unsigned)(B * 2)), b2(256);
      if (kind == 0) hipLaunchKernelGGL((ansatz_tensor_wave_d16_kernel<0>), g2, b2, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active);
      else hipLaunchKernelGGL((ansatz_tensor_wave_d16_kernel<3>), g2, b2, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active);
      return hipGetLastError();
    }
  }
  switch (kind) {
    case 0: hipLaunchKernelGGL((ansatz_tensor_kernel<D, 0>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active); break;
    case 1: hipLaunchKernelGGL((ansatz_tensor_kernel<D, 1>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active); break;
    case 2:
      if (D != 2) return hipErrorInvalidValue;
      hipLaunchKernelGGL((ansatz_tensor_kernel<2, 2>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active);
      break;
    case 3: hipLaunchKernelGGL((ansatz_tensor_kernel<D, 3>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active); break;
    case 4: hipLaunchKernelGGL((ansatz_tensor_kernel<D, 4>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active); break;
    case 5: hipLaunchKernelGGL((ansatz_tensor_kernel<D, 5>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active); break;
    case 6:
      if (D != 2) return hipErrorInvalidValue;
      hipLaunchKernelGGL((ansatz_tensor_kernel<2, 6>), grid, block, 0, st, params, n_params, (double2*)A, B, nsh, i_ptr, fd_h, active);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_ansatz_shifted(int D, int kind, const double* params, int n_params, void* A, int64_t B, int nsh, const int* i_ptr,
                                 hipStream_t st) {
  if (B <= 0) return hipSuccess;
  switch (D) {
    case 2: return launch_ansatz_d<2>(kind, params, n_params, A, B, nsh, i_ptr, st);
    case 4: return launch_ansatz_d<4>(kind, params, n_params, A, B, nsh, i_ptr, st);
    case 8: return launch_ansatz_d<8>(kind, params, n_params, A, B, nsh, i_ptr, st);
    case 16: return launch_ansatz_d<16>(kind, params, n_params, A, B, nsh, i_ptr, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ansatz_fd(int D, int kind, const double* params, int n_params, void* A, int64_t rows, double h, hipStream_t st, const unsigned char* active) {
  const int64_t B = rows * 2 * n_params;
  if (B <= 0) return hipSuccess;
  switch (D) {
    case 2: return launch_ansatz_d<2>(kind, params, n_params, A, B, 2 * n_params, nullptr, st, h, active);
    case 4: return launch_ansatz_d<4>(kind, params, n_params, A, B, 2 * n_params, nullptr, st, h, active);
    case 8: return launch_ansatz_d<8>(kind, params, n_params, A, B, 2 * n_params, nullptr, st, h, active);
    case 16: return launch_ansatz_d<16>(kind, params, n_params, A, B, 2 * n_params, nullptr, st, h, active);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ansatz_masked(int D, int kind, const double* params, int n_params, void* A, int64_t B, const unsigned char* active, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  switch (D) {
    case 2: return launch_ansatz_d<2>(kind, params, n_params, A, B, 0, nullptr, st, 0.0, active);
    case 4: return launch_ansatz_d<4>(kind, params, n_params, A, B, 0, nullptr, st, 0.0, active);
    case 8: return launch_ansatz_d<8>(kind, params, n_params, A, B, 0, nullptr, st, 0.0, active);
    case 16: return launch_ansatz_d<16>(kind, params, n_params, A, B, 0, nullptr, st, 0.0, active);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ansatz(int D, int kind, const double* params, int n_params, void* A, int64_t B, hipStream_t st) {
  return launch_ansatz_shifted(D, kind, params, n_params, A, B, 0, nullptr, st);
}

// ------------------------------------------------------------------------------------------
// Kernel 3c: device-resident rotosolve (qmps/rotosolve.py:154-181).  For parameter i,
// R restarts x 3 shifts {0, +pi/2, -pi/2} form one batch; the closed-form update
//   theta* = -pi/2 - atan2(2 e0 - e+ - e-, e+ - e-),  params[i] = wrap(params[i] + wrap(theta*))
// runs on the device, so a whole sweep needs no host round trip.
// ------------------------------------------------------------------------------------------

// (wrap_pi, double_sinusoid_argmin: qmps_roto_math.h)
__global__ __launch_bounds__(256) void roto_update_kernel(double* __restrict__ base, const double* __restrict__ E,
                                                          const int32_t* __restrict__ status, int R, int P,
                                                          int* __restrict__ i_ptr, int n_terms, int nsh, int rule) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = *i_ptr;
  // the LAST workgroup to finish advances the parameter index for the next graph replay
  __shared__ int s_last;
  if (r < R) {
    double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    bool ok = true;
    for (int k = 0; k < nsh; ++k) {
      double v = 0.0;
      for (int q = 0; q < n_terms; ++q) v += E[((int64_t)r * nsh + k) * n_terms + q];   // M(x) = sum over terms
      e[k] = v;
      ok = ok && overlap_usable(status[(int64_t)r * nsh + k]);      // (energy path: status 0; overlap path at D = 2: also a tie's common modulus)
    }
    if (ok) {          // (an evaluation without a valid environment leaves this restart's parameter untouched)
      double theta;
      if (nsh == 3) {
        theta = -1.5707963267948966 - atan2(2.0 * e[0] - e[1] - e[2], e[1] - e[2]);
      } else {
        // samples at {0, pi, +pi/2, -pi/2, +pi/4, -pi/4}: a, b, c, d -> P sin(2x + u) + Q sin(x + v)  (tools.py:434-447)
        const double A = e[0] + e[1], Bv = e[0] - e[1], C = e[2] + e[3], Dv = e[2] - e[3], Ev = e[4] - e[5];
        const double a = 0.25 * (2.0 * Ev - 1.4142135623730951 * Dv), b = 0.25 * (A - C), c = 0.5 * Dv, d = 0.5 * Bv;
        theta = double_sinusoid_step(a, b, c, d, rule);  // P sin(2x + u) = a sin 2x + b cos 2x,  Q sin(x + v) = c sin x + d cos x; in [-pi, pi]
      }
      const double moved = base[(int64_t)r * P + i] + (nsh == 3 ? wrap_pi(theta) : theta);
      base[(int64_t)r * P + i] = nsh == 3 ? wrap_pi(moved) : moved;   // the double-frequency driver does not re-wrap (tools.py:453-454)
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = atomicAdd(i_ptr + 1, 1);        // i_ptr[1]: arrival counter
    s_last = (done == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    i_ptr[1] = 0;
    i_ptr[0] = (i + 1 == P) ? 0 : i + 1;             // every block has read *i_ptr before its arrival
    if (i + 1 == P) i_ptr[2] += 1;                   // i_ptr[2]: sweeps finished (roto_record_kernel)
    __threadfence();
  }
}

// sweep_ptr (nullable): device counter of finished sweeps (advanced by the update kernel that wraps the parameter index):
// the record of sweep n lands in hist[(n - 1) R ...], so ONE captured graph serves every sweep
// stride: evaluations per restart in E (1: the R base vectors; nsh: a shifted batch, whose shift-0 row IS the evaluation of the
// base vector - the record of a sweep is taken from the first shifted batch of the NEXT one, so a sweep costs n_params
// batches, not n_params + 1); nothing is written before the first sweep has finished
__global__ __launch_bounds__(256) void roto_record_kernel(const double* __restrict__ E, double* __restrict__ hist, int R,
                                                          int n_terms, const int* __restrict__ sweep_ptr, int stride) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int64_t sw = sweep_ptr != nullptr ? (int64_t)(*sweep_ptr - 1) : 0;
  if (sw < 0) return;
  double v = 0.0;
  for (int q = 0; q < n_terms; ++q) v += E[(int64_t)r * stride * n_terms + q];
  hist[sw * R + r] = v;
}

hipError_t launch_roto_update(double* base, const double* E, const int32_t* status, int R, int P, int* i_ptr, int n_terms,
                              int nsh, int rule, hipStream_t st) {
  hipLaunchKernelGGL(roto_update_kernel, dim3((R + 255) / 256), dim3(256), 0, st, base, E, status, R, P, i_ptr, n_terms, nsh, rule);
  return hipGetLastError();
}
// the update rule alone on caller-given fit coefficients (qmps_roto_rule_probe: parity of the device build of qmps_roto_rule.h with
// the answers scipy gave the reference, tests/test_refshim_gpu.py)
__global__ __launch_bounds__(64) void roto_rule_probe_kernel(const double* __restrict__ abcd, int64_t n, int rule, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = double_sinusoid_step(abcd[4 * i], abcd[4 * i + 1], abcd[4 * i + 2], abcd[4 * i + 3], rule);
}
hipError_t launch_roto_rule_probe(const double* abcd, int64_t n, int rule, double* out, hipStream_t st) {
  hipLaunchKernelGGL(roto_rule_probe_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, abcd, n, rule, out);
  return hipGetLastError();
}
hipError_t launch_roto_record(const double* E, double* hist, int R, int n_terms, const int* sweep_ptr, int stride, hipStream_t st) {
  hipLaunchKernelGGL(roto_record_kernel, dim3((R + 255) / 256), dim3(256), 0, st, E, hist, R, n_terms, sweep_ptr, stride);
  return hipGetLastError();
}

hipError_t launch_unitary_to_tensor(const void* U, void* A, int D, int64_t B, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int64_t total = B * 2 * D * D;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(unitary_to_tensor_kernel, dim3(grid), dim3(256), 0, st, (const double2*)U, (double2*)A, D, total);
  return hipGetLastError();
}

}  // namespace qmps
