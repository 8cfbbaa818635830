// qmps_device.h - small device-side helpers shared by the kernel translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qmps {

typedef double v4f64 __attribute__((ext_vector_type(4)));     // accumulator / B operand of v_mfma_f64_16x16x4

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- wave-level reductions (VALU only: DPP row rotations + gfx950 permlane swaps) ----
// sum over the 16 lanes of a row group (lanes 16 g .. 16 g + 15); result in every lane of the group.
// DPP row rotations (v_mov_b32_dpp row_ror:n, VALU only - no LDS crossbar traffic).
template <int N>
__device__ __forceinline__ double row_ror(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + N, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += row_ror<8>(v);
  v += row_ror<4>(v);
  v += row_ror<2>(v);
  v += row_ror<1>(v);
  return v;
}
// sum over the four row groups, per lane position: lane (g, c) receives sum_g' v(g', c).
// gfx950 v_permlane32_swap / v_permlane16_swap: VALU only, no LDS crossbar, no SGPR round trip.
__device__ __forceinline__ double group4_sum(double v) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  u2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  u2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(x);
  hi = __double2hiint(x);
  a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// the same sum on the matrix pipe: one v_mfma_f64_4x4x4_4b with an all-ones A operand.  Lane 16 x + 4 q + z
// holds B_q[k = x][j = z]; D_q[i][j] = sum_k B_q[k][j] lands in lane (x = i, q, z = j): every row group
// receives the column sums (16 cycles instead of ~10 VALU instructions; summation order differs).
__device__ __forceinline__ double group4_sum_mfma(double v) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
}
// wave-uniform copy of lane 0's value
__device__ __forceinline__ double lane0(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) { return group4_sum(row16_sum(v)); }


// ---- DPP quad helpers (a quad = 4 consecutive lanes; v_mov_b32_dpp quad_perm, VALU only) ----
// value of lane L of the own quad, in every lane of the quad
template <int L>
__device__ __forceinline__ double quad_bcast(double v) {
  constexpr int ctrl = L * 0x55;   // quad_perm:[L,L,L,L]
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the four lanes of the quad, result in every lane
__device__ __forceinline__ double quad_sum(double v) {
  v += quad_perm<0xB1>(v);   // [1,0,3,2]
  v += quad_perm<0x4E>(v);   // [2,3,0,1]
  return v;
}
// v_rcp_f64 + one Newton step: relative error ~1e-16
__device__ __forceinline__ double fast_rcp(double t) {
  const double x = __builtin_amdgcn_rcp(t);
  return dfma(dfma(-t, x, 1.0), x, x);
}

// v_rsq_f64 + one Newton step: relative error ~1e-16 (x > 0, finite)
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  return dfma(0.5 * y, dfma(-x * y, y, 1.0), y);
}

}  // namespace qmps
