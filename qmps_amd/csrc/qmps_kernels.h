// qmps_kernels.h - internal interface between the C-ABI host code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qmps {

enum { QMPS_ST_OK = 0, QMPS_ST_NOT_CONVERGED = 1, QMPS_ST_NOT_PD = 2,
       QMPS_ST_PENDING = 3 /* internal, between the kernels of one launch: handed to the Krylov fall-back (D = 16 environment) */,
       QMPS_ST_TIED = 4 /* overlap path, D = 2, 4: dominant eigenvalues tied in modulus - eta is their common modulus, the vector is NOT a fixed point */ };
// an overlap evaluation whose eta (hence the objective -sqrt|eta|) may be used: converged, or the common modulus of a tie (QMPS_STATUS_TIED)
__host__ __device__ inline bool overlap_usable(int status) { return status == QMPS_ST_OK || status == QMPS_ST_TIED; }

// Kernel arguments of the energy kernels (zero-initialise, then fill) (all pointers are HBM addresses).
struct LaneArgs {
  const void* A;      // [B][2][D][D] complex128
  const void* h;      // [n_terms][4][4] complex128
  const void* r_in;   // nullable [B][D][D] complex128: warm start (SOLVE) or the environment (!SOLVE)
  void* r_out;        // nullable [B][D][D] complex128
  void* rho_out;      // nullable [B][4][4] complex128
  double* E;          // [B][n_terms]
  int32_t* iters;     // [B]
  int32_t* status;    // [B]
  int64_t B;
  int n_terms;
  int max_iter;
  double tol;
  // hybrid solve: `handoff` plain power steps, then the repeated-squaring tail (0 = plain only)
  int handoff;
  int hybrid;                 // 1: squaring tail enabled (handoff may be 0 = squaring from the start)
  int skip;                   // D = 2 in-lane tail: squarings before the iterate is tracked
  int32_t* work_count;        // D = 4: slow items are appended to work_idx (wave-aggregated atomics)
  int32_t* work_idx;
  // list mode (energy-only pass over the worklist): evaluation ids come from idx_list[0 .. *idx_count)
  const int32_t* idx_list;
  const int32_t* idx_count;
  int check_pd;               // !SOLVE: Cholesky test of the resident r, status 0 -> 2 on failure
  double* partial;            // nullable [n_terms][gridDim.x]: per-wave sums of the energies (lane kernels)
  // exact in-kernel cost accumulation (energy_direct_d4_kernel): every wave adds ONE 64-bit word per term to
  // acc[term * kAccMaxShards + (tile % acc_shards)] - its partial sum as a fixed-point integer (scale acc_scale, offset
  // 2^51 so that the addend is positive) in bits 0..57 and an arrival count of 1 in bits 58..63.  Integer addition
  // commutes: the sum is exact and does not depend on the order the waves finish in; the count tells a reader on another
  // stream when every wave has arrived (no event on the compute stream).  A partial beyond acc_bound (tensor not an
  // isometry, NaN) goes to the double acc[kAccOver + term] first and counts with value 0.
  // acc_zero: accumulator of a LATER step, cleared here.
  long long* acc;
  long long* acc_zero;
  double acc_scale;           // 2^k with acc_bound * 2^k <= 2^51
  double acc_bound;
  int acc_shards;             // power of two, waves per shard <= 60
  // energy_direct_d4_kernel with the ansatz fused in front: the state tensor is built in LDS from
  // ans_P parameters per evaluation (ans_kind = QMPS_ANSATZ_*) and never travels through HBM.  ans_nsh > 0: rotosolve
  // shift batches - evaluation b = ans_nsh r + k evaluates restart r (parameter row r) with shift k added to parameter *ans_i
  const double* ans_params;   // nullable [rows][ans_P]
  const int* ans_i;           // device pointer to the index of the parameter being updated (ans_nsh > 0)
  int ans_P, ans_kind, ans_nsh;
  int direct;                 // D = 2 lane kernel: QMPS_ENV_DIRECT (4 x 4 fixed-point solve in the lane, then the squaring tail)
  // D = 16 environment: Krylov fall-back (qmps_overlap_krylov.hip, env_mode).  The power iteration hands an evaluation over
  // (status PENDING, iterate in r_out, no energy written) once its residual history predicts more than krylov_after further steps;
  // the fall-back replaces r_out by the fixed point; a FINISHING pass of the same kernel (only_pending = 1, r_in = that buffer)
  // accepts it by its own test and writes energy, status, iterations.  kry_counter: [0..2] as OverlapArgs, [3] evaluations pending
  // (counted by the fall-back), [4] exit tickets of the finishing pass - all zero between launches.
  int krylov_after;
  int* kry_counter;
  int only_pending;
};

// D = 8 direct fixed-point solve, one wave per evaluation: writes the environments r[B][8][8] (the warm start / result
// that energy_block_kernel<8, true> then accepts with one power step)
hipError_t launch_env_direct_d8(const void* A, void* r_out, int64_t B, hipStream_t st);
// layout of one cost accumulator (long long units): 16 terms x kAccMaxShards words, then 16 doubles of overflow sums
constexpr int kAccMaxShards = 2048, kAccOver = 16 * kAccMaxShards, kAccWords = kAccOver + 16;
constexpr int kAccValueBits = 58, kAccOffsetBits = 51, kAccMaxWavesPerShard = 60;
// decode one word: (count, value) - value = sum of the fixed-point partials of `count` waves
__host__ __device__ inline void acc_decode(long long w, long long& count, long long& value) {
  count = (long long)((unsigned long long)w >> kAccValueBits);
  value = (w & ((1LL << kAccValueBits) - 1)) - count * (1LL << kAccOffsetBits);
}
#if defined(__HIPCC__)
// one arrival: `s` = the partial sum of this wave / evaluation for term t (see LaneArgs::acc)
__device__ __forceinline__ void acc_arrive(long long* acc, int shards, int t, unsigned id, double s, double bound, double scale) {
  long long fx = 0;
  if (fabs(s) <= bound) {
    fx = __double2ll_rn(s * scale);
  } else {
    atomicAdd((double*)(acc + kAccOver) + t, s);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the overflow sum is in before this arrival counts
  }
  atomicAdd((unsigned long long*)acc + t * kAccMaxShards + (id & (unsigned)(shards - 1)),
            (unsigned long long)(fx + (1LL << kAccOffsetBits) + (1LL << kAccValueBits)));
}
// clear a whole accumulator (called by the threads of ONE workgroup)
__device__ __forceinline__ void acc_clear(long long* acc, int n_terms, int tid, int nthreads) {
  for (int t = 0; t < n_terms; ++t)
    for (int i = tid; i < kAccMaxShards; i += nthreads) acc[t * kAccMaxShards + i] = 0;
  if (tid < 16) acc[kAccOver + tid] = 0;
}
#endif
// acc -> cost[t] = (sum of the shards) / scale + overflow sum; one wave.  expect > 0: first POLL until `expect` waves
// per term have arrived (the producer kernel may still be running on another stream), at most max_polls sweeps - then
// cost = NaN and *err = 1.
hipError_t launch_cost_finish(const long long* acc, int n_shards, long long expect, int max_polls, double inv_scale,
                              int n_terms, double* cost, int* err, hipStream_t st);
// D = 4 repeated-squaring tail over the worklist (one wave per item, MFMA f64 16x16x4)
struct SquareArgs {
  const void* A;
  const void* r_in;           // nullable: r after `done` plain steps / a warm start (null = 1/D)
  void* r_out;                // converged environment
  int32_t* iters;
  int32_t* status;
  const int32_t* work_count;  // item ids = work_idx[0 .. *work_count), or 0 .. B-1 when work_idx == nullptr
  const int32_t* work_idx;
  int64_t B;
  int done;                   // plain steps already taken
  int skip;                   // squarings before the iterate is tracked
  int period;                 // mat-vecs with R_m between two further squarings (<= 0: never square again)
  int max_iter;
  double tol;
};

// D = 16 on the matrix cores: power iteration + Cholesky test + energy epilogue, one wave per evaluation
hipError_t launch_energy_mfma(int D, const LaneArgs& a, bool solve, hipStream_t st);
hipError_t launch_square_tail(int D, const SquareArgs& a, int grid, hipStream_t st);
// D = 4 energy-only pass with two lanes per evaluation (a.r_in resident, no worklist); partials: one per 32 items
hipError_t launch_energy_pair_d4(const LaneArgs& a, hipStream_t st);
// D = 4 energy-only pass in the quad layout of the fused kernel (no worklist mode): qmps_direct.hip
hipError_t launch_energy_only_d4(const LaneArgs& a, hipStream_t st);
// D = 4 plain power iteration, a quad per evaluation, persistent waves drawing evaluations from `counter` (zero at launch); writes r_out, iters, status
hipError_t launch_env_power_d4(const LaneArgs& a, int* counter, int waves, hipStream_t st);
// D = 4 DIRECT solve fused with the energies: one DPP quad per evaluation (qmps_direct.hip); partials: one per 16 items
hipError_t launch_energy_direct_d4(const LaneArgs& a, hipStream_t st);

// Whole D = 2 rotosolve run in one launch (restarts are independent): base [R][P] in/out, hist [n_sweeps][R] out
struct RotoArgs {
  double* base;
  const void* h;
  double* hist;
  int R, P, n_terms, n_sweeps, max_iter, skip;
  double tol;
  int direct;     // QMPS_ENV_DIRECT: try the 4 x 4 fixed-point solve before squaring
  int nsh;        // 3: single-frequency rotosolve, 6: double-frequency
  int rule;       // double-frequency update: QMPS_ROTO_REFERENCE (scipy's bounded Brent search, tools.py:451) | QMPS_ROTO_GLOBAL_ARGMIN
};
hipError_t launch_rotosolve_fused_d2(int kind, const RotoArgs& a, hipStream_t st);
// the same at D = 8 (ShallowCNOT, ShallowCNOT3): one workgroup per restart, one wave per shift (qmps_roto_d8.hip)
hipError_t launch_rotosolve_fused_d8(int kind, const RotoArgs& a, hipStream_t st);

// Two-site unit cell (NonSparseFullTwoSiteEnergyOptimizer): state unitaries U1, U2 [B][2D][2D].
struct Cell2Args {
  const void* U1;
  const void* U2;
  const void* h;
  double* E;        // [B][n_terms]  (E1 + E2)/2
  double* E12;      // nullable [B][n_terms][2]
  int32_t* iters;
  int32_t* status;
  int64_t B;
  int n_terms;
  int max_iter;
  double tol;
};

hipError_t launch_cell2(int D, const Cell2Args& a, hipStream_t st);
hipError_t launch_energy(int D, const LaneArgs& a, bool solve, hipStream_t st);
// D = 8, 16: one evaluation per workgroup of D x D threads (qmps_energy_block.hip); launch_energy dispatches here
hipError_t launch_energy_block(int D, const LaneArgs& a, bool solve, hipStream_t st);
// ansatz parameters [B][n_params] -> state tensors A [B][2][D][D]; kind: 0 ShallowCNOT, 1 QAOA, 2 ShallowFull (D=2), 3 ShallowCNOT3
hipError_t launch_ansatz(int D, int kind, const double* params, int n_params, void* A, int64_t B, hipStream_t st);
// ... rows with active[row] == 0 left alone (active nullable)
hipError_t launch_ansatz_masked(int D, int kind, const double* params, int n_params, void* A, int64_t B, const unsigned char* active, hipStream_t st);
// the same for rotosolve shift batches: B = nsh R evaluations, evaluation nsh r + k = row r with shift k on parameter *i_ptr
hipError_t launch_ansatz_shifted(int D, int kind, const double* params, int n_params, void* A, int64_t B, int nsh, const int* i_ptr,
                                 hipStream_t st);
// SU(N) parameters [B][stride_params] (the first N^2 - 1 of each row) -> unitaries U[B][N][N] (out_tensor = 0) or state tensors
// A[B][2][N/2][N/2] (out_tensor = 1), N in {4, 8, 16, 32}: qmps_su.hip
hipError_t launch_su_exp(int N, const double* params, int64_t B, int stride_params, void* out, int out_tensor, hipStream_t st);
// central-difference batches: 2 n_params evaluations per row, evaluation 2 P r + k = row r with +h (k < P) / -h (k >= P) on parameter k mod P
hipError_t launch_ansatz_fd(int D, int kind, const double* params, int n_params, void* A, int64_t rows, double h, hipStream_t st,
                            const unsigned char* active = nullptr);      // active: nullable [rows], 0 = leave that row's tensors alone
// time-evolution overlap (D = 2): dominant eigenvalue of the mixed two-site transfer map
struct OverlapArgs {
  const void* A;     // [B or 1][2][2][2] current state tensor(s)
  const void* Bt;    // [B][2][2][2] candidate tensors
  const void* WW;    // [4][4] two-site operator
  void* eta;         // [B] complex dominant eigenvalue
  void* r_out;       // nullable [B][2][2] unit-Frobenius right fixed point
  int32_t* iters;    // [B] squaring rounds used
  int32_t* status;   // [B]
  int64_t B;
  int a_shared;      // 1: one A for the whole batch
  int max_rounds;
  double tol;
  // time-evolution drivers (qmps_evolve_rotosolve, grouped candidates)
  int group;                   // > 0: candidate b is compared with reference tensor b / group (trajectory-major batches)
  double* f_out;               // nullable [B]: the objective -sqrt(|eta|) (qmps/new_time_evolve.py:221)
  const void* x_in;            // nullable [B][D][D]: warm start of the power method (D = 8, 16); an all-zero matrix = cold start
  int x_in_group;              // > 0: candidate b starts from x_in[b / x_in_group] (a ladder of step lengths starts from its trajectory's fixed point)
  const int* slot_ptr;         // nullable: x_in and r_out are displaced by *slot_ptr * slot_stride bytes (rotosolve keeps one
  int64_t slot_stride;         //   set of fixed points per parameter: the candidates of parameter i return to the same slot every sweep)
  unsigned long long* stats;   // nullable [kOverlapStatShards][4]: evaluations, sum of rounds, max rounds, not converged (atomics, sharded by evaluation index)
  const unsigned char* active; // nullable [B / max(group, 1)]: 0 = SKIP every candidate of that trajectory (its outputs keep the values of
                               //   the previous launch) - the lock-step optimiser drivers stop paying for trajectories that have converged
  int* queue;                  // nullable (D = 16, four waves per evaluation): counter the workgroups draw their evaluations from (zeroed by the host)
  int adjoint;                 // 1 (D = 8, 16): the LEFT fixed point - power method on the adjoint map y -> sum_s C_s^+ y Bm_s
  int no_deflation;            // 1: plain power method only (QMPS_NO_DEFLATION: the A/B switch of the deflation steps at D = 8)
  void* l_out;                 // nullable (D = 4 squaring kernel): ALSO the left fixed points [B][D][D] - the largest row of the squared map comes out
                               // of the same squarings as the largest column; eta / rounds / status of the left solve at index B + b
                               //   (eigenvalue conj(eta); eta_out receives eta itself)
  // Krylov fall-back (D = 8, 16; qmps_overlap_krylov.hip): the power kernels GIVE A CANDIDATE UP (status 1, steps used < max_rounds,
  // iterate in r_out - which must then be non-null) as soon as its residual history predicts more than `krylov_after` further
  // steps (or after 4 krylov_after steps in all); launch_overlap_d / _pair_* then run overlap_krylov_kernel over the batch.
  int env_mode;                // 1 (D = 16): the ENVIRONMENT map r -> sum_{s<2} B_s r B_s^+ of the tensors Bt (A, WW unused); candidates with status
                               //   PENDING are solved, the fixed point is rotated to a positive trace, status stays PENDING (see LaneArgs), eta is not written
  int krylov_after;            // 0: plain power method to max_rounds (QMPS_NO_KRYLOV)
  const double* tol_in;        // nullable [B / max(group, 1)]: residual tolerance PER TRAJECTORY instead of `tol` (the lock-step BFGS: a gradient's
                               //   solves need not be more accurate than 1e-3 of the gradient itself); D = 8, 16 power / Krylov kernels
  int* kry_counter;            // the fall-back's three counters [work, exit tickets, candidates given up] - zero between launches (the
                               //   power kernels count what they give up, the fall-back clears all three when it is done); null = no fall-back
};
// optimiser algebra of the lock-step BFGS time evolution on device-resident state (qmps_evolve_lockstep.hip)
struct LockstepArgs {
  double* X;        // [T][P] iterates
  double* G;        // [T][P] gradients
  double* H;        // [T][P][P] inverse Hessians
  double* F;        // [T] objective at the iterates
  double* Dv;       // [T][P] directions
  double* slope;    // [T]
  double* Xc;       // [T][P] candidates x + alpha_0 d: the parameter rows of the next evaluation
  const double* fb;      // the evaluation's objectives: [0, T) iterates / candidates, [T, T + 2 P T) central-difference neighbours
  const int32_t* st;     // ... and statuses: [0, T) right solves, [T, 2 T) left solves
  unsigned char* active; // [T] trajectories still iterating
  unsigned char* eff;    // [T] mask of the next evaluation (active, or all zero when the chain has nothing to do)
  unsigned char* need;   // [T] rejected the full step: waiting for the host's ladder
  int* ctl;              // [0] active trajectories, [1] trajectories waiting for the ladder, [2] iterations done, [3] stop;
                         // [16 + k]: trajectories that rejected the full step of iteration k of this time step
  double* fh_start;      // nullable [T]: record of the objective at the start of the time step (mode 1)
  double* fh_end;        // [T] record of the objective after the last finished iteration (rewritten by every live launch)
  double* ph;            // [T][P] ... and of the parameters
  int T, P, maxiter, reset_h;
  int step_id;           // 1, 2, ...: the time step this launch belongs to (ctl[5] = id of the last finished step, ctl[6] = its iteration count)
  unsigned char* head_mask;   // [T] 1 everywhere once the time step has finished, else 0: the mask of the NEXT step's speculative head
  int hist_off;          // offset of this time step's rejection pattern in ctl[16 ...] (two regions, alternating)
  int mode;              // 0: finish an iteration from its evaluation, open the next; 1: the same after the FIRST evaluation of a time step
                         // (f, g, active set from the batch; H^-1 = 1 if reset_h); 2: open the next iteration only (the host has finished one);
                         // 3: finish an iteration that stopped on rejected full steps, after their ladder and the gradient at the accepted points;
                         // 4: mode 1 enqueued SPECULATIVELY behind the previous time step's chain - does nothing unless that step has finished
  double h, gtol, c1, alpha0;
  // the backtracking ladder of the trajectories that rejected the full step (lockstep_ladder_*_kernel)
  double* F0;            // [T] objective of the rejected full step (rung 0)
  double* asel;          // [T] the step length the ladder chose (0: no rung decreased f)
  const double* alphas;  // [NA] the ladder (device copy); alphas[0] = alpha0
  double* cand;          // [T][NA - 1][P] candidates x + alphas[r + 1] d (trajectory-major: the rows of the ladder's batch)
  int NA;
  // per-trajectory tolerance of the NEXT evaluation's eigen-solves (QMPS_BFGS_ADAPTIVE_GRADIENT; tol_next nullptr = off):
  //   clamp(tol_rel * max|g|, tol_min, tol_max) with g the trajectory's current gradient, or - when the time step ends here - the
  //   gradient its first evaluation found (g0max: the next time step starts about as far from its minimum)
  double* tol_next;      // [T]
  double* g0max;         // [T]
  double tol_min, tol_max, tol_rel;
  // grid barrier of the step kernel: ctl[8 .. 12] (see lockstep_step_kernel); epoch = 1, 2, ... counts the launches on this control word
  int epoch, blocks;
};
hipError_t launch_lockstep_step(const LockstepArgs& a, hipStream_t st);       // n_params <= 32
int lockstep_step_blocks(int T, int n_params);
hipError_t launch_lockstep_ladder_cand(const LockstepArgs& a, hipStream_t st);
// fl / stl: objectives and statuses of the ladder's batch [T (NA - 1)]; writes asel and the accepted points into Xc
hipError_t launch_lockstep_ladder_pick(const LockstepArgs& a, const double* fl, const int32_t* stl, hipStream_t st);
// two-sided first-order evaluation of central-difference neighbours (qmps_overlap_gradient; qmps_overlap_grad.hip)
struct OverlapGradArgs {
  const void* A;       // [T][2][D][D] reference tensors
  const void* WW;      // [4][4]
  const void* r;       // [T][D][D] right fixed points of the iterates
  const void* y;       // [T][D][D] left fixed points
  void* G;             // [T][4][D][D] scratch: G_s = y^+ C_s r
  void* yr;            // [T] complex scratch: <y, r>
  const void* Bt;      // [T * 2P][2][D][D] neighbour tensors
  double* f_out;       // [T * 2P]: -sqrt|eta'|,  eta' = <y, T'(r)>/<y, r>
  int64_t T;
  int G2P;             // neighbours per trajectory (2 P)
  const unsigned char* active;   // nullable [T]: 0 = skip the trajectory (outputs keep their previous values)
  const void* Bc;      // nullable [T][2][D][D]: the iterates' own tensors - their objective by the same two-sided quotient
  double* fc_out;      // [T]: -sqrt|<y, T(r)>/<y, r>|  (error ~ the PRODUCT of the residuals of y and r)
  // D = 16, ShallowCNOT families: the probe kernel BUILDS every neighbour's tensor itself (wave-distributed circuit, LDS) instead of
  // reading it from Bt - no neighbour tensors in HBM at all.  fd_params != nullptr selects it (overlap_probe_fusable).
  const double* fd_params;   // nullable [T][G2P / 2] parameter rows of the iterates (device)
  double fd_h;               // central-difference step: neighbour k < P has +fd_h on parameter k, neighbour P + k has -fd_h
  int kind;                  // QMPS_ANSATZ_* of the rows
};
hipError_t launch_overlap_grad(int D, const OverlapGradArgs& a, hipStream_t st);
// can the D = 16 probe kernel build the neighbours of this ansatz kind itself?
inline bool overlap_probe_fusable(int D, int kind, int n_params) { return D == 16 && (kind == 0 || kind == 3) && n_params >= 1 && n_params <= 64; }
constexpr int kOverlapStatShards = 1024;
#if defined(__HIPCC__)
__device__ __forceinline__ int64_t overlap_ref_index(const OverlapArgs& p, int64_t b) { return p.group > 0 ? b / p.group : (p.a_shared ? 0 : b); }
__device__ __forceinline__ double overlap_tol2(const OverlapArgs& p, int64_t b) {
  const double t = p.tol_in != nullptr ? p.tol_in[p.group > 0 ? b / p.group : b] : p.tol;
  return t * t;
}
__device__ __forceinline__ bool overlap_skipped(const OverlapArgs& p, int64_t b) { return p.active != nullptr && p.active[p.group > 0 ? b / p.group : b] == 0; }
__device__ __forceinline__ int64_t overlap_slot_offset(const OverlapArgs& p) { return p.slot_ptr != nullptr ? (int64_t)(*p.slot_ptr) * p.slot_stride : 0; }
// results of one evaluation (called by ONE lane)
__device__ __forceinline__ void overlap_store(const OverlapArgs& p, int64_t b, double eta_r, double eta_i, int rounds, int status) {
  ((double2*)p.eta)[b] = make_double2(eta_r, eta_i);
  p.iters[b] = rounds;
  p.status[b] = status;
  if (p.f_out != nullptr) p.f_out[b] = -__builtin_sqrt(__builtin_sqrt(eta_r * eta_r + eta_i * eta_i));
  if (p.stats != nullptr) {
    // kOverlapStatShards sets of four counters (the host adds them up): 65 536 evaluations hammering ONE address cost 1.8 ms of
    // a 0.5 ms launch at D = 4
    unsigned long long* st = p.stats + 4 * ((unsigned)b & (kOverlapStatShards - 1));
    atomicAdd(st + 0, 1ULL);
    atomicAdd(st + 1, (unsigned long long)rounds);
    atomicMax(st + 2, (unsigned long long)rounds);
    if (!overlap_usable(status)) atomicAdd(st + 3, 1ULL);
  }
}
// Cold start of the power method on a mixed transfer map: the identity (the natural guess: for a candidate close to the reference the fixed
// point is the environment, positive with trace 1) PLUS 2^-12 of a fixed pseudo-random complex matrix.  The plain identity is orthogonal to the
// dominant eigenvector - or lies in the kernel of the map - at symmetric points of the ansatz (angles on the pi/4 grid): the iteration then
// converged, status 0, to eta = 0 after two steps or to the SECOND eigenvalue after thousands (found by the randomised stress of round 5,
// profiles/EXPERIMENTS.md; the reference's ARPACK starts from a random vector).  A generic start contains every eigenvector; the residual test
// cannot pass while a growing component is present.  Element (i, j) of the D x D start matrix, Frobenius norm ~ 1.
__device__ __forceinline__ double2 overlap_cold_start(int i, int j, int D) {
  unsigned h = ((unsigned)i * 131u + (unsigned)j * 31u + 7u) * 2654435761u;
  const double gr = (double)((h >> 8) & 0xFFFFu) * (1.0 / 32768.0) - 1.0;
  h = h * 2246822519u + 374761393u;
  const double gi = (double)((h >> 8) & 0xFFFFu) * (1.0 / 32768.0) - 1.0;
  const double s = D == 16 ? 0.25 : (D == 4 ? 0.5 : 1.0 / __builtin_sqrt((double)D)), eps = 1.0 / 4096.0;
  return make_double2(s * ((i == j ? 1.0 : 0.0) + eps * gr), s * eps * gi);
}
// power method -> Krylov hand-over (see OverlapArgs::krylov_after).  Called at a convergence test of step k (k >= 16) with the
// squared residual: every 32 steps or more the decay rate since the last look (bits per step) is extrapolated to tol2.
// The caller that acts on `true` counts the candidate: atomicAdd(p.kry_counter + 2, 1) by ONE lane.
// (k_ref, l_ref): the last look (k_ref = 0: none - also after anything that makes the residual jump, e.g. a deflation step).
__device__ __forceinline__ bool power_gives_up(int k, double res2, double tol2, int limit, int& k_ref, float& l_ref) {
  if (limit <= 0 || k < 16) return false;
  if (k >= 4 * limit) return true;
  if (k_ref == 0) {
    k_ref = k;
    l_ref = __log2f((float)res2);
    return false;
  }
  if (k - k_ref < 32) return false;
  const float l = __log2f((float)res2), rate = (l_ref - l) / (float)(k - k_ref);
  k_ref = k;
  l_ref = l;
  if (!(rate > 0.0f)) return true;                          // no progress over 32 steps
  return (l - __log2f((float)tol2)) > rate * (float)limit;
}
#endif
hipError_t launch_overlap(const OverlapArgs& a, hipStream_t st);   // D = 2 (lane kernel, squaring)
// D = 2: the whole BFGS time evolution in one launch, one wave per trajectory (qmps_evolve_d2.hip)
constexpr int kEvolveMaxAlphas = 16;
struct EvolveD2Args {
  double* params;        // [T][P] in: start, out: final
  const void* WW;        // [4][4] complex
  double* hinv;          // nullable [T][P][P]: in (carry) / out
  double* params_hist;   // nullable [n_steps][T][P]
  double* f_hist;        // [n_steps][2][T]: objective at the start and at the end of every time step
  int32_t* nit;          // [n_steps][T] BFGS iterations of trajectory t in step s
  double* nfev;          // nullable [T]: objective evaluations (candidates) of the whole run
  double* rounds;        // nullable [T]: squarings spent on them
  int32_t* fail;         // nullable [T]: evaluations that ended with status != 0
  int64_t T;
  int P, n_steps, maxiter, NA, max_rounds, carry_in, carry;
  int probe;             // timing experiments (QMPS_EVOLVE_PROBE, debug builds): bit 0 skip the eigen-solve, bit 1 skip the circuits
  double gtol, h, c1, tol;
  double alphas[kEvolveMaxAlphas];
  // D = 16 (qmps_evolve_d16.hip): residual at which the two solves of a gradient stop (0: tol) - the objective comes from the two-sided
  // quotient - and, with adaptive != 0, clamp(1e-3 max|g|, grad_tol, 1e-6) per gradient (QMPS_BFGS_ADAPTIVE_GRADIENT)
  double grad_tol;
  int adaptive;
  double* prof;          // nullable [T][8] (tuning builds, QMPS_EVOLVE_PROF): 100 MHz ticks of thread 0 in total / tensor / solves / G + first
                         // neighbour / other neighbours + probes / ladder, gradient passes, power steps of the slower team
};
hipError_t launch_evolve_bfgs_d2(int kind, const EvolveD2Args& a, hipStream_t st);
// D = 4: a workgroup of eight waves per trajectory (qmps_evolve_d4.hip); n_alphas - 1 <= 8, kinds 0, 1, 3
hipError_t launch_evolve_bfgs_d4(int kind, const EvolveD2Args& a, hipStream_t st);
// D = 16: a workgroup of eight waves per trajectory, two teams of four (qmps_evolve_d16.hip); ShallowCNOT / CNOT3 (kinds 0, 3);
// max_rounds = cap on the power steps of a backtracking point (gradient solves: max(max_rounds, 100 000))
hipError_t launch_evolve_bfgs_d16(int kind, const EvolveD2Args& a, hipStream_t st);
// D = 8, 16: thick-restart Arnoldi over the candidates the power kernels gave up (status 1, iters < max_rounds); a.r_out holds their
// iterates and receives the fixed points; `counter` zeroed by the caller (qmps_overlap_krylov.hip)
hipError_t launch_overlap_krylov(int D, const OverlapArgs& a, int* counter, hipStream_t st);
hipError_t launch_overlap_krylov_pair(int D, const OverlapArgs& right, const OverlapArgs& left, hipStream_t st);   // both solves of a pair launch
// D = 4, 8, 16: operator-form power method (qmps_overlap.hip); tensors [2][D][D]; max_rounds = cap on power steps;
// mfma: D = 16 on the matrix cores (one wave per evaluation) instead of the generic LDS-tile kernel
hipError_t launch_overlap_d(int D, const OverlapArgs& a, bool mfma, hipStream_t st);
// psi[0] of the overlap circuit for GIVEN environments (qmps_overlap_amp.hip): a.x_in = q [B][D][D], a.eta = amplitudes [B]
hipError_t launch_overlap_amplitude(int D, const OverlapArgs& a, hipStream_t st);
// D = 16, at most 8192 evaluations in all: right fixed points (`right`) and left fixed points (`left`, adjoint map) in ONE launch, four waves per evaluation
// (krylov_now = false: the caller launches the fall-back itself - launch_overlap_krylov_pair - if and when a status asks for it)
// central-difference neighbours built by the surplus workgroups of the D = 16 pair launch (ShallowCNOT families; see overlap_mfma_d16x4_pair_kernel)
struct NeighbourBuildArgs {
  const double* params;          // [rows][n_params] parameter rows of the iterates (device); nullptr: nothing to build
  void* out;                     // [rows * 2 n_params][2][16][16]: neighbour 2 P r + k = row r with +h (k < P) / -h (k >= P) on parameter k mod P
  int64_t rows;
  int n_params, kind;            // kind 0 (ShallowCNOT) or 3 (ShallowCNOT3)
  double h;
  const unsigned char* active;   // nullable [rows]: 0 = skip the row
};
inline bool neighbour_build_in_pair(int D, int kind, int n_params) { return D == 16 && (kind == 0 || kind == 3) && n_params >= 1 && n_params <= 64; }
hipError_t launch_overlap_pair_d16(const OverlapArgs& right, const OverlapArgs& left, hipStream_t st, bool krylov_now = true, const NeighbourBuildArgs* build = nullptr);
hipError_t launch_overlap_pair_d8(const OverlapArgs& right, const OverlapArgs& left, hipStream_t st, bool krylov_now = true);
// brick-wall (new_tdvp) contractions: what = 0 two-site <O>, 1 four-site <O>, 2 environment matrix + eigenpair, 3 manifold overlap
struct BwArgs {
  const void *U1, *U2, *U1p, *U2p;   // [B][4][4] complex
  const void* O;                      // operator: 4x4 / 16x16 (expval), 16x16 W (manifold); shared or per item
  const void *Mr, *Ml;                // [B or 1][2][2]
  void* out;                          // [B] complex
  void* mat_out;                      // nullable [B][4][4]
  void* vec_out;                      // [B][2][2]
  int32_t* status;
  int64_t B;
  int o_shared, m_shared, side, max_rounds;
  double tol;
};
hipError_t launch_bw(int what, const BwArgs& a, hipStream_t st);
hipError_t launch_opt_env(const double* params, const void* h, double k, double* f, double* parts, int64_t B,
                          hipStream_t st);
// i_ptr[0] = index of the parameter being updated, i_ptr[1] = arrival counter (both zero-initialised)
// nsh = 3: single-frequency rotosolve (shifts 0, +-pi/2); nsh = 6: double-frequency (0, pi, +-pi/2, +-pi/4)
hipError_t launch_roto_rule_probe(const double* abcd, int64_t n, int rule, double* out, hipStream_t st);
hipError_t launch_roto_update(double* base, const double* E, const int32_t* status, int R, int P, int* i_ptr, int n_terms,
                              int nsh, int rule, hipStream_t st);
hipError_t launch_roto_record(const double* E, double* hist, int R, int n_terms, const int* sweep_ptr, int stride, hipStream_t st);
hipError_t launch_unitary_to_tensor(const void* U, void* A, int D, int64_t B, hipStream_t st);
hipError_t launch_sum(const double* E, int64_t B, int n_terms, double* partial, int n_partial, double* cost,
                      hipStream_t st);
hipError_t launch_sum_final(const double* partial, int n_partial, int n_terms, double* cost, hipStream_t st);
hipError_t launch_probe_fp64(double* out, int blocks, int iters, hipStream_t st);
hipError_t launch_probe_mfma_f64(double* out, int blocks, int iters, hipStream_t st);
hipError_t launch_probe_copy(const void* src, void* dst, int64_t n16, hipStream_t st);
// copy of n8 x 8 bytes by a kernel on the given stream (pinned host memory <-> HBM without a copy-queue hop)
hipError_t launch_stage_copy(const void* src, void* dst, int64_t n8, hipStream_t st);
hipError_t launch_stage_copy2(const void* src1, void* dst1, int64_t n1, const void* src2, void* dst2, int64_t n2, hipStream_t st);

}  // namespace qmps
