/* qmps_hip.h - C-ABI of libqmps_hip.so: the MI355X (gfx950) implementation of qmps's classical
 * inner loop (batched transfer-matrix contraction -> right-environment fixed point -> two-site
 * energy).
 *
 * The reference (fergusfinn/qmps) has NO FFI/plugin layer: it is pure Python.  The de-facto
 * boundary this library sits behind is the objective callable the optimisers hand to scipy /
 * rotosolve (qmps/tools.py:242-264), whose body is
 *
 *     qmps/ground_state.py:150-168   SparseFullEnergyOptimizer.objective_function_exact_environment
 *     qmps/ground_state.py:251-266   NonSparseFullEnergyOptimizer.objective_function
 *     qmps/ground_state.py:299-331   NonSparseFullTwoSiteEnergyOptimizer.objective_function
 *
 * i.e. per evaluation: U -> A (qmps/tools.py:151-154), A -> r (qmps/tools.py:176-182, xmps
 * TransferMatrix(A).eigs()), (A, r, h) -> E (qmps/represent.py:258-262 +
 * qmps/ground_state.py:159-167).  Each entry point below cites the reference lines it replaces.
 * The Python binding a maintainer would add is shown in INTEGRATION.md and shipped in
 * qmps_amd/_lib.py (ctypes).
 *
 * Conventions
 *   - plain pointers and sizes only; every complex array is interleaved (re, im) float64, in
 *     numpy C order, so a `numpy.complex128` array can be passed as-is;
 *   - the caller owns every host buffer; the library copies in/out and retains nothing;
 *   - every function returns 0 on success or a negative QMPS_ERR_* code; qmps_last_error()
 *     returns a thread-local description of the last failure.  Nothing throws across the ABI;
 *   - per-evaluation numerical outcomes are reported in `status` (QMPS_STATUS_*), never as a
 *     function error;
 *   - a context owns one device, one HIP stream and its HBM buffers.  NOT thread-safe per
 *     context; use one context per thread/device.  Launch functions are asynchronous on the
 *     context's stream; qmps_sync() or any qmps_get_* call waits for them;
 *   - there is NO CPU fallback: without a usable gfx950 device qmps_create() fails.
 */
#ifndef QMPS_HIP_H
#define QMPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMPS_ABI_VERSION 6
#define QMPS_ABI_MINOR 5

/* error codes */
#define QMPS_OK 0
#define QMPS_ERR_ARG (-1)       /* bad argument (null pointer, size out of range, ...) */
#define QMPS_ERR_HIP (-2)       /* HIP runtime error (message in qmps_last_error()) */
#define QMPS_ERR_NO_DEVICE (-3) /* no usable gfx950 device */
#define QMPS_ERR_STATE (-4)     /* call order violated (e.g. launch before states were set) */
#define QMPS_ERR_RCCL (-5)      /* RCCL error */

/* per-evaluation status (status_out) */
#define QMPS_STATUS_OK 0            /* converged, r positive definite */
#define QMPS_STATUS_NOT_CONVERGED 1 /* power iteration hit max_iter */
#define QMPS_STATUS_NOT_PD 2        /* Cholesky of r fails: the reference's LinAlgError branch
                                       (qmps/ground_state.py:153-157) */
/* (3 is internal to the library and never returned) */
#define QMPS_STATUS_TIED 4          /* overlap path, D = 2 (ABI 6.4) and D = 4 (6.5): the dominant eigenvalues of the mixed transfer map are TIED in modulus;
                                       eta_out = their common modulus (real) and the objective -sqrt|eta| is valid - the library's optimisers
                                       use it - but there is NO unique fixed point: r_out is a mixture, not an eigenvector.
                                       Status 0 keeps meaning "r_out is the fixed point to tol" (ABI 6.2 / 6.3 returned 0 here) */

/* input kinds for qmps_energy_batch / qmps_env_batch */
#define QMPS_INPUT_TENSOR 0  /* A[B][2][D][D]  complex128 - state tensors            */
#define QMPS_INPUT_UNITARY 1 /* U[B][2D][2D]   complex128 - state unitaries; the     */
                             /* library applies unitary_to_tensor (tools.py:151-154)  */

/* environment solver selection (flags of qmps_energy_launch) */
#define QMPS_ENV_POWER 0 /* plain normalised power iteration (`krylov`, PowerCircuit) to convergence */
/* Power iteration with a repeated-squaring tail (D = 2, 4; identical to QMPS_ENV_POWER for D = 8, 16):
 * `handoff` plain steps (default 0 = squaring from the start; see qmps_set_handoff), then items that have not converged
 * continue with the power method applied 2^m steps at a time, P_m = T^(2^m) obtained by squaring the
 * D^2 x D^2 transfer matrix, r_m = herm(P_m r)/tr, until ||r_m - r_{m-1}||_F < tol.  Same fixed
 * point, same tolerance; `iters` reports the equivalent number of power steps handoff + 2^m.
 * D = 2 (ABI 6.3): when max_iter ends the chain (handoff + 2^(m+1) > max_iter) one plain step decides about the last iterate -
 * ||herm(T r_m)/tr - r_m||_F < tol accepts it with iters = handoff + 2^m + 1 - so that max_iter = 10 000 no longer means
 * "converged within 4 096 steps" there (D = 4 continues in steps of 2^m and never had that gap).
 * This is the default of the one-shot entry points. */
#define QMPS_ENV_POWER_SQUARING 1
/* DIRECT fixed-point solve (D = 2, 4 and 8; D = 16 runs QMPS_ENV_POWER_SQUARING): what the reference itself
 * does at qmps/tools.py:176-182 - an exact solve, not an iteration.  For a left isometry the transfer map preserves
 * the trace, so the environment solves the real D^2 x D^2 linear system (R - 1 + e t^T) u = e (R: the map in real
 * coordinates of the Hermitian r, t: trace functional), done by Gauss-Jordan elimination in registers.  The result is
 * ACCEPTED only if one power step moves it by less than tol (||T(r)/tr - r||_F < tol: the very criterion of the
 * iterative solvers, so `status` keeps its meaning); an evaluation that fails the test (tensor not an isometry,
 * degenerate transfer spectrum) continues inside the same launch with the power method 2^m steps at a time from
 * r_0 = 1/D.  `iters` = 1 for an accepted direct solve, else 1 + 2^m (ABI 6.3, D = 4 as D = 2: + 1 when max_iter ends the chain and one
 * plain step accepts its last iterate - max_iter = 10 000 no longer means 'converged within 4 096 steps' for a rejected solve).  A warm start (qmps_set_env_guess,
 * QMPS_FLAG_WARM_RESIDENT) is tried first at D = 4 - see QMPS_FLAG_WARM_RESIDENT - and ignored at D = 2 and 8.
 * D = 2: the 4 x 4 system is solved in the lane, in front of the squaring tail (which takes over what is not accepted).
 * D = 4: environment and energy are fused: one read of A, one store of E per evaluation.
 * D = 8: the solve (one wave per evaluation, a row of the real 64 x 64 system per lane) writes r; the power-iteration
 * kernel starts from it - its first step is the acceptance test (`iters` = 1), its plain iteration the fall-back
 * (`iters` = steps taken). */
#define QMPS_ENV_DIRECT 2
/* flag (OR into `flags` of qmps_energy_launch): do not store the environments r[B][D][D] (QMPS_ENV_DIRECT only; saves
 * 16 D^2 bytes of HBM writes per evaluation).  qmps_get_env / qmps_get_rdm / qmps_energy_only_launch then fail with
 * QMPS_ERR_STATE until a launch without the flag has run. */
#define QMPS_FLAG_NO_ENV_OUT 0x100
/* flag: the launch also accumulates cost[t] = sum_b E[b][t] inside the energy kernel (the fused D = 4 kernel of
 * QMPS_ENV_DIRECT, the D = 2 / D = 4 lane kernels of the iterative solvers - not the D = 4 squaring path, whose energy
 * pass is a separate kernel - and the D = 8, 16 kernels, where every evaluation arrives by itself) - every wave
 * adds ONE 64-bit word per term to one of up to 1024 shards: its partial sum as a FIXED-POINT integer (scale 2^k chosen
 * from ||h||_F) in the low 58 bits and an arrival count of 1 in the high 6.  Integer addition commutes, so the sum is
 * exact and independent of the order the waves finish in.  The qmps_cost_launch(B) that follows consumes it WITHOUT
 * launching a reduction kernel, and - with a communicator - without any event on the compute stream: the one-wave
 * conversion kernel in front of the all-reduce runs on the COMMUNICATION stream and polls the arrival counts (a bounded
 * number of sweeps; it cannot hang).  Measured at world size 1: 28.7-29.1 us per step with an exchange per step against
 * 28.5 us without a communicator - and 41.3 us when the exchange was ordered by an event on the compute stream.
 * Partial sums beyond the isometric bound 16 ||h||_F (tensors that are not isometries, NaN) are added to a double
 * instead.  Contract: the next call that launches must be qmps_cost_launch with the same B and window; a second
 * accumulating launch before that fails with QMPS_ERR_STATE.  At most 2048 x 60 arrivals per launch (1.9 M evaluations
 * at D = 4 direct, 7.8 M on the lane kernels, 122 880 at D = 8, 16). */
#define QMPS_FLAG_ACCUMULATE_COST 0x200
/* QMPS_FLAG_WARM_RESIDENT: start from the RESIDENT environments (left by an earlier launch that stored them, or by
 * qmps_set_env_guess) - the device-side form of the warm start, no host round trip.  With QMPS_ENV_DIRECT at D = 4 an
 * evaluation whose resident environment passes the acceptance test (one power step moves it by less than tol) skips the
 * matrix build and the elimination altogether (iterations = 1; ~7 instead of 15 kflop); the others are solved as usual
 * (iterations = 2: the rejected step + the solve's acceptance step).  The iterative solvers start their iteration from it. */
#define QMPS_FLAG_WARM_RESIDENT 0x400
/* D = 8 (ABI 6.3; ignored elsewhere - D = 2, 4 square, D = 16 always hands over): evaluations whose power iteration - the whole solve under
 * QMPS_ENV_POWER_SQUARING, the fall-back behind a direct solve that was not accepted (the elimination does not pivot: structural zeros at
 * special angles of the ansatz) under QMPS_ENV_DIRECT - predicts more than 256 further steps are finished by the thick-restart Arnoldi kernel
 * on the environment map and accepted by a finishing pass of the energy kernel (its own test, Cholesky test, energies, cost accumulation),
 * as D = 16 does: a few hundred map applications where the power method needs 30 / (1 - |lambda_2|).  Not with QMPS_ENV_POWER (the plain
 * iteration stays plain), not with max_iter <= 64.  Costs two nearly-empty launches (~3 us) per call: the one-shot entry points
 * (qmps_energy_batch*, qmps_env_batch) set it, qmps_energy_launch leaves it to the caller. */
#define QMPS_FLAG_KRYLOV_FALLBACK 0x800
/* With handoff == 0 (squaring from the start) the iterate is not tracked during the first
 * QMPS_SKIP_ROUNDS_D* squarings (no state converges in fewer than 2^skip power steps); the first
 * convergence test compares T^(2^(skip+1)) r_0 with T^(2^skip) r_0. */
#define QMPS_SKIP_ROUNDS_D2 3
#define QMPS_SKIP_ROUNDS_D4 6
/* D = 4: after the untracked squarings the power method continues with R_m = T^(2^m) itself
 * (z <- R_m z / tr: one mat-vec = 2^m power steps; stop when ||z' - z||_F < tol), and R_m is squared once
 * more after every QMPS_MATVEC_PERIOD_D4 unconverged mat-vecs.  iterations = power steps applied to r_0. */
#define QMPS_MATVEC_PERIOD_D4 4
/* Range: the untracked squarings are not rescaled, so T^(2^skip) must stay inside the double range - dominant transfer
 * eigenvalue between ~1e-4 and ~6e4 (tensors within a factor ~250 of an isometry) at D = 4; state unitaries and
 * ansatz-built tensors are exact isometries (eigenvalue 1).  Later squarings are rescaled by the measured eigenvalue. */

/* ---- environment switches -----------------------------------------------------------------
 * The library reads the environment in ONE place (qmps_amd/csrc/qmps_knobs.h).  The switches below are part of its
 * documented behaviour: each selects between two implementations of the SAME computation, so that the test-suite can run
 * both against the oracle.  Set (to any value) before the first call that would pick a kernel.
 *   QMPS_NO_FUSED_ROTO    D = 2, 8 rotosolve: one launch per parameter update instead of the whole run in one kernel
 *   QMPS_NO_FUSED_ANSATZ  D = 4: materialise ansatz-built tensors in HBM instead of building them inside the energy kernel
 *   QMPS_NO_GRAPH         rotosolve / time evolution: plain launches instead of a captured hipGraph per sweep
 *   QMPS_OVERLAP_POWER    D = 4 overlap objective: operator-form power method instead of squaring the 16 x 16 map
 *   QMPS_D16_BLOCK        D = 16: the generic LDS-tile kernels instead of the matrix-core kernels
 *   QMPS_NO_DEFLATION     D = 8, 16 overlap objective: plain power method, without the occasional shifted step that removes a slowly
 *                         decaying second eigenvector (same results; candidates with |eta_2 / eta_1| > 0.9 take 2 - 10 x more steps;
 *                         D = 16: cold starts only - warm-started batches run the lean loop)
 *   QMPS_D16_ONE_WAVE     D = 16 overlap objective, batches above 2 048 candidates: one wave per candidate with a static stride
 *                         (round 2) instead of four waves per candidate drawn from a work queue (no Krylov fall-back)
 *   QMPS_NEIGHBOURS_BESIDE  D = 16 gradient batches (ShallowCNOT families): the 2 P central-difference neighbours' tensors built by a kernel
 *                         of their own on a second stream beside the eigen-solves (round 4: two cross-stream dependencies per batch) instead
 *                         of by the surplus workgroups of the eigen-solve launch itself (same tensors to rounding: the one-lane-per-column
 *                         and the wave-distributed circuit round differently)
 *   QMPS_FUSED_PROBE      ... built inside the probe kernel, never written to HBM (measured slower: the probe kernel becomes
 *                         instruction-bound, 43 us instead of 19 for 4 352 probes; kept as the third implementation of the same numbers)
 *   QMPS_EVOLVE_SPECULATIVE_HEAD  qmps_evolve_bfgs at D = 8, 16: the head of a time step (references, first gradient batch) is enqueued behind the
 *                         previous step's chain, masked by a device-side 'finished' word, instead of after the host has read back that the
 *                         previous step finished (same numbers; measured no gain: the host is back before the device runs dry)
 *   QMPS_EVOLVE_HOST_ALGEBRA  qmps_evolve_bfgs at D = 8, 16: directions, Armijo tests, H^-1 updates and masks on the HOST between two
 *                         gradient evaluations (the round-4 loop: a synchronisation and two staged copies per evaluation) instead of
 *                         in kernels on device-resident state with the host enqueueing chains of iterations.  Same numbers, bit for bit.
 *   QMPS_NO_KRYLOV        D = 8, 16 fixed-point solves: the power method alone, to max_rounds (same results wherever it converges;
 *                         ~1/(1 - |eta_2 / eta_1|) steps)
 *   QMPS_EVOLVE_D2_SQUARING  qmps_evolve_bfgs_device at D = 2: the eigenvalue of every candidate's 4 x 4 map by squaring the map until rank one
 *                         (rounds 4-5: log2(28 / gap) rounds) instead of as the largest root of its characteristic polynomial (round 6: one product, the
 *                         four roots by Aberth's iteration, a root per lane - the same eta to ~1e-15 |eta| / gap, whatever the gap)
 *   QMPS_POWER_LANE       D = 4, QMPS_ENV_POWER: one LANE per evaluation (rounds 1-5: a wave waits for the slowest of its 64 evaluations)
 *                         instead of a DPP quad per evaluation in persistent waves that draw their evaluations from a work counter
 *                         (round 6; same iterates, same iteration counts and statuses; energies to rounding)
 * Everything else that used to be tunable from the environment (thresholds, schedules: profiles/EXPERIMENTS.md) is compiled
 * in only with -DQMPS_DEBUG_KNOBS; the shipped library ignores those variables (tests/test_cabi.py checks both lists). */

typedef struct qmps_ctx qmps_ctx;

/* ---- library / device ------------------------------------------------------------------ */
int qmps_abi_version(void);
/* additions that keep every existing signature (new entry points, new flag bits): bumps QMPS_ABI_MINOR only.
 * 6.1: qmps_set_roto_rule / qmps_get_roto_rule / qmps_roto_rule_probe, qmps_abi_minor; flags QMPS_BFGS_ADAPTIVE_GRADIENT, QMPS_BFGS_TIME_STEPS;
 *      qmps_evolve_opts_init / qmps_evolve_bfgs_opts / qmps_evolve_bfgs_device_opts (versioned option structs).
 * 6.2: qmps_evolve_bfgs_device accepts D = 16 (it refused anything but D = 2, 4 before); fixed-point solves of the overlap path: generic
 *      cold start, eta = 0 for nilpotent maps, the Gelfand route of the Krylov certificate (same signatures, see qmps_overlap_batch).
 * 6.3: qmps_overlap_amplitude (the overlap circuit's amplitude for given environments: the reference's variational route); the D = 2
 *      squaring chain of QMPS_ENV_POWER_SQUARING decides about its last iterate with one plain step when max_iter ends it;
 *      QMPS_FLAG_KRYLOV_FALLBACK (D = 8 environment solves hand long tails to the Arnoldi kernel, as D = 16 always did).
 * 6.4: QMPS_STATUS_TIED (the D = 2 overlap solves report a tie of the dominant eigenvalues by its own status instead of 0: eta is usable, r_out is
 *      not a fixed point); qmps_evolve_bfgs_device_opts: max_rounds = 0 means 100 000 power steps at D = 16 as documented (it meant 60);
 *      qmps_bw_env: the relaxed rank-one acceptance honours the caller's tol; D = 4 QMPS_ENV_POWER runs env_power_d4_kernel (QMPS_POWER_LANE: the old
 *      kernel), the D = 2 device-resident driver solves by the characteristic polynomial (QMPS_EVOLVE_D2_SQUARING: by squaring) - same results.
 * 6.5: the D = 4 overlap solves (qmps_overlap_*, qmps_evolve_bfgs_device at D = 4) report tied dominant eigenvalues as the D = 2 ones do - 30 to 44
 *      squarings without a rank-one power: eta = the common modulus (real), QMPS_STATUS_TIED; they returned status 1 - or, with max_rounds > 50, where
 *      rounding noise breaks the tie, the quotient of a noise-picked direction with status 0 (|eta| = 1.0008, 0.54 at points of the special grid where
 *      it is 1, 0.999).  A rank-one power later than round 44 is no longer believed.  The D = 4 device-resident driver eigen-solves the 2 P neighbours of
 *      a tied point one by one (their second-order expansion has no fixed points to expand round) and leaves it as scipy's BFGS does. */
int qmps_abi_minor(void);
const char* qmps_last_error(void);
/* Test hook for the contract above ("nothing throws across the ABI"): raises a C++ exception inside the library - kind 1
 * std::bad_alloc, 2 std::length_error (a vector of absurd size), 3 a foreign type - behind the same function-try-block every entry
 * point is wrapped in; returns QMPS_ERR_ARG with a message in qmps_last_error().  kind 0: QMPS_OK.  Needs no device. */
int qmps_selftest_exception(int kind);
/* number of visible HIP devices (0 and QMPS_OK when there are none / no driver) */
int qmps_device_count(int* count);
/* name / arch / CU count / HBM bytes of a device; name buffers are caller-owned */
int qmps_device_info(int device, char* name, int name_len, char* arch, int arch_len, int* compute_units,
                     int64_t* hbm_bytes);

/* ---- context --------------------------------------------------------------------------- */
/* D in {2,4,8,16}; max_batch = capacity of the HBM buffers (evaluations per launch). */
int qmps_create(int device, int D, int64_t max_batch, qmps_ctx** out);
int qmps_destroy(qmps_ctx* ctx);
int qmps_sync(qmps_ctx* ctx);

/* ---- inputs (host -> HBM, asynchronous on the context stream) --------------------------- */
/* replaces qmps/tools.py:151-154 (unitary_to_tensor) when kind == QMPS_INPUT_UNITARY */
int qmps_set_states(qmps_ctx* ctx, int64_t B, const double* states, int kind);
/* Ansatz parameters -> state tensors ON THE DEVICE (replaces gate construction + cirq.unitary,
 * qmps/ground_state.py:151-154, and unitary_to_tensor): params[B][n_params] float64.
 * Gate lists: qmps/represent.py:288-310 (ShallowCNOT, the optimisers' default), :268-285 (QAOA),
 * :382-404 (ShallowFull, D = 2, 15 angles), :334-354 (ShallowCNOT3), :312-332 (_nonuniform), :356-380 (ExactAfter4),
 * :406-423 (StateGate, D = 2).
 * The parameters stay resident.  At D = 4 (ShallowCNOT, QAOA, ShallowCNOT3) nothing else happens in this call: a
 * following qmps_energy_launch with QMPS_ENV_DIRECT builds each tensor in LDS in front of the solve (lane q of the
 * evaluation's quad simulates column q of the circuit) - 8 n_params bytes per evaluation from HBM instead of 512, no
 * tensor ever written; the tensors are materialised in HBM only when something asks for them (qmps_get_states, the
 * other solvers, qmps_energy_only_launch, the overlap objective).  Elsewhere the tensors are built here. */
#define QMPS_ANSATZ_SHALLOW_CNOT 0
#define QMPS_ANSATZ_SHALLOW_QAOA 1
#define QMPS_ANSATZ_SHALLOW_FULL 2
#define QMPS_ANSATZ_SHALLOW_CNOT3 3
#define QMPS_ANSATZ_SHALLOW_CNOT_NONUNIFORM 4 /* represent.py:312-332: 2 (log2 D + 1) angles per layer, one rz and one rx angle per qubit */
#define QMPS_ANSATZ_EXACT_AFTER4 5            /* represent.py:356-380: six angles per layer on qubits 0, 1, CNOT ladder, cyclic SWAPs */
#define QMPS_ANSATZ_STATE_GATE 6              /* represent.py:406-423: rx, rx, rz, rz, XX**e, YY**f on two qubits (D = 2) */
int qmps_set_states_ansatz(qmps_ctx* ctx, int64_t B, int kind, int n_params, const double* params);
/* SU(2D) parameters -> state tensors ON THE DEVICE: params[B][(2D)^2 - 1], U = exp(-i/2 sum_k p_k G_k) with G_k the generalised
 * Gell-Mann matrices (order: for a < b the symmetric then the antisymmetric one, rows first; then the 2D - 1 diagonal ones) -
 * what NonSparseFullEnergyOptimizer does per evaluation on the host, `U = SU(u_params, 2 D)` (qmps/ground_state.py:245, 251-266;
 * scripts/bond_dimension.py:21-50: D = 2 .. 16, 1 023 parameters at D = 16).  `SU` lives in xmps.spin, which is not in the reference
 * tree: the convention is this library's (documentation-pinned; the optimisation problem is the same for any smooth onto map).
 * Scaling and squaring of a degree-13 Taylor polynomial, one workgroup of 2D x 2D threads per evaluation; the tensor
 * A[s][i][j] = U[2 i + s][j] (qmps/tools.py:151-154) is written directly.  8 ((2D)^2 - 1) bytes per evaluation cross PCIe
 * instead of 64 D^2. */
int qmps_set_states_su(qmps_ctx* ctx, int64_t B, const double* params);
/* parameters -> energies in one round trip (the call shape of NonSparseFullEnergyOptimizer.objective_function) */
int qmps_energy_batch_su(qmps_ctx* ctx, int64_t B, const double* params, const double* h, int n_terms, int max_iter, double tol,
                         double* E_out, int32_t* iters_out, int32_t* status_out);
/* the unitaries themselves, U_out[B][N][N] complex128, N in {4, 8, 16, 32} (any context; tests, U4 = SU(., 4)) */
int qmps_su_unitaries(qmps_ctx* ctx, int64_t B, int N, const double* params /* [B][N^2 - 1] */, double* U_out);
/* Device-resident single-frequency rotosolve (qmps/rotosolve.py:154-181; the caller shape of SURVEY
 * 8(a)-10/(f)-2): R restarts in lock-step; for each parameter i ONE batch of 3 R evaluations (shifts
 * 0, +pi/2, -pi/2) - ansatz build, environment, energy - and the closed-form update all run on the device,
 * n_sweeps x n_params times, without a host round trip.  params[R][n_params] is updated in place;
 * E_hist[n_sweeps][R] receives the energy (summed over the resident Hamiltonian terms, like the
 * reference's M(x) = np.sum(eps)) after each sweep.  Needs 3 R <= max_batch and a Hamiltonian.
 * A whole sweep (n_params updates, the evaluation of the updated vectors, its record) is ONE hipGraph launch.  D = 4
 * with the direct solver: two kernels per parameter update - the energy kernel, which builds evaluation 3 r + k's
 * tensor from restart r's parameters with shift k added to the parameter being updated, and the update kernel.
 * D = 2 and D = 8 (ShallowCNOT families): every sweep of every restart inside ONE kernel launch (qmps_double_rotosolve too) -
 * restarts are independent, so a quad of lanes (D = 2) or a workgroup with one wave per shift (D = 8) owns a restart. */
int qmps_rotosolve(qmps_ctx* ctx, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter,
                   double tol, double* E_hist);
/* Device-resident DOUBLE-frequency rotosolve (qmps/tools.py:422-457, what Optimizer.optimize('Rotosolve') runs): per
 * parameter ONE batch of 6 R evaluations (shifts 0, pi, +-pi/2, +-pi/4), the fit P sin(2x + u) + Q sin(x + v)
 * (tools.py:440-447) and the minimiser the reference's `minimize_scalar(f, bounds=[-pi, pi])` (tools.py:451) returns for it,
 * all on the device: scipy's 'bounded' method, i.e. Brent's golden-section / parabolic search (xatol 1e-5, first point
 * -pi + (3 - sqrt 5) pi), restated step for step (qmps_roto_math.h) - a LOCAL minimiser of the fit, and the one the
 * reference's trajectory follows.  The minimiser is added to the parameter, which is not re-wrapped (tools.py:453-454).
 * Needs 6 R <= max_batch. */
int qmps_double_rotosolve(qmps_ctx* ctx, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter,
                          double tol, double* E_hist);
/* Update rule of the double-frequency drivers (qmps_double_rotosolve, qmps_evolve_rotosolve with 6 shifts):
 *   QMPS_ROTO_REFERENCE      (default) the reference's bounded scalar search, see above;
 *   QMPS_ROTO_GLOBAL_ARGMIN  the GLOBAL minimiser of the fitted curve on [-pi, pi) (32-point grid + bisection + Newton).
 *                            Departs from qmps/tools.py:451: never worse on the fitted curve, but a different parameter
 *                            trajectory (in the reference-run fixtures 13 % of the updates differ). */
#define QMPS_ROTO_REFERENCE 0
#define QMPS_ROTO_GLOBAL_ARGMIN 1
int qmps_set_roto_rule(qmps_ctx* ctx, int rule);
int qmps_get_roto_rule(qmps_ctx* ctx, int* rule);
/* The update rule alone, on the device: theta[i] = the step `rule` takes for the fit a sin 2x + b cos 2x + c sin x + d cos x with
 * abcd[i][4] = (a, b, c, d).  Test / audit hook: lets a caller compare the device's decisions with scipy's
 * `minimize_scalar(f, bounds=[-pi, pi]).x` on the same fits (qmps/tools.py:451) without an energy landscape around them. */
int qmps_roto_rule_probe(qmps_ctx* ctx, int64_t n, const double* abcd, int rule, double* theta);
/* read back the resident state tensors A[B][2][D][D] (tests / debugging) */
int qmps_get_states(qmps_ctx* ctx, int64_t B, double* A);
/* h[n_terms][4][4] complex128, row/col index = 2*s1+s2, s1 = left site
 * (qmps/ground_state.py:82-88 Hamiltonian.to_matrix).  1 <= n_terms <= 16. */
int qmps_set_hamiltonian(qmps_ctx* ctx, int n_terms, const double* h);
/* Batch window: a context may hold several resident batches side by side (one per Hamiltonian term group, restart
 * group, ...; qmps_set_states / qmps_set_states_ansatz fill [0, B) of the buffers).  After qmps_set_window(first) the
 * launch and read-back calls - qmps_energy_launch, qmps_energy_only_launch, qmps_cost_launch, qmps_sum_energies,
 * qmps_get_energies, qmps_get_env, qmps_get_rdm - address evaluations [first, first + B) of the resident arrays
 * (first + B <= number of resident states).  The qmps_set_* calls reset the window to 0. */
int qmps_set_window(qmps_ctx* ctx, int64_t first);
/* optional warm start r0[B][D][D] complex128 (Hermitian, any positive trace); NULL = identity/D */
int qmps_set_env_guess(qmps_ctx* ctx, int64_t B, const double* r0);

/* ---- the hot path ---------------------------------------------------------------------- */
/* One pass over the resident batch: power-iteration right environment + two-site energy for
 * each of the n_terms Hamiltonians.  Replaces qmps/tools.py:176-182 (get_env_exact) +
 * qmps/tools.py:97-108 (environment_to_unitary - dropped, only column 0 of V is ever used) +
 * qmps/represent.py:258-262 (State) + qmps/ground_state.py:159-167 (psi^+ H psi).
 * tol: stop when ||r' - r||_F < tol;  max_iter >= 1;  flags: QMPS_ENV_*.  Asynchronous. */
int qmps_energy_launch(qmps_ctx* ctx, int64_t B, int max_iter, double tol, int flags);
/* number of plain power steps before the squaring tail of QMPS_ENV_POWER_SQUARING (0 disables it) */
int qmps_set_handoff(qmps_ctx* ctx, int handoff);
int qmps_get_handoff(qmps_ctx* ctx, int* handoff);
/* the squaring schedule in force for this context (QMPS_SKIP_ROUNDS_D*, QMPS_MATVEC_PERIOD_D4 unless a tuning
 * knob overrode them): untracked squarings, and D = 4 mat-vecs between further squarings (0 for D != 4) */
int qmps_get_squaring_schedule(qmps_ctx* ctx, int* skip_rounds, int* matvec_period);
/* solver used by qmps_energy_batch / qmps_env_batch / qmps_rotosolve (default: QMPS_ENV_DIRECT at D = 2, 4 and 8,
 * QMPS_ENV_POWER_SQUARING otherwise) */
int qmps_set_default_solver(qmps_ctx* ctx, int solver);
/* Energy only, from the resident states and the resident environments (no solve): the
 * contraction chain A-Abar-h-A-Abar of the north star.  Asynchronous. */
int qmps_energy_only_launch(qmps_ctx* ctx, int64_t B);
/* sum_b E[b][t] -> cost[t] on the device (one block-reduction kernel), copied to the host.
 * This is what rotosolve's M(x) = np.sum(eps(...)) consumes (qmps/tools.py:432-433). */
int qmps_sum_energies(qmps_ctx* ctx, int64_t B, double* cost /* [n_terms] */);

/* ---- outputs (HBM -> host; each waits for outstanding launches) ------------------------- */
int qmps_get_energies(qmps_ctx* ctx, int64_t B, double* E /* [B][n_terms] */, int32_t* iters /* [B] or NULL */,
                      int32_t* status /* [B] or NULL */);
/* per-evaluation status of the last launch over the window (energy or overlap), without the energies */
int qmps_get_status(qmps_ctx* ctx, int64_t B, int32_t* status /* [B] */);
/* r[B][D][D] complex128, Hermitian, tr r = 1 (the dominant right eigen-matrix returned by
 * xmps TransferMatrix(A).eigs() at qmps/tools.py:181, up to its normalisation) */
int qmps_get_env(qmps_ctx* ctx, int64_t B, double* r);
/* two-site reduced density matrix rho[B][4][4] complex128: rho[t][s] = tr(B_t r B_s^+)/tr r */
int qmps_get_rdm(qmps_ctx* ctx, int64_t B, double* rho);

/* ---- one-shot host-buffer convenience (SURVEY 8(b) energy_batch / env_batch) ------------ */
int qmps_energy_batch(qmps_ctx* ctx, int64_t B, const double* states, int kind, const double* h, int n_terms,
                      const double* r0 /* nullable */, int max_iter, double tol, double* E_out, int32_t* iters_out,
                      int32_t* status_out);
/* The optimisers' own call shape in one round trip (qmps/ground_state.py:150-168: parameters -> gates -> unitary ->
 * environment -> energy): params[B][n_params] of ansatz `ansatz_kind` (QMPS_ANSATZ_*) in, energies out; inputs go to the
 * device asynchronously and the call synchronises once, at the read-back.  Leaves the states resident like
 * qmps_set_states_ansatz (at D = 4 with the direct solver no environments are stored). */
int qmps_energy_batch_ansatz(qmps_ctx* ctx, int64_t B, int ansatz_kind, int n_params, const double* params, const double* h,
                             int n_terms, int max_iter, double tol, double* E_out, int32_t* iters_out, int32_t* status_out);
int qmps_env_batch(qmps_ctx* ctx, int64_t B, const double* states, int kind, const double* r0 /* nullable */,
                   int max_iter, double tol, double* r_out, int32_t* iters_out, int32_t* status_out);

/* Two-site unit cell, D = 2 (qmps/ground_state.py:291-331 NonSparseFullTwoSiteEnergyOptimizer):
 * U1, U2 [B][4][4] complex128 state unitaries; environment of merge(A1, A2)
 * (qmps/time_evolve_tools.py:20-23) by power iteration, E = (E1 + E2)/2 with the two circuits
 * V1.U2.U1 and V2.U1.U2 in closed form.  status 2 if either environment is not positive definite. */
int qmps_cell2_energy_batch(qmps_ctx* ctx, int64_t B, const double* U1, const double* U2, const double* h,
                            int n_terms, int max_iter, double tol, double* E_out, int32_t* iters_out,
                            int32_t* status_out);

/* The same from the optimiser's 30 parameters per evaluation: U1 = U4(p[:15]), U2 = U4(p[15:]) (qmps/ground_state.py:300-301,
 * U4 = SU(., 4)) built on the device. */
int qmps_cell2_energy_batch_su(qmps_ctx* ctx, int64_t B, const double* params /* [B][30] */, const double* h, int n_terms, int max_iter,
                               double tol, double* E_out, int32_t* iters_out, int32_t* status_out);

/* Time-evolution overlap objective (qmps/new_time_evolve.py:193-221, scripts/loschmidt.py:209-239):
 * eta_b = dominant eigenvalue of x -> sum_{s<4} (WW . merge(A, A))_s x merge(B_b, B_b)_s^+ ; the reference's
 * 6-qubit circuit measures 2 |psi[0]| = |eta| and minimises -sqrt(|eta|).  Any bond dimension of the context
 * (the reference's `merge`, qmps/time_evolve_tools.py:20-23, hard-codes D = 2; the map itself does not).
 * A: one tensor [2][D][D] shared by the batch (a_shared = 1, the usual case: the current state) or one per item.
 * Candidates B_b: tensors (QMPS_INPUT_TENSOR), unitaries (QMPS_INPUT_UNITARY) or ansatz parameters built on the device
 * (QMPS_INPUT_ANSATZ_BASE + QMPS_ANSATZ_*, n_params each).  eta_out [B] complex128; r_out nullable [B][D][D]
 * (unit-Frobenius right fixed point, what xmps Map.right_fixed_point returns up to phase); status 1 = no unique
 * dominant eigenvalue within max_rounds.
 * Solver: D = 2 and 4 SQUARE the D^2 x D^2 matrix of the map (max_rounds <= 60 squarings, rounds_out = squarings used:
 * O(log) rounds whatever the spectral gap) - D = 2 in a lane, D = 4 as one complex 16 x 16 tile on the matrix cores,
 * until it is rank one (||M M - tr(M) M||_F < tol ||M M||_F), eta = tr(M E)/tr(M); a map whose powers collapse to rounding
 * noise within the first rounds is NILPOTENT (reference and candidate orthogonal): eta = 0, status 0 (ABI 6.2; it used to return noise);
 * D = 2 (ABI 6.2), D = 4 (ABI 6.5): dominant eigenvalues TIED in modulus (a complex-conjugate pair on the symmetric manifolds BFGS ends up on, a ring;
 * 1, 1, -1, -1 of a non-injective state on the special grid) - 30 and more squarings without a rank-one power (D = 4: none is believed after round 44,
 * where rounding noise starts to break a tie) - return their common MODULUS as a real eta with status QMPS_STATUS_TIED (ABI 6.4; 6.2 / 6.3: status 0) - the
 * reference's objective -sqrt|eta| is the same for whichever member ARPACK returns, and every optimiser of this library treats the status as usable;
 * r_out is then the largest column of the last power, a mixture, not an eigenvector;
 * D = 8, 16 run the power method in operator form from x_0 = (1 + 2^-12 G)/sqrt(D), G a fixed pseudo-random complex matrix (ABI 6.2: the
 * plain identity lies in the kernel of the map at symmetric points of the ansatz; ARPACK starts from a random vector), eta = <x, T x>, stop when
 * ||T x - eta x||_F < tol - at D = 16 on the matrix cores (v_mfma_f64_16x16x4, four waves per evaluation) - WITH A KRYLOV
 * FALL-BACK (ABI 5; the reference's route is ARPACK: xmps Map.right_fixed_point -> scipy eigs, qmps/new_time_evolve.py:201-203,
 * qmps/time_evolve_tools.py:84-91): a candidate whose residual history predicts more than 256 further power steps (looked at
 * every 32 steps from step 48 on; at the latest after 1 024 steps) is handed to a thick-restart Arnoldi solver - one workgroup per
 * candidate, 16 basis vectors and their images in LDS, the 5 dominant Schur vectors of the 16 x 16 projected map by squaring on
 * the matrix cores, restart with those (qmps_amd/csrc/qmps_overlap_krylov.hip).  It declares convergence on the same test
 * (one explicit application, ||T u - <u, T u> u||_F < tol, ||u||_F = 1) AND only once the second Schur pair is itself converged
 * far enough to be ranked below the first (|theta_2| + 100 (res_1 + res_2) < |theta_1|) - or (ABI 6.2) a Gelfand bound on the spectral
 * radius of the projected map without its first Schur vector stays 3 % inside |theta_1| (a dominant eigenvalue over a RING of equal
 * moduli, whose second pair never converges): Haar-random candidates take ~100 (D = 8) /
 * ~250 (D = 16) map applications where the power method needs 10^3 .. 10^5, pairs with |eta_2 / eta_1| = 1 - 1e-8 take 65.
 * max_rounds = cap on MAP APPLICATIONS (power steps + Arnoldi steps), rounds_out = applications used; status 1 = not converged
 * within them - two dominant eigenvalues of EQUAL modulus (no unique fixed point) always end that way. */
#define QMPS_INPUT_ANSATZ_BASE 16
int qmps_overlap_batch(qmps_ctx* ctx, int64_t B, const double* A, int a_shared, const double* states, int kind,
                       int n_params, const double* WW, int max_rounds, double tol, double* eta_out, double* r_out,
                       int32_t* rounds_out, int32_t* status_out);
/* The same in three steps, for candidates that stay resident (qmps_set_states / qmps_set_states_ansatz): reference
 * tensor(s) + two-site operator in, asynchronous launch over the resident candidates [window, window + B), results out.
 * want_r: also keep the fixed points (read back by qmps_overlap_get with r_out != NULL). */
int qmps_overlap_set(qmps_ctx* ctx, int64_t n_ref /* 1 = shared, else one per candidate */, const double* A, const double* WW);
/* The reference states given as ansatz parameters ref_params[n_ref][n_params] (QMPS_ANSATZ_*): the tensors are built on
 * the device (the reference does this per time step: A_ = iMPS([unitary_to_tensor(cirq.unitary(gate(params)))]),
 * qmps/new_time_evolve.py:281-283, scripts/loschmidt.py:368). */
int qmps_overlap_set_refs_ansatz(qmps_ctx* ctx, int64_t n_ref, int kind, int n_params, const double* ref_params, const double* WW);
/* Trajectory-major candidate batches: with group = G > 0 candidate b is compared with reference tensor b / G (G candidates -
 * rotosolve shifts, finite-difference columns, line-search points, simplex vertices - per trajectory); needs
 * n_ref G >= window + B and a window that starts at a multiple of G.  group = 0 (default, and after qmps_overlap_set*):
 * one shared reference (n_ref = 1) or one per candidate. */
int qmps_overlap_set_group(qmps_ctx* ctx, int64_t group);
/* One-shot mask for the NEXT qmps_overlap_launch / qmps_overlap_eval_ansatz / qmps_overlap_gradient: active[t] == 0 skips every
 * candidate of trajectory t (candidate group t with qmps_overlap_set_group, candidate t otherwise; iterate t of
 * qmps_overlap_gradient) - eta, objective, status, fixed points and gradient entries of a skipped trajectory keep the values the
 * previous launch left.  The lock-step optimiser drivers evaluate every trajectory in every iteration so that batch shapes and
 * warm-start slots never change; with the mask a trajectory that has converged costs nothing.  n = 0 or active = NULL disarms. */
int qmps_overlap_set_active(qmps_ctx* ctx, int64_t n, const unsigned char* active);
/* flags of qmps_overlap_launch (the argument was `want_r` in ABI 2: bit 0 keeps that meaning) */
#define QMPS_OVERLAP_WANT_R 1 /* keep the unit-Frobenius right fixed points resident (qmps_overlap_get r_out) */
/* Warm start (D = 8, 16: the power method; ignored by the squaring solvers of D = 2, 4, whose cost does not depend on the
 * start): every candidate starts from the RESIDENT fixed point of its slot - what the previous launch with
 * QMPS_OVERLAP_WANT_R left there - instead of 1/sqrt(D).  The optimisers re-evaluate nearby candidates (the reference
 * warm-starts every time step from the previous parameters, scripts/loschmidt.py:373): a slot whose candidate moved by
 * delta converges in log(delta/tol)/log(|eta_1/eta_2|) steps.  An all-zero slot means cold start.  Needs
 * QMPS_OVERLAP_WANT_R (the new fixed points replace the old ones). */
#define QMPS_OVERLAP_WARM 2
int qmps_overlap_launch(qmps_ctx* ctx, int64_t B, int max_rounds, double tol, int flags);
int qmps_overlap_get(qmps_ctx* ctx, int64_t B, double* eta_out, double* r_out, int32_t* rounds_out, int32_t* status_out);
/* the objective itself, f_b = -sqrt(|eta_b|) (qmps/new_time_evolve.py:221, scripts/loschmidt.py:238-239), computed by the kernel */
int qmps_overlap_get_objective(qmps_ctx* ctx, int64_t B, double* f_out /* [B] */);
/* The overlap CIRCUIT's amplitude for GIVEN environments (ABI 6.3) - the variational route of the reference, which does not solve for
 * the fixed point: `get_overlap` (qmps/time_evolve_tools.py:95-131: R = put_env_on_left_site(r), L = put_env_on_right_site(r^+), objective
 * -2 |psi[0]| minimised over the 8 reals of r by Nelder-Mead) and `obj_state` (qmps/new_time_evolve.py:223-247: R a StateGate).  For the
 * resident candidates [window, window + B) against the references / operator of qmps_overlap_set* (same addressing as qmps_overlap_launch,
 * candidate groups included): amp_out[b] = psi[0] = 1/2 <q^_b, T_b(q^_b)>_F with q^ = q / ||q||_F (an all-zero q gives 0), T_b the mixed
 * two-site transfer map of qmps_overlap_batch.  q [B][D][D] complex128 (row-major), amp_out [B] complex128; D = 2, 4, 8, 16 (the reference
 * has D = 2 only).  For the exact fixed point the amplitude is eta / 2.  Synchronous; leaves eta, the objective and the resident fixed
 * points of earlier launches alone. */
int qmps_overlap_amplitude(qmps_ctx* ctx, int64_t B, const double* q, double* amp_out);
/* Solver statistics accumulated by every overlap evaluation of this context since the last reset (device-side atomics):
 * evaluations, sum and maximum of their rounds (squarings at D = 2, 4; power steps at D = 8, 16), evaluations that ended
 * with status != 0.  Waits for the stream. */
int qmps_overlap_stats(qmps_ctx* ctx, int64_t* evaluations, int64_t* rounds_sum, int64_t* rounds_max, int64_t* not_converged, int reset);

/* One round trip for the optimiser drivers that keep their update on the host (lock-step BFGS, simplex methods): candidate
 * parameters params[B][n_params] (QMPS_ANSATZ_* kind) in; tensors built on the device; overlap objective against the
 * resident references (qmps_overlap_set / _set_refs_ansatz / _set_group) with the flags of qmps_overlap_launch; f_out[B] =
 * -sqrt|eta_b| and status_out[B] (nullable) back - ONE synchronisation.  eta, rounds, fixed points stay resident
 * (qmps_overlap_get). */
int qmps_overlap_eval_ansatz(qmps_ctx* ctx, int64_t B, int kind, int n_params, const double* params, int max_rounds, double tol,
                             int flags, double* f_out, int32_t* status_out);

/* Central-difference GRADIENT of the overlap objective for T iterates at once (the "finite-difference columns" of scipy's BFGS,
 * which the reference runs per time step: qmps/new_time_evolve.py:284, scripts/loschmidt.py:371) from ONE pair of eigen-solves
 * per iterate instead of 2 n_params + 1.  params[T][n_params]: iterate t is compared with resident reference t.  The library
 * solves the RIGHT and the LEFT fixed point of the iterate's map (T(r) = eta r, T^+(y) = conj(eta) y; the left one by the power
 * method on the adjoint map y -> sum_s C_s^+ y Bm_s) and evaluates each of the 2 n_params neighbours params_t +- h e_k by
 *     eta' = <y, T'(r)> / <y, r>       (exact to second order in h: error O(h^2) ~ 1e-12 at h = 1e-6)
 * - one application of the neighbour's map instead of a power iteration.  f_out[T] = -sqrt|eta_t| (from the solve),
 * g_out[T][n_params] = (f(+h e_k) - f(-h e_k)) / 2h, status_out[T] = worst of the two solves.  QMPS_OVERLAP_WARM: both power
 * iterations start from the fixed points the previous call left resident (same T).
 * QMPS_OVERLAP_TWO_SIDED_F: f_out comes from the two-sided quotient as well, f = -sqrt|<y, T(r)> / <y, r>| - its error is the
 * PRODUCT of the residuals of y and r, so the two solves may stop at tol ~ 1e-8 and still deliver eta to ~1e-16 (measured at
 * D = 16: |f - f_exact| < 1e-13 at tol = 1e-8, 16 power steps fewer than tol = 1e-12); the gradient inherits the residual to
 * first order (2e-8 at tol = 1e-8; central differences with h = 1e-6 carry 2e-10 of rounding themselves, scipy's own
 * forward differences ~1e-8).  D = 4, 8, 16; needs T (1 + 2 n_params) <= max_batch.  One synchronisation. */
#define QMPS_OVERLAP_TWO_SIDED_F 4
int qmps_overlap_gradient(qmps_ctx* ctx, int64_t T, int kind, int n_params, const double* params, double h, int max_rounds, double tol,
                          int flags, double* f_out, double* g_out, int32_t* status_out);

/* TIME EVOLUTION by lock-step BFGS, the whole run in ONE C call (BASELINE.json configs[4]; the reference's loop:
 * qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 `for _ in T: A_ = tensor(params); params =
 * minimize(obj, params, (A_, WW)).x` - scipy's BFGS with finite-difference gradients, one trajectory and one scalar objective
 * call at a time).  T independent trajectories, params[T][n_params] in / out.  Per time step the reference tensors
 * A_t = tensor(params_t) are built on the device and every trajectory runs BFGS from its current parameters, all of them in
 * LOCK-STEP - the iteration of qmps_amd/tools.py:batched_bfgs(speculative=True), decision for decision, with its host
 * arithmetic (directions, Armijo tests, rank-two inverse-Hessian updates: O(T n_params^2) per iteration) in C++ inside the
 * library instead of numpy behind five ctypes calls:
 *   objective + central-difference gradient of all iterates: qmps_overlap_gradient (one right + one left eigen-solve each,
 *     warm-started from the previous batch's fixed points; converged trajectories masked out);
 *   the full quasi-Newton step x + alphas[0] d is evaluated WITH its gradient first; only if some active trajectory fails the
 *     Armijo test f(x + a d) <= f + c1 a g.d there, the remaining rungs alphas[1 ..] are evaluated as one batch of
 *     T (n_alphas - 1) candidates (cold start) and the gradient at the accepted points follows;
 *   inverse-Hessian update skipped when s.y <= 1e-12 |s||y|; a trajectory stops when max|g| < gtol, when no rung decreases f,
 *     or after maxiter iterations.
 * flags: QMPS_BFGS_CARRY_HESSIAN - a time step starts from the inverse Hessians the previous one ended with (scipy, and the
 *   reference, start from the identity every time); QMPS_BFGS_WARM - the first gradient batch of this call starts from the
 *   fixed points a previous qmps_evolve_bfgs / qmps_overlap_gradient call left resident (same T).
 * hinv (nullable) [T][n_params][n_params]: in - the initial inverse Hessians of the first time step when QMPS_BFGS_CARRY_HESSIAN
 *   and QMPS_BFGS_WARM are both set (a continued evolution), else ignored; out - the final ones.
 * Outputs per time step: params_hist (nullable) [n_steps][T][n_params], f_hist [n_steps][2][T] objectives -sqrt|eta| at
 * the start (the previous parameters against the new references) and at the end of the time step,
 * nit_out (nullable) [n_steps] lock-step iterations, counters_out (nullable) [4] = gradient batches, ladder batches, objective
 * evaluations in scipy's count (2 n_params + 1 per gradient), summed HIP-event milliseconds of the gradient batches (with
 * counters_out NULL no events are recorded: they cost the stream ~12 us per batch).
 * tol: the accuracy of eta / of the objective.  The gradient batches use QMPS_OVERLAP_TWO_SIDED_F with their two eigen-solves
 * stopped at residual max(tol, 1e-8) (objective error ~ residual^2, gradient error ~ residual: 2e-8 against gtol); the ladder
 * batches, one-sided, iterate to tol.  QMPS_BFGS_TIGHT_GRADIENT: the gradient solves iterate to tol as well.
 * Any D; at D = 2 (the reference's own bond dimension) the 2 n_params + 1 central-difference candidates of a gradient batch are
 * eigen-solved themselves.  Needs T max(2 n_params + 1, n_alphas - 1) <= max_batch.
 * LOCK-STEP GROUPS (ABI 6).  The reference minimises every trajectory on its own (one scipy `minimize` per trajectory and time
 * step, qmps/new_time_evolve.py:286; trajectories in parallel, poincare_map/2body_scars.py:445,607); one lock-step over all T makes
 * a time step cost the slowest trajectory's iteration count and leaves the device idle during the host arithmetic between two
 * batches.  From T = 512 on the trajectories are therefore split into K <= 4 contiguous groups of >= 256 (qmps_set_evolve_groups
 * overrides), each a lock-step of its own on a context of its own (stream, resident fixed points; the first group on this context, the
 * others on internal ones created on first use, kept, and freed by qmps_destroy) driven by a host thread of its own inside this call: one group's kernels fill the gaps of the others, a
 * straggler holds back its own group only.  Every trajectory's decisions and numbers are the same as in one group.  nit_out is the
 * maximum over the groups, the counters are sums (the kernel milliseconds of overlapping streams add up: they can exceed the wall
 * time), qmps_overlap_stats pools the groups.  A QMPS_BFGS_WARM call continues where the previous call left its fixed points
 * (grouped the same way, or on this context). */
#define QMPS_BFGS_CARRY_HESSIAN 1
#define QMPS_BFGS_TIGHT_GRADIENT 4
#define QMPS_BFGS_WARM 2
/* QMPS_BFGS_ADAPTIVE_GRADIENT (D = 8, 16; ignored with QMPS_BFGS_TIGHT_GRADIENT): the two eigen-solves behind a trajectory's gradient stop at
 * residual clamp(1e-3 max|g|, max(tol, 1e-8), 1e-6) - g the trajectory's current gradient; for the first batch of a time step the
 * gradient the previous step's first batch found; for the first step of a call the tight end.  An inexact-gradient rule: relative
 * gradient error <= ~1e-3, objective (two-sided quotient) error <= 1e-12, i.e. 1e-6 |g|^2 against the Armijo margin 1e-4 |g|^2.  Same
 * minima (measured: final objectives to 1e-9, iteration counts unchanged), a third fewer power steps.  Without the flag every batch
 * solves to max(tol, 1e-8) as in ABI 6.0. */
#define QMPS_BFGS_ADAPTIVE_GRADIENT 8
/* QMPS_BFGS_TIME_STEPS (with counters_out; D = 8, 16 device-resident algebra): counters_out[3] becomes the device time of the call -
 * one HIP event pair per time step, from its first kernel to the last one enqueued - instead of the sum over the gradient batches;
 * the run itself is the un-instrumented one (chains of iterations, no event pair per batch; counters_out[0 .. 2] stay 0). */
#define QMPS_BFGS_TIME_STEPS 16
int qmps_evolve_bfgs(qmps_ctx* ctx, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps, int maxiter,
                     double gtol, double h, double c1, int n_alphas, const double* alphas, int flags, int max_rounds, double tol,
                     double* hinv, double* params_hist, double* f_hist, int32_t* nit_out, double* counters_out);

/* Lock-step groups of qmps_evolve_bfgs: 0 = automatic (the default: min(4, T / 256) from T = 512 on, else one), 1 = one group (the
 * plain lock-step), 2 .. 16 = that many.  Host threads = groups. */
int qmps_set_evolve_groups(qmps_ctx* ctx, int groups);
/* ... and the number a qmps_evolve_bfgs call with T trajectories would use. */
int qmps_get_evolve_groups(qmps_ctx* ctx, int64_t T, int* groups);

/* The same time evolution with the optimiser ON THE DEVICE, per trajectory (ABI 5; D = 2 - the reference's own bond dimension - and D = 4:
 * qmps/new_time_evolve.py:186-187, 276-292): ONE launch for the whole run, one wave per trajectory - its lanes are the
 * 2 n_params + 1 central-difference candidates and the n_alphas - 1 backtracking points of an evaluation pass (each lane simulates
 * the ansatz circuit of its parameter vector and eigen-solves its own 4 x 4 transfer map), x, g, H^-1 live in the wave's LDS, and the
 * BFGS iteration of qmps_evolve_bfgs runs to convergence time step after time step WITHOUT returning to the host and without
 * waiting for any other trajectory (qmps_amd/csrc/qmps_evolve_d2.hip).  Same decisions on the same numbers as qmps_evolve_bfgs for
 * every trajectory (the host driver's floating-point expressions are reproduced operation for operation); what differs is the
 * schedule: a trajectory's time step costs ITS iteration count, not the slowest trajectory's.
 * Arguments as qmps_evolve_bfgs (flags: QMPS_BFGS_CARRY_HESSIAN, and QMPS_BFGS_WARM = "hinv holds the inverse Hessians to continue
 * from"), except: nit_out (nullable) [n_steps][T] - iterations of EVERY trajectory in every step; counters_out (nullable) [4] =
 * objective evaluations (candidates) of the whole run, evaluations that ended with status != 0, launch milliseconds (HIP events),
 * squarings spent on the evaluations (D = 2, ABI 6.4: Aberth iterations - a candidate's eta is the largest root of the characteristic polynomial of
 * its 4 x 4 map, one product + Newton's identities + a root per lane of the candidate's quad, instead of the map squared until rank one
 * (QMPS_EVOLVE_D2_SQUARING: the squaring solve for every map, then max_rounds and tol apply as before); the iteration of a pass starts from the
 * eigenvalues of the point of the pass before (a BFGS step away: 3 iterations where a cold start takes 5 - 8) - a continued run (QMPS_BFGS_WARM) therefore
 * agrees with the one-call run to the rounding of a solve, not bit for bit; the same eta to ~1e-15 |eta| / gap, tied
 * moduli need no special path.  What a quartic cannot answer - a MULTIPLE largest root (eps^(1/m) conditioning), a root within 1e-3 of the largest,
 * a nilpotent map - the kernel hands to the squaring solve, quad by quad (points of the special grid of multiples of pi / 4; those evaluations count
 * their Aberth iterations plus their squarings, and max_rounds / tol apply to them).
 * n_params <= 16, n_alphas <= 16, 2 n_params + n_alphas <= 64.  Any T (no max_batch limit: nothing is staged per candidate).
 * D = 4 (qmps_amd/csrc/qmps_evolve_d4.hip): a WORKGROUP per trajectory, its waves are the candidates - each builds its tensor (four lanes
 * simulate the four columns of the ansatz unitary); the point itself is eigen-solved by squaring its 16 x 16 map on the matrix cores (the code of the D = 4 overlap
 * kernel); the 2 n_params neighbours of a point are evaluated to second order in h from its right and left fixed points (the largest
 * column and row of the squared map), eta' = <y, T'(r)>/<y, r>, as in qmps_overlap_gradient.
 * n_alphas <= 9, ShallowCNOT / QAOA / CNOT3; the backtracking points are eigen-solved (eight waves side by side) only when the full step is rejected.
 * D = 16 (ABI 6.2, qmps_amd/csrc/qmps_evolve_d16.hip; config 4 of BASELINE.json): a WORKGROUP of eight waves per trajectory on a compute
 * unit of its own (143 KB of LDS) - two teams of four waves iterate the right and the left fixed point of the iterate's map on the
 * matrix cores (the loop of qmps_overlap_gradient's pair launch), G_s = y^+ C_s r, then every wave builds central-difference
 * neighbours' tensors in LDS and evaluates them by the two-sided quotient; backtracking points two at a time.  ShallowCNOT / CNOT3.
 * flags also QMPS_BFGS_TIGHT_GRADIENT / QMPS_BFGS_ADAPTIVE_GRADIENT (the tolerance of a gradient's two solves, as qmps_evolve_bfgs);
 * max_rounds in [1, 2^24] = power steps of a backtracking point's solve (a gradient's solves: max(max_rounds, 100 000)); no Krylov
 * fall-back - a solve that exhausts its cap counts in counters_out[1] and its point is treated as rejected.  An ALTERNATIVE to the
 * lock-step driver, not its replacement: measured slower than qmps_evolve_bfgs at 32 ... 2 048 trajectories (a trajectory whose map
 * has a small gap holds its compute unit - and the launch - several times longer than the others; profiles/EXPERIMENTS.md). */
int qmps_evolve_bfgs_device(qmps_ctx* ctx, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps, int maxiter,
                            double gtol, double h, double c1, int n_alphas, const double* alphas, int flags, int max_rounds, double tol,
                            double* hinv, double* params_hist, double* f_hist, int32_t* nit_out, double* counters_out);

/* ---- the two evolve drivers behind ONE versioned options struct (ABI 6.1) ----------------------------------------------------------
 * qmps_evolve_bfgs / qmps_evolve_bfgs_device take 21 positional arguments; a caller that binds them by position has nothing to lean
 * on when an option is added.  The *_opts entry points take the same inputs as two structs whose FIRST field is the size the caller
 * compiled against: the library reads only that many bytes (fields beyond it take their defaults), so a struct may grow at its end
 * without breaking callers; a size larger than the library's is refused with QMPS_ERR_ARG.  qmps_evolve_opts_init fills the defaults
 * (scipy's BFGS: gtol 1e-5, the ladder 1, 1/2, ..., 1/4096, h 1e-6, c1 1e-4, maxiter 200, one time step, tol 1e-12).
 * The positional entry points above stay and are thin wrappers over the same code. */
typedef struct qmps_evolve_opts {
  uint32_t size;          /* sizeof(qmps_evolve_opts) of the caller's header */
  int32_t n_steps;
  int32_t maxiter;
  int32_t n_alphas;       /* 0: the default ladder (alphas ignored) */
  int32_t flags;          /* QMPS_BFGS_* */
  int32_t max_rounds;     /* 0: the driver's default (60 squarings at D = 2, 4; 100 000 power steps at D = 8, 16) */
  double gtol, h, c1, tol;
  const double* alphas;   /* [n_alphas] */
} qmps_evolve_opts;
typedef struct qmps_evolve_out {
  uint32_t size;          /* sizeof(qmps_evolve_out) of the caller's header */
  uint32_t reserved;
  double* hinv;           /* nullable, in / out as in qmps_evolve_bfgs */
  double* params_hist;    /* nullable [n_steps][T][n_params] */
  double* f_hist;         /* [n_steps][2][T] */
  int32_t* nit;           /* nullable: [n_steps] (qmps_evolve_bfgs_opts) / [n_steps][T] (qmps_evolve_bfgs_device_opts) */
  double* counters;       /* nullable [4] */
} qmps_evolve_out;
int qmps_evolve_opts_init(qmps_evolve_opts* opts);
int qmps_evolve_bfgs_opts(qmps_ctx* ctx, int64_t T, int kind, int n_params, double* params, const double* WW, const qmps_evolve_opts* opts,
                          const qmps_evolve_out* out);
int qmps_evolve_bfgs_device_opts(qmps_ctx* ctx, int64_t T, int kind, int n_params, double* params, const double* WW, const qmps_evolve_opts* opts,
                                 const qmps_evolve_out* out);

/* Device-resident TIME EVOLUTION by rotosolve on the overlap objective (BASELINE.json configs[4]; the reference's loop:
 * qmps/new_time_evolve.py:276-292 / scripts/loschmidt.py:367-375 `for _ in T: A_ = tensor(params); params =
 * minimize(obj, params, (A_, WW)).x`, with the rotosolve update of qmps/rotosolve.py:154-181 (nsh = 3) or
 * qmps/tools.py:422-457 (nsh = 6; the variant the reference sketches at new_time_evolve.py:292 and scripts/rotosolve.py:270-294)
 * as the minimiser).  T independent trajectories in lock-step, params[T][n_params] in / out (QMPS_ANSATZ_* kind).  Per time
 * step: reference tensors A_t = tensor(params_t) built on the device; n_sweeps sweeps; per parameter ONE batch of nsh T
 * candidates (trajectory-major: candidate nsh t + k = trajectory t with shift k on the parameter) - ansatz, overlap
 * objective f = -sqrt|eta| against A_t, closed-form / fitted update - and per sweep one batch of the T updated vectors
 * whose objective is the sweep's record: f_hist[n_steps][n_sweeps][T].  params_hist (nullable) [n_steps][T][n_params]: the
 * parameters after each time step.  Warm start from the previous parameters, as in the reference (scripts/loschmidt.py:373).
 * No host round trip inside the call: a sweep is one hipGraph; at D = 8, 16 the fixed points of every (parameter, candidate)
 * slot stay resident between sweeps and time steps (warm start of the power method, see QMPS_OVERLAP_WARM).
 * A candidate whose solve does not converge within max_rounds leaves its trajectory's parameter untouched in that update
 * (like the energy rotosolve).  Needs nsh T <= max_batch.  Statistics: qmps_overlap_stats. */
int qmps_evolve_rotosolve(qmps_ctx* ctx, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps,
                          int n_sweeps, int nsh, int max_rounds, double tol, double* params_hist, double* f_hist);

/* Variational-environment objective, D = 2 (qmps/ground_state.py:170-228, selected by
 * SparseFullEnergyOptimizer(optimize_environment=True); the objective the reference's own Rotosolve test drives,
 * tests/test_ground_state.py:250-257): params[B][30] = [U angles (15) | V angles (15)], both ShallowFullStateTensor;
 * f = energy + k (tr rho_u^2 + tr rho_v^2 - 2 tr rho_u rho_v) from four simulated circuits (k = 1 in the
 * reference).  parts_out nullable [B][4] = (energy, u_purity, v_purity, uv_purity). */
int qmps_opt_env_objective(qmps_ctx* ctx, int64_t B, const double* params, const double* h, double k, double* f_out,
                           double* parts_out);

/* ---- brick-wall ("new_tdvp") classical contractions (new_tdvp/ClassicalTDVPStripped.py) -------------
 * U1, U2 (and the primed pair) are [B][4][4] complex128 two-qubit unitaries, big-endian; the reference's
 * U.reshape(2,2,2,2) is [out0, out1, in0, in1].  Any context can be used (its bond dimension is ignored). */
/* OverlapCalculator.qbt2_exp_val (:511-544, sites = 2, O 4x4) / qbt4_exp_val (:464-496, sites = 4, O 16x16):
 * out[B] complex = <psi| 1 x O x 1 |psi>, psi the brick-wall state on 2 (sites/2 + 1) qubits (the reference
 * returns the real part).  O: one operator shared by the batch (o_shared = 1) or one per item. */
int qmps_bw_expval(qmps_ctx* ctx, int64_t B, int sites, const double* U1, const double* U2, const double* O,
                   int o_shared, double* out);
/* RightEnvironment (side = 0, :399-431) / LeftEnvironment (side = 1, :316-347): the 4x4 environment matrix
 * (mat_out nullable [B][4][4]) and its eigenpair with the reference's rule eta[np.argmax(eta)] (largest real
 * part): eta_out [B] complex, vec_out [B][2][2] (unit 2-norm, largest entry real positive).
 * status 0: eigen-residual ||M v - eta v|| < tol ||v|| of a power exp(cM)^(2^m) that is rank one (or, for a degenerate leading eigenvalue, a
 * column whose eigenvalue has the largest real part there is); an ILL-CONDITIONED eigenvector (eigenvalues clustered next to the leading one,
 * condition ~1e3: the residual stalls at ~1e-13, as numpy's eig does) is accepted once the power has been rank one for three rounds with a
 * residual below min(1e-10, 1e3 tol) (ABI 6.4: the bound follows tol; it was 1e-10 for every tol).  status 1: none of these within max_rounds. */
int qmps_bw_env(qmps_ctx* ctx, int64_t B, int side, const double* U1, const double* U2, const double* U1p,
                const double* U2p, int max_rounds, double tol, double* mat_out, double* eta_out, double* vec_out,
                int32_t* status_out);
/* ManifoldOverlap.circuit (:239-275): <0| U2'^3 (1 x U1'^2 x 1)(Ml x W x Mr)(1 x U1^2 x 1) U2^3 |0>;
 * Mr, Ml [2][2] and W [16][16] shared by the batch or per item. */
int qmps_bw_manifold(qmps_ctx* ctx, int64_t B, const double* U1, const double* U2, const double* U1p,
                     const double* U2p, const double* Mr, const double* Ml, int m_shared, const double* W, int w_shared,
                     double* out);

/* ---- timing on the context stream (HIP events) ------------------------------------------ */
int qmps_timer_begin(qmps_ctx* ctx);
int qmps_timer_end(qmps_ctx* ctx, float* milliseconds); /* waits for the end event */

/* Average duration (HIP events on the context stream, recorded around the kernel on every launch) of the
 * DOMINANT kernel of the last n_last (<= 64) qmps_energy_launch calls, and that kernel's name.  Waits
 * for the stream.  This is what bench.py's roofline.achieved is computed from. */
int qmps_kernel_time(qmps_ctx* ctx, int n_last, float* avg_ms, char* name, int name_len);
/* The two events cost ~3 us each on the stream (they fence the command processor): time only every period-th
 * qmps_energy_launch / overlap launch (0 = never, the default since ABI 4; 1 = every launch).  qmps_kernel_time then averages the timed launches
 * among the last n_last ones.  Measured at D = 4, B = 65536: 0.118 ms per step with period 1, 0.111 ms untimed. */
int qmps_set_kernel_timing_period(qmps_ctx* ctx, int period);

/* ---- multi-GPU: one process per GPU, one RCCL all-reduce of the summed cost over xGMI ---- */
#define QMPS_UNIQUE_ID_BYTES 128
int qmps_comm_unique_id(char id[QMPS_UNIQUE_ID_BYTES]); /* rank 0 creates, host code broadcasts */
int qmps_comm_init(qmps_ctx* ctx, const char id[QMPS_UNIQUE_ID_BYTES], int rank, int nranks);
int qmps_comm_destroy(qmps_ctx* ctx);
/* ranks that joined the communicator (ncclCommCount); 1 and QMPS_OK without a communicator */
int qmps_comm_count(qmps_ctx* ctx, int* nranks);
/* in-place sum over ranks of a small float64 vector held on the host (staged through HBM) */
int qmps_allreduce_sum(qmps_ctx* ctx, double* inout, int n);
/* the same with ncclMin: the best cost over the restarts of all ranks (BASELINE.json configs[3]: restarts sharded over the GPUs) */
int qmps_allreduce_min(qmps_ctx* ctx, double* inout, int n);
/* COLLECTIVE CALLS.  With a communicator, every function that exchanges costs must be called by ALL ranks, in the same
 * order, with the same exchange period: qmps_cost_launch (closes a group every `period` calls), and the calls that
 * flush a partly filled group - qmps_sync, qmps_get_cost, qmps_allreduce_cost, qmps_set_exchange_period - as well as
 * qmps_allreduce_sum.  A rank that calls one of them alone (e.g. only rank 0 logging the cost) blocks in ncclAllReduce. */
/* Asynchronous: device-side cost[t] = sum_b E[b][t] on the context stream, followed - when a
 * communicator exists - by ONE ncclAllReduce(sum, double, n_terms) over all ranks on the context's
 * COMMUNICATION stream (ordered after the sum by an event, results in a 4-slot ring), so the exchange
 * step of one batch overlaps the kernels of the next.  This is the path's single exchange step (the
 * summed cost of rotosolve's M(x), qmps/tools.py:432-433).  qmps_sync waits for both streams. */
int qmps_cost_launch(qmps_ctx* ctx, int64_t B);
/* Health of the exchange pipeline (accumulating launches with a communicator): how often the host-side slot guard was
 * asked, how often the exchange that last used the ring slot was still in flight (the host then waits: it issues steps
 * faster than the GPU runs them and is throttled at the ring), and for how long in all.  reset != 0 zeroes the counters. */
int qmps_exchange_stats(qmps_ctx* ctx, int64_t* checks, int64_t* blocked, double* blocked_ms, int reset);
/* Exchange granularity: the summed costs of `steps` consecutive qmps_cost_launch calls (1..16, default 1) travel in ONE
 * all-reduce.  Every step's cost is still reduced exactly once; qmps_get_cost / qmps_sync / a change of the period
 * exchange a partly filled group at once.  Why: the cross-stream ordering of one exchange (two event records and a
 * stream wait) costs ~14 us of command-processor time - measured with a single-rank communicator: 0.120 ms per step
 * with an exchange per step against 0.105 ms without, independent of the all-reduce itself - and shards that own
 * whole restarts do not need each other's cost before the next parameter update. */
int qmps_set_exchange_period(qmps_ctx* ctx, int steps);
/* waits for the stream and copies the (all-reduced) cost[n_terms] to the host */
int qmps_get_cost(qmps_ctx* ctx, double* cost /* [n_terms] */);
/* qmps_cost_launch + qmps_get_cost; requires a communicator */
int qmps_allreduce_cost(qmps_ctx* ctx, int64_t B, double* cost /* [n_terms] */);

/* ---- diagnostics ----------------------------------------------------------------------- */
/* FP64 FMA micro-benchmark (register-resident v_fma_f64 loop on every CU): achieved TFLOP/s */
int qmps_probe_fp64_peak(qmps_ctx* ctx, double* tflops);
/* v_mfma_f64_16x16x4_f64 issue-rate micro-benchmark with `waves_per_simd` resident waves */
int qmps_probe_fp64_mfma_peak(qmps_ctx* ctx, int waves_per_simd, double* tflops);
/* HBM streaming copy micro-benchmark: achieved GB/s (read + write bytes) */
int qmps_probe_hbm_peak(qmps_ctx* ctx, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* QMPS_HIP_H */
