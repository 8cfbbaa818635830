// qmps_ctx.h - internal to libqmps_hip.so: the context behind the opaque `qmps_ctx*` of include/qmps_hip.h and the host-side
// helpers shared by the translation units of the C-ABI (qmps_capi.hip: lifetime, states, energy path, rotosolve, exchange;
// qmps_capi_overlap.hip: the time-evolution overlap objective and its gradient; qmps_capi_evolve.hip: the evolve drivers).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <chrono>
#include <cmath>
#include <new>
#include <vector>

#include "qmps_hip.h"
#include "qmps_kernels.h"
#include "qmps_knobs.h"

namespace qmps_host {

using qmps::documented_switch;
using qmps::tuning_knob;

int fail(int code, const char* fmt, ...);       // formats qmps_last_error()'s message, returns `code`

constexpr int kMaxTerms = 16;      // (energy_block_kernel stages kMaxTerms x 16 entries in LDS: qmps_energy_block.hip kHMax)
constexpr int kSumBlocks = 256;

}  // namespace qmps_host

#define HIP_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) return qmps_host::fail(QMPS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// "Nothing throws across the ABI" (include/qmps_hip.h): every extern "C" entry point is a function-try-block closed by this macro
// (host-side std::vector / new of the optimiser drivers may throw std::bad_alloc on absurd sizes).
#define QMPS_API_CATCH                                                                                  \
  catch (const std::bad_alloc&) {                                                                       \
    return qmps_host::fail(QMPS_ERR_ARG, "out of host memory (sizes too large?)");                      \
  }                                                                                                     \
  catch (const std::exception& ex_) {                                                                   \
    return qmps_host::fail(QMPS_ERR_ARG, "C++ exception inside the library: %s", ex_.what());           \
  }                                                                                                     \
  catch (...) {                                                                                         \
    return qmps_host::fail(QMPS_ERR_ARG, "unknown C++ exception inside the library");                   \
  }

#define RCCL_TRY(expr)                                                                                           \
  do {                                                                                                           \
    ncclResult_t r_ = (expr);                                                                                    \
    if (r_ != ncclSuccess) return qmps_host::fail(QMPS_ERR_RCCL, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)

struct qmps_ctx {
  int device = -1;
  int D = 0;
  int64_t max_batch = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // ring of event pairs around the DOMINANT kernel of each qmps_energy_launch (read by qmps_kernel_time)
  static constexpr int kRing = 64;
  hipEvent_t kev0[kRing] = {}, kev1[kRing] = {};
  int64_t launches = 0;
  bool capturing = false;   // inside hipStreamBeginCapture: skip the timing events
  const char* dominant = "";
  // HBM
  void* d_A = nullptr;       // [max_batch][2][D][D] c128
  void* d_U = nullptr;       // [max_batch][2D][2D] c128 (lazy)
  void* d_U2 = nullptr;      // second unitary of a two-site unit cell (lazy)
  double* d_params = nullptr;  // ansatz parameters [max_batch][params_cap] (lazy)
  void* d_ww = nullptr;        // two-site operator of the overlap objective (lazy)
  void* d_eta = nullptr;       // overlap eigenvalues [max_batch] complex (lazy)
  void* d_ref = nullptr;       // reference tensors of the overlap objective [ref_cap][2][D][D] (lazy; its own buffer: d_U is
  int64_t ref_cap = 0;         //   overwritten by qmps_set_states(kind = UNITARY) and the two-site unit cell)
  double* d_f = nullptr;       // overlap objective -sqrt|eta| [max_batch] (lazy)
  unsigned long long* d_ostats = nullptr;   // overlap solver statistics [4] (lazy)
  char* h_pin = nullptr;       // pinned staging for the optimiser drivers' small host <-> device transfers (lazy, grown on demand):
  size_t h_pin_bytes = 0;      //   pageable buffers make every hipMemcpyAsync a blocking, internally staged copy
  unsigned char* d_active = nullptr;   // qmps_overlap_set_active: per-trajectory mask consumed by the next overlap launch (lazy, [max_batch])
  int64_t active_n = 0;                //   entries armed (0: none)
  int active_stage = 0;                //   staging slot of the last mask (two, alternating)
  hipEvent_t active_ev[2] = {};        //   ... and the event behind its copy kernel
  bool active_inflight[2] = {};
  int64_t warm_from_group = 0;         // one-shot (qmps_evolve_bfgs): the next qmps_overlap_launch starts candidate b from the resident fixed point b / warm_from_group
  bool stash_masks = false;            //   the evolve drivers (one synchronisation per batch): the mask waits in its staging slot and rides
  int64_t mask_stash_n = 0;            //   with the NEXT parameter upload of qmps_set_states_ansatz - one copy kernel instead of two
  const unsigned char* mask_stash = nullptr;
  const unsigned char* mask_host = nullptr;   //   the host copy of the mask the next launch consumes (stash mode; cleared with it): points into mask_copy
  std::vector<unsigned char> mask_copy;       //   ... which is ordinary host memory: no staging region can overwrite it
  double* d_tolarr = nullptr;                 // qmps_evolve_bfgs (host loop, QMPS_BFGS_ADAPTIVE_GRADIENT): per-trajectory tolerances of the next gradient batch
  const double* grad_tol_in = nullptr;        //   one-shot: consumed by the next qmps_overlap_gradient
  hipEvent_t step_ev0 = nullptr, step_ev1 = nullptr;   // QMPS_BFGS_TIME_STEPS: one event pair per time step
  int* h_ctl = nullptr;                       // pinned: the control word of the device-resident lock-step BFGS, read back once per chain
  void* d_lock = nullptr;                     // device-resident state of the lock-step BFGS (qmps_evolve_lockstep.hip; lazy, grown on demand)
  size_t d_lock_bytes = 0;
  unsigned char* h_mask = nullptr;            // pinned staging of the masks, SEPARATE from h_pin (never moves, never shares an offset with a result region):
                                              //   two upload slots of kMaskSlot bytes + one for the fall-back pass of qmps_overlap_gradient
  static constexpr size_t kMaskSlot = (size_t)1 << 19;
  hipEvent_t fork_after_copy = nullptr;   // one-shot: qmps_set_states_ansatz records it between the parameter upload and the tensor build
  int* d_queue = nullptr;      // overlap kernels: counters the workgroups draw their evaluations from (qmps_create; [0, 1] D = 16 queue kernels, [2, 8) Krylov fall-back of the overlap solves: two sets of three, [8, 13) of the D = 16 environment, [13] the work counter of env_power_d4_kernel - cleared in front of its launch)
  void* d_kry = nullptr;       // D = 8, 16: iterates handed from the power kernels to the Krylov fall-back when the caller keeps no fixed points [max_batch][D][D]
  void* d_y = nullptr;         // qmps_overlap_gradient: LEFT fixed points [max_batch][D][D] (lazy)
  int64_t grad_warm_T = 0;     // d_r / d_y hold the fixed points of this many trajectories' iterates (qmps_overlap_gradient)
  // qmps_evolve_bfgs: the trajectories as independent lock-step groups, a context and a host thread each (lazy; qmps_set_evolve_groups)
  std::vector<qmps_ctx*> lockstep;
  int64_t lockstep_T = 0, lockstep_cap = 0;     // trajectories they were split for / evaluations each one holds
  int evolve_groups = 0;                    // 0 = automatic, 1 = one group (the plain lock-step), K = that many
  bool one_stream = false;                  // a lock-step group at work: qmps_overlap_gradient keeps the neighbour tensors on the main stream
  void* d_xwarm = nullptr;     // qmps_evolve_rotosolve: fixed points per (parameter, candidate) (lazy, grown on demand)
  size_t xwarm_bytes = 0;
  int64_t overlap_group = 0;   // > 0: candidate b is compared with reference b / overlap_group
  void* d_scratch = nullptr;   // brick-wall inputs / outputs (lazy, grown on demand)
  size_t scratch_bytes = 0;
  int params_cap = 0;
  void* d_h = nullptr;       // [16][4][4] c128
  void* d_r = nullptr;       // [max_batch][D][D] c128
  void* d_rho = nullptr;     // [max_batch][4][4] c128 (lazy)
  double* d_E = nullptr;     // [max_batch][n_terms]
  int64_t E_capacity = 0;    // in doubles
  int32_t* d_iters = nullptr;
  int32_t* d_status = nullptr;
  double* d_partial = nullptr;  // [16][max(qmps_host::kSumBlocks, waves of the lane kernels)]
  int64_t partial_cap = 0;      // entries per term
  int64_t partials_B = -1;      // >= 0: the last launch left per-wave partial sums for this batch size
  int partials_n = 0;           //       ... in this many entries per term
  double* d_cost = nullptr;     // [16]
  double* h_cost = nullptr;     // pinned [16]
  int32_t* d_work_count = nullptr;  // [1]  hybrid solve: number of slow items handed to the squaring tail
  int32_t* d_work_idx = nullptr;    // [max_batch]
  int handoff = 0;                  // plain power steps before the squaring tail (set in qmps_create)
  int default_solver = 1;           // solver of the one-shot entry points (QMPS_ENV_POWER_SQUARING)
  int roto_rule = 0;                // double-frequency rotosolve update (QMPS_ROTO_REFERENCE: scipy's bounded search, as tools.py:451)
  int skip_rounds = 0;              // untracked squarings when handoff == 0 (set in qmps_create)
  int timing_period = 0;            // HIP events around the dominant kernel on every timing_period-th launch (0 = never, the default:
                                    // a pair of event records costs the stream several us; qmps_set_kernel_timing_period)
  int64_t samples = 0;              // launches timed so far (ring index)
  bool timed = false;               // this launch is one of them
  bool no_pair = false;             // QMPS_NO_PAIR: D = 4 energy-only launches with one lane per evaluation (tuning knob)
  bool pair_in_step = true;         // QMPS_LANE_IN_STEP: one-lane energy pass inside qmps_energy_launch
  int n_cus = 256;                  // compute units of the device (set in qmps_create)
  int matvec_period = QMPS_MATVEC_PERIOD_D4;   // D = 4: mat-vecs with T^(2^m) between two further squarings
  // state
  int n_terms = 0;
  int64_t n_states = 0;
  int64_t window = 0;               // first evaluation addressed by the launch / read-back calls (qmps_set_window)
  int64_t overlap_refs = 0;         // reference tensors resident for the overlap objective (1 = shared by the batch)
  bool have_guess = false;
  bool have_env = false;
  bool have_overlap_x = false;       // d_r holds the fixed points of the last overlap launch (QMPS_OVERLAP_WANT_R)
  bool want_rho = false;
  bool defer_sync = false;          // one-shot entry points: the setters leave their H2D copies in flight, ONE synchronisation at the end
  // ansatz-parametrised states: the parameters stay resident (d_params, or ans_src during a rotosolve run); at D = 4 the
  // direct kernel builds the tensor itself, so d_A is materialised only when something else asks for the tensors
  bool ans_have = false;            // the resident states ARE ansatz(kind, P) of the resident parameters
  bool tensors_valid = true;        // d_A holds the tensors of the resident states
  int ans_kind = 0, ans_P = 0;
  const double* ans_src = nullptr;  // parameter rows (nullptr: d_params)
  // rotosolve work buffers and the captured sweep are kept between calls (a call used to spend ~0.6 ms on hipMalloc /
  // hipFree / graph capture + instantiation - as much as three sweeps at D = 4)
  double* roto_base = nullptr;
  double* roto_hist = nullptr;
  int* roto_idx = nullptr;
  size_t roto_base_bytes = 0, roto_hist_bytes = 0;
  hipGraph_t roto_graph = nullptr;
  hipGraphExec_t roto_exec = nullptr;
  struct RotoKey {
    int64_t R = -1;
    int kind = 0, P = 0, nsh = 0, max_iter = 0, n_terms = 0, solver = 0, handoff = 0, rule = 0;
    double tol = 0.0;
    bool fused = false;
    const void *base = nullptr, *hist = nullptr, *params = nullptr, *E = nullptr;
    bool operator==(const RotoKey& o) const {
      return R == o.R && kind == o.kind && P == o.P && nsh == o.nsh && max_iter == o.max_iter && n_terms == o.n_terms && solver == o.solver &&
             handoff == o.handoff && rule == o.rule && tol == o.tol && fused == o.fused && base == o.base && hist == o.hist && params == o.params && E == o.E;
    }
  } roto_key;
  const int* ans_i = nullptr;       // rotosolve: device index of the parameter being updated
  int ans_nsh = 0;                  // rotosolve: shifts per restart (0: one parameter row per evaluation)
  // RCCL: the all-reduce runs on its own stream so that it overlaps the next step's kernels
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  hipStream_t aux_stream = nullptr;     // qmps_overlap_gradient: the neighbour tensors are built beside the eigen-solves (lazy)
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  hipStream_t comm_stream = nullptr;
  // a second communicator (ncclCommSplit of the first) on its own stream: the exchanges of consecutive ring slots
  // alternate between the two, so two small all-reduces can be in flight - the exchange keeps up with the compute stream
  // as long as an all-reduce takes less than TWO steps (a step is ~28 us; a small all-reduce over 8 GPUs 15-40 us)
  ncclComm_t comm2 = nullptr;
  hipStream_t comm_stream2 = nullptr;
  ncclComm_t comm_of(int slot) const { return (slot & 1) && comm2 ? comm2 : comm; }
  hipStream_t comm_stream_of(int slot) const { return (slot & 1) && comm2 ? comm_stream2 : comm_stream; }
  static constexpr int kCostSlots = 8;        // ring: step n's all-reduce may still be in flight while the next steps sum
  static constexpr int kMaxGroup = 16;        // steps whose summed costs may travel in ONE all-reduce
  double* d_cost_ring = nullptr;             // [kCostSlots][kMaxGroup][16]: a slot = one group of steps
  int exchange_period = 1;                   // steps per all-reduce (qmps_set_exchange_period)
  int group_fill = 0;                        // steps summed into the current group so far
  bool slot_waited = false;                  // the compute stream already waits for the current slot's previous exchange
  int64_t slot_checks = 0, slot_blocks = 0;  // host-side slot guard: times asked / times the previous exchange was still in flight
  double slot_block_ms = 0.0;                // ... and how long the host then waited (qmps_exchange_stats)
  int64_t groups = 0;                        // groups closed (exchanged or, without a communicator, just filled)
  int last_slot = -1, last_pos = -1;         // where the newest cost lives
  hipEvent_t cost_ready[kCostSlots] = {};    // sum kernels done (main stream)
  hipEvent_t cost_reduced[kCostSlots] = {};  // all-reduce done (comm stream)
  int64_t cost_launches = 0;
  // exact in-kernel cost accumulation (QMPS_FLAG_ACCUMULATE_COST): one fixed-point accumulator per ring position
  long long* d_acc = nullptr;                // [kCostSlots][kMaxGroup][kAccWords]
  long long* h_acc = nullptr;                // pinned [kAccWords]
  bool acc_is[kCostSlots][kMaxGroup] = {};   // the position's cost lives in its accumulator (not yet a double in the ring)
  bool acc_dirty[kCostSlots][kMaxGroup] = {};  // the accumulator has been added to since it was last cleared
  bool acc_after_event[kCostSlots][kMaxGroup] = {};  // cleared by a memset on the compute stream: its finish kernel must wait for an event
  double acc_scale[kCostSlots][kMaxGroup] = {};
  int acc_shards[kCostSlots][kMaxGroup] = {};
  long long acc_expect[kCostSlots][kMaxGroup] = {};   // waves (tiles of 16 evaluations) that add to each term's shards
  int* d_acc_err = nullptr;                  // set by a finish kernel whose producer never arrived (bounded poll)
  bool acc_pending = false;                  // an accumulating launch waits for its qmps_cost_launch
  int64_t acc_B = 0, acc_window = 0;
  int acc_slot = 0, acc_pos = 0;
  double h_fro = 0.0;                        // max_t ||h_t||_F (qmps_set_hamiltonian)
  long long* acc_at(int slot, int pos) const { return d_acc + ((size_t)slot * kMaxGroup + pos) * qmps::kAccWords; }
};

namespace qmps_host {

// restores a context field when the scope is left - by `return`, by an early HIP_TRY return or by an exception
template <class T>
struct Restore {
  T& ref;
  T saved;
  Restore(T& r, T now) : ref(r), saved(r) { ref = now; }
  ~Restore() { ref = saved; }
  Restore(const Restore&) = delete;
  Restore& operator=(const Restore&) = delete;
};

int bind(qmps_ctx* c);
inline size_t tensor_bytes(const qmps_ctx* c) { return (size_t)32 * c->D * c->D; }
inline size_t env_bytes(const qmps_ctx* c) { return (size_t)16 * c->D * c->D; }
int ensure_scratch(qmps_ctx* c, size_t bytes);
int ensure_E(qmps_ctx* c, int n_terms);
int check_B(const qmps_ctx* c, int64_t B);
int check_window(const qmps_ctx* c, int64_t B);      // launch / read-back calls: the window [window, window + B) must lie inside the buffers
// addresses of the window's first evaluation
inline char* win_A(const qmps_ctx* c) { return (char*)c->d_A + (size_t)c->window * tensor_bytes(c); }
inline char* win_r(const qmps_ctx* c) { return (char*)c->d_r + (size_t)c->window * env_bytes(c); }
inline double* win_E(const qmps_ctx* c) { return c->d_E + c->window * (c->n_terms > 0 ? c->n_terms : 1); }
inline int32_t* win_iters(const qmps_ctx* c) { return c->d_iters + c->window; }
inline int32_t* win_status(const qmps_ctx* c) { return c->d_status + c->window; }
bool fusable_ansatz(const qmps_ctx* c, int kind);
int ensure_pinned(qmps_ctx* c, size_t bytes);
int check_ansatz(const qmps_ctx* c, int kind, int n_params);
int ensure_tensors(qmps_ctx* c);

}  // namespace qmps_host
