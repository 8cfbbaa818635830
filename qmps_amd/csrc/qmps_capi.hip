// qmps_capi.hip - the C-ABI of libqmps_hip.so (declared in include/qmps_hip.h).
// Host-side runtime: context = one device + one HIP stream + HBM buffers; asynchronous launches;
// pinned staging for small results; native RCCL communicator for the summed-cost all-reduce.
#include <hip/hip_runtime.h>
#include <chrono>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <time.h>

#include <new>
#include <vector>

#include "qmps_hip.h"
#include "qmps_kernels.h"
#include "qmps_knobs.h"

using qmps::documented_switch;
using qmps::tuning_knob;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(QMPS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define RCCL_TRY(expr)                                                                               \
  do {                                                                                               \
    ncclResult_t r_ = (expr);                                                                        \
    if (r_ != ncclSuccess) return fail(QMPS_ERR_RCCL, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)

constexpr int kMaxTerms = 16;      // (energy_block_kernel stages kMaxTerms x 16 entries in LDS: qmps_energy_block.hip kHMax)
constexpr int kSumBlocks = 256;

}  // namespace

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
  int* d_queue = nullptr;      // D = 16 overlap kernels: two counters the workgroups draw their evaluations from (lazy)
  void* d_y = nullptr;         // qmps_overlap_gradient: LEFT fixed points [max_batch][D][D] (lazy)
  int64_t grad_warm_T = 0;     // d_r / d_y hold the fixed points of this many trajectories' iterates (qmps_overlap_gradient)
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
  double* d_partial = nullptr;  // [16][max(kSumBlocks, waves of the lane kernels)]
  int64_t partial_cap = 0;      // entries per term
  int64_t partials_B = -1;      // >= 0: the last launch left per-wave partial sums for this batch size
  int partials_n = 0;           //       ... in this many entries per term
  double* d_cost = nullptr;     // [16]
  double* h_cost = nullptr;     // pinned [16]
  int32_t* d_work_count = nullptr;  // [1]  hybrid solve: number of slow items handed to the squaring tail
  int32_t* d_work_idx = nullptr;    // [max_batch]
  int handoff = 0;                  // plain power steps before the squaring tail (set in qmps_create)
  int default_solver = 1;           // solver of the one-shot entry points (QMPS_ENV_POWER_SQUARING)
  int skip_rounds = 0;              // untracked squarings when handoff == 0 (set in qmps_create)
  int timing_period = 1;            // HIP events around the dominant kernel on every timing_period-th launch (0 = never)
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
    int kind = 0, P = 0, nsh = 0, max_iter = 0, n_terms = 0, solver = 0, handoff = 0;
    double tol = 0.0;
    bool fused = false;
    const void *base = nullptr, *hist = nullptr, *params = nullptr, *E = nullptr;
    bool operator==(const RotoKey& o) const {
      return R == o.R && kind == o.kind && P == o.P && nsh == o.nsh && max_iter == o.max_iter && n_terms == o.n_terms && solver == o.solver &&
             handoff == o.handoff && tol == o.tol && fused == o.fused && base == o.base && hist == o.hist && params == o.params && E == o.E;
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

namespace {

int bind(qmps_ctx* c) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  HIP_TRY(hipSetDevice(c->device));
  return QMPS_OK;
}

size_t tensor_bytes(const qmps_ctx* c) { return (size_t)32 * c->D * c->D; }
size_t env_bytes(const qmps_ctx* c) { return (size_t)16 * c->D * c->D; }

int ensure_scratch(qmps_ctx* c, size_t bytes) {
  if (bytes > c->scratch_bytes) {
    if (c->d_scratch) HIP_TRY(hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
  }
  return QMPS_OK;
}

int ensure_E(qmps_ctx* c, int n_terms) {
  const int64_t need = c->max_batch * n_terms;
  if (need > c->E_capacity) {
    if (c->d_E) HIP_TRY(hipFree(c->d_E));
    c->d_E = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_E, (size_t)need * sizeof(double)));
    c->E_capacity = need;
  }
  return QMPS_OK;
}

int check_B(const qmps_ctx* c, int64_t B) {
  if (B < 0 || B > c->max_batch) return fail(QMPS_ERR_ARG, "B=%lld outside [0, max_batch=%lld]", (long long)B, (long long)c->max_batch);
  return QMPS_OK;
}

// launch / read-back calls: the window [window, window + B) must lie inside the buffers
int check_window(const qmps_ctx* c, int64_t B) {
  if (B < 0 || c->window + B > c->max_batch)
    return fail(QMPS_ERR_ARG, "window [%lld, %lld) outside [0, max_batch=%lld]", (long long)c->window, (long long)(c->window + B), (long long)c->max_batch);
  return QMPS_OK;
}

// addresses of the window's first evaluation
char* win_A(const qmps_ctx* c) { return (char*)c->d_A + (size_t)c->window * tensor_bytes(c); }
char* win_r(const qmps_ctx* c) { return (char*)c->d_r + (size_t)c->window * env_bytes(c); }
double* win_E(const qmps_ctx* c) { return c->d_E + c->window * (c->n_terms > 0 ? c->n_terms : 1); }
int32_t* win_iters(const qmps_ctx* c) { return c->d_iters + c->window; }
int32_t* win_status(const qmps_ctx* c) { return c->d_status + c->window; }

// kinds the D = 4 direct kernel builds in front of the solve (three-qubit circuits with a per-layer gate list)
bool fusable_ansatz(const qmps_ctx* c, int kind) {
  static const bool off = documented_switch("QMPS_NO_FUSED_ANSATZ") != nullptr;   // A/B knob
  return !off && c->D == 4 && (kind == QMPS_ANSATZ_SHALLOW_CNOT || kind == QMPS_ANSATZ_SHALLOW_QAOA || kind == QMPS_ANSATZ_SHALLOW_CNOT3);
}

int ensure_pinned(qmps_ctx* c, size_t bytes) {
  if (bytes > c->h_pin_bytes) {
    if (c->h_pin) HIP_TRY(hipHostFree(c->h_pin));
    c->h_pin = nullptr;
    c->h_pin_bytes = 0;
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    HIP_TRY(hipHostMalloc((void**)&c->h_pin, want, hipHostMallocDefault));
    c->h_pin_bytes = want;
  }
  return QMPS_OK;
}

// (kind, n_params) of an ansatz the device builders know (qmps/represent.py:268-404)
int check_ansatz(const qmps_ctx* c, int kind, int n_params) {
  if (n_params < 1 || n_params > 4096) return fail(QMPS_ERR_ARG, "n_params=%d outside [1,4096]", n_params);
  if (kind < 0 || kind > 6) return fail(QMPS_ERR_ARG, "unknown ansatz kind %d", kind);
  if (kind == QMPS_ANSATZ_SHALLOW_FULL && (c->D != 2 || n_params != 15))
    return fail(QMPS_ERR_ARG, "ShallowFullStateTensor is a two-qubit gate: D = 2, 15 parameters");
  if (kind == QMPS_ANSATZ_STATE_GATE && (c->D != 2 || n_params < 6))
    return fail(QMPS_ERR_ARG, "StateGate is a two-qubit gate: D = 2, 6 parameters");
  if (kind == QMPS_ANSATZ_EXACT_AFTER4 && n_params % 6) return fail(QMPS_ERR_ARG, "ExactAfter4 takes six angles per layer");
  if (kind == QMPS_ANSATZ_SHALLOW_CNOT_NONUNIFORM) {
    int nq = 1;
    while ((1 << (nq - 1)) < c->D) ++nq;      // n + 1 qubits
    if (n_params % (2 * nq)) return fail(QMPS_ERR_ARG, "ShallowCNOTStateTensor_nonuniform takes %d angles per layer at D = %d", 2 * nq, c->D);
  }
  if ((kind == QMPS_ANSATZ_SHALLOW_CNOT || kind == QMPS_ANSATZ_SHALLOW_QAOA) && n_params % 2)
    return fail(QMPS_ERR_ARG, "this ansatz takes (beta, gamma) pairs");
  if (kind == QMPS_ANSATZ_SHALLOW_CNOT3 && n_params % 3) return fail(QMPS_ERR_ARG, "this ansatz takes (beta, gamma, omega) triples");
  return QMPS_OK;
}

// d_A <- tensors of the resident ansatz parameters, if nothing has built them yet
int ensure_tensors(qmps_ctx* c) {
  if (c->tensors_valid) return QMPS_OK;
  if (!c->ans_have || c->ans_nsh != 0) return fail(QMPS_ERR_STATE, "no resident states");
  HIP_TRY(qmps::launch_ansatz(c->D, c->ans_kind, c->ans_src ? c->ans_src : c->d_params, c->ans_P, c->d_A, c->n_states, c->stream));
  c->tensors_valid = true;
  return QMPS_OK;
}

qmps::LaneArgs make_args(qmps_ctx* c, int64_t B, int max_iter, double tol, bool solve) {
  qmps::LaneArgs a;
  memset(&a, 0, sizeof(a));
  a.A = win_A(c);
  a.h = c->d_h;
  a.r_in = solve ? (c->have_guess ? win_r(c) : nullptr) : win_r(c);
  a.r_out = solve ? win_r(c) : nullptr;
  a.rho_out = c->want_rho ? (char*)c->d_rho + (size_t)c->window * 256 : nullptr;
  a.E = win_E(c);
  a.iters = win_iters(c);
  a.status = win_status(c);
  a.B = B;
  a.n_terms = c->n_terms;
  a.max_iter = max_iter;
  a.tol = tol;
  return a;
}

int close_group(qmps_ctx* c);
// QMPS_FLAG_ACCUMULATE_COST: point the energy kernel at the accumulator of the ring position the following
// qmps_cost_launch will use.  adds = arrivals per term (waves or evaluations that add one word each), per_add =
// evaluations behind one arrival (bounds the partial sum: per_add ||h||_F).
int setup_accumulator(qmps_ctx* c, qmps::LaneArgs& a, int64_t B, int64_t adds, int per_add) {
  // The position the following qmps_cost_launch will use: its accumulator must be clear BEFORE the finish kernel of
  // this step starts to poll it on a communication stream (a stale word of the previous lap carries a full arrival
  // count).  Consecutive ring slots alternate between two communication streams, so every launch clears the same
  // position TWO slots ahead: the finish kernel of this step (same stream as that later slot's) completes only when
  // every wave of this kernel - the clearing one included - has arrived, and the later slot's finish kernel is queued
  // behind it.
  const int slot = (int)(c->groups % qmps_ctx::kCostSlots), pos = c->group_fill;
  const int nslot = (int)((c->groups + 2) % qmps_ctx::kCostSlots), npos = pos;
  c->acc_after_event[slot][pos] = false;
  if (c->acc_dirty[slot][pos]) {
    // unusual call order (exchange period changed, a partly filled group, an accumulated cost that was dropped): clear
    // it now on the compute stream, and order this position's finish kernel behind that by an event
    HIP_TRY(hipMemsetAsync(c->acc_at(slot, pos), 0, qmps::kAccWords * sizeof(long long), c->stream));
    c->acc_dirty[slot][pos] = false;
    c->acc_after_event[slot][pos] = true;
  }
  // a slot of the ring is touched again only after its previous exchange has finished.  Asked on the HOST (that
  // exchange, kCostSlots - 2 groups ago, has normally finished long ago): a stream wait would put a barrier packet
  // on the compute stream in every step (+4 us measured), and the compute stream carries no event either
#ifdef QMPS_DEBUG_KNOBS
  static const bool dbg_nohostwait = getenv("QMPS_DBG_NOHOSTWAIT") != nullptr;   // timing dissection only (unsafe slot reuse)
#else
  constexpr bool dbg_nohostwait = false;
#endif
  if (c->comm && !dbg_nohostwait && c->groups + 2 >= qmps_ctx::kCostSlots) {
    c->slot_checks++;
    if (hipEventQuery(c->cost_reduced[nslot]) != hipSuccess) {
      (void)hipGetLastError();
      timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      HIP_TRY(hipEventSynchronize(c->cost_reduced[nslot]));
      clock_gettime(CLOCK_MONOTONIC, &t1);
      c->slot_blocks++;
      c->slot_block_ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }
  }
  int shards = 32;
  while (shards * (int64_t)qmps::kAccMaxWavesPerShard < adds && shards < qmps::kAccMaxShards) shards *= 2;
  if (shards * (int64_t)qmps::kAccMaxWavesPerShard < adds)
    return fail(QMPS_ERR_ARG, "B=%lld too large for QMPS_FLAG_ACCUMULATE_COST (at most %lld evaluations per launch on this path)", (long long)B,
                (long long)qmps::kAccMaxShards * qmps::kAccMaxWavesPerShard * per_add);
  a.acc = c->acc_at(slot, pos);
  a.acc_zero = c->acc_dirty[nslot][npos] ? c->acc_at(nslot, npos) : nullptr;
  a.acc_shards = shards;
  // partial sums (per_add evaluations each) beyond per_add ||h||_F bypass the fixed-point sum; scale 2^k with bound 2^k <= 2^51
  const double hf = c->h_fro > 1e-300 ? c->h_fro : 1.0;
  a.acc_bound = (double)per_add * hf * (1.0 + 1e-6);
  int k = (int)floor((double)qmps::kAccOffsetBits - 1e-9 - log2(a.acc_bound));
  if (k > 1000) k = 1000;
  if (k < -1000) k = -1000;
  a.acc_scale = ldexp(1.0, k);
  c->acc_shards[slot][pos] = shards;
  c->acc_expect[slot][pos] = adds;
  c->acc_scale[slot][pos] = a.acc_scale;
  c->acc_dirty[slot][pos] = true;
  c->acc_dirty[nslot][npos] = false;
  c->acc_pending = true; c->acc_B = B; c->acc_window = c->window; c->acc_slot = slot; c->acc_pos = pos;
  c->partials_B = -1;
  return QMPS_OK;
}

}  // namespace

extern "C" {

int qmps_abi_version(void) { return QMPS_ABI_VERSION; }

const char* qmps_last_error(void) { return g_err; }

int qmps_device_count(int* count) {
  if (!count) return fail(QMPS_ERR_ARG, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return QMPS_OK;
}

int qmps_device_info(int device, char* name, int name_len, char* arch, int arch_len, int* compute_units,
                     int64_t* hbm_bytes) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (name && name_len > 0) snprintf(name, name_len, "%s", prop.name);
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return QMPS_OK;
}

int qmps_create(int device, int D, int64_t max_batch, qmps_ctx** out) {
  if (!out) return fail(QMPS_ERR_ARG, "null out");
  *out = nullptr;
  if (D != 2 && D != 4 && D != 8 && D != 16) return fail(QMPS_ERR_ARG, "bond dimension D=%d not in {2,4,8,16}", D);
  if (max_batch < 1) return fail(QMPS_ERR_ARG, "max_batch must be >= 1");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    return fail(QMPS_ERR_NO_DEVICE, "no HIP device visible: libqmps_hip has no CPU fallback");
  }
  if (device < 0 || device >= n) return fail(QMPS_ERR_ARG, "device %d outside [0,%d)", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(QMPS_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device,
                prop.gcnArchName);
  qmps_ctx* c = new (std::nothrow) qmps_ctx();
  if (!c) return fail(QMPS_ERR_ARG, "out of host memory");
  c->device = device;
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->D = D;
  c->max_batch = max_batch;
  int rc = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    for (int i = 0; i < qmps_ctx::kRing; ++i) {
      HIP_TRY(hipEventCreate(&c->kev0[i]));
      HIP_TRY(hipEventCreate(&c->kev1[i]));
    }
    HIP_TRY(hipMalloc(&c->d_A, (size_t)max_batch * tensor_bytes(c)));
    HIP_TRY(hipMalloc(&c->d_r, (size_t)max_batch * env_bytes(c)));
    HIP_TRY(hipMalloc(&c->d_h, (size_t)kMaxTerms * 256));
    HIP_TRY(hipMalloc((void**)&c->d_iters, (size_t)(max_batch + 2) * sizeof(int32_t)));     // (+ 2: read in 8-byte units by the staging copy)
    HIP_TRY(hipMalloc((void**)&c->d_status, (size_t)(max_batch + 2) * sizeof(int32_t)));
    c->partial_cap = (max_batch + 15) / 16 > kSumBlocks ? (max_batch + 15) / 16 : kSumBlocks;   // one partial per 16 (direct kernel), 32 (pair kernel) or 64 items
    HIP_TRY(hipMalloc((void**)&c->d_partial, (size_t)kMaxTerms * c->partial_cap * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&c->d_cost, kMaxTerms * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&c->d_cost_ring, (size_t)qmps_ctx::kCostSlots * qmps_ctx::kMaxGroup * kMaxTerms * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->d_cost_ring, 0, (size_t)qmps_ctx::kCostSlots * qmps_ctx::kMaxGroup * kMaxTerms * sizeof(double), c->stream));
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream2, hipStreamNonBlocking));
    for (int i = 0; i < qmps_ctx::kCostSlots; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&c->cost_ready[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->cost_reduced[i], hipEventDisableTiming));
    }
    HIP_TRY(hipHostMalloc((void**)&c->h_cost, kMaxTerms * sizeof(double), hipHostMallocDefault));
    const size_t acc_bytes = (size_t)qmps_ctx::kCostSlots * qmps_ctx::kMaxGroup * qmps::kAccWords * sizeof(long long);
    HIP_TRY(hipMalloc((void**)&c->d_acc, acc_bytes));
    HIP_TRY(hipMemsetAsync(c->d_acc, 0, acc_bytes, c->stream));
    HIP_TRY(hipHostMalloc((void**)&c->h_acc, qmps::kAccWords * sizeof(long long), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&c->d_acc_err, sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->d_acc_err, 0, sizeof(int), c->stream));
    HIP_TRY(hipMalloc((void**)&c->d_work_count, sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&c->d_work_idx, (size_t)max_batch * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(c->d_work_count, 0, sizeof(int32_t), c->stream));
    c->handoff = 0;   // D = 2, 4: squaring from the start (fastest); D = 8, 16 have no squaring path
    c->default_solver = (D == 2 || D == 4 || D == 8) ? QMPS_ENV_DIRECT : QMPS_ENV_POWER_SQUARING;
    c->skip_rounds = (D == 2) ? QMPS_SKIP_ROUNDS_D2 : QMPS_SKIP_ROUNDS_D4;
    if (const char* e = tuning_knob("QMPS_SKIP_ROUNDS")) c->skip_rounds = atoi(e);   // tuning knob
    if (const char* e = tuning_knob("QMPS_MATVEC_PERIOD")) c->matvec_period = atoi(e);   // tuning knob
    c->no_pair = tuning_knob("QMPS_NO_PAIR") != nullptr;
    c->pair_in_step = tuning_knob("QMPS_LANE_IN_STEP") == nullptr;
    return QMPS_OK;
  }();
  if (rc != QMPS_OK) {
    char keep[512];
    snprintf(keep, sizeof(keep), "%s", g_err);
    qmps_destroy(c);
    snprintf(g_err, sizeof(g_err), "%s", keep);
    return rc;
  }
  *out = c;
  return QMPS_OK;
}

int qmps_destroy(qmps_ctx* c) {
  if (!c) return QMPS_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamSynchronize(c->comm_stream2);
  if (c->roto_exec) (void)hipGraphExecDestroy(c->roto_exec);
  if (c->roto_graph) (void)hipGraphDestroy(c->roto_graph);
  if (c->comm2) (void)ncclCommDestroy(c->comm2);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  for (int i = 0; i < qmps_ctx::kCostSlots; ++i) {
    if (c->cost_ready[i]) (void)hipEventDestroy(c->cost_ready[i]);
    if (c->cost_reduced[i]) (void)hipEventDestroy(c->cost_reduced[i]);
  }
  if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
  if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
  if (c->aux_join) (void)hipEventDestroy(c->aux_join);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamDestroy(c->comm_stream2);
  void* bufs[] = {c->d_A, c->d_U, c->d_U2, c->d_params, c->d_ww, c->d_eta, c->d_ref, c->d_f, c->d_ostats, c->d_xwarm, c->d_y, c->d_queue, c->d_active, c->d_scratch, c->d_h, c->d_r, c->d_rho, c->d_E, c->d_iters, c->d_status, c->d_partial, c->d_cost, c->d_cost_ring, c->d_work_count, c->d_work_idx, c->d_acc, c->d_acc_err, c->roto_base, c->roto_hist, c->roto_idx};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (c->h_cost) (void)hipHostFree(c->h_cost);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_acc) (void)hipHostFree(c->h_acc);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int i = 0; i < qmps_ctx::kRing; ++i) {
    if (c->kev0[i]) (void)hipEventDestroy(c->kev0[i]);
    if (c->kev1[i]) (void)hipEventDestroy(c->kev1[i]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return QMPS_OK;
}

int qmps_sync(qmps_ctx* c) {
  if (int rc = bind(c)) return rc;
  if (int rc = close_group(c)) return rc;     // costs still waiting for their exchange go out now
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->comm_stream));
  HIP_TRY(hipStreamSynchronize(c->comm_stream2));
  return QMPS_OK;
}

int qmps_set_states(qmps_ctx* c, int64_t B, const double* states, int kind) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!states && B > 0) return fail(QMPS_ERR_ARG, "null states");
  if (kind == QMPS_INPUT_TENSOR) {
    HIP_TRY(hipMemcpyAsync(c->d_A, states, (size_t)B * tensor_bytes(c), hipMemcpyHostToDevice, c->stream));
  } else if (kind == QMPS_INPUT_UNITARY) {
    if (!c->d_U) HIP_TRY(hipMalloc(&c->d_U, (size_t)c->max_batch * 2 * tensor_bytes(c)));
    HIP_TRY(hipMemcpyAsync(c->d_U, states, (size_t)B * 2 * tensor_bytes(c), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(qmps::launch_unitary_to_tensor(c->d_U, c->d_A, c->D, B, c->stream));
  } else {
    return fail(QMPS_ERR_ARG, "unknown input kind %d", kind);
  }
  // the caller's host buffer may be pageable and re-used right after the call returns
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_states = B;
  c->window = 0;
  c->have_guess = false;
  c->have_env = false;
  c->ans_have = false;
  c->tensors_valid = true;
  return QMPS_OK;
}

int qmps_set_window(qmps_ctx* c, int64_t first) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (first < 0 || first > c->n_states) return fail(QMPS_ERR_ARG, "window start %lld outside the %lld resident states", (long long)first, (long long)c->n_states);
  c->window = first;
  c->partials_B = -1;
  return QMPS_OK;
}

int qmps_set_states_ansatz(qmps_ctx* c, int64_t B, int kind, int n_params, const double* params) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!params && B > 0) return fail(QMPS_ERR_ARG, "null params");
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  if (n_params > c->params_cap) {
    if (c->d_params) HIP_TRY(hipFree(c->d_params));
    c->d_params = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_params, (size_t)c->max_batch * n_params * sizeof(double)));
    c->params_cap = n_params;
  }
  {
    const size_t pb = (size_t)B * n_params * sizeof(double);
    if (c->defer_sync && pb <= (8u << 20)) {
      // one-round-trip callers: through pinned memory, moved by a kernel on the context stream (no copy-queue hop)
      if (int rc = ensure_pinned(c, (16u << 20))) return rc;
      memcpy(c->h_pin, params, pb);
      HIP_TRY(qmps::launch_stage_copy(c->h_pin, c->d_params, (int64_t)(pb / 8), c->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(c->d_params, params, pb, hipMemcpyHostToDevice, c->stream));
    }
  }
  c->ans_have = true; c->ans_kind = kind; c->ans_P = n_params; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
  c->tensors_valid = false;
  c->n_states = B;
  // D = 4: the direct kernel builds the tensors itself (8 P bytes per evaluation instead of 512); d_A is filled on demand
  if (!fusable_ansatz(c, kind))
    if (int rc = ensure_tensors(c)) return rc;
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->window = 0;
  c->have_guess = false;
  c->have_env = false;
  return QMPS_OK;
}

int qmps_set_states_su(qmps_ctx* c, int64_t B, const double* params) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!params && B > 0) return fail(QMPS_ERR_ARG, "null params");
  const int N = 2 * c->D, np_ = N * N - 1;
  if (np_ > c->params_cap) {
    if (c->d_params) HIP_TRY(hipFree(c->d_params));
    c->d_params = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_params, (size_t)c->max_batch * np_ * sizeof(double)));
    c->params_cap = np_;
  }
  HIP_TRY(hipMemcpyAsync(c->d_params, params, (size_t)B * np_ * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(qmps::launch_su_exp(N, c->d_params, B, np_, c->d_A, 1, c->stream));
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_states = B;
  c->window = 0;
  c->have_guess = false;
  c->have_env = false;
  c->ans_have = false;
  c->tensors_valid = true;
  return QMPS_OK;
}

int qmps_energy_batch_su(qmps_ctx* c, int64_t B, const double* params, const double* h, int n_terms, int max_iter, double tol,
                         double* E_out, int32_t* iters_out, int32_t* status_out) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  c->defer_sync = true;
  int rc = qmps_set_states_su(c, B, params);
  if (!rc) rc = qmps_set_hamiltonian(c, n_terms, h);
  if (!rc) rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver);
  c->defer_sync = false;
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}

int qmps_su_unitaries(qmps_ctx* c, int64_t B, int N, const double* params, double* U_out) {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !params || !U_out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (N != 4 && N != 8 && N != 16 && N != 32) return fail(QMPS_ERR_ARG, "N=%d not in {4, 8, 16, 32}", N);
  const size_t pb = (size_t)B * (N * N - 1) * sizeof(double), ub = (size_t)B * N * N * 16;
  if (int rc = ensure_scratch(c, ((pb + 255) & ~(size_t)255) + ub + 256)) return rc;
  double* d_p = (double*)c->d_scratch;
  void* d_u = (char*)c->d_scratch + ((pb + 255) & ~(size_t)255);
  HIP_TRY(hipMemcpyAsync(d_p, params, pb, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(qmps::launch_su_exp(N, d_p, B, N * N - 1, d_u, 0, c->stream));
  HIP_TRY(hipMemcpyAsync(U_out, d_u, ub, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

namespace {
int rotosolve_impl(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter, double tol,
                   double* E_hist, int nsh);
}

int qmps_rotosolve(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter,
                   double tol, double* E_hist) {
  return rotosolve_impl(c, R, kind, n_params, params, n_sweeps, max_iter, tol, E_hist, 3);
}

int qmps_double_rotosolve(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter,
                          double tol, double* E_hist) {
  return rotosolve_impl(c, R, kind, n_params, params, n_sweeps, max_iter, tol, E_hist, 6);
}

namespace {
int rotosolve_impl(qmps_ctx* c, int64_t R, int kind, int n_params, double* params, int n_sweeps, int max_iter, double tol,
                   double* E_hist, int nsh) {
  if (int rc = bind(c)) return rc;
  if (R < 1 || nsh * R > c->max_batch) return fail(QMPS_ERR_ARG, "%d R = %lld evaluations exceed max_batch = %lld", nsh, (long long)(nsh * R), (long long)c->max_batch);
  c->window = 0;
  if (!params || !E_hist) return fail(QMPS_ERR_ARG, "null argument");
  if (n_sweeps < 1) return fail(QMPS_ERR_ARG, "n_sweeps must be >= 1");
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "qmps_set_hamiltonian has not been called");
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  if (n_params > c->params_cap) {
    if (c->d_params) HIP_TRY(hipFree(c->d_params));
    c->d_params = nullptr;
    HIP_TRY(hipMalloc((void**)&c->d_params, (size_t)c->max_batch * n_params * sizeof(double)));
    c->params_cap = n_params;
  }
  auto grow = [&](double*& buf, size_t& have, size_t need) -> int {
    if (need > have) {
      if (buf) HIP_TRY(hipFree(buf));
      buf = nullptr;
      have = 0;
      HIP_TRY(hipMalloc((void**)&buf, need));
      have = need;
    }
    return QMPS_OK;
  };
  if (int rc = grow(c->roto_base, c->roto_base_bytes, (size_t)R * n_params * sizeof(double))) return rc;
  // The run's results (final parameters, energy history) come back through the context's pinned buffer when they fit: the
  // first LARGE copy into pageable memory makes the runtime set up its internal staging, ~8 ms once per process (measured
  // in the first 160-sweep call after an 8-sweep one: 27.7 instead of 19.5 us per parameter update at D = 8).
  auto download_results = [&](size_t hist_doubles) -> int {
    const size_t pb = (size_t)R * n_params * sizeof(double), hb = hist_doubles * sizeof(double);
    if (pb + hb <= (8u << 20)) {
      if (int e = ensure_pinned(c, (16u << 20))) return e;
      HIP_TRY(hipMemcpyAsync(c->h_pin, c->roto_base, pb, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_pin + pb, c->roto_hist, hb, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      memcpy(params, c->h_pin, pb);
      memcpy(E_hist, c->h_pin + pb, hb);
      return QMPS_OK;
    }
    HIP_TRY(hipMemcpyAsync(params, c->roto_base, pb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(E_hist, c->roto_hist, hb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMPS_OK;
  };
#ifdef QMPS_D8_PROFILE        // scratch instrumentation build (tools/scratch/d8_profile.py): 16 phase clocks behind the history
  constexpr size_t kHistExtra = 16 + 3 * 4096;
#else
  constexpr size_t kHistExtra = 0;
#endif
#ifdef QMPS_D8_PROFILE
  const auto g0 = std::chrono::steady_clock::now();
#endif
  if (int rc = grow(c->roto_hist, c->roto_hist_bytes, ((size_t)R * n_sweeps + kHistExtra) * sizeof(double))) return rc;
#ifdef QMPS_D8_PROFILE
  fprintf(stderr, "[d8 profile] history buffer: %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g0).count());
#endif
  if (!c->roto_idx) HIP_TRY(hipMalloc((void**)&c->roto_idx, 4 * sizeof(int)));
  double *d_base = c->roto_base, *d_hist = c->roto_hist;
  int* d_idx = c->roto_idx;
  int rc = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(d_base, params, (size_t)R * n_params * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_idx, 0, 3 * sizeof(int), c->stream));   // parameter index, arrival counter, finished sweeps
    if (kHistExtra) HIP_TRY(hipMemsetAsync(d_hist + (size_t)R * n_sweeps, 0, kHistExtra * sizeof(double), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const bool saved_guess = c->have_guess;
    c->have_guess = false;
    // D = 2 with the library's default solver: the whole run is ONE launch (restarts are independent, see
    // rotosolve_fused_d2_kernel); afterwards one ordinary evaluation of the final parameters leaves the context's
    // resident tensors / energies / statuses exactly as the step-by-step path does.
    if ((nsh == 3 || nsh == 6) && c->D == 2 && c->handoff == 0 && (c->default_solver == QMPS_ENV_POWER_SQUARING || c->default_solver == QMPS_ENV_DIRECT) && n_params <= 64 &&
        documented_switch("QMPS_NO_FUSED_ROTO") == nullptr) {
      qmps::RotoArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.base = d_base; ra.h = c->d_h; ra.hist = d_hist;
      ra.R = (int)R; ra.P = n_params; ra.n_terms = c->n_terms; ra.n_sweeps = n_sweeps; ra.max_iter = max_iter;
      ra.skip = c->skip_rounds; ra.tol = tol; ra.direct = c->default_solver == QMPS_ENV_DIRECT ? 1 : 0; ra.nsh = nsh;
      HIP_TRY(qmps::launch_rotosolve_fused_d2(kind, ra, c->stream));
      HIP_TRY(qmps::launch_ansatz(c->D, kind, d_base, n_params, c->d_A, R, c->stream));
      c->n_states = R; c->ans_have = false; c->tensors_valid = true;
      if (int e = qmps_energy_launch(c, R, max_iter, tol, c->default_solver)) return e;
      c->have_guess = saved_guess;
      return download_results((size_t)R * n_sweeps);
    }
    // D = 8 (ShallowCNOT families, direct solver): the whole run in ONE launch as well - a workgroup per restart, a wave per
    // shift (qmps_roto_d8.hip); afterwards one ordinary evaluation of the final parameters, as above
    // (six shifts: every wave evaluates two of them in turn).  A restart occupies a CU for the whole run, so this is the path of
    // the SMALL runs (BASELINE.json configs[3]: 256 restarts): measured against the step-by-step path below, us per update,
    // three shifts: R = 256: 19.9 / 34, 512: 40.8 / 45.4, 1 024: 77 / 63, 21 845: 1 552 / 785; six shifts: R = 128: 40.6 / 35.6, 256: 41.0 / 44.1.
    const bool d8_fused_pays = nsh == 3 ? R <= 512 : (R <= 256 && 6 * R > 1024);
    if (c->D == 8 && (nsh == 3 || nsh == 6) && d8_fused_pays && c->default_solver == QMPS_ENV_DIRECT && (kind == QMPS_ANSATZ_SHALLOW_CNOT || kind == QMPS_ANSATZ_SHALLOW_CNOT3) &&
        n_params <= 64 && documented_switch("QMPS_NO_FUSED_ROTO") == nullptr) {
      qmps::RotoArgs ra;
      memset(&ra, 0, sizeof(ra));
      ra.base = d_base; ra.h = c->d_h; ra.hist = d_hist;
      ra.R = (int)R; ra.P = n_params; ra.n_terms = c->n_terms; ra.n_sweeps = n_sweeps; ra.max_iter = max_iter;
      ra.tol = tol; ra.direct = 1; ra.nsh = nsh;
#ifdef QMPS_D8_PROFILE
      auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      const double h0 = now();
#endif
      HIP_TRY(qmps::launch_rotosolve_fused_d8(kind, ra, c->stream));
#ifdef QMPS_D8_PROFILE
      const double h1 = now();
      HIP_TRY(hipStreamSynchronize(c->stream));
      const double h2 = now();
#endif
      HIP_TRY(qmps::launch_ansatz(c->D, kind, d_base, n_params, c->d_A, R, c->stream));
      c->n_states = R; c->ans_have = false; c->tensors_valid = true;
      if (int e = qmps_energy_launch(c, R, max_iter, tol, c->default_solver)) return e;
      c->have_guess = saved_guess;
      if (int e = download_results((size_t)R * n_sweeps + kHistExtra)) return e;
#ifdef QMPS_D8_PROFILE
      fprintf(stderr, "[d8 profile] launch call %.0f us, kernel until sync %.0f us, final evaluation + downloads %.0f us\n", h1 - h0, h2 - h1, now() - h2);
#endif
      return QMPS_OK;
    }
    // One parameter update = shift build -> ansatz -> environment + energy -> closed-form update.  The
    // parameter index lives in HBM and is advanced by the update kernel, so the sequence is captured ONCE
    // into a hipGraph and replayed n_params x n_sweeps times: the sweep is launch-bound at small R.
    // D = 4 with the direct solver: shift build and ansatz happen INSIDE the energy kernel (evaluation nsh r + k builds
    // the tensor of restart r with shift k on parameter *d_idx straight into LDS): two kernels per parameter update
    const bool fused = c->default_solver == QMPS_ENV_DIRECT && fusable_ansatz(c, kind);
    auto evaluate = [&](int shifts) -> int {      // shifts = nsh: the shifted batch;  0: the R base vectors
      const int64_t n = shifts > 0 ? (int64_t)shifts * R : R;
      if (fused) {
        c->ans_have = true; c->ans_kind = kind; c->ans_P = n_params; c->ans_src = d_base; c->ans_i = d_idx; c->ans_nsh = shifts;
        c->tensors_valid = false;
      } else {
        // shifted tensors straight from the base vectors (the shift build is folded into the ansatz kernel)
        HIP_TRY(qmps::launch_ansatz_shifted(c->D, kind, d_base, n_params, c->d_A, n, shifts, d_idx, c->stream));
        c->ans_have = false; c->tensors_valid = true;
      }
      c->n_states = n;
      return qmps_energy_launch(c, n, max_iter, tol, c->default_solver);
    };
    auto one_update = [&](bool first_of_sweep) -> int {
      if (int e = evaluate(nsh)) return e;
      // the shift-0 row of a sweep's first batch is the evaluation of the vectors the PREVIOUS sweep left: its record
      if (first_of_sweep) HIP_TRY(qmps::launch_roto_record(c->d_E, d_hist, (int)R, c->n_terms, d_idx + 2, nsh, c->stream));
      HIP_TRY(qmps::launch_roto_update(d_base, c->d_E, c->d_status, (int)R, n_params, d_idx, c->n_terms, nsh, c->stream));
      return QMPS_OK;
    };
    // One sweep = n_params updates (the first one also records the previous sweep from its shift-0 rows).  The parameter index and the
    // sweep counter live in HBM and are advanced by the update kernel, so the sweep is captured ONCE into a hipGraph and
    // replayed n_sweeps times (a graph launch costs ~15 us: per update it was a third of the time, per sweep it is noise)
    auto one_sweep = [&]() -> int {
      for (int i = 0; i < n_params; ++i)
        if (int e = one_update(i == 0)) return e;
      return QMPS_OK;
    };
    const bool use_graph = documented_switch("QMPS_NO_GRAPH") == nullptr && n_params <= 256;
    if (use_graph) {
      qmps_ctx::RotoKey key;
      key.R = R; key.kind = kind; key.P = n_params; key.nsh = nsh; key.max_iter = max_iter; key.n_terms = c->n_terms;
      key.solver = c->default_solver; key.handoff = c->handoff; key.tol = tol; key.fused = fused;
      key.base = d_base; key.hist = d_hist; key.params = c->d_params; key.E = c->d_E;
      if (!(c->roto_exec && key == c->roto_key)) {
        if (c->roto_exec) (void)hipGraphExecDestroy(c->roto_exec);
        if (c->roto_graph) (void)hipGraphDestroy(c->roto_graph);
        c->roto_exec = nullptr; c->roto_graph = nullptr;
        c->capturing = true;
        HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        const int e = one_sweep();
        const hipError_t ce = hipStreamEndCapture(c->stream, &c->roto_graph);
        c->capturing = false;
        if (e) return e;
        HIP_TRY(ce);
        HIP_TRY(hipGraphInstantiate(&c->roto_exec, c->roto_graph, nullptr, nullptr, 0));
        c->roto_key = key;
      }
    }
    for (int sw = 0; sw < n_sweeps; ++sw) {
      if (use_graph) HIP_TRY(hipGraphLaunch(c->roto_exec, c->stream));
      else if (int e = one_sweep()) return e;
    }
    // the last sweep's record, and the resident state the call leaves: one evaluation of the final vectors
    if (int e = evaluate(0)) return e;
    HIP_TRY(qmps::launch_roto_record(c->d_E, d_hist, (int)R, c->n_terms, d_idx + 2, 1, c->stream));
    // The context's view of what is resident - a replayed graph runs no host code, so it is stated here, not inherited from
    // the capture: the R final parameter vectors, their energies / statuses / environments
    c->n_states = R;
    c->window = 0;
    c->have_env = true;
    c->partials_B = -1;
    c->acc_pending = false;
    if (fused) {
      // as a qmps_set_states_ansatz of the final parameters would leave it: rows resident in d_params, tensors on demand
      HIP_TRY(hipMemcpyAsync(c->d_params, d_base, (size_t)R * n_params * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      c->ans_have = true; c->ans_kind = kind; c->ans_P = n_params; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
      c->tensors_valid = false;
    } else {
      c->ans_have = false; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
      c->tensors_valid = true;
    }
    c->have_guess = saved_guess;
    return download_results((size_t)R * n_sweeps);
  }();
  c->capturing = false;
  (void)hipStreamSynchronize(c->stream);
  if (c->ans_src != nullptr) {     // an error left the context pointing at the run's own buffers
    c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0; c->ans_have = false; c->tensors_valid = true; c->n_states = 0;
  }
  if (rc != QMPS_OK && c->roto_exec) {     // do not trust a sweep captured by a failed run
    (void)hipGraphExecDestroy(c->roto_exec);
    if (c->roto_graph) (void)hipGraphDestroy(c->roto_graph);
    c->roto_exec = nullptr; c->roto_graph = nullptr;
  }
  return rc;
}
}  // namespace

int qmps_get_states(qmps_ctx* c, int64_t B, double* A) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!A) return fail(QMPS_ERR_ARG, "null A");
  if (B > c->n_states) return fail(QMPS_ERR_STATE, "only %lld states are resident", (long long)c->n_states);
  if (int rc = ensure_tensors(c)) return rc;
  HIP_TRY(hipMemcpyAsync(A, c->d_A, (size_t)B * tensor_bytes(c), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_set_hamiltonian(qmps_ctx* c, int n_terms, const double* h) {
  if (int rc = bind(c)) return rc;
  if (n_terms < 1 || n_terms > kMaxTerms) return fail(QMPS_ERR_ARG, "n_terms=%d outside [1,%d]", n_terms, kMaxTerms);
  if (!h) return fail(QMPS_ERR_ARG, "null h");
  if (int rc = ensure_E(c, n_terms)) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_h, h, (size_t)n_terms * 256, hipMemcpyHostToDevice, c->stream));
  if (!c->defer_sync) HIP_TRY(hipStreamSynchronize(c->stream));
  c->h_fro = 0.0;
  for (int t = 0; t < n_terms; ++t) {
    double f = 0.0;
    for (int i = 0; i < 32; ++i) f += h[32 * t + i] * h[32 * t + i];
    f = sqrt(f);
    if (f > c->h_fro) c->h_fro = f;
  }
  c->n_terms = n_terms;
  return QMPS_OK;
}

int qmps_set_env_guess(qmps_ctx* c, int64_t B, const double* r0) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  c->window = 0;
  if (!r0) {
    c->have_guess = false;
    return QMPS_OK;
  }
  HIP_TRY(hipMemcpyAsync(c->d_r, r0, (size_t)B * env_bytes(c), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->have_guess = true;
  c->have_env = true;
  c->have_overlap_x = false;
  return QMPS_OK;
}

int qmps_energy_launch(qmps_ctx* c, int64_t B, int max_iter, double tol, int flags) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "qmps_set_hamiltonian has not been called");
  if (max_iter < 1) return fail(QMPS_ERR_ARG, "max_iter must be >= 1");
  if (!(tol > 0.0)) return fail(QMPS_ERR_ARG, "tol must be > 0");
  int solver = flags & 0xff;
  if (solver != QMPS_ENV_POWER && solver != QMPS_ENV_POWER_SQUARING && solver != QMPS_ENV_DIRECT)
    return fail(QMPS_ERR_ARG, "unknown environment solver %d", solver);
  if ((flags & ~0xff) & ~(QMPS_FLAG_NO_ENV_OUT | QMPS_FLAG_ACCUMULATE_COST | QMPS_FLAG_WARM_RESIDENT)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags & ~0xff);
  const bool warm_resident = (flags & QMPS_FLAG_WARM_RESIDENT) != 0;
  if (warm_resident && !c->have_env) return fail(QMPS_ERR_STATE, "QMPS_FLAG_WARM_RESIDENT: no resident environments (run a launch that stores them, or qmps_set_env_guess)");
  const bool direct = solver == QMPS_ENV_DIRECT && c->D == 4;
  if ((flags & QMPS_FLAG_NO_ENV_OUT) && !direct) return fail(QMPS_ERR_ARG, "QMPS_FLAG_NO_ENV_OUT needs QMPS_ENV_DIRECT at D = 4");
  const bool accumulate = (flags & QMPS_FLAG_ACCUMULATE_COST) != 0;
  if (accumulate && c->D == 4 && !direct && solver == QMPS_ENV_POWER_SQUARING)
    return fail(QMPS_ERR_ARG, "QMPS_FLAG_ACCUMULATE_COST: at D = 4 use QMPS_ENV_DIRECT or QMPS_ENV_POWER");
  if ((flags & QMPS_FLAG_ACCUMULATE_COST) && c->acc_pending)
    return fail(QMPS_ERR_STATE, "the cost accumulated by the previous launch has not been consumed by qmps_cost_launch");
  if ((flags & QMPS_FLAG_ACCUMULATE_COST) && c->capturing) return fail(QMPS_ERR_STATE, "no cost accumulation inside a graph capture");
  const bool direct8 = solver == QMPS_ENV_DIRECT && c->D == 8;
  const bool direct2 = solver == QMPS_ENV_DIRECT && c->D == 2;     // 4 x 4 solve in the lane, in front of the squaring tail
  if (solver == QMPS_ENV_DIRECT && !direct) solver = QMPS_ENV_POWER_SQUARING;   // D = 2: the lane kernel's squaring path with the 4 x 4 solve in front (direct2); D = 16 iterates (documented)
  c->acc_pending = false;   // whatever an earlier launch accumulated no longer describes the resident energies
  c->have_overlap_x = false;   // d_r is about to hold environments, not overlap fixed points
  c->grad_warm_T = 0;
  const bool fused = direct && c->ans_have && fusable_ansatz(c, c->ans_kind);
  if (!fused)
    if (int rc = ensure_tensors(c)) return rc;
  qmps::LaneArgs a = make_args(c, B, max_iter, tol, true);
  if (warm_resident) a.r_in = win_r(c);
  if (direct8) {
    // D = 8: the direct solve (one wave per evaluation) hands its result to the power iteration of the block kernel: its
    // first step is the acceptance test, its loop the fall-back.  Small batches (all launch latency: BASELINE configs[3]
    // is 96 evaluations per GPU) run both in ONE launch; large ones keep two kernels - the block kernel alone runs four
    // waves per SIMD, the solve two.
    static const int64_t fuse_below = tuning_knob("QMPS_D8_FUSE_BELOW") ? atoll(tuning_knob("QMPS_D8_FUSE_BELOW")) : 4096;   // A/B knob
    if (B <= fuse_below) {
      a.direct = 1;
      a.r_in = nullptr;
    } else {
      HIP_TRY(qmps::launch_env_direct_d8(win_A(c), win_r(c), B, c->stream));
      a.r_in = win_r(c);
    }
  }
  const bool hybrid = solver == QMPS_ENV_POWER_SQUARING && c->D <= 4 && c->handoff < max_iter;
  c->timed = !c->capturing && c->timing_period > 0 && c->launches % c->timing_period == 0;
  const int slot = (int)(c->samples % qmps_ctx::kRing);
  c->partials_B = -1;
  const int lane_waves = (int)((B + 63) / 64);
  if (direct) {
    // D = 4: direct fixed-point solve + acceptance power step + energies in ONE kernel (a DPP quad per evaluation);
    // one read of A, one store of E (and, unless switched off, of r) per evaluation
    // a.r_in (qmps_set_env_guess / QMPS_FLAG_WARM_RESIDENT): evaluations whose guess passes the acceptance test skip the solve
    a.r_out = (flags & QMPS_FLAG_NO_ENV_OUT) ? nullptr : win_r(c);
    if (fused) {
      const double* rows = c->ans_src ? c->ans_src : c->d_params;
      a.ans_params = c->ans_nsh > 0 ? rows : rows + (size_t)c->window * c->ans_P;
      a.ans_P = c->ans_P; a.ans_kind = c->ans_kind; a.ans_nsh = c->ans_nsh; a.ans_i = c->ans_i;
    }
    if (flags & QMPS_FLAG_ACCUMULATE_COST) {
      if (int rc = setup_accumulator(c, a, B, (B + 15) / 16, 16)) return rc;
    } else {
      a.partial = c->d_partial; c->partials_B = B; c->partials_n = (int)((B + 15) / 16);
    }
    c->dominant = "energy_direct_d4_kernel";
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy_direct_d4(a, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
    if (c->timed) c->samples++;
    if (!c->capturing) c->launches++;
    if (a.r_out != nullptr) c->have_env = true;
    else if (a.r_in == nullptr) c->have_env = false;      // (a warm launch that stores nothing leaves the resident guesses in place)
    return QMPS_OK;
  }
  if (c->D == 16 && !documented_switch("QMPS_D16_BLOCK")) {
    // D = 16: power iteration on the matrix cores (one wave per evaluation), then the energy pass
    c->dominant = "energy_mfma_d16_kernel<true>";
    if (accumulate) if (int rc = setup_accumulator(c, a, B, B, 1)) return rc;
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy_mfma(c->D, a, true, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
  } else if (!hybrid) {
    c->dominant = c->D <= 4 ? "energy_lane_kernel<D,true>" : "energy_block_kernel<D,true>";
    if (accumulate) {
      if (int rc = setup_accumulator(c, a, B, c->D <= 4 ? lane_waves : B, c->D <= 4 ? 64 : 1)) return rc;
    } else if (c->D <= 4) { a.partial = c->d_partial; c->partials_B = B; c->partials_n = lane_waves; }
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy(c->D, a, true, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
  } else if (c->D == 2) {
    a.handoff = c->handoff;  // the squaring tail runs in-lane (real 4 x 4 transfer matrix in registers)
    a.hybrid = 1;
    a.direct = direct2 ? 1 : 0;
    a.skip = c->handoff == 0 ? c->skip_rounds : 0;
    if (accumulate) {
      if (int rc = setup_accumulator(c, a, B, lane_waves, 64)) return rc;
    } else { a.partial = c->d_partial; c->partials_B = B; c->partials_n = lane_waves; }
    c->dominant = "energy_lane_kernel<2,true>";
    if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    HIP_TRY(qmps::launch_energy(c->D, a, true, c->stream));
    if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
  } else {
    // D = 4: (1) lane kernel: `handoff` plain steps, slow items -> worklist (skipped when handoff == 0:
    // every item goes straight to the squaring kernel); (2) wave-per-item MFMA squaring over the
    // worklist; (3) energy-only pass over the worklist.  No host round trip: the later kernels read the
    // item count from HBM.
    qmps::SquareArgs q;
    memset(&q, 0, sizeof(q));
    q.A = win_A(c); q.r_out = win_r(c); q.iters = win_iters(c); q.status = win_status(c);
    q.B = B; q.done = c->handoff; q.max_iter = max_iter; q.tol = tol;
    q.skip = c->handoff == 0 ? c->skip_rounds : 0;
    q.period = c->matvec_period;
    qmps::LaneArgs e = make_args(c, B, 1, 1.0, false);
    e.check_pd = 1;
    if (c->handoff > 0) {
      HIP_TRY(hipMemsetAsync(c->d_work_count, 0, sizeof(int32_t), c->stream));
      a.handoff = c->handoff;
      a.hybrid = 1;
      a.work_count = c->d_work_count;
      a.work_idx = c->d_work_idx;
      c->dominant = "energy_lane_kernel<4,true>";
      if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
      HIP_TRY(qmps::launch_energy(c->D, a, true, c->stream));
      if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
      q.r_in = win_r(c);
      q.work_count = c->d_work_count;
      q.work_idx = c->d_work_idx;
      e.idx_list = c->d_work_idx;
      e.idx_count = c->d_work_count;
    } else {
      q.r_in = c->have_guess ? win_r(c) : nullptr;
      e.partial = c->d_partial; c->partials_B = B; c->partials_n = lane_waves;   // the energy pass covers every item
    }
    // grid-stride workgroups of 4 waves: whole generations of the resident capacity (5 workgroups per CU), at most three
    // (measured at B = 65536 with settled clocks: 1280 / 2560 / 3840 / 5120 workgroups -> 0.0876 / 0.0870 / 0.0859 / 0.0875 ms;
    // 2048 and 3072, which end in a partial generation, 0.0900 and 0.0878)
    int grid = (int)((B + 15) / 16);
    const int generation = c->n_cus * 5;
    if (grid > generation) {
      grid = (grid / generation) * generation;
      if (grid > 3 * generation) grid = 3 * generation;
    }
    if (const char* e = tuning_knob("QMPS_SQ_GRID")) grid = atoi(e) < grid ? atoi(e) : grid;   // tuning knob
    if (grid < 1) grid = 1;
    if (c->handoff == 0) {
      c->dominant = "env_square_d4_kernel";
      if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
    }
    HIP_TRY(qmps::launch_square_tail(c->D, q, grid, c->stream));
    if (c->handoff == 0) if (c->timed) HIP_TRY(hipEventRecord(c->kev1[slot], c->stream));
    // (with settled clocks the step is 0.9 % shorter with the pair kernel: 0.1112 against 0.1122 ms at B = 65536;
    // QMPS_LANE_IN_STEP keeps the one-lane pass)
    if (e.idx_list == nullptr && !c->no_pair && c->pair_in_step) {
      if (e.partial != nullptr) c->partials_n = (int)((B + 31) / 32);      // two lanes per evaluation: one partial per 32 items
      HIP_TRY(qmps::launch_energy_pair_d4(e, c->stream));
    } else {
      HIP_TRY(qmps::launch_energy(c->D, e, false, c->stream));
    }
  }
  if (c->timed) c->samples++;
  if (!c->capturing) c->launches++;
  c->have_env = true;
  return QMPS_OK;
}

int qmps_set_kernel_timing_period(qmps_ctx* c, int period) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (period < 0) return fail(QMPS_ERR_ARG, "period must be >= 0");
  c->timing_period = period;
  c->samples = 0;          // earlier samples belong to another schedule
  return QMPS_OK;
}

int qmps_set_handoff(qmps_ctx* c, int handoff) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (handoff < 0) return fail(QMPS_ERR_ARG, "handoff must be >= 0");
  c->handoff = handoff;
  return QMPS_OK;
}

int qmps_set_default_solver(qmps_ctx* c, int solver) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (solver != QMPS_ENV_POWER && solver != QMPS_ENV_POWER_SQUARING && solver != QMPS_ENV_DIRECT)
    return fail(QMPS_ERR_ARG, "unknown solver %d", solver);
  c->default_solver = solver;
  return QMPS_OK;
}

int qmps_get_handoff(qmps_ctx* c, int* handoff) {
  if (!c || !handoff) return fail(QMPS_ERR_ARG, "null argument");
  *handoff = c->handoff;
  return QMPS_OK;
}

int qmps_get_squaring_schedule(qmps_ctx* c, int* skip_rounds, int* matvec_period) {
  if (!c || !skip_rounds || !matvec_period) return fail(QMPS_ERR_ARG, "null argument");
  *skip_rounds = c->skip_rounds;
  *matvec_period = c->D == 4 ? c->matvec_period : 0;
  return QMPS_OK;
}

int qmps_energy_only_launch(qmps_ctx* c, int64_t B) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "qmps_set_hamiltonian has not been called");
  if (!c->have_env) return fail(QMPS_ERR_STATE, "no resident environment: run qmps_energy_launch or qmps_set_env_guess first");
  if (int rc = ensure_tensors(c)) return rc;
  qmps::LaneArgs a = make_args(c, B, 1, 1.0, false);
  c->partials_B = -1;
  if (c->D == 16 && !documented_switch("QMPS_D16_BLOCK"))
    HIP_TRY(qmps::launch_energy_mfma(c->D, a, false, c->stream));
  else if (c->D == 4 && tuning_knob("QMPS_ENERGY_PAIR") == nullptr)
    HIP_TRY(qmps::launch_energy_only_d4(a, c->stream));     // quad layout, 4+ waves per SIMD (round 1: two lanes per evaluation)
  else if (c->D == 4 && !c->no_pair)
    HIP_TRY(qmps::launch_energy_pair_d4(a, c->stream));
  else
    HIP_TRY(qmps::launch_energy(c->D, a, false, c->stream));
  return QMPS_OK;
}

static int sum_on_device(qmps_ctx* c, int64_t B) {
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  c->partials_B = -1;   // d_partial is about to be overwritten by the generic two-pass reduction
  HIP_TRY(qmps::launch_sum(win_E(c), B, c->n_terms, c->d_partial, kSumBlocks, c->d_cost, c->stream));
  return QMPS_OK;
}

int qmps_sum_energies(qmps_ctx* c, int64_t B, double* cost) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!cost) return fail(QMPS_ERR_ARG, "null cost");
  if (int rc = sum_on_device(c, B)) return rc;
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost, c->n_terms * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(cost, c->h_cost, c->n_terms * sizeof(double));
  return QMPS_OK;
}

int qmps_get_energies(qmps_ctx* c, int64_t B, double* E, int32_t* iters, int32_t* status) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  if (E) HIP_TRY(hipMemcpyAsync(E, win_E(c), (size_t)B * c->n_terms * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (iters) HIP_TRY(hipMemcpyAsync(iters, win_iters(c), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (status) HIP_TRY(hipMemcpyAsync(status, win_status(c), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_get_status(qmps_ctx* c, int64_t B, int32_t* status) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!status) return fail(QMPS_ERR_ARG, "null status");
  HIP_TRY(hipMemcpyAsync(status, win_status(c), (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_get_env(qmps_ctx* c, int64_t B, double* r) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!r) return fail(QMPS_ERR_ARG, "null r");
  if (!c->have_env) return fail(QMPS_ERR_STATE, "no resident environment");
  HIP_TRY(hipMemcpyAsync(r, win_r(c), (size_t)B * env_bytes(c), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_get_rdm(qmps_ctx* c, int64_t B, double* rho) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!rho) return fail(QMPS_ERR_ARG, "null rho");
  if (!c->have_env) return fail(QMPS_ERR_STATE, "no resident environment");
  if (!c->d_rho) HIP_TRY(hipMalloc(&c->d_rho, (size_t)c->max_batch * 256));
  // recompute from the resident (A, r): the energy-only kernel writes rho when asked to
  c->want_rho = true;
  int rc = qmps_energy_only_launch(c, B);
  c->want_rho = false;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(rho, (char*)c->d_rho + (size_t)c->window * 256, (size_t)B * 256, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_energy_batch(qmps_ctx* c, int64_t B, const double* states, int kind, const double* h, int n_terms,
                      const double* r0, int max_iter, double tol, double* E_out, int32_t* iters_out,
                      int32_t* status_out) {
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  if (int rc = qmps_set_states(c, B, states, kind)) return rc;
  if (int rc = qmps_set_hamiltonian(c, n_terms, h)) return rc;
  if (int rc = qmps_set_env_guess(c, B, r0)) return rc;
  if (int rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver)) return rc;
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}

int qmps_energy_batch_ansatz(qmps_ctx* c, int64_t B, int ansatz_kind, int n_params, const double* params, const double* h,
                             int n_terms, int max_iter, double tol, double* E_out, int32_t* iters_out, int32_t* status_out) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  // the host buffers stay the caller's until this function returns: the copies in may stay in flight until the ONE
  // synchronisation of the read-back (a scalar objective call is all latency: three round trips -> one)
  c->defer_sync = true;
  int rc = qmps_set_states_ansatz(c, B, ansatz_kind, n_params, params);
  if (!rc) rc = qmps_set_hamiltonian(c, n_terms, h);
  if (!rc) rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver | ((c->D == 4 && c->default_solver == QMPS_ENV_DIRECT) ? QMPS_FLAG_NO_ENV_OUT : 0));
  c->defer_sync = false;
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}

int qmps_env_batch(qmps_ctx* c, int64_t B, const double* states, int kind, const double* r0, int max_iter, double tol,
                   double* r_out, int32_t* iters_out, int32_t* status_out) {
  if (!r_out) return fail(QMPS_ERR_ARG, "null r_out");
  if (int rc = qmps_set_states(c, B, states, kind)) return rc;
  if (c->n_terms < 1) {
    // the solve kernel always evaluates at least one Hamiltonian term; use h = 0
    double zero[32];
    memset(zero, 0, sizeof(zero));
    if (int rc = qmps_set_hamiltonian(c, 1, zero)) return rc;
  }
  if (int rc = qmps_set_env_guess(c, B, r0)) return rc;
  if (int rc = qmps_energy_launch(c, B, max_iter, tol, c->default_solver)) return rc;
  if (int rc = qmps_get_energies(c, B, nullptr, iters_out, status_out)) return rc;
  return qmps_get_env(c, B, r_out);
}

int qmps_cell2_energy_batch(qmps_ctx* c, int64_t B, const double* U1, const double* U2, const double* h, int n_terms,
                             int max_iter, double tol, double* E_out, int32_t* iters_out, int32_t* status_out) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (c->D != 2) return fail(QMPS_ERR_ARG, "the two-site unit cell path is D = 2 only (qmps/ground_state.py:276)");
  c->window = 0;      // a one-shot call: results at the start of the buffers, like the qmps_set_* calls
  if ((!U1 || !U2) && B > 0) return fail(QMPS_ERR_ARG, "null unitaries");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  if (max_iter < 1 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_iter / tol");
  if (int rc = qmps_set_hamiltonian(c, n_terms, h)) return rc;
  const size_t ub = 2 * tensor_bytes(c);
  if (!c->d_U) HIP_TRY(hipMalloc(&c->d_U, (size_t)c->max_batch * ub));
  if (!c->d_U2) HIP_TRY(hipMalloc(&c->d_U2, (size_t)c->max_batch * ub));
  HIP_TRY(hipMemcpyAsync(c->d_U, U1, (size_t)B * ub, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_U2, U2, (size_t)B * ub, hipMemcpyHostToDevice, c->stream));
  qmps::Cell2Args a;
  a.U1 = c->d_U; a.U2 = c->d_U2; a.h = c->d_h; a.E = c->d_E; a.E12 = nullptr;
  a.iters = c->d_iters; a.status = c->d_status; a.B = B; a.n_terms = n_terms; a.max_iter = max_iter; a.tol = tol;
  c->partials_B = -1;
  HIP_TRY(qmps::launch_cell2(c->D, a, c->stream));
  c->n_states = 0;  // the resident single-site states (if any) are no longer what d_E refers to
  c->have_env = false;
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}

int qmps_cell2_energy_batch_su(qmps_ctx* c, int64_t B, const double* params, const double* h, int n_terms, int max_iter, double tol,
                               double* E_out, int32_t* iters_out, int32_t* status_out) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (c->D != 2) return fail(QMPS_ERR_ARG, "the two-site unit cell path is D = 2 only (qmps/ground_state.py:276)");
  c->window = 0;
  if (!params && B > 0) return fail(QMPS_ERR_ARG, "null params");
  if (!E_out) return fail(QMPS_ERR_ARG, "null E_out");
  if (max_iter < 1 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_iter / tol");
  if (int rc = qmps_set_hamiltonian(c, n_terms, h)) return rc;
  const size_t ub = 2 * tensor_bytes(c);
  if (!c->d_U) HIP_TRY(hipMalloc(&c->d_U, (size_t)c->max_batch * ub));
  if (!c->d_U2) HIP_TRY(hipMalloc(&c->d_U2, (size_t)c->max_batch * ub));
  if (int rc = ensure_scratch(c, (size_t)B * 30 * sizeof(double) + 256)) return rc;
  double* d_p = (double*)c->d_scratch;
  HIP_TRY(hipMemcpyAsync(d_p, params, (size_t)B * 30 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  // U1 = U4(p[:15]), U2 = U4(p[15:])  (qmps/ground_state.py:300-301), both built on the device
  HIP_TRY(qmps::launch_su_exp(4, d_p, B, 30, c->d_U, 0, c->stream));
  HIP_TRY(qmps::launch_su_exp(4, d_p + 15, B, 30, c->d_U2, 0, c->stream));
  qmps::Cell2Args a;
  a.U1 = c->d_U; a.U2 = c->d_U2; a.h = c->d_h; a.E = c->d_E; a.E12 = nullptr;
  a.iters = c->d_iters; a.status = c->d_status; a.B = B; a.n_terms = n_terms; a.max_iter = max_iter; a.tol = tol;
  c->partials_B = -1;
  HIP_TRY(qmps::launch_cell2(c->D, a, c->stream));
  c->n_states = 0;
  c->have_env = false;
  return qmps_get_energies(c, B, E_out, iters_out, status_out);
}

int qmps_kernel_time(qmps_ctx* c, int n_last, float* avg_ms, char* name, int name_len) {
  if (int rc = bind(c)) return rc;
  if (!avg_ms || n_last < 1) return fail(QMPS_ERR_ARG, "bad arguments");
  if (c->samples < 1) return fail(QMPS_ERR_STATE, "no timed energy launch yet (qmps_set_kernel_timing_period)");
  HIP_TRY(hipStreamSynchronize(c->stream));
  // the timed launches among the last n_last ones
  int64_t n = c->timing_period > 0 ? (n_last + c->timing_period - 1) / c->timing_period : 1;
  if (n < 1) n = 1;
  if (n > c->samples) n = c->samples;
  if (n > qmps_ctx::kRing) n = qmps_ctx::kRing;
  double sum = 0.0;
  for (int64_t k = 0; k < n; ++k) {
    const int slot = (int)((c->samples - 1 - k) % qmps_ctx::kRing);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->kev0[slot], c->kev1[slot]));
    sum += ms;
  }
  *avg_ms = (float)(sum / (double)n);
  if (name && name_len > 0) snprintf(name, name_len, "%s", c->dominant);
  return QMPS_OK;
}

namespace {
int ensure_refs(qmps_ctx* c, int64_t n_ref) {
  if (n_ref > c->ref_cap) {
    if (c->d_ref) HIP_TRY(hipFree(c->d_ref));
    c->d_ref = nullptr;
    c->ref_cap = 0;
    HIP_TRY(hipMalloc(&c->d_ref, (size_t)n_ref * tensor_bytes(c)));
    c->ref_cap = n_ref;
  }
  return QMPS_OK;
}
int set_ww(qmps_ctx* c, const double* WW) {
  if (!c->d_ww) HIP_TRY(hipMalloc(&c->d_ww, 256));
  HIP_TRY(hipMemcpyAsync(c->d_ww, WW, 256, hipMemcpyHostToDevice, c->stream));
  return QMPS_OK;
}
int ensure_overlap_outputs(qmps_ctx* c) {
  if (!c->d_eta) HIP_TRY(hipMalloc(&c->d_eta, (size_t)c->max_batch * 16));
  if (!c->d_f) HIP_TRY(hipMalloc((void**)&c->d_f, (size_t)c->max_batch * sizeof(double)));
  if (!c->d_ostats) {
    HIP_TRY(hipMalloc((void**)&c->d_ostats, (size_t)qmps::kOverlapStatShards * 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(c->d_ostats, 0, (size_t)qmps::kOverlapStatShards * 4 * sizeof(unsigned long long), c->stream));
  }
  return QMPS_OK;
}
// D = 16 batches above 2 048 evaluations: the four-waves-per-evaluation kernel with a work queue (zeroed here, on the stream)
int arm_queue(qmps_ctx* c, qmps::OverlapArgs& a, int which) {
  a.queue = nullptr;
  if (c->D != 16 || a.B <= 2048 || documented_switch("QMPS_D16_ONE_WAVE") != nullptr || documented_switch("QMPS_D16_BLOCK") != nullptr) return QMPS_OK;
  if (!c->d_queue) HIP_TRY(hipMalloc((void**)&c->d_queue, 2 * sizeof(int)));
  HIP_TRY(hipMemsetAsync(c->d_queue + which, 0, sizeof(int), c->stream));
  a.queue = c->d_queue + which;
  return QMPS_OK;
}

// the kernel the overlap launch of this context runs (name for qmps_kernel_time) and whether it counts squarings
bool overlap_squares(const qmps_ctx* c) { return c->D == 2 || (c->D == 4 && !documented_switch("QMPS_OVERLAP_POWER")); }
int launch_overlap_kernels(qmps_ctx* c, const qmps::OverlapArgs& a_in) {
  qmps::OverlapArgs a = a_in;
  if (int rc = arm_queue(c, a, 0)) return rc;
  const bool squaring = overlap_squares(c);
  c->dominant = c->D == 2 ? "overlap_lane_kernel" : (c->D == 4 && squaring ? "overlap_square_d4_kernel" :
                (c->D == 16 && !documented_switch("QMPS_D16_BLOCK") ? "overlap_mfma_d16_kernel" : "overlap_block_kernel<D>"));
  c->timed = !c->capturing && c->timing_period > 0 && c->launches % c->timing_period == 0;
  const int slot = (int)(c->samples % qmps_ctx::kRing);
  if (c->timed) HIP_TRY(hipEventRecord(c->kev0[slot], c->stream));
  a.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
  if (c->D == 2) HIP_TRY(qmps::launch_overlap(a, c->stream));
  else HIP_TRY(qmps::launch_overlap_d(c->D, a, c->D == 4 ? squaring : documented_switch("QMPS_D16_BLOCK") == nullptr, c->stream));
  if (c->timed) { HIP_TRY(hipEventRecord(c->kev1[slot], c->stream)); c->samples++; }
  if (!c->capturing) c->launches++;
  return QMPS_OK;
}
}  // namespace

int qmps_overlap_set(qmps_ctx* c, int64_t n_ref, const double* A, const double* WW) {
  if (int rc = bind(c)) return rc;
  if (!A || !WW) return fail(QMPS_ERR_ARG, "null argument");
  if (n_ref < 1 || n_ref > c->max_batch) return fail(QMPS_ERR_ARG, "n_ref=%lld outside [1, max_batch]", (long long)n_ref);
  // the reference state(s) live in their own buffer: nothing else in the context writes it
  if (int rc = ensure_refs(c, n_ref)) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_ref, A, (size_t)n_ref * tensor_bytes(c), hipMemcpyHostToDevice, c->stream));
  if (int rc = set_ww(c, WW)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->overlap_refs = n_ref;
  c->overlap_group = 0;
  return QMPS_OK;
}

int qmps_overlap_set_refs_ansatz(qmps_ctx* c, int64_t n_ref, int kind, int n_params, const double* params, const double* WW) {
  if (int rc = bind(c)) return rc;
  if (!params || !WW) return fail(QMPS_ERR_ARG, "null argument");
  if (n_ref < 1 || n_ref > c->max_batch) return fail(QMPS_ERR_ARG, "n_ref=%lld outside [1, max_batch]", (long long)n_ref);
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  if (int rc = ensure_refs(c, n_ref)) return rc;
  // parameter rows through the scratch arena, tensors built on the device straight into the reference buffer
  if (int rc = ensure_scratch(c, (size_t)n_ref * n_params * sizeof(double))) return rc;
  HIP_TRY(hipMemcpyAsync(c->d_scratch, params, (size_t)n_ref * n_params * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(qmps::launch_ansatz(c->D, kind, (const double*)c->d_scratch, n_params, c->d_ref, n_ref, c->stream));
  if (int rc = set_ww(c, WW)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->overlap_refs = n_ref;
  c->overlap_group = 0;
  return QMPS_OK;
}

int qmps_overlap_set_group(qmps_ctx* c, int64_t group) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (group < 0) return fail(QMPS_ERR_ARG, "group must be >= 0");
  c->overlap_group = group;
  return QMPS_OK;
}

int qmps_overlap_set_active(qmps_ctx* c, int64_t n, const unsigned char* active) {
  if (int rc = bind(c)) return rc;
  if (n < 0 || n > c->max_batch) return fail(QMPS_ERR_ARG, "n=%lld outside [0, max_batch]", (long long)n);
  if (n == 0 || !active) {
    c->active_n = 0;
    return QMPS_OK;
  }
  if (!c->d_active) HIP_TRY(hipMalloc((void**)&c->d_active, (size_t)c->max_batch));
  if (int rc = ensure_pinned(c, (16u << 20))) return rc;
  // through the pinned staging buffer (its last MiB: the parameter / result regions may be in use by the same driver)
  unsigned char* stage = (unsigned char*)c->h_pin + (15u << 20);
  if ((size_t)n > (1u << 20)) return fail(QMPS_ERR_ARG, "mask longer than 2^20 entries");
  memcpy(stage, active, (size_t)n);
  memset(stage + n, 1, (size_t)((8 - n % 8) % 8));
  HIP_TRY(qmps::launch_stage_copy(stage, c->d_active, (n + 7) / 8, c->stream));
  c->active_n = n;
  return QMPS_OK;
}

int qmps_overlap_launch(qmps_ctx* c, int64_t B, int max_rounds, double tol, int flags) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->window + B > c->n_states) return fail(QMPS_ERR_STATE, "window [%lld, %lld) but only %lld states are resident", (long long)c->window, (long long)(c->window + B), (long long)c->n_states);
  if (c->overlap_refs < 1) return fail(QMPS_ERR_STATE, "qmps_overlap_set has not been called");
  if (flags & ~(QMPS_OVERLAP_WANT_R | QMPS_OVERLAP_WARM)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  const bool want_r = (flags & QMPS_OVERLAP_WANT_R) != 0, warm = (flags & QMPS_OVERLAP_WARM) != 0;
  if (warm && !want_r) return fail(QMPS_ERR_ARG, "QMPS_OVERLAP_WARM needs QMPS_OVERLAP_WANT_R (the fixed points stay resident for the next launch)");
  if (warm && !c->have_overlap_x) return fail(QMPS_ERR_STATE, "QMPS_OVERLAP_WARM: no resident fixed points (run a launch with QMPS_OVERLAP_WANT_R first)");
  const int64_t group = c->overlap_group;
  if (group > 0) {
    if (c->overlap_refs * group < c->window + B) return fail(QMPS_ERR_STATE, "%lld reference tensors x group %lld for window end %lld", (long long)c->overlap_refs, (long long)group, (long long)(c->window + B));
  } else if (c->overlap_refs != 1 && c->overlap_refs < c->window + B) {
    return fail(QMPS_ERR_STATE, "%lld reference tensors for window end %lld", (long long)c->overlap_refs, (long long)(c->window + B));
  }
  const bool squaring = overlap_squares(c);   // these square the matrix of the map: rounds, not steps
  const int cap = squaring ? 60 : (1 << 24);
  if (max_rounds < 1 || max_rounds > cap || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol (D = %d: max_rounds in [1, %d])", c->D, cap);
  if (int rc = ensure_overlap_outputs(c)) return rc;
  qmps::OverlapArgs a;
  memset(&a, 0, sizeof(a));
  const bool shared = group == 0 && c->overlap_refs == 1;
  // (the window displaces the candidates; with a group the references are addressed by the candidate's GLOBAL index)
  a.A = (shared || group > 0) ? (char*)c->d_ref + (group > 0 ? (size_t)(c->window / group) * tensor_bytes(c) : 0)
                              : (char*)c->d_ref + (size_t)c->window * tensor_bytes(c);
  if (group > 0 && c->window % group) return fail(QMPS_ERR_ARG, "with a candidate group the window must start at a multiple of it");
  if (int rc = ensure_tensors(c)) return rc;
  a.Bt = win_A(c);
  a.WW = c->d_ww;
  a.eta = (char*)c->d_eta + (size_t)c->window * 16;
  a.f_out = c->d_f + c->window;
  a.r_out = want_r ? win_r(c) : nullptr;
  a.x_in = warm ? win_r(c) : nullptr;
  a.stats = c->d_ostats;
  a.group = (int)group;
  a.iters = win_iters(c); a.status = win_status(c); a.B = B; a.a_shared = shared ? 1 : 0; a.max_rounds = max_rounds; a.tol = tol;
  if (c->active_n > 0) {           // one-shot mask of qmps_overlap_set_active: one entry per trajectory (candidate group)
    const int64_t need = group > 0 ? (c->window + B + group - 1) / group : c->window + B;
    if (c->active_n < need) return fail(QMPS_ERR_STATE, "qmps_overlap_set_active: %lld entries for %lld trajectories", (long long)c->active_n, (long long)need);
    a.active = c->d_active + (group > 0 ? c->window / group : c->window);
    c->active_n = 0;
  }
  if (int rc = launch_overlap_kernels(c, a)) return rc;
  c->have_env = false;
  c->have_guess = false;
  if (want_r) {                 // (a launch that keeps no fixed points leaves the resident ones alone)
    c->have_overlap_x = true;
    c->grad_warm_T = 0;
  }
  c->acc_pending = false;
  c->partials_B = -1;
  return QMPS_OK;
}

int qmps_overlap_get(qmps_ctx* c, int64_t B, double* eta_out, double* r_out, int32_t* rounds_out, int32_t* status_out) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!eta_out) return fail(QMPS_ERR_ARG, "null eta_out");
  if (!c->d_eta) return fail(QMPS_ERR_STATE, "qmps_overlap_launch has not been called");
  HIP_TRY(hipMemcpyAsync(eta_out, (char*)c->d_eta + (size_t)c->window * 16, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  if (r_out) {
    if (!c->have_overlap_x) return fail(QMPS_ERR_STATE, "the last overlap launch did not keep the fixed points (QMPS_OVERLAP_WANT_R)");
    HIP_TRY(hipMemcpyAsync(r_out, win_r(c), (size_t)B * env_bytes(c), hipMemcpyDeviceToHost, c->stream));
  }
  if (rounds_out) HIP_TRY(hipMemcpyAsync(rounds_out, win_iters(c), (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  if (status_out) HIP_TRY(hipMemcpyAsync(status_out, win_status(c), (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_overlap_get_objective(qmps_ctx* c, int64_t B, double* f_out) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (!f_out) return fail(QMPS_ERR_ARG, "null f_out");
  if (!c->d_f) return fail(QMPS_ERR_STATE, "qmps_overlap_launch has not been called");
  HIP_TRY(hipMemcpyAsync(f_out, c->d_f + c->window, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_overlap_stats(qmps_ctx* c, int64_t* evaluations, int64_t* rounds_sum, int64_t* rounds_max, int64_t* not_converged, int reset) {
  if (int rc = bind(c)) return rc;
  unsigned long long h[4] = {0, 0, 0, 0};
  if (c->d_ostats) {
    std::vector<unsigned long long> sh((size_t)qmps::kOverlapStatShards * 4);
    HIP_TRY(hipMemcpyAsync(sh.data(), c->d_ostats, sh.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (reset) HIP_TRY(hipMemsetAsync(c->d_ostats, 0, sh.size() * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < qmps::kOverlapStatShards; ++k) {
      h[0] += sh[4 * k];
      h[1] += sh[4 * k + 1];
      if (sh[4 * k + 2] > h[2]) h[2] = sh[4 * k + 2];
      h[3] += sh[4 * k + 3];
    }
  }
  if (evaluations) *evaluations = (int64_t)h[0];
  if (rounds_sum) *rounds_sum = (int64_t)h[1];
  if (rounds_max) *rounds_max = (int64_t)h[2];
  if (not_converged) *not_converged = (int64_t)h[3];
  return QMPS_OK;
}

int qmps_overlap_eval_ansatz(qmps_ctx* c, int64_t B, int kind, int n_params, const double* params, int max_rounds, double tol,
                             int flags, double* f_out, int32_t* status_out) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!f_out) return fail(QMPS_ERR_ARG, "null f_out");
  // one round trip: parameters in, ansatz + overlap kernels, objective and status out - ONE synchronisation (the optimiser
  // drivers call this twice per iteration; three separate calls cost three synchronisations and two extra launch gaps)
  c->defer_sync = true;
  int rc = qmps_set_states_ansatz(c, B, kind, n_params, params);
  c->defer_sync = false;
  if (!rc) rc = qmps_overlap_launch(c, B, max_rounds, tol, flags);
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  const size_t fb = (size_t)B * sizeof(double), sb = (size_t)B * sizeof(int32_t);
  if (fb + sb <= (8u << 20) && c->h_pin_bytes >= (16u << 20)) {
    char* out = c->h_pin + (8u << 20);
    HIP_TRY(qmps::launch_stage_copy(c->d_f, out, B, c->stream));
    if (status_out) HIP_TRY(qmps::launch_stage_copy(c->d_status, out + fb, (B + 1) / 2, c->stream));     // (d_status has max_batch >= B + 1 entries or the tail is never read)
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(f_out, out, fb);
    if (status_out) memcpy(status_out, out + fb, sb);
    return QMPS_OK;
  }
  HIP_TRY(hipMemcpyAsync(f_out, c->d_f, fb, hipMemcpyDeviceToHost, c->stream));
  if (status_out) HIP_TRY(hipMemcpyAsync(status_out, c->d_status, sb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_overlap_gradient(qmps_ctx* c, int64_t T, int kind, int n_params, const double* params, double h, int max_rounds, double tol,
                          int flags, double* f_out, double* g_out, int32_t* status_out) {
  if (int rc = bind(c)) return rc;
  if (!params || !f_out || !g_out) return fail(QMPS_ERR_ARG, "null argument");
  if (c->D < 4) return fail(QMPS_ERR_ARG, "qmps_overlap_gradient: D = 4, 8, 16 (at D = 2 evaluate the central-difference neighbours themselves)");
  if (flags & ~(QMPS_OVERLAP_WARM | QMPS_OVERLAP_TWO_SIDED_F)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  if (!(h > 0.0)) return fail(QMPS_ERR_ARG, "h must be > 0");
  const int P = n_params;
  if (T < 1 || T * (1 + 2 * (int64_t)P) > c->max_batch) return fail(QMPS_ERR_ARG, "T (1 + 2 n_params) = %lld evaluations exceed max_batch = %lld", (long long)(T * (1 + 2 * (int64_t)P)), (long long)c->max_batch);
  if (c->overlap_refs < T) return fail(QMPS_ERR_STATE, "%lld reference tensors resident, %lld trajectories (qmps_overlap_set / qmps_overlap_set_refs_ansatz)", (long long)c->overlap_refs, (long long)T);
  if (max_rounds < 1 || max_rounds > (1 << 24) || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol");
  const bool warm = (flags & QMPS_OVERLAP_WARM) != 0;
  if (warm && c->grad_warm_T != T) return fail(QMPS_ERR_STATE, "QMPS_OVERLAP_WARM: the resident fixed points belong to %lld trajectories, not %lld", (long long)c->grad_warm_T, (long long)T);
  if (int rc = ensure_overlap_outputs(c)) return rc;
  if (!c->d_y) HIP_TRY(hipMalloc(&c->d_y, (size_t)c->max_batch * env_bytes(c)));
  const size_t nD = (size_t)c->D * c->D;
  if (int rc = ensure_scratch(c, (size_t)T * (4 * nD + 1) * 16 + 256)) return rc;
  {      // pinned staging for the results, sized BEFORE anything is in flight through it
    const size_t need = (size_t)T * (1 + 2 * P) * sizeof(double) + (size_t)2 * T * sizeof(int32_t) + (8u << 20);
    if (int e = ensure_pinned(c, need > (16u << 20) ? need : (16u << 20))) return e;
  }
  // the iterates: parameters -> tensors in d_A[0, T)
  c->defer_sync = true;
  int rc = qmps_set_states_ansatz(c, T, kind, P, params);
  c->defer_sync = false;
  if (!rc) rc = ensure_tensors(c);
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  const bool squaring = overlap_squares(c);       // D = 4: the right fixed point comes from the squaring kernel (largest column)
  qmps::OverlapArgs a;
  memset(&a, 0, sizeof(a));
  a.A = c->d_ref; a.Bt = c->d_A; a.WW = c->d_ww; a.eta = c->d_eta; a.f_out = c->d_f; a.r_out = c->d_r;
  a.x_in = (warm && !squaring) ? c->d_r : nullptr;
  a.stats = c->d_ostats; a.iters = c->d_iters; a.status = c->d_status; a.B = T; a.a_shared = 0;
  a.max_rounds = squaring && max_rounds > 60 ? 60 : max_rounds; a.tol = tol;
  const unsigned char* mask = nullptr;
  if (c->active_n > 0) {
    if (c->active_n < T) return fail(QMPS_ERR_STATE, "qmps_overlap_set_active: %lld entries for %lld trajectories", (long long)c->active_n, (long long)T);
    mask = c->d_active;
    c->active_n = 0;
  }
  a.active = mask;
  // HIP events around the WHOLE gradient evaluation (right solve, left solve, neighbour tensors, G, probes): qmps_kernel_time
  c->dominant = c->D == 16 ? "overlap_mfma_d16_kernel + adjoint + neighbour probes" : "overlap solve + adjoint + neighbour probes";
  c->timed = !c->capturing && c->timing_period > 0 && c->launches % c->timing_period == 0;
  const int tslot = (int)(c->samples % qmps_ctx::kRing);
  if (c->timed) HIP_TRY(hipEventRecord(c->kev0[tslot], c->stream));
  // the 2 P central-difference neighbours of every iterate (evaluated below to second order in h from (y, r)): their tensors
  // need the parameters only, so they are built on a second stream BESIDE the eigen-solves (at small T a gradient batch is the
  // latency of its slowest solve; the neighbour tensors were a fifth of it in front of the probes)
  if (!c->aux_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
  }
  // (beyond ~1 000 iterates the solves fill the chip by themselves: T = 2 048 measured 5 % slower with the second stream)
  const bool beside = T <= 1024;
  if (beside) {
    HIP_TRY(hipEventRecord(c->aux_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->aux_fork, 0));
    HIP_TRY(qmps::launch_ansatz_fd(c->D, kind, c->d_params, P, (char*)c->d_A + (size_t)T * tensor_bytes(c), T, h, c->aux_stream));
    HIP_TRY(hipEventRecord(c->aux_join, c->aux_stream));
  }
  // the left fixed points: power method on the adjoint map; results behind the iterates' (eta, rounds, status at [T, 2T))
  qmps::OverlapArgs l = a;
  l.adjoint = 1; l.eta = (char*)c->d_eta + (size_t)T * 16; l.f_out = nullptr; l.r_out = c->d_y; l.x_in = warm ? c->d_y : nullptr;
  l.iters = c->d_iters + T; l.status = c->d_status + T; l.max_rounds = max_rounds;
  if (c->D == 16 && documented_switch("QMPS_D16_BLOCK") == nullptr && documented_switch("QMPS_D16_ONE_WAVE") == nullptr) {
    // both solves in ONE launch: the iteration chains are latency-bound, so the left solve rides along (more than 2 048
    // iterates: the workgroups draw them from two queues)
    a.no_deflation = l.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    if (int e = arm_queue(c, a, 0)) return e;
    if (int e = arm_queue(c, l, 1)) return e;
    HIP_TRY(qmps::launch_overlap_pair_d16(a, l, c->stream));
  } else {
    a.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    HIP_TRY(qmps::launch_overlap_d(c->D, a, c->D == 4 ? squaring : documented_switch("QMPS_D16_BLOCK") == nullptr, c->stream));
    // (D = 4: the squaring kernel again - the left fixed point is the largest row of the squared map, whatever the spectral gap)
    if (c->D == 4 && squaring) l.max_rounds = a.max_rounds;
    l.no_deflation = documented_switch("QMPS_NO_DEFLATION") != nullptr ? 1 : 0;
    HIP_TRY(qmps::launch_overlap_d(c->D, l, c->D == 4 ? squaring : (c->D == 16 && documented_switch("QMPS_D16_BLOCK") == nullptr), c->stream));
  }
  if (beside) HIP_TRY(hipStreamWaitEvent(c->stream, c->aux_join, 0));
  else HIP_TRY(qmps::launch_ansatz_fd(c->D, kind, c->d_params, P, (char*)c->d_A + (size_t)T * tensor_bytes(c), T, h, c->stream));
  qmps::OverlapGradArgs g;
  memset(&g, 0, sizeof(g));
  g.A = c->d_ref; g.WW = c->d_ww; g.r = c->d_r; g.y = c->d_y; g.G = c->d_scratch; g.yr = (char*)c->d_scratch + (size_t)T * 4 * nD * 16;
  g.Bt = (char*)c->d_A + (size_t)T * tensor_bytes(c); g.f_out = c->d_f + T; g.T = T; g.G2P = 2 * P; g.active = mask;
  if (flags & QMPS_OVERLAP_TWO_SIDED_F) { g.Bc = c->d_A; g.fc_out = c->d_f; }       // (overwrites the right solve's own estimate)
  HIP_TRY(qmps::launch_overlap_grad(c->D, g, c->stream));
  if (c->timed) { HIP_TRY(hipEventRecord(c->kev1[tslot], c->stream)); c->samples++; }
  c->launches++;
  // f of the iterates and of their neighbours are contiguous in d_f: one copy; statuses of both solves: one copy (pinned staging)
  const size_t fbytes = (size_t)T * (1 + 2 * P) * sizeof(double), sbytes = (size_t)2 * T * sizeof(int32_t);
  double* fall = (double*)(c->h_pin + (8u << 20));
  int32_t* st = (int32_t*)(c->h_pin + (8u << 20) + fbytes);
  HIP_TRY(qmps::launch_stage_copy(c->d_f, fall, (int64_t)(fbytes / 8), c->stream));
  HIP_TRY(qmps::launch_stage_copy(c->d_status, st, (int64_t)(sbytes / 8), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(f_out, fall, (size_t)T * sizeof(double));
  const double* fn = fall + T;
  for (int64_t t = 0; t < T; ++t) {
    for (int k = 0; k < P; ++k) g_out[t * P + k] = (fn[(size_t)t * 2 * P + k] - fn[(size_t)t * 2 * P + P + k]) / (2.0 * h);
    if (status_out) status_out[t] = st[t] > st[T + t] ? st[t] : st[T + t];
  }
  c->window = 0;
  c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1;
  c->grad_warm_T = T;
  return QMPS_OK;
}

int qmps_evolve_bfgs(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps, int maxiter,
                     double gtol, double h, double c1, int n_alphas, const double* alphas, int flags, int max_rounds, double tol,
                     double* hinv, double* params_hist, double* f_hist, int32_t* nit_out, double* counters_out) {
  if (int rc = bind(c)) return rc;
  if (!params || !WW || !f_hist || !alphas) return fail(QMPS_ERR_ARG, "null argument");
  if (flags & ~(QMPS_BFGS_CARRY_HESSIAN | QMPS_BFGS_WARM | QMPS_BFGS_TIGHT_GRADIENT)) return fail(QMPS_ERR_ARG, "unknown flag bits 0x%x", flags);
  const int P = n_params, NA = n_alphas;
  if (NA < 1 || NA > 64) return fail(QMPS_ERR_ARG, "n_alphas outside [1, 64]");
  const int64_t G = NA - 1;
  if (T < 1 || T * (1 + 2 * (int64_t)P) > c->max_batch || T * G > c->max_batch)
    return fail(QMPS_ERR_ARG, "T max(2 n_params + 1, n_alphas - 1) = %lld evaluations exceed max_batch = %lld",
                (long long)(T * ((1 + 2 * (int64_t)P) > G ? (1 + 2 * (int64_t)P) : G)), (long long)c->max_batch);
  if (n_steps < 1 || maxiter < 0 || !(gtol > 0.0) || !(h > 0.0)) return fail(QMPS_ERR_ARG, "bad n_steps / maxiter / gtol / h");
  if (int rc = check_ansatz(c, kind, P)) return rc;
  const bool carry = (flags & QMPS_BFGS_CARRY_HESSIAN) != 0;
  bool warm = (flags & QMPS_BFGS_WARM) != 0;
  const bool two_sided = c->D >= 4;       // D = 2: the 2 P + 1 central-difference candidates are eigen-solved themselves (a lane each)
  if (warm && two_sided && c->grad_warm_T != T) return fail(QMPS_ERR_STATE, "QMPS_BFGS_WARM: the resident fixed points belong to %lld trajectories, not %lld", (long long)c->grad_warm_T, (long long)T);
  const bool squaring = overlap_squares(c);
  const int ladder_rounds = squaring ? (max_rounds > 60 ? 60 : max_rounds) : max_rounds;
  const int grad_rounds = max_rounds > 100000 ? max_rounds : 100000;       // (as _GroupedObjective.value_and_grad)
  // objective by the two-sided quotient (error ~ residual^2): the gradient batches' solves stop at 1e-8 (see qmps_hip.h)
  const double grad_tol = (flags & QMPS_BFGS_TIGHT_GRADIENT) ? tol : (tol > 1e-8 ? tol : 1e-8);
  const size_t TP = (size_t)T * P;
  const double nan = __builtin_nan("");
  std::vector<double> X(params, params + TP), Hinv(TP * P), f(T), g(TP), d(TP), slope(T), fs(T), gs(TP), fn(T), gn(TP), Xc(TP), Xn(TP), s(TP), Fc((size_t)T * NA),
      cand, Fl, Hy(P);
  std::vector<int32_t> st(T), stl;
  std::vector<unsigned char> active(T), moved(T), need(T);
  const int saved_period = c->timing_period;
  if (counters_out) c->timing_period = 1;
  double n_grad = 0.0, n_ladder = 0.0, nfev = 0.0, grad_ms = 0.0;
  auto set_identity = [&](int64_t t) {
    double* Ht = &Hinv[(size_t)t * P * P];
    for (int a = 0; a < P; ++a)
      for (int b = 0; b < P; ++b) Ht[a * P + b] = a == b ? 1.0 : 0.0;
  };
  if (carry && warm && hinv) memcpy(Hinv.data(), hinv, TP * P * sizeof(double));
  else for (int64_t t = 0; t < T; ++t) set_identity(t);
  // objective + gradient of a batch of iterates; trajectories with a failed solve come back as NaN (tools.py / new_time_evolve.py)
  std::vector<double> fdc, fdf;
  std::vector<int32_t> fds;
  auto value_and_grad = [&](const double* Z, double* fo, double* go, const unsigned char* mask) -> int {
    if (!two_sided) {
      // tools.batched_fd_gradient: candidate t (2 P + 1) + 0 = the iterate, + 1 + k = +h e_k, + 1 + P + k = -h e_k
      const int64_t G1 = 2 * (int64_t)P + 1;
      fdc.resize((size_t)T * G1 * P);
      fdf.resize((size_t)T * G1);
      fds.resize((size_t)T * G1);
      for (int64_t t = 0; t < T; ++t)
        for (int64_t r = 0; r < G1; ++r)
          for (int k = 0; k < P; ++k)
            fdc[((size_t)t * G1 + r) * P + k] = Z[(size_t)t * P + k] + (r >= 1 && (r - 1) % P == k ? (r <= P ? h : -h) : 0.0);
      if (int e = qmps_overlap_set_group(c, G1)) return e;
      if (mask) { if (int e = qmps_overlap_set_active(c, T, mask)) return e; }
      int e = qmps_overlap_eval_ansatz(c, T * G1, kind, P, fdc.data(), ladder_rounds, tol, 0, fdf.data(), fds.data());
      (void)qmps_overlap_set_group(c, 0);
      if (e) return e;
      for (int64_t t = 0; t < T; ++t) {
        const double* F = &fdf[(size_t)t * G1];
        const int32_t* S = &fds[(size_t)t * G1];
        fo[t] = S[0] == qmps::QMPS_ST_OK ? F[0] : nan;
        for (int k = 0; k < P; ++k)
          go[(size_t)t * P + k] = (S[1 + k] == qmps::QMPS_ST_OK && S[1 + P + k] == qmps::QMPS_ST_OK) ? (F[1 + k] - F[1 + P + k]) / (2.0 * h) : nan;
      }
      n_grad += 1.0;
      nfev += (double)T * (2 * P + 1);
      if (counters_out) {
        float ms = 0.f;
        if (qmps_kernel_time(c, 1, &ms, nullptr, 0) == QMPS_OK) grad_ms += ms;
      }
      return QMPS_OK;
    }
    if (mask) { if (int e = qmps_overlap_set_active(c, T, mask)) return e; }
    if (int e = qmps_overlap_gradient(c, T, kind, P, Z, h, grad_rounds, grad_tol, (warm ? QMPS_OVERLAP_WARM : 0) | QMPS_OVERLAP_TWO_SIDED_F, fo, go, st.data())) return e;
    warm = true;
    for (int64_t t = 0; t < T; ++t)
      if (st[t] != qmps::QMPS_ST_OK) {
        fo[t] = nan;
        for (int k = 0; k < P; ++k) go[(size_t)t * P + k] = nan;
      }
    n_grad += 1.0;
    nfev += (double)T * (2 * P + 1);
    if (counters_out) {
      float ms = 0.f;
      if (qmps_kernel_time(c, 1, &ms, nullptr, 0) == QMPS_OK) grad_ms += ms;
    }
    return QMPS_OK;
  };
  auto gmax_at_least = [&](const double* gt, double bound) {       // np.abs(g).max() >= bound, NaN-propagating: false with any NaN
    double m = 0.0;
    for (int k = 0; k < P; ++k) {
      if (gt[k] != gt[k]) return false;
      const double a = fabs(gt[k]);
      m = a > m ? a : m;
    }
    return m >= bound;
  };
  int rc = QMPS_OK;
  for (int step = 0; step < n_steps && rc == QMPS_OK; ++step) {
    // the step's references: A_t = tensor(current parameters)
    if ((rc = qmps_overlap_set_refs_ansatz(c, T, kind, P, X.data(), WW))) break;
    if (!(carry && (step > 0 || (warm && hinv))))
      for (int64_t t = 0; t < T; ++t) set_identity(t);
    if ((rc = value_and_grad(X.data(), f.data(), g.data(), nullptr))) break;
    bool any_active = false;
    for (int64_t t = 0; t < T; ++t) { active[t] = gmax_at_least(&g[(size_t)t * P], gtol) ? 1 : 0; any_active |= active[t] != 0; }
    int nit = 0;
    while (nit < maxiter && any_active) {
      for (int64_t t = 0; t < T; ++t) {
        const double* Ht = &Hinv[(size_t)t * P * P];
        const double* gt = &g[(size_t)t * P];
        double* dt = &d[(size_t)t * P];
        double sl = 0.0;
        for (int a = 0; a < P; ++a) {
          double acc = 0.0;
          for (int b = 0; b < P; ++b) acc += Ht[a * P + b] * gt[b];
          dt[a] = -acc;
        }
        for (int a = 0; a < P; ++a) sl += gt[a] * dt[a];
        if (!(sl < 0.0)) {                                  // not a descent direction: restart from steepest descent
          set_identity(t);
          sl = 0.0;
          for (int a = 0; a < P; ++a) { dt[a] = -gt[a]; sl -= gt[a] * gt[a]; }
        }
        slope[t] = sl;
        if (!active[t]) for (int a = 0; a < P; ++a) dt[a] = 0.0;
      }
      // the full step with its gradient, straight away
      for (size_t q = 0; q < TP; ++q) Xc[q] = X[q] + alphas[0] * d[q];
      if ((rc = value_and_grad(Xc.data(), fs.data(), gs.data(), active.data()))) break;
      bool all_accept = true;
      for (int64_t t = 0; t < T; ++t) {
        if (!active[t]) {                                   // (rows of skipped trajectories: their last values)
          fs[t] = f[t];
          memcpy(&gs[(size_t)t * P], &g[(size_t)t * P], P * sizeof(double));
        }
        double* Ft = &Fc[(size_t)t * NA];
        for (int r = 0; r < NA; ++r) Ft[r] = INFINITY;
        Ft[0] = std::isfinite(fs[t]) ? fs[t] : INFINITY;
        // need: the trajectories that rejected the full step - the ladder and the gradient at the accepted point are for them only
        need[t] = (active[t] && !(Ft[0] <= f[t] + c1 * alphas[0] * slope[t])) ? 1 : 0;
        if (need[t]) all_accept = false;
      }
      bool have_new = all_accept;
      if (all_accept) {
        fn = fs;
        gn = gs;
      } else if (G > 0) {
        cand.resize((size_t)T * G * P);
        Fl.resize((size_t)T * G);
        stl.resize((size_t)T * G);
        for (int64_t t = 0; t < T; ++t)
          for (int64_t r = 0; r < G; ++r)
            for (int k = 0; k < P; ++k) cand[((size_t)t * G + r) * P + k] = X[(size_t)t * P + k] + alphas[r + 1] * d[(size_t)t * P + k];
        if ((rc = qmps_overlap_set_group(c, G))) break;
        if ((rc = qmps_overlap_set_active(c, T, need.data()))) break;
        rc = qmps_overlap_eval_ansatz(c, T * G, kind, P, cand.data(), ladder_rounds, tol, 0, Fl.data(), stl.data());
        (void)qmps_overlap_set_group(c, 0);
        if (rc) break;
        n_ladder += 1.0;
        nfev += (double)T * G;
        for (int64_t t = 0; t < T; ++t)
          for (int64_t r = 0; r < G; ++r) {
            const double v = (need[t] && stl[(size_t)t * G + r] == qmps::QMPS_ST_OK) ? Fl[(size_t)t * G + r] : nan;
            Fc[(size_t)t * NA + r + 1] = std::isfinite(v) ? v : INFINITY;
          }
      }
      for (int64_t t = 0; t < T; ++t) {
        const double* Ft = &Fc[(size_t)t * NA];
        int first = -1, best = 0;
        for (int r = 0; r < NA; ++r) {
          if (first < 0 && Ft[r] <= f[t] + c1 * alphas[r] * slope[t]) first = r;
          if (Ft[r] < Ft[best]) best = r;
        }
        if (first < 0) first = best;
        moved[t] = (active[t] && Ft[first] < f[t]) ? 1 : 0;
        const double a = moved[t] ? alphas[first] : 0.0;
        for (int k = 0; k < P; ++k) {
          s[(size_t)t * P + k] = a * d[(size_t)t * P + k];
          Xn[(size_t)t * P + k] = X[(size_t)t * P + k] + s[(size_t)t * P + k];
        }
      }
      if (!have_new) {
        if ((rc = value_and_grad(Xn.data(), fn.data(), gn.data(), need.data()))) break;
        for (int64_t t = 0; t < T; ++t)
          if (!need[t]) {                                   // accepted the full step: its values are the speculative batch's
            fn[t] = fs[t];
            memcpy(&gn[(size_t)t * P], &gs[(size_t)t * P], P * sizeof(double));
          }
      }
      any_active = false;
      for (int64_t t = 0; t < T; ++t) {
        double* gt = &g[(size_t)t * P];
        const double* gnt = &gn[(size_t)t * P];
        const double* sv = &s[(size_t)t * P];
        if (moved[t]) {
          double sy = 0.0, ss = 0.0, yy = 0.0;
          for (int k = 0; k < P; ++k) { const double y = gnt[k] - gt[k]; sy += sv[k] * y; ss += sv[k] * sv[k]; yy += y * y; }
          if (sy > 1e-12 * sqrt(ss * yy) && sy > 0.0) {
            // H' = H - rho (s (Hy)^T + (Hy) s^T) + rho (1 + rho y^T H y) s s^T
            double* Ht = &Hinv[(size_t)t * P * P];
            const double rho = 1.0 / sy;
            double yHy = 0.0;
            for (int a = 0; a < P; ++a) {
              double acc = 0.0;
              for (int b = 0; b < P; ++b) acc += Ht[a * P + b] * (gnt[b] - gt[b]);
              Hy[a] = acc;
            }
            for (int a = 0; a < P; ++a) yHy += (gnt[a] - gt[a]) * Hy[a];
            const double coef = rho * (1.0 + rho * yHy);
            for (int a = 0; a < P; ++a)
              for (int b = 0; b < P; ++b) Ht[a * P + b] = Ht[a * P + b] - (rho * sv[a] * Hy[b] + rho * sv[b] * Hy[a]) + coef * sv[a] * sv[b];
          }
          f[t] = fn[t];
          memcpy(gt, gnt, P * sizeof(double));
        }
        active[t] = (active[t] && moved[t] && gmax_at_least(gt, gtol)) ? 1 : 0;
        any_active |= active[t] != 0;
      }
      X = Xn;
      ++nit;
    }
    if (rc) break;
    memcpy(f_hist + (size_t)step * T, f.data(), (size_t)T * sizeof(double));
    if (params_hist) memcpy(params_hist + (size_t)step * TP, X.data(), TP * sizeof(double));
    if (nit_out) nit_out[step] = nit;
  }
  c->timing_period = saved_period;
  if (rc) return rc;
  memcpy(params, X.data(), TP * sizeof(double));
  if (hinv) memcpy(hinv, Hinv.data(), TP * P * sizeof(double));
  if (counters_out) { counters_out[0] = n_grad; counters_out[1] = n_ladder; counters_out[2] = nfev; counters_out[3] = grad_ms; }
  return QMPS_OK;
}

int qmps_evolve_rotosolve(qmps_ctx* c, int64_t T, int kind, int n_params, double* params, const double* WW, int n_steps,
                          int n_sweeps, int nsh, int max_rounds, double tol, double* params_hist, double* f_hist) {
  if (int rc = bind(c)) return rc;
  if (!params || !WW || !f_hist) return fail(QMPS_ERR_ARG, "null argument");
  if (nsh != 3 && nsh != 6) return fail(QMPS_ERR_ARG, "nsh must be 3 (single-frequency) or 6 (double-frequency)");
  if (T < 1 || nsh * T > c->max_batch) return fail(QMPS_ERR_ARG, "%d T = %lld candidates exceed max_batch = %lld", nsh, (long long)(nsh * T), (long long)c->max_batch);
  if (n_steps < 1 || n_sweeps < 1) return fail(QMPS_ERR_ARG, "n_steps and n_sweeps must be >= 1");
  if (int rc = check_ansatz(c, kind, n_params)) return rc;
  const bool squaring = overlap_squares(c);
  const int cap = squaring ? 60 : (1 << 24);
  if (max_rounds < 1 || max_rounds > cap || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol (D = %d: max_rounds in [1, %d])", c->D, cap);
  const int P = n_params;
  const int64_t n_rec = (int64_t)n_steps * n_sweeps;
  auto grow = [&](double*& buf, size_t& have, size_t need) -> int {
    if (need > have) {
      if (buf) HIP_TRY(hipFree(buf));
      buf = nullptr;
      have = 0;
      HIP_TRY(hipMalloc((void**)&buf, need));
      have = need;
    }
    return QMPS_OK;
  };
  if (int rc = grow(c->roto_base, c->roto_base_bytes, (size_t)T * P * sizeof(double))) return rc;
  if (int rc = grow(c->roto_hist, c->roto_hist_bytes, (size_t)T * n_rec * sizeof(double))) return rc;
  if (!c->roto_idx) HIP_TRY(hipMalloc((void**)&c->roto_idx, 4 * sizeof(int)));
  if (int rc = ensure_refs(c, T)) return rc;
  if (int rc = ensure_E(c, c->n_terms > 0 ? c->n_terms : 1)) return rc;
  if (int rc = ensure_overlap_outputs(c)) return rc;
  if (int rc = ensure_scratch(c, (size_t)n_steps * T * P * sizeof(double))) return rc;    // parameter history
  // fixed points of the power method (D = 8, 16), one set per parameter plus one for the unshifted evaluation of a sweep:
  // the candidates of parameter i come back to the same slot in the next sweep and in the next time step - by then the
  // parameters have moved by one sweep's updates, so the resident fixed point is the natural warm start
  const bool warm = !squaring;
  const size_t slot_bytes = (size_t)nsh * T * env_bytes(c);
  if (warm) {
    const size_t need = (size_t)(P + 1) * slot_bytes;
    if (need > c->xwarm_bytes) {
      if (c->d_xwarm) HIP_TRY(hipFree(c->d_xwarm));
      c->d_xwarm = nullptr;
      c->xwarm_bytes = 0;
      HIP_TRY(hipMalloc(&c->d_xwarm, need));
      c->xwarm_bytes = need;
    }
    HIP_TRY(hipMemsetAsync(c->d_xwarm, 0, need, c->stream));       // all zero = cold start
  }
  double *d_base = c->roto_base, *d_hist = c->roto_hist, *d_phist = (double*)c->d_scratch;
  int* d_idx = c->roto_idx;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int rc = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(d_base, params, (size_t)T * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int idx0[4] = {0, 0, 0, P};      // parameter index, arrival counter, finished sweeps, slot of the unshifted evaluation
    HIP_TRY(hipMemcpyAsync(d_idx, idx0, sizeof(idx0), hipMemcpyHostToDevice, c->stream));
    if (int e = set_ww(c, WW)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->window = 0;
    c->have_env = false; c->have_guess = false; c->have_overlap_x = false; c->acc_pending = false; c->partials_B = -1;
    c->ans_have = false; c->ans_src = nullptr; c->ans_i = nullptr; c->ans_nsh = 0;
    auto evaluate = [&](int shifts) -> int {      // shifts = nsh: the shifted batch of parameter *d_idx;  0: the T base vectors
      const int64_t n = shifts > 0 ? (int64_t)shifts * T : T;
      HIP_TRY(qmps::launch_ansatz_shifted(c->D, kind, d_base, P, c->d_A, n, shifts, d_idx, c->stream));
      qmps::OverlapArgs a;
      memset(&a, 0, sizeof(a));
      a.A = c->d_ref; a.Bt = c->d_A; a.WW = c->d_ww; a.eta = c->d_eta; a.f_out = c->d_E;
      a.iters = c->d_iters; a.status = c->d_status; a.B = n; a.group = shifts > 0 ? shifts : 1;
      a.max_rounds = max_rounds; a.tol = tol; a.stats = c->d_ostats;
      if (warm) {
        a.x_in = c->d_xwarm; a.r_out = c->d_xwarm;
        a.slot_ptr = shifts > 0 ? d_idx : d_idx + 3; a.slot_stride = (int64_t)slot_bytes;
      }
      return launch_overlap_kernels(c, a);
    };
    auto one_sweep = [&]() -> int {
      for (int i = 0; i < P; ++i) {
        if (int e = evaluate(nsh)) return e;
        HIP_TRY(qmps::launch_roto_update(d_base, c->d_E, c->d_status, (int)T, P, d_idx, 1, nsh, c->stream));
      }
      // the sweep's record: the objective of the updated vectors against this time step's reference states
      if (int e = evaluate(0)) return e;
      HIP_TRY(qmps::launch_roto_record(c->d_E, d_hist, (int)T, 1, d_idx + 2, 1, c->stream));
      return QMPS_OK;
    };
    const bool use_graph = documented_switch("QMPS_NO_GRAPH") == nullptr && P <= 256;
    if (use_graph) {
      c->capturing = true;
      HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      const int e = one_sweep();
      const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
      c->capturing = false;
      if (e) return e;
      HIP_TRY(ce);
      HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    for (int step = 0; step < n_steps; ++step) {
      // the states the step starts from are the reference: A_t = tensor(params_t)  (new_time_evolve.py:281-283)
      HIP_TRY(qmps::launch_ansatz(c->D, kind, d_base, P, c->d_ref, T, c->stream));
      for (int sw = 0; sw < n_sweeps; ++sw) {
        if (use_graph) HIP_TRY(hipGraphLaunch(exec, c->stream));
        else if (int e = one_sweep()) return e;
      }
      HIP_TRY(hipMemcpyAsync(d_phist + (size_t)step * T * P, d_base, (size_t)T * P * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(params, d_base, (size_t)T * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(f_hist, d_hist, (size_t)T * n_rec * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (params_hist) HIP_TRY(hipMemcpyAsync(params_hist, d_phist, (size_t)n_steps * T * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMPS_OK;
  }();
  c->capturing = false;
  (void)hipStreamSynchronize(c->stream);
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  // what the call leaves resident: the T final candidates (tensors, eta, objective, status) against the last step's references
  c->n_states = rc == QMPS_OK ? T : 0;
  c->tensors_valid = true;
  c->overlap_refs = rc == QMPS_OK ? T : 0;
  c->overlap_group = 0;
  return rc;
}

int qmps_overlap_batch(qmps_ctx* c, int64_t B, const double* A, int a_shared, const double* states, int kind,
                       int n_params, const double* WW, int max_rounds, double tol, double* eta_out, double* r_out,
                       int32_t* rounds_out, int32_t* status_out) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_B(c, B)) return rc;
  if (!A || !WW || !eta_out || (!states && B > 0)) return fail(QMPS_ERR_ARG, "null argument");
  // candidates -> d_A: tensors, unitaries or ansatz parameters
  if (kind == QMPS_INPUT_TENSOR || kind == QMPS_INPUT_UNITARY) {
    if (int rc = qmps_set_states(c, B, states, kind)) return rc;
  } else if (kind >= QMPS_INPUT_ANSATZ_BASE && kind <= QMPS_INPUT_ANSATZ_BASE + 6) {
    if (int rc = qmps_set_states_ansatz(c, B, kind - QMPS_INPUT_ANSATZ_BASE, n_params, states)) return rc;
  } else {
    return fail(QMPS_ERR_ARG, "unknown input kind %d", kind);
  }
  if (B == 0) return QMPS_OK;
  if (int rc = qmps_overlap_set(c, a_shared ? 1 : B, A, WW)) return rc;
  if (int rc = qmps_overlap_launch(c, B, max_rounds, tol, r_out != nullptr ? QMPS_OVERLAP_WANT_R : 0)) return rc;
  return qmps_overlap_get(c, B, eta_out, r_out, rounds_out, status_out);
}

// ---- brick-wall (new_tdvp) contractions -------------------------------------------------------
namespace {
// bump allocator over the scratch arena: copies a host array in, returns the device address
struct Arena {
  qmps_ctx* c;
  size_t off = 0;
  void* put(const void* host, size_t bytes, hipError_t* err) {
    void* d = (char*)c->d_scratch + off;
    off += (bytes + 255) & ~(size_t)255;
    if (host) *err = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->stream);
    return d;
  }
};
}  // namespace

int qmps_bw_expval(qmps_ctx* c, int64_t B, int sites, const double* U1, const double* U2, const double* O, int o_shared,
                   double* out) {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !U1 || !U2 || !O || !out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (sites != 2 && sites != 4) return fail(QMPS_ERR_ARG, "sites must be 2 or 4");
  const size_t no = sites == 2 ? 16 : 256;
  const size_t ob = (o_shared ? 1 : (size_t)B) * no * 16;
  if (int rc = ensure_scratch(c, (size_t)B * (256 + 256 + 16 + 256) + ob + 4096)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  qmps::BwArgs k;
  memset(&k, 0, sizeof(k));
  k.U1 = a.put(U1, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2 = a.put(U2, (size_t)B * 256, &e); HIP_TRY(e);
  k.O = a.put(O, ob, &e); HIP_TRY(e);
  k.out = a.put(nullptr, (size_t)B * 16, &e);
  k.B = B; k.o_shared = o_shared ? 1 : 0;
  HIP_TRY(qmps::launch_bw(sites == 2 ? 0 : 1, k, c->stream));
  HIP_TRY(hipMemcpyAsync(out, k.out, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_bw_env(qmps_ctx* c, int64_t B, int side, const double* U1, const double* U2, const double* U1p,
                const double* U2p, int max_rounds, double tol, double* mat_out, double* eta_out, double* vec_out,
                int32_t* status_out) {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !U1 || !U2 || !U1p || !U2p || !eta_out || !vec_out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (side != 0 && side != 1) return fail(QMPS_ERR_ARG, "side must be 0 (right) or 1 (left)");
  if (max_rounds < 1 || max_rounds > 60 || !(tol > 0.0)) return fail(QMPS_ERR_ARG, "bad max_rounds / tol");
  if (int rc = ensure_scratch(c, (size_t)B * (4 * 256 + 256 + 16 + 64 + 16) + 8192)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  qmps::BwArgs k;
  memset(&k, 0, sizeof(k));
  k.U1 = a.put(U1, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2 = a.put(U2, (size_t)B * 256, &e); HIP_TRY(e);
  k.U1p = a.put(U1p, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2p = a.put(U2p, (size_t)B * 256, &e); HIP_TRY(e);
  k.mat_out = mat_out ? a.put(nullptr, (size_t)B * 256, &e) : nullptr;
  k.out = a.put(nullptr, (size_t)B * 16, &e);
  k.vec_out = a.put(nullptr, (size_t)B * 64, &e);
  k.status = (int32_t*)a.put(nullptr, (size_t)B * 4, &e);
  k.B = B; k.side = side; k.max_rounds = max_rounds; k.tol = tol;
  HIP_TRY(qmps::launch_bw(2, k, c->stream));
  if (mat_out) HIP_TRY(hipMemcpyAsync(mat_out, k.mat_out, (size_t)B * 256, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(eta_out, k.out, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(vec_out, k.vec_out, (size_t)B * 64, hipMemcpyDeviceToHost, c->stream));
  if (status_out) HIP_TRY(hipMemcpyAsync(status_out, k.status, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_bw_manifold(qmps_ctx* c, int64_t B, const double* U1, const double* U2, const double* U1p, const double* U2p,
                     const double* Mr, const double* Ml, int m_shared, const double* W, int w_shared, double* out) {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !U1 || !U2 || !U1p || !U2p || !Mr || !Ml || !W || !out) return fail(QMPS_ERR_ARG, "bad arguments");
  const size_t mb = (m_shared ? 1 : (size_t)B) * 64, wb = (w_shared ? 1 : (size_t)B) * 4096;
  if (int rc = ensure_scratch(c, (size_t)B * (4 * 256 + 16) + 2 * mb + wb + 8192)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  qmps::BwArgs k;
  memset(&k, 0, sizeof(k));
  k.U1 = a.put(U1, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2 = a.put(U2, (size_t)B * 256, &e); HIP_TRY(e);
  k.U1p = a.put(U1p, (size_t)B * 256, &e); HIP_TRY(e);
  k.U2p = a.put(U2p, (size_t)B * 256, &e); HIP_TRY(e);
  k.Mr = a.put(Mr, mb, &e); HIP_TRY(e);
  k.Ml = a.put(Ml, mb, &e); HIP_TRY(e);
  k.O = a.put(W, wb, &e); HIP_TRY(e);
  k.out = a.put(nullptr, (size_t)B * 16, &e);
  k.B = B; k.m_shared = m_shared ? 1 : 0; k.o_shared = w_shared ? 1 : 0;
  HIP_TRY(qmps::launch_bw(3, k, c->stream));
  HIP_TRY(hipMemcpyAsync(out, k.out, (size_t)B * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_opt_env_objective(qmps_ctx* c, int64_t B, const double* params, const double* h, double k, double* f_out,
                           double* parts_out) {
  if (int rc = bind(c)) return rc;
  if (B < 0 || !params || !h || !f_out) return fail(QMPS_ERR_ARG, "bad arguments");
  if (int rc = ensure_scratch(c, (size_t)B * (240 + 8 + 32) + 4096)) return rc;
  Arena a{c};
  hipError_t e = hipSuccess;
  const double* d_p = (const double*)a.put(params, (size_t)B * 240, &e); HIP_TRY(e);
  const void* d_h = a.put(h, 256, &e); HIP_TRY(e);
  double* d_f = (double*)a.put(nullptr, (size_t)B * 8, &e);
  double* d_parts = parts_out ? (double*)a.put(nullptr, (size_t)B * 32, &e) : nullptr;
  HIP_TRY(qmps::launch_opt_env(d_p, d_h, k, d_f, d_parts, B, c->stream));
  HIP_TRY(hipMemcpyAsync(f_out, d_f, (size_t)B * 8, hipMemcpyDeviceToHost, c->stream));
  if (parts_out) HIP_TRY(hipMemcpyAsync(parts_out, d_parts, (size_t)B * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return QMPS_OK;
}

int qmps_timer_begin(qmps_ctx* c) {
  if (int rc = bind(c)) return rc;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return QMPS_OK;
}

int qmps_timer_end(qmps_ctx* c, float* ms) {
  if (int rc = bind(c)) return rc;
  if (!ms) return fail(QMPS_ERR_ARG, "null ms");
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return QMPS_OK;
}

// ---- RCCL ---------------------------------------------------------------------------------
int qmps_comm_unique_id(char id[QMPS_UNIQUE_ID_BYTES]) {
  if (!id) return fail(QMPS_ERR_ARG, "null id");
  static_assert(sizeof(ncclUniqueId) <= QMPS_UNIQUE_ID_BYTES, "ncclUniqueId larger than QMPS_UNIQUE_ID_BYTES");
  ncclUniqueId u;
  RCCL_TRY(ncclGetUniqueId(&u));
  memset(id, 0, QMPS_UNIQUE_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return QMPS_OK;
}

int qmps_comm_init(qmps_ctx* c, const char id[QMPS_UNIQUE_ID_BYTES], int rank, int nranks) {
  if (int rc = bind(c)) return rc;
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(QMPS_ERR_ARG, "bad communicator arguments");
  if (c->comm) return fail(QMPS_ERR_STATE, "communicator already initialised");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  RCCL_TRY(ncclCommInitRank(&c->comm, nranks, u, rank));
  if (!tuning_knob("QMPS_ONE_COMM")) {
    // second communicator over the same ranks (collective, like the init itself); without it everything runs on the first
    ncclResult_t r2 = ncclCommSplit(c->comm, 0, rank, &c->comm2, nullptr);
    if (r2 != ncclSuccess) c->comm2 = nullptr;
    // every rank must take the same decision (slot -> communicator): agree on min over ranks of "I have the second one"
    HIP_TRY(hipStreamSynchronize(c->stream));
    double* flag = c->d_cost;
    const double mine = c->comm2 ? 1.0 : 0.0;
    double all = 0.0;
    HIP_TRY(hipMemcpyAsync(flag, &mine, sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
    RCCL_TRY(ncclAllReduce(flag, flag, 1, ncclDouble, ncclMin, c->comm, c->comm_stream));
    HIP_TRY(hipMemcpyAsync(&all, flag, sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    if (all < 0.5 && c->comm2) {
      (void)ncclCommDestroy(c->comm2);
      c->comm2 = nullptr;
    }
  }
  c->rank = rank;
  c->nranks = nranks;
  return QMPS_OK;
}

int qmps_comm_destroy(qmps_ctx* c) {
  if (int rc = bind(c)) return rc;
  if (c->comm) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->comm_stream));
    HIP_TRY(hipStreamSynchronize(c->comm_stream2));
    if (c->comm2) RCCL_TRY(ncclCommDestroy(c->comm2));
    c->comm2 = nullptr;
    RCCL_TRY(ncclCommDestroy(c->comm));
    c->comm = nullptr;
    c->nranks = 1;
    c->rank = 0;
  }
  return QMPS_OK;
}

int qmps_comm_count(qmps_ctx* c, int* nranks) {
  if (int rc = bind(c)) return rc;
  if (!nranks) return fail(QMPS_ERR_ARG, "null nranks");
  *nranks = 1;
  if (c->comm) RCCL_TRY(ncclCommCount(c->comm, nranks));
  return QMPS_OK;
}

int qmps_allreduce_sum(qmps_ctx* c, double* inout, int n) {
  if (int rc = bind(c)) return rc;
  if (!inout || n < 1 || n > kMaxTerms) return fail(QMPS_ERR_ARG, "n=%d outside [1,%d]", n, kMaxTerms);
  if (!c->comm) return fail(QMPS_ERR_STATE, "qmps_comm_init has not been called");
  memcpy(c->h_cost, inout, n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->d_cost, c->h_cost, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RCCL_TRY(ncclAllReduce(c->d_cost, c->d_cost, n, ncclDouble, ncclSum, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(inout, c->h_cost, n * sizeof(double));
  return QMPS_OK;
}

int qmps_allreduce_min(qmps_ctx* c, double* inout, int n) {
  if (int rc = bind(c)) return rc;
  if (!inout || n < 1 || n > kMaxTerms) return fail(QMPS_ERR_ARG, "n=%d outside [1,%d]", n, kMaxTerms);
  if (!c->comm) return fail(QMPS_ERR_STATE, "qmps_comm_init has not been called");
  memcpy(c->h_cost, inout, n * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->d_cost, c->h_cost, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  RCCL_TRY(ncclAllReduce(c->d_cost, c->d_cost, n, ncclDouble, ncclMin, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(inout, c->h_cost, n * sizeof(double));
  return QMPS_OK;
}

namespace {
// close the current group: ONE ncclAllReduce of its `fill` x 16 doubles on the communication stream, ordered after the
// device-side sums by an event, so the exchange overlaps the next steps' kernels instead of stalling the compute stream
int close_group(qmps_ctx* c) {
  if (c->group_fill == 0) return QMPS_OK;
  const int slot = (int)(c->groups % qmps_ctx::kCostSlots);
  double* base = c->d_cost_ring + (size_t)slot * qmps_ctx::kMaxGroup * kMaxTerms;
  if (c->comm) {
#ifdef QMPS_DEBUG_KNOBS   // timing dissections only (they produce WRONG costs): compiled in with -DQMPS_DEBUG_KNOBS, never in the shipped library
    static const bool dbg_noevent = getenv("QMPS_DBG_NOEVENT") != nullptr, dbg_noar = getenv("QMPS_DBG_NOAR") != nullptr,
                      dbg_nofinish = getenv("QMPS_DBG_NOFINISH") != nullptr, dbg_nopoll = getenv("QMPS_DBG_NOPOLL") != nullptr;
#else
    constexpr bool dbg_noevent = false, dbg_noar = false, dbg_nofinish = false, dbg_nopoll = false;
#endif
    // positions whose cost lives in a fixed-point accumulator need no ordering on the compute stream: their finish
    // kernel polls the arrival counts.  Only costs written by reduction kernels on the compute stream need the event.
    bool need_event = false;
    for (int pos = 0; pos < c->group_fill; ++pos) need_event = need_event || !c->acc_is[slot][pos] || c->acc_after_event[slot][pos];
    if (need_event && !dbg_noevent) {
      HIP_TRY(hipEventRecord(c->cost_ready[slot], c->stream));
      HIP_TRY(hipStreamWaitEvent(c->comm_stream_of(slot), c->cost_ready[slot], 0));
    }
    for (int pos = 0; pos < c->group_fill; ++pos)
      if (c->acc_is[slot][pos]) {   // fixed-point accumulators -> doubles, off the compute stream
        if (!dbg_nofinish)
          HIP_TRY(qmps::launch_cost_finish(c->acc_at(slot, pos), c->acc_shards[slot][pos], c->acc_expect[slot][pos], dbg_nopoll ? 0 : 1 << 22,
                                           1.0 / c->acc_scale[slot][pos], c->n_terms, base + (size_t)pos * kMaxTerms,
                                           c->d_acc_err, c->comm_stream_of(slot)));
        c->acc_is[slot][pos] = false;
      }
#ifdef QMPS_DEBUG_KNOBS
    // robustness drill for the exchange pipeline at world size 1, where the real all-reduce is instantaneous: a busy kernel in
    // front of it makes every exchange last QMPS_DBG_SLOW_AR probe iterations (~1300 = 40 us, longer than a step), so the ring
    // fills up, the host-side slot guard blocks and the finish kernels queue behind exchanges that are still in flight
    static const int slow_ar = getenv("QMPS_DBG_SLOW_AR") ? atoi(getenv("QMPS_DBG_SLOW_AR")) : 0;
    if (slow_ar > 0) HIP_TRY(qmps::launch_probe_fp64((double*)c->d_work_idx, 1, slow_ar, c->comm_stream_of(slot)));
#endif
    if (!dbg_noar)
      RCCL_TRY(ncclAllReduce(base, base, (size_t)c->group_fill * kMaxTerms, ncclDouble, ncclSum, c->comm_of(slot), c->comm_stream_of(slot)));
    HIP_TRY(hipEventRecord(c->cost_reduced[slot], c->comm_stream_of(slot)));
  }
  c->group_fill = 0;
  c->groups++;
  c->slot_waited = false;
  return QMPS_OK;
}
}  // namespace

int qmps_exchange_stats(qmps_ctx* c, int64_t* checks, int64_t* blocked, double* blocked_ms, int reset) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (checks) *checks = c->slot_checks;
  if (blocked) *blocked = c->slot_blocks;
  if (blocked_ms) *blocked_ms = c->slot_block_ms;
  if (reset) { c->slot_checks = 0; c->slot_blocks = 0; c->slot_block_ms = 0.0; }
  return QMPS_OK;
}

int qmps_set_exchange_period(qmps_ctx* c, int steps) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (steps < 1 || steps > qmps_ctx::kMaxGroup) return fail(QMPS_ERR_ARG, "exchange period must be in [1, %d]", qmps_ctx::kMaxGroup);
  if (int rc = bind(c)) return rc;
  if (int rc = close_group(c)) return rc;     // costs summed under the old period are exchanged now
  c->exchange_period = steps;
  return QMPS_OK;
}

int qmps_cost_launch(qmps_ctx* c, int64_t B) {
  if (int rc = bind(c)) return rc;
  if (int rc = check_window(c, B)) return rc;
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  // device-side sum into this step's place in the current group of the ring (main stream) ...
  const int slot = (int)(c->groups % qmps_ctx::kCostSlots);
  double* dst = c->d_cost_ring + ((size_t)slot * qmps_ctx::kMaxGroup + c->group_fill) * kMaxTerms;
  c->acc_is[slot][c->group_fill] = false;
  const bool in_kernel = c->acc_pending && c->acc_B == B && c->acc_window == c->window && c->acc_slot == slot && c->acc_pos == c->group_fill;
  // A slot is reused only after its previous all-reduce has finished.  Costs written by a reduction kernel on the compute
  // stream need that as a stream dependency; a cost that lives in a fixed-point accumulator is converted on the slot's own
  // communication stream, behind that all-reduce, and puts nothing on the compute stream (no barrier packet per step).
  if (c->comm && !in_kernel && c->groups >= qmps_ctx::kCostSlots && !c->slot_waited) {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->cost_reduced[slot], 0));
    c->slot_waited = true;
  }
  if (in_kernel) {
    // the energy kernel has summed the batch itself (exact fixed-point accumulator): nothing to launch
    c->acc_is[slot][c->group_fill] = true;
  } else if (c->partials_B == B)   // the energy kernel already left per-wave partial sums: only the final pass is needed
    HIP_TRY(qmps::launch_sum_final(c->d_partial, c->partials_n, c->n_terms, dst, c->stream));
  else {
    c->partials_B = -1;   // the generic two-pass reduction reuses d_partial
    HIP_TRY(qmps::launch_sum(win_E(c), B, c->n_terms, c->d_partial, kSumBlocks, dst, c->stream));
  }
  c->acc_pending = false;
  c->last_slot = slot;
  c->last_pos = c->group_fill;
  c->group_fill++;
  c->cost_launches++;
  // ... then, once per `exchange_period` steps, the exchange step
  if (c->group_fill >= c->exchange_period)
    if (int rc = close_group(c)) return rc;
  return QMPS_OK;
}

int qmps_get_cost(qmps_ctx* c, double* cost) {
  if (int rc = bind(c)) return rc;
  if (!cost) return fail(QMPS_ERR_ARG, "null cost");
  if (c->n_terms < 1) return fail(QMPS_ERR_STATE, "no energies resident");
  if (c->cost_launches < 1) return fail(QMPS_ERR_STATE, "qmps_cost_launch has not been called");
  if (int rc = close_group(c)) return rc;     // a partly filled group is exchanged now
  hipStream_t st = c->comm ? c->comm_stream_of(c->last_slot) : c->stream;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->acc_is[c->last_slot][c->last_pos]) {
    // no communicator: the cost still lives in its fixed-point accumulator; sum the shards on the host (exact)
    HIP_TRY(hipMemcpyAsync(c->h_acc, c->acc_at(c->last_slot, c->last_pos), qmps::kAccWords * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double inv = 1.0 / c->acc_scale[c->last_slot][c->last_pos];
    for (int t = 0; t < c->n_terms; ++t) {
      long long cnt = 0, hi = 0, lo = 0;
      for (int sh = 0; sh < c->acc_shards[c->last_slot][c->last_pos]; ++sh) {
        long long k, v;
        qmps::acc_decode(c->h_acc[t * qmps::kAccMaxShards + sh], k, v);
        cnt += k;
        hi += v >> 20;
        lo += v & 0xFFFFF;
      }
      if (cnt != c->acc_expect[c->last_slot][c->last_pos])
        return fail(QMPS_ERR_STATE, "cost accumulator: %lld of %lld waves arrived", cnt, c->acc_expect[c->last_slot][c->last_pos]);
      cost[t] = ((double)hi * 1048576.0 + (double)lo) * inv + ((const double*)(c->h_acc + qmps::kAccOver))[t];
    }
    return QMPS_OK;
  }
  HIP_TRY(hipMemcpyAsync(c->h_cost, c->d_cost_ring + ((size_t)c->last_slot * qmps_ctx::kMaxGroup + c->last_pos) * kMaxTerms,
                         c->n_terms * sizeof(double), hipMemcpyDeviceToHost, st));
  int acc_err = 0;
  if (c->comm) HIP_TRY(hipMemcpyAsync(&acc_err, c->d_acc_err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (acc_err) {
    // a finish kernel gave up waiting for its energy kernel's waves (bounded poll): the cost it wrote is NaN
    (void)hipMemsetAsync(c->d_acc_err, 0, sizeof(int), st);
    return fail(QMPS_ERR_STATE, "cost accumulator: a step's energy kernel did not arrive within the polling bound (was it launched?)");
  }
  memcpy(cost, c->h_cost, c->n_terms * sizeof(double));
  return QMPS_OK;
}

int qmps_allreduce_cost(qmps_ctx* c, int64_t B, double* cost) {
  if (!c) return fail(QMPS_ERR_ARG, "null context");
  if (!c->comm) return fail(QMPS_ERR_STATE, "qmps_comm_init has not been called");
  if (int rc = qmps_cost_launch(c, B)) return rc;
  return qmps_get_cost(c, cost);
}

// ---- probes -------------------------------------------------------------------------------
int qmps_probe_fp64_peak(qmps_ctx* c, double* tflops) {
  if (int rc = bind(c)) return rc;
  if (!tflops) return fail(QMPS_ERR_ARG, "null tflops");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int blocks = prop.multiProcessorCount * 8;  // 2 waves per SIMD
  const int iters = 20000;
  HIP_TRY(qmps::launch_probe_fp64(c->d_cost, blocks, 200, c->stream));  // warm-up
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(qmps::launch_probe_fp64(c->d_cost, blocks, iters, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms < best) best = ms;
  }
  const double flops = 2.0 * 16.0 * iters * 256.0 * blocks;
  *tflops = flops / (best * 1e-3) * 1e-12;
  return QMPS_OK;
}

int qmps_probe_fp64_mfma_peak(qmps_ctx* c, int waves_per_simd, double* tflops) {
  if (int rc = bind(c)) return rc;
  if (!tflops || waves_per_simd < 1 || waves_per_simd > 8) return fail(QMPS_ERR_ARG, "bad arguments");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int blocks = prop.multiProcessorCount * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  const int iters = 20000;
  HIP_TRY(qmps::launch_probe_mfma_f64(c->d_cost, blocks, 200, c->stream));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(qmps::launch_probe_mfma_f64(c->d_cost, blocks, iters, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms < best) best = ms;
  }
  const double flops = 4.0 * 2048.0 * iters * 4.0 * blocks;  // 4 MFMAs x 2048 flop, 4 waves per block
  *tflops = flops / (best * 1e-3) * 1e-12;
  return QMPS_OK;
}

int qmps_probe_hbm_peak(qmps_ctx* c, double* gbps) {
  if (int rc = bind(c)) return rc;
  if (!gbps) return fail(QMPS_ERR_ARG, "null gbps");
  const size_t bytes = (size_t)1 << 30;  // 1 GiB each way: well past the 256 MiB Infinity Cache
  void *src = nullptr, *dst = nullptr;
  HIP_TRY(hipMalloc(&src, bytes));
  if (hipMalloc(&dst, bytes) != hipSuccess) {
    (void)hipFree(src);
    return fail(QMPS_ERR_HIP, "hipMalloc failed in the HBM probe");
  }
  int rc = [&]() -> int {
    HIP_TRY(hipMemsetAsync(src, 1, bytes, c->stream));
    HIP_TRY(qmps::launch_probe_copy(src, dst, (int64_t)(bytes / 16), c->stream));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      HIP_TRY(hipEventRecord(c->ev0, c->stream));
      HIP_TRY(qmps::launch_probe_copy(src, dst, (int64_t)(bytes / 16), c->stream));
      HIP_TRY(hipEventRecord(c->ev1, c->stream));
      HIP_TRY(hipEventSynchronize(c->ev1));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
      if (ms < best) best = ms;
    }
    *gbps = 2.0 * (double)bytes / (best * 1e-3) * 1e-9;
    return QMPS_OK;
  }();
  (void)hipFree(src);
  (void)hipFree(dst);
  return rc;
}

}  // extern "C"
