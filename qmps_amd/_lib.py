"""ctypes binding of libqmps_hip.so (C-ABI: include/qmps_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is usable the
product path raises.  Loading the library itself needs no GPU (the `-m "not gpu"` tests check
that every symbol of include/qmps_hip.h is exported); only `qmps_create` touches the device.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libqmps_hip.so')

QMPS_OK = 0
QMPS_ERR_ARG, QMPS_ERR_HIP, QMPS_ERR_NO_DEVICE, QMPS_ERR_STATE, QMPS_ERR_RCCL = -1, -2, -3, -4, -5
STATUS_OK, STATUS_NOT_CONVERGED, STATUS_NOT_PD = 0, 1, 2
STATUS_TIED = 4          # overlap path, D = 2, 4: eta = the common modulus of tied dominant eigenvalues (objective usable, r_out no fixed point)


def overlap_usable(st):
    """Overlap statuses whose eta / objective may be used: converged, or the common modulus of a tie (QMPS_STATUS_TIED)."""
    import numpy as _np
    st = _np.asarray(st)
    return (st == STATUS_OK) | (st == STATUS_TIED)

INPUT_TENSOR, INPUT_UNITARY = 0, 1
INPUT_ANSATZ_BASE = 16
ANSATZ_SHALLOW_CNOT, ANSATZ_SHALLOW_QAOA, ANSATZ_SHALLOW_FULL, ANSATZ_SHALLOW_CNOT3 = 0, 1, 2, 3
ANSATZ_SHALLOW_CNOT_NONUNIFORM, ANSATZ_EXACT_AFTER4, ANSATZ_STATE_GATE = 4, 5, 6
ENV_POWER = 0
ENV_POWER_SQUARING = 1
ENV_DIRECT = 2
FLAG_NO_ENV_OUT = 0x100
FLAG_ACCUMULATE_COST = 0x200
FLAG_WARM_RESIDENT = 0x400
FLAG_KRYLOV_FALLBACK = 0x800
OVERLAP_WANT_R, OVERLAP_WARM, OVERLAP_TWO_SIDED_F = 1, 2, 4
BFGS_CARRY_HESSIAN, BFGS_WARM, BFGS_TIGHT_GRADIENT, BFGS_ADAPTIVE_GRADIENT, BFGS_TIME_STEPS = 1, 2, 4, 8, 16
ROTO_REFERENCE, ROTO_GLOBAL_ARGMIN = 0, 1
# ansatz kinds the device-resident BFGS of D = 2 has kernels for (launch_evolve_bfgs_d2); the others run the host loop qmps_evolve_bfgs
EVOLVE_DEVICE_KINDS_D2 = (ANSATZ_SHALLOW_CNOT, ANSATZ_SHALLOW_QAOA, ANSATZ_SHALLOW_FULL, ANSATZ_SHALLOW_CNOT3, ANSATZ_STATE_GATE)
UNIQUE_ID_BYTES = 128

_dp = POINTER(c_double)
_ip = POINTER(c_int32)


class EvolveOpts(ctypes.Structure):
    """qmps_evolve_opts (include/qmps_hip.h): `size` first, so the struct may grow at its end."""
    _fields_ = [('size', ctypes.c_uint32), ('n_steps', c_int32), ('maxiter', c_int32), ('n_alphas', c_int32), ('flags', c_int32), ('max_rounds', c_int32),
                ('gtol', c_double), ('h', c_double), ('c1', c_double), ('tol', c_double), ('alphas', _dp)]


class EvolveOut(ctypes.Structure):
    """qmps_evolve_out (include/qmps_hip.h)."""
    _fields_ = [('size', ctypes.c_uint32), ('reserved', ctypes.c_uint32), ('hinv', _dp), ('params_hist', _dp), ('f_hist', _dp), ('nit', _ip), ('counters', _dp)]


# name -> (restype, argtypes): every entry point declared in include/qmps_hip.h
SIGNATURES = {
    'qmps_abi_version': (c_int, []),
    'qmps_abi_minor': (c_int, []),
    'qmps_selftest_exception': (c_int, [c_int]),
    'qmps_last_error': (c_char_p, []),
    'qmps_device_count': (c_int, [POINTER(c_int)]),
    'qmps_device_info': (c_int, [c_int, c_char_p, c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_int64)]),
    'qmps_create': (c_int, [c_int, c_int, c_int64, POINTER(c_void_p)]),
    'qmps_destroy': (c_int, [c_void_p]),
    'qmps_sync': (c_int, [c_void_p]),
    'qmps_set_states': (c_int, [c_void_p, c_int64, _dp, c_int]),
    'qmps_set_states_ansatz': (c_int, [c_void_p, c_int64, c_int, c_int, _dp]),
    'qmps_set_states_su': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_energy_batch_su': (c_int, [c_void_p, c_int64, _dp, _dp, c_int, c_int, c_double, _dp, _ip, _ip]),
    'qmps_su_unitaries': (c_int, [c_void_p, c_int64, c_int, _dp, _dp]),
    'qmps_cell2_energy_batch_su': (c_int, [c_void_p, c_int64, _dp, _dp, c_int, c_int, c_double, _dp, _ip, _ip]),
    'qmps_rotosolve': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, c_int, c_int, c_double, _dp]),
    'qmps_double_rotosolve': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, c_int, c_int, c_double, _dp]),
    'qmps_set_roto_rule': (c_int, [c_void_p, c_int]),
    'qmps_get_roto_rule': (c_int, [c_void_p, POINTER(c_int)]),
    'qmps_roto_rule_probe': (c_int, [c_void_p, c_int64, _dp, c_int, _dp]),
    'qmps_get_states': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_set_hamiltonian': (c_int, [c_void_p, c_int, _dp]),
    'qmps_set_env_guess': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_set_window': (c_int, [c_void_p, c_int64]),
    'qmps_energy_launch': (c_int, [c_void_p, c_int64, c_int, c_double, c_int]),
    'qmps_set_handoff': (c_int, [c_void_p, c_int]),
    'qmps_get_handoff': (c_int, [c_void_p, POINTER(c_int)]),
    'qmps_get_squaring_schedule': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    'qmps_set_default_solver': (c_int, [c_void_p, c_int]),
    'qmps_energy_only_launch': (c_int, [c_void_p, c_int64]),
    'qmps_sum_energies': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_get_energies': (c_int, [c_void_p, c_int64, _dp, _ip, _ip]),
    'qmps_get_status': (c_int, [c_void_p, c_int64, _ip]),
    'qmps_get_env': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_get_rdm': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_energy_batch': (c_int, [c_void_p, c_int64, _dp, c_int, _dp, c_int, _dp, c_int, c_double, _dp, _ip, _ip]),
    'qmps_energy_batch_ansatz': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp, c_int, c_int, c_double, _dp, _ip, _ip]),
    'qmps_env_batch': (c_int, [c_void_p, c_int64, _dp, c_int, _dp, c_int, c_double, _dp, _ip, _ip]),
    'qmps_cell2_energy_batch': (c_int, [c_void_p, c_int64, _dp, _dp, _dp, c_int, c_int, c_double, _dp, _ip, _ip]),
    'qmps_kernel_time': (c_int, [c_void_p, c_int, POINTER(c_float), c_char_p, c_int]),
    'qmps_set_kernel_timing_period': (c_int, [c_void_p, c_int]),
    'qmps_overlap_batch': (c_int, [c_void_p, c_int64, _dp, c_int, _dp, c_int, c_int, _dp, c_int, c_double, _dp, _dp, _ip, _ip]),
    'qmps_overlap_set': (c_int, [c_void_p, c_int64, _dp, _dp]),
    'qmps_overlap_launch': (c_int, [c_void_p, c_int64, c_int, c_double, c_int]),
    'qmps_overlap_get': (c_int, [c_void_p, c_int64, _dp, _dp, _ip, _ip]),
    'qmps_overlap_set_refs_ansatz': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp]),
    'qmps_overlap_set_group': (c_int, [c_void_p, c_int64]),
    'qmps_overlap_set_active': (c_int, [c_void_p, c_int64, c_char_p]),
    'qmps_overlap_get_objective': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_overlap_amplitude': (c_int, [c_void_p, c_int64, _dp, _dp]),
    'qmps_overlap_stats': (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), c_int]),
    'qmps_overlap_eval_ansatz': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, c_int, c_double, c_int, _dp, _ip]),
    'qmps_overlap_gradient': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, c_double, c_int, c_double, c_int, _dp, _dp, _ip]),
    'qmps_evolve_rotosolve': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp, c_int, c_int, c_int, c_int, c_double, _dp, _dp]),
    'qmps_evolve_bfgs': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp, c_int, c_int, c_double, c_double, c_double, c_int, _dp, c_int, c_int, c_double,
                                 _dp, _dp, _dp, _ip, _dp]),
    'qmps_set_evolve_groups': (c_int, [c_void_p, c_int]),
    'qmps_get_evolve_groups': (c_int, [c_void_p, c_int64, POINTER(c_int)]),
    'qmps_evolve_opts_init': (c_int, [POINTER(EvolveOpts)]),
    'qmps_evolve_bfgs_opts': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp, POINTER(EvolveOpts), POINTER(EvolveOut)]),
    'qmps_evolve_bfgs_device_opts': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp, POINTER(EvolveOpts), POINTER(EvolveOut)]),
    'qmps_evolve_bfgs_device': (c_int, [c_void_p, c_int64, c_int, c_int, _dp, _dp, c_int, c_int, c_double, c_double, c_double, c_int, _dp, c_int, c_int, c_double,
                                        _dp, _dp, _dp, _ip, _dp]),
    'qmps_opt_env_objective': (c_int, [c_void_p, c_int64, _dp, _dp, c_double, _dp, _dp]),
    'qmps_bw_expval': (c_int, [c_void_p, c_int64, c_int, _dp, _dp, _dp, c_int, _dp]),
    'qmps_bw_env': (c_int, [c_void_p, c_int64, c_int, _dp, _dp, _dp, _dp, c_int, c_double, _dp, _dp, _dp, _ip]),
    'qmps_bw_manifold': (c_int, [c_void_p, c_int64, _dp, _dp, _dp, _dp, _dp, _dp, c_int, _dp, c_int, _dp]),
    'qmps_timer_begin': (c_int, [c_void_p]),
    'qmps_timer_end': (c_int, [c_void_p, POINTER(c_float)]),
    'qmps_comm_unique_id': (c_int, [c_char_p]),
    'qmps_comm_init': (c_int, [c_void_p, c_char_p, c_int, c_int]),
    'qmps_comm_destroy': (c_int, [c_void_p]),
    'qmps_comm_count': (c_int, [c_void_p, POINTER(c_int)]),
    'qmps_allreduce_sum': (c_int, [c_void_p, _dp, c_int]),
    'qmps_allreduce_min': (c_int, [c_void_p, _dp, c_int]),
    'qmps_cost_launch': (c_int, [c_void_p, c_int64]),
    'qmps_set_exchange_period': (c_int, [c_void_p, c_int]),
    'qmps_exchange_stats': (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_double), c_int]),
    'qmps_get_cost': (c_int, [c_void_p, _dp]),
    'qmps_allreduce_cost': (c_int, [c_void_p, c_int64, _dp]),
    'qmps_probe_fp64_peak': (c_int, [c_void_p, _dp]),
    'qmps_probe_fp64_mfma_peak': (c_int, [c_void_p, c_int, _dp]),
    'qmps_probe_hbm_peak': (c_int, [c_void_p, _dp]),
}


class QmpsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libqmps_hip error {code}: {message}')
        self.code = code


_lib = None


def load():
    """Load libqmps_hip.so and bind every symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('QMPS_HIP_LIB', LIB_PATH)     # tuning experiments: an alternative build of the same library
    if not os.path.exists(path):
        raise ImportError(
            f'{path} not found: build it with `make -C qmps_amd/csrc` (or __graft_entry__.build()). '
            'qmps_amd has no CPU fallback.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.qmps_abi_version() != 6:
        raise ImportError('libqmps_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc):
    if rc != QMPS_OK:
        raise QmpsError(rc, load().qmps_last_error().decode('utf-8', 'replace'))


def device_count():
    n = c_int(0)
    check(load().qmps_device_count(byref(n)))
    return n.value


def device_info(device=0):
    name = ctypes.create_string_buffer(256)
    arch = ctypes.create_string_buffer(64)
    cus = c_int(0)
    hbm = c_int64(0)
    check(load().qmps_device_info(device, name, 256, arch, 64, byref(cus), byref(hbm)))
    return {'name': name.value.decode(), 'arch': arch.value.decode(), 'compute_units': cus.value,
            'hbm_bytes': hbm.value}
