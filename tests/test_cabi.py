"""CPU: the C-ABI library loads without a GPU, exports every symbol include/qmps_hip.h declares,
and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'qmps_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(qmps_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    from qmps_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/qmps_hip.h but not exported by libqmps_hip.so'
    # the ctypes table binds exactly the declared entry points
    assert sorted(_lib.SIGNATURES) == names
    assert lib.qmps_abi_version() == 6 and lib.qmps_abi_minor() >= 1


def test_no_cpu_fallback_without_device():
    from qmps_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('a GPU is visible here')
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    rc = lib.qmps_create(0, 4, 16, ctypes.byref(ctx))
    assert rc == _lib.QMPS_ERR_NO_DEVICE and not ctx.value
    assert b'no CPU fallback' in lib.qmps_last_error()
    from qmps_amd import EnergyEngine
    with pytest.raises(_lib.QmpsError):
        EnergyEngine(4, 16)
    # the reference-API objective fails loudly as well (it never routes to a CPU path)
    import numpy as np
    from qmps_amd.ground_state import Hamiltonian, SparseFullEnergyOptimizer
    opt = SparseFullEnergyOptimizer(Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix(), 2, 1,
                                    initial_guess=np.array([0.1, 0.2]))
    with pytest.raises(_lib.QmpsError):
        opt.objective_function(np.array([0.1, 0.2]))


def test_argument_validation_needs_no_device():
    from qmps_amd import _lib
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    assert lib.qmps_create(0, 3, 16, ctypes.byref(ctx)) == _lib.QMPS_ERR_ARG      # D not a supported power of two
    assert lib.qmps_create(0, 4, 0, ctypes.byref(ctx)) == _lib.QMPS_ERR_ARG
    assert lib.qmps_sync(None) == _lib.QMPS_ERR_ARG
    n = ctypes.c_int(-1)
    assert lib.qmps_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under qmps_amd/ may import or load it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'qmps_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f'{f} imports the oracle'
                assert 'libqmps_oracle' not in src and 'qmps_oracle_energy' not in src, f'{f} links the oracle'
                assert not re.search(r'#include\s+[<"].*oracle', src), f'{f} includes oracle sources'


def test_environment_switches_are_documented_or_compiled_out():
    """The library reads the environment only through qmps_knobs.h: `documented_switch` names must be listed in
    include/qmps_hip.h, `tuning_knob` names are dead unless built with -DQMPS_DEBUG_KNOBS, and a raw getenv() appears
    nowhere else except inside `#ifdef QMPS_DEBUG_KNOBS` blocks (the QMPS_DBG_* timing dissections)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'qmps_hip.h')).read()
    documented, tuning = set(), set()
    for f in glob.glob(os.path.join(root, 'qmps_amd', 'csrc', '*')):
        if not f.endswith(('.hip', '.h')) or f.endswith('qmps_knobs.h'):
            continue
        src = open(f).read()
        documented |= set(re.findall(r'documented_switch\("([A-Z0-9_]+)"\)', src))
        tuning |= set(re.findall(r'tuning_knob\("([A-Z0-9_]+)"\)', src))
        # raw getenv only between #ifdef QMPS_DEBUG_KNOBS and its #else / #endif
        depth_dbg = False
        for line in src.splitlines():
            t = line.strip()
            if t.startswith('#ifdef QMPS_DEBUG_KNOBS'):
                depth_dbg = True
            elif depth_dbg and (t.startswith('#else') or t.startswith('#endif')):
                depth_dbg = False
            if 'getenv(' in line:
                assert depth_dbg, (f, line)
    assert documented, 'no documented switches found'
    for name in documented:
        assert name in header, f'{name} is read by the library but not documented in include/qmps_hip.h'
    assert not (documented & tuning)
    knobs = open(os.path.join(root, 'qmps_amd', 'csrc', 'qmps_knobs.h')).read()
    assert '#ifdef QMPS_DEBUG_KNOBS' in knobs and knobs.count('getenv(') == 2


def test_nothing_throws_across_the_abi():
    """include/qmps_hip.h: "Nothing throws across the ABI".  Every extern "C" body is a function-try-block closed by QMPS_API_CATCH
    (checked in the sources), and an exception raised inside the library - std::bad_alloc, std::length_error from a vector of absurd
    size, a foreign type - comes back as an error code with a message, not as an abort of the Python process."""
    from qmps_amd import _lib
    lib = _lib.load()
    assert lib.qmps_selftest_exception(0) == 0
    for kind, word in ((1, b'out of host memory'), (2, b'exception'), (3, b'unknown C++ exception')):
        assert lib.qmps_selftest_exception(kind) == _lib.QMPS_ERR_ARG
        assert word in lib.qmps_last_error(), lib.qmps_last_error()
    csrc = os.path.join(ROOT, 'qmps_amd', 'csrc')
    for f in ('qmps_capi.hip', 'qmps_capi_overlap.hip', 'qmps_capi_evolve.hip'):
        src = open(os.path.join(csrc, f)).read()
        entry = re.findall(r'^int (qmps_\w+)\(', src, flags=re.M)
        guarded = re.findall(r'^int (qmps_\w+)\([^{;]*?\) try \{', src, flags=re.M | re.S)
        bare = sorted(set(entry) - set(guarded) - {'qmps_abi_version', 'qmps_abi_minor'})
        assert not bare, f'{f}: entry points without a function-try-block: {bare}'
        assert src.count('QMPS_API_CATCH') == len(guarded)
    # context fields that are switched for the duration of a call are restored by scope guards, not by hand
    for f in ('qmps_capi.hip', 'qmps_capi_overlap.hip', 'qmps_capi_evolve.hip'):
        src = open(os.path.join(csrc, f)).read()
        assert 'c->defer_sync = true' not in src and 'saved_period' not in src


def test_boundary_end_to_end_through_the_cpu_build_of_the_abi():
    """"a CPU build of the same ABI must exist so the boundary is testable without a GPU".  tests/csrc/libqmps_cpuabi.so
    exports qmps_create / qmps_destroy / qmps_energy_batch / qmps_env_batch / qmps_last_error with the header's prototypes (bound
    here through the PRODUCT's ctypes signature table) at D = 4, computing through qmps_direct_core.h on the host: argument
    marshalling, caller-owned buffers, return codes and error strings are exercised end to end and the numbers are the oracle's.
    Test infrastructure: qmps_amd never loads it (checked), the product still has no CPU fallback."""
    import subprocess
    import numpy as np
    from oracle import qmps_oracle as O
    from qmps_amd import _lib
    here = os.path.join(ROOT, 'tests', 'csrc')
    subprocess.check_call(['make', '-C', here, '--no-print-directory'], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(here, 'libqmps_cpuabi.so'))
    names = ('qmps_abi_version', 'qmps_last_error', 'qmps_create', 'qmps_destroy', 'qmps_energy_batch', 'qmps_env_batch')
    for n in names:                                     # the product's own signature table describes these entry points
        res, args = _lib.SIGNATURES[n]
        getattr(lib, n).restype, getattr(lib, n).argtypes = res, args
    assert lib.qmps_abi_version() == _lib.load().qmps_abi_version()
    ctx = ctypes.c_void_p()
    assert lib.qmps_create(0, 8, 16, ctypes.byref(ctx)) == _lib.QMPS_ERR_ARG and b'D = 4 only' in lib.qmps_last_error()
    assert lib.qmps_create(1, 4, 16, ctypes.byref(ctx)) == _lib.QMPS_ERR_NO_DEVICE
    assert lib.qmps_create(0, 4, 64, ctypes.byref(ctx)) == 0 and ctx.value
    rng = np.random.default_rng(44)
    B = 50
    A = np.ascontiguousarray(O.unitary_to_tensor(O.haar_unitaries(rng, 8, B)), dtype=np.complex128)      # (caller-owned, C-ordered buffers)
    h = np.ascontiguousarray(np.stack([O.hamiltonian_matrix({'ZZ': -1, 'X': 1}), O.hamiltonian_matrix({'XX': 1, 'YY': 1, 'ZZ': 0.5})]), dtype=np.complex128)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    f64 = lambda a: a.ctypes.data_as(dp)
    E = np.empty((B, 2))
    it, st = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
    rc = lib.qmps_energy_batch(ctx, B, f64(A.view(np.float64)), _lib.INPUT_TENSOR, f64(h.view(np.float64)), 2, None, 10000, 1e-13, f64(E),
                               it.ctypes.data_as(ip), st.ctypes.data_as(ip))
    assert rc == 0 and np.all(st == 0)
    ref = np.array([[O.energy_closed_form(a, hh) for hh in h] for a in A])
    assert np.abs(E - ref).max() < 1e-10
    r = np.empty((B, 4, 4), dtype=np.complex128)
    assert lib.qmps_env_batch(ctx, B, f64(A.view(np.float64)), _lib.INPUT_TENSOR, None, 10000, 1e-13, f64(r.view(np.float64)), None, None) == 0
    for b in range(0, B, 7):
        rr = r[b] / np.trace(r[b])
        assert np.abs(O.apply_transfer(A[b], rr) - rr).max() < 1e-11 and np.abs(rr - rr.conj().T).max() < 1e-13
    # error behaviour of the boundary: codes + messages, nothing written, nothing thrown
    assert lib.qmps_energy_batch(ctx, 65, f64(A.view(np.float64)), 0, f64(h.view(np.float64)), 2, None, 100, 1e-13, f64(E), None, None) == _lib.QMPS_ERR_ARG
    assert b'max_batch' in lib.qmps_last_error()
    assert lib.qmps_energy_batch(ctx, B, None, 0, f64(h.view(np.float64)), 2, None, 100, 1e-13, f64(E), None, None) == _lib.QMPS_ERR_ARG
    assert lib.qmps_energy_batch(ctx, B, f64(A.view(np.float64)), _lib.INPUT_UNITARY, f64(h.view(np.float64)), 2, None, 100, 1e-13, f64(E), None, None) == _lib.QMPS_ERR_ARG
    assert lib.qmps_energy_batch(None, B, f64(A.view(np.float64)), 0, f64(h.view(np.float64)), 2, None, 100, 1e-13, f64(E), None, None) == _lib.QMPS_ERR_ARG
    assert lib.qmps_destroy(ctx) == 0
    # never part of the product
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'qmps_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', 'Makefile')):
                assert 'cpuabi' not in open(os.path.join(dirpath, f)).read(), f


def test_evolve_option_structs_are_versioned_by_their_size_field():
    """ABI 6.1 (VERDICT r04 engineering item 9): qmps_evolve_bfgs[_device] behind structs whose first field is the caller's sizeof.
    Needs no device: defaults, and the size checks that run before anything touches the GPU."""
    import ctypes
    from qmps_amd import _lib
    lib = _lib.load()
    o = _lib.EvolveOpts()
    assert lib.qmps_evolve_opts_init(ctypes.byref(o)) == 0
    assert o.size == ctypes.sizeof(_lib.EvolveOpts) and (o.n_steps, o.maxiter, o.n_alphas, o.flags, o.max_rounds) == (1, 200, 0, 0, 0)
    assert (o.gtol, o.h, o.c1, o.tol) == (1e-5, 1e-6, 1e-4, 1e-12)
    assert lib.qmps_evolve_opts_init(None) == _lib.QMPS_ERR_ARG
    out = _lib.EvolveOut(size=ctypes.sizeof(_lib.EvolveOut))
    # a null context, a size from a NEWER header, a size too small to hold itself: refused with a message, nothing dereferenced
    assert lib.qmps_evolve_bfgs_opts(None, 1, 0, 2, None, None, ctypes.byref(o), ctypes.byref(out)) == _lib.QMPS_ERR_ARG
    ctx = ctypes.c_void_p(1)        # never dereferenced: the struct checks come first
    o.size = ctypes.sizeof(_lib.EvolveOpts) + 8
    assert lib.qmps_evolve_bfgs_opts(ctx, 1, 0, 2, None, None, ctypes.byref(o), ctypes.byref(out)) == _lib.QMPS_ERR_ARG
    assert b'newer header' in lib.qmps_last_error()
    o.size = 4
    assert lib.qmps_evolve_bfgs_device_opts(ctx, 1, 0, 2, None, None, ctypes.byref(o), ctypes.byref(out)) == _lib.QMPS_ERR_ARG
    o.size = ctypes.sizeof(_lib.EvolveOpts)
    assert lib.qmps_evolve_bfgs_opts(ctx, 1, 0, 2, None, None, ctypes.byref(o), ctypes.byref(out)) == _lib.QMPS_ERR_ARG     # f_hist is required
    assert b'f_hist' in lib.qmps_last_error()
    # the header and the ctypes mirror agree on the layout
    hdr = open(os.path.join(ROOT, 'include', 'qmps_hip.h')).read()
    body = hdr[hdr.index('typedef struct qmps_evolve_opts {'):hdr.index('} qmps_evolve_opts;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = re.findall(r'\b(\w+)(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?;', body)
    flat = [n for grp in names for n in grp if n]
    assert flat == [f[0] for f in _lib.EvolveOpts._fields_], flat
