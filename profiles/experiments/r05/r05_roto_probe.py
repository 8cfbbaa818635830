"""GPU probe: where does the device double-frequency rotosolve leave the reference-run trajectory?"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from oracle import qmps_oracle as O
from qmps_amd import EnergyEngine, _lib as L
g = np.load(os.path.join(ROOT, 'tests/golden/refshim_golden.npz'))
lib = ctypes.CDLL(os.path.join(ROOT, 'tests/csrc/libroto_emu.so'))
lib.roto_emu_step.restype = ctypes.c_double
lib.roto_emu_step.argtypes = [ctypes.c_double] * 4 + [ctypes.c_int]
SH = np.array([0.0, np.pi, np.pi / 2, -np.pi / 2, np.pi / 4, -np.pi / 4])
for tag, D, hn in (('D2_d2', 2, 'h_tfim'), ('D4_d2', 4, 'h_tfim')):
    h = g[hn]; x0 = g[f'roto_{tag}_x0']; xref = g[f'refshim_droto_{tag}_x']; Eref = g[f'refshim_droto_{tag}_E']
    R, P = x0.shape
    with EnergyEngine(D, 4096) as eng:
        eng.set_hamiltonian(h)
        # (1) device energies at the first parameter's six shifts vs oracle
        cand = np.repeat(x0[:, None, :], 6, axis=1); cand[:, :, 0] += SH
        E, it, st = eng.energies_from_params(0, cand.reshape(-1, P), h)
        Eo = np.array([O.energy_closed_form(O.unitary_to_tensor(O.shallow_cnot_unitary(D, p)), h) for p in cand.reshape(-1, P)])
        print(tag, 'six-shift energies: max |E_dev - E_oracle|', np.abs(E[:, 0] - Eo).max(), 'status', np.unique(st))
        # (2) manual lock-step replay with DEVICE energies and the host build of the rule: follows the reference?
        p = x0.copy()
        for i in range(P):
            cand = np.repeat(p[:, None, :], 6, axis=1); cand[:, :, i] += SH
            E, it, st = eng.energies_from_params(0, cand.reshape(-1, P), h)
            e = E[:, 0].reshape(R, 6)
            for r in range(R):
                A = e[r, 0] + e[r, 1]; Bv = e[r, 0] - e[r, 1]; C = e[r, 2] + e[r, 3]; Dv = e[r, 2] - e[r, 3]; Ev = e[r, 4] - e[r, 5]
                p[r, i] += lib.roto_emu_step(0.25 * (2 * Ev - np.sqrt(2) * Dv), 0.25 * (A - C), 0.5 * Dv, 0.5 * Bv, 0)
        print(tag, 'host-rule replay on device energies, 1 sweep: max |x - x_ref|', np.abs(p - xref[0]).max(1))
        # (3) the device driver, 1 sweep
        for env in (None, 'QMPS_NO_FUSED_ROTO'):
            if env: os.environ[env] = '1'
            es, pd = eng.double_rotosolve(0, x0, 1)
            if env: del os.environ[env]
            print(tag, env, 'device driver 1 sweep: |x - x_ref| per restart', np.abs(pd - xref[0]).max(1), '|E - E_ref|', np.abs(es[0] - Eref[0]))
            print(tag, env, 'per-parameter dx of restart 0..', (pd - xref[0])[:3])
        es, pg = eng.double_rotosolve(0, x0, 1, rule=L.ROTO_GLOBAL_ARGMIN)
        print(tag, 'global rule: |x - x_ref|', np.abs(pg - xref[0]).max(1))
