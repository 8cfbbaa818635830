"""A/B of two builds of the library on the D = 8 paths: energies of the one-launch solve + energy kernel and a whole rotosolve run,
compared bit for bit.  QMPS_HIP_LIB_A / QMPS_HIP_LIB_B name the two libraries (each run in its own process)."""
import os, sys, subprocess
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, root)
    from qmps_amd import EnergyEngine, _lib as L
    from qmps_amd.engine import _f64
    eng = EnergyEngine(8, 4096)
    X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0, -1.0]).astype(complex)
    h = np.kron(X, X) + np.kron(Y, Y) + 0.5 * np.kron(Z, Z)
    eng.set_hamiltonian(h)
    rng = np.random.default_rng(3)
    P = rng.standard_normal((768, 6))
    E, it, st = eng.energies_from_params(L.ANSATZ_SHALLOW_CNOT, P, h)
    R, sweeps = 256, 6
    Pc = np.ascontiguousarray(rng.standard_normal((R, 6)))
    hist = np.zeros(sweeps * R + 16)
    L.check(eng._lib.qmps_rotosolve(eng._ctx, R, 0, 6, _f64(Pc), sweeps, 10000, 1e-13, _f64(hist)))
    np.savez(sys.argv[2], E=E, it=it, st=st, Pc=Pc, hist=hist[:sweeps * R])
    sys.exit(0)
out = []
for tag in 'AB':
    env = dict(os.environ, QMPS_HIP_LIB=os.environ['QMPS_HIP_LIB_' + tag])
    f = '/tmp/d8_ab_%s.npz' % tag
    subprocess.check_call([sys.executable, __file__, 'child', f], env=env)
    out.append(np.load(f))
for k in out[0].files:
    a, b = out[0][k], out[1][k]
    print(k, 'identical' if np.array_equal(a, b) else 'max |diff| %.3e' % np.abs(a - b).max())
