"""Host-side mirror of the pieces of `qmps/time_evolve_tools.py` the energy path touches.

  merge                    qmps/time_evolve_tools.py:20-23   (any D here; the reference hard-codes D = 2)
  Nsphere                  qmps/time_evolve_tools.py:25-36
  put_env_on_left_site     qmps/time_evolve_tools.py:38-53
  get_env_off_left_site    qmps/time_evolve_tools.py:55-57
  put_env_on_right_site    qmps/time_evolve_tools.py:59-70
  get_env_off_right_site   qmps/time_evolve_tools.py:72-74

  gate, egate, get_overlap_exact   qmps/time_evolve_tools.py:76-92   (device: qmps_overlap_batch with W = 1)

  get_overlap              qmps/time_evolve_tools.py:95-131   the variational-environment route: Nelder-Mead over the 8 reals of an
                           environment r put on the circuit's outer qubits; the circuit amplitude psi[0] = 1/2 <r^, T(r^)>_F comes from
                           `qmps_overlap_amplitude` (device), the simplex stays scipy's as in the reference
"""
import numpy as np
from scipy.linalg import null_space

_SWAP = np.eye(4)[[0, 2, 1, 3]]


def merge(A, B):
    """-A- -B- -> -AB-: merge(A, B)[2 s1 + s2] = A[s1] @ B[s2]."""
    D = A.shape[1]
    return np.einsum('sij,tjk->stik', A, B).reshape(A.shape[0] * B.shape[0], D, D)


def Nsphere(v):
    """Point on the unit len(v)-sphere from len(v) angles (time_evolve_tools.py:25-36): x_k = sin v_1 .. sin v_(k-1) cos v_k, last entry the
    product of all sines."""
    v = np.asarray(v, dtype=float)
    sines = np.concatenate([[1.0], np.cumprod(np.sin(v))])
    return np.concatenate([sines[:-1] * np.cos(v), sines[-1:]])


def put_env_on_left_site(q, ret_n=False):
    """4x4 unitary whose (|0> on the ancilla) block carries the 2x2 matrix q/||q||."""
    a, b, c, d = np.asarray(q).T.reshape(-1)
    n = np.sqrt(abs(a) ** 2 + abs(b) ** 2 + abs(c) ** 2 + abs(d) ** 2)
    guess = np.array([[a, np.conj(c), b, np.conj(d)], [c, -np.conj(a), d, -np.conj(b)]]) / n
    A = _SWAP @ np.concatenate([guess, null_space(guess).conj().T], axis=0)
    return (A, n) if ret_n else A


def get_env_off_left_site(A):
    return np.asarray(A).reshape(2, 2, 2, 2)[:, 0, :, 0].T


def put_env_on_right_site(q, ret_n=False):
    a, b, c, d = np.asarray(q).reshape(-1)
    n = np.sqrt(abs(a) ** 2 + abs(b) ** 2 + abs(c) ** 2 + abs(d) ** 2)
    guess = np.array([[a, b, np.conj(d), -np.conj(c)], [c, d, -np.conj(b), np.conj(a)]]) / n
    A = np.concatenate([guess, null_space(guess).conj().T], axis=0)
    return (A, n) if ret_n else A


def get_env_off_right_site(A):
    return np.asarray(A).reshape(2, 2, 2, 2)[0, :, 0, :]


# ---- overlap between two one-site iMPS (time_evolve_tools.py:76-92) ---------------------------------------
def gate(v, symbol='U'):
    """The state gate the time-evolution code parameterises (time_evolve_tools.py:76-79)."""
    from .represent import ShallowFullStateTensor
    return ShallowFullStateTensor(2, v, symbol)


def egate(v, symbol='R'):
    from .represent import StateGate
    return StateGate(v, symbol)


def overlap_of_tensors(A, B, want_r=False):
    """|x|^2 with x the dominant eigenvalue of the mixed transfer map  r -> sum_s A_s r B_s^+  of two D = 2 tensors
    (what xmps `Map(A, B).right_fixed_point()` returns, squared), and its unit-Frobenius right fixed point r.
    Device path: the two-site map with W = 1 is that map applied twice - same fixed point, eigenvalue x^2 - so
    `qmps_overlap_batch` answers directly.  A: (2,2,2) or (B,2,2,2); B: (2,2,2) or (B,2,2,2)."""
    from . import _lib as L
    from . import _runtime
    A = np.asarray(A, dtype=complex)
    Bt = np.asarray(B, dtype=complex)
    single = Bt.ndim == 3
    cand = Bt[None] if single else Bt
    eng = _runtime.engine(2, len(cand))
    eta, _, st, r = eng.overlaps(A, cand, np.eye(4), kind='tensor', want_r=True)
    # QMPS_STATUS_TIED (D = 2: dominant eigenvalues tied in modulus) has a valid |eta| but no fixed point: good enough without `want_r` only
    if np.any(~L.overlap_usable(st)) or (want_r and np.any(st != L.STATUS_OK)):
        raise np.linalg.LinAlgError('mixed transfer map has no unique dominant eigenvalue')
    x2 = np.abs(eta)
    if single:
        x2, r = float(x2[0]), r[0]
    return (x2, r) if want_r else x2


def get_overlap_exact(p1, p2, gate=gate, testing=True):
    """Fidelity per site |x|^2 between the iMPS of gate(p1) and gate(p2) (time_evolve_tools.py:84-91); with
    `testing` also the right fixed point (unit Frobenius norm; its phase is arbitrary, as in xmps)."""
    from .represent import unitary
    from .tools import unitary_to_tensor
    A = unitary_to_tensor(unitary(gate(p1)))
    B = unitary_to_tensor(unitary(gate(p2)))
    x2, r = overlap_of_tensors(A, B, want_r=True)
    return (x2, r) if testing else x2


def overlap_amplitudes(A, B, WW, q):
    """psi[0] of the reference's 6-qubit overlap circuit (time_evolve_tools.py:113-127, scripts/loschmidt.py:228-238) with the
    environment q on its outer qubits - R = put_env_on_left_site(q), L = put_env_on_right_site(q^+) - for a batch:
    A (2,D,D) shared or (n,2,D,D), candidates B (n,2,D,D), q (n,D,D); the norm of q drops out.  Device: `qmps_overlap_amplitude`."""
    from . import _runtime
    B = np.asarray(B, dtype=complex)
    q = np.asarray(q, dtype=complex)
    D = B.shape[-1]
    eng = _runtime.engine(D, len(B))
    eng.set_tensors(B)
    eng.overlap_set(A, WW)
    return eng.overlap_amplitudes(q)


def get_overlap(p1, p2, gate=gate, egate=egate, initial=None, options=None):
    """time_evolve_tools.py:95-131: minimise -2 |psi[0]| of the overlap circuit (W = 1) over an environment r = (rs[:4] + i rs[4:]) put
    on both outer sites, Nelder-Mead from `initial` (random if None); returns the minimum.  The reference rotates r by a phase
    (xmps `rotate_to_hermitian`) and normalises it: neither moves |psi[0]|.  (What it converges to is the numerical radius of the
    mixed two-site transfer map, >= |x|^2 of `get_overlap_exact`; equal when the map is normal.)  `options`: scipy's, default
    {'disp': True} as in the reference."""
    from scipy.optimize import minimize
    from . import _runtime
    from .represent import unitary
    from .tools import unitary_to_tensor
    initial = np.random.randn(8) if initial is None else np.asarray(initial, dtype=float)
    A = unitary_to_tensor(unitary(gate(p1)))
    B = unitary_to_tensor(unitary(gate(p2)))
    eng = _runtime.engine(2, 1)
    eng.set_tensors(B[None])
    eng.overlap_set(A, np.eye(4))

    def obj(rs):
        r = (rs[:4] + 1j * rs[4:]).reshape(1, 2, 2)
        return -2.0 * float(np.abs(eng.overlap_amplitudes(r)[0]))

    res = minimize(obj, initial, method='Nelder-Mead', options={'disp': True} if options is None else options)
    return res.fun
