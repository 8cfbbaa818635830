"""Host-side mirror of the pieces of `qmps/time_evolve_tools.py` the energy path touches.

  merge                    qmps/time_evolve_tools.py:20-23   (any D here; the reference hard-codes D = 2)
  put_env_on_left_site     qmps/time_evolve_tools.py:38-53
  get_env_off_left_site    qmps/time_evolve_tools.py:55-57
  put_env_on_right_site    qmps/time_evolve_tools.py:59-70
  get_env_off_right_site   qmps/time_evolve_tools.py:72-74

The overlap objective built on them (time evolution) is a "next" row.
"""
import numpy as np
from scipy.linalg import null_space

_SWAP = np.eye(4)[[0, 2, 1, 3]]


def merge(A, B):
    """-A- -B- -> -AB-: merge(A, B)[2 s1 + s2] = A[s1] @ B[s2]."""
    D = A.shape[1]
    return np.einsum('sij,tjk->stik', A, B).reshape(A.shape[0] * B.shape[0], D, D)


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
