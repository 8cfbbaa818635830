"""Prototype (numpy): power method with momentum for the evolve workload's fixed-point solves at D = 16.
Spectra of the mixed transfer map for candidates near their reference (what a BFGS iterate is), rounds of the plain power method
against x_{k+1} = T x_k - beta x_{k-1} with beta from an estimate of |lambda_2|."""
import os, sys
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from oracle import qmps_oracle as O
import overlap_cases as OC
import bench

D, P = 16, 8
rng = np.random.default_rng(3)
WW = expm(-1j * 0.05 * bench.tfim_h(1.0))
tens = lambda p: O.unitary_to_tensor(O.shallow_cnot_unitary(D, p))


def power(M, x0, tol=1e-8, momentum=None, maxit=2000):
    x = x0 / np.linalg.norm(x0)
    xp = np.zeros_like(x)
    beta = 0.0
    res_hist = []
    for k in range(1, maxit + 1):
        n = M @ x
        eta = np.vdot(x, n)
        res = np.linalg.norm(n - eta * x)
        res_hist.append(res)
        if res < tol:
            return k, eta
        if momentum == 'auto' and k >= 4 and beta == 0.0:
            # estimate |lambda_2 / lambda_1| from the residual decay
            rho = (res_hist[-1] / res_hist[-3]) ** 0.5
            if rho < 0.98:
                beta = (rho * abs(eta)) ** 2 / 4
        y = n - beta * xp
        nrm = np.linalg.norm(y)
        xp = x / nrm
        x = y / nrm
    return maxit, eta


rows = []
for trial in range(40):
    x = rng.standard_normal(P)
    A = tens(x)
    step = rng.standard_normal(P)
    step *= 10 ** rng.uniform(-3, -1) / np.linalg.norm(step)
    B0, B1 = tens(x + step), tens(x + 1.5 * step)       # previous iterate (warm start) and the next one
    M0, M1 = OC.dense_map(A, B0, WW), OC.dense_map(A, B1, WW)
    w0, v0 = np.linalg.eig(M0)
    r0 = v0[:, np.argmax(abs(w0))]
    w1 = np.linalg.eigvals(M1)
    order = np.argsort(-abs(w1))
    l1, l2 = w1[order[0]], w1[order[1]]
    kp, _ = power(M1, r0)
    km, em = power(M1, r0, momentum='auto')
    rows.append((abs(l2 / l1), np.angle(l2 / l1), kp, km, abs(em - l1)))
rows = np.array(rows)
print('|l2/l1| mean %.3f max %.3f; phase of l2/l1: mean |phi| %.2f' % (rows[:, 0].mean(), rows[:, 0].max(), np.abs(rows[:, 1]).mean()))
print('rounds plain: mean %.1f max %d; momentum(auto): mean %.1f max %d; worst |eta error| %.1e' % (rows[:, 2].mean(), rows[:, 2].max(), rows[:, 3].mean(), rows[:, 3].max(), rows[:, 4].max()))
for r in rows[:12]:
    print('  rho %.3f phase %+.2f plain %3d momentum %3d' % (r[0], r[1], r[2], r[3]))
