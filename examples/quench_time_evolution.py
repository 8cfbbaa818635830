#!/usr/bin/env python3
"""Time evolution after a quench with the reference's loop on an MI355X.

The `__main__` block of `qmps/new_time_evolve.py` / `scripts/loschmidt.py` of fergusfinn/qmps - per time step
`A_ = tensor(params); params = minimize(obj, params, (A_, WW)).x`, recording one-site expectation values and the Loschmidt echo -
through qmps_amd.new_time_evolve (same `obj(p, A, WW)` signature; `evolve` runs every BFGS iteration of every time step of every
trajectory in one C call):

    python examples/quench_time_evolution.py [--D 2] [--steps 40] [--dt 0.05] [--trajectories 4]

Each trajectory starts from a random product-like state of the ansatz family; H = -sum ZZ + g sum X."""
import argparse
import os
import sys

import numpy as np
from scipy.linalg import expm

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmps_amd import new_time_evolve as NT, represent as R  # noqa: E402
from qmps_amd.ground_state import Hamiltonian  # noqa: E402
from qmps_amd.tools import unitary_to_tensor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--D', type=int, default=2)
    ap.add_argument('--g', type=float, default=0.2)
    ap.add_argument('--dt', type=float, default=0.05)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--trajectories', type=int, default=4)
    ap.add_argument('--seed', type=int, default=3)
    args = ap.parse_args()
    D = args.D
    cls = R.ShallowFullStateTensor if D == 2 else R.ShallowCNOTStateTensor
    P = 15 if D == 2 else 2 * int(np.log2(D))
    WW = expm(-1j * Hamiltonian({'ZZ': -1.0, 'X': args.g}).to_matrix() * args.dt)
    rng = np.random.default_rng(args.seed)
    X0 = 0.3 * rng.standard_normal((args.trajectories, P))
    H, info = NT.evolve(X0, WW, args.steps, method='BFGS', D=D, state_tensor=cls, tol=1e-12, return_info=True)
    f_end = np.array([f[-1] for f in info['fun']])               # (steps, T): -sqrt|eta| of every time step, -1 = the step lost nothing
    tensor = lambda p: unitary_to_tensor(R.unitary(R.build_gate(cls, D, p)))      # noqa: E731
    echo = np.array([[NT.loschmidt_overlap(tensor(H[k][t]), tensor(H[0][t])) for t in range(args.trajectories)] for k in range(args.steps + 1)]) if D == 2 else None
    print(f'D = {D}, {args.trajectories} trajectories, {args.steps} steps of dt = {args.dt}: worst step fidelity 1 - {1.0 - (f_end.min() ** 2):.2e}')
    if echo is not None:
        for k in range(0, args.steps + 1, max(1, args.steps // 8)):
            print(f't = {k * args.dt:5.2f}   Loschmidt echo per site ' + '  '.join(f'{e:.6f}' for e in echo[k]))
    return H, f_end, echo


if __name__ == '__main__':
    main()
