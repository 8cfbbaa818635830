#!/usr/bin/env python3
"""Variational phase diagram of the transverse-field Ising chain - the reference's `plot_phase_diagram`
(scripts/ground_state_finding.py:166-200: 21 couplings x ansatz sizes x random restarts, one scipy BFGS after the other, every energy a
cirq simulation) with its loops turned into the batch axis of the MI355X kernels:

    python examples/phase_diagram_tfim.py [--points 21] [--restarts 16] [--D 2] [--depth 2]

`ground_state_sweep` advances all points x restarts minimisations in ONE lock-step BFGS; a launch returns the energies of the two terms
(-ZZ and (XI + IX)/2) of every candidate and each trajectory combines them with its own coupling.  Prints lambda, the variational energy
per site (best restart), the exact value and the gap."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmps_amd.ground_state import Hamiltonian, ground_state_sweep  # noqa: E402
from qmps_amd.represent import ShallowCNOTStateTensor  # noqa: E402


def exact_energy(g, n=200001):
    k = np.linspace(0.0, np.pi, n)
    return -np.trapezoid(np.sqrt(1.0 + g * g - 2.0 * g * np.cos(k)), k) / np.pi


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=21)
    ap.add_argument('--restarts', type=int, default=16)
    ap.add_argument('--D', type=int, default=2)
    ap.add_argument('--depth', type=int, default=2)
    ap.add_argument('--seed', type=int, default=3)
    args = ap.parse_args(argv)
    lams = np.linspace(0.0, 2.0, args.points)
    terms = [Hamiltonian({'ZZ': -1.0}).to_matrix(), Hamiltonian({'X': 1.0}).to_matrix()]
    coef = np.stack([np.ones_like(lams), lams], axis=1)
    t0 = time.perf_counter()
    out = ground_state_sweep(terms, coef, D=args.D, depth=args.depth, state_tensor=ShallowCNOTStateTensor, restarts=args.restarts,
                             rng=np.random.default_rng(args.seed), maxiter=300)
    dt = time.perf_counter() - t0
    exact = np.array([exact_energy(g) for g in lams])
    for g, e, x in zip(lams, out['energy'], exact):
        print(f'lambda {g:5.2f}   E_var {e:+.8f}   E_exact {x:+.8f}   gap {e - x:.2e}')
    print(f'{args.points} couplings x {args.restarts} restarts = {args.points * args.restarts} BFGS minimisations in lock-step: '
          f'{out["nit"]} iterations, {out["nfev"]} energy evaluations, {dt:.2f} s')
    return lams, out, exact


if __name__ == '__main__':
    main()
