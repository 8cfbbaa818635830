#!/usr/bin/env python3
"""Variational energy against bond dimension - the reference's `scripts/bond_dimension.py` (XY chain, `NonSparseFullEnergyOptimizer` with the
full SU(2D) parameterisation at D = 2, 4, 8[, 16], every optimum embedded as the starting point of the next bond dimension) through the
drop-in modules:

    python examples/bond_dimension.py [--Ds 2 4 8] [--maxiter 150]

The reference runs Nelder-Mead (tol 1e-5) over up to 1 023 parameters, every energy a cirq simulation; here the optimiser is BFGS with the
2 P central-difference neighbours of an iterate evaluated as ONE device batch - the unitaries exp(-i sum p_k G_k / 2) built on the device
(`qmps_energy_batch_su`).  Prints D, the energy per site and the exact value -4/pi of H = sum XX + YY."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmps_amd.ground_state import Hamiltonian, NonSparseFullEnergyOptimizer, embed_bond_dimension  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--Ds', type=int, nargs='+', default=[2, 4, 8])
    ap.add_argument('--maxiter', type=int, default=150)
    ap.add_argument('--seed', type=int, default=2)
    args = ap.parse_args(argv)
    XY = Hamiltonian({'XX': 1, 'YY': 1}).to_matrix()
    rng = np.random.default_rng(args.seed)
    guess, es = None, []
    for D in args.Ds:
        x0 = rng.standard_normal((2 * D) ** 2 - 1) if guess is None else guess
        opt = NonSparseFullEnergyOptimizer(XY, D, initial_guess=x0)
        opt.change_settings({'verbose': False, 'store_values': False, 'maxiter': args.maxiter, 'tol': 1e-7})
        e_start = opt.objective_function(x0)
        t0 = time.perf_counter()
        res = opt.optimize_restarts(x0[None], method='BFGS')[0]
        es.append((D, e_start, float(res.fun), int(res.nit), time.perf_counter() - t0))
        print(f'D = {D:2d}: {len(x0):4d} parameters, start {e_start:+.6f} -> {res.fun:+.8f} after {res.nit} BFGS iterations ({es[-1][4]:.1f} s);  exact {-4 / np.pi:+.8f}')
        guess = embed_bond_dimension(res.x)           # scripts/bond_dimension.py:50
    return es


if __name__ == '__main__':
    main()
