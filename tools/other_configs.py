#!/usr/bin/env python3
"""Step times of the other BASELINE.json configurations (parity-test cases, not bench lines): HIP events on the
context stream around `reps` steps (environment + energy + device-side sum) after a clock-settle phase.
usage (GPU box): python tools/other_configs.py > gpurun_out/other_configs.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from qmps_amd import EnergyEngine


def run(D, B, solver, reps=200):
    A = bench.haar_tensors(7 + D + B, D, B)
    h = bench.tfim_h(1.0)
    with EnergyEngine(D, B) as eng:
        eng.set_tensors(A)
        eng.set_hamiltonian(h)
        for _ in range(10):
            eng.probe_fp64_tflops()

        def step():
            eng.launch(B, solver=solver, store_env=False if (D == 4 and solver == 'direct') else True,
                       accumulate_cost=not (D == 4 and solver == 'squaring'))
            eng.cost_launch(B)
        for _ in range(50):
            step()
        eng.sync()
        eng.timer_begin()
        for _ in range(reps):
            step()
        ms = eng.timer_end() / reps
        E, it, st = eng.results(B)
        return {'D': D, 'B': B, 'solver': solver, 'ms_per_step': ms, 'evals_per_s': B / (ms * 1e-3),
                'mean_iters': float(it.mean()), 'not_ok': int((st != 0).sum())}


if __name__ == '__main__':
    out = []
    for D, B, solver in ((2, 4096, 'direct'), (2, 65536, 'direct'), (2, 4096, 'squaring'), (2, 65536, 'squaring'),
                         (8, 96, 'direct'), (8, 768, 'direct'), (8, 65536, 'direct'),
                         (8, 96, 'plain'), (8, 768, 'plain'), (8, 65536, 'plain'),
                         (16, 96, 'squaring'), (16, 768, 'squaring'), (16, 16384, 'squaring'),
                         (4, 4096, 'direct'), (4, 65536, 'direct'), (4, 65536, 'squaring')):
        r = run(D, B, solver, reps=200 if B <= 4096 else 30)
        out.append(r)
        print(json.dumps(r), flush=True)
