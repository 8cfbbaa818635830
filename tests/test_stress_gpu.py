"""-m gpu: short runs of the randomised stress scripts of round 5 (profiles/experiments/r05/stress_*.py: device against the oracle on random inputs, a third of
them at symmetric angles / special unitaries - the inputs that found five silent errors of the fixed-point solves, profiles/EXPERIMENTS.md).  The full
campaign is minutes of GPU time per script; here a few dozen cases each with fixed seeds, so that the classes of error it found stay found."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'profiles', 'experiments', 'r05', script)] + [str(a) for a in args], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().split('\n')[-1])


def test_energy_path_has_no_silent_error():
    r = run('stress_energy.py', 25, 21)
    assert r['evals'] > 600 and r['anomalies'] == 0, r['bad']
    assert r['max_dE_unique_status0'] < 1e-10


def test_overlap_path_has_no_silent_error():
    r = run('stress_overlap.py', 80, 22)
    # what may be flagged: numerically nilpotent maps (numpy's eigenvalues are noise), crowded rings and tied pairs (status 1 / a member within 0.3 % of
    # the top - the reference's ARPACK does no better: profiles/EXPERIMENTS.md); what may NOT: a status-0 eigenvalue that is clearly not the dominant one
    silent = [b for b in r['bad'] if b.get('what', '').startswith('status 0') and b.get('sep', 0.0) > 0.01 and max(abs(b['dominant'][0]), abs(b['dominant'][1])) > 1e-6]
    assert r['evals'] > 900 and not silent, silent
    assert r['max_r_residual_status0'] < 1e-10


def test_evolve_device_algebra_equals_the_host_loop_on_random_cases():
    r = run('stress_evolve.py', 120, 23)
    assert r['cases'] == 120 and r['not_identical_or_mismatch'] == 0, r['bad']


def test_device_resident_drivers_record_the_oracles_objectives():
    r = run('stress_evolve_device.py', 80, 24)
    assert r['oracle_checked'] > 500 and max(r['max_d_oracle_dev'], r['max_d_oracle_host']) < 1e-10
    assert r['nan_device_only'] == 0 and r['device_failed_evaluations'] == 0 and r['anomalies'] == 0, r['bad']


def test_brickwall_eigenpairs_follow_the_reference_rule():
    r = run('stress_brickwall.py', 6, 25)
    assert r['max_d_expval2'] < 1e-12 and r['max_d_envmat'] < 1e-13 and r['max_d_eta_status0'] < 1e-8
    assert not [b for b in r['bad'] if b['what'].startswith('status 0')], r['bad']


def test_rotosolve_cell_and_variational_environment():
    r = run('stress_rotosolve.py', 30, 26)
    assert r['final_checked'] > 300 and r['anomalies'] == 0 and r['max_dE_final'] < 1e-9, r['bad']
    c = run('stress_cell2_optenv.py', 3, 27)
    assert c['anomalies'] == 0 and c['cell_max_dE'] < 1e-10 and c['optenv_max_d'] < 1e-10, c['bad']


def test_api_fuzzers_find_no_state_leak():
    """Random call sequences on ONE context against stateless evaluations on a second one (stress_api_state.py: tensors / Hamiltonian terms / windows /
    every launch flavour / cost exchange; stress_api_overlap.py: references from parameters, candidate groups, one-shot masks, warm starts, gradients) -
    the fuzzers that found the stale cost-accumulator terms and the warm start on an unwritten window (round 5)."""
    for D, seed in ((4, 31), (8, 32), (16, 33)):
        r = run('stress_api_state.py', 12, seed, D)
        assert r['checks'] > 60 and r['anomalies'] == 0 and r['max_dE'] < 1e-9, r['bad']
    for D, seed in ((8, 34), (2, 35)):
        r = run('stress_api_overlap.py', 12, seed, D)
        assert r['checks'] > 40 and r['anomalies'] == 0 and r['max_df'] < 1e-10 and r['max_dg'] < 1e-8, r['bad']
    s = run('stress_su.py', 2, 36)
    assert s['anomalies'] == 0 and s['max_dU'] < 1e-12 and s['max_dE'] < 1e-10, s['bad']
