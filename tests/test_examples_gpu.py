"""-m gpu: the two example scripts (examples/) - the reference's `__main__` workflows through the drop-in modules - run end to end and
produce physics: variational energies above the exact one, the universal D = 2 gate at the reference's quoted D = 2 optimum; a quench evolution whose
every time step keeps the state (objective -> -1) while the Loschmidt echo decays from 1."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'examples', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ground_state_example(monkeypatch):
    mod = load('ground_state_tfim')
    monkeypatch.setattr(sys, 'argv', ['ground_state_tfim.py', '--restarts', '32', '--sweeps', '12'])
    out, e0 = mod.main()
    assert abs(e0 - (-4.0 / np.pi)) < 1e-9                       # TFIM at g = 1: -4/pi
    for label, (e_roto, e_bfgs) in out.items():
        assert e_bfgs <= e_roto + 1e-10, label                   # the polish never loses
        assert e0 < e_bfgs < -1.2, label                         # variational bound (tests/test_ground_state.py:101-102 of the reference)
    # scripts/noisy_optimization.py:93 draws D2_gse = -1.269909412573 as "the D = 2 line"; the optimum of the D = 2 manifold at g = 1 is
    # -1.2725424859 (tests/test_oracle.py::test_d2_optimum_of_the_tfim: the universal gate AND a free 4 x 2 isometry, CPU oracle) -
    # the device-backed optimiser finds it
    assert out['ShallowFull D=2 (universal)'][1] < -1.269909412573
    assert abs(out['ShallowFull D=2 (universal)'][1] - (-1.2725424859)) < 1e-6
    assert out['ShallowFull D=2 (universal)'][1] <= out['ShallowCNOT D=2 depth 2'][1] + 1e-9      # the universal gate contains the shallow one


def test_quench_example(monkeypatch):
    mod = load('quench_time_evolution')
    monkeypatch.setattr(sys, 'argv', ['quench_time_evolution.py', '--steps', '16', '--trajectories', '3'])
    H, f_end, echo = mod.main()
    assert H.shape == (17, 3, 15) and np.all(np.isfinite(H))
    assert f_end.max() < -0.999                                  # every time step is represented inside the manifold
    assert np.abs(echo[0] - 1.0).max() < 1e-9 and np.all(echo[-1] < echo[0]) and np.all(echo > 0.0)
    monkeypatch.setattr(sys, 'argv', ['quench_time_evolution.py', '--D', '4', '--steps', '4', '--trajectories', '2'])
    H4, f4, _ = mod.main()
    assert H4.shape == (5, 2, 4) and f4.max() < -0.99


def test_phase_diagram_example():
    """examples/phase_diagram_tfim.py: the reference's `plot_phase_diagram` loops (scripts/ground_state_finding.py:166-200) as one lock-step:
    every point above the exact energy, the classical limit lambda = 0 exact (a product state is in the manifold), the gap largest near the
    critical point; the same through the D = 4 kernels."""
    mod = load('phase_diagram_tfim')
    lams, out, exact = mod.main(['--points', '9', '--restarts', '12'])
    gap = out['energy'] - exact
    assert np.all(gap > -1e-10) and gap[0] < 1e-8 and np.all(gap < 0.05)        # (the depth-2 D = 2 circuit: 2.6e-2 at lambda = 1.25)
    assert 0.5 <= lams[np.argmax(gap)] <= 1.5
    # (the shallow families of different D are not nested and have local minima: no ordering between them is asserted - 12 restarts of the D = 4
    # depth-3 circuit end at -1.2590 at lambda = 1 where the D = 2 depth-2 circuit reaches its optimum -1.2655)
    _, out4, exact4 = mod.main(['--points', '3', '--restarts', '12', '--D', '4', '--depth', '3'])
    gap4 = out4['energy'] - exact4
    assert np.all(gap4 > -1e-10) and gap4[0] < 1e-8 and np.all(gap4 < 0.05)


def test_bond_dimension_example():
    """examples/bond_dimension.py (scripts/bond_dimension.py of the reference): the optimum of D embedded at 2 D starts within a few per cent of
    where D ended (eps = 4e-2 off the embedded state), every bond dimension ends at or below the previous one and above the exact -4/pi."""
    mod = load('bond_dimension')
    es = mod.main(['--Ds', '2', '4', '--maxiter', '120'])
    (d2, s2, e2, _, _), (d4, s4, e4, _, _) = es
    assert (d2, d4) == (2, 4) and -4 / np.pi < e4 <= e2 + 1e-9 < -0.99          # (this seed: -1.0000 at D = 2, -1.2500 at D = 4; exact -1.2732)
    assert abs(s4 - e2) < 0.1 and e4 < s4
