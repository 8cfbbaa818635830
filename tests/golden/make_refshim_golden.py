#!/usr/bin/env python3
"""Generate tests/golden/refshim_golden.npz: outputs of the REFERENCE'S OWN Python, executed in the build container.

    python tests/golden/make_refshim_golden.py          (needs /root/reference; never runs on the GPU box)

Where tests/golden/make_golden.py imports the reference with arithmetic-free stubs (and can therefore only reach its pure
numpy helpers), this script runs the reference's circuit-level code - the gate classes' `_decompose_` lists, `State`,
`SparseFullEnergyOptimizer.objective_function_exact_environment / _opt_environment`, `NonSparseFull[TwoSite]EnergyOptimizer.
objective_function`, the overlap objectives `scripts/loschmidt.py:obj` and `qmps/new_time_evolve.py:obj`, the embeddings
`put_env_on_left_site / put_env_on_right_site / get_env_off_*`, and the drivers `qmps.tools.double_rotosolve`,
`qmps.rotosolve.rotosolve / double_rotosolve`, and the time-evolution loop (`minimize(obj, params, (A_, WW))` step after step) - on top of the documented-convention stand-ins of tests/golden/cirq_shim.py
(read its header for exactly what the stand-ins supply: named-gate matrices, a big-endian state-vector pass, a dense
eigen-solve for xmps's fixed points).  Arrays are named `refshim_*` so that no reader mistakes them for outputs of the
reference on top of the real cirq / xmps; inputs carry no prefix.

No reference source text is stored: inputs and numeric outputs only.
"""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
from scipy.linalg import expm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def haar(rng, n, B):
    z = rng.standard_normal((B, n, n)) + 1j * rng.standard_normal((B, n, n))
    return np.stack([np.linalg.qr(x)[0] for x in z])


def main():
    assert os.path.isdir(REF), 'reference checkout not mounted - fixtures can only be regenerated in the build container'
    import cirq_shim
    cirq = cirq_shim.install()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'scripts'))
    from qmps import ground_state as rgs
    from qmps import represent as rrep
    from qmps import tools as rtools
    from qmps import time_evolve_tools as rtet
    from qmps import rotosolve as rroto
    import loschmidt as rlos                      # /root/reference/scripts/loschmidt.py (module level: definitions only)

    out = {}
    rng = np.random.default_rng(20261003)
    h_tfim = rgs.Hamiltonian({'ZZ': -1, 'X': 1}).to_matrix()
    h_xxz = rgs.Hamiltonian({'XX': 1, 'YY': 1, 'ZZ': 0.5}).to_matrix()
    out['h_tfim'], out['h_xxz'] = h_tfim, h_xxz

    # ---------------------------------------------------------------------------------------------------------
    # (1) a-2 / a-7: ansatz classes -> unitary (represent.py:268-423) and the exact-environment objective
    #     (ground_state.py:150-168) for every class x bond dimension the library builds on the device
    # ---------------------------------------------------------------------------------------------------------
    def energy_case(tag, cls, D, P, h, n):
        prm = rng.standard_normal((n, P))
        Us, Es = [], []
        for p in prm:
            opt = rgs.SparseFullEnergyOptimizer(h, D, 1, state_tensor=cls, initial_guess=p.copy())
            buf = io.StringIO()
            with redirect_stdout(buf):
                Es.append(opt.objective_function(p.copy()))
            assert 'LinAlgError' not in buf.getvalue(), tag
            Us.append(cirq.unitary(cls(D, p)))
        out[f'params_{tag}'] = prm
        out[f'refshim_U_{tag}'] = np.stack(Us)
        out[f'refshim_E_{tag}'] = np.array(Es)

    C = rrep.ShallowCNOTStateTensor
    energy_case('cnot_D2_d1', C, 2, 2, h_tfim, 6)                    # BASELINE.json configs[0] / [1]
    energy_case('cnot_D2_d2', C, 2, 4, h_tfim, 6)
    energy_case('cnot_D4_d2', C, 4, 4, h_tfim, 6)                    # configs[2]'s ansatz family
    energy_case('cnot_D8_d3_xxz', C, 8, 6, h_xxz, 4)                 # configs[3]
    energy_case('cnot_D16_d4', C, 16, 8, h_tfim, 2)                  # configs[4]'s ansatz
    energy_case('qaoa_D4_d2', rrep.ShallowQAOAStateTensor, 4, 4, h_tfim, 4)
    energy_case('cnot3_D4_d2', rrep.ShallowCNOTStateTensor3, 4, 6, h_tfim, 4)
    energy_case('nonuniform_D2_d2', rrep.ShallowCNOTStateTensor_nonuniform, 2, 8, h_tfim, 4)
    energy_case('nonuniform_D4_d2', rrep.ShallowCNOTStateTensor_nonuniform, 4, 12, h_tfim, 4)
    energy_case('exactafter4_D2_d2', rrep.ExactAfter4, 2, 12, h_tfim, 4)
    energy_case('exactafter4_D4_d2', rrep.ExactAfter4, 4, 12, h_tfim, 4)
    energy_case('full_D2', rrep.ShallowFullStateTensor, 2, 15, h_tfim, 6)
    sg = rng.standard_normal((4, 6))
    out['params_stategate'] = sg
    out['refshim_U_stategate'] = np.stack([cirq.unitary(rrep.StateGate(p)) for p in sg])

    # the full register of State(U, V, 2) (represent.py:258-262) for D = 2: amplitudes, not only the energy
    p = out['params_cnot_D2_d2'][0]
    U = rrep.ShallowCNOTStateTensor(2, p)
    V = rrep.FullEnvironment(rtools.get_env_exact(cirq.unitary(U)))
    qbs = cirq.LineQubit.range(4)
    out['refshim_state_psi_cnot_D2_d2_0'] = cirq.Simulator().simulate(cirq.Circuit().from_ops(rrep.State(U, V, 2)(*qbs))).final_state
    out['refshim_state_V_cnot_D2_d2_0'] = cirq.unitary(V)

    # ---------------------------------------------------------------------------------------------------------
    # (2) a-8 / a-9: NonSparseFullEnergyOptimizer and the two-site unit cell (ground_state.py:251-266, 291-331).
    #     xmps.spin.SU is not in the reference: a look-up stands in for it, so the CIRCUIT is pinned, the SU map is not.
    # ---------------------------------------------------------------------------------------------------------
    table = {}
    rgs.SU = lambda v, n: table[(float(v[0]), n)]
    for D in (2, 4):
        Us = haar(rng, 2 * D, 4)
        Es = []
        for k, u in enumerate(Us):
            table[(float(k), 2 * D)] = u
            opt = rgs.NonSparseFullEnergyOptimizer(h_tfim, D, initial_guess=np.full((2 * D) ** 2 - 1, float(k)))
            Es.append(opt.objective_function(np.full((2 * D) ** 2 - 1, float(k))))
        out[f'U_nonsparse_D{D}'] = Us
        out[f'refshim_E_nonsparse_D{D}'] = np.array(Es)
    U1s, U2s = haar(rng, 4, 4), haar(rng, 4, 4)
    Es = []
    for k in range(4):
        table[(100.0 + k, 4)], table[(200.0 + k, 4)] = U1s[k], U2s[k]
        opt = rgs.NonSparseFullTwoSiteEnergyOptimizer(h_tfim)
        Es.append(opt.objective_function(np.concatenate([np.full(15, 100.0 + k), np.full(15, 200.0 + k)])))
    out['U1_cell'], out['U2_cell'] = U1s, U2s
    out['refshim_E_cell'] = np.array(Es)

    # ---------------------------------------------------------------------------------------------------------
    # (3) a-12: variational-environment objective (ground_state.py:170-229), 30 angles
    # ---------------------------------------------------------------------------------------------------------
    p30 = rng.standard_normal((4, 30))
    opt = rgs.SparseFullEnergyOptimizer(h_tfim, 2, 2, optimize_environment=True, initial_guess=p30[0].copy())
    out['params_optenv'] = p30
    out['refshim_optenv'] = np.array([opt.objective_function(q.copy()) for q in p30])

    # ---------------------------------------------------------------------------------------------------------
    # (4) f-3: environment embeddings (time_evolve_tools.py:38-74) and the overlap objectives
    #     scripts/loschmidt.py:209-239 (6 qubits, ShallowCNOT gate) and qmps/new_time_evolve.py:193-221 (5 qubits, ShallowFull)
    # ---------------------------------------------------------------------------------------------------------
    q = rng.standard_normal((6, 2, 2)) + 1j * rng.standard_normal((6, 2, 2))
    out['embed_q'] = q
    L_, R_ = [rtet.put_env_on_left_site(x, ret_n=True) for x in q], [rtet.put_env_on_right_site(x, ret_n=True) for x in q]
    out['refshim_put_left'] = np.stack([a for a, _ in L_])
    out['refshim_put_left_n'] = np.array([n for _, n in L_])
    out['refshim_put_right'] = np.stack([a for a, _ in R_])
    out['refshim_put_right_n'] = np.array([n for _, n in R_])
    out['refshim_off_left'] = np.stack([rtet.get_env_off_left_site(a) for a, _ in L_])
    out['refshim_off_right'] = np.stack([rtet.get_env_off_right_site(a) for a, _ in R_])

    last = {}
    sim_cls = cirq.Simulator

    class Recording(sim_cls):
        def simulate(self, circuit):
            res = super().simulate(circuit)
            last['psi'] = res.final_state
            return res

    T = np.linspace(0, 6, 300)
    dt = T[1] - T[0]
    WW_l = expm(-1j * rgs.Hamiltonian({'ZZ': -1, 'X': 0.2}).to_matrix() * 2 * dt)          # scripts/loschmidt.py:342
    WW_n = expm(-1j * h_tfim * (1.0 / 9))                                                  # qmps/new_time_evolve.py:254-267
    out['WW_loschmidt'], out['WW_nte'] = WW_l, WW_n
    # The 6-qubit objective of scripts/loschmidt.py:209-239 with its own gate (ShallowCNOTStateTensor(2, v), :204-207) and with
    # the gate qmps/new_time_evolve.py:186-187 parameterises (ShallowFullStateTensor, 15 angles) swapped in through the
    # module-level name `gate`, which is how the reference itself switches ansatz.  NOT generated: the 5-qubit variant
    # qmps/new_time_evolve.py:193-221 - it feeds R = put_env_on_left_site(r) the state |00>, so its amplitude reads the
    # null_space() completion rows of that matrix (arbitrary basis): under these stand-ins it returns -0.34 .. -0.70 at
    # B = A, W = 1 where an overlap objective must give -1.  (DESIGN.md section 2 records the probe.)
    rlos.cirq.Simulator = Recording
    for name, gate_cls, P, WW in (('loschmidt', rrep.ShallowCNOTStateTensor, 8, WW_l), ('loschmidt_full', rrep.ShallowFullStateTensor, 15, WW_n)):
        rlos.gate = (lambda cls: (lambda v, symbol='U': cls(2, v)))(gate_cls)
        p0 = rng.standard_normal((4, P))
        # candidates near the reference state (what a time step sees) and far from it
        cand = np.concatenate([p0 + 0.05 * rng.standard_normal((4, P)), rng.standard_normal((4, P))])
        ref_idx = np.concatenate([np.arange(4), np.arange(4)])
        for wname, W in (('W', WW), ('I', np.eye(4, dtype=complex))):
            f, amp = [], []
            for pc, k in zip(cand, ref_idx):
                A = rtools.unitary_to_tensor(cirq.unitary(gate_cls(2, p0[k])))
                f.append(rlos.obj(pc.copy(), A, W))
                amp.append(last['psi'][0])
            out[f'refshim_{name}_obj_{wname}'] = np.array(f)
            out[f'refshim_{name}_psi0_{wname}'] = np.array(amp)            # objective = -sqrt(2 |psi[0]|)
        out[f'{name}_p_ref'], out[f'{name}_p_cand'], out[f'{name}_ref_idx'] = p0, cand, ref_idx
    rlos.cirq.Simulator = sim_cls

    # ---------------------------------------------------------------------------------------------------------
    # (5) a-10 / f-2: the rotosolve drivers, run on the reference's own objective
    # ---------------------------------------------------------------------------------------------------------
    fits = []
    real_ms = rtools.minimize_scalar

    def recording_ms(f, bounds=None, **kw):
        res = real_ms(f, bounds=bounds, **kw)
        xs = np.array([0.0, 0.7, 1.9, -1.1, 2.6, -2.4])
        M = np.stack([np.sin(2 * xs), np.cos(2 * xs), np.sin(xs), np.cos(xs)], axis=1)
        coef = np.linalg.lstsq(M, np.array([f(x) for x in xs]), rcond=None)[0]      # f = a sin 2x + b cos 2x + c sin x + d cos x
        fits.append(np.concatenate([coef, [res.x, res.fun, res.nfev]]))
        return res

    rtools.minimize_scalar = recording_ms
    rroto.minimize_scalar = recording_ms
    rroto.sinusoids = lambda *a, **k: None           # figure tooling (and it indexes 15 parameters whatever len(params) is)
    rroto.tqdm = lambda x, *a, **k: x               # qmps/rotosolve.py:216 uses tqdm and :175 uses π without defining them
    rroto.π = np.pi                                  # (scripts/rotosolve.py:15, 20, the runnable copy of the same file, has both)

    def run_tools_double(tag, D, depth, h, R, sweeps):
        x0 = rng.standard_normal((R, 2 * depth))
        hist_E = np.zeros((sweeps, R))
        hist_x = np.zeros((sweeps, R, 2 * depth))
        for r in range(R):
            for s in range(1, sweeps + 1):
                opt = rgs.SparseFullEnergyOptimizer(h, D, depth, initial_guess=x0[r].copy())
                buf = io.StringIO()
                with redirect_stdout(buf):
                    res = rtools.double_rotosolve(opt.objective_function, x0[r].copy(), s, False)
                assert 'LinAlgError' not in buf.getvalue(), (tag, r)
                hist_E[s - 1, r] = res.fun
                hist_x[s - 1, r] = res.x
                assert np.allclose(res.history[:s - 1], hist_E[:s - 1, r], atol=1e-13)
        out[f'roto_{tag}_x0'] = x0
        out[f'refshim_droto_{tag}_E'] = hist_E
        out[f'refshim_droto_{tag}_x'] = hist_x

    run_tools_double('D2_d2', 2, 2, h_tfim, 8, 2)
    run_tools_double('D4_d2', 4, 2, h_tfim, 6, 2)
    run_tools_double('D8_d3_xxz', 8, 3, h_xxz, 3, 2)

    def state_fn_factory(cls, D):
        n_q = int(2 + 2 * np.log2(D))
        qb = cirq.LineQubit.range(n_q)

        def state_fn(x):
            U = cls(D, x)
            V = rrep.FullEnvironment(rtools.get_env_exact(cirq.unitary(U)))
            return cirq.Simulator().simulate(cirq.Circuit().from_ops(rrep.State(U, V, 2)(*qb))).final_state
        return state_fn

    def run_old_api(tag, D, depth, h, R, sweeps):
        Hfull = np.kron(np.kron(np.eye(D), h), np.eye(D))
        fn = state_fn_factory(rrep.ShallowCNOTStateTensor, D)
        x0 = rng.standard_normal((R, 2 * depth))
        es1, xs1, es2, xs2 = [], [], [], []
        for r in range(R):
            es, S = rroto.rotosolve(Hfull, fn, x0[r].copy(), (), sweeps)
            es1.append(es)
            xs1.append(np.stack(S))
            e2 = [rroto.double_rotosolve(Hfull, fn, x0[r].copy(), (), s) for s in range(1, sweeps + 1)]
            es2.append(e2[-1][0])
            xs2.append(np.stack([x for _, x in e2]))
        out[f'roto_old_{tag}_x0'] = x0
        out[f'refshim_roto_old_{tag}_E'] = np.array(es1).T                      # (sweeps, R)
        out[f'refshim_roto_old_{tag}_x'] = np.stack(xs1).transpose(1, 0, 2)     # (sweeps, R, P)
        out[f'refshim_droto_old_{tag}_E'] = np.array(es2).T
        out[f'refshim_droto_old_{tag}_x'] = np.stack(xs2).transpose(1, 0, 2)

    run_old_api('D2_d2', 2, 2, h_tfim, 8, 3)
    run_old_api('D4_d2', 4, 2, h_tfim, 6, 2)
    # every fit the reference handed to scipy's minimize_scalar during these runs, with scipy's answer: (a, b, c, d, x, f(x), nfev)
    out['refshim_roto_fits'] = np.stack(fits)

    # ---------------------------------------------------------------------------------------------------------
    # (6) N-2: the reference's time-evolution LOOP - per time step `A_ = tensor(params); res = minimize(obj, params, (A_, WW));
    #     params = res.x` (qmps/new_time_evolve.py:276-292, scripts/loschmidt.py:367-375 with prob = 0) - run here with the reference's
    #     6-qubit objective scripts/loschmidt.py:209-239, its gate and the new_time_evolve gate, scipy's default BFGS (forward
    #     differences, gtol 1e-5): parameters and minima of three consecutive time steps for three starting points per gate.
    #     (A generator of its own: the arrays above do not move when this section changes.)
    # ---------------------------------------------------------------------------------------------------------
    from scipy.optimize import minimize
    rng6 = np.random.default_rng(20261004)
    n_steps = 3
    for name, gate_cls, P, WW in (('loschmidt', rrep.ShallowCNOTStateTensor, 8, WW_l), ('loschmidt_full', rrep.ShallowFullStateTensor, 15, WW_n)):
        rlos.gate = (lambda cls: (lambda v, symbol='U': cls(2, v)))(gate_cls)
        starts = rng6.standard_normal((3, P))
        xs, fs, nits, nfevs, f0s = [], [], [], [], []
        for x0 in starts:
            params = x0.copy()
            hx, hf, hn, hv, h0 = [params.copy()], [], [], [], []
            for _ in range(n_steps):
                A_ = rlos.iMPS([rtools.unitary_to_tensor(cirq.unitary(rlos.gate(params)))]).left_canonicalise()
                h0.append(rlos.obj(params.copy(), A_[0], WW))             # where the step starts (W moved the state away from A_)
                res = minimize(rlos.obj, params, (A_[0], WW))
                params = res.x
                hx.append(params.copy()); hf.append(res.fun); hn.append(res.nit); hv.append(res.nfev)
            xs.append(np.stack(hx)); fs.append(hf); nits.append(hn); nfevs.append(hv); f0s.append(h0)
        out[f'evolve_{name}_x0'] = starts
        out[f'refshim_evolve_{name}_x'] = np.stack(xs)                # (3, n_steps + 1, P): parameters before / after every time step
        out[f'refshim_evolve_{name}_f'] = np.array(fs)                # (3, n_steps): scipy's minimum of every time step
        out[f'refshim_evolve_{name}_f_start'] = np.array(f0s)         # (3, n_steps): the objective at the step's starting point
        out[f'refshim_evolve_{name}_nit'] = np.array(nits)
        out[f'refshim_evolve_{name}_nfev'] = np.array(nfevs)

    # ---------------------------------------------------------------------------------------------------------
    # (7) the variational route of the overlap: `get_overlap` (qmps/time_evolve_tools.py:95-131) - its Nelder-Mead minimum and its
    #     objective closure at recorded points - and `obj_state` (qmps/new_time_evolve.py:223-247).  xmps `rotate_to_hermitian` is not
    #     in /root/reference; the objective -2 |psi[0]| cannot depend on it as long as it rotates its argument by a phase (R carries
    #     r, L carries r^+): run with TWO stand-ins (divide by the phase of the trace; multiply by i) and assert equal minima.
    # ---------------------------------------------------------------------------------------------------------
    from qmps import new_time_evolve as rnte
    rng7 = np.random.default_rng(20261005)
    real_min = rtet.minimize
    seen = {}

    def recording_min(fun, x0, args=(), **kw):
        seen['fun'] = fun
        return real_min(fun, x0, args, **kw)

    rtet.minimize = recording_min
    P1, P2, INIT = rng7.standard_normal((3, 15)), rng7.standard_normal((3, 15)), rng7.standard_normal((3, 8))
    P2[2] = P1[2] + 0.05 * rng7.standard_normal(15)                     # a candidate next to the reference state
    probes = rng7.standard_normal((3, 5, 8))
    mins, vals = [], []
    for k in range(3):
        got = []
        for stand_in in (lambda M: M * np.exp(-1j * np.angle(np.trace(M))), lambda M: 1j * M):
            rtet.rotate_to_hermitian = stand_in
            with redirect_stdout(io.StringIO()):
                got.append(rtet.get_overlap(P1[k], P2[k], initial=INIT[k].copy()))
        assert abs(got[0] - got[1]) < 1e-12, got
        mins.append(got[0])
        vals.append([seen['fun'](x.copy()) for x in probes[k]])             # the closure of the last run: -2 |psi[0]| at given rs
    rtet.minimize = real_min
    out['varenv_p1'], out['varenv_p2'], out['varenv_initial'], out['varenv_probe_rs'] = P1, P2, INIT, probes
    out['refshim_get_overlap_min'] = np.array(mins)
    out['refshim_get_overlap_obj'] = np.array(vals)
    PS = np.concatenate([P2, rng7.standard_normal((3, 6))], axis=1)          # 15 gate angles + 6 StateGate angles
    psis = []
    for k in range(3):
        A = rtools.unitary_to_tensor(cirq.unitary(rnte.gate(P1[k])))
        psis.append(rnte.obj_state(PS[k].copy(), A, WW_n if k else np.eye(4)))
    out['nsphere_v'] = rng7.standard_normal((4, 7))
    out['refshim_nsphere'] = np.stack([rtet.Nsphere(v) for v in out['nsphere_v']])           # time_evolve_tools.py:25-36
    out['obj_state_p'] = PS
    out['refshim_obj_state_psi'] = np.stack(psis)                            # (3, 32); entries 16.. read null_space() completion rows of L

    # ---------------------------------------------------------------------------------------------------------
    # (8) the reference's OWN in-file assertion suite, executed over the stand-ins: qmps/new_time_evolve.py:run_tests (:50-184)
    #     - embeddings round trips, unitarity, and the circuit identities 2 psi[0] = tr(g r), x tr(g r),
    #     x^2 tr(g r), tr(g l*), x tr(g l*), x^2 tr(g l*), x^2 tr(l^+ r) for random left-canonical tensors.  They passed on the authors'
    #     cirq / xmps; that they pass HERE pins what the stand-ins had only documented: simulator endianness, H / CNOT / Pauli matrices,
    #     the right AND left fixed-point conventions of `Map` (the left one was wrong until this section existed: cirq_shim._dominant).
    #     Two numpy-1.x idioms of the 2019 code are bridged: np.prod((matrix, scalar)) multiplied them; nothing else is touched.
    # ---------------------------------------------------------------------------------------------------------
    class Numpy1x:
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def prod(a, *args, **kw):
            if isinstance(a, tuple) and len(a) == 2 and np.ndim(a[0]) == 2 and np.ndim(a[1]) == 0:
                return a[0] * a[1]
            return np.prod(a, *args, **kw)

    state = np.random.get_state()
    np.random.seed(20261006)
    rnte.np = Numpy1x()
    rnte.run_tests(6)                                   # raises AssertionError on the first identity that fails
    rnte.np = np
    np.random.set_state(state)
    # (scripts/loschmidt.py:test is the same suite with a typo of its own - `*cirq.H(qbs[1])` at :130 unpacks an Operation - and cannot run anywhere)
    out['refshim_reference_selftests_passed'] = np.array([6])           # iterations of run_tests that passed

    path = os.path.join(HERE, 'refshim_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays,', len(fits), 'scalar-minimiser calls recorded')


if __name__ == '__main__':
    main()
