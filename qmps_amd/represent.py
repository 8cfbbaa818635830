"""Host-side mirror of `qmps/represent.py`: state tensors, environments and ansatz gates as plain
unitary-producing objects (no cirq).

The reference expresses everything as `cirq.Gate` subclasses and lets cirq decompose them; here a
gate is any object with `num_qubits()` and either `_unitary_()` or `_decompose_(qubits)` (a list
of `Op`s, possibly nested), and `unitary(gate)` plays the role of `cirq.unitary`.  Registers are
big-endian like cirq's: qubit 0 is the most significant bit of a row/column/amplitude index.

  Tensor / FullStateTensor / FullEnvironment   qmps/represent.py:188-232
  PowerCircuit                                 qmps/represent.py:235-248
  State                                        qmps/represent.py:251-265
  ShallowQAOAStateTensor                       qmps/represent.py:268-285
  ShallowCNOTStateTensor (+_nonuniform, 3)     qmps/represent.py:288-354
  ExactAfter4                                  qmps/represent.py:356-380
  ShallowFullStateTensor                       qmps/represent.py:382-404
  StateGate                                    qmps/represent.py:406-423
  ShallowEnvironment                           qmps/represent.py:425-442
"""
import numpy as np

from .tools import split_2s, split_3s, split_ns

# ---------------------------------------------------------------------------------------------
# a minimal circuit model: ops, unitaries, state vectors
# ---------------------------------------------------------------------------------------------
_SQ2 = 1.0 / np.sqrt(2.0)


class Op:
    """A gate placed on specific qubits (integers)."""
    __slots__ = ('gate', 'qubits')

    def __init__(self, gate, qubits):
        self.gate = gate
        self.qubits = tuple(int(q) for q in qubits)


class Gate:
    """Base class: calling a gate on qubits yields an Op (like cirq's gate.on)."""

    def num_qubits(self):
        raise NotImplementedError

    def __call__(self, *qubits):
        if len(qubits) != self.num_qubits():
            raise ValueError(f'{type(self).__name__} acts on {self.num_qubits()} qubits, got {len(qubits)}')
        return Op(self, qubits)

    on = __call__


class MatrixGate(Gate):
    def __init__(self, matrix):
        self._m = np.asarray(matrix, dtype=complex)
        self._n = int(round(np.log2(self._m.shape[0])))

    def num_qubits(self):
        return self._n

    def _unitary_(self):
        return self._m


def line_qubits(n):
    """Stand-in for cirq.LineQubit.range(n)."""
    return list(range(n))


def rx(t):
    c, s = np.cos(0.5 * t), np.sin(0.5 * t)
    return MatrixGate([[c, -1j * s], [-1j * s, c]])


def ry(t):
    c, s = np.cos(0.5 * t), np.sin(0.5 * t)
    return MatrixGate([[c, -s], [s, c]])


def rz(t):
    return MatrixGate(np.diag([np.exp(-0.5j * t), np.exp(0.5j * t)]))


H = MatrixGate([[_SQ2, _SQ2], [_SQ2, -_SQ2]])
X = MatrixGate([[0, 1], [1, 0]])
CNOT = MatrixGate(np.eye(4)[[0, 1, 3, 2]])
SWAP = MatrixGate(np.eye(4)[[0, 2, 1, 3]])


def _pauli_power(P, t):
    """cirq's P**t for an involutory P: eigenvalue +1 -> 1, eigenvalue -1 -> e^{i pi t}."""
    n = P.shape[0]
    ph = np.exp(1j * np.pi * t)
    return 0.5 * (1 + ph) * np.eye(n) + 0.5 * (1 - ph) * P


_X = np.array([[0, 1], [1, 0]], dtype=complex)
_Y = np.array([[0, -1j], [1j, 0]])
_Z = np.diag([1.0 + 0j, -1.0])


def x_pow(t):
    return MatrixGate(_pauli_power(_X, t))


def xx_pow(t):
    return MatrixGate(_pauli_power(np.kron(_X, _X), t))


def yy_pow(t):
    return MatrixGate(_pauli_power(np.kron(_Y, _Y), t))


def zz_pow(t):
    return MatrixGate(_pauli_power(np.kron(_Z, _Z), t))


def _flatten(ops):
    if isinstance(ops, Op):
        yield ops
    elif ops is not None:
        for o in ops:
            yield from _flatten(o)


def _apply(matrix, qubits, tensor, n):
    """Apply a k-qubit matrix to axes `qubits` of a (2,)*n (+ trailing) tensor."""
    k = len(qubits)
    g = matrix.reshape((2,) * (2 * k))
    out = np.tensordot(g, tensor, axes=(list(range(k, 2 * k)), list(qubits)))
    return np.moveaxis(out, list(range(k)), list(qubits))


def _run(ops, tensor, n):
    for op in _flatten(ops):
        g = op.gate
        if hasattr(g, '_unitary_'):
            tensor = _apply(np.asarray(g._unitary_(), dtype=complex), op.qubits, tensor, n)
        else:
            tensor = _run(g._decompose_(list(op.qubits)), tensor, n)
    return tensor


def unitary(gate):
    """The matrix of a gate on its own register (role of cirq.unitary; ground_state.py:154)."""
    if hasattr(gate, '_unitary_'):
        return np.asarray(gate._unitary_(), dtype=complex)
    n = gate.num_qubits()
    full = np.eye(2 ** n, dtype=complex).reshape((2,) * (2 * n))
    out = _run(gate._decompose_(list(range(n))), full, n)
    return out.reshape(2 ** n, 2 ** n)


def final_state(ops, n_qubits):
    """State vector after applying `ops` to |0...0> (role of cirq.Simulator().simulate(C).final_state,
    ground_state.py:165), complex128."""
    psi = np.zeros((2,) * n_qubits, dtype=complex)
    psi[(0,) * n_qubits] = 1.0
    return _run(ops, psi, n_qubits).reshape(-1)


# ---------------------------------------------------------------------------------------------
# Tensor / State wiring
# ---------------------------------------------------------------------------------------------
class Tensor(Gate):
    def __init__(self, unitary, symbol):
        self.U = np.asarray(unitary)
        self.n_qubits = int(np.log2(self.U.shape[0]))
        self.symbol = symbol

    def _unitary_(self):
        return self.U

    def num_qubits(self):
        return self.n_qubits

    def __pow__(self, power, modulo=None):
        if power == -1:
            return self.__class__(self.U.conj().T, symbol=self.symbol + '†')
        return self.__class__(np.linalg.matrix_power(self.U, power), symbol=self.symbol)


class StateTensor(Tensor):
    pass


class Environment(Tensor):
    pass


class FullStateTensor(StateTensor):
    """State tensor as a 2D x 2D unitary."""

    def __init__(self, unitary, symbol='U'):
        super().__init__(unitary, symbol)

    def raise_power(self, power):
        return PowerCircuit(state=self, power=power)


class FullEnvironment(Environment):
    """Environment as a D^2 x D^2 unitary."""

    def __init__(self, unitary, symbol='V'):
        super().__init__(unitary, symbol)


class PowerCircuit(Gate):
    """K staggered copies of U: the quantum statement of the power method (represent.py:235-248)."""

    def __init__(self, state, power):
        self.power = power
        self.state = state

    def _decompose_(self, qubits):
        n = self.state.num_qubits()
        return [FullStateTensor(self.state.U)(*qubits[i:n + i]) for i in reversed(range(self.power))]

    def num_qubits(self):
        return self.state.num_qubits() + (self.power - 1)

    def _set_power(self, power):
        self.power = power


class State(Gate):
    """V on qubits[n:], then U on qubits[i:i+nu] for i = n-1 ... 0 (represent.py:258-262); final
    register layout [a | sigma_1 ... sigma_n | b]."""

    def __init__(self, u, v, n=1):
        self.u = u
        self.v = v
        self.n_phys_qubits = n
        self.bond_dim = int(2 ** (u.num_qubits() - 1))

    def _decompose_(self, qubits):
        nv, nu, n = self.v.num_qubits(), self.u.num_qubits(), self.n_phys_qubits
        return [self.v(*qubits[n:n + nv])] + [self.u(*qubits[i:i + nu]) for i in reversed(range(n))]

    def num_qubits(self):
        return self.n_phys_qubits + self.v.num_qubits()


# ---------------------------------------------------------------------------------------------
# parameterised ansatz gates
# ---------------------------------------------------------------------------------------------
class _Ansatz(Gate):
    symbol = 'U'

    def __init__(self, bond_dim, βγs):
        self.βγs = βγs
        self.p = len(βγs)
        self.n_qubits = int(np.log2(bond_dim)) + 1

    def num_qubits(self):
        return self.n_qubits

    def _cnot_ladder(self, qubits):
        return [CNOT(qubits[i], qubits[i + 1]) for i in reversed(range(self.n_qubits - 1))]


class ShallowQAOAStateTensor(_Ansatz):
    """Per (beta, gamma): X**beta on every qubit, ZZ**gamma on neighbours (represent.py:268-285)."""
    device_kind = 1      # QMPS_ANSATZ_SHALLOW_QAOA: parameters -> tensor runs on the GPU for batches

    def _decompose_(self, qubits):
        return [[x_pow(b)(q) for q in qubits] + [zz_pow(g)(qubits[i], qubits[i + 1]) for i in range(self.n_qubits - 1)]
                for b, g in split_2s(self.βγs)]


class ShallowCNOTStateTensor(_Ansatz):
    """Per (beta, gamma): rz(beta) on all, rx(gamma) on all, H(q0), CNOT ladder from the bottom
    (represent.py:288-310).  The default state tensor of SparseFullEnergyOptimizer."""
    device_kind = 0      # QMPS_ANSATZ_SHALLOW_CNOT

    @staticmethod
    def params_per_iter():
        return 2

    def _decompose_(self, qubits):
        return [[rz(b)(q) for q in qubits] + [rx(g)(q) for q in qubits] + [H(qubits[0])] + self._cnot_ladder(qubits)
                for b, g in split_2s(self.βγs)]


class ShallowCNOTStateTensor_nonuniform(_Ansatz):
    """Per layer 2(n) angles: rz(p[i]) and rx(p[i+n]) on qubit i, CNOT ladder (represent.py:312-332)."""
    device_kind = 4      # QMPS_ANSATZ_SHALLOW_CNOT_NONUNIFORM

    def __init__(self, bond_dim, βγs):
        super().__init__(bond_dim, βγs)
        self.D = bond_dim

    @staticmethod
    def params_per_iter(D):
        return int((np.log2(D) + 1) * 2)

    def _decompose_(self, qubits):
        n = self.n_qubits
        return [[rz(p[i])(q) for i, q in enumerate(qubits)] + [rx(p[i + n])(q) for i, q in enumerate(qubits)] +
                self._cnot_ladder(qubits) for p in split_ns(self.βγs, 2 * n)]


class ShallowCNOTStateTensor3(_Ansatz):
    """rz, rx, rz on all qubits, H(q0), CNOT ladder (represent.py:334-354)."""
    device_kind = 3      # QMPS_ANSATZ_SHALLOW_CNOT3

    def _decompose_(self, qubits):
        return [[rz(b)(q) for q in qubits] + [rx(g)(q) for q in qubits] + [rz(w)(q) for q in qubits] + [H(qubits[0])] +
                self._cnot_ladder(qubits) for b, g, w in split_3s(self.βγs)]


class ExactAfter4(_Ansatz):
    """Six angles per layer on qubits 0,1, CNOT ladder, cyclic SWAPs (represent.py:356-380)."""
    device_kind = 5      # QMPS_ANSATZ_EXACT_AFTER4

    def __init__(self, bond_dim, βγs):
        super().__init__(bond_dim, βγs)
        self.params_per_iter = 6

    def _decompose_(self, qubits):
        n = self.n_qubits
        return [[rz(a)(qubits[0]), rz(d)(qubits[1]), rx(b)(qubits[0]), rx(e)(qubits[1]), rz(c)(qubits[0]),
                 rz(f)(qubits[1])] + self._cnot_ladder(qubits) +
                [SWAP(qubits[i], qubits[i + 1 if i != n - 1 else 0]) for i in range(n)]
                for a, b, c, d, e, f in split_ns(self.βγs, 6)]


class ShallowFullStateTensor(_Ansatz):
    """Universal two-qubit gate, 15 angles (represent.py:382-404)."""
    device_kind = 2      # QMPS_ANSATZ_SHALLOW_FULL (D = 2)

    def __init__(self, bond_dim, βγs, symbol='U'):
        super().__init__(bond_dim, βγs)
        self.symbol = symbol

    def _decompose_(self, qubits):
        v, (a, b) = self.βγs, qubits[:2]
        return [rz(v[0])(a), rx(v[1])(a), rz(v[2])(a), rz(v[3])(b), rx(v[4])(b), rz(v[5])(b),
                CNOT(a, b), ry(v[6])(a), CNOT(b, a), ry(v[7])(a), rz(v[8])(b), CNOT(a, b),
                rz(v[9])(a), rx(v[10])(a), rz(v[11])(a), rz(v[12])(b), rx(v[13])(b), rz(v[14])(b)]


class StateGate(Gate):
    """rx, rx, rz, rz, XX**e, YY**f on two qubits (represent.py:406-423)."""
    device_kind = 6      # QMPS_ANSATZ_STATE_GATE (D = 2; built as StateGate(params), no bond-dimension argument)

    def __init__(self, βγs, symbol='U'):
        self.βγs = βγs
        self.p = len(βγs)
        self.n_qubits = 2
        self.symbol = symbol

    def num_qubits(self):
        return 2

    def _decompose_(self, qubits):
        a, b, c, d, e, f = self.βγs[:6]
        q0, q1 = qubits
        return [rx(a)(q0), rx(b)(q1), rz(c)(q0), rz(d)(q1), xx_pow(e)(q0, q1), yy_pow(f)(q0, q1)]


def build_gate(cls, D, params):
    """Gate object of class `cls` at bond dimension D: every state-tensor class of this module is `cls(D, params)` except
    StateGate, whose constructor takes the parameters alone (represent.py:406-410 of the reference; D = 2 only)."""
    if cls is StateGate:
        if D != 2:
            raise ValueError('StateGate exists at D = 2 only')
        return cls(params)
    return cls(D, params)


class ShallowEnvironment(Gate):
    """QAOA-style environment ansatz on 2 log2(D) qubits (represent.py:425-442)."""

    def __init__(self, bond_dim, βγs):
        self.βγs = βγs
        self.p = len(βγs)
        self.n_qubits = 2 * int(np.log2(bond_dim))

    def num_qubits(self):
        return self.n_qubits

    def _decompose_(self, qubits):
        n = self.n_qubits
        return [[x_pow(b)(q) for q in qubits] + [zz_pow(g)(qubits[i], qubits[i + 1]) for i in range(n - 1)]
                for b, g in split_2s(self.βγs)]


# ---------------------------------------------------------------------------------------------
# environment consistency objective (represent.py:18-56, 88-114), D = 2, exact (state-vector) form
# ---------------------------------------------------------------------------------------------
def bloch_vector_of(state, qubit, n_qubits=None):
    """(<X>, <Y>, <Z>) of one qubit of a state vector (role of cirq's `bloch_vector_of`; big-endian qubit order)."""
    state = np.asarray(state, dtype=complex)
    n = int(np.log2(state.size)) if n_qubits is None else n_qubits
    psi = np.moveaxis(state.reshape((2,) * n), qubit, 0).reshape(2, -1)
    rho = psi @ psi.conj().T
    return np.array([2 * rho[1, 0].real, 2 * rho[1, 0].imag, (rho[0, 0] - rho[1, 1]).real])


def full_tomography_env_objective_function(U, V):
    """Norm of the difference of the Bloch vectors of qubit 0 in State(U, V)|000> and in V|00>
    (represent.py:88-114): zero exactly when V carries the right environment of U.  U: FullStateTensor-like gate on
    two qubits, V: environment gate on two qubits."""
    lhs = final_state(State(U, V, 1)._decompose_(line_qubits(3)), 3)
    rhs = final_state([V(*line_qubits(2))], 2)
    return float(np.linalg.norm(bloch_vector_of(lhs, 0) - bloch_vector_of(rhs, 0)))


def get_env(U, C0=None, sample=False, reps=100000):
    """Variational environment (represent.py:18-56): Nelder-Mead over the 8 real numbers of a 2 x 2 matrix C,
    minimising `full_tomography_env_objective_function(U, environment_to_unitary(C))`.  The sampled (shot-noise)
    objective of the reference is out of scope; `sample=True` raises."""
    from scipy.optimize import minimize
    from .tools import environment_to_unitary, from_real_vector, to_real_vector
    if sample:
        raise NotImplementedError('sampled (shot-noise) objectives are out of scope')
    if C0 is None:
        C0 = np.random.randn(2, 2) + 1j * np.random.randn(2, 2)
    state = FullStateTensor(np.asarray(U))

    def f_obj(v):
        return full_tomography_env_objective_function(state, FullEnvironment(environment_to_unitary(from_real_vector(v))))
    res = minimize(f_obj, to_real_vector(np.asarray(C0).reshape(-1)), method='Nelder-Mead')
    return environment_to_unitary(from_real_vector(res.x))
