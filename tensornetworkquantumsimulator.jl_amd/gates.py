"""Gate registry: circuit tuples ("Rzz", [v1, v2], theta) -> d^k x d^k matrices, qiskit convention.
Mirrors src/Apply/gate_definitions.jl (GateSpec :12-17, GATES :21-64, ALIASES :74-83, toitensor :110-153,
register_gate!/register_alias!/unregister_gate! :189-239, in-house gates :248-281, Levenshtein suggestions
:97-105 and src/utils.jl:115-135).  Matrices are built in complex128 and cast to the state's precision by the
library (adapt_gate, src/Apply/apply_gates.jl:41-44).  First listed vertex = most significant index."""
from __future__ import annotations

import cmath
import math
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np


class GateSpec:
    def __init__(self, fn: Callable[..., np.ndarray], nparams: int = 0, rescale: Callable = lambda *p: p):
        self.fn, self.nparams, self.rescale = fn, nparams, rescale


_PX = np.array([[0, 1], [1, 0]], dtype=complex)
_PY = np.array([[0, -1j], [1j, 0]], dtype=complex)
_PZ = np.array([[1, 0], [0, -1]], dtype=complex)
_PAULI = {"X": _PX, "Y": _PY, "Z": _PZ}


def _rot(p: np.ndarray, phi: float) -> np.ndarray:
    """exp(-i phi P) for an involutory P (P^2 = 1)"""
    return math.cos(phi) * np.eye(p.shape[0], dtype=complex) - 1j * math.sin(phi) * p


def _exp_herm(h: np.ndarray, t: float) -> np.ndarray:
    w, v = np.linalg.eigh(h)
    return (v * np.exp(-1j * t * w)) @ v.conj().T


def _controlled(u: np.ndarray) -> np.ndarray:
    m = np.eye(4, dtype=complex)
    m[2:, 2:] = u
    return m


def _half(t):
    return (t / 2,)


def _upstream_only(name: str):
    """"Rz+" and "Rz+z+" are entries of the reference's registry (gate_definitions.jl:33,58) whose matrices the reference does not define:
    the name is forwarded to `ITensors.op(name, ...)`, and no such op exists in the ITensors sources the reference pins (0.9) nor anywhere
    under the reference tree -- building the gate fails there as well.  The names are registered here (locked built-ins, lowercase aliases,
    one parameter) so that name resolution, aliasing and `register_gate` conflicts behave like the reference; asking for the matrix raises
    with this explanation instead of guessing a convention.  A user who has the definition registers it under another name."""
    def fn(*_params):
        raise ValueError(f'gate "{name}" is listed in the reference registry but its matrix is only defined by an upstream ITensors.op '
                         f'method that is not part of the reference sources; register the matrix under a custom name with register_gate')
    return fn


GATES: Dict[str, GateSpec] = {
    "Rz+": GateSpec(_upstream_only("Rz+"), 1), "Rz+z+": GateSpec(_upstream_only("Rz+z+"), 1),
    "X": GateSpec(lambda: _PX.copy()), "Y": GateSpec(lambda: _PY.copy()), "Z": GateSpec(lambda: _PZ.copy()),
    "H": GateSpec(lambda: np.array([[1, 1], [1, -1]], dtype=complex) / math.sqrt(2)),
    "Rx": GateSpec(lambda t: _rot(_PX, t / 2), 1), "Ry": GateSpec(lambda t: _rot(_PY, t / 2), 1),
    "Rz": GateSpec(lambda t: _rot(_PZ, t / 2), 1),
    "P": GateSpec(lambda p: np.diag([1.0, cmath.exp(1j * p)]), 1),
    "CNOT": GateSpec(lambda: _controlled(_PX)), "CX": GateSpec(lambda: _controlled(_PX)),
    "CY": GateSpec(lambda: _controlled(_PY)), "CZ": GateSpec(lambda: _controlled(_PZ)),
    "SWAP": GateSpec(lambda: np.eye(4, dtype=complex)[[0, 2, 1, 3]]),
    "iSWAP": GateSpec(lambda: np.array([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], dtype=complex)),
    "√SWAP": GateSpec(lambda: np.array([[1, 0, 0, 0], [0, (1 + 1j) / 2, (1 - 1j) / 2, 0],
                                         [0, (1 - 1j) / 2, (1 + 1j) / 2, 0], [0, 0, 0, 1]], dtype=complex)),
    "√iSWAP": GateSpec(lambda: np.array([[1, 0, 0, 0], [0, 1 / math.sqrt(2), 1j / math.sqrt(2), 0],
                                          [0, 1j / math.sqrt(2), 1 / math.sqrt(2), 0], [0, 0, 0, 1]], dtype=complex)),
    # qiskit Rxx(theta) = exp(-i theta XX / 2); the reference forwards phi = theta/2 to ITensors (:46-51)
    "Rxx": GateSpec(lambda p: _rot(np.kron(_PX, _PX), p), 1, _half),
    "Ryy": GateSpec(lambda p: _rot(np.kron(_PY, _PY), p), 1, _half),
    "Rzz": GateSpec(lambda p: _rot(np.kron(_PZ, _PZ), p), 1, _half),
    "CRx": GateSpec(lambda t: _controlled(_rot(_PX, t / 2)), 1), "CRy": GateSpec(lambda t: _controlled(_rot(_PY, t / 2)), 1),
    "CRz": GateSpec(lambda t: _controlled(_rot(_PZ, t / 2)), 1),
    "CPHASE": GateSpec(lambda p: np.diag([1, 1, 1, cmath.exp(1j * p)]), 1),
    "Rxxyy": GateSpec(lambda t: _exp_herm(0.5 * (np.kron(_PX, _PX) + np.kron(_PY, _PY)), t), 1),
    "Rxxyyzz": GateSpec(lambda t: _exp_herm(0.5 * (np.kron(_PX, _PX) + np.kron(_PY, _PY) + np.kron(_PZ, _PZ)), t), 1),
    "xx_plus_yy": GateSpec(lambda t, b: np.array(
        [[1, 0, 0, 0], [0, math.cos(t / 2), -1j * math.sin(t / 2) * cmath.exp(-1j * b), 0],
         [0, -1j * math.sin(t / 2) * cmath.exp(1j * b), math.cos(t / 2), 0], [0, 0, 0, 1]], dtype=complex), 2),
}
BUILTIN_GATES = frozenset(GATES)


def _default_aliases() -> Dict[str, str]:
    m = {}
    for canon in GATES:
        low = canon.lower()
        if low != canon:
            m[low] = canon
    m["cp"] = "CPHASE"
    return m


ALIASES: Dict[str, str] = _default_aliases()


def levenshtein(a: str, b: str) -> int:
    if not a:
        return len(b)
    if not b:
        return len(a)
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(cur[j - 1] + 1, prev[j] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


def _suggestions(name: str, topk: int = 3, maxdist: int = 2):
    scored = sorted(((levenshtein(name.lower(), k.lower()), k) for k in GATES))
    return [k for (dist, k) in scored if dist <= maxdist][:topk]


def _resolve(name: str) -> Optional[GateSpec]:
    if name in GATES:
        return GATES[name]
    canon = ALIASES.get(name)
    return GATES[canon] if canon is not None else None


def register_gate(name: str, fn: Callable[..., np.ndarray], nparams: int = 0, rescale: Optional[Callable] = None):
    """register_gate! (:189-203).  `fn(*params)` returns the matrix; built-in names are locked."""
    if name in BUILTIN_GATES:
        raise ValueError(f'"{name}" is a built-in gate and cannot be overwritten. Choose a different name for your custom gate.')
    GATES[name] = GateSpec(fn, nparams, rescale if rescale is not None else (lambda *p: p))
    return name


def register_alias(alias: str, canonical: str):
    if canonical not in GATES:
        raise ValueError(f'Cannot register alias "{alias}" → "{canonical}": canonical gate is not registered.')
    ALIASES[alias] = canonical
    return alias


def unregister_gate(name: str):
    if name in BUILTIN_GATES:
        raise ValueError(f'"{name}" is a built-in gate and cannot be unregistered.')
    GATES.pop(name, None)
    for a, c in list(ALIASES.items()):
        if c == name:
            del ALIASES[a]
    return name


# (name, params) -> (GateSpec it was built from, matrix, its column-major flattening); a layer of a circuit repeats a handful of
# distinct gates a thousand times, and building each matrix costs an eigendecomposition.  Entries are read-only arrays and are
# only reused while the registry still maps the name to the same GateSpec object (GATES / ALIASES are public dicts).
_MATRIX_CACHE: Dict[tuple, tuple] = {}
_PAULI_STRING = object()


def _cached_gate(name: str, params: tuple):
    if len(params) == 1 and isinstance(params[0], (tuple, list)):
        params = tuple(params[0])
    key = (name, params)
    try:
        hit = _MATRIX_CACHE.get(key)
    except TypeError:                 # unhashable parameter (e.g. an array): not cached
        key, hit = None, None
    pauli = len(name) > 1 and all(c in "XYZxyz" for c in name)          # Pauli-string sugar (:123-128)
    spec = _PAULI_STRING if pauli else _resolve(name)
    if hit is not None and hit[0] is spec:
        return hit
    if pauli:
        m = np.ones((1, 1), dtype=complex)
        for c in name:
            m = np.kron(m, _PAULI[c.upper()])
    else:
        if spec is None:
            sug = _suggestions(name)
            msg = f'Unknown gate "{name}".'
            msg += (" Did you mean: " + ", ".join(f'"{s}"' for s in sug) + "?") if sug else f" Registered gates: {sorted(GATES)}."
            raise ValueError(msg)
        if spec.nparams == 0:
            m = np.array(spec.fn(), dtype=complex)
        else:
            if len(params) != spec.nparams:
                raise ValueError(f'Gate "{name}" expects {spec.nparams} parameter(s), got {len(params)}.')
            m = np.array(spec.fn(*spec.rescale(*params)), dtype=complex)
    flat = np.ascontiguousarray(m.ravel(order="F"))
    m.flags.writeable = False; flat.flags.writeable = False
    entry = (spec, m, flat)
    if key is not None:
        if len(_MATRIX_CACHE) >= 4096:
            _MATRIX_CACHE.clear()
        _MATRIX_CACHE[key] = entry
    return entry


def gate_matrix(name, *params) -> np.ndarray:
    """name -> complex128 matrix (toitensor :118-153).  The returned array is shared and read-only; copy it before modifying."""
    if isinstance(name, np.ndarray):
        return np.asarray(name, dtype=complex)
    return _cached_gate(name, params)[1]


def _verts_of(gate, graph) -> list:
    verts = gate[1]
    if type(verts) is list:                      # fast path: the usual ("Rzz", [v1, v2], theta) / ("Rx", [v], theta) forms
        index = graph.index
        n = len(verts)
        if n == 2:
            a, b = verts
            if a in index and b in index and a != b:
                return verts
        elif n == 1 and verts[0] in index:
            return verts
    if not isinstance(verts, (list,)):
        verts = [verts] if verts in graph.index else list(verts)
    verts = list(verts)
    for v in verts:
        if v not in graph.index:
            raise RuntimeError("Vertex does not match the vertex type of the tensor network")
    if len(set(verts)) != len(verts):
        raise RuntimeError("Repeated vertex in collection")
    return verts


def resolve_gate_flat(gate, graph) -> Tuple[np.ndarray, list]:
    """circuit tuple -> (column-major flattened complex128 matrix, [vertices]): what tnqs_apply_gates takes"""
    name = gate[0]
    verts = _verts_of(gate, graph)
    if isinstance(name, np.ndarray):
        return np.asarray(name, dtype=np.complex128).ravel(order="F"), verts
    params = gate[2:] if type(gate) is tuple else tuple(gate[2:])
    try:                                         # fast path: an exact cache hit whose GateSpec is still the registered one
        hit = _MATRIX_CACHE[(name, params)]
        if hit[0] is GATES.get(name):
            return hit[2], verts
    except (KeyError, TypeError):
        pass
    return _cached_gate(name, params)[2], verts


def resolve_gate(gate, graph) -> Tuple[np.ndarray, list]:
    """circuit tuple -> (matrix, [vertices]) with collect_vertices semantics (src/utils.jl:137-160)"""
    return gate_matrix(gate[0], *gate[2:]), _verts_of(gate, graph)
