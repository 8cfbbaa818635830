"""Process-wide cache of EnergyEngine contexts (one per bond dimension), so that the scalar
objective functions of the reference API do not pay hipMalloc on every call."""
import atexit

_engines = {}


def engine(D, min_batch=1, device=0):
    """Return a cached engine for bond dimension D with capacity >= min_batch (grown geometrically)."""
    from .engine import EnergyEngine
    key = (int(D), int(device))
    eng = _engines.get(key)
    if eng is None or eng.max_batch < min_batch:
        if eng is not None:
            eng.close()
        cap = 1
        while cap < max(min_batch, 1024):
            cap *= 2
        eng = EnergyEngine(D, cap, device=device)
        _engines[key] = eng
    return eng


def shutdown():
    for e in _engines.values():
        e.close()
    _engines.clear()


atexit.register(shutdown)
