import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from qmps_amd import EnergyEngine
g = np.load(os.path.join(ROOT, 'tests/golden/refshim_golden.npz'))
F = g['refshim_roto_fits']
lib = ctypes.CDLL(os.path.join(ROOT, 'tests/csrc/libroto_emu.so'))
dp = ctypes.POINTER(ctypes.c_double)
lib.roto_emu_steps.argtypes = [ctypes.c_long, dp, ctypes.c_int, dp]
ab = np.ascontiguousarray(F[:, :4]); host = np.empty(len(F))
lib.roto_emu_steps(len(F), ab.ctypes.data_as(dp), 0, host.ctypes.data_as(dp))
with EnergyEngine(2, 4096) as eng:
    dev = eng.roto_rule_probe(ab, 0)
    devg = eng.roto_rule_probe(ab, 1)
d = np.abs(dev - F[:, 4])
print('device vs scipy: max', d.max(), 'frac > 1e-9', (d > 1e-9).mean(), 'percentiles', np.percentile(d, [50, 90, 99]))
print('host vs scipy max', np.abs(host - F[:, 4]).max())
k = np.argsort(d)[-5:]
for i in k:
    print(i, F[i, :4], 'scipy', F[i, 4], 'dev', dev[i], 'host', host[i], 'nfev', F[i, 6])
