import sys, numpy as np
sys.path.insert(0,'/root/repo')
from qmps_amd import EnergyEngine, _lib as L
import bench
eng=EnergyEngine(8,4096)
h=bench.xxz_h(0.5); eng.set_hamiltonian(h)
rng=np.random.default_rng(20241022)
p0=rng.standard_normal((256,6))
for sweeps in (0,3,20):
    p=p0.copy()
    if sweeps: hist,p=eng.rotosolve(0,p0,sweeps)
    sh=np.repeat(p,3,axis=0); sh[:,2]+=np.tile([0,np.pi/2,-np.pi/2],256)
    eng.set_ansatz_params(0,sh); eng.launch(768,solver='direct')
    E,it,st=eng.results(768)
    print('after',sweeps,'sweeps: iters hist',np.bincount(np.minimum(it,12)),'max',it.max(),'status',np.bincount(st))
