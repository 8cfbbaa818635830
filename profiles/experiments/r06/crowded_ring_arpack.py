"""CPU: what the reference's own eigen-solver (scipy.sparse.linalg.eigs = ARPACK, k = 1, 'LM': xmps Map.right_fixed_point) returns on four grid pairs of
D = 8 whose mixed transfer map has a CROWDED RING of eigenvalues on top (offenders of grid_overlaps_probe.py), for eight start vectors - beside the
60-digit truth's moduli (numpy agrees to 1e-9 here) and the device's answer."""
import os, sys
R = os.path.abspath(os.path.join(os.path.dirname(__file__), '../../..')); sys.path.insert(0, R + '/tests'); sys.path.insert(0, R)
import numpy as np, evolve_replay as ER
from oracle import qmps_oracle as O
from scipy.linalg import expm
from scipy.sparse.linalg import eigs
H=O.hamiltonian_matrix({'ZZ':-1.0,'X':1.0})
cases=[(8,[2,-6,-4,-4,0,-6],[-6,4,0,-6,-8,-2],0.05,0.5000024706098224),(8,[-4,0,0,-4,-4,-8],[8,-6,2,-6,8,-8],0.05,0.5005337763354931),(8,[-8,-8,-2,4,6,0],[0,0,-6,6,-4,0],0.3,0.5009865081488195),(8,[6,6,2,0,0,-8],[8,2,-4,-4,-8,-4],0.3,0.5102551297767004)]
for D,a,b,dt,dev in cases:
    WW=expm(-1j*dt*H); a=np.array(a)*np.pi/4; b=np.array(b)*np.pi/4
    A=ER.tensor(0,D,a); B=ER.tensor(0,D,b)
    E=O.transfer_matrix(np.tensordot(WW,O.merge(A,A),[1,0]),O.merge(B,B))
    w=np.linalg.eigvals(E); w=np.sort(abs(w))[::-1]
    res=[]
    for seed in range(8):
        v0=np.random.default_rng(seed).standard_normal(D*D)+1j*np.random.default_rng(seed+100).standard_normal(D*D)
        try:
            val=eigs(E,k=1,which='LM',v0=v0)[0][0]; res.append(round(abs(val),7))
        except Exception as e: res.append(type(e).__name__)
    print('top moduli',w[:6].round(7),'device',round(dev,7),'ARPACK',res)
