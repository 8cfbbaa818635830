import sys, itertools
R = __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '../../../..')); sys.path.insert(0, R); sys.path.insert(0, R + '/tests'); sys.path.insert(0, __import__('os').path.dirname(__file__))
import numpy as np
import evolve_replay as ER
from oracle import qmps_oracle as O
from scipy.linalg import expm
from aberth import solve
H=O.hamiltonian_matrix({'ZZ':-1.0,'X':1.0})
rng=np.random.default_rng(5)
bad=0; tot=0; its=[]; fb=0
def chk(E,tag):
    global bad,tot,fb
    tr=max(abs(np.linalg.eigvals(E))); tot+=1
    e,it,f=solve(E); its.append(it); fb+=f
    if abs(abs(e)-tr)>1e-9*max(tr,1e-3):
        bad+=1
        if bad<10: print(tag,'true',tr,'got',abs(e),it,f,np.round(np.linalg.eigvals(E),5))
for dt in (0.0,0.05):
  WW=expm(-1j*dt*H)
  for kind,P in ((0,2),(0,8),(2,15)):
    for trial in range(3000 if P>2 else 0):
        a=rng.integers(-4,5,P)*np.pi/4 if trial%2 else rng.integers(-2,3,P)*np.pi/2
        b=rng.integers(-4,5,P)*np.pi/4 if trial%2 else rng.integers(-2,3,P)*np.pi/2
        A=ER.tensor(kind,2,a); B=ER.tensor(kind,2,b)
        C=np.tensordot(WW,O.merge(A,A),[1,0]); chk(O.transfer_matrix(C,O.merge(B,B)),(dt,kind,P))
    if P==2:
        g=np.arange(-4,5)*np.pi/4
        for a in itertools.product(g,g):
          for b in itertools.product(g,g):
            A=ER.tensor(kind,2,np.array(a)); B=ER.tensor(kind,2,np.array(b))
            C=np.tensordot(WW,O.merge(A,A),[1,0]); chk(O.transfer_matrix(C,O.merge(B,B)),(dt,kind,P))
print('grid: bad',bad,'of',tot,'fallback frac',fb/tot,'its mean',np.mean(its),'max',max(its))
# typical
bad=tot=fb=0; its=[]
WW=expm(-0.05j*H); rng=np.random.default_rng(7)
for kind,P in ((2,15),(0,8),(0,2)):
    for t in range(1500):
        a=rng.standard_normal(P); b=a+ (0.05 if t%2 else 1.0)*rng.standard_normal(P)
        A=ER.tensor(kind,2,a); B=ER.tensor(kind,2,b)
        C=np.tensordot(WW,O.merge(A,A),[1,0]); chk(O.transfer_matrix(C,O.merge(B,B)),('typ',kind,P))
print('typical: bad',bad,'of',tot,'fallback frac',fb/tot,'its mean',np.mean(its),'max',max(its),np.bincount(its))
