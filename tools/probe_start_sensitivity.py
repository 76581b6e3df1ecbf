"""CPU only: the oracle's own sensitivity of one multi-start end value (compaction case N = 300, D = 6, Matern, start 50) to last-place
changes of the start and of the model, and its trajectory round by round -> profiles/r05_start50_probe.log."""
import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from oracle import oracle_py as oracle
oracle.build()
from util import synth_problem, synth_candidates
D,N,S,n_local,kernel=6,300,700,30,1
X,y,theta,b=synth_problem(oracle,D,N)
starts=synth_candidates(oracle,D,S); starts[:, ::7]=np.round(starts[:, ::7])
ref=oracle.Regressor(X,y,theta,b,kernel=kernel)
i=50
ro=ref.acq_maximize(starts[:, i:i+1], n_local, diag=True)
print("start", starts[:,i]); print("end value", ro["y_stars"][0], "x", ro["x_stars"][:,0], "margin", ro["armijo_margin"][0])
print({k: (v if np.ndim(v)==0 else np.asarray(v).shape) for k,v in ro.items()})
base=ro["y_stars"][0]
full=ref.acq_maximize(starts, n_local, diag=True); scale=np.abs(full["y_stars"]).max(); print("scale", scale, "value of start 50 in the full run", full["y_stars"][50])
worst=0
for toward in (0.5,2.0,-1.0):
    s1=starts[:, i:i+1].copy(); s1[:,0]=np.nextafter(s1[:,0],toward)
    r1=ref.acq_maximize(s1,n_local,diag=True); d=abs(r1["y_stars"][0]-base)/scale; worst=max(worst,d); print("start ulp toward",toward,d)
for which,toward,n_ulp in ((0,2.0,1),(0,0.0,1),(1,1.0,1),(1,0.0,1),(0,2.0,8),(0,0.0,8),(1,1.0,8),(1,0.0,8)):
    th1,b1=theta.copy(),b
    for _ in range(n_ulp):
        if which==0: th1[0]=np.nextafter(th1[0],toward)
        else: b1=float(np.nextafter(b1,toward))
    r1=oracle.Regressor(X,y,th1,b1,kernel=kernel).acq_maximize(starts[:, i:i+1],n_local,diag=True); d=abs(r1["y_stars"][0]-base)/scale; worst=max(worst,d); print("model",which,toward,n_ulp,d)
print("worst",worst)
# trajectory: run with n_local = 1..30 and print the value / x per budget
prev=None
for n in range(1,31):
    r=ref.acq_maximize(starts[:, i:i+1], n, diag=True)
    x=r["x_stars"][:,0]
    print(n, repr(r["y_stars"][0]), np.array2string(x,precision=6), "at bounds:", [int(k) for k in np.nonzero((x<=0)|(x>=1))[0]])
print("---- larger perturbations")
for n_ulp in (16, 64, 256):
    for which,toward in ((0,2.0),(0,0.0),(1,1.0),(1,0.0)):
        th1,b1=theta.copy(),b
        for _ in range(n_ulp):
            if which==0: th1[0]=np.nextafter(th1[0],toward)
            else: b1=float(np.nextafter(b1,toward))
        r1=oracle.Regressor(X,y,th1,b1,kernel=kernel).acq_maximize(starts[:, i:i+1],n_local,diag=True); print("model",which,toward,n_ulp,abs(r1["y_stars"][0]-base)/scale, abs(r1["y_stars"][0]-base)/abs(base))
    for toward in (0.5,2.0,-1.0):
        s1=starts[:, i:i+1].copy()
        for _ in range(n_ulp): s1[:,0]=np.nextafter(s1[:,0],toward)
        r1=ref.acq_maximize(s1,n_local,diag=True); print("start",toward,n_ulp,abs(r1["y_stars"][0]-base)/scale)
