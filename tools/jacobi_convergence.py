"""Numpy simulation of the Hermitian Jacobi iteration of siegel_math.hpp on the benchmark tables:
sweep-by-sweep off-diagonal ratios for several pivot orders, error of (k sweeps + finishing sweep),
statistics of the finishing-sweep certificate.  Evidence for DESIGN.md section 5; CPU only."""
import numpy as np, torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from sympa_amd import data
def gramH(table, pairs):
    z1=table[pairs[:,0]].numpy(); z2=table[pairs[:,1]].numpy()
    L1=np.linalg.cholesky(z1[:,1]); L2=np.linalg.cholesky(z2[:,1])
    D=(z2[:,0]-z1[:,0])+1j*(z2[:,1]-z1[:,1])
    F=np.linalg.solve(L1.astype(complex),D)
    E=np.swapaxes(np.linalg.solve(L2.astype(complex),np.swapaxes(F,-1,-2)),-1,-2)
    return np.conj(np.swapaxes(E,-1,-2))@E, E
def offratio(H):
    d=np.real(np.einsum('bii->bi',H)); 
    off=H.copy(); idx=np.arange(H.shape[1]); off[:,idx,idx]=0
    return np.sqrt((np.abs(off)**2).sum((1,2))/2/(d**2).sum(1))
def sweep(H, order):
    b,n,_=H.shape
    for (p,q) in order:
        beta=H[:,p,q]; a2=np.abs(beta)**2; delta=(H[:,q,q]-H[:,p,p]).real
        rad=np.sqrt(delta**2+4*a2); den=np.abs(delta)+rad
        u=np.where(den>0, np.where(delta>=0,2.0,-2.0)/np.where(den>0,den,1),0)
        t2=u*u*a2; c=1/np.sqrt(1+t2); w=c*u*beta
        J=np.tile(np.eye(n,dtype=complex),(b,1,1))
        J[:,p,p]=c; J[:,q,q]=c; J[:,p,q]=w; J[:,q,p]=-np.conj(w)
        H=np.conj(np.swapaxes(J,-1,-2))@H@J
        H[:,p,q]=0; H[:,q,p]=0
    return H
n=4
cyc=[(p,q) for p in range(n) for q in range(p+1,n)]
rr=[(0,1),(2,3),(0,2),(1,3),(0,3),(1,2)]
rr2=[(0,3),(1,2),(0,2),(1,3),(0,1),(2,3)]
table=data.trained_like_table(5041,4); pairs=data.sample_pairs(5041,8192)
H0,E=gramH(table,pairs)
def report(name,H,order,presort=False):
    H=H.copy()
    if presort:
        d=np.real(np.einsum('bii->bi',H)); perm=np.argsort(-d,axis=1)
        H=np.take_along_axis(np.take_along_axis(H,perm[:,:,None],1),perm[:,None,:],2)
    out=[]
    for k in range(5):
        H=sweep(H,order); r=offratio(H)
        out.append('k=%d: med %.1e p98 %.1e max %.1e'%(k+1,np.median(r),np.percentile(r,98),r.max()))
    print(name); [print('   ',o) for o in out]
report('cyclic rows',H0,cyc); report('round robin',H0,rr); report('rr2',H0,rr2)
report('presort desc + cyclic',H0,cyc,True); report('presort desc + rr',H0,rr,True)
print('==== more data sets, presort+rr vs rr')
for name,tab in (('init',data.init_table(5041,4)),('trained1.0',data.trained_like_table(5041,4,scale=1.0)),('trained0.1',data.trained_like_table(5041,4,scale=0.1))):
    H1,_=gramH(tab,pairs)
    print(name); report(' rr',H1,rr); report(' presort desc+rr',H1,rr,True)
# ascending
def report_asc(name,H,order):
    d=np.real(np.einsum('bii->bi',H)); perm=np.argsort(d,axis=1)
    H=np.take_along_axis(np.take_along_axis(H,perm[:,:,None],1),perm[:,None,:],2)
    report(name,H,order)
report_asc('presort ASC + rr',H0,rr)
print('==== error of (k sweeps + final diag-only sweep) vs eps before final')
def final_diag(H, order):
    H=H.copy(); b,n,_=H.shape
    d=np.real(np.einsum('bii->bi',H)).copy()
    for (p,q) in order:
        a2=np.abs(H[:,p,q])**2; delta=d[:,q]-d[:,p]
        r=np.sqrt(delta**2+4*a2); den=np.abs(delta)+r
        ua2=np.where(den>0,np.where(delta>=0,2.0,-2.0)*a2/np.where(den>0,den,1),0)
        d[:,p]-=ua2; d[:,q]+=ua2
    return d
def riem(lam): 
    lam=np.maximum(lam,0)/4; return np.sqrt((4*np.arcsinh(np.sqrt(lam))**2).sum(1))
for name,tab in (('trained0.3',table),('init',data.init_table(5041,4)),('trained1.0',data.trained_like_table(5041,4,scale=1.0))):
    H1,_=gramH(tab,pairs)
    exact=riem(np.linalg.eigvalsh(H1))
    for presort in (False,True):
        H=H1.copy()
        if presort:
            d=np.real(np.einsum('bii->bi',H)); perm=np.argsort(-d,axis=1)
            H=np.take_along_axis(np.take_along_axis(H,perm[:,:,None],1),perm[:,None,:],2)
        for k in range(1,5):
            H=sweep(H,rr)
            eps=offratio(H)
            err_nofinal=np.abs(riem(np.real(np.einsum('bii->bi',H)))-exact)/exact
            err=np.abs(riem(final_diag(H,rr))-exact)/exact
            # bucket by eps
            msg=[]
            for lo,hi in ((1e-2,1),(1e-3,1e-2),(1e-4,1e-3),(1e-5,1e-4),(0,1e-5)):
                m=(eps>=lo)&(eps<hi)
                if m.any(): msg.append('eps[%g,%g) n=%d maxerr %.1e (nofinal %.1e)'%(lo,hi,m.sum(),err[m].max(),err_nofinal[m].max()))
            print(name,'presort' if presort else 'plain','k=%d'%k,' | '.join(msg))
print('==== angle criterion after k sweeps, 65536 pairs')
pairs=data.sample_pairs(5041,65536)
for name,tab in (('trained0.3',data.trained_like_table(5041,4)),('init',data.init_table(5041,4))):
    H1,_=gramH(tab,pairs)
    exact=riem(np.linalg.eigvalsh(H1))
    H=H1.copy()
    for k in range(1,5):
        H=sweep(H,rr)
        d=np.real(np.einsum('bii->bi',H)); diag2=(d**2).sum(1)
        tmax=np.zeros(len(H))
        for (p,q) in rr:
            a2=np.abs(H[:,p,q])**2; dl=(d[:,q]-d[:,p])**2
            t2=np.where(a2<=1e-24*diag2,0,a2/np.maximum(dl,1e-300))
            tmax=np.maximum(tmax,np.sqrt(t2))
        err=np.abs(riem(final_diag(H,rr))-exact)/exact
        print(name,'k=%d'%k,'frac t>3e-4: %.4f  t>1e-2: %.5f  t>0.1: %.6f'%((tmax>3e-4).mean(),(tmax>1e-2).mean(),(tmax>0.1).mean()),' max err after final: %.2e'%err.max(), ' worst err among t>1e-2: %.2e'%(err[tmax>1e-2].max() if (tmax>1e-2).any() else 0))
print('==== certificate failure stats after 3 sweeps')
tab=data.trained_like_table(5041,4)
worst=0
for bid in range(4):
    pairs=data.sample_pairs(5041,65536,bid)
    H1,_=gramH(tab,pairs); exact=riem(np.linalg.eigvalsh(H1))
    H=H1.copy()
    for k in range(3): H=sweep(H,rr)
    d=np.real(np.einsum('bii->bi',H)); diag2=(d**2).sum(1)
    tmax=np.zeros(len(H)); off2=np.zeros(len(H))
    for (p,q) in rr:
        a2=np.abs(H[:,p,q])**2; dl=(d[:,q]-d[:,p])**2
        t2=np.where(a2<=1e-24*diag2,0,a2/np.maximum(dl,1e-300)); tmax=np.maximum(tmax,np.sqrt(t2)); off2+=a2
    eps=np.sqrt(off2/diag2)
    err=np.abs(riem(final_diag(H,rr))-exact)/exact
    print('batch',bid,'n(t>1e-2)=',(tmax>1e-2).sum(),'n(t>3e-3)=',(tmax>3e-3).sum(),'n(eps>2e-3)=',(eps>2e-3).sum(),'n(eps>1e-2)=',(eps>1e-2).sum(),'max eps %.1e max t %.1e maxerr %.1e'%(eps.max(),tmax.max(),err.max()))
print('==== third-order bound stats after 3 sweeps')
import itertools
for name,tab in (('trained0.3',data.trained_like_table(5041,4)),('init',data.init_table(5041,4)),('trained1.0',data.trained_like_table(5041,4,scale=1.0))):
  for bid in range(2):
    pairs=data.sample_pairs(5041,65536,bid)
    H1,_=gramH(tab,pairs); exact=riem(np.linalg.eigvalsh(H1))
    H=H1.copy()
    for k in range(3): H=sweep(H,rr)
    d=np.real(np.einsum('bii->bi',H)); diag2=(d**2).sum(1)
    a2={}; t2={}
    for p in range(4):
        for q in range(p+1,4):
            a=np.abs(H[:,p,q])**2; dl=(d[:,q]-d[:,p])**2
            a2[(p,q)]=a2[(q,p)]=a
            t2[(p,q)]=t2[(q,p)]=np.where(a<=1e-24*diag2,0,a/np.maximum(dl,1e-300))
    S=np.zeros(len(H))   # sum of squares of the 12 terms
    B=np.zeros(len(H))   # exact bound max_i sum
    for i in range(4):
        others=[j for j in range(4) if j!=i]
        Bi=np.zeros(len(H))
        for j,k in itertools.combinations(others,2):
            term2=4*a2[(j,k)]*t2[(i,j)]*t2[(i,k)]
            S+=term2; Bi+=np.sqrt(term2)
        B=np.maximum(B,Bi)
    rel=B/np.sqrt(diag2); relS=np.sqrt(3*S)/np.sqrt(diag2)
    err=np.abs(riem(final_diag(H,rr))-exact)/exact
    print(name,bid,'B/|diag|: med %.1e p99.9 %.1e max %.1e | cauchy version max %.1e | n(B>1e-9)=%d n(B>1e-8)=%d | true err max %.1e'%(np.median(rel),np.percentile(rel,99.9),rel.max(),relS.max(),(relS>1e-9).sum(),(relS>1e-8).sum(),err.max()))
