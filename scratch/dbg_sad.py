import sys; sys.path.insert(0,'.')
import numpy as np
from ma_amd import engine as E, synth, drivers as D
from oracle import prox as px, operators as oo
from oracle.chambolle_pock import chambolle_pock as cp_ref
m,n=40,50
R,T=synth.shifted_rect_pair(m,n,(3,-2))
R=D.normalize(R); T=D.normalize(T)
lam,tau=4.0,1.5
sigma=(1-1e-4)/(8*tau)
u0=np.zeros(2*m*n); v0=np.zeros(4*m*n)
K=oo.pairwise_gradient(m,n,(1.0,1.0),2)
F=lambda v,c,nargout: px.TV_registration(v,sigma,c,nargout)
G=lambda u,c,nargout: px.SAD_registration(u,u0,T,R,(1.0,1.0),lam,tau,c,nargout)
for it in (1,2):
    xr,yr,phr,dhr=cp_ref(F,G,K,u0,v0,1.0,tau,sigma,it)
    eng=E.Engine(E.MODEL_TV_L1_REG,m,n,lam=lam)
    eng.set_images(T.reshape(-1,order='F'),R.reshape(-1,order='F'))
    eng.linearize()
    Tg,gg,bg=eng.get_linearization()
    eng.solve(1.0,tau,sigma,it,1e-3)
    x=eng.primal(); y=eng.dual(); ph,dh=eng.history()
    print('it',it,'x err',np.abs(x-xr).max(),'y err',np.abs(y-yr).max())
    bad=np.nonzero(np.abs(x-xr)>1e-12)[0]
    print('nbad',bad.size, bad[:10])
    Tu,g=oo.evaluate_displacement(T,(1.0,1.0),u0.reshape((-1,2),order='F'))
    b=Tu.reshape(-1,order='F')-R.reshape(-1,order='F')
    for q in bad[:6]:
        p=q%(m*n)
        ngs=g[p,0]**2+g[p,1]**2
        print(q,p,'x',x[q],xr[q],'g',g[p],'b',b[p],'ngs',ngs,'tl*ngs',tau*lam*ngs, 'gpu g',gg[p],gg[m*n+p],'gpu b',bg[p])
    print(ph[:3]); print(phr[:3])
    eng.close()
