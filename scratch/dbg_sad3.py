import sys; sys.path.insert(0,'.')
import numpy as np
from ma_amd import engine as E, synth, drivers as D
from oracle import prox as px, operators as oo
from oracle.chambolle_pock import chambolle_pock as cp_ref
m,n=40,50
fr=synth.dynamic_test_image(m,n,2).numpy()
R=D.normalize(fr[0]); T=D.normalize(fr[1])
lam,tau=4.0,1.5
sigma=(1-1e-4)/(8*tau)
u0=np.zeros(2*m*n); v0=np.zeros(4*m*n)
K=oo.pairwise_gradient(m,n,(1.0,1.0),2)
F=lambda v,c,nargout: px.TV_registration(v,sigma,c,nargout)
eng=E.Engine(E.MODEL_TV_L1_REG,m,n,lam=lam)
eng.set_images(T.reshape(-1,order='F'),R.reshape(-1,order='F'))
for step in range(6):
    G=lambda u,c,nargout,u0_=u0.copy(): px.SAD_registration(u,u0_,T,R,(1.0,1.0),lam,tau,c,nargout)
    eng.set_primal(u0); eng.set_dual(v0)
    eng.linearize()
    Tg,gg,bg=eng.get_linearization()
    Tu,g=oo.evaluate_displacement(T,(1.0,1.0),u0.reshape((-1,2),order='F'))
    b=Tu.reshape(-1,order='F')-(g[:,0]*u0[:m*n]+g[:,1]*u0[m*n:])-R.reshape(-1,order='F')
    print('step',step,'lin err', np.abs(Tg-Tu.reshape(-1,order='F')).max(), np.abs(gg-g.reshape(-1,order='F')).max(), np.abs(bg-b).max())
    for it in (1,2,5,30):
        eng.set_primal(u0); eng.set_dual(v0)
        xr,yr,phr,dhr=cp_ref(F,G,K,u0,v0,1.0,tau,sigma,it)
        eng.solve(1.0,tau,sigma,it,1e-3)
        x=eng.primal(); y=eng.dual(); ph,dh=eng.history()
        ex=np.abs(x-xr).max(); ey=np.abs(y-yr).max()
        print('  it',it,'x err',ex,'y err',ey,'hist err',np.abs(ph-phr).max(),np.abs(dh-dhr).max())
        if ex>1e-9:
            bad=np.nonzero(np.abs(x-xr)>1e-9)[0]
            print('  nbad',bad.size,bad[:8])
            for q in bad[:5]:
                p=q%(m*n); ngs=g[p,0]**2+g[p,1]**2
                print('   ',q,p,(p%m,p//m),'x',x[q],xr[q],'g',g[p],'b',b[p],'ngs',ngs)
            break
    u0=xr.copy(); v0=yr.copy()
eng.close()
