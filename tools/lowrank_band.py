#!/usr/bin/env python3
"""Offline (CPU: numpy + the oracle) emulation of the LOW-RANK centred-remainder screening pass (kernels.h: kLrK):
the band of the projected operand (orthonormal basis B of the HAF slots' span, y^ = fp16(B^'p^), q~_n = (B'B^)^-1 B'q_n,
r_n = (I - BB')q_n - (B^ - B)q~_n) term by term against the band of the full-rank form, the share of evaluations each leaves
undecided, and the actual error of an emulated pass (sums exact) as a fraction of the band -- for the bench generator's seeds and the
trained 8964-SV model on the C5 cloud.  The noise |nu| = |p - p_lin| enters with a factor (x1, x2, x4) to show what a loose bound costs.

  python tools/lowrank_band.py [grid]          (default 512; about a minute)
Not a test; no GPU.
"""
import sys,os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'tools'))
import numpy as np, scipy.linalg as sl, tempfile
from lowrank_study import *
from centre_study import psi
LN2=np.log(2.0)
tmp=tempfile.mkdtemp(); tiny=os.path.join(tmp,'t.model'); models.write_random_model(tiny,8,seed=1,balanced=True)
o=O.Oracle(F,R,tiny)
Afull=linear_map(o); A=Afull[:302]
lower,upper,fmin,fmax,present=o.range_table()
alpha=np.array([(upper-lower)/(fmax[a+1]-fmin[a+1]) for a in range(302)])
As=A*alpha[:,None]
Uu,ss,Vt=np.linalg.svd(As,full_matrices=False)
r=158; BH=Uu[:,:r]
D=323
B=np.zeros((D,r+21)); B[:302,:r]=BH; B[302:,r:]=np.eye(21)
Bh=f16(B); G=B.T@Bh; Gi=np.linalg.inv(G)
print('sigma(B^) %.5f sigma(|B^|) %.3f cond(G) %.4f'%(np.linalg.norm(Bh,2),np.linalg.norm(np.abs(Bh),2),np.linalg.cond(G)))
grid=int(sys.argv[1]) if len(sys.argv)>1 else 512
wins,_=windows(o,grid,500)
X,XL=rows(o,wins,Afull,323)
KAP=12.0
def cr_consts(Q,Qh,b):
    Bm=b[:,None]; qn,qhn,dqn=np.linalg.norm(Q,axis=1),np.linalg.norm(Qh,axis=1),np.linalg.norm(Qh-Q,axis=1)
    return dict(qhmax=qhn.max(),dqmax=dqn.max(),Babs=np.abs(b).sum(),Cq1=(np.abs(b)*qhn).sum())
for name,kind,poly in (('random42',42,False),('random11',11,False),('random7',7,False),('trained','t',True)):
    if kind=='t':
        pth=os.path.join(tmp,'tr.model'); models.unpack_trained_model(os.path.join(ROOT,'tests','golden','trained.model.npz'),pth)
    else:
        pth=os.path.join(tmp,'r%s.model'%kind); models.write_random_model(pth,4096,seed=kind,balanced=True,rho=0.01,gamma=1.0/323)
    om=O.Oracle(F,R,pth); m=om.model_arrays(); g=m['gamma']; c=np.sqrt(2*g*np.log2(np.e)); coef=m['coef']; rho=m['rho']
    U,UL,V=X*c,XL*c,m['sv']*c
    t=-0.5*(V*V).sum(1); ax=0.5*(U*U).sum(1)
    K=np.exp2(U@V.T+t[None,:]-ax[:,None]); dec=K@coef-rho
    wgt=np.abs(coef)*np.exp2(t); mu=(wgt[:,None]*V).sum(0)/wgt.sum()
    # affine subspace of exact-linear HAF attributes: UL_H = dH + BH y ; project mu_H into it
    ULH=UL[:,:302]; dH=(ULH-(ULH@BH)@BH.T).mean(0)
    assert np.abs(ULH-(ULH@BH)@BH.T-dH).max()<1e-9
    mu2=mu.copy(); mu2[:302]=dH+BH@(BH.T@(mu[:302]-dH))
    P,Q=U-mu2,V-mu2
    b=coef*np.exp2(-0.5*(Q*Q).sum(1)); A_=np.exp2(-0.5*(P*P).sum(1)); Bm=b[:,None]
    Ph,Qh=f16(P),f16(Q)
    pn,phn,dn=np.linalg.norm(P,axis=1),np.linalg.norm(Ph,axis=1),np.linalg.norm(Ph-P,axis=1)
    nu=np.zeros_like(U); nu[:,:302]=(U-UL)[:,:302]; nun=np.linalg.norm(nu,axis=1)
    pperp=P-(P@B)@B.T
    print('== %s: |p| %.3g dn %.3g |nu| %.3g |p_perp| %.3g'%(name,np.median(pn),np.median(dn),np.median(nun),np.median(np.linalg.norm(pperp,axis=1))))
    def full_band():
        nN=sigma_upper((Q*Bm).T@Qh); Ms=(Q*Bm).T@(Qh-Q); nM=sigma_upper(0.5*(Ms+Ms.T))
        qn,qhn,dqn=np.linalg.norm(Q,axis=1),np.linalg.norm(Qh,axis=1),np.linalg.norm(Qh-Q,axis=1)
        nHabs,nDabs=sigma_upper(Qh*np.sqrt(np.abs(Bm)))**2,sigma_upper((Qh-Q)*np.sqrt(np.abs(Bm)))**2
        nHaa=sigma_upper(np.abs(Qh)*np.sqrt(np.abs(Bm)))**2
        accr=10*KAP*U24
        eps=dn*qhn.max()+pn*dqn.max()+accr*phn*qhn.max(); zmax=phn*qhn.max()+eps
        acc_sum=np.minimum(phn*pn*(np.abs(b)*qhn*qn).sum(),(np.sqrt(nHabs)+np.sqrt(nDabs))*pn*np.sqrt(nHaa)*phn)
        quad1=LN2**2*(nN*pn*dn+nM*pn**2+accr*acc_sum); quad2=1.5*LN2**2*(nHabs*dn**2+nDabs*pn**2)
        cub2=LN2**2*(np.exp2(zmax)-1)*eps**2*np.abs(b).sum()
        kpsi=LN2*eps*1.01; cabs=quad1+quad2+cub2
        if poly: kpsi=kpsi+4.1*(LN2*zmax)**4/360+6e-7
        else: kpsi=kpsi+2.4e-7; cabs=cabs+2.4e-7*(np.abs(b).sum()+LN2*phn*(np.abs(b)*qhn).sum())
        Spsi=psi(Ph@Qh.T)@np.abs(b); g0=(34+(len(b)/2/32)/8)*U24
        return A_*(cabs+(g0+kpsi)*Spsi), dict(q1=np.median(A_*quad1),cub2=np.median(A_*cub2),kS=np.median(A_*kpsi*Spsi),eps=np.median(eps))
    def lr_band(nufac):
        Qt=(Q@B)@Gi.T          # rows q~_n = G^-1 B' q_n
        Qth=f16(Qt); dQt=Qth-Qt
        Rm=(Q-(Q@B)@B.T)-Qt@(Bh-B).T      # r_n
        y32=Ph@Bh; yh=f16(y32); ye=P@Bh
        sdy=np.linalg.norm(yh-y32,axis=1)
        sB,sAB=np.linalg.norm(Bh,2),np.linalg.norm(np.abs(Bh),2)
        dyn=sdy+10*KAP*U24*sAB*phn+sB*dn
        yen=sB*pn; yhn=yen+dyn
        nn=nufac*nun+5e-6*pn
        qtn,qthn,dqtn,rn=np.linalg.norm(Qt,axis=1),np.linalg.norm(Qth,axis=1),np.linalg.norm(dQt,axis=1),np.linalg.norm(Rm,axis=1)
        accr=6*KAP*U24
        eps=dyn*qthn.max()+yen*dqtn.max()+accr*yhn*qthn.max()+nn*rn.max(); zmax=yhn*qthn.max()+eps
        nN1=sigma_upper((Q*Bm).T@Qth); M1=((Q*Bm).T@dQt)@Bh.T; nM1=sigma_upper(0.5*(M1+M1.T)); nN2=sigma_upper((Q*Bm).T@Rm)
        sb=np.sqrt(np.abs(Bm))
        nHabs,nDabs,nRabs=sigma_upper(Qth*sb)**2,sigma_upper(dQt*sb)**2,sigma_upper(Rm*sb)**2
        sQb,sQtaa=sigma_upper(Q*sb),sigma_upper(np.abs(Qth)*sb)
        acc_sum=np.minimum(yhn*pn*(np.abs(b)*qthn*np.linalg.norm(Q,axis=1)).sum(), sQb*pn*sQtaa*yhn)
        quad1=LN2**2*(nN1*pn*dyn+nM1*pn**2+accr*acc_sum+nN2*pn*nn)
        quad2=2*LN2**2*(nHabs*dyn**2+nDabs*yen**2+nRabs*nn**2)
        cub2=LN2**2*(np.exp2(zmax)-1)*eps**2*np.abs(b).sum()
        kpsi=LN2*eps*1.01; cabs=quad1+quad2+cub2
        if poly: kpsi=kpsi+4.1*(LN2*zmax)**4/360+6e-7
        else: kpsi=kpsi+2.4e-7; cabs=cabs+2.4e-7*(np.abs(b).sum()+LN2*yhn*(np.abs(b)*qthn).sum())
        zt=yh@Qth.T
        Spsi=psi(zt)@np.abs(b); g0=(34+(len(b)/2/32)/8)*U24
        # actual error of the emulated low-rank pass (sums exact)
        ztrue=P@Q.T
        err=np.abs(A_*((psi(zt)-psi(ztrue))@b))
        return A_*(cabs+(g0+kpsi)*Spsi), dict(q1=np.median(A_*quad1),q1_dy=np.median(A_*LN2**2*nN1*pn*dyn),q1_nu=np.median(A_*LN2**2*nN2*pn*nn),cub2=np.median(A_*cub2),kS=np.median(A_*kpsi*Spsi),eps=np.median(eps),sdy=np.median(sdy),dyn=np.median(dyn),qtmax=qthn.max(),rmax=rn.max()),err
    bf,tf=full_band()
    print('   full   : undecided %.4f band median %.3g'%(np.mean(np.abs(dec)<=bf),np.median(bf)),{k:'%.3g'%v for k,v in tf.items()})
    for nf in (1.0,2.0,4.0):
        bl,tl,err=lr_band(nf)
        print('   lowrank (nu x%.0f): undecided %.4f band median %.3g max err/band %.3f'%(nf,np.mean(np.abs(dec)<=bl),np.median(bl),(err/bl).max()),{k:'%.3g'%v for k,v in tl.items()} if nf==1 else '')
