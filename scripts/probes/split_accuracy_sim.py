"""CPU simulation behind the two-term fp16 split of the CNN contractions (cnn.h, CNN_SPLIT = 2): the trained networks of the three
proteins (tests/golden/real_<tag>_cnn.npz) evaluated in fp64, in fp32 (numpy, what a reference-like evaluation rounds) and with
every contraction operand replaced by its two fp16 terms (products and sums exact): errors of fitness and input gradient against
fp64, as fractions of the tolerances of the parity tests.   python scripts/probes/split_accuracy_sim.py"""
import numpy as np, sys
import os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rng=np.random.default_rng(0)
def split16(a, bound=None, cross=True):
    a=np.asarray(a,np.float64)
    b = np.abs(a).max() if bound is None else bound
    s = 2.0**np.floor(np.log2(32768.0/b))
    x=a*s
    a1=x.astype(np.float16).astype(np.float64)
    a2=(x-a1).astype(np.float32).astype(np.float16).astype(np.float64)
    return a1/s, a2/s
def mm16(A,B,boundA=None):
    a1,a2=split16(A,boundA); b1,b2=split16(B)
    return a1@b1 + a1@b2 + a2@b1
def run(tag,L):
    d=np.load(os.path.join(REPO, 'tests', 'golden', f'real_{tag}_cnn.npz'))
    n=256
    idx=rng.integers(0,20,size=(n,L))
    X=np.zeros((n,20,L)); 
    for i in range(n): X[i,idx[i],np.arange(L)]=1
    res={}
    for mode in ('f64','f32','s16'):
        fits=[];grads=[]
        for k in range(3):
            W=d[f'net{k}.encoder.weight'].astype(np.float64); b=d[f'net{k}.encoder.bias'].astype(np.float64)
            We=d[f'net{k}.embedding.0.weight'].astype(np.float64); be=d[f'net{k}.embedding.0.bias'].astype(np.float64)
            wd=d[f'net{k}.decoder.weight'].astype(np.float64)[0]; bd=float(d[f'net{k}.decoder.bias'][0])
            C,_,K=W.shape; T=L-K+1
            if mode=='s16':
                w1,w2=split16(W); Wc=w1+w2   # table in 2-term fp16
            else: Wc=W
            # conv
            pre=np.zeros((n,T,C))
            for t in range(K):
                # pre[n,t',c] += Wc[c,letter(n,t'+t),t]
                pre+=Wc[:, idx[:, t:t+T], t].transpose(1,2,0)
            pre+=b
            if mode=='f32': pre=pre.astype(np.float32).astype(np.float64)
            h1=np.maximum(pre,0)
            bound=(np.abs(b)+np.abs(W).max(axis=1).sum(axis=1)).max()
            if mode=='f64': h2=h1@We.T
            elif mode=='f32': h2=(h1.astype(np.float32)@We.T.astype(np.float32)).astype(np.float64)
            else: h2=mm16(h1.reshape(-1,C),We.T,bound).reshape(n,T,-1)
            h2=np.maximum(h2+be,0)
            am=h2.argmax(1); m=h2.max(1)
            fit=bd+m@wd
            # backward: G[n,t,c] = sum_f [am[n,f]==t] wd_f (m>0) We[f,c]; gate
            coef=wd*(m>0)
            G=np.zeros((n,T,C))
            for i in range(n):
                np.add.at(G[i], am[i], coef[i][:,None]*We)
            G*= (pre>0)
            boundG=(np.abs(wd)[:,None]*np.abs(We)).sum(0).max()
            Wt=W.transpose(0,2,1).reshape(C,K*20)  # [c][tap*20+letter]
            if mode=='f64': O=G@Wt
            elif mode=='f32': O=(G.astype(np.float32)@Wt.astype(np.float32)).astype(np.float64)
            else: O=mm16(G.reshape(-1,C),Wt,boundG).reshape(n,T,K*20)
            g=np.zeros((n,L,20))
            for t in range(K):
                g[:,t:t+T,:]+=O[:,:,t*20:(t+1)*20]
            fits.append(fit);grads.append(g)
        res[mode]=(np.mean(fits,0),np.mean(grads,0),am)
    f64=res['f64']
    for mode in ('f32','s16'):
        fe=np.abs(res[mode][0]-f64[0]); ge=np.abs(res[mode][1]-f64[1]).reshape(n,-1).max(1)
        same=(res[mode][2]==f64[2]).all(1)
        gm=np.abs(f64[1]).max()
        print(tag,mode,'fit err max %.2e (tol 5e-6 -> %.2f) | grad err max (same argmax chains %d/%d) %.2e, max|g| %.2f tol ratio %.2f'%(fe.max(),fe.max()/5e-6,same.sum(),n,ge[same].max(),gm,ge[same].max()/(2e-6*max(1,gm))))
run('pabp',96); run('ube4b',104); run('gfp',237)
