import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
g.build()
import tensor_fft_amd as tf
from oracle import orc
def _c(a,b): return np.asarray(a,np.float64)+1j*np.asarray(b,np.float64)
def run(re,im):
    batch,n=re.shape
    dev=torch.from_numpy(np.ascontiguousarray(np.stack([re,im],axis=1))).cuda().reshape(-1)
    out=torch.empty_like(dev); p=tf.TfftPlan(n,batch,0); p.exec(dev,dev[n:],out,out[n:]); torch.cuda.synchronize()
    o=out.cpu().numpy().reshape(batch,2,n); return o[:,0],o[:,1]
def dist(g,r):
    gg=np.concatenate([g.real.ravel(),g.imag.ravel()]); rr=np.concatenate([r.real.ravel(),r.imag.ravel()])
    rms=np.sqrt(np.mean(rr*rr)); mx=np.abs(rr).max()
    d=np.abs(gg-rr)
    u_el=2.0**(np.floor(np.log2(np.maximum(np.abs(rr),max(rms,2.0**-14))))-10)
    return (d/u_el).max(), d.max()/2.0**(np.floor(np.log2(rms))-10), d.max()/2.0**(np.floor(np.log2(mx))-10)
for lg in (8,9,10,12,13,14,16,18,20):
    n=1<<lg
    for name in ("uniform","bench"):
        if name=="uniform":
            rng=np.random.default_rng(900+lg); re=rng.uniform(-1,1,(1,n)).astype(np.float16); im=rng.uniform(-1,1,(1,n)).astype(np.float16)
        else:
            a,b=orc.sine_superposition(n,orc.random_weights(10,42),orc.random_weights(10,4242),10); re,im=a[None],b[None]
        gr,gi=run(re,im); got=_c(gr,gi); ex=_c(*orc.dft64(re,im))
        row=f"N=2^{lg:2d} {name:8s} HIP-vs-fp64: elem {dist(got,ex)[0]:7.2f} rms {dist(got,ex)[1]:7.2f} max {dist(got,ex)[2]:6.2f} |"
        for mode in ((0,) if n<4096 else (0,1)):
            ref=_c(*orc.ref_fft(re,im,mode))
            a=dist(got,ref); b=dist(ref,ex)
            row+=f" mode{mode}: HIP-vs-restatement elem {a[0]:7.2f} rms {a[1]:7.2f} max {a[2]:6.2f}; restatement-vs-fp64 elem {b[0]:7.2f} rms {b[1]:7.2f} max {b[2]:6.2f} |"
        print(row, flush=True)
