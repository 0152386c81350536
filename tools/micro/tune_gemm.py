import sys, os, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, collections, numpy as np
from tc_light_amd import sd15
from tc_light_amd.unet import UNetEngine, Ops
from tc_light_amd.vidtome import VidToMe
from tc_light_amd.lib import lib
L=lib(); H=torch.float16
def st(): return torch.cuda.current_stream().cuda_stream
calls=collections.Counter()
og, oc = Ops.gemm, Ops.conv3x3
def gemm(self,a,w,bias=None,resid=None,act=0,out=None,M=None,lda=None,N=None,K=None,ldw=None,ldc=None):
    n,k = (N,K) if N is not None else w.shape
    m = M if M is not None else a.numel()//k
    calls[('g',m,n,k,act)]+=1
    return og(self,a,w,bias,resid,act,out,M,lda,N,K,ldw,ldc)
def conv(self,x,B,Hh,Ww,cin,w,bias,resid=None,stride=1,pad=1,up=None):
    calls[('c',B,Hh,Ww,cin,w.shape[0],stride,up)]+=1
    return oc(self,x,B,Hh,Ww,cin,w,bias,resid,stride,pad,up)
Ops.gemm, Ops.conv3x3 = gemm, conv
sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
eng = UNetEngine(sd,'cuda',VidToMe('cuda',seed=1))
text = torch.randn(2,154,768,device='cuda').half(); text_t=torch.randn(2,77,768,device='cuda').half()
def run(F,Hh,Ww,txt,n):
    for _ in range(n):
        x = torch.randn(2*F,Hh,Ww,8,device='cuda').half()
        eng.forward_nhwc(x,F,Hh,Ww,801.0,txt)
run(4,90,120,text,2); xy=dict(calls); calls.clear()
run(4,30,90,text_t,2); yt=dict(calls); calls.clear()
Ops.gemm, Ops.conv3x3 = og, oc
def timeit(fn,n=6):
    try: fn()
    except RuntimeError: return None
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
def mk(key):
    if key[0]=='g':
        _,M,N,K,act=key
        A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H)
        return (lambda: L.tcl_gemm_f16(A,W,0,0,C,M,N,K,K,K,N//2 if act==2 else N,N,act,st())), 2.0*M*N*K
    _,B,Hh,Ww,ci,co,stride,up=key
    x=torch.randn(B,Hh,Ww,ci,device='cuda').to(H); w=torch.randn(co,9*ci,device='cuda').to(H)
    Hu,Wu = up if up else (Hh,Ww); Ho=(Hu-1)//stride+1; Wo=(Wu-1)//stride+1
    y=torch.empty(B,Ho,Wo,co,device='cuda',dtype=H)
    return (lambda: L.tcl_conv3x3_f16(x,w,0,0,y,B,Hh,Ww,ci,co,stride,1,up[0] if up else 0,up[1] if up else 0,0,st())), 2.0*B*Ho*Wo*9*ci*co
CFGS=[1,2,3,4,11,5,6,7,8]
res=[]
for name,d,mult in (('xy',xy,8/2),('yt',yt,31/2)):
    tot_def=tot_best=0
    rows=[]
    for k,c in d.items():
        fn,fl=mk(k)
        L.tcl_gemm_tune(0,0); t0=timeit(fn)
        best=(t0,0,0); alls={}
        for cfg in CFGS:
            for sp in ([1] if cfg in (5,6,7,8) else [1,2,3,4,6,8,12,16]):
                L.tcl_gemm_tune(cfg,sp); t=timeit(fn)
                if t is None: break
                alls[(cfg,sp)]=round(t,1)
                if t<best[0]: best=(t,cfg,sp)
        L.tcl_gemm_tune(0,0)
        n=c*mult
        tot_def+=t0*n; tot_best+=best[0]*n
        rows.append((t0*n, k, n, t0, best, alls))
    rows.sort(key=lambda r:-r[0])
    print(f"== {name}: default {tot_def/1e3:.1f} ms/step, best-per-shape {tot_best/1e3:.1f} ms/step")
    for tn_,k,n,t0,best,alls in rows:
        top=sorted(alls.items(), key=lambda kv: kv[1])[:4]
        print(f"  {tn_/1e3:6.1f} ms x{int(n):4d} def {t0:7.1f} us best {best[0]:7.1f} cfg{best[1]} sp{best[2]}  {k}  top: {top}")
        res.append(dict(pass_=name,key=list(map(str,k)),n=n,default=t0,best=best,alls={f"{a}_{b}":v for (a,b),v in alls.items()}))
json.dump(res, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/tune_gemm.json"),"w"))
