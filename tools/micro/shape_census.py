import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    calls[('g',m,n,k)]+=1
    return og(self,a,w,bias,resid,act,out,M,lda,N,K,ldw,ldc)
def conv(self,x,B,Hh,Ww,cin,w,bias,resid=None,stride=1,pad=1,up=None):
    calls[('c',B,Hh,Ww,cin,w.shape[0],stride,up)]+=1
    return oc(self,x,B,Hh,Ww,cin,w,bias,resid,stride,pad,up)
Ops.gemm, Ops.conv3x3 = gemm, conv
sd = sd15.random_state_dict(sd15.unet_param_shapes(), seed=1)
eng = UNetEngine(sd,'cuda',VidToMe('cuda',seed=1))
text = torch.randn(2,154,768,device='cuda').half(); text_t=torch.randn(2,77,768,device='cuda').half()
def run(Fs,Hh,Ww,txt,n):
    for _ in range(n):
        x = torch.randn(2*sum(Fs),Hh,Ww,8,device='cuda').half()
        eng.forward_many(x,Fs,Hh,Ww,801.0,txt)
        eng.tome.reset_global_tokens()
# emulate a step's mix for config 2: 8 xy chunks (F=4, 90x120) w/ bank, 31 yt chunks (F=4, 30x90)
run([2]+[4]*7,90,120,text,2); xy=dict(calls); calls.clear()
run([4]*30,30,90,text_t,2); yt=dict(calls); calls.clear()
Ops.gemm, Ops.conv3x3 = og, oc
def timeit(fn,n=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
def bench(key):
    if key[0]=='g':
        _,M,N,K=key
        A=torch.randn(M,K,device='cuda').to(H); W=torch.randn(N,K,device='cuda').to(H); C=torch.empty(M,N,device='cuda',dtype=H)
        return timeit(lambda: L.tcl_gemm_f16(A,W,0,0,C,M,N,K,K,K,N,N,0,st())), 2.0*M*N*K
    _,B,Hh,Ww,ci,co,stride,up=key
    x=torch.randn(B,Hh,Ww,ci,device='cuda').to(H); w=torch.randn(co,9*ci,device='cuda').to(H)
    Hu,Wu = up if up else (Hh,Ww); Ho=(Hu-1)//stride+1; Wo=(Wu-1)//stride+1
    y=torch.empty(B,Ho,Wo,co,device='cuda',dtype=H)
    return timeit(lambda: L.tcl_conv3x3_f16(x,w,0,0,y,B,Hh,Ww,ci,co,stride,1,up[0] if up else 0,up[1] if up else 0,0,st())), 2.0*B*Ho*Wo*9*ci*co
for name,d,mult in (('xy',xy,1/2),('yt',yt,1/2)):
    rows=[]
    for k,c in d.items():
        ms,fl=bench(k); rows.append((ms*c*mult,k,c,ms,fl))
    rows.sort(reverse=True)
    tot=sum(r[0] for r in rows); totfl=sum(r[4]*r[2]*mult for r in rows)
    print(f"== {name}: per-step GEMM/conv time {tot:.1f} ms, {totfl/1e12:.1f} TFLOP -> {totfl/tot/1e9:.0f} TF/s")
    for t,k,c,ms,fl in rows[:22]: print(f"  {t:7.1f} ms  x{int(c*mult):4d}  {ms*1e3:8.1f} us  {fl/ms/1e9:6.0f} TF/s  {k}")
