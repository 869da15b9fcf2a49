import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg import _lib, ops
dev = torch.device("cuda:0")
_lib.load()
N=32
g=torch.Generator().manual_seed(1)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    best=1e9
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/n*1e3)
    return best
for (H,W,C,K,R) in [(12,16,128,128,3),(12,16,136,133,3),(12,16,128,133,3)]:
    d=ops.conv_desc(N,H,W,C,K,R,R,1,"SAME", ldx=(C+3)//4*4)
    Cp=(C+3)//4*4; Kp=(K+3)//4*4
    x=torch.randn(N,H,W,Cp,generator=g).to(dev); gy=(torch.randn(N,H,W,Kp,generator=g)*1e-3).to(dev)
    w=(torch.randn(R,R,C,Kp,generator=g)*0.05).to(dev); b=torch.zeros(Kp,device=dev)
    y=torch.zeros(N,H,W,Kp,device=dev); dx=torch.zeros(N,H,W,Cp,device=dev); dw=torch.zeros(R,R,C,Kp,device=dev); db=torch.zeros(Kp,device=dev)
    plan=ops.Plan(dev,eager=True)
    for cfg in [dict(), dict(splitk_handoff=0), dict(splitk_target=192), dict(splitk_target=384), dict(splitk_target=1536), dict(splitk_cut=1)]:
        _lib.configure(**cfg)
        tf=timeit(lambda: ops.conv2d_fwd(plan,d,x,w,b,y))
        td=timeit(lambda: ops.conv2d_dgrad(plan,d,gy,Kp,w,dx))
        tw=timeit(lambda: ops.conv2d_wgrad(plan,d,x,gy,Kp,dw,db))
        print("%dx%d %d->%d  %-28s fwd %6.1f  dgrad %6.1f  wgrad %6.1f us"%(H,W,C,K,str(cfg),tf,td,tw))
    _lib.configure()
# launch floor: an empty-ish kernel
z=torch.zeros(1024,device=dev)
print("zero kernel (launch floor in a back-to-back stream): %.1f us"%timeit(lambda: ops.zero(plan,z,1024) if hasattr(ops,'zero') else z.zero_()))
