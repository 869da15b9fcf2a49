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
kin, hn = 28416, 300
d=ops.conv_desc(N,1,1,kin,hn,1,1,1,"VALID",ldx=kin,ldy=hn,ldw=hn)
x=torch.randn(N,kin,generator=g).to(dev); w=(torch.randn(kin,hn,generator=g)*0.01).to(dev); b=torch.zeros(hn,device=dev)
y=torch.zeros(N,hn,device=dev); gy=torch.randn(N,hn,generator=g).to(dev); dx=torch.zeros(N,kin,device=dev); dw=torch.zeros(kin,hn,device=dev); db=torch.zeros(hn,device=dev)
plan=ops.Plan(dev,eager=True)
for cfg in [dict(), dict(splitk_handoff=0), dict(splitk_target=192), dict(splitk_target=384), dict(splitk_target=2048)]:
    _lib.configure(**cfg)
    tf=timeit(lambda: ops.conv2d_fwd(plan,d,x,w,b,y))
    td=timeit(lambda: ops.conv2d_dgrad(plan,d,gy,hn,w,dx))
    tw=timeit(lambda: ops.conv2d_wgrad(plan,d,x,gy,hn,dw,db))
    print("heads 32x28416->300 %-26s fwd %6.1f dgrad %6.1f wgrad %6.1f us (weights 34 MB: 5.4 us at 6.3 TB/s)"%(str(cfg),tf,td,tw))
_lib.configure()
ref=(x.double()@w.double()).float()
print("fwd err", float((y-ref).abs().max()/ref.abs().max()))
