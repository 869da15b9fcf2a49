import sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/acoustic-image-generation_amd")
import torch
from acimg import ops
dev = torch.device("cuda:0"); N = 32
g = torch.Generator().manual_seed(1)
for (H,W,C,K,stats) in [(56,75,32,512,True),(56,75,64,512,True),(56,75,128,512,True),(56,75,256,512,True),(56,75,128,512,False),(56,75,32,512,False)]:
    d = ops.conv_desc(N,H,W,C,K,1,1,1,"SAME")
    rows = N*H*W
    lo = -(-rows*C*2//256)*256
    x = torch.rand(rows, C, generator=g).to(dev)
    planes = torch.zeros(2*lo, dtype=torch.uint8, device=dev)
    plan = ops.Plan(dev, eager=True)
    ops.bn_relu_split(plan, x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1, planes, lo, rows, C)
    w = (torch.randn(1,1,C,K, generator=g)*0.05).to(dev)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
    ops.conv2d_split3_prepare(plan, d, w, wsplit)
    y = torch.empty(N,H,W,K, device=dev)
    st = torch.zeros(4096*2*K, device=dev) if stats else None
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=dev)
    for _ in range(3): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/10*1e3
    print("C=%3d K=%d stats=%s  %.1f us   out %.0f MB -> %.2f TB/s (out+in)" % (C,K,stats,us, rows*K*4/1e6, (rows*K*4+rows*C*4)/us/1e6))
