"""conv_few16_kernel against fp64, forward and data gradient, with the error broken down (round 4 debugging aid)"""
import os
import sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
from acimg import ops  # noqa: E402

dev = torch.device("cuda:0")
for (N, H, W, Cc, K) in [(2, 200, 180, 8, 8), (2, 147, 161, 8, 32), (3, 150, 161, 16, 8), (2, 151, 170, 8, 16)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, Cc, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, Cc, K, generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(K, generator=g, dtype=torch.float64)
    gy = torch.randn(N, H, W, K, generator=g, dtype=torch.float64)
    res = torch.randn(N, H, W, Cc, generator=g, dtype=torch.float64)
    xr = x.permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, w.permute(3, 2, 0, 1), b, padding=1)
    (gx,) = torch.autograd.grad(yr, (xr,), gy.permute(0, 3, 1, 2))
    yref, gxref = yr.detach().permute(0, 2, 3, 1), gx.permute(0, 2, 3, 1)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME")
    plan = ops.Plan(dev, eager=True)
    y = torch.zeros(N, H, W, K, device=dev)
    rows = ops.conv2d_stats_rows(d)
    st = torch.zeros(rows, 2, K, device=dev)
    ops.conv2d_fwd(plan, d, x.float().to(dev), w.float().to(dev), b.float().to(dev), y, stats=st)
    torch.cuda.synchronize()
    e = (y.cpu().double() - yref).abs()
    print((N, H, W, Cc, K), "rows", rows, "fwd err %.3e" % (e.max() / yref.abs().max()), "per channel", (e.amax((0, 1, 2)) / yref.abs().max()).tolist()[:8])
    for with_res in (False, True):
        dx = torch.full((N, H, W, Cc), 3.0, device=dev)
        ops.conv2d_dgrad(plan, d, gy.float().to(dev), K, w.float().to(dev), dx, res.float().to(dev) if with_res else None, Cc if with_res else 0)
        torch.cuda.synchronize()
        ref = gxref + (res if with_res else 0)
        e = (dx.cpu().double() - ref).abs()
        print("   dgrad res=%s err %.3e" % (with_res, e.max() / ref.abs().max()), "per channel", ["%.1e" % v for v in (e.amax((0, 1, 2)) / ref.abs().max()).tolist()[:16]],
              "untouched", int((dx == 3.0).sum()), "corr", float((dx.cpu().double() * ref).sum() / (ref * ref).sum()))

# the sequence of tests/test_ops_gpu.py::test_few_channel_direct_conv[case0]
print("--- test sequence")
N, H, W, Cc, K = 2, 200, 180, 8, 8
g = torch.Generator().manual_seed(5 + 2 + 200 + 180 + 8 + 8 + 3 + 3 + 1)
x = torch.randn(N, H, W, Cc, generator=g, dtype=torch.float64)
w = torch.randn(3, 3, Cc, K, generator=g, dtype=torch.float64) * 0.2
b = torch.randn(K, generator=g, dtype=torch.float64)
d1 = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME", act=1)
d0 = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME", act=0)
plan = ops.Plan(dev, eager=True)
y = torch.zeros(N, H, W, K, device=dev)
stats = torch.zeros(ops.conv2d_stats_rows(d1), 2, K, device=dev)
wd, xd, bd = w.float().to(dev), x.float().to(dev), b.float().to(dev)
for first in (False, True):
    if first:
        ops.conv2d_fwd(plan, d1, xd, wd, bd, y, stats=stats)
        torch.cuda.synchronize()
    ops.conv2d_fwd(plan, d0, xd, wd, bd, y, stats=stats)
    torch.cuda.synchronize()
    gy = torch.randn(N, H, W, K, generator=g, dtype=torch.float64)
    res = torch.randn(N, H, W, Cc, generator=g, dtype=torch.float64)
    xr = x.permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, w.permute(3, 2, 0, 1), b, padding=1)
    (gx,) = torch.autograd.grad(yr, (xr,), gy.permute(0, 3, 1, 2))
    dx = torch.full((N, H, W, Cc), 3.0, device=dev)
    gyd, resd = gy.float().to(dev), res.float().to(dev)
    ops.conv2d_dgrad(plan, d0, gyd, K, wd, dx, resd, Cc)
    torch.cuda.synchronize()
    ref = gx.permute(0, 2, 3, 1) + res
    e = (dx.cpu().double() - ref).abs().max() / ref.abs().max()
    e2 = (dx.cpu().double() - res).abs().max()
    wsb = plan.ws.buf
    print("direct fwd first =", first, "dgrad err %.3e" % e, "|dx - res| max %.3e" % e2, "ws bytes", None if wsb is None else wsb.numel(),
          "image nonzero", None if wsb is None else int((wsb.view(torch.uint8)[:6144] != 0).sum()))
