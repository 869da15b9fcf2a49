"""runs one split3p trunk conv shape a few times (for rocprofv3 --pmc); args: H W C K R stride [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg import ops
H, W, C, K, R, s = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0"); N = 32
d = ops.conv_desc(N, H, W, C, K, R, R, s, "SAME" if s == 1 else (1 if R == 3 else "SAME"))
rows = N * H * W
lo = -(-rows * C * 2 // 256) * 256
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, C, generator=g).to(dev)
planes = torch.zeros(2 * lo, dtype=torch.uint8, device=dev)
w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
y = torch.empty(N, d.OH, d.OW, K, device=dev)
stats = torch.zeros(4096 * 2 * K, device=dev)
plan = ops.Plan(dev, eager=True)
ops.bn_relu_split(plan, x, None, None, 1, planes, lo, rows, C)
ops.conv2d_split3_prepare(plan, d, w, wsplit)
for _ in range(iters):
    ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, stats)
torch.cuda.synchronize()
print("done", ops.conv2d_fwd_split3_tiling(d))
