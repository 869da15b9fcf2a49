"""weight gradient / forward of a 3x3 layer with and without the producer's affine applied on load: event timings (round 4)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
from acimg import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (N, H, W, Cc, K, prec) in [(32, 224, 298, 8, 8, 0), (32, 112, 149, 8, 32, 0), (32, 112, 149, 32, 32, 2), (32, 224, 298, 16, 8, 0)]:
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME")
    plan = ops.Plan(dev, eager=True)
    x = torch.randn(N, H, W, Cc, device=dev)
    xm = torch.relu(x)
    sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev)
    gy = torch.randn(N, H, W, K, device=dev) * 1e-3
    w = torch.randn(3, 3, Cc, K, device=dev) * 0.1
    b = torch.zeros(K, device=dev)
    dw, db = torch.zeros(3, 3, Cc, K, device=dev), torch.zeros(K, device=dev)
    y = torch.zeros(N, H, W, K, device=dev)
    bf16 = prec == 2
    if prec:
        wimg = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
        ops.conv2d_split3_prepare(plan, d, w, wimg, bf16=bf16)
        t_w0 = timeit(lambda: ops.conv2d_wgrad_split3(plan, d, xm, gy, K, dw, db, bf16=bf16))
        t_f0 = timeit(lambda: ops.conv2d_fwd_split3(plan, d, xm, wimg, y, bias=b, bf16=bf16))
        t_f1 = timeit(lambda: ops.conv2d_fwd_split3(plan, d, x, wimg, y, bias=b, bf16=bf16, in_scale=sc, in_shift=sh, in_relu=1))
    else:
        t_w0 = timeit(lambda: ops.conv2d_wgrad(plan, d, xm, gy, K, dw, db))
        t_f0 = timeit(lambda: ops.conv2d_fwd(plan, d, xm, w, b, y))
        t_f1 = timeit(lambda: ops.conv2d_fwd(plan, d, x, w, b, y, in_scale=sc, in_shift=sh, in_relu=1))
    t_w1 = timeit(lambda: ops.conv2d_wgrad_affine(plan, d, prec, x, sc, sh, gy, K, dw, db))
    t_w2 = timeit(lambda: ops.conv2d_wgrad_affine(plan, d, prec, xm, sc, sh, gy, K, dw, db))
    print((N, H, W, Cc, K, prec), "wgrad %.1f us, with affine %.1f us (same kernel on the rectified tensor %.1f) | forward %.1f us, with affine %.1f us"
          % (t_w0, t_w1, t_w2, t_f0, t_f1))
