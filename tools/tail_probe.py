"""times one trunk conv shape under forced tail-split factors; usage: tail_probe.py H W C K R stride"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
if sys.argv[1] == "child":
    import torch
    from acimg import ops
    H, W, C, K, R, s = [int(v) for v in sys.argv[2:8]]
    dev = torch.device("cuda:0"); N = 32
    d = ops.conv_desc(N, H, W, C, K, R, R, s, "SAME" if s == 1 else (1 if R == 3 else "SAME"))
    rows = N * H * W
    lo = -(-rows // 16) * 16 * C * 2     # acimg_split_plane_bytes: whole 16-pixel bricks
    g = torch.Generator().manual_seed(0)
    x = torch.rand(rows, C, generator=g).to(dev)
    planes = torch.zeros(2 * lo, dtype=torch.uint8, device=dev)
    one = torch.ones(C, device=dev); zero = torch.zeros(C, device=dev)
    plan = ops.Plan(dev, eager=True)
    ops.bn_relu_split(plan, x, one, zero, 1, planes, lo, rows, C)
    w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
    ops.conv2d_split3_prepare(plan, d, w, wsplit)
    y = torch.empty(N, d.OH, d.OW, K, device=dev)
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=dev)
    for _ in range(3): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, None, tail_ws=tws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, None, tail_ws=tws)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps(e0.elapsed_time(e1) / 10 * 1e3))
else:
    for s in ("0", "2", "3", "4", "6", "8", "12", "16"):
        env = dict(os.environ)
        if s == "0": env["ACIMG_NO_TAIL_SPLIT"] = "1"
        else: env["ACIMG_TAIL_S"] = s
        r = subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=env, capture_output=True, text=True)
        print("s=%-3s %s us" % (s, r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "FAILED " + r.stderr[-300:]))
