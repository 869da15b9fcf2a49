"""times each distinct trunk conv shape (batch 32) under each split3p tiling; run on the GPU box"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from acimg import ops
    dev = torch.device("cuda:0")
    shapes = [(56,75,64,64,3,1),(56,75,64,256,1,1),(56,75,256,64,1,1),(56,75,256,128,1,1),(56,75,128,128,3,1),(56,75,128,512,1,1),
              (56,75,256,512,1,1),(56,75,512,128,1,1),(56,75,128,128,3,2),(28,38,128,512,1,1),(28,38,512,256,1,1),(28,38,256,256,3,1),
              (28,38,256,1024,1,1),(28,38,512,1024,1,1),(28,38,1024,256,1,1),(28,38,256,256,3,2),(14,19,256,1024,1,1),(14,19,1024,512,1,1),
              (14,19,512,512,3,1),(14,19,512,2048,1,1),(14,19,1024,2048,1,1),(14,19,2048,512,1,1)]
    res = {}
    N = 32
    for (H,W,C,K,R,s) in shapes:
        d = ops.conv_desc(N,H,W,C,K,R,R,s,"SAME" if s == 1 else (1 if R == 3 else "SAME"))
        rows = N*H*W
        lo = -(-rows*C*2//256)*256
        planes = torch.zeros(2*lo, dtype=torch.uint8, device=dev)
        wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
        y = torch.empty(N, d.OH, d.OW, K, device=dev)
        stats = torch.zeros(4096*2*K, device=dev)
        plan = ops.Plan(dev, eager=True)
        for _ in range(3): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, stats)
        e1.record(); torch.cuda.synchronize()
        res["%dx%d %d->%d %dx%d/%d" % (H,W,C,K,R,R,s)] = e0.elapsed_time(e1) / 10 * 1e3
    print(json.dumps(res))
else:
    out = {}
    for tile in ("128x128", "64x128", "128x64"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, ACIMG_SPLIT3_TILE=tile), capture_output=True, text=True)
        out[tile] = json.loads(r.stdout.strip().split("\n")[-1])
    keys = list(out["128x128"].keys())
    print("%-28s %10s %10s %10s" % ("shape", "128x128", "64x128", "128x64"))
    for k in keys:
        print("%-28s %10.1f %10.1f %10.1f" % (k, out["128x128"][k], out["64x128"][k], out["128x64"][k]))
