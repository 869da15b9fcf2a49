"""times each distinct trunk conv shape (batch 32, random operands) under split3p staging variants; run on the GPU box
   usage: python tools/tune_dma.py "NAME=ENV1=V1,ENV2=V2" ...   (NAME 'base' = no env)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
SHAPES = [(56,75,64,64,3,1,2),(56,75,64,256,1,1,3),(56,75,256,64,1,1,2),(56,75,256,128,1,1,1),(56,75,128,128,3,1,3),(56,75,128,512,1,1,4),
          (56,75,256,512,1,1,1),(56,75,512,128,1,1,3),(56,75,128,128,3,2,1),(28,38,128,512,1,1,1),(28,38,512,256,1,1,1),(28,38,256,256,3,1,5),
          (28,38,256,1024,1,1,6),(28,38,512,1024,1,1,1),(28,38,1024,256,1,1,5),(28,38,256,256,3,2,1),(14,19,256,1024,1,1,1),(14,19,1024,512,1,1,1),
          (14,19,512,512,3,1,3),(14,19,512,2048,1,1,3),(14,19,1024,2048,1,1,1),(14,19,2048,512,1,1,2)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from acimg import ops
    dev = torch.device("cuda:0")
    res = {}
    N = 32
    g = torch.Generator(device="cpu").manual_seed(1)
    for (H,W,C,K,R,s,cnt) in SHAPES:
        d = ops.conv_desc(N,H,W,C,K,R,R,s,"SAME" if s == 1 else (1 if R == 3 else "SAME"))
        rows = N*H*W
        lo = -(-rows*C*2//256)*256
        x = torch.rand(rows, C, generator=g).to(dev)
        planes = torch.zeros(2*lo, dtype=torch.uint8, device=dev)
        one = torch.ones(C, device=dev); zero = torch.zeros(C, device=dev)
        plan = ops.Plan(dev, eager=True)
        ops.bn_relu_split(plan, x, one, zero, 1, planes, lo, rows, C)
        w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
        wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
        ops.conv2d_split3_prepare(plan, d, w, wsplit)
        y = torch.empty(N, d.OH, d.OW, K, device=dev)
        stats = torch.zeros(4096*2*K, device=dev)
        tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=dev)
        for _ in range(3): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, stats, tail_ws=tws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, stats, tail_ws=tws)
        e1.record(); torch.cuda.synchronize()
        res["%dx%d %d->%d %dx%d/%d" % (H,W,C,K,R,R,s)] = (e0.elapsed_time(e1) / 10 * 1e3, float(y.double().abs().sum().item()))
    print(json.dumps(res))
else:
    variants = [v.split("=", 1) for v in sys.argv[1:]] or [["base", ""]]
    out = {}
    for name, envs in variants:
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        if r.returncode:
            print(name, "FAILED", r.stderr[-2000:])
            continue
        out[name] = json.loads(r.stdout.strip().split("\n")[-1])
    names = list(out.keys())
    keys = list(out[names[0]].keys())
    print("%-28s %3s" % ("shape", "n") + "".join(" %14s" % n for n in names) + "   TF(best)  checksum-match")
    tot = {n: 0.0 for n in names}
    for k, sh in zip(keys, SHAPES):
        H,W,C,K,R,s,cnt = sh
        oh, ow = (H, W) if s == 1 else ((H + 1) // 2, (W + 1) // 2)
        fl = 2.0 * 32 * oh * ow * K * R * R * C
        best = min(out[n][k][0] for n in names)
        cs = [out[n][k][1] for n in names]
        ok = all(abs(c - cs[0]) <= 1e-6 * abs(cs[0]) for c in cs)
        print("%-28s %3d" % (k, cnt) + "".join(" %14.1f" % out[n][k][0] for n in names) + "   %7.1f   %s" % (fl / best / 1e6, ok))
        for n in names: tot[n] += cnt * out[n][k][0]
    print("%-32s" % "trunk total (us, weighted)" + "".join(" %14.1f" % tot[n] for n in names))
