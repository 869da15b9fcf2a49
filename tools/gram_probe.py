"""per-kernel timing of acimg_gram_stats on the three trunk shapes at batch 32 (run under rocprofv3 --kernel-trace --stats):
   python tools/gram_probe.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch  # noqa: E402

from acimg import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
plan = ops.Plan(dev, eager=True)
for rows, Cc, K in ((134400, 64, 256), (134400, 128, 512), (34048, 256, 1024), (8512, 512, 2048)):
    lo = -(-rows // 16) * 16 * Cc * 2
    x = torch.relu(torch.randn(rows, Cc, device=dev) + 0.3)
    xp = torch.zeros(lo * 2, dtype=torch.uint8, device=dev)
    ops.bn_relu_split(plan, x, torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), 1, xp, lo, rows, Cc)
    w = torch.randn(Cc, K, device=dev) * (2.6 / Cc) ** 0.5
    ws = torch.zeros(ops.gram_stats_workspace(rows, Cc), dtype=torch.uint8, device=dev)
    sc, sh = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    mm, mv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.gram_stats(plan, xp, lo, rows, Cc, w, K, K, g, b, mm, mv, sc, sh, ws)
    e0.record()
    for _ in range(reps):
        ops.gram_stats(plan, xp, lo, rows, Cc, w, K, K, g, b, mm, mv, sc, sh, ws)
    e1.record()
    torch.cuda.synchronize()
    print("rows %6d C %3d K %4d: %.1f us per call (3 launches), workspace %.1f MB" % (rows, Cc, K, e0.elapsed_time(e1) / reps * 1e3,
                                                                                  ws.numel() / 1e6), flush=True)
