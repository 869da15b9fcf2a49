"""rate of the audio front end: 1024-sample int32 frames -> 12 MFCCs, HIP kernel vs the NumPy
restatement of dataloader/outdoor_data_mfcc.py:796-876 on the host; and find_logen + IoU (row f2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import numpy as np, torch
from acimg.frontend import FrontEnd
from acimg import evaluate
from oracle import frontend as ofe
dev = torch.device("cuda:0")
fe = FrontEnd(dev)
n = 1 << 18                                   # 262144 frames = 1 GiB of int32 samples
x = torch.randint(-30000, 30000, (n, 1024), dtype=torch.int32, device=dev)
out = torch.empty(n, 12, device=dev)
for _ in range(2): fe._build_spectrograms_function(x, True, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fe._build_spectrograms_function(x, True, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("GPU MFCC front end: %.2f Mframes/s, %.0f GB/s of int32 samples (HBM-bound kernel: 4 KiB in, 48 B out per frame)" %
      (n / dt / 1e6, n * 4096 / dt / 1e9))
xs = x[:2048].cpu().numpy()
t0 = time.perf_counter(); ref = np.stack([ofe.normalize_mfcc(v) for v in ofe.mfcc(xs)]); dt_cpu = time.perf_counter() - t0
print("NumPy restatement on one host core: %.1f kframes/s  (GPU/CPU = %.0fx)" % (2048 / dt_cpu / 1e3, (n / dt) / (2048 / dt_cpu)))
print("max |diff| vs NumPy on the sample: %.2e" % float(np.abs(out[:2048].cpu().numpy() - ref).max()))
ev = evaluate.EnergyIoU(dev)
N = 4096
a = torch.rand(N, 36, 48, 12, device=dev) * 4 - 2; b = a + 0.5 * torch.randn_like(a)
for _ in range(2): ev.iou(a, b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): iou = ev.iou(a, b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("find_logen x2 + mask IoU: %.0f kimages/s" % (N / dt / 1e3))
