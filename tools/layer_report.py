"""Per-launch report for the implicit-GEMM kernels: aligns a rocprofv3 kernel trace of `bench.py` with
the recorded plan (rebuilt on the CPU, no GPU needed) and prints time, TFLOP/s and GB/s per launch.
usage: python tools/layer_report.py <kernel_trace.csv> [batch]"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg.flags import FLAGS
from acimg.session import Session
from acimg.trainer import Trainer
from acimg.unet_acresnet import UNetAc
from acimg.vision import ResNet50Model

trace, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32
FLAGS.model = "UNet"
sess = Session(torch.device("cpu"))
tr = Trainer(UNetAc(input_shape=[36, 48, 12]), ResNet50Model(input_shape=[224, 298, 3]), session=sess)
g = tr._build_functions(batch_size=B)
calls = []
for name, fn, a in g.plan_train.calls:
    if name in ("conv2d_fwd", "conv2d_fwd_split3", "conv2d_fwd_split3p", "conv2d_dgrad", "deconv_fwd", "deconv_dgrad"):
        calls.append((name, a[0]._obj))
rows = list(csv.DictReader(open(trace)))
ig = [r for r in rows if "igemm_f32_kernel" in r["Kernel_Name"] or "igemm_split3" in r["Kernel_Name"]]
last = ig[-len(calls):]
print("%-4s %-13s %-30s %-22s %8s %8s %8s" % ("#", "op", "HxW C->K RxS/s", "kernel", "us", "TFLOP/s", "GB/s"))
tot = {}
for i, ((name, d), r) in enumerate(zip(calls, last)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fl = 2.0 * d.N * d.OH * d.OW * d.K * d.R * d.S * d.C
    if name.startswith("deconv"):
        fl = 2.0 * d.N * d.H * d.W * d.K * d.R * d.S * d.C
    by = 4.0 * (d.N * d.H * d.W * d.C + d.N * d.OH * d.OW * d.K + d.R * d.S * d.C * d.K)
    kn = r["Kernel_Name"].split("<")[1].split(">")[0].replace(" ", "")
    tot[name] = tot.get(name, 0) + us
    print("%-4d %-13s %-30s %-22s %8.1f %8.1f %8.0f" % (i, name, "%dx%d %d->%d %dx%d/%d" % (d.H, d.W, d.C, d.K, d.R, d.S, d.stride),
          kn, us, fl / us / 1e6, by / us / 1e3))
print("totals (us):", tot)
