"""Per-launch report for the forward-conv kernel: aligns a rocprofv3 kernel trace of `bench.py` with the
recorded plan (rebuilt on the CPU, no GPU needed) and prints time, TFLOP/s and GB/s per conv launch.
usage: python tools/layer_report.py <kernel_trace.csv> [batch]"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg import ops
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
convs = []
for name, fn, a in g.plan_train.calls:
    if name == "conv2d_fwd":
        d = a[0]._obj
        bm, bn, sp = ops.conv2d_fwd_tiling(d)
        convs.append((d, bm, bn, sp))
rows = list(csv.DictReader(open(trace)))
kname = [r for r in rows if "igemm_f32_kernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"]]
# the last full step: take the trailing len(convs) forward launches (backward uses NT kernels / deconv NN uses few)
nn_per_step = len([c for c in convs])
# deconv_dgrad also launches an NN kernel once per step (after the forward convs): drop it by name order
per_step = nn_per_step + 1
last = kname[-per_step:]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last][:nn_per_step]
print("%-4s %-28s %-9s %8s %8s %8s %7s" % ("#", "conv (HxW C->K RxS/s)", "tile", "us", "TFLOP/s", "GB/s", "kiters"))
tot = 0
for i, ((d, bm, bn, sp), us) in enumerate(zip(convs, durs)):
    fl = 2.0 * d.N * d.OH * d.OW * d.K * d.R * d.S * d.C
    by = 4.0 * (d.N * d.H * d.W * d.C + d.N * d.OH * d.OW * d.K + d.R * d.S * d.C * d.K)
    tot += us
    print("%-4d %-28s %-9s %8.1f %8.1f %8.0f" % (i, "%dx%d %d->%d %dx%d/%d" % (d.H, d.W, d.C, d.K, d.R, d.S, d.stride),
          "%dx%d/%d" % (bm, bn, sp), us, fl / us / 1e6, by / us / 1e3))
print("total us", tot)
