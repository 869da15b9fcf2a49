"""Per-call report of the recorded train step, measured in place: every C-ABI call of the plan is bracketed with
events (Plan.run_probed) over a few steps.  Run on the GPU box.  usage: python tools/op_report.py [batch] [first_call] [trainer_mask|unet_rgb|unet_sound] [precision of the U-Net: split|bf16|f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg.flags import FLAGS
from acimg.session import Session
from acimg.trainer import Trainer
from acimg.unet_acresnet import UNetAc
from acimg.vision import ResNet50Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
workload = sys.argv[3] if len(sys.argv) > 3 else "trainer_mask"
dev = torch.device("cuda:0")
sess = Session(dev)
gen = torch.Generator().manual_seed(1234)
if workload == "trainer_mask":
    FLAGS.model, FLAGS.ae, FLAGS.num_skip_conn = "UNet", 0, 1
    # per-op attribution: one stream, one trunk stage (overlapped ops would be charged each other's time)
    tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1, side_lane=False),
                 ResNet50Model(input_shape=[224, 298, 3], num_classes=None, side_lane=False), learning_rate=1e-4, session=sess)
    g = tr._build_functions(batch_size=B)
    tr.modelimages.initialize(seed=1238)
    tr.modelac.initialize(seed=1239)
    g.video.copy_(torch.rand(B, 224, 298, 3, generator=gen))
    g.mfcc.copy_(torch.rand(B, 12, generator=gen))
    g.acoustic.copy_(torch.rand(B, 36, 48, 12, generator=gen))
else:
    from acimg.trainer_vae import TrainerVAE
    from acimg.unet_vae import UNet, UNetSound
    prec = sys.argv[4] if len(sys.argv) > 4 else "split"
    tr = TrainerVAE((UNet if workload == "unet_rgb" else UNetSound)(precision=prec), learning_rate=1e-4, session=sess)
    g = tr._build_functions(batch_size=B)
    tr.model.initialize(seed=1240)
    g.images.copy_(torch.rand(*g.images.shape, generator=gen))
for _ in range(3):
    tr.train_step(sync=False)
torch.cuda.synchronize()
plan = g.plan_train
n = len(plan.calls)
acc = [0.0] * n
R = 5
for _ in range(R):
    out = []
    plan.run_probed(set(range(n)), out)
    torch.cuda.synchronize()
    for i, e0, e1 in out:
        acc[i] += e0.elapsed_time(e1) * 1e3 / R
tot = {}
for i, (name, fn, a) in enumerate(plan.calls):
    if a is None:
        continue
    shape = ""
    d = getattr(a[0], "_obj", None)
    if d is not None and hasattr(d, "OH"):
        shape = "%dx%d %d->%d %dx%d/%d" % (d.H, d.W, d.C, d.K, d.R, d.S, d.stride)
    elif name.startswith("bn_") or name in ("maxpool_fwd", "grad_slice"):
        shape = " ".join(str(x) for x in a if isinstance(x, int))[:28]
    tot[name] = tot.get(name, [0, 0.0]); tot[name][0] += 1; tot[name][1] += acc[i]
    if i >= first:
        print("%4d %-24s %-28s %8.1f" % (i, name, shape, acc[i]))
print("sum %.1f us" % sum(acc))
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-26s %4d %9.1f us" % (k, c, us))
