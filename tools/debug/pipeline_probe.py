"""How much would a two-lane software pipeline buy: lane A = the frozen trunk forward of batch t+1, lane B = conv_map +
generator forward / backward + Adam of batch t.  Timing experiment only (lane B reads whatever lane A left in the
buffers): the recorded train plan is cut after the trunk's calls and the two halves are replayed one after the other on
one stream, and side by side on two streams."""
import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg.flags import FLAGS
from acimg.session import Session
from acimg.trainer import Trainer
from acimg.unet_acresnet import UNetAc
from acimg.vision import ResNet50Model
from acimg import ops
import bench as B

dev = torch.device("cuda:0")
FLAGS.model, FLAGS.ae, FLAGS.num_skip_conn = "UNet", 0, 1
tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1),
             ResNet50Model(input_shape=[224, 298, 3], num_classes=None), learning_rate=1e-4, session=Session(dev))
g = tr._build_functions(batch_size=32)
tr.modelimages.initialize(seed=1238); tr.modelac.initialize(seed=1239)
B.fill_inputs(g, 32, 4321)
for _ in range(3): tr.train_step(sync=False)
torch.cuda.synchronize()
full = g.plan_train
names = [n for n, _, _ in full.calls]
# cut right before the conv_map layer (first tapconv_pack), i.e. after the last frozen layer
cut = names.index("tapconv_pack") if "tapconv_pack" in names else names.index("conv2d_split3_prepare_multi")
print("calls", len(full.calls), "cut at", cut, names[cut-2:cut+3])
def sub(lo, hi):
    p = ops.Plan(dev, ws=full.ws)
    p.calls = full.calls[lo:hi]
    p.side = set(i - lo for i in full.side if lo <= i < hi)
    return p
pA, pB = sub(0, cut), sub(cut, len(full.calls))
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
tA = timed(lambda: pA.run()); tB = timed(lambda: pB.run())
tSeq = timed(lambda: (pA.run(), pB.run()))
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def both():
    with torch.cuda.stream(sA): pA.run()
    with torch.cuda.stream(sB): pB.run()
tPar = timed(both)
print("lane A (trunk) %.2f ms, lane B (conv_map + generator + backward) %.2f ms, one stream %.2f ms, two streams %.2f ms" % (tA, tB, tSeq, tPar))
