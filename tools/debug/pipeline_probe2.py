"""Would a THREE-stage pipeline pay: trunk first half (batch t+2) | trunk second half (batch t+1) | trained part (batch t)?
Timing experiment only (data dependencies between the stages ignored): the recorded train plan is cut at a unit
boundary of the trunk and after the trunk; the parts are replayed one after the other and side by side."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
lo, cut = g.head_calls, g.head_calls + g.frozen_calls
names = [n for n, _, _ in full.calls]
# unit boundaries inside the trunk = the bn_add_relu_split calls
ends = [i + 1 for i in range(lo, cut) if names[i] == "bn_add_relu_split"]
print("trunk calls", lo, cut, "unit ends", ends)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
pB = full.slice(cut, len(full.calls))
sA2, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for k in (3, 5, 7, 9, 13):      # cut after unit k (block1 = 3 units, block2 = 4, block3 = 6, block4 = 3)
    mid = ends[k - 1]
    pA1, pA2 = full.slice(lo, mid), full.slice(mid, cut)
    t1, t2 = timed(lambda: pA1.run()), timed(lambda: pA2.run())
    def two():
        pA1.run()
        with torch.cuda.stream(sA2): pA2.run()
    def three():
        pA1.run()
        with torch.cuda.stream(sA2): pA2.run()
        with torch.cuda.stream(sB): pB.run()
    def twoB():
        pA1.run(); pA2.run()
        with torch.cuda.stream(sB): pB.run()
    print("cut after unit %2d: A1 %.2f + A2 %.2f = %.2f ms; A1|A2 %.2f ms; (A1 A2)|B %.2f ms; A1|A2|B %.2f ms" %
          (k, t1, t2, t1 + t2, timed(two), timed(twoB), timed(three)))
