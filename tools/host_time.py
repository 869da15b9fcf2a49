"""Host time of one recorded train step (time for train_step(sync=False) to RETURN, GPU drained before each step) vs the
GPU step time.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg.flags import FLAGS
from acimg.session import Session
from acimg.trainer import Trainer
from acimg.unet_acresnet import UNetAc
from acimg.vision import ResNet50Model
dev = torch.device("cuda:0")
FLAGS.model, FLAGS.ae, FLAGS.num_skip_conn = "UNet", 0, 1
sess = Session(dev)
tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1),
             ResNet50Model(input_shape=[224, 298, 3], num_classes=None), learning_rate=1e-4, session=sess)
g = tr._build_functions(batch_size=32)
tr.modelimages.initialize(seed=1238); tr.modelac.initialize(seed=1239)
for _ in range(3):
    tr.train_step(sync=False)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("host ms/step: min %.2f median %.2f; step (drained start) ms: median %.2f; calls %d" %
      (min(host), sorted(host)[5], sorted(total)[5], len(g.plan_train.calls)))
