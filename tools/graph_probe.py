"""does replaying the recorded step as ONE hipGraph beat launching its ~276 kernels one by one?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from acimg.flags import FLAGS
from acimg.session import Session
from acimg.trainer import Trainer
from acimg.unet_acresnet import UNetAc
from acimg.vision import ResNet50Model
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
FLAGS.model, FLAGS.ae, FLAGS.num_skip_conn = "UNet", 0, 1
B = 32
sess = Session(dev)
tr = Trainer(UNetAc(input_shape=[36, 48, 12]), ResNet50Model(input_shape=[224, 298, 3], num_classes=None), session=sess)
g = tr._build_functions(batch_size=B)
tr.modelimages.initialize(seed=1238); tr.modelac.initialize(seed=1239)
gen = torch.Generator().manual_seed(1)
g.video.copy_(torch.rand(B, 224, 298, 3, generator=gen)); g.mfcc.copy_(torch.rand(B, 12, generator=gen)); g.acoustic.copy_(torch.rand(B, 36, 48, 12, generator=gen))
for _ in range(5): tr.train_step(sync=False)
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager plan        : %.3f ms/step" % timeit(lambda: tr.train_step(sync=False)))
loss_eager = tr._scalars(g)["loss"]
s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    g.plan_train.run()          # warm on the side stream
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(graph, stream=s):
    g.plan_train.run()
torch.cuda.synchronize()
def step_graph():
    tr._noise(g, None)
    graph.replay()
print("hipGraph replay   : %.3f ms/step (plan only, no Adam)" % timeit(step_graph))
def step_eager_plan():
    tr._noise(g, None)
    g.plan_train.run()
print("eager plan only   : %.3f ms/step (no Adam)" % timeit(step_eager_plan))
