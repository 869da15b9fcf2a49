"""debug: train-step parity case (2 skips, f16x3): per step, the per-sample top-2 of the conv_map feature in the HIP run
and in the oracle, and the worst conv_map gradients"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch
from tests.test_trainstep_gpu import build, saved_activations, rel_err
from oracle import trainer as otr
dev = torch.device("cuda:0")
tr, orc, sess = build(dev, 2, False, 2, 1e-3, "f16x3")
store = sess.store
ac, mf, vid, eps = otr.synthetic_batch(2, seed=99)
for step in range(3):
    store.load_state(orc.state_dict(), strict=True)
    store.load_slots(orc.m, orc.v)
    tr.global_step = orc.step
    got = tr.train_step((ac, mf, vid), eps=eps)
    g = tr.primary
    acts = saved_activations(g)
    grads = store.grad_dict()
    masks = dict((k, v > 0) for k, v in acts.items())
    ep = {}
    ref = orc.train_step(ac, mf, vid, eps, end_points=ep, keep_grads=True, relu_masks=masks)
    a, b = acts["conv_map"].double(), ep["resnet_v1_50/conv_map"].detach().double()
    print("step", step, "feature rel err %.2e" % rel_err(a, b))
    for n in range(2):
        ta = a[n].flatten().topk(3); tb = b[n].flatten().topk(3)
        print("  sample", n, "hip top3", [("%.7f" % v, int(i)) for v, i in zip(ta.values, ta.indices)],
              "oracle top3", [("%.7f" % v, int(i)) for v, i in zip(tb.values, tb.indices)])
    for k in ("resnet_v1_50/conv_map/BatchNorm/beta", "resnet_v1_50/conv_map/BatchNorm/gamma", "resnet_v1_50/conv_map/weights"):
        print("  grad", k, "%.3e" % rel_err(grads[k], ref["grads"][k]), "ref max %.3e" % float(ref["grads"][k].abs().max()))
    if step == 2:
        print("  beta hip", grads["resnet_v1_50/conv_map/BatchNorm/beta"].flatten().tolist())
        print("  beta ref", ref["grads"]["resnet_v1_50/conv_map/BatchNorm/beta"].flatten().tolist())
