"""debug helper: per-variable error after N optimiser steps (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_trainstep_gpu import build, rel_err, l2_err, saved_activations
from oracle import trainer as otr

dev = torch.device("cuda:0")
tr, orc, sess = build(dev, 1, False, 2)
ac, mf, vid, eps = otr.synthetic_batch(2, seed=99)
for step in range(2):
    got = tr.train_step((ac, mf, vid), eps=eps)
    g = tr.primary
    acts = saved_activations(g)
    masks = dict((k, v > 0) for k, v in acts.items())
    ep = {}
    ref = orc.train_step(ac, mf, vid, eps, end_points=ep, keep_grads=True, relu_masks=masks)
    print("step", step, {k: (got[k], ref[k]) for k in ("mse", "latent", "loss")})
    print("  mean err %.3e  output err %.3e feat err %.3e features err %.3e" % (rel_err(g.modelac.mean, ref["mean"]),
          rel_err(g.modelac.output, ref["output"]), rel_err(acts["conv_map"], ep["resnet_v1_50/conv_map"]),
          rel_err(g.modelac.network["features"], ep["features"])))
    grads = sess.store.grad_dict()
    sd = sess.store.state_dict()
    m = sess.store.slot_dict("m"); v = sess.store.slot_dict("v")
    osd = orc.state_dict()
    rows = []
    for k in orc.train_names:
        rows.append((rel_err(sd[k], osd[k]), k, rel_err(grads[k], ref["grads"][k]), rel_err(m[k], orc.m[k]), rel_err(v[k], orc.v[k]),
                     float((sd[k].double() - osd[k].detach().double()).abs().max())))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("  var err %.3e %-40s grad err %.3e m err %.3e v err %.3e  abs %.3e" % r)
    bn = [(rel_err(sd[k], osd[k]), k) for k in osd if "moving" in k]
    bn.sort(reverse=True)
    print("  worst BN moving stat:", bn[:3])
