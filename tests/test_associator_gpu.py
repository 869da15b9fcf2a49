"""Latent associators and joint-latent fusion MLPs (models/multimodal.py) and the
single-associator step of trainer/trainer_proietta.py against the frozen unet_z decoder: outputs, losses and every
gradient vs the fp64 oracle, through the C ABI; only the associator's variables move."""
import os
import sys
from collections import OrderedDict

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "acoustic-image-generation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("which", ["AssociatorVideoAc", "AssociatorAudioAc"])
def test_associator_step(which):
    from acimg import multimodal
    from acimg.session import Session
    from acimg.trainer_associator import TrainerAssociator
    from acimg.unet_acoustic import UNetAcZ
    from oracle import multimodal as om
    from oracle import unet_acoustic as oa
    from tests.test_unet_acoustic_gpu import _masks

    dev = torch.device("cuda:0")
    N = 6
    cls = getattr(multimodal, which)
    sess = Session(dev)
    tr = TrainerAssociator(cls(), UNetAcZ(), learning_rate=1e-3, session=sess)
    g = tr._build_functions(batch_size=N)
    pa = om.init_params(which, seed=21, dtype=torch.float64, bias_std=0.05)
    pd = oa.init_params(seed=22, dtype=torch.float64, bias_std=0.05)
    tr.modelassociator.initialize(state={k: v.float() for k, v in pa.items()})
    tr.modelac.initialize(state={k: v.float() for k, v in pd.items()})
    gen = torch.Generator().manual_seed(23)
    din = cls.DIN
    stats = torch.randn(N, 2 * din, generator=gen, dtype=torch.float64)
    x = torch.rand(N, 36, 48, 12, generator=gen, dtype=torch.float64)
    eps = torch.randn(N, 150, generator=gen, dtype=torch.float64)
    before = {k: v.clone() for k, v in sess.store.state_dict().items()}
    r = tr.train_step((stats.float().to(dev), x.float().to(dev)), eps.float().to(dev), apply=False)
    torch.cuda.synchronize()
    ma, md = tr.modelassociator, tr.modelac
    masks_a = {}
    for name, d, xx, off, ldx, y, ldy, tower, last in ma.layers:
        if not last:
            masks_a[name] = (y[:, :d.K] > 0).cpu()
    p = {k: v.clone().requires_grad_(True) for k, v in pa.items()}
    ref = om.step_loss(p, which, pd, x, eps, stats[:, :din], stats[:, din:], masks_a, _masks(md))
    assert rel(ma.mean, ref["mean"].detach()) < 1e-4 and rel(ma.std, ref["std"].detach()) < 1e-4
    assert rel(md.output, ref["output"].detach()) < 1e-4
    for k in ("mse", "huber", "latent", "loss"):
        assert abs(r[k] - float(ref[k])) <= 1e-4 * abs(float(ref[k])) + 1e-12, (k, r[k], float(ref[k]))
    grads = dict(zip(p.keys(), torch.autograd.grad(ref["loss"], list(p.values()))))
    got = sess.store.grad_dict()
    worst = max((rel(got[k], v), k) for k, v in grads.items())
    print("%s: worst gradient %s %.2e" % (which, worst[1], worst[0]))
    assert worst[0] < 1e-3, worst
    # a few optimisation steps: the loss falls, and only the associator's variables move
    first = tr.train_step(None, eps.float().to(dev))
    for _ in range(20):
        last = tr.train_step(None, eps.float().to(dev))
    assert last["loss"] < first["loss"]
    after = sess.store.state_dict()
    for k in before:
        assert (not torch.equal(before[k], after[k])) == k.startswith(which + "/"), k


@pytest.mark.parametrize("which,widths", [("Jointmvae", (128, 512, 128)), ("JointTwomvae", (512, 128)),
                                          ("JointTwomvae2", (512, 128))])
def test_joint_mlp(which, widths):
    """joint-latent fusion MLPs (models/multimodal.py:287-465) on 12x16 feature maps: head outputs, every parameter
    gradient and the gradient w.r.t. the concatenated inputs vs the fp64 oracle, under a weighted squared-error
    loss on each head"""
    from acimg import multimodal
    from acimg.session import Session
    from oracle import multimodal as om

    dev = torch.device("cuda:0")
    N, H, W = 3, 12, 16
    cls = getattr(multimodal, which)
    sess = Session(dev)
    ctot = sum(widths)
    buf = sess.zeros(N, H, W, ctot)
    ins, o = [], 0
    for w in widths:
        ins.append(buf[..., o:o + w])
        o += w
    m = cls()
    m._build_model(*ins, session=sess)
    g_bufs = OrderedDict((a, sess.zeros(m.rows, m.heads[a].shape[1])) for a, _ in cls.HEADS)
    bp = sess.new_plan()
    m.record_backward(bp, {a: (g, g.shape[1]) for a, g in g_bufs.items()})
    sess.finalize()
    pa = om.joint_init_params(which, ctot, seed=31, dtype=torch.float64, bias_std=0.05)
    m.initialize(state={k: v.float() for k, v in pa.items()})
    gen = torch.Generator().manual_seed(32)
    x = torch.rand(N, H, W, ctot, generator=gen, dtype=torch.float64).float().double()
    buf.copy_(x.float())
    m.plan_fwd.run()
    torch.cuda.synchronize()
    masks = {}
    for name, d, xx, ldx, y, attr in m.layers:
        masks[name] = (y[:, :d.K] > 0).cpu().view(N, H, W, d.K)
    p = {k: v.clone().requires_grad_(True) for k, v in pa.items()}
    xin = x.clone().requires_grad_(True)
    parts, o = [], 0
    for w in widths:
        parts.append(xin[..., o:o + w])
        o += w
    outs, _ = om.joint_forward(p, which, parts, relu_masks=masks)
    loss = 0
    for k, (attr, w) in enumerate(cls.HEADS):
        got = getattr(m, attr)
        assert got.shape == (N, H, W, w) and rel(got, outs[attr].detach()) < 1e-4
        tgt = torch.rand(N, H, W, w, generator=gen, dtype=torch.float64)
        wk = 0.5 + k
        loss = loss + 0.5 * wk * ((outs[attr] - tgt) ** 2).sum()
        g_bufs[attr][:, :w] = (wk * (got.double().cpu() - tgt)).float().view(-1, w).to(dev)
    bp.run()
    torch.cuda.synchronize()
    names = list(p.keys())
    grads = torch.autograd.grad(loss, [p[k] for k in names] + [xin])
    got = sess.store.grad_dict()
    worst = max((rel(got[k], v), k) for k, v in zip(names, grads[:-1]))
    print("%s: worst gradient %s %.2e" % (which, worst[1], worst[0]))
    assert worst[0] < 1e-3, worst
    assert rel(m.g_input.view(N, H, W, ctot), grads[-1]) < 1e-3


def test_conv_associator_audio_step():
    """`AssociatorAudio` (models/multimodal.py:139-285: the conv associator - conv_conv_pool x 5 on a 193x257x1
    spectrogram, batch norm in training mode, two 12x16 VALID heads, std = softplus) in front of the frozen `unet_z`
    decoder, the single-associator step of trainer/trainer_proietta.py:104-146 with the kernel regularisers that
    tf.losses.get_total_loss() collects: mean / std / generated images, loss terms, every gradient of the associator
    (kernels, biases, BN gamma / beta, heads), BN moving statistics, vs the fp64 oracle; a few steps lower the loss and
    move only `AssociatorAudio/` variables."""
    from acimg import multimodal
    from acimg.session import Session
    from acimg.trainer_associator import TrainerAssociator
    from acimg.unet_acoustic import UNetAcZ
    from oracle import multimodal as om
    from oracle import unet_acoustic as oa
    from oracle import unet_vae as ouv
    from tests.test_unet_acoustic_gpu import _masks

    dev = torch.device("cuda:0")
    N = 4
    sess = Session(dev)
    tr = TrainerAssociator(multimodal.AssociatorAudio(), UNetAcZ(), learning_rate=1e-3, session=sess)
    g = tr._build_functions(batch_size=N)
    pa = ouv.init_params("AssociatorAudio", seed=31, dtype=torch.float64, bias_std=0.05, bn_jitter=0.1)
    pd = oa.init_params(seed=32, dtype=torch.float64, bias_std=0.05)
    tr.modelassociator.initialize(state={k: v.float() for k, v in pa.items()})
    tr.modelac.initialize(state={k: v.float() for k, v in pd.items()})
    gen = torch.Generator().manual_seed(33)
    spec = torch.rand(N, 193, 257, 1, generator=gen, dtype=torch.float64)
    x = torch.rand(N, 36, 48, 12, generator=gen, dtype=torch.float64)
    eps = torch.randn(N, 150, generator=gen, dtype=torch.float64)
    before = {k: v.clone() for k, v in sess.store.state_dict().items()}
    r = tr.train_step((spec.float().to(dev), x.float().to(dev)), eps.float().to(dev), apply=False)
    torch.cuda.synchronize()
    ma, md = tr.modelassociator, tr.modelac
    masks_a = {}
    for name, L in ma.layers.items():
        masks_a[name] = (L.relu_output() > 0).cpu()
    free = om.step_loss_audio(pa, pd, spec, x, eps)
    flips = sum(int((free["masks_a"][k] != masks_a[k].reshape(free["masks_a"][k].shape)).sum()) for k in masks_a)
    total = sum(v.numel() for v in masks_a.values())
    print("AssociatorAudio: ReLU pattern differs from the fp64 oracle's in %d of %d places" % (flips, total))
    assert flips < 200
    assert rel(ma.mean, free["mean"]) < 1e-4 and rel(ma.std, free["std"]) < 1e-4       # forward vs the free-running oracle
    p = {k: v.clone().requires_grad_(True) for k, v in pa.items()}
    ref = om.step_loss_audio(p, pd, spec, x, eps, {k: v.reshape(free["masks_a"][k].shape) for k, v in masks_a.items()},
                             _masks(md))
    assert rel(ma.mean, ref["mean"].detach()) < 1e-4 and rel(ma.std, ref["std"].detach()) < 1e-4
    assert rel(md.output, ref["output"].detach()) < 1e-4
    for k in ("mse", "huber", "latent", "reg", "loss"):
        assert abs(r[k] - float(ref[k])) <= 1e-4 * abs(float(ref[k])) + 1e-12, (k, r[k], float(ref[k]))
    names = [k for k in p if ouv.trainable(k)]
    grads = dict(zip(names, torch.autograd.grad(ref["loss"], [p[k] for k in names])))
    got = sess.store.grad_dict()
    worst = ("", 0.0)
    for k, v in grads.items():
        if k.endswith("/bias") and "/layer" in k:
            # a conv bias under a batch norm has an exactly-zero gradient: ~0 relative to its kernel's on both sides
            kmax = float(grads[k[:-4] + "kernel"].abs().max())
            assert float(got[k].abs().max()) < 1e-4 * kmax and float(v.abs().max()) < 1e-6 * kmax, k
            continue
        e = rel(got[k], v)
        worst = max(worst, (k, e), key=lambda t: t[1])
    print("AssociatorAudio: worst gradient %s %.2e" % worst)
    assert worst[1] < 1e-3, worst
    after_bn = sess.store.state_dict()
    for k, v in ref["new_stats"].items():
        assert rel(after_bn[k], v.detach()) < 1e-4, "BN moving statistic " + k
    # a few optimisation steps: the loss falls, and only the associator's variables move
    first = tr.train_step(None, eps.float().to(dev))
    for _ in range(10):
        last = tr.train_step(None, eps.float().to(dev))
    assert last["loss"] < first["loss"]
    after = sess.store.state_dict()
    for k, v in before.items():
        moved = not torch.equal(v, after[k])
        assert moved == k.startswith("AssociatorAudio/"), (k, moved)
