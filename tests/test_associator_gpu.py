"""Latent associators (models/multimodal.py) and the single-associator step of
trainer/trainer_proietta.py against the frozen unet_z decoder: outputs, losses and every associator gradient vs
the fp64 oracle, through the C ABI; only the associator's variables move."""
import os
import sys

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
