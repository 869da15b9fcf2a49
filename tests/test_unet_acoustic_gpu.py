"""The acoustic-image VAE `UNetAc` (scope 'UNetAcoustic'; SURVEY A.3 last row, §8 a9): models/unet_noconc.py as a
stand-alone VAE train step (trainer/trainer.py, encoder_type 'Ac') and models/unet_z.py as the decoder driven by
external (mean2, std2) — outputs, losses, every gradient (incl. d loss / d (mean2, std2)) vs the fp64 oracle."""
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


def _masks(m):
    acts = {"layer1/conv_1": m.c11, "layer1/conv_2": m.conv1, "layer1/pool_2": m.pool1, "layer3/conv_1": m.c31,
            "layer3/conv_2": m.conv2, "conv2d": m.net, "layer4/conv_1": m.c41, "layer4/conv_2": m.conv4,
            "layer5/conv_1": m.c51, "layer5/conv_2": m.conv5}
    out = {k: (a.t[..., :a.C] > 0).cpu() for k, a in acts.items()}
    out["dense"] = (m.dns.t > 0).cpu()
    return out


@pytest.mark.parametrize("precision", ["split", "f32"])
def test_unet_noconc_vae_step(precision):
    from acimg.session import Session
    from acimg.trainer_vae import TrainerVAE
    from acimg.unet_acoustic import UNetAcNoConc
    from oracle import unet_acoustic as oa

    dev = torch.device("cuda:0")
    N = 10            # 17280 pixels: above the split-kernel threshold
    sess = Session(dev)
    tr = TrainerVAE(UNetAcNoConc(precision=precision), learning_rate=1e-3, session=sess)
    tr._build_functions(batch_size=N)
    params = oa.init_params(seed=8, dtype=torch.float64, bias_std=0.05)
    tr.model.initialize(state={k: v.float() for k, v in params.items()})
    g = torch.Generator().manual_seed(12)
    x = torch.rand(N, 36, 48, 12, generator=g, dtype=torch.float64)
    eps = torch.randn(N, 150, generator=g, dtype=torch.float64)
    r = tr.train_step(x.float().to(dev), eps.float().to(dev), apply=False)
    torch.cuda.synchronize()
    m = tr.model
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    fw = oa.forward(p, x, eps, relu_masks=_masks(m))
    ls = oa.vae_losses(x, fw)
    grads = dict(zip(p.keys(), torch.autograd.grad(ls["loss"], list(p.values()))))
    assert rel(m.output, fw["output"].detach()) < 1e-4 and rel(m.mean, fw["mean"].detach()) < 1e-4
    assert rel(m.std, fw["std"].detach()) < 1e-4
    for k in ("mse", "huber", "latent", "loss"):
        assert abs(r[k] - float(ls[k])) <= 1e-4 * abs(float(ls[k])) + 1e-10, (k, r[k], float(ls[k]))
    got = sess.store.grad_dict()
    worst = max((rel(got[k], v), k) for k, v in grads.items())
    print("unet_noconc %s: worst gradient %s %.2e" % (precision, worst[1], worst[0]))
    assert worst[0] < 1e-3, worst


def test_unet_z_external_latent():
    from acimg import ops
    from acimg.session import Session
    from acimg.unet_acoustic import UNetAcZ
    from oracle import unet_acoustic as oa

    dev = torch.device("cuda:0")
    N = 4
    sess = Session(dev)
    m = UNetAcZ()
    g = torch.Generator().manual_seed(14)
    x = torch.rand(N, 36, 48, 12, generator=g, dtype=torch.float64)
    eps = torch.randn(N, 150, generator=g, dtype=torch.float64)
    mean2 = torch.randn(N, 150, generator=g, dtype=torch.float64) * 0.5
    std2 = torch.rand(N, 150, generator=g, dtype=torch.float64) + 0.2
    ext = torch.cat([mean2, std2], 1).float().to(dev)
    xd, epsd = x.float().to(dev), eps.float().to(dev)
    m._build_model(xd, ext[:, :150], ext[:, 150:], session=sess, eps=epsd)
    sums, g_logit = sess.zeros(4), sess.zeros(N, 36, 48, 12)
    count = N * 36 * 48 * 12
    klw = 1e-6 / N                     # latent_loss * mean_b(0.5 * sum_j ...), as the associator trainers weight it
    plan = sess.new_plan()
    plan.extend(m.plan_fwd)
    ops.recon_loss(plan, m.yhat.t, xd, g_logit, sums, count, 1.0, 1.0)
    m.record_backward(plan, g_logit, klw)
    sess.finalize()
    params = oa.init_params(seed=9, dtype=torch.float64, bias_std=0.05)
    m.initialize(state={k: v.float() for k, v in params.items()})
    plan.run()
    torch.cuda.synchronize()
    masks = _masks(m)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    m2, s2 = mean2.clone().requires_grad_(True), std2.clone().requires_grad_(True)
    fw = oa.forward(p, x, eps, m2, s2, relu_masks=masks)
    from oracle import tfsem
    kl = 0.5 * (m2 * m2 + s2 * s2 - torch.log(1e-8 + s2 * s2) - 1).sum(1)
    loss = tfsem.mse_loss(x, fw["output"]) + tfsem.huber_loss(x, fw["output"]) + 1e-6 * kl.mean(0)
    dec = [k for k in p if any(t in k for t in ("dense", "conv2d", "upsample", "layer4", "layer5", "final"))]
    gr = torch.autograd.grad(loss, [m2, s2] + [p[k] for k in dec])
    assert rel(m.output, fw["output"].detach()) < 1e-4
    assert rel(m.mean, fw["mean"].detach()) < 1e-4 and rel(m.std, fw["std"].detach()) < 1e-4   # own statistics
    assert rel(m.g_ext[:, :150], gr[0]) < 1e-3 and rel(m.g_ext[:, 150:], gr[1]) < 1e-3
    got = sess.store.grad_dict()
    for k, gv in zip(dec, gr[2:]):
        assert rel(got[k], gv) < 1e-3, k
