"""End-to-end GPU parity of the HIP train step against the CPU oracle on identical weights, inputs and
noise: loss terms, generated image, mean/std, EVERY gradient, post-Adam weights after 1 and 3 steps,
batch-norm moving statistics, and the evaluation pass.

Tolerance (BASELINE.json north_star): outputs and losses within 1e-3 relative (max-norm, relative to
the tensor's max magnitude).  Both sides compute in fp32 (the HIP path on exact-f32 MFMA), observed
error ~1e-5.  Gradients are held to the same 1e-3 (max-norm and L2) on an input whose ReLU masks agree
exactly with the oracle's: a pre-activation within fp32 rounding of zero (observed with seed 99: one
element of 49k in `conv4`, 2.3e-7 vs 0.0) legitimately takes the other subgradient, so the test counts
mask flips, bounds their damage loosely, and demands strict parity on a flip-free seed.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3


def rel_err(got, ref):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-12))


def l2_err(got, ref):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


def build(device, num_skip, embedding, batch, lr=1e-3):
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer import Trainer
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model
    from oracle import trainer as otr

    FLAGS.model = "UNet"
    FLAGS.ae = int(embedding)
    FLAGS.latent_loss = 1e-3  # larger than the default 1e-6 so the KL path is visible in the gradients
    orc = otr.Oracle(num_skip=num_skip, embedding=embedding, learning_rate=lr, latent_loss=FLAGS.latent_loss,
                     randomize=True)
    sess = Session(device)
    mi = ResNet50Model(input_shape=[224, 298, 3], num_classes=None)
    ma = UNetAc(input_shape=[36, 48, 12], embedding=embedding, num_skip=num_skip)
    tr = Trainer(ma, mi, learning_rate=lr, session=sess)
    tr._build_functions(batch_size=batch)
    loaded = sess.store.load_state(orc.state_dict(), strict=True)
    assert len(loaded) == len(orc.state_dict())
    return tr, orc, sess


MASK_PAIRS = [("c11", "layer1/conv_1"), ("conv1", "conv1"), ("pool1", "pool1"), ("c21", "layer2/conv_1"),
              ("conv2_0", "conv2_0"), ("dns", "dense"), ("net", "conv2d"), ("c41", "layer4/conv_1"), ("conv4", "conv4"),
              ("c51", "layer5/conv_1"), ("conv5", "conv5"), ("c61", "layer6/conv_1"), ("conv6", "conv6"),
              ("c71", "layer7/conv_1"), ("conv7", "conv7")]


def relu_mask_flips(ma, ep):
    """number of post-ReLU activations that are zero on one side and positive on the other (a
    pre-activation within fp32 rounding of zero); every activation is also checked to 1e-3"""
    flips = 0
    for attr, key in MASK_PAIRS:
        a = getattr(ma, attr)
        t = a.t.cpu().reshape(a.N, a.H, a.W, -1)[..., a.off:a.off + a.C]
        r = ep[key].detach()
        assert rel_err(t, r) < TOL, "activation " + key
        flips += int(((t > 0) != (r > 0)).sum())
    return flips


def run_case(device, num_skip, embedding, seed):
    """returns True when the case ran with zero ReLU-mask flips and passed the strict gradient checks"""
    from oracle import trainer as otr

    B = 2
    tr, orc, sess = build(device, num_skip, embedding, B)
    ac, mf, vid, eps = otr.synthetic_batch(B, seed=seed)
    # state-dict round trip through the padded internal layouts is lossless
    sd = sess.store.state_dict()
    for k, v in orc.state_dict().items():
        assert torch.equal(sd[k], v.detach()), k
    strict = True
    for step in range(3):
        ep = {}
        ref = orc.train_step(ac, mf, vid, eps, end_points=ep, keep_grads=True)
        got = tr.train_step((ac, mf, vid), eps=eps)
        for k in ("mse", "huber", "latent", "reg", "loss"):
            assert abs(got[k] - ref[k]) <= TOL * max(abs(ref[k]), 1e-8), (step, k, got[k], ref[k])
        g = tr.primary
        assert rel_err(g.modelimages.output, ep["resnet_v1_50/conv_map"]) < TOL, "resnet feature"
        assert rel_err(g.modelac.output, ref["output"]) < TOL, "generated image"
        assert rel_err(g.modelac.mean, ref["mean"]) < TOL, "mean"
        if not embedding:
            assert rel_err(g.modelac.std, ref["std"]) < TOL, "std"
        if step == 0:
            assert rel_err(g.modelac.network["features"], ep["features"]) < TOL, "145-ch feature map"
            flips = relu_mask_flips(g.modelac, ep)
            strict = flips == 0
            grads = sess.store.grad_dict()
            worst, worst_l2 = ("", 0.0), ("", 0.0)
            for k, gr in ref["grads"].items():
                e = rel_err(grads[k], gr)
                if e > worst[1]:
                    worst = (k, e)
                e = l2_err(grads[k], gr)
                if e > worst_l2[1]:
                    worst_l2 = (k, e)
            if strict:
                assert worst[1] < TOL, "gradient %s max-norm err %.3e" % worst
                assert worst_l2[1] < TOL, "gradient %s L2 err %.3e" % worst_l2
            else:  # one subgradient differs: the damage must stay small and local
                assert worst_l2[1] < 2e-2, "gradient %s L2 err %.3e with %d mask flips" % (worst_l2 + (flips,))
                return False
    # weights, Adam slots and BN moving statistics after 3 steps
    sd = sess.store.state_dict()
    worst = ("", 0.0)
    for k, v in orc.state_dict().items():
        e = rel_err(sd[k], v)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < TOL, "variable %s rel err %.3e after 3 steps" % worst
    m = sess.store.slot_dict("m")
    for k in orc.train_names:
        assert l2_err(m[k], orc.m[k]) < 2e-3, "adam m " + k
    # evaluation pass (BN inference mode)
    refe = orc.eval_step(ac, mf, vid, eps)
    gote = tr.eval_step((ac, mf, vid), eps=eps)
    for k in ("mse", "mse0", "mse1", "mse2", "mse3"):
        assert abs(gote[k] - refe[k]) <= TOL * refe[k], (k, gote[k], refe[k])
    assert rel_err(tr.primary.modelac.output, refe["output"]) < TOL
    return True


@pytest.mark.parametrize("num_skip,embedding", [(1, False), (2, False), (0, True)])
def test_train_step_matches_oracle(device, num_skip, embedding):
    """strict parity must hold on an input whose ReLU masks agree exactly with the oracle's (tries up to
    three seeds; a seed with a flipped mask still has to stay within the loose bound)"""
    for seed in (99, 100, 101):
        if run_case(device, num_skip, embedding, seed):
            return
    pytest.fail("no seed without ReLU-mask flips among 3")


def test_partial_batch_and_device_noise(device):
    """a smaller last batch gets its own plans over the same variables; eps=None samples on device"""
    tr, orc, sess = build(device, 1, False, 2)
    from oracle import trainer as otr

    ac, mf, vid, eps = otr.synthetic_batch(1, seed=5)
    ref = orc.eval_step(ac, mf, vid, eps)
    got = tr.eval_step((ac, mf, vid), eps=eps)
    assert abs(got["mse"] - ref["mse"]) <= TOL * ref["mse"]
    r1 = tr.train_step((ac, mf, vid))
    assert np.isfinite(r1["loss"])
    e = tr.graphs[1].eps.cpu()
    assert abs(float(e.mean())) < 0.5 and 0.5 < float(e.std()) < 1.5
