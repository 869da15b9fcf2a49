"""CPU tests of the oracle: pinned against the reference's own NumPy outputs (tests/golden/, made by
tests/golden/make_frontend_golden.py) and the known answers recorded in SURVEY.md App. D; plus
checks of each TensorFlow semantic the network restatement relies on."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend, resnet50, tfsem, trainer, unet_acresnet

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz"))


def test_mfcc_matches_reference_golden():
    got = frontend.mfcc(GOLD["frames"])
    assert got.dtype == np.float32 and got.shape == (14, 12)
    np.testing.assert_allclose(got, GOLD["mfcc"], rtol=1e-6, atol=1e-6)


def test_mfcc_known_answers_survey_appendix_d():
    got = frontend.mfcc(GOLD["frames"][:12])
    row0 = [-9.510008, -1.000083, -3.269133, 3.592514, 0.124461, -3.20517, -0.679309, 2.78446, -4.455116,
            -3.4477, 0.066911, -5.548341]
    np.testing.assert_allclose(got[0], row0, rtol=0, atol=2e-5)
    assert abs(float(got.sum()) - (-262.86496)) < 1e-3
    assert abs(float(np.abs(got).max()) - 9.510008) < 1e-5


def test_mel_filters_match_reference():
    f = frontend.createfilters()
    np.testing.assert_array_equal(f, GOLD["filters"])
    colsum = [5.5, 6, 7, 7.5, 8, 9.5, 10, 10.5, 12, 13.5, 14.5, 15.5, 17, 19, 21, 23, 25, 27.5, 30, 33, 36.5, 40,
              43.5, 48]
    np.testing.assert_allclose(f.sum(0), colsum, atol=1e-9)


def test_lowpass_and_silence_mfcc_match_reference():
    lp = frontend.butter_lowpass_filter(GOLD["frames"][:12].astype(np.float64))
    np.testing.assert_allclose(lp, GOLD["lowpassed"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(frontend.mfcc(GOLD["lowpassed"]), GOLD["mfcc_lowpassed"], rtol=1e-6, atol=1e-6)


def test_find_logen_matches_reference():
    np.testing.assert_allclose(frontend.find_logen(GOLD["img64"]).reshape(36, 48), GOLD["logen64"], rtol=1e-12)
    np.testing.assert_allclose(frontend.find_logen(GOLD["img32"]).reshape(36, 48), GOLD["logen32"], rtol=1e-6)


def test_normalize_mfcc_range():
    v = frontend.normalize_mfcc(GOLD["mfcc"])
    assert v.dtype == np.float32
    np.testing.assert_array_equal(v.min(1), 0)
    np.testing.assert_array_equal(v.max(1), 1)


# ---- TF semantics ----------------------------------------------------------------------------------
def test_same_padding_is_asymmetric_with_stride():
    assert tfsem.same_pads(224, 3, 2) == (112, 0, 1)
    assert tfsem.same_pads(298, 3, 2) == (149, 0, 1)
    assert tfsem.same_pads(36, 3, 3) == (12, 0, 0)
    assert tfsem.same_pads(112, 3, 2) == (56, 0, 1)   # pool1 rows
    assert tfsem.same_pads(149, 3, 2) == (75, 1, 1)   # pool1 cols
    x = torch.arange(25.0).reshape(1, 5, 5, 1)
    w = torch.ones(3, 3, 1, 1)
    y = tfsem.conv2d(x, w, None, 2, "SAME")
    assert y.shape == (1, 3, 3, 1)
    assert float(y[0, 0, 0, 0]) == 0 + 1 + 5 + 6  # pad 1 before


def test_conv2d_transpose_valid_size_and_bias_only_gaps():
    x = torch.ones(1, 12, 16, 4)
    w = torch.ones(2, 2, 3, 4)
    b = torch.tensor([0.5, 1.5, 2.5])
    y = tfsem.conv2d_transpose_valid(x, w, b, 3)
    assert y.shape == (1, 36, 48, 3)
    np.testing.assert_allclose(y[0, 0, 0].numpy(), [4.5, 5.5, 6.5])
    np.testing.assert_allclose(y[0, 2, 0].numpy(), [0.5, 1.5, 2.5])   # row 2 of each cell: bias only
    np.testing.assert_allclose(y[0, 35, 47].numpy(), [0.5, 1.5, 2.5])


def test_batch_norm_moving_variance_is_unbiased():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 3, 3, 2, generator=g, dtype=torch.float64)
    y, mm, mv, mean, var = tfsem.batch_norm(x, torch.ones(2, dtype=torch.float64), torch.zeros(2, dtype=torch.float64),
                                            torch.zeros(2, dtype=torch.float64), torch.ones(2, dtype=torch.float64), True)
    flat = x.reshape(-1, 2)
    np.testing.assert_allclose(mv.numpy(), 0.997 + 0.003 * flat.var(0, unbiased=True).numpy(), rtol=1e-12)
    np.testing.assert_allclose(y.reshape(-1, 2).var(0, unbiased=False).numpy(), 1.0, rtol=1e-4)


def test_adam_is_tf1_form():
    p, g = torch.tensor([1.0]), torch.tensor([0.5])
    p2, m, v = tfsem.adam_tf1(p, g, torch.zeros(1), torch.zeros(1), 1, 0.1)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = 1.0 - lr_t * 0.05 / (np.sqrt(0.00025) + 1e-8)
    assert abs(float(p2) - exp) < 1e-6


def test_minmax_gradient_splits_ties():
    x = torch.tensor([[0.0, 0.0, 1.0, 2.0, 2.0]], dtype=torch.float64, requires_grad=True)
    o = tfsem.minmax_norm(x, (1,))
    o.backward(torch.tensor([[1.0, 2.0, 3.0, 4.0, 5.0]], dtype=torch.float64))
    s1, s2, D = 15.0, (3 * 0.5 + 4 + 5), 2.0
    gmin, gmax = -(s1 - s2) / D / 2, -s2 / D / 2
    exp = np.array([1 / D + gmin, 2 / D + gmin, 3 / D, 4 / D + gmax, 5 / D + gmax])
    np.testing.assert_allclose(x.grad.numpy()[0], exp, rtol=1e-12)
    # prescribed argmin / argmax sets (parity tests take them from the implementation under test): same result
    x2 = x.detach().clone().requires_grad_(True)
    o2 = tfsem.minmax_norm(x2, (1,), sel=(x2.detach() == 0.0, x2.detach() == 2.0))
    o2.backward(torch.tensor([[1.0, 2.0, 3.0, 4.0, 5.0]], dtype=torch.float64))
    assert torch.equal(o2.detach(), o.detach())
    np.testing.assert_allclose(x2.grad.numpy()[0], exp, rtol=1e-12)


def test_losses_reductions():
    y = torch.tensor([[0.0, 0.5], [2.0, -3.0]])
    t = torch.zeros(2, 2)
    assert abs(float(tfsem.mse_loss(t, y)) - (0.25 + 4 + 9) / 4) < 1e-7
    hub = (0.5 * 0.25 + (2 - 0.5) + (3 - 0.5)) / 4
    assert abs(float(tfsem.huber_loss(t, y)) - hub) < 1e-7


# ---- network restatement ---------------------------------------------------------------------------
def test_parameter_inventory():
    n_gen = sum(int(np.prod(s)) for s in unet_acresnet.param_shapes(1).values())
    assert n_gen == 10558532                      # 10.56 M
    heads = 2 * (12 * 16 * 145 * 150 + 150)
    assert heads == 8352300                       # 8.35 M in the mean/std heads
    shapes = resnet50.param_shapes()
    assert len([k for k in shapes if k.endswith("/weights")]) == 54   # 53 trunk convs + conv_map
    assert shapes["resnet_v1_50/conv_map/weights"] == (3, 4, 2048, 12)
    assert resnet50.train_var_names()[0] == "resnet_v1_50/conv_map/weights"
    assert "UNetAcRes/layer4/conv_1/kernel" in unet_acresnet.param_shapes(2)
    assert unet_acresnet.param_shapes(2)["UNetAcRes/layer4/conv_1/kernel"] == (3, 3, 266, 128)
    assert unet_acresnet.param_shapes(0)["UNetAcRes/layer6/conv_1/kernel"] == (3, 3, 128, 128)


# The trunk geometry WRITTEN OUT from /root/reference/models/resnet50.py:229-250 (`resnet_v1_block`: units 1 .. n-1 have
# stride 1, the LAST unit carries the block's stride) and :261-266 (block1: 64 x 3 stride 1; block2: 128 x 4 stride 2;
# block3: 256 x 6 stride 2; block4: 512 x 3 stride 1), with :104-125 (shortcut = 1x1 conv only where the depth changes,
# else subsample) and :205-209 (7x7/2 root conv + 3x3/2 SAME pool: 224x298 -> 112x149 -> 56x75).  One literal row per
# unit, independent of the generators `oracle.resnet50.units()` / `acimg.vision.ResNet50Model._units()` (which share an
# author): (block, unit, depth_in, depth_bottleneck, depth_out, stride, projection shortcut?, H x W in, H x W out).
TRUNK_UNITS = [
    ("block1", 1,   64,  64,  256, 1, True,  (56, 75), (56, 75)),
    ("block1", 2,  256,  64,  256, 1, False, (56, 75), (56, 75)),
    ("block1", 3,  256,  64,  256, 1, False, (56, 75), (56, 75)),
    ("block2", 1,  256, 128,  512, 1, True,  (56, 75), (56, 75)),
    ("block2", 2,  512, 128,  512, 1, False, (56, 75), (56, 75)),
    ("block2", 3,  512, 128,  512, 1, False, (56, 75), (56, 75)),
    ("block2", 4,  512, 128,  512, 2, False, (56, 75), (28, 38)),
    ("block3", 1,  512, 256, 1024, 1, True,  (28, 38), (28, 38)),
    ("block3", 2, 1024, 256, 1024, 1, False, (28, 38), (28, 38)),
    ("block3", 3, 1024, 256, 1024, 1, False, (28, 38), (28, 38)),
    ("block3", 4, 1024, 256, 1024, 1, False, (28, 38), (28, 38)),
    ("block3", 5, 1024, 256, 1024, 1, False, (28, 38), (28, 38)),
    ("block3", 6, 1024, 256, 1024, 2, False, (28, 38), (14, 19)),
    ("block4", 1, 1024, 512, 2048, 1, True,  (14, 19), (14, 19)),
    ("block4", 2, 2048, 512, 2048, 1, False, (14, 19), (14, 19)),
    ("block4", 3, 2048, 512, 2048, 1, False, (14, 19), (14, 19)),
]


def test_trunk_geometry_against_the_written_out_table():
    """per-unit depths, strides, shortcut kinds and spatial sizes of BOTH generators (oracle and product host) against
    the literal table above, and the oracle's forward pass against the table's spatial sizes (product and
    oracle shared one author-written table; this pins it independently)"""
    from acimg.vision import ResNet50Model

    ou = list(resnet50.units())
    pu = list(ResNet50Model(input_shape=[224, 298, 3], num_classes=None)._units())
    assert len(ou) == len(pu) == len(TRUNK_UNITS) == 16
    convs_o = dict((sc, (kh, kw, ci, co)) for sc, kh, kw, ci, co in resnet50.conv_layers())
    for (blk, u, din, db, d, s, proj, hw_in, hw_out), o, q in zip(TRUNK_UNITS, ou, pu):
        scope = "resnet_v1_50/%s/unit_%d/bottleneck_v1" % (blk, u)
        assert tuple(o) == (scope, din, d, db, s), (o, scope)
        assert (q[0].split("resnet_v1_50/")[1], q[1], q[2], q[3], q[4]) == (scope.split("resnet_v1_50/")[1], din, d, db, s), q
        assert ((scope + "/shortcut") in convs_o) == proj
        if proj:
            assert convs_o[scope + "/shortcut"] == (1, 1, din, d)
        assert convs_o[scope + "/conv1"] == (1, 1, din, db)
        assert convs_o[scope + "/conv2"] == (3, 3, db, db)
        assert convs_o[scope + "/conv3"] == (1, 1, db, d)
        # TF SAME / conv2d_same output size at stride s: ceil(in / s)
        assert tuple(-(-x // s) for x in hw_in) == hw_out
    # spatial chain of the table is consistent and ends where conv_map (3x4 VALID) gives 12 x 16
    for a, b in zip(TRUNK_UNITS[:-1], TRUNK_UNITS[1:]):
        assert a[8] == b[7] and a[4] == b[2]
    assert (TRUNK_UNITS[-1][8][0] - 3 + 1, TRUNK_UNITS[-1][8][1] - 4 + 1) == (12, 16)
    # the oracle's forward pass produces exactly these maps (batch 1, inference statistics: cheap)
    p = resnet50.init_params(seed=3)
    ep = {}
    torch.manual_seed(0)
    feat, _ = resnet50.forward(p, torch.rand(1, 224, 298, 3), False, end_points=ep)
    assert tuple(ep["resnet_v1_50/conv1"].shape) == (1, 112, 149, 64)
    assert tuple(ep["resnet_v1_50/pool1"].shape) == (1, 56, 75, 64)
    for blk, u, din, db, d, s, proj, hw_in, hw_out in TRUNK_UNITS:
        t = ep["resnet_v1_50/%s/unit_%d/bottleneck_v1" % (blk, u)]
        assert tuple(t.shape) == (1, hw_out[0], hw_out[1], d), (blk, u, t.shape)
    assert tuple(feat.shape) == (1, 12, 16, 12)


@pytest.mark.parametrize("num_skip,embedding", [(1, False), (2, False), (0, True)])
def test_oracle_step_shapes_and_learning(num_skip, embedding):
    o = trainer.Oracle(num_skip=num_skip, embedding=embedding, randomize=True, learning_rate=1e-3)
    ac, mf, vid, eps = trainer.synthetic_batch(2)
    ep = {}
    r0 = o.train_step(ac, mf, vid, eps, end_points=ep, keep_grads=True)
    assert r0["output"].shape == (2, 36, 48, 12)
    assert ep["resnet_v1_50/conv1"].shape == (2, 112, 149, 64) and ep["resnet_v1_50/pool1"].shape == (2, 56, 75, 64)
    assert ep["resnet_v1_50/block1/unit_3/bottleneck_v1"].shape == (2, 56, 75, 256)
    assert ep["resnet_v1_50/block2/unit_4/bottleneck_v1"].shape == (2, 28, 38, 512)
    assert ep["resnet_v1_50/block3/unit_6/bottleneck_v1"].shape == (2, 14, 19, 1024)
    assert ep["resnet_v1_50/block4/unit_3/bottleneck_v1"].shape == (2, 14, 19, 2048)
    assert ep["resnet_v1_50/conv_map"].shape == (2, 12, 16, 12)
    assert ep["features"].shape == (2, 12, 16, 145)
    assert all(torch.isfinite(g).all() for g in r0["grads"].values())
    assert float(r0["grads"]["resnet_v1_50/conv_map/weights"].abs().max()) > 0
    assert (r0["latent"] == 0) == embedding
    for _ in range(3):
        r = o.train_step(ac, mf, vid, eps)
    assert r["mse"] < r0["mse"]
    e = o.eval_step(ac, mf, vid, eps)
    assert abs(np.mean([e["mse%d" % i] for i in range(4)]) - e["mse"]) < 1e-6


# ---- RGB / spectrogram U-Net VAEs (BASELINE configs[0]/[1]) ---------------------------------
def test_unet_vae_oracle_inventory_and_shapes():
    from oracle import unet_vae as ouv

    # parameter counts of models/unet_architecture.py and models/unet_sound.py
    for model, want in (("UNet", 8.87e6), ("UNetSound", 3.68e6)):
        n = sum(int(torch.tensor(s).prod()) for k, s in ouv.param_shapes(model).items() if ouv.trainable(k))
        assert abs(n - want) / want < 2e-3, (model, n)
    names = ouv.param_shapes("UNet")
    # tf.layers names incl. the `pool_2` / `bn_pool_2` quirk (the loop index is reused, unet_architecture.py:171-178)
    for k in ("UNet/layer1/conv_1/kernel", "UNet/layer1/bn_2/moving_variance", "UNet/layer4/pool_2/kernel",
              "UNet/layer4/bn_pool_2/gamma", "UNet/mean/kernel", "UNet/variance/bias", "UNet/dense/kernel",
              "UNet/conv2d/kernel", "UNet/upsample_6/kernel", "UNet/final/bias"):
        assert k in names, k
    assert names["UNet/upsample_6/kernel"] == (2, 3, 64, 128) and names["UNet/mean/kernel"] == (14, 18, 128, 128)
    assert ouv.param_shapes("UNetSound")["UNetAudio/upsample_9/kernel"] == (3, 3, 8, 8)
    for model in ("UNet", "UNetSound"):
        p = ouv.init_params(model, dtype=torch.float64)
        x, eps = ouv.synthetic_batch(model, 1, dtype=torch.float64)
        fw, stats = ouv.forward(p, x, eps, model, True)
        assert fw["output"].shape == x.shape and fw["mean"].shape == (1, 128)
        ls = ouv.losses(p, x, fw, model)
        # trainer/trainer.py:58-73: total = MSE + Huber + regularisers + KL/1e6
        assert abs(float(ls["loss"]) - float(ls["mse"] + ls["huber"] + ls["reg"] + ls["latent"])) < 1e-12
        # moving variance advances with the UNBIASED batch variance, momentum 0.99
        k = [n for n in stats if n.endswith("layer1/bn_1/moving_variance")][0]
        assert float((stats[k] - 1.0).abs().max()) < 0.011


def test_unet_vae_host_graph_builds_on_cpu():
    """the recorded plan (no launches) and the TF-named state dict of the HIP host model agree with the oracle"""
    from acimg.session import Session
    from acimg.trainer_vae import TrainerVAE
    from acimg.unet_vae import UNet, UNetSound
    from oracle import unet_vae as ouv

    for cls, model in ((UNet, "UNet"), (UNetSound, "UNetSound")):
        sess = Session(torch.device("cpu"))
        tr = TrainerVAE(cls(), session=sess)
        g = tr._build_functions(batch_size=2)
        assert len(g.plan_train) > 100
        tr.model.initialize(seed=1)
        sd = sess.store.state_dict()
        want = ouv.param_shapes(model)
        assert set(sd) == set(want)
        for k, shape in want.items():
            assert tuple(sd[k].shape) == tuple(shape), k
        assert sorted(tr.model.train_vars) == sorted(k for k in want if ouv.trainable(k))


def test_dualcamnet_oracle_and_host_graph():
    """0.25 M parameters, conv3d SAME temporal padding 5/6, clip-mean logits; host graph builds on CPU"""
    from acimg.dualcamnet import DualCamHybridModel
    from acimg.session import Session
    from oracle import dualcamnet as odc

    shapes = odc.param_shapes(14)
    n = sum(int(torch.tensor(s).prod()) for s in shapes.values())
    assert abs(n - 0.25e6) / 0.25e6 < 0.05, n
    p = odc.init_params(14, dtype=torch.float64, std=0.05)
    x = torch.rand(24, 36, 48, 12, dtype=torch.float64)
    fl, _ = odc.forward(p, x)
    assert fl.shape == (24, 14)
    # temporal SAME padding: frame f of the output sees frames f-5 .. f+6 -> changing frame 11 of clip 0 cannot
    # reach output frame 5-6=... of clip 1, and reaches clip 0's frames 5..11 only
    x2 = x.clone()
    x2[11] += 1.0
    fl2, _ = odc.forward(p, x2)
    changed = (fl2 - fl).abs().amax(1) > 1e-12
    assert changed[5:12].all() and not changed[:5].any() and not changed[12:].any()
    loss, acc, logits = odc.loss_and_accuracy(fl, torch.tensor([1, 2]))
    assert logits.shape == (2, 14) and torch.allclose(logits[0], fl[:12].mean(0))
    sess = Session(torch.device("cpu"))
    m = DualCamHybridModel(input_shape=[36, 48, 12], num_classes=14)
    m._build_model(torch.zeros(24, 36, 48, 12), session=sess)
    sess.finalize()
    m.initialize()
    sd = m.state_dict_tf()
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)


def test_unet_acoustic_oracle_and_host_graph():
    """SURVEY A.3 last row: unet_noconc / unet_z, 9.32 M parameters; TF-named state of the HIP host model"""
    from acimg.session import Session
    from acimg.trainer_vae import TrainerVAE
    from acimg.unet_acoustic import UNetAcNoConc, UNetAcZ
    from oracle import unet_acoustic as oa

    want = oa.param_shapes()
    n = sum(int(torch.tensor(s).prod()) for s in want.values())
    assert abs(n - 9.32e6) / 9.32e6 < 2e-3, n
    p = oa.init_params(dtype=torch.float64)
    x, eps = torch.rand(2, 36, 48, 12, dtype=torch.float64), torch.randn(2, 150, dtype=torch.float64)
    own = oa.forward(p, x, eps)
    ext = oa.forward(p, x, eps, own["mean"], own["std"])          # unet_z fed its own statistics = unet_noconc
    assert torch.allclose(own["output"], ext["output"], atol=1e-12)
    sess = Session(torch.device("cpu"))
    tr = TrainerVAE(UNetAcNoConc(), session=sess)
    g = tr._build_functions(batch_size=2)
    tr.model.initialize()
    sd = sess.store.state_dict()
    assert set(sd) == set(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    sess2 = Session(torch.device("cpu"))
    m = UNetAcZ()
    e = torch.zeros(2, 300)
    m._build_model(torch.zeros(2, 36, 48, 12), e[:, :150], e[:, 150:], session=sess2)
    assert len(m.plan_fwd) == len(tr.model.plan_fwd) + 1      # + the external reparameterisation


@pytest.mark.parametrize("which,n_params", [("AssociatorVideoAc", 2 * (1024 * 512 + 512 * 512 + 512 * 256 + 256 * 256
                                                                    + 256 * 150 + 150 * 150 + 512 + 512 + 256 + 256
                                                                    + 150 + 150)),
                                            ("AssociatorAudioAc", 2 * (256 * 256 + 256 * 256 + 256 * 150 + 256 + 256
                                                                    + 150))])
def test_associator_oracle_and_host_graph(which, n_params):
    """models/multimodal.py:5-137: two dense towers, tf.layers default names in creation order, softplus std;
    the host graph registers the same TF-named variables and its Adam range covers exactly the associator"""
    from acimg import multimodal
    from acimg.session import Session
    from acimg.trainer_associator import TrainerAssociator
    from acimg.unet_acoustic import UNetAcZ
    from oracle import multimodal as om

    want = om.param_shapes(which)
    assert sum(int(torch.tensor(s).prod()) for s in want.values()) == n_params
    din, widths = om.TOWERS[which]
    assert list(want)[0] == which + "/dense/kernel" and list(want)[-1] == "%s/dense_%d/bias" % (which, 2 * len(widths) - 1)
    p = om.init_params(which, dtype=torch.float64)
    m, s, raw, masks = om.forward(p, which, torch.randn(3, din, dtype=torch.float64),
                                  torch.rand(3, din, dtype=torch.float64))
    assert m.shape == s.shape == (3, 150) and bool((s > 0).all())
    assert torch.allclose(s, torch.log1p(torch.exp(raw)))
    assert len(masks) == 2 * (len(widths) - 1)
    cls = getattr(multimodal, which)
    sess = Session(torch.device("cpu"))
    tr = TrainerAssociator(cls(), UNetAcZ(), session=sess)
    g = tr._build_functions(batch_size=2)
    tr.modelassociator.initialize()
    tr.modelac.initialize()
    sd = {k: v for k, v in sess.store.state_dict().items() if k.startswith(which + "/")}
    assert set(sd) == set(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    assert tr.modelassociator.train_vars and all(n.startswith(which + "/") for n in tr.modelassociator.train_vars)
    # the Adam range [off, off + numel) of the flat buffer holds the associator's variables and no others
    for n, o, c in sess.store.train_ranges():
        assert (g.off <= o and o + c <= g.off + g.numel) == n.startswith(which + "/"), n


def test_triplet_oracle_against_loops():
    """trainer/trainer_three.py:551-732 restated with broadcasting vs explicit loops over (anchor, positive,
    negative); the distance keeps the reference's broadcast of the two norms"""
    from oracle import triplet as ot
    g = torch.Generator().manual_seed(5)
    B, D, margin = 9, 4, 0.4
    e0 = torch.randn(B, D, generator=g, dtype=torch.float64)
    e1 = torch.randn(B, D, generator=g, dtype=torch.float64)
    labels = torch.tensor([0, 0, 1, 1, 2, 0, 1, 2, 2])
    scenario = torch.tensor([0, 0, 0, 1, 1, 0, 1, 1, 0])
    d = ot.pairwise_distances(e0, e1)
    same = lambda a, b: bool(labels[a] == labels[b] and scenario[a] == scenario[b])
    for i in range(B):
        for j in range(B):
            want = max(float((e0[j] ** 2).sum() - 2 * (e0[i] * e1[j]).sum() + (e1[i] ** 2).sum()), 0.0)
            assert abs(float(d[i, j]) - want) < 1e-12
    s, npos, nvalid = 0.0, 0, 0
    for a in range(B):
        for p in range(B):
            for n in range(B):
                if same(a, p) and not same(a, n):
                    nvalid += 1
                    t = max(float(d[a, p] - d[a, n]) + margin, 0.0)
                    s += t
                    npos += t > 1e-16
    loss, frac, np_, nv_ = ot.mix_all(e0, e1, labels, scenario, margin)
    assert abs(float(loss) - s / npos) < 1e-12 and int(np_) == npos and int(nv_) == nvalid
    assert abs(float(frac) - npos / nvalid) < 1e-12
    tl = []
    for a in range(B):
        hp = max(float(d[a, p]) if same(a, p) else 0.0 for p in range(B))
        rm = float(d[a].max())
        hn = min(float(d[a, n]) + (rm if same(a, n) else 0.0) for n in range(B))
        tl.append(max(hp - hn + margin, 0.0))
    loss_h, frac_h, _, _ = ot.mix_data_hard(e0, e1, labels, scenario, margin)
    assert abs(float(loss_h) - sum(tl) / B) < 1e-12
    assert abs(float(frac_h) - sum(t > 1e-16 for t in tl) / nvalid) < 1e-12
    # tied maxima share the gradient (tf.reduce_max), the hinge passes it at exactly zero (tf.maximum)
    x = torch.tensor([[1.0, 3.0, 3.0]], dtype=torch.float64, requires_grad=True)
    torch.amax(x, dim=1).sum().backward()
    assert x.grad.tolist() == [[0.0, 0.5, 0.5]]
    z = torch.zeros(2, dtype=torch.float64, requires_grad=True)
    ot._tf_max0(z).sum().backward()
    assert z.grad.tolist() == [1.0, 1.0]


@pytest.mark.parametrize("which,widths", [("Jointmvae", (128, 512, 128)), ("JointTwomvae", (512, 128)),
                                          ("JointTwomvae2", (512, 128))])
def test_joint_mlp_oracle_and_host_graph(which, widths):
    """models/multimodal.py:287-465: concat -> 3 x dense 512 -> ReLU heads (133 / 512 / 128); per-pixel on 12x16 maps"""
    from acimg import multimodal
    from acimg.session import Session
    from oracle import multimodal as om

    ctot = sum(widths)
    want = om.joint_param_shapes(which, ctot)
    heads = om.JOINT_HEADS[which]
    assert len(want) == 2 * (3 + len(heads)) and want[which + "/dense/kernel"] == (ctot, 512)
    p = om.joint_init_params(which, ctot, dtype=torch.float64)
    ins = [torch.rand(2, 12, 16, w, dtype=torch.float64) for w in widths]
    outs, masks = om.joint_forward(p, which, ins)
    assert [tuple(v.shape) for v in outs.values()] == [(2, 12, 16, w) for _, w in heads]
    assert all(bool((v >= 0).all()) for v in outs.values()) and len(masks) == 3 + len(heads)
    sess = Session(torch.device("cpu"))
    buf = sess.zeros(2, 12, 16, ctot)
    parts, o = [], 0
    for w in widths:
        parts.append(buf[..., o:o + w])
        o += w
    m = getattr(multimodal, which)()
    m._build_model(*parts, session=sess)
    sess.finalize()
    m.initialize()
    sd = {k: v for k, v in sess.store.state_dict().items() if k.startswith(which + "/")}
    assert set(sd) == set(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    with pytest.raises(AssertionError):                       # inputs in the wrong order are not a concat
        getattr(multimodal, which)()._build_model(*parts[::-1], session=Session(torch.device("cpu")))


@pytest.mark.parametrize("mode", ["all", "fusion", "onlyaudiovideo"])
def test_joint_trainer_branches_oracle_and_host_graph(mode):
    """the three branches of trainer/trainermulti.py `_build_functions` (FLAGS.fusion / FLAGS.onlyaudiovideo, main.py:194-201):
    the oracle's loss terms and trained variable set per mode, and the device plan recorded on a CPU session (which MLP
    the Adam range covers, which decoders exist)"""
    from acimg.multimodal import Jointmvae, JointTwomvae, JointTwomvae2
    from acimg.session import Session
    from acimg.trainer_multi import TrainerMulti
    from acimg.unet_joint import UNetAc2, UNetSound22, Unet2
    from oracle import joint

    scope = {"all": "Jointmvae", "fusion": "JointTwomvae2", "onlyaudiovideo": "JointTwomvae"}[mode]
    o = joint.Oracle(learning_rate=1e-3, mode=mode)
    assert all(k.startswith(scope + "/") for k in o.pj)
    assert o.pj[scope + "/dense/kernel"].shape[0] == (773 if mode == "all" else 640)     # 133 + 512 + 128 | 512 + 128
    assert (o.pj0 is not None) == (mode == "onlyaudiovideo")
    b, e = joint.synthetic_batch(1)
    r = o.train_step(b, e, apply=False, moddrop_on=0.0 if mode == "all" else None)
    ls = r["losses"]
    if mode == "onlyaudiovideo":
        assert "feature" in ls and "mse_video" not in ls and len(r["grads"]) == 8
        assert abs(ls["loss"] - (ls["mse"] + ls["huber"] + ls["latent"] + ls["feature"] + ls["reg"])) < 1e-6
        full = sum(float(joint.regulariser(model, o.params[m])) for m, model in joint.ORDER)
        assert 0 < ls["reg"] < full                    # encoder regularisers only: the video / audio decoders are not built
    else:
        assert set(k for k in ls if k.startswith("mse_")) == {"mse_ac", "mse_video", "mse_audio"} and len(r["grads"]) == 12
    if mode == "all":                                  # modDrop with the step's draw at 0: the acoustic map is zeros
        assert float(r["feats"]["ac"].abs().max()) == 0.0
    sess = Session(torch.device("cpu"))
    assoc = {"all": Jointmvae, "fusion": JointTwomvae2, "onlyaudiovideo": Jointmvae}[mode]()
    assoc1 = JointTwomvae() if mode == "onlyaudiovideo" else None
    tr = TrainerMulti(UNetAc2([36, 48, 12]), UNetSound22([193, 257, 1]), Unet2([224, 298, 3]), assoc, assoc1, session=sess,
                      mode=mode, moddrop=mode == "all")
    g = tr._build_functions(batch_size=2)
    g.plan_train.finalize()
    names = [n for n, off, c in sess.store.train_ranges() if g.off <= off < g.off + g.numel]
    assert names and all(n.startswith(scope + "/") for n in names), names[:3]
    assert list(g.mods) == (["ac"] if mode == "onlyaudiovideo" else ["ac", "audio", "video"])
    assert (g.moddrop_mask is not None) == (mode == "all") and (g.feat_sum is not None) == (mode == "onlyaudiovideo")
    with pytest.raises(AssertionError):
        TrainerMulti(UNetAc2([36, 48, 12]), UNetSound22([193, 257, 1]), Unet2([224, 298, 3]), Jointmvae(), None,
                     session=Session(torch.device("cpu")), mode="onlyaudiovideo")


def test_stft_and_resize_restatements_known_answers():
    """oracle.frontend.stft_mag / resize_bilinear (TensorFlow definitions, unpinned): closed-form cases"""
    n = 12288
    t = np.arange(n, dtype=np.float32)
    spec = frontend.stft_mag(np.cos(2 * np.pi * 32 * t / 512)[None, :].astype(np.float32))
    assert spec.shape == (1, 99, 257)
    assert np.all(spec[0].argmax(1) == 32)
    # a windowed cosine on an exact bin: |X[32]| = sum(window) / 2 = 246 / 4 (periodic Hann sums to N / 2)
    np.testing.assert_allclose(spec[0, :, 32], 61.5, rtol=2e-4)
    # DC input: bin 0 = sum of the window = 123
    dc = frontend.stft_mag(np.ones((1, 1000), np.float32))
    assert dc.shape == (1, 7, 257)
    np.testing.assert_allclose(dc[0, :, 0], 123.0, rtol=1e-6)
    w = frontend.build_wav(np.array([[3, -8], [4, 2]], np.int32))
    np.testing.assert_array_equal(w, np.array([3, -8, 4, 2], np.float32) / 8)
    # resize: identity at equal size; 2 -> 4 rows samples src rows 0, .5, 1, 1(clamped)
    x = np.arange(12, dtype=np.float32).reshape(1, 2, 3, 2)
    np.testing.assert_array_equal(frontend.resize_bilinear(x, 2, 3), x)
    up = frontend.resize_bilinear(x, 4, 3)
    np.testing.assert_array_equal(up[0, :, 0, 0], [0, 3, 6, 6])
    up = frontend.resize_bilinear(np.zeros((1, 99, 257, 1), np.float32), 193, 257)
    assert up.shape == (1, 193, 257, 1)


def test_host_butterworth_design_equals_scipy():
    """acimg.frontend.butter_lowpass / lfilter_zi (NumPy-only host code of the product) reproduce SciPy's design of
    dataloader/outdoor_data_mfcc.py:565-569 bit for bit, and SURVEY App. D's known answers"""
    from scipy import signal

    from acimg import frontend as fe

    b, a = fe.butter_lowpass(125, 10)
    b2, a2 = signal.butter(10, 125 / (0.5 * 12288), btype="low", analog=False)
    np.testing.assert_array_equal(b, b2)
    np.testing.assert_array_equal(a, a2)
    np.testing.assert_array_equal(fe.lfilter_zi(b, a), signal.lfilter_zi(b2, a2))
    assert len(b) == 11 and abs(b[0] - 9.0892e-16) < 1e-19 and abs(a[1] + 9.5914255) < 1e-6
    np.testing.assert_allclose(fe.hann_window_periodic(246).sum(), 123.0, rtol=1e-6)


def test_f16_operand_mode_of_the_trunk_oracle():
    """oracle/resnet50.py f16_operands (the reference statement of the product's precision="f16"): conv operands
    carry 11 significant bits, everything else stays fp32; the effect on the feature is real but bounded"""
    from oracle import resnet50 as ores
    from oracle import trainer as otr

    t = torch.tensor([1.0, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20, 300.0, 1e-5])   # exact, tie -> even, above the tie
    q = ores._f16_operand(t, 0.25)
    assert q[0] == 1.0 and q[1] == 1.0 and q[2] == 1.0 + 2.0 ** -10 and abs(q[3] - 300.0) <= 0.125
    assert abs(float(q[4]) - 1e-5) < 2.5e-7            # 2^-2 scaling keeps small activations out of fp16's subnormals
    w = ores._f16_operand(torch.tensor([1e-3, 60.0]), 1024.0)
    assert abs(float(w[0]) - 1e-3) < 1e-6 and float(w[1]) == 60.0
    o32 = otr.Oracle(randomize=True)
    o16 = otr.Oracle(randomize=True, f16_operands=True)
    ac, mf, vid, eps = otr.synthetic_batch(1, seed=3)
    with torch.no_grad():
        f32_, _ = ores.forward(o32.res, vid, False)
        f16a, _ = ores.forward(o16.res, vid, False, f16_operands=True)
        f16b, _ = ores.forward(o16.res, vid, False, f16_operands=True)
    assert torch.equal(f16a, f16b)
    d = float((f16a - f32_).abs().max() / f32_.abs().max())
    assert 1e-5 < d < 0.5, d


def test_bf16_operand_mode_of_the_unet_oracle():
    """oracle/unet_vae.py `_Bf16Conv` (the reference statement of the product's precision="bf16", BASELINE
    configs[1]): forward = conv of the bf16-rounded operands; data gradient from (r(gy), r(w)); weight gradient from
    (r(x), r(gy)); bias gradient = sum r(gy) — checked against independent F.conv2d / conv_transpose2d statements;
    the layer predicate matches acimg/unet_vae.py `_use_split`."""
    import torch.nn.functional as F
    from oracle import unet_vae as ouv

    t = torch.tensor([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -16, 3e38, 1e-30], dtype=torch.float64)
    q = ouv.bf16_round(t)           # exact, tie -> even, above the tie; fp32's range is kept
    assert q[0] == 1.0 and q[1] == 1.0 and q[2] == 1.0 + 2.0 ** -7 and abs(q[3] / 3e38 - 1) < 2.0 ** -8 and q[4] > 0
    assert ouv.bf16_layer(2, 112, 149, 32, 32, 1) and not ouv.bf16_layer(2, 112, 149, 8, 32, 1)
    assert not ouv.bf16_layer(2, 112, 149, 32, 32, 2) and not ouv.bf16_layer(2, 14, 18, 128, 128, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 9, 11, 32, generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn(3, 3, 32, 32, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = torch.randn(32, generator=g, dtype=torch.float64).requires_grad_(True)
    gy = torch.randn(2, 9, 11, 32, generator=g, dtype=torch.float64)
    y = ouv._Bf16Conv.apply(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    r = ouv.bf16_round
    xr, wr, gr = r(x.detach()).permute(0, 3, 1, 2), r(w.detach()).permute(3, 2, 0, 1), r(gy).permute(0, 3, 1, 2)
    assert torch.allclose(y.detach().permute(0, 3, 1, 2), F.conv2d(xr, wr, b.detach(), padding=1), rtol=0, atol=1e-12)
    assert torch.allclose(gx.permute(0, 3, 1, 2), F.conv_transpose2d(gr, wr, padding=1), rtol=0, atol=1e-12)
    want_gw = torch.stack([torch.stack([(F.pad(xr, (1, 1, 1, 1))[:, :, i:i + 9, j:j + 11].unsqueeze(1) *
                                         gr.unsqueeze(2)).sum((0, 3, 4)) for j in range(3)]) for i in range(3)])
    assert torch.allclose(gw, want_gw.permute(0, 1, 3, 2), rtol=0, atol=1e-11)     # [kh][kw][cin][cout]
    assert torch.allclose(gb, gr.sum((0, 2, 3)), rtol=0, atol=1e-12)
    # the rounding is in the gradients too: an un-rounded gy gives a different data gradient
    assert not torch.allclose(gx.permute(0, 3, 1, 2), F.conv_transpose2d(gy.permute(0, 3, 1, 2), wr, padding=1), rtol=0, atol=1e-6)
