"""DualCamNet classifier head (BASELINE configs[4]): parity of forward, clip-level softmax
cross-entropy, accuracy and every gradient against the CPU oracle (fp64), and the classifier-on-generated-images
train step of trainer/trainer_reconstructed_class.py end to end."""
import os
import sys

import numpy as np
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


def test_dualcamnet_forward_backward():
    from acimg import ops
    from acimg.dualcamnet import DualCamHybridModel
    from acimg.params import up4
    from acimg.session import Session
    from oracle import dualcamnet as odc

    dev = torch.device("cuda:0")
    K, clips = 14, 3
    NF = clips * 12
    sess = Session(dev)
    m = DualCamHybridModel(input_shape=[36, 48, 12], num_classes=K)
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(NF, 36, 48, 12, generator=gen, dtype=torch.float64)
    labels = torch.tensor([3, 0, 13])
    xd = x.float().to(dev)
    m._build_model(xd, session=sess)
    kp = up4(K)
    out = sess.zeros(4)
    g_logits = sess.zeros(NF, kp)
    lab = labels.to(torch.int32).to(dev)
    p = sess.new_plan()
    ops.zero(p, out)
    p.extend(m.plan_fwd)
    ops.clip_softmax_ce(p, m.logits, kp, clips, 12, K, lab, out, g_logits, kp)
    m.record_backward(p, g_logits)
    sess.finalize()
    # larger-than-default weights so every layer carries signal (sigma 0.01 squashes the logits to ~1e-6)
    params = odc.init_params(K, seed=5, dtype=torch.float64, std=0.05, bias_std=0.05)
    m.initialize(state={k: v.float() for k, v in params.items()})
    p.run()
    torch.cuda.synchronize()
    masks = {"conv1": (m.relu1.t > 0).cpu(), "conv2": (m.relu2.t > 0).cpu(), "conv3": (m.relu3.t > 0).cpu(),
             "full1": (m.relu4 > 0).cpu()}
    ref = odc.train_step_grads(params, x, labels, relu_masks=masks)
    assert rel(m.logits[:, :K], ref["frame_logits"]) < 1e-4
    loss, correct = out[:2].tolist()
    assert abs(loss - ref["loss"]) < 1e-5 * abs(ref["loss"]) + 1e-7
    assert abs(correct / clips - ref["accuracy"]) < 1e-6
    grads = sess.store.grad_dict()
    for name, gref in ref["grads"].items():
        g = grads[name].reshape(gref.shape)
        assert rel(g, gref) < 1e-3, (name, rel(g, gref))
    # TF-shaped export of the conv3d kernel
    assert tuple(m.state_dict_tf()["DualCamNet/conv1/weights"].shape) == (12, 1, 1, 12, 12)


def test_classifier_on_generated_images_step():
    """trainer_reconstructed_class.py end to end at 2 clips: the loss falls when the same batch is repeated, only
    DualCamNet variables move, the generator and trunk stay frozen."""
    from acimg.dualcamnet import DualCamHybridModel
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer_class import TrainerClass
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model

    dev = torch.device("cuda:0")
    FLAGS.model = "DualCamNet"
    sess = Session(dev)
    tr = TrainerClass(DualCamHybridModel(input_shape=[36, 48, 12], num_classes=14),
                      ResNet50Model(input_shape=[224, 298, 3], num_classes=None),
                      UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1), learning_rate=1e-3, session=sess)
    g = tr._build_functions(batch_size=24)
    tr.model_encoder_images.initialize(seed=1238)
    tr.model_encoder_acoustic.initialize(seed=1239)
    tr.model.initialize(seed=1241)
    gen = torch.Generator().manual_seed(9)
    video = torch.rand(24, 224, 298, 3, generator=gen)
    mfcc = torch.rand(24, 12, generator=gen)
    labels = torch.tensor([2, 7])
    eps = torch.randn(24, 150, generator=gen)
    before = {k: v.clone() for k, v in sess.store.state_dict().items()}
    first = tr.train_step((mfcc, video, labels), eps)
    for _ in range(30):
        last = tr.train_step(None, eps)
    assert abs(first["loss"] - 2.639) < 0.05            # ln(14): near-uniform logits at initialisation
    assert last["loss"] < first["loss"] - 0.05, (first, last)
    after = sess.store.state_dict()
    for k in before:
        moved = not torch.equal(before[k], after[k])
        assert moved == k.startswith("DualCamNet/"), k
    ev = tr.eval_step(None, eps)
    assert abs(ev["loss"] - tr.train_step(None, eps)["loss"]) < 1e-5


def test_configs4_classifier_step_at_60_frames():
    """BASELINE configs[4] at its per-GPU size (60 frames = 5 clips of 12 per GPU, global 480 on 8 GPUs):
    the train step of trainer/trainer_reconstructed_class.py:31-75 — ResNet-50-mod + UNetAcRes in inference mode
    (is_training 0, :183-186), clips of 12 generated frames (:44), DualCamNet per frame, clip-mean logits (:47-51),
    softmax CE (:52-56), Adam on DualCamNet/* (:61,71-73) — against the oracle on identical weights, inputs and noise:
    generated images 1e-3, loss / accuracy, every DualCamNet gradient 1e-3, TF-1 Adam update."""
    from acimg.dualcamnet import DualCamHybridModel
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer_class import TrainerClass
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model
    from oracle import dualcamnet as odc
    from oracle import tfsem
    from oracle import trainer as otr

    dev = torch.device("cuda:0")
    FLAGS.model, FLAGS.ae = "DualCamNet", 0
    NF, K, lr = 60, 14, 1e-3
    clips = NF // 12
    sess = Session(dev)
    m = DualCamHybridModel(input_shape=[36, 48, 12], num_classes=K)
    tr = TrainerClass(m, ResNet50Model(input_shape=[224, 298, 3], num_classes=None),
                      UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1), learning_rate=lr, session=sess)
    g = tr._build_functions(batch_size=NF)
    orc = otr.Oracle(num_skip=1, randomize=True)
    params = odc.init_params(K, seed=11, dtype=torch.float32, std=0.05, bias_std=0.05)
    state = dict(orc.state_dict())
    state.update(params)
    sess.store.load_state(m._to_internal(state), strict=True)
    _, mfcc, video, eps = otr.synthetic_batch(NF, seed=123)
    labels = torch.tensor([1, 5, 0, 13, 7])
    before = sess.store.state_dict()
    got = tr.train_step((mfcc, video, labels), eps)
    with torch.no_grad():
        _, _, gen_ref, _ = orc.forward(video, mfcc, eps, False)
    assert rel(tr.model_encoder_acoustic.output, gen_ref) < 1e-3, "generated frames"
    masks = {"conv1": (m.relu1.t > 0).cpu(), "conv2": (m.relu2.t > 0).cpu(), "conv3": (m.relu3.t > 0).cpu(),
             "full1": (m.relu4 > 0).cpu()}
    # the classifier differentiates from OUR generated frames (the generator is checked above, and frozen here)
    x = tr.model_encoder_acoustic.output.detach().cpu().double()
    p64 = {k: v.double() for k, v in params.items()}
    ref = odc.train_step_grads(p64, x, labels, relu_masks=masks)
    free = odc.train_step_grads(p64, gen_ref.double(), labels)
    flips = sum(int((masks[k].reshape(-1) != free["masks"][k].reshape(-1)).sum()) for k in masks)
    elems = sum(v.numel() for v in masks.values())
    assert flips <= max(8, 3.2e-6 * elems), (flips, elems)
    assert abs(got["loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (got, ref["loss"])
    assert abs(got["loss"] - free["loss"]) <= 1e-3 * abs(free["loss"])
    assert abs(got["accuracy"] - ref["accuracy"]) < 1e-6
    assert rel(m.logits[:, :K], ref["frame_logits"]) < 1e-4
    grads = sess.store.grad_dict()
    for name, gref in ref["grads"].items():
        assert rel(grads[name].reshape(gref.shape), gref) < 1e-3, (name, rel(grads[name].reshape(gref.shape), gref))
    after = sess.store.state_dict()
    for k in before:
        if k.startswith("DualCamNet/"):
            p2, _, _ = tfsem.adam_tf1(before[k].double(), grads[k].double().reshape(before[k].shape),
                                      torch.zeros_like(before[k]).double(), torch.zeros_like(before[k]).double(), 1, lr)
            assert float((after[k].double() - p2).abs().max()) <= 2e-6 * max(float(p2.abs().max()), 1e-3), k
        else:
            assert torch.equal(before[k], after[k]), "frozen variable moved: " + k


def test_configs4_fp16_operand_storage():
    """BASELINE configs[4] "fp16 with fp32 loss accumulation": ResNet50Model(precision="f16") runs the 52 trunk convs
    on the hi fp16 plane of activations and weights only (one MFMA per product, fp32 accumulation, fp32 batch-norm
    statistics, fp32 losses).  Checked against the oracle with the SAME operand rounding (oracle/resnet50.py
    f16_operands): generated frames within 1e-3; and against the full-precision oracle to show what the storage costs
    on the random-initialised trunk."""
    from acimg.dualcamnet import DualCamHybridModel
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer_class import TrainerClass
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model
    from oracle import dualcamnet as odc
    from oracle import trainer as otr

    dev = torch.device("cuda:0")
    FLAGS.model, FLAGS.ae = "DualCamNet", 0
    NF, K = 24, 14
    sess = Session(dev)
    m = DualCamHybridModel(input_shape=[36, 48, 12], num_classes=K)
    tr = TrainerClass(m, ResNet50Model(input_shape=[224, 298, 3], num_classes=None, precision="f16"),
                      UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1), learning_rate=1e-3, session=sess)
    tr._build_functions(batch_size=NF)
    orc16 = otr.Oracle(num_skip=1, randomize=True, f16_operands=True)
    orc32 = otr.Oracle(num_skip=1, randomize=True)
    params = odc.init_params(K, seed=11, dtype=torch.float32, std=0.05, bias_std=0.05)
    state = dict(orc16.state_dict())
    state.update(params)
    sess.store.load_state(m._to_internal(state), strict=True)
    _, mfcc, video, eps = otr.synthetic_batch(NF, seed=55)
    labels = torch.tensor([3, 9])
    got = tr.train_step((mfcc, video, labels), eps)
    with torch.no_grad():
        ep16, ep32 = {}, {}
        _, _, gen16, _ = orc16.forward(video, mfcc, eps, False, ep16)
        _, _, gen32, _ = orc32.forward(video, mfcc, eps, False, ep32)
    # the same rounded arithmetic evaluated in fp64 (same fp32 weights): what the ORACLE ITSELF moves by when only its
    # accumulation precision changes - values within fp32 rounding of an fp16 tie round apart and the 53-layer random-init
    # trunk amplifies it.  The product is held to 3x that spread (the bound of tests/test_unet_vae_gpu.py's bf16 mode): the
    # 1e-2 cap below is not the bar, the oracle's own discontinuity is ("10x the stated bar")
    orc64 = otr.Oracle(num_skip=1, randomize=True, f16_operands=True, dtype=torch.float64)
    for k in orc64.res:
        orc64.res[k] = orc16.res[k].double()
    for k in orc64.gen:
        orc64.gen[k] = orc16.gen[k].double()
    with torch.no_grad():
        ep64 = {}
        orc64.forward(video.double(), mfcc.double(), eps.double(), False, ep64)
    feat = tr.model_encoder_images.output
    spread = rel(ep16["resnet_v1_50/conv_map"], ep64["resnet_v1_50/conv_map"])
    e_feat64 = rel(feat, ep64["resnet_v1_50/conv_map"])
    print("fp16 operand storage: oracle fp32 vs fp64 under the same rounding %.2e; product vs the fp64 evaluation %.2e"
          % (spread, e_feat64))
    assert e_feat64 <= 3.0 * spread + 1e-4, (e_feat64, spread)
    e_feat = rel(feat, ep16["resnet_v1_50/conv_map"])
    e_gen = rel(tr.model_encoder_acoustic.output, gen16)
    cost = rel(ep16["resnet_v1_50/conv_map"], ep32["resnet_v1_50/conv_map"])
    print("fp16 operand storage: feature vs same-rounding oracle %.2e, frames %.2e; cost of the storage itself "
          "(oracle f16 vs f32 feature) %.2e" % (e_feat, e_gen, cost))
    # both sides round every conv operand to fp16, but a value within fp32 rounding of an fp16 tie rounds apart and
    # the 53-layer random-init trunk amplifies that (the kernel itself is exact on rounded operands:
    # tests/test_ops_gpu.py::test_fp16_operand_storage_conv): the bar is 1e-2 on the feature, 1e-3 on the frames it
    # conditions, and the storage format itself moves the feature by no more than that either
    assert e_feat < 1e-2 and e_gen < 1e-3 and cost < 1e-2, (e_feat, e_gen, cost)
    assert np.isfinite(got["loss"]) and 0.0 <= got["accuracy"] <= 1.0
