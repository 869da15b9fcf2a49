"""The reference's trainer surface around the hot loop, end to end on the device:
`Trainer.train()` (trainer/mfcctrainer.py:249-398: _init_model, 'epoch_random' checkpoint, epoch loop over the loader,
validation, every-10-epochs and best-loss checkpoints, model.txt), `_save_checkpoint` (:400-406, a TensorFlow Saver-V2
bundle with the reference Saver's variable set), `_restore_model` (:236-247), `test()` (:476-536, the
test_accuracy_<epoch>.txt line with the four per-3-channel losses), `_evaluate` (:411-442)."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make(device, lr=1e-3, epochs=1):
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer import Trainer
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model

    FLAGS.model, FLAGS.ae, FLAGS.latent_loss = "UNet", 0, 1e-6
    sess = Session(device)
    tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=False, num_skip=1),
                 ResNet50Model(input_shape=[224, 298, 3], num_classes=None), display_freq=1, learning_rate=lr,
                 num_epochs=epochs, session=sess)
    return tr, sess


def test_train_checkpoint_restore_test(device, tmp_path):
    from acimg import tfio
    from acimg.data import SyntheticDataLoader
    from acimg.flags import FLAGS

    FLAGS.checkpoint_dir, FLAGS.exp_name = str(tmp_path), "exp"
    FLAGS.restore_checkpoint = FLAGS.init_checkpoint = None
    FLAGS.acoustic_init_checkpoint = FLAGS.visual_init_checkpoint = None
    tr, sess = make(device, epochs=3)
    lines = []
    tr.log = lines.append
    train_data = SyntheticDataLoader(5, 2, seed=10)       # batches of 2, 2, 1: the partial batch gets its own plans
    valid_data = SyntheticDataLoader(3, 2, seed=20)
    best = tr.train(train_data, valid_data)
    d = os.path.join(str(tmp_path), "exp")
    # --- log lines: 3 iterations per epoch, one validation line per epoch, the final best-epoch line (:352-357,:369-398)
    it = [ln for ln in lines if "Training_mse_Loss" in ln]
    va = [float(re.search(r"Validation_mse_Loss: ([0-9.]+)", ln).group(1)) for ln in lines if "- Epoch:" in ln]
    assert len(it) == 9 and len(va) == 3 and tr.global_step == 9
    assert re.search(r"Iteration: \[  0\]\t Training_mse_Loss: [0-9.]+\t Training_Loss: [0-9.]+", it[0])
    # --- best-epoch rule of :380-395: `<=`, so a later epoch wins ties; model.txt names it
    best_epoch = max(i for i, v in enumerate(va) if v == min(va))
    assert abs(best - min(va)) < 1e-6
    txt = open(os.path.join(d, "model.txt")).read()
    assert "Best Epoch: %d\n" % best_epoch in txt and "Validation_mse_Loss: %.6f\n" % min(va) in txt and "exp\n" in txt
    assert "Best Epoch: %d" % best_epoch in lines[-1]
    # --- checkpoints: 'random' before the loop (:322), epoch 0 (epoch %% 10 == 0 and first best), every new best
    saved = sorted(f[:-6] for f in os.listdir(d) if f.endswith(".index"))
    expect = {"epoch_random.ckpt", "epoch_0.ckpt"}
    run_best = 1e9
    for i, v in enumerate(va):
        if v <= run_best:
            run_best = v
            expect.add("epoch_%d.ckpt" % i)
    assert set(saved) == expect, (saved, expect)
    state = open(os.path.join(d, "checkpoint")).read()
    assert 'model_checkpoint_path: "epoch_%d.ckpt"' % best_epoch in state
    # --- the bundle holds the reference Saver's variable set under TF names
    ck = tfio.read_checkpoint(os.path.join(d, "epoch_%d.ckpt" % best_epoch), verify=True)
    sd = sess.store.state_dict()
    assert set(sd) <= set(ck)
    assert "UNetAcRes/layer6/conv_1/kernel/Adam" in ck and "UNetAcRes/mean/kernel/Adam_1" in ck
    assert "resnet_v1_50/conv_map/weights/Adam" in ck and "resnet_v1_50/block1/unit_1/bottleneck_v1/conv1/weights/Adam" not in ck
    assert ck["global_step"].dtype == np.int64 and int(ck["global_step"]) == 3 * (best_epoch + 1)
    assert abs(float(ck["beta1_power"]) - 0.9 ** (3 * (best_epoch + 1) + 1)) < 1e-7
    assert ck["UNetAcRes/mean/kernel"].shape == (12, 16, 145, 150) and ck["UNetAcRes/upsample_1/kernel"].shape == (2, 2, 128, 128)
    rnd = tfio.read_checkpoint(os.path.join(d, "epoch_random.ckpt"))
    assert int(rnd["global_step"]) == 0 and float(np.abs(rnd["UNetAcRes/dense/kernel/Adam"]).max()) == 0
    assert not np.array_equal(rnd["UNetAcRes/dense/kernel"], ck["UNetAcRes/dense/kernel"])      # training moved it
    assert np.array_equal(rnd["resnet_v1_50/conv1/weights"], ck["resnet_v1_50/conv1/weights"])  # trunk frozen
    assert not np.array_equal(rnd["resnet_v1_50/conv1/BatchNorm/moving_mean"],
                              ck["resnet_v1_50/conv1/BatchNorm/moving_mean"])                    # UPDATE_OPS ran
    # --- save the CURRENT weights, evaluate, then restore into a fresh trainer and run test()
    tr._save_checkpoint(sess, 99)
    test_data = SyntheticDataLoader(5, 2, seed=30)
    tr._noise_calls = 0
    want = tr._evaluate(sess, "test", test_data)
    FLAGS.restore_checkpoint = os.path.join(d, "epoch_99.ckpt")
    tr2, sess2 = make(device)
    out = []
    tr2.log = out.append
    got = tr2.test(test_data)
    assert abs(got - want) <= 1e-6 * want, (got, want)
    for k, v in sess.store.state_dict().items():
        assert torch.equal(v, sess2.store.state_dict()[k]), k
    line = open(os.path.join(d, "test_accuracy_99.txt")).read()
    m = re.match(r".* - Testing_Loss: ([0-9.]+)\t  Testing_Loss0: ([0-9.]+)\t Testing_Loss1: ([0-9.]+)\t "
                 r"Testing_Loss2: ([0-9.]+)\t Testing_Loss3: ([0-9.]+)$", line)
    assert m, line
    vals = [float(x) for x in m.groups()]
    assert abs(vals[0] - got) < 1e-6 and abs(sum(vals[1:]) / 4 - vals[0]) < 2e-6     # 4 x 3 channels partition the 12
    assert out[-1] == line
    # --- _init_model: --init_checkpoint loads the generator only (:188-212); trunk stays as initialised
    FLAGS.restore_checkpoint = None
    FLAGS.init_checkpoint = os.path.join(d, "epoch_99.ckpt")
    tr3, sess3 = make(device)
    tr3._build_functions(batch_size=2)
    tr3._init_model(sess3)
    sd3 = sess3.store.state_dict()
    assert torch.equal(sd3["UNetAcRes/final/kernel"], sess.store.state_dict()["UNetAcRes/final/kernel"])
    FLAGS.init_checkpoint = None
    # --- Saver(max_to_keep=11): the oldest bundles are deleted
    for e in range(100, 112):
        tr._save_checkpoint(sess, e)
    left = [f for f in os.listdir(d) if f.endswith(".index")]
    assert len(left) == 11 and "epoch_random.ckpt.index" not in left and "epoch_111.ckpt.index" in left


def test_init_model_from_known_answer_bundle(device):
    """`init_model(session, checkpoint_file)` of both models (models/unet_acresnet.py:33-41; models/vision.py:26-43
    = trainer/mfcctrainer.py:214-225: restore resnet_v1_50/* EXCEPT logits and conv_map) on the hand-built Saver-V2
    bundle of tests/golden/make_bundle_golden.py"""
    import importlib.util
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_bundle_golden", os.path.join(gold, "make_bundle_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    want = mk.tensors()
    prefix = os.path.join(gold, "bundle_known_answer", "model.ckpt")
    tr, sess = make(device)
    tr._build_functions(batch_size=2)
    tr.modelimages.initialize(seed=1)
    tr.modelac.initialize(seed=2)
    before = sess.store.state_dict()
    loaded_v = tr.modelimages.init_model(sess, prefix)
    loaded_a = tr.modelac.init_model(sess, prefix)
    after = sess.store.state_dict()
    assert sorted(loaded_v) == ["resnet_v1_50/conv1/BatchNorm/gamma", "resnet_v1_50/conv1/BatchNorm/moving_mean"]
    assert sorted(loaded_a) == ["UNetAcRes/final/bias", "UNetAcRes/layer7/conv_2/bias", "UNetAcRes/layer7/conv_2/kernel"]
    for k in loaded_v + loaded_a:
        assert np.array_equal(after[k].numpy(), want[k]), k
    # conv_map is NOT restored from a visual checkpoint (:218-221), everything else is untouched
    assert torch.equal(after["resnet_v1_50/conv_map/BatchNorm/beta"], before["resnet_v1_50/conv_map/BatchNorm/beta"])
    for k in before:
        if k not in loaded_v + loaded_a:
            assert torch.equal(before[k], after[k]), k
    # and the step still runs on the restored variables
    from acimg.data import SyntheticDataLoader
    b = next(iter(SyntheticDataLoader(2, 2, seed=3).data))
    r = tr.train_step((b[0], b[1], b[2]))
    assert np.isfinite(r["loss"])


def test_tfrecord_loader_feeds_the_trainer(device, tmp_path):
    """`TFRecordDataLoader` (acimg/data.py): the reference's on-disk format end to end - GZIP TFRecords of SequenceExamples
    (convert_data.py:247-279) -> native reader (`_parse_sequence`, dataloader/outdoor_data_mfcc.py:263-343, LR + UD flip)
    -> device low-pass + MFCC with `_normalize_mfcc` (:558-575, :696-703, :796-876) -> per-frame maps (:634-679) ->
    frames re-batched (:99-104) -> `Trainer.train()`.  Every tensor of every batch against the pinned CPU oracle of the
    front end and a NumPy restatement of the maps; then one epoch of training + validation straight from the files."""
    from collections import OrderedDict

    from acimg import tfio
    from acimg.data import TFRecordDataLoader
    from acimg.flags import FLAGS
    from oracle import frontend as ofe

    rng = np.random.RandomState(3)
    recs, truth = [], []
    for r in range(2):
        ai = rng.rand(12, 36, 48, 12).astype(np.float32) * 5 - 1
        sa = (rng.randn(12, 1024) * 800).astype(np.int32)
        vi = rng.randint(0, 256, size=(12, 224, 298, 3)).astype(np.uint8)
        ctx = OrderedDict([("classes", np.array([3 + r])), ("location", np.array([7])),
                           ("audio_image/height", np.array([36])), ("audio_image/width", np.array([48])),
                           ("audio_image/depth", np.array([12])), ("audio_data/mics", np.array([1])),
                           ("audio_data/samples", np.array([1024])), ("video/height", np.array([224])),
                           ("video/width", np.array([298])), ("video/depth", np.array([3]))])
        lists = OrderedDict([("audio/image", [a.tobytes() for a in ai]), ("audio/data", [s.tobytes() for s in sa]),
                             ("video/image", [v.tobytes() for v in vi])])
        recs.append(tfio.build_sequence_example(ctx, lists))
        truth.append((ai, sa, vi, 3 + r))
    paths = []
    for i, rec in enumerate(recs):
        p = str(tmp_path / ("part%d.tfrecord" % i))
        tfio.write_tfrecord(p, [rec], compression="GZIP")
        paths.append(p)
    listing = tmp_path / "train.txt"
    listing.write_text("\n".join(paths) + "\n")
    dl = TFRecordDataLoader(str(listing), 8, device=device)
    assert dl.num_samples == 24 and dl.total_batches == 3
    batches = list(dl.data)
    assert [b[0].shape[0] for b in batches] == [8, 8, 8]
    ac = torch.cat([b[0] for b in batches]).numpy()
    mf = torch.cat([b[1] for b in batches]).numpy()
    vid = torch.cat([b[2] for b in batches]).numpy()
    act = torch.cat([b[3] for b in batches]).numpy()
    mfl = torch.cat([b[5] for b in batches]).numpy()
    for r, (ai, sa, vi, cls) in enumerate(truth):
        sl = slice(12 * r, 12 * r + 12)
        a = ai[:, ::-1, ::-1, :].astype(np.float32)                      # flip_left_right + flip_up_down (:314-315)
        a = a - a.min(axis=(1, 2, 3), keepdims=True)
        a = a / a.max(axis=(1, 2, 3), keepdims=True)
        np.testing.assert_array_equal(ac[sl], a)
        np.testing.assert_array_equal(vid[sl], vi[..., ::-1].astype(np.float32) * np.float32(1.0 / 255.0))
        want = np.stack([ofe.normalize_mfcc(v) for v in ofe.mfcc(sa)])
        np.testing.assert_allclose(mf[sl], want, rtol=2e-5, atol=2e-6)
        low = ofe.butter_lowpass_filter(sa)
        want_low = np.stack([ofe.normalize_mfcc(v) for v in ofe.mfcc(low)])
        np.testing.assert_allclose(mfl[sl], want_low, rtol=2e-5, atol=2e-6)
        assert (act[sl].argmax(1) == cls).all() and act[sl].sum() == 12
    # one epoch of Trainer.train() straight from the files (training and validation loaders over the same two files)
    FLAGS.checkpoint_dir, FLAGS.exp_name = None, "tfr"
    FLAGS.restore_checkpoint = FLAGS.init_checkpoint = None
    FLAGS.acoustic_init_checkpoint = FLAGS.visual_init_checkpoint = None
    tr, sess = make(device, epochs=1)
    lines = []
    tr.log = lines.append
    best = tr.train(TFRecordDataLoader(paths, 8, device=device), TFRecordDataLoader(paths[:1], 8, device=device))
    assert tr.global_step == 3 and np.isfinite(best) and 0 < best < 1
    assert sum("Training_mse_Loss" in ln for ln in lines) == 3
