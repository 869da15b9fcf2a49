"""`TrainerClass`: trains the DualCamNet classifier on GENERATED acoustic images, MI355X-native.

Mirrors trainer/trainer_reconstructed_class.py:14-75: ResNet-50 image encoder + `UNetAc` generator run in
inference mode (`is_training: 0`, :183-186), their output is viewed as clips of 12 frames (:44), DualCamNet
classifies each frame, logits are averaged per clip (:47-48), loss = tf.losses.softmax_cross_entropy (:49-54),
Adam on the `DualCamNet/` variables only (:58-70).  `train_step` is the body of the loop at :176-190.

One step = ONE recorded plan: tile MFCC -> trunk forward (moving statistics) -> generator forward -> DualCamNet
forward -> clip softmax-CE (+ gradient, + accuracy count) -> DualCamNet backward; then Adam over the
contiguous DualCamNet range of the flat parameter buffer.
"""
from collections import OrderedDict

import torch

from . import _lib, ops
from .params import up4
from .session import Session
from .unet_acresnet import Z


class _Graph(object):
    pass


class TrainerClass(object):

    def __init__(self, model, model_encoder_images, model_encoder_acoustic, display_freq=1, learning_rate=0.0001,
                 num_classes=14, num_epochs=1, nr_frames=12, temporal_pooling=False, session=None):
        self.model = model
        self.model_encoder_images = model_encoder_images
        self.model_encoder_acoustic = model_encoder_acoustic
        self.display_freq = display_freq
        self.learning_rate = learning_rate
        self.num_classes = num_classes
        self.num_epochs = num_epochs
        self.nr_frames = nr_frames
        self.temporal_pooling = temporal_pooling
        self.session = session
        self.global_step = 0
        self.noise_seed = 1237

    def _build_functions(self, data=None, batch_size=None):
        """batch_size = number of FRAMES (a multiple of 12: clips of 12 consecutive frames)"""
        N = int(batch_size or getattr(data, "batch_size", None) or 24)
        assert N % self.nr_frames == 0, "the frame batch must hold whole clips of %d frames" % self.nr_frames
        if self.session is None:
            self.session = Session()
        sess = self.session
        z = sess.zeros
        g = _Graph()
        g.N, g.clips = N, N // self.nr_frames
        g.mfcc = z(N, 12)
        g.video = z(N, 224, 298, 3)
        g.eps = z(N, Z)
        g.mfccmap = z(N, 36, 48, 12)
        g.labels = torch.zeros(g.clips, dtype=torch.int32, device=sess.device)
        mi, ma, m = self.model_encoder_images, self.model_encoder_acoustic, self.model
        mi._build_model(g.video, session=sess)
        ma._build_model(g.mfccmap, mi.output, session=sess, eps=g.eps)
        m._build_model(ma.output, session=sess)
        kp = up4(self.num_classes)
        g.out = z(4)                  # loss, #correct
        g.g_logits = z(N, kp)
        p = sess.new_plan()
        ops.zero(p, g.out)
        ops.tile_mfcc(p, g.mfcc, g.mfccmap, N, 36 * 48, 12)
        p.extend(mi.plan_eval)
        p.extend(ma.plan_fwd)
        p.extend(m.plan_fwd)
        e = sess.new_plan()
        e.extend(p)
        ops.clip_softmax_ce(p, m.logits, kp, g.clips, self.nr_frames, self.num_classes, g.labels, g.out, g.g_logits, kp)
        m.record_backward(p, g.g_logits)
        ops.clip_softmax_ce(e, m.logits, kp, g.clips, self.nr_frames, self.num_classes, g.labels, g.out, None, kp)
        g.plan_train, g.plan_eval = p, e
        sess.finalize()
        # the DualCamNet variables are one contiguous run of the flat trainable buffer
        rng = [(n, o, c) for n, o, c in sess.store.train_ranges() if n.startswith(m.scope + "/")]
        g.off = rng[0][1]
        g.numel = rng[-1][1] + rng[-1][2] - g.off
        self.primary = g
        return g

    def _feed(self, g, batch, eps):
        if batch is not None:
            mfcc, video, labels = batch
            g.mfcc.copy_(mfcc.reshape(g.N, 12), non_blocking=True)
            g.video.copy_(video.reshape(g.N, 224, 298, 3), non_blocking=True)
            lab = labels.reshape(g.clips, -1)
            lab = lab.argmax(1) if lab.shape[1] > 1 else lab[:, 0]
            g.labels.copy_(lab.to(torch.int32), non_blocking=True)
        if eps is not None:
            g.eps.copy_(eps.reshape(g.N, Z), non_blocking=True)
        else:
            self._noise_calls = getattr(self, "_noise_calls", 0) + 1
            rc = _lib.load().acimg_randn(g.eps.data_ptr(), g.N * Z, self.noise_seed, self._noise_calls * 65536,
                                         ops.current_stream_handle(self.session.device))
            _lib.check(rc, "randn")

    def train_step(self, batch=None, eps=None):
        """batch: (mfcc [N,12], video [N,224,298,3], labels [clips] or one-hot [clips, classes]); returns
        {loss, accuracy}"""
        g = self.primary
        self._feed(g, batch, eps)
        g.plan_train.run()
        st = self.session.store
        self.global_step += 1
        lr_t = ops.adam_lr_t(self.learning_rate, self.global_step)
        o, n = g.off * 4, g.numel
        rc = _lib.load().acimg_adam_step(st.flat["train"].data_ptr() + o, st.grad.data_ptr() + o,
                                         st.adam_m.data_ptr() + o, st.adam_v.data_ptr() + o, n, lr_t, 0.9, 0.999, 1e-8,
                                         1.0, ops.current_stream_handle(self.session.device))
        _lib.check(rc, "adam_step")
        v = g.out[:2].tolist()
        return OrderedDict(loss=v[0], accuracy=v[1] / g.clips)

    def eval_step(self, batch=None, eps=None):
        g = self.primary
        self._feed(g, batch, eps)
        g.plan_eval.run()
        v = g.out[:2].tolist()
        return OrderedDict(loss=v[0], accuracy=v[1] / g.clips)
