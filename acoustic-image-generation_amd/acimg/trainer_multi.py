"""`Trainer` of trainer/trainermulti.py:14-96 for the joint-latent model (FLAGS.jointmvae without `fusion` /
`onlyaudiovideo`; main.py:190-201, 220-225): the encoders of three split VAEs -> the per-pixel fusion MLP `Jointmvae`
-> the three decoders on the MLP's heads; MSE + Huber per modality + KL (summed over the latent, /1e6) + the kernel
regularisers tf.losses.get_total_loss() collects; Adam over `modelassociator.train_vars` ONLY.

One recorded plan: encoders (forward only: nothing upstream of the MLP is trained), the three feature maps gathered
into the MLP's 776-wide input (133 | 512 | 128 channels in tf.concat order + 3 zero pads), MLP, decoders, losses,
decoder backward for DATA gradients (batch norm in training mode: its statistics depend on the features), MLP backward,
Adam over the MLP's contiguous range.  The video / audio models' moving averages are updated by the step as the
reference's update_ops do.
"""
from collections import OrderedDict

from . import _lib, ops
from .session import Session

_LATENT_W = 1e-6


class _Graph(object):
    pass


class TrainerMulti(object):

    def __init__(self, modelac, modelaudio, modelimages, modelassociator, modelassociator1=None, logger=None,
                 display_freq=1, learning_rate=0.0001, num_classes=14, num_epochs=1, nr_frames=12, temporal_pooling=False,
                 session=None):
        self.modelac, self.modelaudio, self.modelimages = modelac, modelaudio, modelimages
        self.modelassociator, self.modelassociator1 = modelassociator, modelassociator1
        self.logger = logger
        self.display_freq = display_freq
        self.learning_rate = learning_rate
        self.num_classes = num_classes
        self.num_epochs = num_epochs
        self.nr_frames = nr_frames
        self.temporal_pooling = temporal_pooling
        self.session = session
        self.global_step = 0
        self.noise_seed = 1239

    def _build_functions(self, data=None, batch_size=None):
        N = int(batch_size or getattr(data, "batch_size", None) or 2)
        if self.session is None:
            self.session = Session()
        sess = self.session
        z = sess.zeros
        mac, mau, mvi, ma = self.modelac, self.modelaudio, self.modelimages, self.modelassociator
        g = _Graph()
        g.N = N
        g.acoustic = z(N, mac.height, mac.width, mac.channels)
        g.mfcc = z(N, mau.height, mau.width, mau.channels)        # (the reference's name for the spectrogram input)
        g.video = z(N, mvi.height, mvi.width, mvi.channels)
        g.eps = OrderedDict(ac=z(N, mac.Z), audio=z(N, mau.Z), video=z(N, mvi.Z))
        # trainermulti.py:44-49: the three encoders
        mac._build_network(g.acoustic, session=sess, eps=g.eps["ac"])
        mvi._build_network(g.video, session=sess, eps=g.eps["video"])
        mau._build_network(g.mfcc, session=sess, eps=g.eps["audio"])
        order = (mac, mvi, mau)                      # tf.concat((inputac, inputvideo, inputaudio)), multimodal.py:306
        ctot = sum(m.FEAT_C for m in order)
        rows = N * 12 * 16
        g.concat = z(N, 12, 16, (ctot + 3) & ~3)
        ld = g.concat.shape[-1]
        p = sess.new_plan()
        for m in order:
            p.extend(m.plan_enc)
        off, views = 0, []
        for m in order:      # features are post-ReLU and dense per model: a strided copy into the slice
            ops.grad_slice(p, m.features._base if m.features._base is not None else m.features, m.FEAT_LD,
                           ops.Ptr(g.concat, off), ld, None, 0, rows, m.FEAT_C)
            views.append(g.concat[..., off:off + m.FEAT_C])
            off += m.FEAT_C
        ma._build_model(*views, session=sess)
        p.extend(ma.plan_fwd)
        # trainermulti.py:53-56: the decoders on the MLP's heads
        mac._build_model(ma.outputac)
        mau._build_model(ma.outputaudio)
        mvi._build_model(ma.outputvideo)
        g.mods = OrderedDict(ac=(mac, g.acoustic, "outputac"), audio=(mau, g.mfcc, "outputaudio"),
                             video=(mvi, g.video, "outputvideo"))
        g.sums, g.losses, g.g_logit = OrderedDict(), OrderedDict(), OrderedDict()
        for k, (m, target, _) in g.mods.items():
            p.extend(m.plan_fwd)
            g.sums[k], g.losses[k] = z(4), z(8)
            g.g_logit[k] = z(*m.yhat.t.shape)
            count = target.numel()                  # the mean's divisor
            padded = m.yhat.t.numel()               # floats the loss kernel walks: output and target padded to 4 channels
            ratio = float(padded) / float(count)    # (pad channels are 0 - 0), as acimg/trainer_vae.py
            ops.zero(p, g.sums[k])
            ops.recon_loss(p, m.yhat.t, m.xpad.t, g.g_logit[k], g.sums[k], padded, ratio, ratio)
            ops.loss_finalize(p, g.sums[k], m.kl, N, count, _LATENT_W, 0.0, 1.0, 1.0, g.losses[k])
        # kernel regularisers of the video / audio models (constants of this step: reported, no gradient wanted)
        st = sess.store
        g.reg = []
        for k, (m, _, _) in g.mods.items():
            for wd, roff, n in m.reg_ranges():
                buf = z(4)
                ops.zero(p, buf)
                ops.sumsq(p, ops.LazyPtr(lambda roff=roff, n=n: st.flat["train"][roff:roff + n]), n, buf)
                g.reg.append((wd, buf))
        g_heads = OrderedDict()
        for k, (m, _, attr) in g.mods.items():
            m.record_backward(p, g.g_logit[k], _LATENT_W / N)
            g_heads[attr] = (m.g_feat, m.feat_ld)
        ma.record_backward(p, g_heads, need_input_grad=False)
        g.plan_train = p
        sess.finalize()
        rng = [(n, o, c) for n, o, c in st.train_ranges() if n.startswith(ma.scope + "/")]
        g.off = rng[0][1]
        g.numel = rng[-1][1] + rng[-1][2] - g.off
        self.primary = g
        return g

    def train_step(self, batch=None, eps=None, apply=True):
        """batch: (acoustic [N,36,48,12], spectrogram [N,193,257,1], video [N,224,298,3]) or None to reuse the resident
        inputs; eps: {'ac' | 'audio' | 'video': [N, Z]} or None (device normal noise)"""
        g = self.primary
        if batch is not None:
            for dst, src in zip((g.acoustic, g.mfcc, g.video), batch):
                dst.copy_(src.reshape(dst.shape), non_blocking=True)
        if eps is not None:
            for k, v in eps.items():
                g.eps[k].copy_(v.reshape(g.eps[k].shape), non_blocking=True)
        else:
            self._noise_calls = getattr(self, "_noise_calls", 0) + 1
            for i, t in enumerate(g.eps.values()):
                rc = _lib.load().acimg_randn(t.data_ptr(), t.numel(), self.noise_seed, (self._noise_calls * 4 + i) << 24,
                                             ops.current_stream_handle(self.session.device))
                _lib.check(rc, "randn")
        g.plan_train.run()
        if apply:
            st = self.session.store
            self.global_step += 1
            lr_t = ops.adam_lr_t(self.learning_rate, self.global_step)
            o = g.off * 4
            rc = _lib.load().acimg_adam_step(st.flat["train"].data_ptr() + o, st.grad.data_ptr() + o,
                                             st.adam_m.data_ptr() + o, st.adam_v.data_ptr() + o, g.numel, lr_t, 0.9,
                                             0.999, 1e-8, 1.0, ops.current_stream_handle(self.session.device))
            _lib.check(rc, "adam_step")
        out = OrderedDict()
        mse = hub = lat = 0.0
        for k in g.mods:
            v = g.losses[k][:3].tolist()
            out["mse_" + k], out["huber_" + k] = v[0], v[1]
            mse, hub, lat = mse + v[0], hub + v[1], lat + v[2]
        reg = sum(0.5 * wd * float(buf[0]) for wd, buf in g.reg)
        out.update(mse=mse, huber=hub, latent=lat, reg=reg, loss=mse + hub + lat + reg)
        return out
