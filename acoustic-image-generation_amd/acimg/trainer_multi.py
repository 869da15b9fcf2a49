"""`Trainer` of trainer/trainermulti.py:14-129 for the joint-latent model (FLAGS.jointmvae; main.py:190-201, 220-225): the
encoders of three split VAEs -> the per-pixel fusion MLP `Jointmvae` -> the three decoders on the MLP's heads; MSE + Huber
per modality + KL (summed over the latent, /1e6) + the kernel regularisers tf.losses.get_total_loss() collects; Adam over
`modelassociator.train_vars` ONLY.  Round 4: the two other branches of `_build_functions`, chosen like the reference by
FLAGS.fusion / FLAGS.onlyaudiovideo (or the `mode` argument) -
  "fusion" (:50-51): the MLP is `JointTwomvae2` on (video, audio) only; decoders, losses and optimiser as above;
  "onlyaudiovideo" (:97-125): `modelassociator` (Jointmvae, FROZEN) on the three maps gives the acoustic feature target,
      `modelassociator1` (JointTwomvae) on (video, audio) feeds the ACOUSTIC decoder alone; loss = MSE(target feature,
      feature) + MSE + Huber + KL/1e6 + the encoders' kernel regularisers (the video / audio decoders are never built in that
      graph); Adam over `modelassociator1.train_vars`;
and FLAGS.moddrop (:46-47, `modDrop` :446-450): the acoustic feature map times ONE 0 / 1 draw per step (an explicit input
of `train_step`, like eps; drawn on the host otherwise).

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
                 session=None, mode=None, moddrop=None):
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
        from .flags import FLAGS
        if mode is None:
            mode = "fusion" if FLAGS.fusion else ("onlyaudiovideo" if FLAGS.onlyaudiovideo else "all")
        assert mode in ("all", "fusion", "onlyaudiovideo"), mode
        assert mode != "onlyaudiovideo" or modelassociator1 is not None, "onlyaudiovideo needs modelassociator1 (JointTwomvae)"
        self.mode = mode
        self.moddrop = bool(FLAGS.moddrop if moddrop is None else moddrop)

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
        order3 = (mac, mvi, mau)                     # tf.concat((inputac, inputvideo, inputaudio)), multimodal.py:306
        rows = N * 12 * 16
        p = sess.new_plan()
        mode = self.mode
        # encoders the graph really needs: the acoustic one feeds nothing in "fusion" (TF prunes it)
        for m in (order3 if mode != "fusion" else (mvi, mau)):
            p.extend(m.plan_enc)
        g.moddrop_mask = None
        if self.moddrop and mode != "fusion":
            g.moddrop_mask = z(rows, mac.FEAT_LD)
            g.moddrop_mask.fill_(1.0)

        def gather(models):
            """tf.concat of the models' feature maps: consecutive channel slices of one buffer (+ zero pad channels)"""
            ctot = sum(m.FEAT_C for m in models)
            buf = z(N, 12, 16, (ctot + 3) & ~3)
            ld = buf.shape[-1]
            off, views = 0, []
            for m in models:      # features are post-ReLU and dense per model: a strided copy into the slice
                drop = g.moddrop_mask if (m is mac and g.moddrop_mask is not None) else None
                ops.grad_slice(p, m.features._base if m.features._base is not None else m.features, m.FEAT_LD,
                               ops.Ptr(buf, off), ld, drop, 0 if drop is None else mac.FEAT_LD, rows, m.FEAT_C)
                views.append(buf[..., off:off + m.FEAT_C])
                off += m.FEAT_C
            return buf, views

        if mode == "fusion":
            g.concat, views = gather((mvi, mau))                 # JointTwomvae2(outputvideo, outputaudio), :50-51
            ma._build_model(*views, session=sess)
            trained = ma
        elif mode == "onlyaudiovideo":
            g.concat, views = gather(order3)                     # the frozen Jointmvae: feature target, forward only
            ma._build_model(*views, session=sess)
            g.concat1, views1 = gather((mvi, mau))
            ma1 = self.modelassociator1
            ma1._build_model(*views1, session=sess)              # :98
            trained = ma1
        else:
            g.concat, views = gather(order3)
            ma._build_model(*views, session=sess)
            trained = ma
        p.extend(ma.plan_fwd)
        if trained is not ma:
            p.extend(trained.plan_fwd)
        # trainermulti.py:53-56 / :99: the decoders on the trained MLP's heads
        mac._build_model(trained.outputac)
        g.mods = OrderedDict(ac=(mac, g.acoustic, "outputac"))
        if mode != "onlyaudiovideo":
            mau._build_model(trained.outputaudio)
            mvi._build_model(trained.outputvideo)
            g.mods["audio"] = (mau, g.mfcc, "outputaudio")
            g.mods["video"] = (mvi, g.video, "outputvideo")
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
        # kernel regularisers of the video / audio models (constants of this step: reported, no gradient wanted); in
        # "onlyaudiovideo" only their encoder halves exist in the reference's graph
        st = sess.store
        g.reg = []
        for m in (mac, mau, mvi):
            ranges = m.reg_ranges()
            if mode == "onlyaudiovideo":
                ranges = ranges[:1]
            for wd, roff, n in ranges:
                buf = z(4)
                ops.zero(p, buf)
                ops.sumsq(p, ops.LazyPtr(lambda roff=roff, n=n: st.flat["train"][roff:roff + n]), n, buf)
                g.reg.append((wd, buf))
        g_heads = OrderedDict()
        for k, (m, _, attr) in g.mods.items():
            m.record_backward(p, g.g_logit[k], _LATENT_W / N)
            g_heads[attr] = (m.g_feat, m.feat_ld)
        g.feat_sum = None
        if mode == "onlyaudiovideo":
            # l2_feature = tf.losses.mean_squared_error(modelassociator.outputac, modelassociator1.outputac) (:100): the mean
            # over N*12*16*133 elements; its gradient w.r.t. the trained head joins the decoder's (both [rows][136], pad
            # channels zero on both sides)
            a1, a0 = trained.heads["outputac"], ma.heads["outputac"]
            n_el, count = a1.numel(), rows * 133
            g.feat_diff, g.feat_sum = z(*a1.shape), z(4)
            ops.zero(p, g.feat_diff)
            ops.axpy(p, 1.0, a1, g.feat_diff, n_el)
            ops.axpy(p, -1.0, a0, g.feat_diff, n_el)
            ops.zero(p, g.feat_sum)
            ops.sumsq(p, g.feat_diff, n_el, g.feat_sum)
            g.feat_count = count
            assert mac.feat_ld == a1.shape[-1]
            ops.axpy(p, 2.0 / count, g.feat_diff, mac.g_feat, n_el)
        trained.record_backward(p, g_heads, need_input_grad=False)
        ma = trained
        g.plan_train = p
        sess.finalize()
        rng = [(n, o, c) for n, o, c in st.train_ranges() if n.startswith(ma.scope + "/")]
        g.off = rng[0][1]
        g.numel = rng[-1][1] + rng[-1][2] - g.off
        self.primary = g
        return g

    def train_step(self, batch=None, eps=None, apply=True, moddrop_on=None):
        """batch: (acoustic [N,36,48,12], spectrogram [N,193,257,1], video [N,224,298,3]) or None to reuse the resident
        inputs; eps: {'ac' | 'audio' | 'video': [N, Z]} or None (device normal noise); moddrop_on (FLAGS.moddrop): the
        step's 0 / 1 draw (None: drawn here, 1 with probability 0.2 as `modDrop(p_mod=0.2)` is written, :47,447)"""
        g = self.primary
        if g.moddrop_mask is not None:
            if moddrop_on is None:
                import random
                moddrop_on = 1.0 if random.random() - 0.2 < 0 else 0.0
            g.moddrop_mask.fill_(float(moddrop_on))
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
        feat = 0.0
        if g.feat_sum is not None:
            feat = float(g.feat_sum[0]) / g.feat_count
            out["feature"] = feat
        out.update(mse=mse, huber=hub, latent=lat, reg=reg, loss=mse + hub + lat + reg + feat)
        return out
