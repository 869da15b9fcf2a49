"""`TrainerAssociator`: trains a latent associator against the frozen `unet_z` decoder, MI355X-native — the
single-associator step of trainer/trainer_proietta.py:104-146: the encoder VAE's statistics go through the
associator, the acoustic-image decoder reconstructs from z = mean + std * eps, loss = MSE + Huber +
mean_b(0.5 * sum_j(mu^2 + s^2 - log(1e-8 + s^2) - 1)) / 1e6, Adam on the associator's variables.  The encoder
statistics arrive as a device buffer [N, 2*DIN] (any encoder's fused [mean | std] head output); the CONV associator
`AssociatorAudio` (models/multimodal.py:139-285) takes the spectrogram [N,193,257,1] itself, runs its batch norms in
training mode, and adds the l2_regularizer(8e-5) terms of its conv_conv_pool kernels, which
tf.losses.get_total_loss() collects, to the loss.

One step = ONE recorded plan: associator forward -> decoder forward -> reconstruction loss (+ gradient) ->
decoder data gradients down to d loss / d (mean, std) (KL term included) -> associator backward; then Adam over
the associator's contiguous range of the flat parameter buffer.
"""
from collections import OrderedDict

from . import _lib, ops
from .session import Session

_LATENT_W = 1e-6


class _Graph(object):
    pass


class TrainerAssociator(object):

    def __init__(self, modelassociator, modelac, display_freq=1, learning_rate=0.0001, num_epochs=1, session=None):
        self.modelassociator = modelassociator
        self.modelac = modelac
        self.display_freq = display_freq
        self.learning_rate = learning_rate
        self.num_epochs = num_epochs
        self.session = session
        self.global_step = 0
        self.noise_seed = 1237

    def _build_functions(self, data=None, batch_size=None):
        N = int(batch_size or getattr(data, "batch_size", None) or 8)
        if self.session is None:
            self.session = Session()
        sess = self.session
        z = sess.zeros
        ma, md = self.modelassociator, self.modelac
        g = _Graph()
        g.N = N
        conv = bool(getattr(ma, "IMAGE_INPUT", False))
        g.acoustic = z(N, 36, 48, 12)
        g.eps = z(N, 150)
        if conv:
            g.stats = z(N, ma.height, ma.width, ma.channels)      # the associator's own input (spectrogram)
            ma._build_model(g.stats, session=sess)
        else:
            g.stats = z(N, 2 * ma.DIN)        # [mean | std] of the encoder VAE
            ma._build_model(g.stats[:, :ma.DIN], g.stats[:, ma.DIN:], session=sess)
        md._build_model(g.acoustic, ma.mean, ma.std, session=sess, eps=g.eps)
        g.sums, g.losses = z(4), z(8)
        g.g_logit = z(N, 36, 48, 12)
        count = N * 36 * 48 * 12
        p = sess.new_plan()
        ops.zero(p, g.sums)
        p.extend(ma.plan_fwd)
        p.extend(md.plan_fwd)
        ops.recon_loss(p, md.yhat.t, g.acoustic, g.g_logit, g.sums, count, 1.0, 1.0)
        st = sess.store
        wd, nreg = 0.0, 0
        if conv:                                  # kernel regularisers of the conv associator: one pass each way
            roff, nreg = ma.reg_range()
            wd = ma.WD
            wreg = ops.LazyPtr(lambda: st.flat["train"][roff:roff + nreg])
            greg = ops.LazyPtr(lambda: st.grad[roff:roff + nreg])
            ops.sumsq(p, wreg, nreg, ops.Ptr(g.sums, 2))
        ops.loss_finalize(p, g.sums, md.kl, N, count, _LATENT_W, 0.5 * wd, 1.0, 1.0, g.losses)
        md.record_backward(p, g.g_logit, _LATENT_W / N)
        ma.record_backward(p, md.g_ext)
        if nreg:
            ops.axpy(p, wd, wreg, greg, nreg)
        g.plan_train = p
        sess.finalize()
        rng = [(n, o, c) for n, o, c in sess.store.train_ranges() if n.startswith(ma.scope + "/")]
        g.off = rng[0][1]
        g.numel = rng[-1][1] + rng[-1][2] - g.off
        self.primary = g
        return g

    def train_step(self, batch=None, eps=None, apply=True):
        """batch: (stats [N, 2*DIN], acoustic [N,36,48,12]) or None to reuse the resident inputs"""
        g = self.primary
        if batch is not None:
            g.stats.copy_(batch[0].reshape(g.stats.shape), non_blocking=True)
            g.acoustic.copy_(batch[1].reshape(g.acoustic.shape), non_blocking=True)
        if eps is not None:
            g.eps.copy_(eps.reshape(g.eps.shape), non_blocking=True)
        else:
            self._noise_calls = getattr(self, "_noise_calls", 0) + 1
            rc = _lib.load().acimg_randn(g.eps.data_ptr(), g.eps.numel(), self.noise_seed, self._noise_calls * 65536,
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
        v = g.losses[:5].tolist()
        return OrderedDict(mse=v[0], huber=v[1], latent=v[2], reg=v[3], loss=v[4])
