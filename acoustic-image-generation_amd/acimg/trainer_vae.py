"""`TrainerVAE`: train / eval step of the single-modality U-Net VAEs (`UNet` on RGB frames, `UNetSound` on STFT
spectrograms), MI355X-native.  Mirrors trainer/trainer.py:16-87 (ctor, `_build_functions`): the model
reconstructs its own input,
    loss = MSE + Huber + sum_k wd*||W_k||^2/2 + mean_b(0.5*mean_j(mu^2 + s^2 - log(1e-8 + s^2) - 1)) / 1e6
(`tf.losses.get_total_loss()` = both reconstruction losses + the kernel regularisers; :58-73), Adam on every
trainable variable of the model with the BN moving-average updates as control dependencies (:81-87).
`train_step` is the body of the reference's loop (one session.run of the train op).

One step = ONE recorded plan: zero sums -> forward (BN batch statistics, moving averages advance) ->
reconstruction loss + its gradient -> L2 term (one pass over the contiguous regularised kernels) -> scalars
-> backward -> L2 gradient (one axpy) ; then Adam (one launch over the flat parameter buffer).
"""
from collections import OrderedDict

from . import _lib, ops
from .session import Session


class _Graph(object):
    pass


class TrainerVAE(object):

    def __init__(self, model, display_freq=1, learning_rate=0.0001, num_epochs=1, session=None):
        self.model = model
        self.display_freq = display_freq
        self.learning_rate = learning_rate
        self.num_epochs = num_epochs
        self.session = session
        self.global_step = 0
        self.noise_seed = 1237
        self.log = print

    def _build_functions(self, data=None, batch_size=None):
        N = int(batch_size or getattr(data, "batch_size", None) or 8)
        if self.session is None:
            self.session = Session()
        sess = self.session
        m = self.model
        z = sess.zeros
        g = _Graph()
        g.N = N
        H, W, Cin = m.height, m.width, m.channels
        g.images = z(N, H, W, Cin)
        g.eps = z(N, m.Z)
        m._build_model(g.images, session=sess, eps=g.eps)
        cp = m.xpad.Cp
        g.sums = z(4)
        g.losses = z(8)
        g.g_logit = z(N, H, W, cp)
        padded = N * H * W * cp            # floats the loss kernel walks (pad channels are 0 - 0)
        count = N * H * W * Cin            # the mean's divisor
        ratio = float(padded) / float(count)
        st = sess.store
        off, n = m.reg_range()
        wreg = ops.LazyPtr(lambda: st.flat["train"][off:off + n])
        greg = ops.LazyPtr(lambda: st.grad[off:off + n])
        latent_w = 1.0 / (1e6 * m.Z)       # kl[n] holds 0.5 * sum_j; the reference takes mean_j, then / 1e6

        p = sess.new_plan()
        ops.zero(p, g.sums)
        p.extend(m.plan_fwd)
        ops.recon_loss(p, m.yhat.t, m.xpad.t, g.g_logit, g.sums, padded, ratio, ratio)
        if n > 0:
            ops.sumsq(p, wreg, n, ops.Ptr(g.sums, 2))
        ops.loss_finalize(p, g.sums, m.kl, N, count, latent_w, 0.5 * m.WD, 1.0, 1.0, g.losses)
        m.record_backward(p, g.g_logit, latent_w / N)
        if n > 0:
            ops.axpy(p, m.WD, wreg, greg, n)
        g.plan_train = p
        sess.finalize()
        self.primary = g
        return g

    def _noise(self, g, eps):
        if eps is not None:
            g.eps.copy_(eps.reshape(g.N, -1), non_blocking=True)
        else:
            self._noise_calls = getattr(self, "_noise_calls", 0) + 1
            rc = _lib.load().acimg_randn(g.eps.data_ptr(), g.eps.numel(), self.noise_seed, self._noise_calls * 65536,
                                         ops.current_stream_handle(self.session.device))
            _lib.check(rc, "randn")

    def train_step(self, batch=None, eps=None, sync=True, apply=True):
        """batch: images [N,H,W,C] (or None to reuse the resident input); returns {mse, huber, latent, reg, loss}"""
        g = self.primary
        if batch is not None:
            g.images.copy_(batch.reshape(g.images.shape), non_blocking=True)
        self._noise(g, eps)
        g.plan_train.run()
        if apply:
            store = self.session.store
            self.global_step += 1
            lr_t = ops.adam_lr_t(self.learning_rate, self.global_step)
            rc = _lib.load().acimg_adam_step(store.flat["train"].data_ptr(), store.grad.data_ptr(),
                                             store.adam_m.data_ptr(), store.adam_v.data_ptr(), store.train_numel(),
                                             lr_t, 0.9, 0.999, 1e-8, 1.0, ops.current_stream_handle(self.session.device))
            _lib.check(rc, "adam_step")
        if not sync:
            return g.losses
        v = g.losses[:5].tolist()
        return OrderedDict(mse=v[0], huber=v[1], latent=v[2], reg=v[3], loss=v[4])
