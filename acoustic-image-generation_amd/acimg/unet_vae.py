"""`UNet` (RGB frames, scope 'UNet') and `UNetSound` (STFT spectrograms, scope 'UNetAudio'): the
conv-BN-ReLU U-Net VAEs of models/unet_architecture.py:46-206 and models/unet_sound.py:49-208, MI355X-native.

Both reference classes are the same code with a different layer table, so one recorder serves both.  Model
protocol as in the reference: `scope`, `init_model(session, checkpoint_file)`, `_build_model(images)` setting
`mean`, `variance`, `output`, `network`, `train_vars`.

Every layer here is small-channel (8..128) at high resolution (up to 224x298), i.e. HBM-bound: the design
minimises passes.  Per conv-BN-ReLU layer, forward: ONE implicit-GEMM launch (conv + bias, raw output, BN
statistics partials in its epilogue) -> bn_finalize -> ONE normalise+ReLU pass that writes straight into the
consumer's buffer (skip tensors land in their slice of the decoder's concat buffer, tf.concat never runs).
Backward: bn_bwd (ordered partial sums, ReLU mask recomputed from the raw output instead of re-reading the
activation) -> weight gradient (+ fused bias gradient) -> data gradient, with the skip branch's gradient added
in the strided conv's data-gradient epilogue (`residual`).  The strided "pool" convs' data gradients and the
overlapping transposed convs use the zero-insertion forms of acimg_conv2d_dgrad / acimg_deconv_fwd.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID
from .params import FusedHeads, Var, up4
from .session import get_default_session
from .unet_acresnet import Act
from .vision import load_state_file

BN_MOMENTUM, BN_EPS = 0.99, 1e-3   # tf.layers.batch_normalization defaults


class _CBR(object):
    """one conv + BN + ReLU layer: descriptor, buffers, variable names"""
    deferred = False

    def relu_output(self):
        """the layer's normalised, rectified output [N, H, W, C]: the buffer the normalise + ReLU pass wrote, or - when the
        layer's consumers form it on load and no pass ever wrote it (`deferred`) - the same values from the raw output"""
        y = self.y
        v = y.t.view(y.N, y.H, y.W, -1)[..., y.off:y.off + y.C]
        if self.deferred:
            v = torch.relu(v * self.scale[:y.C] + self.shift[:y.C])
        return v


class UNetVAE(object):
    SCOPE = None
    CIN = None
    WD = None
    Z = 128
    HEAD = None           # VALID kernel of the mean / variance heads = size of the bottleneck map
    # (layer, filters, pool kernel (kh, kw) or None, pool padding)
    ENC = None
    # (layer, upsample filters, upsample kernel (kh, kw), skip layer)
    DEC = None
    COUT = None
    HEAD_NAMES = ("mean", "variance")
    ENCODER_ONLY = False

    def __init__(self, input_shape=None, precision="split", defer_bn=True):
        """precision: "split" = layers with >= 32 channels on both sides run on the split-MFMA kernels (forward
        f16x3, gradients bf16x3: fp32-class results at 5x the fp32-MFMA rate), "f32" = exact-f32 MFMA everywhere,
        "bf16" = BASELINE configs[1]: the same layers with both GEMM operands ROUNDED to bf16 (one MFMA per product,
        fp32 accumulation; tensors, statistics, losses and Adam stay fp32) in forward, data and weight gradients"""
        assert precision in ("split", "f32", "bf16")
        self.precision = precision
        self._bf16 = precision == "bf16"
        # a conv-BN-ReLU layer whose only consumer is the next 3x3 conv (conv_1 of a block, a "pool" conv) skips its normalise +
        # ReLU pass when that conv and its weight gradient apply the affine while they stage their tiles
        # (ops.conv2d_affine_input_ok: the halo kernels); False = every layer materialises its output
        self.defer_bn = bool(defer_bn)
        self.scope = self.SCOPE
        self.height, self.width, self.channels = input_shape
        assert self.channels == self.CIN
        self.session = None
        self._wsplit_bufs = {}

    def _use_split(self, d):
        return (self.precision in ("split", "bf16") and d.stride == 1 and d.C % 32 == 0 and d.K % 32 == 0 and
                d.N * d.OH * d.OW >= 16384)

    def _begin_prepares(self, plan):
        """the kernels change every step: TWO launches at the head of the forward plan re-split them all (16 jobs each at most:
        the forward images, and - added while the backward is recorded - the flipped / transposed data-gradient images)
        instead of one ~5 us launch in front of every conv"""
        self._prep_fwd = self._prep_bwd = None
        if self.precision in ("split", "bf16"):
            self._prep_fwd, self._prep_bwd = ops.PrepareJobs(), ops.PrepareJobs()
            ops.conv2d_split3_prepare_multi(plan, self._prep_fwd)
            ops.conv2d_split3_prepare_multi(plan, self._prep_bwd)

    def _prec(self, d):
        """precision code of a layer's entry points (include/acimg.h): 0 fp32-class, 1 split3, 2 bf16"""
        return (2 if self._bf16 else 1) if self._use_split(d) else 0

    def _next_takes_affine(self, N, h, w, cin, K):
        """does a 3x3 / stride-1 / SAME conv-BN-ReLU layer [N, h, w, cin] -> K read a RAW producer output through the
        producer's batch-norm affine (forward and weight gradient both)?"""
        if not self.defer_bn:
            return False
        d = ops.conv_desc(N, h, w, up4(cin), K, 3, 3, 1, "SAME", ldx=up4(cin), ldy=up4(K), ldw=up4(K), act=ACT_NONE)
        return ops.conv2d_affine_input_ok(d, self._prec(d))

    def _wsplit(self, name, nbytes, kind):
        key = (name, kind)
        if key not in self._wsplit_bufs:
            self._wsplit_bufs[key] = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.session.device)
        return self._wsplit_bufs[key]

    # ---- variables --------------------------------------------------------------------------------------
    def _layer_table(self):
        """[(kind, name, tf kernel shape)] in forward order; kind: cbr | head | dense | conv | deconv"""
        t = []
        cin = self.CIN
        widths = {}
        for name, F_, pool, _ in self.ENC:
            t.append(("cbr", "layer%s/conv_1" % name, "layer%s/bn_1" % name, (3, 3, cin, F_)))
            t.append(("cbr", "layer%s/conv_2" % name, "layer%s/bn_2" % name, (3, 3, F_, F_)))
            if pool is not None:
                t.append(("cbr", "layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, (pool[0], pool[1], F_, F_)))
            widths[name] = F_
            cin = F_
        t.append(("heads", None, None, (self.HEAD[0], self.HEAD[1], cin, self.Z)))
        if self.ENCODER_ONLY:
            return t
        t.append(("dense", "dense", None, (self.Z, self.HEAD[0] * self.HEAD[1])))
        t.append(("conv", "conv2d", None, (3, 3, 1, 128)))
        cin = 128
        for name, F_, k, skip in self.DEC:
            t.append(("deconv", "upsample_%s" % name, None, (k[0], k[1], F_, cin)))
            t.append(("cbr", "layer%s/conv_1" % name, "layer%s/bn_1" % name, (3, 3, F_ + widths[skip], F_)))
            t.append(("cbr", "layer%s/conv_2" % name, "layer%s/bn_2" % name, (3, 3, F_, F_)))
            cin = F_
        t.append(("conv", "final", None, (1, 1, cin, self.COUT)))
        return t

    def _register(self, store):
        """The kernels that carry kernel_regularizer (conv_conv_pool + upconv_2D, unet_architecture.py:159,
        172,203) are registered first, contiguously: their L2 term and its gradient are one pass each."""
        s = self.scope
        table = self._layer_table()
        first = None
        for kind, name, bn, shape in table:
            if kind == "cbr" or kind == "deconv":
                v = store.add(Var("%s/%s/kernel" % (s, name), shape, "conv" if kind == "cbr" else "deconv", "train"))
                first = first or v
                last = v
        self._reg_range = (first, last)
        for kind, name, bn, shape in table:
            if kind == "cbr":
                store.add(Var("%s/%s/bias" % (s, name), (shape[3],), "vec", "train"))
                store.add(Var("%s/%s/gamma" % (s, bn), (shape[3],), "vec", "train"))
                store.add(Var("%s/%s/beta" % (s, bn), (shape[3],), "vec", "train"))
                store.add(Var("%s/%s/moving_mean" % (s, bn), (shape[3],), "vec", "state"))
                store.add(Var("%s/%s/moving_variance" % (s, bn), (shape[3],), "vec", "state"))
            elif kind == "deconv":
                store.add(Var("%s/%s/bias" % (s, name), (shape[2],), "vec", "train"))
            elif kind == "heads":
                self.heads = FusedHeads(s, shape[2], self.Z, True, hw=self.HEAD, names=self.HEAD_NAMES)
                store.add_fused(self.heads)
            elif kind == "dense":
                store.add(Var("%s/dense/kernel" % s, shape, "dense", "train"))
                store.add(Var("%s/dense/bias" % s, (shape[1],), "vec", "train"))
            else:
                store.add(Var("%s/%s/kernel" % (s, name), shape, "conv", "train"))
                store.add(Var("%s/%s/bias" % (s, name), (shape[3],), "vec", "train"))

    def reg_range(self):
        """(offset, numel) of the regularised kernels inside the flat trainable buffer"""
        a, b = self._reg_range
        return a.offset, b.offset + b.numel - a.offset

    def init_model(self, session, checkpoint_file):
        """models/unet_architecture.py:31-44: restore every model variable of the scope from a TF-named state"""
        state = load_state_file(checkpoint_file)
        store = (session or self.session).store
        return store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def initialize(self, seed=1240, state=None):
        """xavier_initializer() / Glorot-uniform kernels, zero biases, gamma 1, beta 0, moving mean 0 / variance 1"""
        if state is None:
            g = torch.Generator().manual_seed(seed)
            state = OrderedDict()
            s = self.scope

            def xav(shape, fin, fout):
                lim = np.sqrt(6.0 / (fin + fout))
                return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()

            for kind, name, bn, shape in self._layer_table():
                if kind == "cbr":
                    kh, kw, cin, cout = shape
                    state["%s/%s/kernel" % (s, name)] = xav(shape, kh * kw * cin, kh * kw * cout)
                    state["%s/%s/bias" % (s, name)] = torch.zeros(cout)
                    state["%s/%s/gamma" % (s, bn)] = torch.ones(cout)
                    state["%s/%s/beta" % (s, bn)] = torch.zeros(cout)
                    state["%s/%s/moving_mean" % (s, bn)] = torch.zeros(cout)
                    state["%s/%s/moving_variance" % (s, bn)] = torch.ones(cout)
                elif kind == "heads":
                    kh, kw, cin, cout = shape
                    for h in self.HEAD_NAMES:
                        state["%s/%s/kernel" % (s, h)] = xav(shape, kh * kw * cin, kh * kw * cout)
                        state["%s/%s/bias" % (s, h)] = torch.zeros(cout)
                elif kind == "dense":
                    state[s + "/dense/kernel"] = xav(shape, shape[0], shape[1])
                    state[s + "/dense/bias"] = torch.zeros(shape[1])
                elif kind == "deconv":
                    kh, kw, cout, cin = shape
                    state["%s/%s/kernel" % (s, name)] = xav(shape, kh * kw * cin, kh * kw * cout)
                    state["%s/%s/bias" % (s, name)] = torch.zeros(cout)
                else:
                    kh, kw, cin, cout = shape
                    state["%s/%s/kernel" % (s, name)] = xav(shape, kh * kw * cin, kh * kw * cout)
                    state["%s/%s/bias" % (s, name)] = torch.zeros(cout)
        self.session.store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    # ---- pointers ---------------------------------------------------------------------------------------
    def _P(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.p(self.scope + "/" + name))

    def _G(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.g(self.scope + "/" + name))

    # ---- graph ------------------------------------------------------------------------------------------
    def _build_model(self, images, session=None, eps=None, training=True):
        """images: device buffer [N,H,W,cin]; eps: device buffer [N,Z] standing where the reference samples
        tf.random_normal (models/unet_architecture.py:66)."""
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        N = images.shape[0]
        assert tuple(images.shape[1:]) == (self.height, self.width, self.channels)
        self.N = N
        self.training = training
        z = sess.zeros
        H, W = self.height, self.width
        cp = up4(self.CIN)
        self.images = images
        self.xpad = Act(z(N, H, W, cp), N, H, W, self.CIN)
        self.eps = eps if eps is not None else z(N, self.Z)

        # geometry of the encoder, then the decoder's concat buffers (skip tensors live in their slices)
        sizes = {}
        h, w = H, W
        for name, F_, pool, pad in self.ENC:
            sizes[name] = (h, w, F_)
            if pool is not None:
                if pad == "SAME":
                    h, w = -(-h // 2), -(-w // 2)
                else:
                    h, w = (h - pool[0]) // 2 + 1, (w - pool[1]) // 2 + 1
        assert (h, w) == tuple(self.HEAD), "input %dx%d does not reduce to the %s head" % (H, W, self.HEAD)
        self.cat = {}
        for name, F_, k, skip in self.DEC:
            sh, sw, sf = sizes[skip]
            self.cat[skip] = (z(N, sh, sw, F_ + sf), F_, sf)

        self.layers = OrderedDict()
        self.plan_fwd = sess.new_plan()
        self._begin_prepares(self.plan_fwd)
        self._record_forward(self.plan_fwd, sizes)

        Zn = self.Z
        self.mean = self.heads_out[:, :Zn]
        self.variance = self.heads_out[:, Zn:2 * Zn]
        self.std = self.variance            # trainer/trainer.py:61 reads `model.std`
        self.output = self.yhat.t
        self.network = OrderedDict(input=images, is_training=None, keep_prob=None, features=self.conv5.t)
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/") and
                           not n.endswith(("moving_mean", "moving_variance"))]

    def _cbr(self, plan, name, bn, x, K, R, S, stride, padding, out, defer=False):
        """conv + bias -> raw (+ BN statistics) ; bn_finalize ; out = relu(raw*scale + shift).
        defer: no normalise + ReLU pass - the layer's output IS its raw tensor with `.affine = (scale, shift)` attached, and
        the one consumer applies it on load; an input `x` that carries `.affine` is read that way here"""
        z = self.session.zeros
        L = _CBR()
        L.name, L.bn, L.x, L.y = name, bn, x, out
        aff = getattr(x, "affine", None)
        in_scale, in_shift = aff if aff is not None else (None, None)
        L.d = ops.conv_desc(x.N, x.H, x.W, x.Cp if x.off == 0 and x.ld == x.Cp else x.C, K, R, S, stride, padding,
                            ldx=x.ld, ldy=up4(K), ldw=up4(K), act=ACT_NONE)
        assert (L.d.OH, L.d.OW) == (out.H, out.W), (name, L.d.OH, L.d.OW, out.H, out.W)
        L.raw = Act(z(x.N, out.H, out.W, up4(K)), x.N, out.H, out.W, K)
        kp = up4(K)
        L.rows = ops.conv2d_stats_rows(L.d)
        L.stats = z(L.rows, 2, kp)
        L.scale, L.shift, L.save_mean, L.save_invstd = z(kp), z(kp), z(kp), z(kp)
        if self._use_split(L.d):
            L.rows = ops.conv2d_fwd_split3_stats_rows(L.d)
            L.stats = z(L.rows, 2, kp)
            ws = self._wsplit(name, ops.conv2d_split3_weight_bytes(L.d), "fwd")
            jobs = getattr(self, "_prep_fwd", None)
            if jobs is not None and len(jobs.jobs) < 16:
                jobs.add(L.d, self._P(name + "/kernel"), ws, 2 if self._bf16 else 0)
            else:
                ops.conv2d_split3_prepare(plan, L.d, self._P(name + "/kernel"), ws, bf16=self._bf16)
            ops.conv2d_fwd_split3(plan, L.d, x.ptr, ws, L.raw.ptr, in_scale=in_scale, in_shift=in_shift,
                                  in_relu=1 if aff is not None else 0, stats=L.stats if self.training else None,
                                  bias=self._P(name + "/bias"), bf16=self._bf16)
        else:
            ops.conv2d_fwd(plan, L.d, x.ptr, self._P(name + "/kernel"), self._P(name + "/bias"), L.raw.ptr,
                           in_scale=in_scale, in_shift=in_shift, in_relu=1 if aff is not None else 0,
                           stats=L.stats if self.training else None)
        ops.bn_finalize(plan, L.stats if self.training else None, L.rows if self.training else 0, K, kp,
                        out.pixels if self.training else 0, self._P(bn + "/gamma"), self._P(bn + "/beta"),
                        self._P(bn + "/moving_mean"), self._P(bn + "/moving_variance"), L.scale, L.shift,
                        BN_MOMENTUM, BN_EPS, self.training, L.save_mean, L.save_invstd)
        if defer:
            L.deferred = True
            L.y = Act(L.raw.t, x.N, out.H, out.W, K)
            L.y.affine = (L.scale, L.shift)
        else:
            ops.bn_relu(plan, L.raw.ptr, L.scale, L.shift, out.ptr, out.pixels, K, kp, out.ld)
        self.layers[name] = L
        return L

    def _record_forward(self, plan, sizes):
        net = self._record_encoder(plan, sizes)
        self._record_latent_and_decoder(plan, sizes, net)

    def _record_encoder(self, plan, sizes):
        """input padding, the conv-conv-pool encoder (skip tensors into their concat slices) and the fused heads GEMM:
        leaves `conv5` and `heads_out` [N, 2Z] = [head 0 | head 1]"""
        N = self.N
        z = self.session.zeros
        H, W = self.height, self.width
        ops.pad_channels(plan, self.images, self.xpad.t, N * H * W, self.CIN, self.xpad.Cp)
        net = self.xpad
        self.skips = {}
        for li, (name, F_, pool, pad) in enumerate(self.ENC):
            h, w, _ = sizes[name]
            if self._next_takes_affine(N, h, w, F_, F_):       # conv_2 reads conv_1's raw output through its batch norm
                mid = self._cbr(plan, "layer%s/conv_1" % name, "layer%s/bn_1" % name, net, F_, 3, 3, 1, "SAME",
                                Act(None, N, h, w, F_), defer=True).y
            else:
                mid = Act(z(N, h, w, F_), N, h, w, F_)
                self._cbr(plan, "layer%s/conv_1" % name, "layer%s/bn_1" % name, net, F_, 3, 3, 1, "SAME", mid)
            if name in self.cat:    # the skip tensor is written into its slice of the decoder concat buffer
                buf, fup, fs = self.cat[name]
                out = Act(buf, N, h, w, F_, fup + fs, fup)
            else:
                out = Act(z(N, h, w, F_), N, h, w, F_)
            self._cbr(plan, "layer%s/conv_2" % name, "layer%s/bn_2" % name, mid, F_, 3, 3, 1, "SAME", out)
            self.skips[name] = out
            net = out
            if pool is not None:
                if pad == "SAME":
                    ph, pw = -(-h // 2), -(-w // 2)
                else:
                    ph, pw = (h - pool[0]) // 2 + 1, (w - pool[1]) // 2 + 1
                nxt = self.ENC[li + 1][1] if li + 1 < len(self.ENC) else None
                if nxt is not None and self._next_takes_affine(N, ph, pw, F_, nxt):   # the next block's conv_1 is its only reader
                    po = self._cbr(plan, "layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, net, F_, pool[0], pool[1], 2,
                                   pad, Act(None, N, ph, pw, F_), defer=True).y
                else:
                    po = Act(z(N, ph, pw, F_), N, ph, pw, F_)
                    self._cbr(plan, "layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, net, F_, pool[0], pool[1], 2,
                              pad, po)
                net = po
        self.conv5 = net
        hh, hw = self.HEAD
        Zn = self.Z
        C5 = net.C
        # mean | variance heads: one [N, hh*hw*C5] x [., 2Z] GEMM (a VALID conv over the whole map)
        kin = hh * hw * up4(C5)
        self.heads_out = z(N, 2 * Zn)
        self.d_heads = ops.conv_desc(N, 1, 1, kin, 2 * Zn, 1, 1, 1, "VALID", ldx=kin, ldy=2 * Zn, ldw=2 * Zn)
        ops.conv2d_fwd(plan, self.d_heads, net.t, self._P("heads/kernel"), self._P("heads/bias"), self.heads_out)
        return net

    def _record_latent_and_decoder(self, plan, sizes, net):
        N = self.N
        z = self.session.zeros
        H, W = self.height, self.width
        hh, hw = self.HEAD
        Zn = self.Z
        self.zbuf = z(N, Zn)
        self.kl = z(N)
        ops.latent_linear_fwd(plan, self.heads_out, self.eps, self.zbuf, Zn, self.kl, N, Zn)
        nd = hh * hw
        self.dns1 = z(N, nd)                       # dense output = the [N,hh,hw,1] map
        self.d_dense = ops.conv_desc(N, 1, 1, Zn, nd, 1, 1, 1, "VALID", ldx=Zn, ldy=nd, ldw=up4(nd), act=ACT_RELU)
        ops.conv2d_fwd(plan, self.d_dense, self.zbuf, self._P("dense/kernel"), self._P("dense/bias"), self.dns1)
        self.dns = Act(z(N, hh, hw, 4), N, hh, hw, 1)
        ops.pad_channels(plan, self.dns1, self.dns.t, N * nd, 1, 4)
        self.c2d = Act(z(N, hh, hw, 128), N, hh, hw, 128)
        self.d_c2d = ops.conv_desc(N, hh, hw, 4, 128, 3, 3, 1, "SAME", ldx=4, ldy=128, ldw=128, act=ACT_RELU)
        ops.conv2d_fwd(plan, self.d_c2d, self.dns.ptr, self._P("conv2d/kernel"), self._P("conv2d/bias"), self.c2d.ptr)
        net = self.c2d
        self.ups = OrderedDict()
        for name, F_, k, skip in self.DEC:
            buf, fup, fs = self.cat[skip]
            sh, sw, _ = sizes[skip]
            up = Act(buf, N, sh, sw, F_, fup + fs, 0)
            d = ops.deconv_desc(N, net.H, net.W, net.Cp, F_, k[0], k[1], 2, ldx=net.ld, ldy=up.ld, ldw=net.Cp)
            assert (d.OH, d.OW) == (sh, sw), (name, d.OH, d.OW, sh, sw)
            ops.deconv_fwd(plan, d, net.ptr, self._P("upsample_%s/kernel" % name), self._P("upsample_%s/bias" % name),
                           up.ptr)
            self.ups[name] = (d, net, up)
            catin = Act(buf, N, sh, sw, fup + fs)
            if self._next_takes_affine(N, sh, sw, F_, F_):
                mid = self._cbr(plan, "layer%s/conv_1" % name, "layer%s/bn_1" % name, catin, F_, 3, 3, 1, "SAME",
                                Act(None, N, sh, sw, F_), defer=True).y
            else:
                mid = Act(z(N, sh, sw, F_), N, sh, sw, F_)
                self._cbr(plan, "layer%s/conv_1" % name, "layer%s/bn_1" % name, catin, F_, 3, 3, 1, "SAME", mid)
            out = Act(z(N, sh, sw, F_), N, sh, sw, F_)
            self._cbr(plan, "layer%s/conv_2" % name, "layer%s/bn_2" % name, mid, F_, 3, 3, 1, "SAME", out)
            net = out
        self.conv9 = net
        co = up4(self.COUT)
        self.yhat = Act(z(N, H, W, co), N, H, W, self.COUT)
        self.d_final = ops.conv_desc(N, H, W, net.Cp, self.COUT, 1, 1, 1, "SAME", ldx=net.ld, ldy=co, ldw=co,
                                     act=ACT_SIGMOID)
        ops.conv2d_fwd(plan, self.d_final, net.ptr, self._P("final/kernel"), self._P("final/bias"), self.yhat.ptr)

    # ---- backward ---------------------------------------------------------------------------------------
    def _gbuf(self, a):
        z = self.session.zeros
        return Act(z(a.N, a.H, a.W, up4(a.C)), a.N, a.H, a.W, a.C)

    def _cbr_back(self, plan, name, gy, dx, res=None):
        """gy: gradient w.r.t. the layer's ReLU output (overwritten with the pre-BN gradient);
        dx: where the gradient w.r.t. the layer's input goes (None for the first layer)"""
        L = self.layers[name]
        K = L.d.K
        ops.bn_bwd(plan, L.raw.ptr, L.raw.ld, gy.ptr, gy.ld, L.scale, L.shift, L.save_mean, L.save_invstd,
                   self._P(L.bn + "/gamma"), L.y.pixels, up4(K), gy.ptr, gy.ld, self._G(L.bn + "/gamma"),
                   self._G(L.bn + "/beta"))
        aff = getattr(L.x, "affine", None)
        if aff is not None:      # the input was never normalised in memory: the weight gradient forms it on load, like the forward
            ops.conv2d_wgrad_affine(plan, L.d, self._prec(L.d), L.x.ptr, aff[0], aff[1], gy.ptr, gy.ld,
                                    self._G(name + "/kernel"), self._G(name + "/bias"))
        elif self._use_split(L.d):
            ops.conv2d_wgrad_split3(plan, L.d, L.x.ptr, gy.ptr, gy.ld, self._G(name + "/kernel"),
                                    self._G(name + "/bias"), bf16=self._bf16)
        else:
            ops.conv2d_wgrad(plan, L.d, L.x.ptr, gy.ptr, gy.ld, self._G(name + "/kernel"), self._G(name + "/bias"))
        if dx is not None and self._use_split(L.d):
            wt = self._wsplit(name, ops.conv2d_split3_dgrad_weight_bytes(L.d), "dgrad")
            jobs = getattr(self, "_prep_bwd", None)
            if jobs is not None and len(jobs.jobs) < 16:
                jobs.add(L.d, self._P(name + "/kernel"), wt, 1)
            else:
                ops.conv2d_split3_prepare_dgrad(plan, L.d, self._P(name + "/kernel"), wt)
            ops.conv2d_dgrad_split3(plan, L.d, gy.ptr, gy.ld, wt, dx.ptr,
                                    res.ptr if res is not None else None, res.ld if res is not None else 0,
                                    None, 0, lddx=dx.ld, bf16=self._bf16)
        elif dx is not None:
            ops.conv2d_dgrad(plan, L.d, gy.ptr, gy.ld, self._P(name + "/kernel"), dx.ptr,
                             res.ptr if res is not None else None, res.ld if res is not None else 0,
                             None, 0, lddx=dx.ld)


    def record_backward(self, plan, g_logit, kl_weight):
        """g_logit: gradient w.r.t. the PRE-sigmoid output [N,H,W,up4(cout)] (from recon_loss);
        kl_weight: d loss / d kl[n] (trainer/trainer.py:61-73: 1 / (1e6 * N * Z), the 0.5 is inside kl)."""
        N = self.N
        z = self.session.zeros
        Zn = self.Z

        gbuf = self._gbuf

        def cbr_back(name, gy, dx, res=None):
            return self._cbr_back(plan, name, gy, dx, res)

        # final 1x1 conv (sigmoid folded into g_logit)
        g_final = Act(g_logit, N, self.height, self.width, self.COUT)
        g = gbuf(self.conv9)
        ops.conv2d_wgrad(plan, self.d_final, self.conv9.ptr, g_final.ptr, g_final.ld, self._G("final/kernel"),
                         self._G("final/bias"))
        ops.conv2d_dgrad(plan, self.d_final, g_final.ptr, g_final.ld, self._P("final/kernel"), g.ptr)
        g_skip = {}
        for name, F_, k, skip in reversed(self.DEC):
            L2, L1 = self.layers["layer%s/conv_2" % name], self.layers["layer%s/conv_1" % name]
            g_mid = gbuf(L2.x)
            cbr_back(L2.name, g, g_mid)
            buf, fup, fs = self.cat[skip]
            g_cat_t = z(*buf.shape)
            g_cat = Act(g_cat_t, N, L1.x.H, L1.x.W, fup + fs)
            cbr_back(L1.name, g_mid, g_cat)
            g_up = Act(g_cat_t, N, L1.x.H, L1.x.W, fup, fup + fs, 0)
            g_skip[skip] = Act(g_cat_t, N, L1.x.H, L1.x.W, fs, fup + fs, fup)
            d, src, up = self.ups[name]
            ops.deconv_wgrad(plan, d, src.ptr, g_up.ptr, g_up.ld, self._G("upsample_%s/kernel" % name),
                             self._G("upsample_%s/bias" % name))
            g = gbuf(src)
            mask = src if src is self.c2d else None       # conv2d has a plain ReLU; BN layers mask inside bn_bwd
            ops.deconv_dgrad(plan, d, g_up.ptr, g_up.ld, self._P("upsample_%s/kernel" % name), g.ptr,
                             mask.ptr if mask is not None else None, mask.ld if mask is not None else 0)
        # conv2d 1 -> 128 (3x3, ReLU): g is already its pre-activation gradient
        g_dns = gbuf(self.dns)
        ops.conv2d_wgrad(plan, self.d_c2d, self.dns.ptr, g.ptr, g.ld, self._G("conv2d/kernel"), self._G("conv2d/bias"))
        ops.conv2d_dgrad(plan, self.d_c2d, g.ptr, g.ld, self._P("conv2d/kernel"), g_dns.ptr, None, 0, self.dns.ptr, 4)
        nd = self.HEAD[0] * self.HEAD[1]
        g_dns1 = z(N, nd)
        ops.grad_slice(plan, g_dns.ptr, 4, g_dns1, 1, None, 0, N * nd, 1)
        g_z = z(N, Zn)
        ops.conv2d_wgrad(plan, self.d_dense, self.zbuf, g_dns1, nd, self._G("dense/kernel"), self._G("dense/bias"))
        ops.conv2d_dgrad(plan, self.d_dense, g_dns1, nd, self._P("dense/kernel"), g_z)
        g_heads = z(N, 2 * Zn)
        ops.latent_linear_bwd(plan, self.heads_out, self.eps, g_z, Zn, kl_weight, g_heads, N, Zn)
        self._backward_encoder(plan, g_heads, g_skip)
        self._grad_bufs = dict(g_z=g_z, g_heads=g_heads)

    def _backward_encoder(self, plan, g_heads, g_skip):
        """g_heads [N, 2Z]: gradient w.r.t. the fused heads output; heads weight + data gradient, then the encoder"""
        Zn = self.Z
        gbuf = self._gbuf

        def cbr_back(name, gy, dx, res=None):
            return self._cbr_back(plan, name, gy, dx, res)

        g = gbuf(self.conv5)
        ops.conv2d_wgrad(plan, self.d_heads, self.conv5.t, g_heads, 2 * Zn, self._G("heads/kernel"),
                         self._G("heads/bias"))
        ops.conv2d_dgrad(plan, self.d_heads, g_heads, 2 * Zn, self._P("heads/kernel"), g.t)
        for name, F_, pool, pad in reversed(self.ENC):
            if pool is not None:
                # g = gradient w.r.t. the pool layer's output; the skip branch joins at the pool conv's input
                Lp = self.layers["layer%s/pool_2" % name]
                g_y2 = gbuf(Lp.x)
                cbr_back(Lp.name, g, g_y2, res=g_skip.get(name))
                g = g_y2
            L2, L1 = self.layers["layer%s/conv_2" % name], self.layers["layer%s/conv_1" % name]
            g_mid = gbuf(L2.x)
            cbr_back(L2.name, g, g_mid)
            if L1.x is self.xpad:
                cbr_back(L1.name, g_mid, None)
            else:
                g = gbuf(L1.x)
                cbr_back(L1.name, g_mid, g)


class UNet(UNetVAE):
    """models/unet_architecture.py (RGB frames 224x298x3)"""
    SCOPE, CIN, WD, HEAD, COUT = "UNet", 3, 7e-5, (14, 18), 3
    ENC = [("1", 8, (3, 3), "SAME"), ("2", 32, (2, 3), "VALID"), ("3", 32, (3, 3), "SAME"),
           ("4", 64, (2, 3), "VALID"), ("5", 128, None, None)]
    DEC = [("6", 64, (2, 3), "4"), ("7", 32, (2, 2), "3"), ("8", 32, (2, 3), "2"), ("9", 8, (2, 2), "1")]

    def __init__(self, input_shape=None, precision="split", defer_bn=True):
        super(UNet, self).__init__(input_shape or [224, 298, 3], precision, defer_bn)


class UNetSound(UNetVAE):
    """models/unet_sound.py (STFT magnitude 99x257x1)"""
    SCOPE, CIN, WD, HEAD, COUT = "UNetAudio", 1, 6e-5, (6, 16), 1
    ENC = [("1", 8, (3, 3), "VALID"), ("2", 8, (3, 2), "VALID"), ("3", 32, (3, 3), "SAME"),
           ("4", 64, (3, 3), "SAME"), ("5", 128, None, None)]
    DEC = [("6", 64, (2, 2), "4"), ("7", 32, (2, 2), "3"), ("8", 8, (3, 2), "2"), ("9", 8, (3, 3), "1")]

    def __init__(self, input_shape=None, precision="split", defer_bn=True):
        super(UNetSound, self).__init__(input_shape or [99, 257, 1], precision, defer_bn)


class AssociatorAudio(UNetVAE):
    """models/multimodal.py:139-285: the conv associator - a spectrogram ENCODER (conv_conv_pool x 5, strided-conv
    "pools", batch norm in training mode) with two 12x16 VALID heads, mean and std = softplus(.), whose outputs stand
    where the dense associators' do: the halves of ONE [N, 300] buffer that `UNetAcZ` takes as its external latent
    statistics.  Same protocol as the dense associators (acimg/multimodal.py): `_build_model(inputs)` sets `mean`,
    `std`, `plan_fwd`, `train_vars`; `record_backward(plan, g_ext)` consumes d loss / d [mean | std]."""
    SCOPE, CIN, WD, HEAD, COUT = "AssociatorAudio", 1, 8e-5, (12, 16), None
    Z = 150
    ENC = [("1", 16, (3, 3), "VALID"), ("2", 16, (3, 3), "SAME"), ("3", 64, (3, 3), "SAME"), ("4", 128, (3, 3), "SAME"),
           ("5", 128, None, None)]
    DEC = []
    HEAD_NAMES = ("mean", "std")
    ENCODER_ONLY = True
    IMAGE_INPUT = True

    def __init__(self, input_shape=None, precision="split", defer_bn=True):
        super(AssociatorAudio, self).__init__(input_shape or [193, 257, 1], precision, defer_bn)

    def _build_model(self, inputs, session=None, training=True):
        """inputs: device buffer [N,193,257,1] (the STFT-magnitude spectrogram)"""
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        N = inputs.shape[0]
        assert tuple(inputs.shape[1:]) == (self.height, self.width, self.channels)
        self.N, self.training = N, training
        z = sess.zeros
        H, W = self.height, self.width
        self.images = inputs
        self.xpad = Act(z(N, H, W, up4(self.CIN)), N, H, W, self.CIN)
        sizes, h, w = {}, H, W
        for name, F_, pool, pad in self.ENC:
            sizes[name] = (h, w, F_)
            if pool is not None:
                h, w = (-(-h // 2), -(-w // 2)) if pad == "SAME" else ((h - pool[0]) // 2 + 1, (w - pool[1]) // 2 + 1)
        assert (h, w) == tuple(self.HEAD)
        self.cat = {}
        self.layers = OrderedDict()
        p = sess.new_plan()
        self._begin_prepares(p)
        self._record_encoder(p, sizes)
        Zn = self.Z
        self.ext = z(N, 2 * Zn)             # [mean | std = softplus(raw)]
        ops.grad_slice(p, self.heads_out, 2 * Zn, self.ext, 2 * Zn, None, 0, N, Zn)
        ops.softplus_fwd(p, ops.Ptr(self.heads_out, Zn), 2 * Zn, ops.Ptr(self.ext, Zn), 2 * Zn, N, Zn)
        self.plan_fwd = p
        self.mean, self.std = self.ext[:, :Zn], self.ext[:, Zn:]
        self.network = OrderedDict(input=inputs, is_training=None, keep_prob=None, features=self.conv5.t)
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/") and
                           not n.endswith(("moving_mean", "moving_variance"))]

    def record_backward(self, plan, g_ext):
        """g_ext [N, 300]: d loss / d [mean | std] (e.g. `UNetAcZ.g_ext`)"""
        N, Zn = self.N, self.Z
        g_heads = self.session.zeros(N, 2 * Zn)
        ops.grad_slice(plan, g_ext, 2 * Zn, g_heads, 2 * Zn, None, 0, N, Zn)
        ops.softplus_bwd(plan, ops.Ptr(self.heads_out, Zn), 2 * Zn, ops.Ptr(g_ext, Zn), 2 * Zn, ops.Ptr(g_heads, Zn),
                         2 * Zn, N, Zn)
        self._backward_encoder(plan, g_heads, {})
