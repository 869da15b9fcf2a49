"""The acoustic-image VAE `UNetAc` (scope 'UNetAcoustic') of models/unet_noconc.py:46-89 (z from its own heads:
`UNetAcNoConc`) and models/unet_z.py:46-82 (same network, z = mean2 + std2 * eps from EXTERNAL statistics:
`UNetAcZ`, the decoder the associator trainers drive), MI355X-native.

Model protocol as in the reference: `scope`, `init_model`, `_build_model(acoustic_images[, mean2, std2])` setting
`mean`, `std`, `output`, `network`, `train_vars`.  No batch norm (commented out in the reference), so every layer
is one implicit-GEMM launch with bias + ReLU in its epilogue and the backward kernels emit pre-activation
gradients (ReLU masks from the saved activations) exactly as in `unet_acresnet.py`; the 36x48 128-channel layers
run on the split-MFMA kernels.  With `TrainerVAE` the stand-alone model trains as trainer/trainer.py does with
encoder_type 'Ac'; with external statistics `record_backward` leaves d loss / d (mean2 | std2) in `g_ext`.
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

Z = 150


class UNetAcNoConc(object):
    WD = 0.0                      # kernel_regularizer=None everywhere (unet_noconc.py:143,163,197)
    EXTERNAL_Z = False

    def __init__(self, input_shape=None, precision="split"):
        assert precision in ("split", "f32")
        self.precision = precision
        self.scope = 'UNetAcoustic'
        self.height, self.width, self.channels = input_shape or [36, 48, 12]
        self.Z = Z
        self.session = None
        self._wsplit_bufs = {}

    # ---- variables ------------------------------------------------------------------------------------------
    def _conv_specs(self):
        return OrderedDict([
            ("final", (3, 3, 128, 12)),
            ("layer5/conv_2", (3, 3, 128, 128)), ("layer5/conv_1", (3, 3, 128, 128)),
            ("layer4/conv_2", (3, 3, 128, 128)), ("layer4/conv_1", (3, 3, 128, 128)),
            ("conv2d", (3, 3, 12, 133)),
            ("layer3/conv_2", (3, 3, 133, 133)), ("layer3/conv_1", (3, 3, 128, 133)),
            ("layer1/pool_2", (3, 3, 128, 128)), ("layer1/conv_2", (3, 3, 128, 128)),
            ("layer1/conv_1", (3, 3, 12, 128)),
        ])

    def _register(self, store):
        s = self.scope
        specs = self._conv_specs()

        def conv(name):
            store.add(Var("%s/%s/kernel" % (s, name), specs[name], "conv", "train"))
            store.add(Var("%s/%s/bias" % (s, name), (specs[name][3],), "vec", "train"))

        for name in ("final", "layer5/conv_2", "layer5/conv_1", "layer4/conv_2", "layer4/conv_1"):
            conv(name)
        store.add(Var(s + "/upsample_1/kernel", (2, 2, 128, 133), "deconv", "train"))
        store.add(Var(s + "/upsample_1/bias", (128,), "vec", "train"))
        conv("conv2d")
        store.add(Var(s + "/dense/kernel", (Z, 12 * 16 * 12), "dense", "train"))
        store.add(Var(s + "/dense/bias", (12 * 16 * 12,), "vec", "train"))
        self.heads = FusedHeads(s, 133, Z, True, hw=(12, 16), names=("mean", "std"))
        store.add_fused(self.heads)
        for name in ("layer3/conv_2", "layer3/conv_1", "layer1/pool_2", "layer1/conv_2", "layer1/conv_1"):
            conv(name)

    def reg_range(self):
        return 0, 0

    def init_model(self, session, checkpoint_file):
        state = load_state_file(checkpoint_file)
        store = (session or self.session).store
        return store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def initialize(self, seed=1242, state=None):
        if state is None:
            g = torch.Generator().manual_seed(seed)
            state = OrderedDict()

            def xav(shape, fin, fout):
                lim = np.sqrt(6.0 / (fin + fout))
                return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()

            for name, (kh, kw, cin, cout) in self._conv_specs().items():
                state["%s/%s/kernel" % (self.scope, name)] = xav((kh, kw, cin, cout), kh * kw * cin, kh * kw * cout)
                state["%s/%s/bias" % (self.scope, name)] = torch.zeros(cout)
            for h in ("mean", "std"):
                state["%s/%s/kernel" % (self.scope, h)] = xav((12, 16, 133, Z), 12 * 16 * 133, 12 * 16 * Z)
                state["%s/%s/bias" % (self.scope, h)] = torch.zeros(Z)
            state[self.scope + "/dense/kernel"] = xav((Z, 2304), Z, 2304)
            state[self.scope + "/dense/bias"] = torch.zeros(2304)
            state[self.scope + "/upsample_1/kernel"] = xav((2, 2, 128, 133), 4 * 133, 4 * 128)
            state[self.scope + "/upsample_1/bias"] = torch.zeros(128)
        self.session.store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def _P(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.p(self.scope + "/" + name))

    def _G(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.g(self.scope + "/" + name))

    # ---- graph ------------------------------------------------------------------------------------------------
    def _alloc(self, acoustic_images, mean2=None, std2=None, session=None, eps=None):
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        N = acoustic_images.shape[0]
        H, W = self.height, self.width
        assert tuple(acoustic_images.shape[1:]) == (H, W, 12) and H % 3 == 0 and W % 3 == 0
        self.N = N
        z = sess.zeros
        h, w = H // 3, W // 3
        assert (h, w) == (12, 16)
        self.images = acoustic_images
        self.xpad = Act(acoustic_images, N, H, W, 12)          # (TrainerVAE reads the target from here)
        self.eps = eps if eps is not None else z(N, Z)
        if self.EXTERNAL_Z:
            assert mean2 is not None and std2 is not None
            self.ext = mean2._base if mean2._base is not None else mean2
            assert self.ext.shape == (N, 2 * Z), "mean2 / std2 must be the halves of one [N, 300] buffer"
        self.c11 = Act(z(N, H, W, 128), N, H, W, 128)
        self.conv1 = Act(z(N, H, W, 128), N, H, W, 128)
        self.pool1 = Act(z(N, h, w, 128), N, h, w, 128)
        self.c31 = Act(z(N, h, w, 136), N, h, w, 133)
        self.conv2 = Act(z(N, h, w, 136), N, h, w, 133)
        self.heads_out = z(N, 2 * Z)
        self.sigma = z(N, Z)
        self.zbuf = z(N, 152)
        self.kl = z(N)
        self.dns = Act(z(N, h, w, 12), N, h, w, 12)
        self.net = Act(z(N, h, w, 136), N, h, w, 133)
        self.up = Act(z(N, H, W, 128), N, H, W, 128)
        self.c41 = Act(z(N, H, W, 128), N, H, W, 128)
        self.conv4 = Act(z(N, H, W, 128), N, H, W, 128)
        self.c51 = Act(z(N, H, W, 128), N, H, W, 128)
        self.conv5 = Act(z(N, H, W, 128), N, H, W, 128)
        self.yhat = Act(z(N, H, W, 12), N, H, W, 12)

    def _build_model(self, acoustic_images, mean2=None, std2=None, session=None, eps=None):
        """acoustic_images: device buffer [N,36,48,12]; eps: device buffer [N,150].  UNetAcZ: mean2 / std2 are
        views of ONE device buffer `ext` [N, 2*150] = [mean2 | std2] (pass the same tensor's halves)."""
        self._alloc(acoustic_images, mean2, std2, session, eps)
        self.plan_fwd = self.session.new_plan()
        self._record_forward(self.plan_fwd)
        self._publish()

    def _publish(self):
        sess, acoustic_images = self.session, self.images
        self.mean = self.heads_out[:, :Z]
        self.std = self.sigma
        self.output = self.yhat.t
        self.network = OrderedDict(input=acoustic_images, is_training=None, keep_prob=None, features=self.conv2.t)
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/")]

    def _desc(self, x, K, stride=1, y=None, act=ACT_NONE):
        return ops.conv_desc(x.N, x.H, x.W, x.Cp if x.off == 0 and x.ld == x.Cp else x.C, K, 3, 3, stride, "SAME",
                             ldx=x.ld, ldy=(y.ld if y is not None else up4(K)), ldw=up4(K), act=act)

    def _use_split(self, d):
        return (self.precision == "split" and d.stride == 1 and d.C % 32 == 0 and d.K % 32 == 0 and
                d.N * d.OH * d.OW >= 16384)

    def _wsplit(self, name, nbytes, kind):
        key = (name, kind)
        if key not in self._wsplit_bufs:
            self._wsplit_bufs[key] = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.session.device)
        return self._wsplit_bufs[key]

    def _conv(self, plan, name, x, y, stride=1, act=ACT_RELU):
        d = self._desc(x, y.C, stride, y, act)
        self._descs[name] = (d, x, y)
        if self._use_split(d):
            ws = self._wsplit(name, ops.conv2d_split3_weight_bytes(d), "fwd")
            ops.conv2d_split3_prepare(plan, d, self._P(name + "/kernel"), ws)
            ops.conv2d_fwd_split3(plan, d, x.ptr, ws, y.ptr, bias=self._P(name + "/bias"))
        else:
            ops.conv2d_fwd(plan, d, x.ptr, self._P(name + "/kernel"), self._P(name + "/bias"), y.ptr)

    def _record_forward(self, plan):
        self._descs = {}
        self._record_encoder(plan)
        self._record_decoder(plan, self.conv2.t)

    def _record_encoder(self, plan):
        """models/unet_noconc2.py:48-64 (`_build_network`): acoustic image -> the 12x16x133 feature map `conv2`"""
        self._conv(plan, "layer1/conv_1", self.xpad, self.c11)
        self._conv(plan, "layer1/conv_2", self.c11, self.conv1)
        self._conv(plan, "layer1/pool_2", self.conv1, self.pool1, stride=3)
        self._conv(plan, "layer3/conv_1", self.pool1, self.c31)
        self._conv(plan, "layer3/conv_2", self.c31, self.conv2)

    def _record_decoder(self, plan, feat):
        """heads + latent + decoder from a feature map `feat` [N,12,16,136] (133 channels valid): the model's own `conv2`, or
        - models/unet_noconc2.py:66-96, `_build_network2(f)` - the joint MLP's acoustic head"""
        N = self.N
        h, w = self.pool1.H, self.pool1.W
        self.feat = feat
        kin = h * w * 136
        self.d_heads = ops.conv_desc(N, 1, 1, kin, 2 * Z, 1, 1, 1, "VALID", ldx=kin, ldy=2 * Z, ldw=2 * Z)
        ops.conv2d_fwd(plan, self.d_heads, feat, self._P("heads/kernel"), self._P("heads/bias"), self.heads_out)
        if self.EXTERNAL_Z:
            # own statistics are still produced (the associator trainers read them); z comes from outside
            self.own_kl = self.session.zeros(N)
            self.own_z = self.session.zeros(N, 152)
            ops.latent_fwd(plan, self.heads_out, self.eps, self.own_z, 152, self.sigma, self.own_kl, N, Z)
            ops.latent_linear_fwd(plan, self.ext, self.eps, self.zbuf, 152, self.kl, N, Z)
        else:
            ops.latent_fwd(plan, self.heads_out, self.eps, self.zbuf, 152, self.sigma, self.kl, N, Z)
        self.d_dense = ops.conv_desc(N, 1, 1, 152, 2304, 1, 1, 1, "VALID", ldx=152, ldy=2304, ldw=2304, act=ACT_RELU)
        ops.conv2d_fwd(plan, self.d_dense, self.zbuf, self._P("dense/kernel"), self._P("dense/bias"), self.dns.t)
        self._conv(plan, "conv2d", self.dns, self.net)
        self.d_up = ops.deconv_desc(N, h, w, 136, 128, 2, 2, 3, ldx=136, ldy=128, ldw=136)
        ops.deconv_fwd(plan, self.d_up, self.net.ptr, self._P("upsample_1/kernel"), self._P("upsample_1/bias"), self.up.ptr)
        self._conv(plan, "layer4/conv_1", self.up, self.c41)
        self._conv(plan, "layer4/conv_2", self.c41, self.conv4)
        self._conv(plan, "layer5/conv_1", self.conv4, self.c51)
        self._conv(plan, "layer5/conv_2", self.c51, self.conv5)
        self._conv(plan, "final", self.conv5, self.yhat, act=ACT_SIGMOID)

    # ---- backward ---------------------------------------------------------------------------------------------
    def record_backward(self, plan, g_logit, kl_weight, stop_at_features=False):
        """g_logit: gradient w.r.t. the pre-sigmoid output; kl_weight: d loss / d kl[n].  UNetAcZ: the gradient
        w.r.t. the external statistics is left in `self.g_ext` [N, 300] (KL term of (mean2, std2) included) and the
        encoder receives no gradient.  stop_at_features: the decoder was fed an outside feature map (`_record_decoder`):
        d loss / d feat is left in `self.g_feat` [N,12,16,136], unmasked, and the encoder receives no gradient."""
        N = self.N
        z = self.session.zeros
        H, W, h, w = self.height, self.width, self.pool1.H, self.pool1.W

        def gbuf(a):
            return Act(z(a.N, a.H, a.W, up4(a.C)), a.N, a.H, a.W, a.C)

        def back(name, gy, dx=None, mask=None):
            d, x, y = self._descs[name]
            wg = ops.conv2d_wgrad_split3 if (self._use_split(d) and d.K % 64 == 0) else ops.conv2d_wgrad
            wg(plan, d, x.ptr, gy.ptr, gy.ld, self._G(name + "/kernel"), self._G(name + "/bias"))
            if dx is not None and self._use_split(d):
                wt = self._wsplit(name, ops.conv2d_split3_dgrad_weight_bytes(d), "dgrad")
                ops.conv2d_split3_prepare_dgrad(plan, d, self._P(name + "/kernel"), wt)
                ops.conv2d_dgrad_split3(plan, d, gy.ptr, gy.ld, wt, dx.ptr, None, 0,
                                        mask.ptr if mask is not None else None, mask.ld if mask is not None else 0,
                                        lddx=dx.ld)
            elif dx is not None:
                ops.conv2d_dgrad(plan, d, gy.ptr, gy.ld, self._P(name + "/kernel"), dx.ptr, None, 0,
                                 mask.ptr if mask is not None else None, mask.ld if mask is not None else 0, lddx=dx.ld)

        g_final = Act(g_logit, N, H, W, 12)
        g_conv5, g_c51, g_conv4, g_c41, g_up = (gbuf(self.conv5), gbuf(self.c51), gbuf(self.conv4), gbuf(self.c41),
                                                gbuf(self.up))
        back("final", g_final, g_conv5, mask=self.conv5)
        back("layer5/conv_2", g_conv5, g_c51, mask=self.c51)
        back("layer5/conv_1", g_c51, g_conv4, mask=self.conv4)
        back("layer4/conv_2", g_conv4, g_c41, mask=self.c41)
        back("layer4/conv_1", g_c41, g_up)                 # the transposed conv has no activation
        g_net = gbuf(self.net)
        ops.deconv_wgrad(plan, self.d_up, self.net.ptr, g_up.ptr, g_up.ld, self._G("upsample_1/kernel"),
                         self._G("upsample_1/bias"))
        ops.deconv_dgrad(plan, self.d_up, g_up.ptr, g_up.ld, self._P("upsample_1/kernel"), g_net.ptr, self.net.ptr,
                         self.net.ld)
        g_dns = gbuf(self.dns)
        back("conv2d", g_net, g_dns, mask=self.dns)
        g_z = z(N, 152)
        ops.conv2d_wgrad(plan, self.d_dense, self.zbuf, g_dns.t, 2304, self._G("dense/kernel"), self._G("dense/bias"))
        ops.conv2d_dgrad(plan, self.d_dense, g_dns.t, 2304, self._P("dense/kernel"), g_z)
        self._grad_bufs = dict(g_z=g_z)
        if self.EXTERNAL_Z:
            self.g_ext = z(N, 2 * Z)
            ops.latent_linear_bwd(plan, self.ext, self.eps, g_z, 152, kl_weight, self.g_ext, N, Z)
            return
        g_heads = z(N, 2 * Z)
        ops.latent_bwd(plan, self.heads_out, self.eps, self.sigma, g_z, 152, kl_weight, g_heads, N, Z)
        g_conv2 = gbuf(self.conv2)
        ops.conv2d_wgrad(plan, self.d_heads, self.feat, g_heads, 2 * Z, self._G("heads/kernel"), self._G("heads/bias"))
        if stop_at_features:
            ops.conv2d_dgrad(plan, self.d_heads, g_heads, 2 * Z, self._P("heads/kernel"), g_conv2.t)
            self.g_feat = g_conv2.t
            self._grad_bufs.update(g_heads=g_heads)
            return
        ops.conv2d_dgrad(plan, self.d_heads, g_heads, 2 * Z, self._P("heads/kernel"), g_conv2.t, None, 0,
                         self.conv2.t, h * w * 136)        # ReLU mask of conv2, viewed as one 12*16*136-channel pixel
        g_c31, g_pool1, g_conv1, g_c11 = gbuf(self.c31), gbuf(self.pool1), gbuf(self.conv1), gbuf(self.c11)
        back("layer3/conv_2", g_conv2, g_c31, mask=self.c31)
        back("layer3/conv_1", g_c31, g_pool1, mask=self.pool1)
        back("layer1/pool_2", g_pool1, g_conv1, mask=self.conv1)
        back("layer1/conv_2", g_conv1, g_c11, mask=self.c11)
        back("layer1/conv_1", g_c11, None)
        self._grad_bufs.update(g_heads=g_heads, g_conv2=g_conv2)


class UNetAcZ(UNetAcNoConc):
    """models/unet_z.py: `_build_model(acoustic_images, mean2, std2)`"""
    EXTERNAL_Z = True
