"""The three SPLIT VAEs of the joint-latent experiments (main.py:190-201, FLAGS.jointmvae): `Unet2`
(models/unet_architecture_noconc2.py, RGB frames 224x298x3), `UNetSound22` (models/unet_sound22.py, STFT spectrograms
193x257x1) and `UNetAc2` (models/unet_noconc2.py, acoustic images 36x48x12), MI355X-native.

Each reference class builds in two halves around a fusion network (trainer/trainermulti.py:44-57):
    `_build_network(inputs)` -> the 12x16 feature map of the encoder (conv_conv_pool stack, no skip connections),
    `_build_model(f)`        -> two 12x16 VALID heads on an OUTSIDE feature map f (mean, std = softplus), z = mean + std * eps,
                                dense, conv, transposed convs each followed by two conv_conv blocks, sigmoid output.
Here: `_build_network` records `plan_enc` and returns `features`; `_build_model(f)` records `plan_fwd`;
`record_backward(plan, g_logit, kl_weight)` walks the decoder back and leaves d loss / d f in `g_feat` (row stride `feat_ld`)
- the encoder receives no gradient (nothing upstream of the fusion network is trained in this step).

`Unet2` / `UNetSound22` are conv-BN-ReLU (batch norm in training mode, kernel regularisers) on the recorder of
acimg/unet_vae.py (`_cbr` / `_cbr_back`: conv + statistics, bn_finalize, normalise pass; bn_bwd, data gradient);
`UNetAc2` has neither batch norm nor regularisers and is `UNetAcNoConc` (acimg/unet_acoustic.py) recorded in two halves.
"""
from collections import OrderedDict

from . import ops
from .ops import ACT_RELU, ACT_SIGMOID
from .params import up4
from .session import get_default_session
from .unet_acoustic import UNetAcNoConc
from .unet_acresnet import Act
from .unet_vae import UNetVAE

HEAD = (12, 16)


class UNetAc2(UNetAcNoConc):
    """models/unet_noconc2.py:48-110"""
    FEAT_C, FEAT_LD = 133, 136

    def _build_network(self, acoustic_images, session=None, eps=None):
        self._alloc(acoustic_images, None, None, session, eps)
        self.plan_enc = self.session.new_plan()
        self._descs = {}
        self._record_encoder(self.plan_enc)
        self.features = self.conv2.t[..., :133]
        return self.features

    def _build_model(self, f):
        base = f._base if f._base is not None else f
        N = self.N
        assert base.numel() == N * 12 * 16 * 136, "f must be a [N,12,16,133] view of a buffer with 136 channels per pixel"
        self.plan_fwd = self.session.new_plan()
        self._record_decoder(self.plan_fwd, base)
        self._publish()

    def record_backward(self, plan, g_logit, kl_weight):
        UNetAcNoConc.record_backward(self, plan, g_logit, kl_weight, stop_at_features=True)
        self.feat_ld = 136

    def reg_ranges(self):
        return []


class _SplitBN(UNetVAE):
    """conv-BN-ReLU split VAE: class attributes give the layer table.
    ENC row: (layer, filters, pool kernel (kh, kw), pool stride, pool padding) - pool None for the last block;
    DEC row: (upsample name, filters, kernel (kh, kw), stride, (block, block))"""
    HEAD_NAMES = ("mean", "std")
    HEAD = HEAD
    WD_ENC = WD_DEC = 0.0
    DENSE_CH = C2D = None

    def _layer_table(self):
        t = []
        cin = self.CIN
        for name, F_, pool, stride, pad in self.ENC:
            t.append(("cbr", "layer%s/conv_1" % name, "layer%s/bn_1" % name, (3, 3, cin, F_)))
            t.append(("cbr", "layer%s/conv_2" % name, "layer%s/bn_2" % name, (3, 3, F_, F_)))
            if pool is not None:
                t.append(("cbr", "layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, (pool[0], pool[1], F_, F_)))
            cin = F_
        self.FEAT_C, self.FEAT_LD = cin, up4(cin)
        t.append(("heads", None, None, (HEAD[0], HEAD[1], cin, self.Z)))
        t.append(("dense", "dense", None, (self.Z, HEAD[0] * HEAD[1] * self.DENSE_CH)))
        t.append(("conv", "conv2d", None, (3, 3, self.DENSE_CH, self.C2D)))
        cin = self.C2D
        for name, F_, k, stride, blocks in self.DEC:
            t.append(("deconv", "upsample_%s" % name, None, (k[0], k[1], F_, cin)))
            for b in blocks:
                t.append(("cbr", "layer%s/conv_1" % b, "layer%s/bn_1" % b, (3, 3, F_, F_)))
                t.append(("cbr", "layer%s/conv_2" % b, "layer%s/bn_2" % b, (3, 3, F_, F_)))
            cin = F_
        t.append(("conv", "final", None, (1, 1, cin, self.COUT)))
        return t

    def reg_ranges(self):
        """[(weight decay, offset, numel)] of the regularised kernels in the flat trainable buffer: encoder kernels
        (`_build_network`'s weight_decay), then decoder kernels (`_build_network2`'s) - registered in that order"""
        st = self.session.store
        a, b = self._reg_range
        first_dec = st.vars["%s/upsample_%s/kernel" % (self.scope, self.DEC[0][0])]
        end = b.offset + b.numel
        return [(self.WD_ENC, a.offset, first_dec.offset - a.offset), (self.WD_DEC, first_dec.offset, end - first_dec.offset)]

    # ---- encoder --------------------------------------------------------------------------------------------
    def _build_network(self, images, session=None, eps=None, training=True):
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        N = images.shape[0]
        H, W = self.height, self.width
        assert tuple(images.shape[1:]) == (H, W, self.channels)
        self.N, self.training = N, training
        z = sess.zeros
        self.images = images
        self.xpad = Act(z(N, H, W, up4(self.CIN)), N, H, W, self.CIN)
        self.eps = eps if eps is not None else z(N, self.Z)
        self.layers = OrderedDict()
        self.cat = {}
        p = sess.new_plan()
        ops.pad_channels(p, images, self.xpad.t, N * H * W, self.CIN, self.xpad.Cp)
        net = self.xpad
        h, w = H, W
        for name, F_, pool, stride, pad in self.ENC:
            mid = Act(z(N, h, w, F_), N, h, w, F_)
            self._cbr(p, "layer%s/conv_1" % name, "layer%s/bn_1" % name, net, F_, 3, 3, 1, "SAME", mid)
            out = Act(z(N, h, w, F_), N, h, w, F_)
            self._cbr(p, "layer%s/conv_2" % name, "layer%s/bn_2" % name, mid, F_, 3, 3, 1, "SAME", out)
            net = out
            if pool is not None:
                if pad == "SAME":
                    h, w = -(-h // stride), -(-w // stride)
                else:
                    h, w = (h - pool[0]) // stride + 1, (w - pool[1]) // stride + 1
                po = Act(z(N, h, w, F_), N, h, w, F_)
                self._cbr(p, "layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, net, F_, pool[0], pool[1], stride, pad, po)
                net = po
        assert (h, w) == HEAD, "input %dx%d does not reduce to the 12x16 feature map" % (H, W)
        self.conv5 = net
        self.plan_enc = p
        self.features = net.t[..., :net.C]
        return self.features

    # ---- heads + decoder from an outside feature map -----------------------------------------------------
    def _build_model(self, f):
        sess = self.session
        base = f._base if f._base is not None else f
        N, Zn = self.N, self.Z
        C5, C5p = self.FEAT_C, self.FEAT_LD
        assert base.numel() == N * HEAD[0] * HEAD[1] * C5p
        z = sess.zeros
        p = sess.new_plan()
        self.feat = base
        kin = HEAD[0] * HEAD[1] * C5p
        self.heads_out = z(N, 2 * Zn)
        self.d_heads = ops.conv_desc(N, 1, 1, kin, 2 * Zn, 1, 1, 1, "VALID", ldx=kin, ldy=2 * Zn, ldw=2 * Zn)
        ops.conv2d_fwd(p, self.d_heads, base, self._P("heads/kernel"), self._P("heads/bias"), self.heads_out)
        self.ext = z(N, 2 * Zn)                 # [mean | std = softplus(raw)]
        ops.grad_slice(p, self.heads_out, 2 * Zn, self.ext, 2 * Zn, None, 0, N, Zn)
        ops.softplus_fwd(p, ops.Ptr(self.heads_out, Zn), 2 * Zn, ops.Ptr(self.ext, Zn), 2 * Zn, N, Zn)
        self.zbuf, self.kl = z(N, Zn), z(N)
        ops.latent_linear_fwd(p, self.ext, self.eps, self.zbuf, Zn, self.kl, N, Zn)
        D0, D0p = self.DENSE_CH, up4(self.DENSE_CH)
        npix = HEAD[0] * HEAD[1]
        nd = npix * D0
        self.dns1 = z(N, nd)
        self.d_dense = ops.conv_desc(N, 1, 1, Zn, nd, 1, 1, 1, "VALID", ldx=Zn, ldy=nd, ldw=up4(nd), act=ACT_RELU)
        ops.conv2d_fwd(p, self.d_dense, self.zbuf, self._P("dense/kernel"), self._P("dense/bias"), self.dns1)
        self.dns = Act(z(N, HEAD[0], HEAD[1], D0p), N, HEAD[0], HEAD[1], D0)
        ops.pad_channels(p, self.dns1, self.dns.t, N * npix, D0, D0p)
        self.c2d = Act(z(N, HEAD[0], HEAD[1], self.C2D), N, HEAD[0], HEAD[1], self.C2D)
        self.d_c2d = ops.conv_desc(N, HEAD[0], HEAD[1], D0p, self.C2D, 3, 3, 1, "SAME", ldx=D0p, ldy=self.C2D, ldw=self.C2D,
                                   act=ACT_RELU)
        ops.conv2d_fwd(p, self.d_c2d, self.dns.ptr, self._P("conv2d/kernel"), self._P("conv2d/bias"), self.c2d.ptr)
        net = self.c2d
        self.ups = OrderedDict()
        for name, F_, k, stride, blocks in self.DEC:
            d = ops.deconv_desc(N, net.H, net.W, net.Cp, F_, k[0], k[1], stride, ldx=net.ld, ldy=up4(F_), ldw=net.Cp)
            up = Act(z(N, d.OH, d.OW, up4(F_)), N, d.OH, d.OW, F_)
            ops.deconv_fwd(p, d, net.ptr, self._P("upsample_%s/kernel" % name), self._P("upsample_%s/bias" % name), up.ptr)
            self.ups[name] = (d, net, up)
            net = up
            for b in blocks:
                mid = Act(z(N, net.H, net.W, F_), N, net.H, net.W, F_)
                self._cbr(p, "layer%s/conv_1" % b, "layer%s/bn_1" % b, net, F_, 3, 3, 1, "SAME", mid)
                out = Act(z(N, net.H, net.W, F_), N, net.H, net.W, F_)
                self._cbr(p, "layer%s/conv_2" % b, "layer%s/bn_2" % b, mid, F_, 3, 3, 1, "SAME", out)
                net = out
        assert (net.H, net.W) == (self.height, self.width), (net.H, net.W)
        self.conv_last = net
        co = up4(self.COUT)
        self.yhat = Act(z(N, net.H, net.W, co), N, net.H, net.W, self.COUT)
        self.d_final = ops.conv_desc(N, net.H, net.W, net.Cp, self.COUT, 1, 1, 1, "SAME", ldx=net.ld, ldy=co, ldw=co,
                                     act=ACT_SIGMOID)
        ops.conv2d_fwd(p, self.d_final, net.ptr, self._P("final/kernel"), self._P("final/bias"), self.yhat.ptr)
        self.plan_fwd = p
        self.mean, self.std = self.ext[:, :Zn], self.ext[:, Zn:]
        self.output = self.yhat.t
        self.network = OrderedDict(input=self.images, is_training=None, keep_prob=None, features=self.conv5.t)
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/") and
                           not n.endswith(("moving_mean", "moving_variance"))]

    def record_backward(self, plan, g_logit, kl_weight):
        """g_logit: gradient w.r.t. the pre-sigmoid output [N,H,W,up4(cout)]; kl_weight: d loss / d kl[n].  Leaves
        d loss / d f in `g_feat` [N,12,16,feat_ld].  (Parameter gradients of this model are produced on the way and
        land in its slots of the gradient buffer; the joint step's optimiser does not read them.)"""
        N, Zn = self.N, self.Z
        z = self.session.zeros
        gbuf = self._gbuf
        net = self.conv_last
        g_final = Act(g_logit, N, self.height, self.width, self.COUT)
        g = gbuf(net)
        ops.conv2d_wgrad(plan, self.d_final, net.ptr, g_final.ptr, g_final.ld, self._G("final/kernel"), self._G("final/bias"))
        ops.conv2d_dgrad(plan, self.d_final, g_final.ptr, g_final.ld, self._P("final/kernel"), g.ptr)
        for name, F_, k, stride, blocks in reversed(self.DEC):
            for b in reversed(blocks):
                L2, L1 = self.layers["layer%s/conv_2" % b], self.layers["layer%s/conv_1" % b]
                g_mid = gbuf(L2.x)
                self._cbr_back(plan, L2.name, g, g_mid)
                g_in = gbuf(L1.x)
                self._cbr_back(plan, L1.name, g_mid, g_in)
                g = g_in
            d, src, up = self.ups[name]
            ops.deconv_wgrad(plan, d, src.ptr, g.ptr, g.ld, self._G("upsample_%s/kernel" % name),
                             self._G("upsample_%s/bias" % name))
            g_src = gbuf(src)
            mask = src if src is self.c2d else None       # conv2d has a plain ReLU; BN layers mask inside bn_bwd
            ops.deconv_dgrad(plan, d, g.ptr, g.ld, self._P("upsample_%s/kernel" % name), g_src.ptr,
                             mask.ptr if mask is not None else None, mask.ld if mask is not None else 0)
            g = g_src
        g_dns = gbuf(self.dns)
        ops.conv2d_wgrad(plan, self.d_c2d, self.dns.ptr, g.ptr, g.ld, self._G("conv2d/kernel"), self._G("conv2d/bias"))
        ops.conv2d_dgrad(plan, self.d_c2d, g.ptr, g.ld, self._P("conv2d/kernel"), g_dns.ptr, None, 0, self.dns.ptr,
                         self.dns.ld)
        D0 = self.DENSE_CH
        npix = HEAD[0] * HEAD[1]
        nd = npix * D0
        g_dns1 = z(N, nd)
        ops.grad_slice(plan, g_dns.ptr, g_dns.ld, g_dns1, D0, None, 0, N * npix, D0)
        g_z = z(N, Zn)
        ops.conv2d_wgrad(plan, self.d_dense, self.zbuf, g_dns1, nd, self._G("dense/kernel"), self._G("dense/bias"))
        ops.conv2d_dgrad(plan, self.d_dense, g_dns1, nd, self._P("dense/kernel"), g_z)
        g_ext = z(N, 2 * Zn)
        ops.latent_linear_bwd(plan, self.ext, self.eps, g_z, Zn, kl_weight, g_ext, N, Zn)
        g_heads = z(N, 2 * Zn)
        ops.grad_slice(plan, g_ext, 2 * Zn, g_heads, 2 * Zn, None, 0, N, Zn)
        ops.softplus_bwd(plan, ops.Ptr(self.heads_out, Zn), 2 * Zn, ops.Ptr(g_ext, Zn), 2 * Zn, ops.Ptr(g_heads, Zn), 2 * Zn,
                         N, Zn)
        ops.conv2d_wgrad(plan, self.d_heads, self.feat, g_heads, 2 * Zn, self._G("heads/kernel"), self._G("heads/bias"))
        self.g_feat = z(N, HEAD[0], HEAD[1], self.FEAT_LD)
        ops.conv2d_dgrad(plan, self.d_heads, g_heads, 2 * Zn, self._P("heads/kernel"), self.g_feat)
        self.feat_ld = self.FEAT_LD
        self._grad_bufs = dict(g_z=g_z, g_heads=g_heads, g_ext=g_ext)


class Unet2(_SplitBN):
    """models/unet_architecture_noconc2.py:49-109 (scope 'UNet')"""
    SCOPE, CIN, COUT, Z = "UNet", 3, 3, 1024
    WD, WD_ENC, WD_DEC = 7e-5, 7e-5, 7e-5
    ENC = [("1", 32, (3, 3), 3, "VALID"), ("2", 128, (3, 3), 2, "VALID"), ("3", 256, (2, 3), 3, "VALID"),
           ("5", 512, None, None, None)]
    DENSE_CH, C2D = 50, 512
    DEC = [("6", 256, (3, 4), 3, ("6", "7")), ("8", 128, (4, 3), 2, ("8", "9")), ("10", 32, (5, 4), 3, ("10", "11"))]

    def __init__(self, input_shape=None, precision="split"):
        super(Unet2, self).__init__(input_shape or [224, 298, 3], precision)


class UNetSound22(_SplitBN):
    """models/unet_sound22.py:52-117 (scope 'UNetAudio')"""
    SCOPE, CIN, COUT, Z = "UNetAudio", 1, 1, 256
    WD, WD_ENC, WD_DEC = 6e-5, 6e-5, 8e-5
    ENC = [("1", 16, (3, 3), 2, "VALID"), ("2", 16, (3, 3), 2, "SAME"), ("3", 64, (3, 3), 2, "SAME"),
           ("4", 128, (3, 3), 2, "SAME"), ("5", 128, None, None, None)]
    DENSE_CH, C2D = 10, 128
    DEC = [("6", 128, (2, 2), 2, ("6", "7")), ("8", 64, (2, 2), 2, ("8", "9")), ("10", 16, (2, 2), 2, ("10", "11")),
           ("12", 16, (3, 3), 2, ("12", "13"))]

    def __init__(self, input_shape=None, precision="split"):
        super(UNetSound22, self).__init__(input_shape or [193, 257, 1], precision)
