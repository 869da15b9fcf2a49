"""`UNetAc`: the acoustic-image generator (scope 'UNetAcRes'), MI355X-native.

Mirrors the reference's model protocol (models/unet_acresnet.py:21-41,104-134): `scope`,
`init_model`, `_build_model(acoustic_images, resnetfeature)` setting `mean`, `std`, `output`,
`network`, `train_vars`; and its three variants (models/unet_acresnet0skip.py:85,185-196,
models/unet_acresnet2skip.py:82-83) through `num_skip`.  `embedding=True` is the plain auto-encoder
of :63-71 (min-max-normalised code, no std / KL).

MI355X design: tf.concat never materialises — producers write straight into channel slices of the
concat buffers (deconv output and skip `conv1` into one [N,36,48,256] tensor; the two min-max maps
into the [N,12,16,148] latent-head input); the two 12x16-VALID heads are ONE split-K GEMM over the
concatenated [mean|std] weight; every backward kernel emits the PRE-activation gradient of the
layer below (ReLU mask applied in the data-gradient epilogue from the saved activation), so no
standalone ReLU / ReLU-grad / concat / slice pass touches HBM.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, Ptr
from .params import FusedHeads, Var, up4
from .session import get_default_session
from .vision import load_state_file

Z = 150


class Act(object):
    """An NHWC activation: logical channels C inside a buffer with pixel stride ld at channel offset."""

    def __init__(self, t, N, H, W, C, ld=None, off=0):
        self.t, self.N, self.H, self.W, self.C = t, N, H, W, C
        self.ld = up4(C) if ld is None else ld
        self.off = off

    @property
    def ptr(self):
        return Ptr(self.t, self.off)

    @property
    def Cp(self):
        return up4(self.C)

    @property
    def pixels(self):
        return self.N * self.H * self.W


class UNetAc(object):

    def __init__(self, input_shape=None, num_frames=12, embedding=False, num_skip=1, precision="split", side_lane=True):
        """precision: "split" = the large convs on the split-MFMA kernels (fp32-class results), "f32" = every
        conv on the exact-f32 MFMA kernel"""
        assert precision in ("split", "f32")
        self.precision = precision
        self.split_min_rows = 16384      # below this the f32 kernel (64x64 tiles + split-K) fills the chip better
        # backward: weight gradients on the plan's side lane (a second HIP stream), beside the data gradients
        self.side_lane = bool(side_lane)
        self._wsplit_bufs = {}
        self.scope = 'UNetAcRes'
        self.num_frames = num_frames
        self.height = input_shape[0]
        self.width = input_shape[1]
        self.channels = input_shape[2]
        # if true, no vae: use auto encoder
        self.embedding = bool(embedding)
        self.num_skip = int(num_skip)
        assert self.num_skip in (0, 1, 2)
        self.session = None

    # ---- variables ------------------------------------------------------------------------------------
    def _conv_specs(self):
        ns = self.num_skip
        return OrderedDict([
            ("final", (3, 3, 64, 12)),
            ("layer7/conv_2", (3, 3, 64, 64)), ("layer7/conv_1", (3, 3, 128, 64)),
            ("layer6/conv_2", (3, 3, 128, 128)), ("layer6/conv_1", (3, 3, 128 if ns == 0 else 256, 128)),
            ("layer5/conv_2", (3, 3, 128, 128)), ("layer5/conv_1", (3, 3, 128, 128)),
            ("layer4/conv_2", (3, 3, 128, 128)), ("layer4/conv_1", (3, 3, 266 if ns == 2 else 133, 128)),
            ("conv2d", (3, 3, 12, 133)),
            ("layer2/conv_2", (3, 3, 133, 133)), ("layer2/conv_1", (3, 3, 128, 133)),
            ("layer1/pool_2", (3, 3, 128, 128)), ("layer1/conv_2", (3, 3, 128, 128)),
            ("layer1/conv_1", (3, 3, 12, 128)),
        ])

    def _register(self, store):
        """Registration order = the order gradients become final in the backward pass, so the flat
        gradient buffer fills front to back and can be all-reduced in contiguous buckets."""
        s = self.scope
        specs = self._conv_specs()

        def conv(name):
            store.add(Var("%s/%s/kernel" % (s, name), specs[name], "conv", "train"))
            store.add(Var("%s/%s/bias" % (s, name), (specs[name][3],), "vec", "train"))

        for name in ("final", "layer7/conv_2", "layer7/conv_1", "layer6/conv_2", "layer6/conv_1"):
            conv(name)
        store.add(Var(s + "/upsample_1/kernel", (2, 2, 128, 128), "deconv", "train"))
        store.add(Var(s + "/upsample_1/bias", (128,), "vec", "train"))
        for name in ("layer5/conv_2", "layer5/conv_1", "layer4/conv_2", "layer4/conv_1", "conv2d"):
            conv(name)
        store.add(Var(s + "/dense/kernel", (Z, 12 * 16 * 12), "dense", "train"))
        store.add(Var(s + "/dense/bias", (12 * 16 * 12,), "vec", "train"))
        self.heads = FusedHeads(s, 145, Z, not self.embedding)
        store.add_fused(self.heads)
        for name in ("layer2/conv_2", "layer2/conv_1", "layer1/pool_2", "layer1/conv_2", "layer1/conv_1"):
            conv(name)

    def init_model(self, session, checkpoint_file):
        """Initialise every variable of the scope from a TF-named state (models/unet_acresnet.py:33-41)."""
        state = load_state_file(checkpoint_file)
        store = (session or self.session).store
        return store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def initialize(self, seed=1239):
        """xavier_initializer() kernels (models/unet_acresnet.py:165,179,216), zero biases"""
        g = torch.Generator().manual_seed(seed)
        state = OrderedDict()

        def xav(shape, fin, fout):
            lim = np.sqrt(6.0 / (fin + fout))
            return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()

        for name, (kh, kw, cin, cout) in self._conv_specs().items():
            state["%s/%s/kernel" % (self.scope, name)] = xav((kh, kw, cin, cout), kh * kw * cin, kh * kw * cout)
            state["%s/%s/bias" % (self.scope, name)] = torch.zeros(cout)
        heads = ["mean"] + ([] if self.embedding else ["std"])
        for h in heads:
            state["%s/%s/kernel" % (self.scope, h)] = xav((12, 16, 145, Z), 12 * 16 * 145, 12 * 16 * Z)
            state["%s/%s/bias" % (self.scope, h)] = torch.zeros(Z)
        state[self.scope + "/dense/kernel"] = xav((Z, 2304), Z, 2304)
        state[self.scope + "/dense/bias"] = torch.zeros(2304)
        state[self.scope + "/upsample_1/kernel"] = xav((2, 2, 128, 128), 4 * 128, 4 * 128)
        state[self.scope + "/upsample_1/bias"] = torch.zeros(128)
        self.session.store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    # ---- graph ------------------------------------------------------------------------------------------
    def _build_model(self, acoustic_images, resnetfeature, session=None, eps=None):
        """acoustic_images: device buffer [N,36,48,12] (the tiled MFCC map); resnetfeature: device buffer
        [N,12,16,12] (ResNet50Model.output); eps: device buffer [N,150] standing where the reference
        samples tf.random_normal (models/unet_acresnet.py:77)."""
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        N = acoustic_images.shape[0]
        assert tuple(acoustic_images.shape[1:]) == (self.height, self.width, self.channels)
        assert tuple(resnetfeature.shape) == (N, 12, 16, 12)
        self.N = N
        z = sess.zeros
        ns = self.num_skip
        H, W = self.height, self.width
        h, w = H // 3, W // 3
        self.inp = Act(acoustic_images, N, H, W, 12)
        self.feat = Act(resnetfeature, N, h, w, 12)
        self.eps = eps if eps is not None else z(N, Z)
        self.c11 = Act(z(N, H, W, 128), N, H, W, 128)
        if ns >= 1:
            self.up1cat = z(N, H, W, 256)
            self.up = Act(self.up1cat, N, H, W, 128, 256, 0)
            self.conv1 = Act(self.up1cat, N, H, W, 128, 256, 128)
            self.l6in = Act(self.up1cat, N, H, W, 256)
        else:
            self.up = Act(z(N, H, W, 128), N, H, W, 128)
            self.conv1 = Act(z(N, H, W, 128), N, H, W, 128)
            self.l6in = self.up
        self.pool1 = Act(z(N, h, w, 128), N, h, w, 128)
        self.c21 = Act(z(N, h, w, 136), N, h, w, 133)
        if ns == 2:
            self.cat4 = z(N, h, w, 268)
            self.net = Act(self.cat4, N, h, w, 133, 268, 0)
            self.conv2_0 = Act(self.cat4, N, h, w, 133, 268, 133)
            self.l4in = Act(self.cat4, N, h, w, 266)
        else:
            self.net = Act(z(N, h, w, 136), N, h, w, 133)
            self.conv2_0 = Act(z(N, h, w, 136), N, h, w, 133)
            self.l4in = self.net
        self.cat145 = z(N, h, w, 148)
        self.mm_a, self.mm_b, self.mm_z = z(N, 4), z(N, 4), z(N, 4)
        self.hn = self.heads.ncols
        self.heads_out = z(N, self.hn)
        self.zbuf = z(N, 152)
        self.sigma = z(N, Z)
        self.kl = z(N)
        self.dns = Act(z(N, h, w, 12), N, h, w, 12)
        self.c41 = Act(z(N, h, w, 128), N, h, w, 128)
        self.conv4 = Act(z(N, h, w, 128), N, h, w, 128)
        self.c51 = Act(z(N, h, w, 128), N, h, w, 128)
        self.conv5 = Act(z(N, h, w, 128), N, h, w, 128)
        self.c61 = Act(z(N, H, W, 128), N, H, W, 128)
        self.conv6 = Act(z(N, H, W, 128), N, H, W, 128)
        self.c71 = Act(z(N, H, W, 64), N, H, W, 64)
        self.conv7 = Act(z(N, H, W, 64), N, H, W, 64)
        self.yhat = Act(z(N, H, W, 12), N, H, W, 12)

        self.plan_fwd = sess.new_plan()
        self._record_forward(self.plan_fwd)

        self.mean = self.heads_out[:, :Z]
        if not self.embedding:
            self.std = self.sigma
        else:
            self.mean = self.zbuf[:, :Z]
        self.output = self.yhat.t
        self.network = OrderedDict(input=acoustic_images, is_training=None, keep_prob=None,
                                   features=self.cat145[..., :145])
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/")]

    # weight / grad pointers (resolved once the flat buffers exist)
    def _P(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.p(self.scope + "/" + name))

    def _G(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.g(self.scope + "/" + name))

    def _desc(self, x, K, stride=1, y=None, act=ACT_NONE):
        return ops.conv_desc(x.N, x.H, x.W, x.Cp if x.off == 0 and x.ld == x.Cp else x.C, K, 3, 3, stride, "SAME",
                             ldx=x.ld, ldy=(y.ld if y is not None else up4(K)), ldw=up4(K), act=act)

    def _use_split(self, d):
        """big stride-1 convs of the generator run on the split-MFMA kernels (forward f16x3, data gradient
        bf16x3); the 12x16 layers (48 row tiles: they need split-K) and the 12/133-channel layers stay f32"""
        return (self.precision == "split" and d.stride == 1 and d.C % 32 == 0 and d.K % 32 == 0 and
                d.N * d.OH * d.OW >= self.split_min_rows)

    def _wsplit(self, name, nbytes, kind):
        key = (name, kind)
        if key not in self._wsplit_bufs:
            self._wsplit_bufs[key] = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.session.device)
        return self._wsplit_bufs[key]

    def _prepare_job(self, plan, d, name, image, mode):
        """the kernels change every step: ONE launch at the head of the forward plan re-splits them all (forward
        images and, once the backward is recorded, the flipped / transposed data-gradient images)"""
        if getattr(self, "_prep_jobs", None) is None:
            # a plan other than the forward one (should not happen): fall back to a launch in place
            (ops.conv2d_split3_prepare_dgrad if mode else ops.conv2d_split3_prepare)(plan, d, self._P(name + "/kernel"),
                                                                                      image)
            return
        self._prep_jobs.add(d, self._P(name + "/kernel"), image, mode)

    def _conv(self, plan, name, x, y, stride=1, act=ACT_RELU):
        d = self._desc(x, y.C, stride, y, act)
        self._descs[name] = (d, x, y)
        if self._use_split(d):
            # the kernel changes every step: re-split it right before use (one tiny launch)
            ws = self._wsplit(name, ops.conv2d_split3_weight_bytes(d), "fwd")
            self._prepare_job(plan, d, name, ws, 0)
            ops.conv2d_fwd_split3(plan, d, x.ptr, ws, y.ptr, bias=self._P(name + "/bias"))
        else:
            ops.conv2d_fwd(plan, d, x.ptr, self._P(name + "/kernel"), self._P(name + "/bias"), y.ptr)

    def _record_forward(self, plan):
        N = self.N
        self._descs = {}
        h, w = self.pool1.H, self.pool1.W
        self._prep_jobs = None
        if self.precision == "split":
            self._prep_jobs = ops.PrepareJobs()
            ops.conv2d_split3_prepare_multi(plan, self._prep_jobs)
        self._conv(plan, "layer1/conv_1", self.inp, self.c11)
        self._conv(plan, "layer1/conv_2", self.c11, self.conv1)
        self._conv(plan, "layer1/pool_2", self.conv1, self.pool1, stride=3)
        self._conv(plan, "layer2/conv_1", self.pool1, self.c21)
        self._conv(plan, "layer2/conv_2", self.c21, self.conv2_0)
        ops.minmax_fwd(plan, self.conv2_0.ptr, self.conv2_0.ld, self.cat145, 148, self.mm_a, N, h * w, 133)
        ops.minmax_fwd(plan, self.feat.ptr, 12, Ptr(self.cat145, 133), 148, self.mm_b, N, h * w, 12)
        # mean | std heads: one [N, 28416] x [28416, 300] GEMM (a 12x16 VALID conv over the whole map)
        kin = h * w * 148
        self.d_heads = ops.conv_desc(N, 1, 1, kin, self.hn if not self.embedding else Z, 1, 1, 1, "VALID",
                                     ldx=kin, ldy=self.hn, ldw=self.hn)
        ops.conv2d_fwd(plan, self.d_heads, self.cat145, self._P("heads/kernel"), self._P("heads/bias"),
                       self.heads_out)
        if self.embedding:
            ops.minmax_fwd(plan, self.heads_out, self.hn, self.zbuf, 152, self.mm_z, N, 1, Z)
        else:
            ops.latent_fwd(plan, self.heads_out, self.eps, self.zbuf, 152, self.sigma, self.kl, N, Z)
        self.d_dense = ops.conv_desc(N, 1, 1, 152, 2304, 1, 1, 1, "VALID", ldx=152, ldy=2304, ldw=2304,
                                     act=ACT_RELU)
        ops.conv2d_fwd(plan, self.d_dense, self.zbuf, self._P("dense/kernel"), self._P("dense/bias"), self.dns.t)
        self._conv(plan, "conv2d", self.dns, self.net)
        self._conv(plan, "layer4/conv_1", self.l4in, self.c41)
        self._conv(plan, "layer4/conv_2", self.c41, self.conv4)
        self._conv(plan, "layer5/conv_1", self.conv4, self.c51)
        self._conv(plan, "layer5/conv_2", self.c51, self.conv5)
        self.d_up = ops.deconv_desc(N, h, w, 128, 128, 2, 2, 3, ldx=128, ldy=self.up.ld, ldw=128)
        ops.deconv_fwd(plan, self.d_up, self.conv5.ptr, self._P("upsample_1/kernel"), self._P("upsample_1/bias"),
                       self.up.ptr)
        self._conv(plan, "layer6/conv_1", self.l6in, self.c61)
        self._conv(plan, "layer6/conv_2", self.c61, self.conv6)
        self._conv(plan, "layer7/conv_1", self.conv6, self.c71)
        self._conv(plan, "layer7/conv_2", self.c71, self.conv7)
        self._conv(plan, "final", self.conv7, self.yhat, act=ACT_SIGMOID)

    # ---- backward ---------------------------------------------------------------------------------------
    def record_backward(self, plan, g_logit, g_feat, kl_weight, on_ready=None):
        """g_logit: gradient w.r.t. the PRE-sigmoid output [N,36,48,12] (from recon_loss);
        g_feat: where to leave the gradient w.r.t. resnetfeature [N,12,16,12];
        kl_weight: FLAGS.latent_loss / N (trainer/mfcctrainer.py:56-59);
        on_ready(name): called at record time right after the kernels that finish layer `name`'s
        weight gradient were ADDED (lets the trainer place its gradient all-reduce hooks); weight gradients may still
        be in flight on the side lane there: a consumer calls plan.join() first.  The plan ends joined."""
        on_ready = on_ready or (lambda name: None)
        N = self.N
        z = self.session.zeros
        H, W, h, w = self.height, self.width, self.pool1.H, self.pool1.W
        ns = self.num_skip

        def gbuf(a, full=False):
            return Act(z(a.N, a.H, a.W, up4(a.C)), a.N, a.H, a.W, a.C)

        def back(name, gy, dx=None, mask=None, res=None):
            """weight/bias gradient of conv `name`, and its data gradient into dx (if given)"""
            d, x, y = self._descs[name]
            wg = ops.conv2d_wgrad_split3 if (self.precision == "split" and d.K % 64 == 0) else ops.conv2d_wgrad
            # the weight gradients run on the side lane (a second stream, in layer order) beside the chain of data
            # gradients on the main stream: a weight gradient only needs its layer's gy (fork = the side lane waits
            # for it), every gradient buffer is written once, before the fork that publishes it, and nothing on the
            # main stream waits for a weight gradient until a consumer of the parameter gradients joins
            beside = dx is not None and self.side_lane
            if beside:
                plan.fork()
            wg(plan, d, x.ptr, gy.ptr, gy.ld, self._G(name + "/kernel"), self._G(name + "/bias"), side=beside)
            if dx is not None and self._use_split(d):
                wt = self._wsplit(name, ops.conv2d_split3_dgrad_weight_bytes(d), "dgrad")
                self._prepare_job(plan, d, name, wt, 1)
                ops.conv2d_dgrad_split3(plan, d, gy.ptr, gy.ld, wt, dx.ptr,
                                        res.ptr if res is not None else None, res.ld if res is not None else 0,
                                        mask.ptr if mask is not None else None, mask.ld if mask is not None else 0,
                                        lddx=dx.ld)
            elif dx is not None:
                ops.conv2d_dgrad(plan, d, gy.ptr, gy.ld, self._P(name + "/kernel"), dx.ptr,
                                 res.ptr if res is not None else None, res.ld if res is not None else 0,
                                 mask.ptr if mask is not None else None, mask.ld if mask is not None else 0,
                                 lddx=dx.ld)
            on_ready(name)

        g_final = Act(g_logit, N, H, W, 12)
        g_conv7, g_c71, g_conv6, g_c61 = gbuf(self.conv7), gbuf(self.c71), gbuf(self.conv6), gbuf(self.c61)
        back("final", g_final, g_conv7, mask=self.conv7)
        back("layer7/conv_2", g_conv7, g_c71, mask=self.c71)
        back("layer7/conv_1", g_c71, g_conv6, mask=self.conv6)
        back("layer6/conv_2", g_conv6, g_c61, mask=self.c61)
        if ns >= 1:
            g_l6in_t = z(N, H, W, 256)
            g_l6in = Act(g_l6in_t, N, H, W, 256)
            g_up = Act(g_l6in_t, N, H, W, 128, 256, 0)
            g_skip = Act(g_l6in_t, N, H, W, 128, 256, 128)
        else:
            g_l6in = g_up = gbuf(self.up)
            g_skip = None
        back("layer6/conv_1", g_c61, g_l6in)           # no mask: deconv output has no activation
        g_conv5, g_c51, g_conv4, g_c41 = gbuf(self.conv5), gbuf(self.c51), gbuf(self.conv4), gbuf(self.c41)
        if self.side_lane:
            plan.fork()
        ops.deconv_wgrad(plan, self.d_up, self.conv5.ptr, g_up.ptr, g_up.ld, self._G("upsample_1/kernel"),
                         self._G("upsample_1/bias"), side=self.side_lane)
        ops.deconv_dgrad(plan, self.d_up, g_up.ptr, g_up.ld, self._P("upsample_1/kernel"), g_conv5.ptr,
                         self.conv5.ptr, self.conv5.ld)
        back("layer5/conv_2", g_conv5, g_c51, mask=self.c51)
        back("layer5/conv_1", g_c51, g_conv4, mask=self.conv4)
        back("layer4/conv_2", g_conv4, g_c41, mask=self.c41)
        g_net = gbuf(self.net)
        if ns == 2:
            g_cat4 = Act(z(N, h, w, 268), N, h, w, 266)
            back("layer4/conv_1", g_c41, g_cat4)
            ops.grad_slice(plan, g_cat4.ptr, 268, g_net.ptr, g_net.ld, self.net.ptr, self.net.ld, N * h * w, 133)
        else:
            back("layer4/conv_1", g_c41, g_net, mask=self.net)
        g_dns = gbuf(self.dns)
        back("conv2d", g_net, g_dns, mask=self.dns)
        # dense 150 -> 2304
        g_z = z(N, 152)
        ops.conv2d_wgrad(plan, self.d_dense, self.zbuf, g_dns.t, 2304, self._G("dense/kernel"), self._G("dense/bias"))
        ops.conv2d_dgrad(plan, self.d_dense, g_dns.t, 2304, self._P("dense/kernel"), g_z)
        on_ready("dense")
        g_heads = z(N, self.hn)
        if self.embedding:
            ops.minmax_bwd(plan, self.heads_out, self.hn, g_z, 152, self.mm_z, g_heads, self.hn, N, 1, Z)
        else:
            ops.latent_bwd(plan, self.heads_out, self.eps, self.sigma, g_z, 152, kl_weight, g_heads, N, Z)
        g_cat145 = z(N, h, w, 148)
        if self.side_lane:
            plan.fork()
        ops.conv2d_wgrad(plan, self.d_heads, self.cat145, g_heads, self.hn, self._G("heads/kernel"),
                         self._G("heads/bias"), side=self.side_lane)
        ops.conv2d_dgrad(plan, self.d_heads, g_heads, self.hn, self._P("heads/kernel"), g_cat145)
        on_ready("heads")
        # through the two min-max normalisations
        g_conv2_0 = gbuf(self.conv2_0)
        if ns == 2:
            ops.grad_slice(plan, Ptr(g_cat4.t, 133), 268, g_conv2_0.ptr, g_conv2_0.ld, None, 0, N * h * w, 133)
        ops.minmax_bwd(plan, self.conv2_0.ptr, self.conv2_0.ld, g_cat145, 148, self.mm_a, g_conv2_0.ptr,
                       g_conv2_0.ld, N, h * w, 133, accumulate=(ns == 2), mask_relu=True)
        ops.minmax_bwd(plan, self.feat.ptr, 12, Ptr(g_cat145, 133), 148, self.mm_b, g_feat, 12, N, h * w, 12)
        g_c21, g_pool1, g_conv1, g_c11 = gbuf(self.c21), gbuf(self.pool1), gbuf(self.conv1), gbuf(self.c11)
        back("layer2/conv_2", g_conv2_0, g_c21, mask=self.c21)
        back("layer2/conv_1", g_c21, g_pool1, mask=self.pool1)
        back("layer1/pool_2", g_pool1, g_conv1, mask=self.conv1, res=g_skip)
        back("layer1/conv_2", g_conv1, g_c11, mask=self.c11)
        back("layer1/conv_1", g_c11, None)
        plan.join()           # every parameter gradient of the generator is complete from here on
        self._grad_bufs = dict(g_conv7=g_conv7, g_l6in=g_l6in, g_conv5=g_conv5, g_net=g_net, g_z=g_z,
                               g_heads=g_heads, g_cat145=g_cat145, g_conv2_0=g_conv2_0, g_conv1=g_conv1,
                               g_c11=g_c11)
