"""Oracle: the joint-latent step of trainer/trainermulti.py:32-96 (FLAGS.jointmvae, neither `fusion` nor `onlyaudiovideo`):
three split VAEs - `Unet2` (models/unet_architecture_noconc2.py), `UNetSound22` (models/unet_sound22.py), `UNetAc2`
(models/unet_noconc2.py) - whose ENCODERS (`_build_network`) feed the per-pixel fusion MLP `Jointmvae`
(models/multimodal.py:287-347) and whose DECODERS (`_build_model(f)`: two 12x16 VALID heads, std = softplus, z = mean +
std * eps, dense, conv, transposed convs, conv_conv blocks, sigmoid) reconstruct every modality from the MLP's heads.
TEST INFRASTRUCTURE - see oracle/__init__.py.

    loss = sum_m MSE_m + sum_m Huber_m + mean_b(sum_m 0.5 * sum_j(mu^2 + s^2 - log(1e-8 + s^2) - 1)) / 1e6 + regularisers
(tf.losses.get_total_loss() collects the three MSEs, the three Hubers and the kernel regularisers of the video / audio
models; the KL is a SUM over the latent here, trainermulti.py:70-79), minimised over `Jointmvae`'s variables only; the
batch norms of the video / audio models run in training mode (is_training fed 1, trainermulti.py:323-328) and their
moving averages are updated by the step (update_ops).  The acoustic model has no batch norm and no regulariser.
Parity is unpinned at the TensorFlow boundary (no TF here, no fixtures in the reference)."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import multimodal, tfsem

BN_MOMENTUM, BN_EPS = 0.99, 1e-3

# enc row: (layer, filters, pool kernel (kh, kw), pool stride, pool padding) - pool None for the last block
# dec row: (upsample name, filters, kernel (kh, kw), stride, (block, block))
MODELS = {
    "Unet2": dict(scope="UNet", cin=3, input_hw=(224, 298), bn=True, wd_enc=7e-5, wd_dec=7e-5, Z=1024,
                  enc=[("1", 32, (3, 3), 3, "VALID"), ("2", 128, (3, 3), 2, "VALID"), ("3", 256, (2, 3), 3, "VALID"),
                       ("5", 512, None, None, None)],
                  dense_ch=50, c2d=512,
                  dec=[("6", 256, (3, 4), 3, ("6", "7")), ("8", 128, (4, 3), 2, ("8", "9")), ("10", 32, (5, 4), 3, ("10", "11"))],
                  final=(1, 1), cout=3),
    "UNetSound22": dict(scope="UNetAudio", cin=1, input_hw=(193, 257), bn=True, wd_enc=6e-5, wd_dec=8e-5, Z=256,
                        enc=[("1", 16, (3, 3), 2, "VALID"), ("2", 16, (3, 3), 2, "SAME"), ("3", 64, (3, 3), 2, "SAME"),
                             ("4", 128, (3, 3), 2, "SAME"), ("5", 128, None, None, None)],
                        dense_ch=10, c2d=128,
                        dec=[("6", 128, (2, 2), 2, ("6", "7")), ("8", 64, (2, 2), 2, ("8", "9")),
                             ("10", 16, (2, 2), 2, ("10", "11")), ("12", 16, (3, 3), 2, ("12", "13"))],
                        final=(1, 1), cout=1),
    "UNetAc2": dict(scope="UNetAcoustic", cin=12, input_hw=(36, 48), bn=False, wd_enc=0.0, wd_dec=0.0, Z=150,
                    enc=[("1", 128, (3, 3), 3, "SAME"), ("3", 133, None, None, None)],
                    dense_ch=12, c2d=133,
                    dec=[("1", 128, (2, 2), 3, ("4", "5"))],
                    final=(3, 3), cout=12),
}
HEAD = (12, 16)


def feature_channels(model):
    return MODELS[model]["enc"][-1][1]


def param_shapes(model):
    """TF variable name -> shape, encoder variables first; `decoder_start` = index of the first decoder variable"""
    cfg = MODELS[model]
    sc = cfg["scope"]
    s = OrderedDict()

    def conv(name, kh, kw, cin, cout):
        s["%s/%s/kernel" % (sc, name)] = (kh, kw, cin, cout)
        s["%s/%s/bias" % (sc, name)] = (cout,)

    def bn(name, c):
        if cfg["bn"]:
            for v in ("gamma", "beta", "moving_mean", "moving_variance"):
                s["%s/%s/%s" % (sc, name, v)] = (c,)

    def block(name, cin, F_):
        for i in (1, 2):
            conv("layer%s/conv_%d" % (name, i), 3, 3, cin if i == 1 else F_, F_)
            bn("layer%s/bn_%d" % (name, i), F_)

    cin = cfg["cin"]
    for name, F_, pool, stride, pad in cfg["enc"]:
        block(name, cin, F_)
        if pool is not None:
            conv("layer%s/pool_2" % name, pool[0], pool[1], F_, F_)
            bn("layer%s/bn_pool_2" % name, F_)
        cin = F_
    conv("mean", HEAD[0], HEAD[1], cin, cfg["Z"])
    conv("std", HEAD[0], HEAD[1], cin, cfg["Z"])
    nd = HEAD[0] * HEAD[1] * cfg["dense_ch"]
    s[sc + "/dense/kernel"] = (cfg["Z"], nd)
    s[sc + "/dense/bias"] = (nd,)
    conv("conv2d", 3, 3, cfg["dense_ch"], cfg["c2d"])
    cin = cfg["c2d"]
    for name, F_, k, stride, blocks in cfg["dec"]:
        s["%s/upsample_%s/kernel" % (sc, name)] = (k[0], k[1], F_, cin)
        s["%s/upsample_%s/bias" % (sc, name)] = (F_,)
        for b in blocks:
            block(b, F_, F_)
        cin = F_
    conv("final", cfg["final"][0], cfg["final"][1], cin, cfg["cout"])
    return s


def is_encoder_var(model, name):
    cfg = MODELS[model]
    leaf = name[len(cfg["scope"]) + 1:]
    return any(leaf.startswith("layer%s/" % e[0]) for e in cfg["enc"])


def init_params(model, seed=1251, dtype=torch.float32, bias_std=0.0, bn_jitter=0.0):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, shape in param_shapes(model).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "bias":
            p[name] = (bias_std * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        elif leaf == "gamma":
            p[name] = (1.0 + bn_jitter * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        elif leaf == "beta":
            p[name] = (bn_jitter * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        elif leaf == "moving_mean":
            p[name] = torch.zeros(*shape, dtype=dtype)
        elif leaf == "moving_variance":
            p[name] = torch.ones(*shape, dtype=dtype)
        elif name.endswith("dense/kernel"):
            p[name] = tfsem.xavier_uniform(g, shape, shape[0], shape[1], dtype)
        elif "upsample" in name:
            kh, kw, cout, cin = shape
            p[name] = tfsem.xavier_uniform(g, shape, kh * kw * cin, kh * kw * cout, dtype)
        else:
            kh, kw, cin, cout = shape
            p[name] = tfsem.xavier_uniform(g, shape, kh * kw * cin, kh * kw * cout, dtype)
    return p


class _Net(object):
    def __init__(self, model, p, training, relu_masks=None):
        self.cfg, self.p, self.training = MODELS[model], p, training
        self.sc = self.cfg["scope"]
        self.new_stats = OrderedDict()
        self.masks = OrderedDict()
        self.relu_masks = relu_masks

    def relu(self, name, t):
        if self.relu_masks is not None and name in self.relu_masks:
            return t * self.relu_masks[name].to(t.dtype).reshape(t.shape)
        y = torch.relu(t)
        self.masks[name] = y > 0
        return y

    def cbr(self, name, bnname, t, stride=1, padding="SAME"):
        p, sc = self.p, self.sc
        t = tfsem.conv2d(t, p["%s/%s/kernel" % (sc, name)], p["%s/%s/bias" % (sc, name)], stride, padding)
        if self.cfg["bn"]:
            bb = "%s/%s/" % (sc, bnname)
            t, mm, mv, _, _ = tfsem.batch_norm(t, p[bb + "gamma"], p[bb + "beta"], p[bb + "moving_mean"],
                                               p[bb + "moving_variance"], self.training, BN_MOMENTUM, BN_EPS)
            self.new_stats[bb + "moving_mean"], self.new_stats[bb + "moving_variance"] = mm, mv
        return self.relu(name, t)

    def block(self, name, t):
        for i in (1, 2):
            t = self.cbr("layer%s/conv_%d" % (name, i), "layer%s/bn_%d" % (name, i), t)
        return t

    def encoder(self, x):
        """`_build_network`: the feature map [N,12,16,C]"""
        net = x
        for name, F_, pool, stride, pad in self.cfg["enc"]:
            net = self.block(name, net)
            if pool is not None:
                net = self.cbr("layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, net, stride, pad)
        assert tuple(net.shape[1:3]) == HEAD, net.shape
        return net

    def decoder(self, f, eps):
        """`_build_network2(f)`"""
        p, sc, cfg = self.p, self.sc, self.cfg
        N = f.shape[0]
        mean = tfsem.conv2d(f, p[sc + "/mean/kernel"], p[sc + "/mean/bias"], 1, "VALID").reshape(N, -1)
        std = F.softplus(tfsem.conv2d(f, p[sc + "/std/kernel"], p[sc + "/std/bias"], 1, "VALID").reshape(N, -1))
        z = mean + std * eps
        net = self.relu("dense", z @ p[sc + "/dense/kernel"] + p[sc + "/dense/bias"]).reshape(N, HEAD[0], HEAD[1],
                                                                                              cfg["dense_ch"])
        net = self.relu("conv2d", tfsem.conv2d(net, p[sc + "/conv2d/kernel"], p[sc + "/conv2d/bias"], 1, "SAME"))
        for name, F_, k, stride, blocks in cfg["dec"]:
            net = tfsem.conv2d_transpose_valid(net, p["%s/upsample_%s/kernel" % (sc, name)],
                                               p["%s/upsample_%s/bias" % (sc, name)], stride)
            for b in blocks:
                net = self.block(b, net)
        out = torch.sigmoid(tfsem.conv2d(net, p[sc + "/final/kernel"], p[sc + "/final/bias"], 1, "SAME"))
        assert tuple(out.shape[1:3]) == cfg["input_hw"], out.shape
        return dict(output=out, mean=mean, std=std, z=z)


def regulariser(model, p):
    cfg = MODELS[model]
    tot = 0.0
    for n, w in p.items():
        if n.endswith("/kernel") and ("/layer" in n or "/upsample_" in n):
            wd = cfg["wd_enc"] if is_encoder_var(model, n) else cfg["wd_dec"]
            if wd:
                tot = tot + tfsem.l2_regularizer(w, wd)
    return tot


ORDER = (("ac", "UNetAc2"), ("video", "Unet2"), ("audio", "UNetSound22"))     # tf.concat order of Jointmvae's inputs


MODES = ("all", "fusion", "onlyaudiovideo")
# the fusion MLP each mode TRAINS and the modalities it reads (trainer/trainermulti.py:50-53,98; main.py:194-201)
MODE_MLP = {"all": ("Jointmvae", ("ac", "video", "audio")), "fusion": ("JointTwomvae2", ("video", "audio")),
            "onlyaudiovideo": ("JointTwomvae", ("video", "audio"))}


def joint_forward(params, pj, batch, eps, relu_masks=None, mode="all", pj0=None, moddrop_on=None):
    """params {modality: model params}, pj: the trained fusion MLP's params, batch / eps {modality: tensor}.
    mode "all": Jointmvae on the three feature maps, three decoders (trainermulti.py:53-81);
         "fusion": JointTwomvae2 on (video, audio), three decoders (:50-51);
         "onlyaudiovideo": the FROZEN Jointmvae `pj0` on the three maps gives the acoustic feature target, JointTwomvae on
                           (video, audio) feeds the acoustic decoder alone (:97-103).
    moddrop_on: FLAGS.moddrop - the scalar 0 / 1 that multiplies the acoustic feature map (`modDrop`, :446-450; the one
    random draw per step is an explicit input).
    -> (decoder outputs, features, heads of the trained MLP, new moving statistics, relu masks, heads of pj0 or None)"""
    rm = relu_masks or {}
    scope, reads = MODE_MLP[mode]
    nets = OrderedDict((m, _Net(model, params[m], True, rm.get(m))) for m, model in ORDER)
    feats = OrderedDict((m, nets[m].encoder(batch[m])) for m, _ in ORDER)
    if moddrop_on is not None:
        feats["ac"] = feats["ac"] * float(moddrop_on)
    heads, jm = multimodal.joint_forward(pj, scope, [feats[m] for m in reads], rm.get("joint"))
    heads0 = None
    decoded = [m for m, _ in ORDER]
    if mode == "onlyaudiovideo":
        with torch.no_grad():
            heads0, _ = multimodal.joint_forward(pj0, "Jointmvae", [feats[m] for m, _ in ORDER], rm.get("joint0"))
        decoded = ["ac"]
    outs = OrderedDict((m, nets[m].decoder(heads["output" + m], eps[m])) for m in decoded)
    stats = OrderedDict()
    masks = OrderedDict(joint=jm)
    for m, _ in ORDER:
        stats.update(nets[m].new_stats)
        masks[m] = nets[m].masks
    return outs, feats, heads, stats, masks, heads0


def joint_losses(params, batch, outs, mode="all", heads=None, heads0=None):
    """trainer/trainermulti.py:58-81 (all / fusion) and :100-106 (onlyaudiovideo: + the feature-matching MSE; the decoder
    regularisers of the video / audio models are not in that graph: only their `_build_network` halves were built)"""
    ls = OrderedDict()
    mse = hub = kl = 0.0
    for m, model in ORDER:
        if m not in outs:
            continue
        ls["mse_" + m] = tfsem.mse_loss(batch[m], outs[m]["output"])
        ls["huber_" + m] = tfsem.huber_loss(batch[m], outs[m]["output"])
        mu, sg = outs[m]["mean"], outs[m]["std"]
        kl = kl + 0.5 * (mu * mu + sg * sg - torch.log(1e-8 + sg * sg) - 1).sum(1)
        mse, hub = mse + ls["mse_" + m], hub + ls["huber_" + m]
    ls["mse"], ls["huber"] = mse, hub
    ls["latent"] = kl.mean(0) / 1000000
    if mode == "onlyaudiovideo":
        ls["feature"] = tfsem.mse_loss(heads0["outputac"], heads["outputac"])
        reg = 0.0
        for m, model in ORDER:
            cfg = MODELS[model]
            for n, w in params[m].items():
                if n.endswith("/kernel") and ("/layer" in n or "/upsample_" in n) and is_encoder_var(model, n) and cfg["wd_enc"]:
                    reg = reg + tfsem.l2_regularizer(w, cfg["wd_enc"])
        ls["reg"] = reg
        ls["loss"] = ls["latent"] + mse + hub + ls["feature"] + ls["reg"]
    else:
        ls["reg"] = sum(regulariser(model, params[m]) for m, model in ORDER)
        ls["loss"] = ls["latent"] + mse + hub + ls["reg"]
    return ls


class Oracle(object):
    def __init__(self, learning_rate=1e-4, dtype=torch.float32, params=None, joint_params=None, seed=1251, mode="all"):
        assert mode in MODES
        self.lr, self.dtype, self.mode = learning_rate, dtype, mode
        self.params = OrderedDict()
        for i, (m, model) in enumerate(ORDER):
            src = params[m] if params is not None else init_params(model, seed + i, bias_std=0.02, bn_jitter=0.1)
            self.params[m] = OrderedDict((k, v.to(dtype).clone()) for k, v in src.items())
        fc = OrderedDict((m, feature_channels(model)) for m, model in ORDER)
        scope, reads = MODE_MLP[mode]
        cin = sum(fc[m] for m in reads)
        src = joint_params if joint_params is not None else multimodal.joint_init_params(scope, cin, seed + 7, bias_std=0.02)
        self.pj = OrderedDict((k, v.to(dtype).clone()) for k, v in src.items())
        self.pj0 = None
        if mode == "onlyaudiovideo":       # the frozen three-modality MLP that provides the feature target
            src0 = multimodal.joint_init_params("Jointmvae", sum(fc.values()), seed + 8, bias_std=0.02)
            self.pj0 = OrderedDict((k, v.to(dtype).clone()) for k, v in src0.items())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.pj.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.pj.items())
        self.step = 0

    def state_dict(self):
        out = OrderedDict()
        for m, _ in ORDER:
            out.update(self.params[m])
        out.update(self.pj)
        if self.pj0 is not None:
            out.update(self.pj0)
        return out

    def train_step(self, batch, eps, apply=True, relu_masks=None, moddrop_on=None):
        dt = self.dtype
        batch = OrderedDict((k, v.to(dt)) for k, v in batch.items())
        eps = OrderedDict((k, v.to(dt)) for k, v in eps.items())
        pj = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.pj.items())
        outs, feats, heads, stats, masks, heads0 = joint_forward(self.params, pj, batch, eps, relu_masks, self.mode, self.pj0,
                                                                 moddrop_on)
        ls = joint_losses(self.params, batch, outs, self.mode, heads, heads0)
        names = list(pj)
        grads = torch.autograd.grad(ls["loss"], [pj[k] for k in names])
        g = OrderedDict(zip(names, grads))
        if apply:
            self.step += 1
            for k in names:
                self.pj[k], self.m[k], self.v[k] = tfsem.adam_tf1(self.pj[k], g[k], self.m[k], self.v[k], self.step, self.lr)
            for k, v in stats.items():
                for m, _ in ORDER:
                    if k in self.params[m]:
                        self.params[m][k] = v.detach()
        return dict(losses=OrderedDict((k, float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in ls.items()),
                    grads=g, outs=outs, feats=feats, heads=heads, heads0=heads0, new_stats=stats, masks=masks)


def synthetic_batch(n, seed=1234, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    batch, eps = OrderedDict(), OrderedDict()
    for m, model in ORDER:
        cfg = MODELS[model]
        H, W = cfg["input_hw"]
        batch[m] = torch.rand(n, H, W, cfg["cin"], generator=g, dtype=torch.float64).to(dtype)
        eps[m] = torch.randn(n, cfg["Z"], generator=g, dtype=torch.float64).to(dtype)
    return batch, eps
