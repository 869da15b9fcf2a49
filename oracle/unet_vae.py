"""Oracle: the conv-BN-ReLU U-Net VAEs `UNet` (RGB frames, scope 'UNet') and `UNetSound` (STFT
spectrograms, scope 'UNetAudio') and their train step.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows models/unet_architecture.py:46-206 and models/unet_sound.py:49-208 (same code, different layer
table) and the loss / optimiser recipe of trainer/trainer.py:58-87:
    loss = MSE(x, yhat) + Huber(x, yhat) + sum_k wd * ||W_k||^2 / 2 + mean_b(0.5 * mean_j(mu^2 + s^2 - log(1e-8 + s^2) - 1)) / 1e6
(tf.losses.get_total_loss() collects MSE, Huber and the kernel regularisers of conv_conv_pool / upconv_2D;
the second head is used as sigma directly; trainer.py:61 reads it as `model.std`).
TF semantics: asymmetric SAME padding, conv2d_transpose VALID sizes, tf.layers
batch_normalization (momentum .99, eps 1e-3, biased variance to normalise, unbiased into the moving
average), heads flatten (h, w, c), TF-1 Adam.  Parity is unpinned at the TensorFlow boundary (no TF here).
"""
from collections import OrderedDict

import torch

from . import tfsem

BN_MOMENTUM, BN_EPS = 0.99, 1e-3

# name -> (scope, input channels, weight decay, encoder table, head kernel, decoder table, final channels)
# encoder row: (layer, [F, F], pool kernel (kh, kw), pool padding) — pool None for the bottleneck
# decoder row: (layer, F of upsample, upsample kernel (kh, kw), skip layer, [F, F])
CONFIGS = {
    "UNet": dict(scope="UNet", cin=3, wd=7e-5, Z=128, head=(14, 18),
                 enc=[("1", 8, (3, 3), "SAME"), ("2", 32, (2, 3), "VALID"), ("3", 32, (3, 3), "SAME"),
                      ("4", 64, (2, 3), "VALID"), ("5", 128, None, None)],
                 dec=[("6", 64, (2, 3), "4"), ("7", 32, (2, 2), "3"), ("8", 32, (2, 3), "2"), ("9", 8, (2, 2), "1")],
                 cout=3, input_hw=(224, 298)),
    "UNetSound": dict(scope="UNetAudio", cin=1, wd=6e-5, Z=128, head=(6, 16),
                      enc=[("1", 8, (3, 3), "VALID"), ("2", 8, (3, 2), "VALID"), ("3", 32, (3, 3), "SAME"),
                           ("4", 64, (3, 3), "SAME"), ("5", 128, None, None)],
                      dec=[("6", 64, (2, 2), "4"), ("7", 32, (2, 2), "3"), ("8", 8, (3, 2), "2"), ("9", 8, (3, 3), "1")],
                      cout=1, input_hw=(99, 257)),
    # models/multimodal.py:139-285: the conv ENCODER of a spectrogram with two 12x16 VALID heads, std = softplus(.);
    # no decoder (`encoder_only`): it feeds the acoustic-image decoder of models/unet_z.py
    "AssociatorAudio": dict(scope="AssociatorAudio", cin=1, wd=8e-5, Z=150, head=(12, 16),
                            enc=[("1", 16, (3, 3), "VALID"), ("2", 16, (3, 3), "SAME"), ("3", 64, (3, 3), "SAME"),
                                 ("4", 128, (3, 3), "SAME"), ("5", 128, None, None)],
                            dec=[], cout=None, input_hw=(193, 257), heads=("mean", "std"), encoder_only=True),
}


def param_shapes(model="UNet"):
    """TF variable name -> shape (tf.layers names: conv_{i}/pool_2 kernels HWIO, bn_{i}/bn_pool_2,
    upsample_{n} [kh,kw,out,in], unnamed dense / conv2d)."""
    cfg = CONFIGS[model]
    sc = cfg["scope"]
    s = OrderedDict()

    def conv(name, kh, kw, cin, cout):
        s["%s/%s/kernel" % (sc, name)] = (kh, kw, cin, cout)
        s["%s/%s/bias" % (sc, name)] = (cout,)

    def bn(name, c):
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            s["%s/%s/%s" % (sc, name, v)] = (c,)

    cin = cfg["cin"]
    widths = {}
    for name, F_, pool, _ in cfg["enc"]:
        for i in (1, 2):
            conv("layer%s/conv_%d" % (name, i), 3, 3, cin if i == 1 else F_, F_)
            bn("layer%s/bn_%d" % (name, i), F_)
        if pool is not None:
            conv("layer%s/pool_2" % name, pool[0], pool[1], F_, F_)
            bn("layer%s/bn_pool_2" % name, F_)
        widths[name] = F_
        cin = F_
    hh, hw = cfg["head"]
    hn = cfg.get("heads", ("mean", "variance"))
    conv(hn[0], hh, hw, cin, cfg["Z"])
    conv(hn[1], hh, hw, cin, cfg["Z"])
    if cfg.get("encoder_only"):
        return s
    s[sc + "/dense/kernel"] = (cfg["Z"], hh * hw)
    s[sc + "/dense/bias"] = (hh * hw,)
    conv("conv2d", 3, 3, 1, 128)
    cin = 128
    for name, F_, k, skip in cfg["dec"]:
        s["%s/upsample_%s/kernel" % (sc, name)] = (k[0], k[1], F_, cin)
        s["%s/upsample_%s/bias" % (sc, name)] = (F_,)
        for i in (1, 2):
            conv("layer%s/conv_%d" % (name, i), 3, 3, F_ + widths[skip] if i == 1 else F_, F_)
            bn("layer%s/bn_%d" % (name, i), F_)
        cin = F_
    conv("final", 1, 1, cin, cfg["cout"])
    return s


def regularized(name):
    """kernels that carry kernel_regularizer=l2_regularizer(weight_decay): conv_conv_pool + upconv_2D"""
    return name.endswith("/kernel") and ("/layer" in name or "/upsample_" in name)


def trainable(name):
    return not (name.endswith("moving_mean") or name.endswith("moving_variance"))


def init_params(model="UNet", seed=1240, dtype=torch.float32, bias_std=0.0, bn_jitter=0.0):
    """Glorot-uniform kernels, zero biases, gamma 1 / beta 0, moving mean 0 / variance 1.  bias_std / bn_jitter > 0
    randomise biases and gamma/beta so that parity tests exercise those paths."""
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


def bf16_round(t):
    """round-to-nearest-even to bf16, value kept in t's dtype (what `(__bf16)v` / v_cvt_pk_bf16_f32 do)"""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def bf16_layer(n, oh, ow, cin, cout, stride):
    """the layers the product runs with bf16 operands under precision="bf16" (acimg/unet_vae.py `_use_split`)"""
    return stride == 1 and cin % 32 == 0 and cout % 32 == 0 and n * oh * ow >= 16384


class _Bf16Conv(torch.autograd.Function):
    """BASELINE configs[1] 'bf16': a stride-1 SAME conv whose GEMM operands are rounded to bf16 in ALL THREE products,
    fp32 (here: self.dtype) accumulation — forward conv(r(x), r(w)) + b; data gradient from (r(gy), r(w)); weight
    gradient from (r(x), r(gy)); bias gradient = sum r(gy) (it rides in the weight-gradient GEMM as a column of ones).
    Mirrors acimg_conv2d_fwd_bf16 / _dgrad_bf16 / _wgrad_bf16 (include/acimg.h)."""

    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = bf16_round(x), bf16_round(w)
        ctx.save_for_backward(xr, wr)
        return tfsem.conv2d(xr, wr, b, 1, "SAME")

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = bf16_round(gy)
        with torch.enable_grad():
            xl, wl = xr.detach().requires_grad_(True), wr.detach().requires_grad_(True)
            y = tfsem.conv2d(xl, wl, None, 1, "SAME")
            gx, gw = torch.autograd.grad(y, (xl, wl), gr)
        return gx, gw, gr.sum((0, 1, 2))


def forward(p, x, eps, model="UNet", training=True, relu_masks=None, bf16_operands=False):
    """x [N,H,W,cin], eps [N,Z].  Returns (out dict, new moving statistics dict).
    bf16_operands: the conv layers `bf16_layer` selects see bf16-rounded operands in forward and both gradients
    (the arithmetic of the product's precision="bf16"); everything else stays in x's dtype.
    relu_masks {layer name: 0/1 tensor}: the ReLU on/off pattern of the implementation under test, so that both
    sides differentiate the SAME piecewise-linear function (of the ~10^7 pre-activations of a batch a few hundred
    sit within fp32 rounding of zero; fp32 and fp64 evaluations of this very oracle differ by 1e-3..1e-2 in the
    gradients for that reason alone).  Forward values are unaffected beyond that rounding."""
    cfg = CONFIGS[model]
    sc = cfg["scope"]
    new_stats = OrderedDict()
    acts = OrderedDict()
    masks_out = OrderedDict()

    def relu(name, t):
        if relu_masks is not None and name in relu_masks:
            return t * relu_masks[name].to(t.dtype)
        y = torch.relu(t)
        masks_out[name] = (y > 0)
        return y

    def cbr(name, bnname, t, stride=1, padding="SAME"):
        k, b = p["%s/%s/kernel" % (sc, name)], p["%s/%s/bias" % (sc, name)]
        if bf16_operands and padding == "SAME" and bf16_layer(t.shape[0], t.shape[1], t.shape[2], k.shape[2], k.shape[3],
                                                              stride):
            t = _Bf16Conv.apply(t, k, b)
        else:
            t = tfsem.conv2d(t, k, b, stride, padding)
        bb = "%s/%s/" % (sc, bnname)
        y, mm, mv, _, _ = tfsem.batch_norm(t, p[bb + "gamma"], p[bb + "beta"], p[bb + "moving_mean"],
                                           p[bb + "moving_variance"], training, BN_MOMENTUM, BN_EPS)
        new_stats[bb + "moving_mean"], new_stats[bb + "moving_variance"] = mm, mv
        return relu(name, y)

    def block(name, t):
        for i in (1, 2):
            t = cbr("layer%s/conv_%d" % (name, i), "layer%s/bn_%d" % (name, i), t)
        return t

    net = x
    skips = {}
    for name, F_, pool, pad in cfg["enc"]:
        net = block(name, net)
        skips[name] = net
        acts["conv" + name] = net
        if pool is not None:
            net = cbr("layer%s/pool_2" % name, "layer%s/bn_pool_2" % name, net, 2, pad)
            acts["pool" + name] = net
    N = x.shape[0]
    hn = cfg.get("heads", ("mean", "variance"))
    mean = tfsem.conv2d(net, p["%s/%s/kernel" % (sc, hn[0])], p["%s/%s/bias" % (sc, hn[0])], 1, "VALID").reshape(N, -1)
    var = tfsem.conv2d(net, p["%s/%s/kernel" % (sc, hn[1])], p["%s/%s/bias" % (sc, hn[1])], 1, "VALID").reshape(N, -1)
    if cfg.get("encoder_only"):        # models/multimodal.py:172-176: std = softplus(conv)
        return dict(mean=mean, std=torch.nn.functional.softplus(var), raw_std=var, acts=acts, masks=masks_out), new_stats
    z = mean + var * eps
    hh, hw = cfg["head"]
    net = relu("dense", z @ p[sc + "/dense/kernel"] + p[sc + "/dense/bias"]).reshape(N, hh, hw, 1)
    net = relu("conv2d", tfsem.conv2d(net, p[sc + "/conv2d/kernel"], p[sc + "/conv2d/bias"], 1, "SAME"))
    acts["conv2d"] = net
    for name, F_, k, skip in cfg["dec"]:
        up = tfsem.conv2d_transpose_valid(net, p["%s/upsample_%s/kernel" % (sc, name)],
                                          p["%s/upsample_%s/bias" % (sc, name)], 2)
        net = block(name, torch.cat([up, skips[skip]], dim=-1))
        acts["conv" + name] = net
    out = torch.sigmoid(tfsem.conv2d(net, p[sc + "/final/kernel"], p[sc + "/final/bias"], 1, "SAME"))
    return dict(output=out, mean=mean, variance=var, z=z, acts=acts, masks=masks_out), new_stats


def losses(p, x, fw, model="UNet"):
    """trainer/trainer.py:58-75"""
    cfg = CONFIGS[model]
    mse = tfsem.mse_loss(x, fw["output"])
    hub = tfsem.huber_loss(x, fw["output"])
    mu, sg = fw["mean"], fw["variance"]
    kl = 0.5 * (mu * mu + sg * sg - torch.log(1e-8 + sg * sg) - 1).mean(1)
    latent = kl.mean(0) / 1000000
    reg = sum(tfsem.l2_regularizer(w, cfg["wd"]) for n, w in p.items() if regularized(n))
    return dict(loss=latent + mse + hub + reg, mse=mse, huber=hub, latent=latent, reg=reg)


class Oracle(object):
    """CPU train step: forward (batch statistics), losses, autograd, TF-1 Adam, moving-average update."""

    def __init__(self, model="UNet", learning_rate=1e-4, seed=1240, dtype=torch.float32, params=None,
                 bf16_operands=False):
        self.model = model
        self.bf16_operands = bf16_operands
        self.lr = learning_rate
        self.dtype = dtype
        self.params = OrderedDict((k, v.to(dtype).clone()) for k, v in (params or init_params(model, seed)).items())
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.params.items() if trainable(k))
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.params.items() if trainable(k))
        self.step = 0

    def train_step(self, x, eps, apply=True, relu_masks=None):
        p = OrderedDict((k, v.clone().requires_grad_(trainable(k))) for k, v in self.params.items())
        fw, stats = forward(p, x.to(self.dtype), eps.to(self.dtype), self.model, True, relu_masks, self.bf16_operands)
        ls = losses(p, x.to(self.dtype), fw, self.model)
        names = [k for k in p if trainable(k)]
        grads = torch.autograd.grad(ls["loss"], [p[k] for k in names])
        g = OrderedDict(zip(names, grads))
        if apply:
            self.step += 1
            for k in names:
                self.params[k], self.m[k], self.v[k] = tfsem.adam_tf1(self.params[k], g[k], self.m[k], self.v[k],
                                                                      self.step, self.lr)
            for k, v in stats.items():
                self.params[k] = v.detach()
        return dict(losses={k: float(v.detach()) for k, v in ls.items()}, grads=g, fw=fw, new_stats=stats)


def synthetic_batch(model, n, seed=1234, dtype=torch.float32):
    cfg = CONFIGS[model]
    g = torch.Generator().manual_seed(seed)
    H, W = cfg["input_hw"]
    x = torch.rand(n, H, W, cfg["cin"], generator=g, dtype=torch.float64).to(dtype)
    eps = torch.randn(n, cfg["Z"], generator=g, dtype=torch.float64).to(dtype)
    return x, eps
