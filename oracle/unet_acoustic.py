"""Oracle: the acoustic-image VAE `UNetAc` of models/unet_noconc.py:46-89 (z from its own heads) and
models/unet_z.py:46-82 (same network, z from EXTERNAL (mean2, std2): the decoder the associator trainers drive),
scope 'UNetAcoustic'.  TEST INFRASTRUCTURE — see oracle/__init__.py.

conv_conv_pool has no batch norm here (commented out, unet_z.py:143-144,156-157); std = softplus(head);
loss recipe for the stand-alone VAE: trainer/trainer.py:58-75 with encoder_type 'Ac' (MSE + Huber + mean_b(0.5 *
mean_j(mu^2 + s^2 - log(1e-8 + s^2) - 1)) / 1e6; kernel_regularizer=None).  Parity unpinned at the TF boundary.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import tfsem

SCOPE = "UNetAcoustic"
Z = 150


def param_shapes():
    s = OrderedDict()

    def conv(name, kh, kw, cin, cout):
        s["%s/%s/kernel" % (SCOPE, name)] = (kh, kw, cin, cout)
        s["%s/%s/bias" % (SCOPE, name)] = (cout,)

    conv("layer1/conv_1", 3, 3, 12, 128)
    conv("layer1/conv_2", 3, 3, 128, 128)
    conv("layer1/pool_2", 3, 3, 128, 128)
    conv("layer3/conv_1", 3, 3, 128, 133)
    conv("layer3/conv_2", 3, 3, 133, 133)
    conv("mean", 12, 16, 133, Z)
    conv("std", 12, 16, 133, Z)
    s[SCOPE + "/dense/kernel"] = (Z, 12 * 16 * 12)
    s[SCOPE + "/dense/bias"] = (12 * 16 * 12,)
    conv("conv2d", 3, 3, 12, 133)
    s[SCOPE + "/upsample_1/kernel"] = (2, 2, 128, 133)
    s[SCOPE + "/upsample_1/bias"] = (128,)
    conv("layer4/conv_1", 3, 3, 128, 128)
    conv("layer4/conv_2", 3, 3, 128, 128)
    conv("layer5/conv_1", 3, 3, 128, 128)
    conv("layer5/conv_2", 3, 3, 128, 128)
    conv("final", 3, 3, 128, 12)
    return s


def init_params(seed=1242, dtype=torch.float32, bias_std=0.0):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, shape in param_shapes().items():
        if name.endswith("/bias"):
            p[name] = (bias_std * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        elif name.endswith("dense/kernel"):
            p[name] = tfsem.xavier_uniform(g, shape, shape[0], shape[1], dtype)
        elif "upsample" in name:
            kh, kw, cout, cin = shape
            p[name] = tfsem.xavier_uniform(g, shape, kh * kw * cin, kh * kw * cout, dtype)
        else:
            kh, kw, cin, cout = shape
            p[name] = tfsem.xavier_uniform(g, shape, kh * kw * cin, kh * kw * cout, dtype)
    return p


def forward(p, x, eps, mean2=None, std2=None, relu_masks=None):
    """x [N,36,48,12]; eps [N,150]; mean2/std2 given = unet_z (external latent), else unet_noconc"""
    masks = OrderedDict()

    def relu(name, t):
        if relu_masks is not None and name in relu_masks:
            return t * relu_masks[name].to(t.dtype).reshape(t.shape)
        y = torch.relu(t)
        masks[name] = y > 0
        return y

    def c(name, t, stride=1):
        return relu(name, tfsem.conv2d(t, p["%s/%s/kernel" % (SCOPE, name)], p["%s/%s/bias" % (SCOPE, name)], stride, "SAME"))

    N = x.shape[0]
    net = c("layer1/conv_2", c("layer1/conv_1", x))
    net = c("layer1/pool_2", net, 3)
    conv2 = c("layer3/conv_2", c("layer3/conv_1", net))
    mean = tfsem.conv2d(conv2, p[SCOPE + "/mean/kernel"], p[SCOPE + "/mean/bias"], 1, "VALID").reshape(N, Z)
    std = F.softplus(tfsem.conv2d(conv2, p[SCOPE + "/std/kernel"], p[SCOPE + "/std/bias"], 1, "VALID").reshape(N, Z))
    z = (mean2 + std2 * eps) if mean2 is not None else (mean + std * eps)
    net = relu("dense", z @ p[SCOPE + "/dense/kernel"] + p[SCOPE + "/dense/bias"]).reshape(N, 12, 16, 12)
    net = c("conv2d", net)
    net = tfsem.conv2d_transpose_valid(net, p[SCOPE + "/upsample_1/kernel"], p[SCOPE + "/upsample_1/bias"], 3)
    net = c("layer4/conv_2", c("layer4/conv_1", net))
    net = c("layer5/conv_2", c("layer5/conv_1", net))
    out = torch.sigmoid(tfsem.conv2d(net, p[SCOPE + "/final/kernel"], p[SCOPE + "/final/bias"], 1, "SAME"))
    return dict(output=out, mean=mean, std=std, z=z, masks=masks)


def vae_losses(x, fw):
    """trainer/trainer.py:58-73 (encoder_type 'Ac')"""
    mse = tfsem.mse_loss(x, fw["output"])
    hub = tfsem.huber_loss(x, fw["output"])
    mu, sg = fw["mean"], fw["std"]
    latent = (0.5 * (mu * mu + sg * sg - torch.log(1e-8 + sg * sg) - 1).mean(1)).mean(0) / 1000000
    return dict(loss=latent + mse + hub, mse=mse, huber=hub, latent=latent)
