"""Oracle: the acoustic-image generator `UNetAc` (scope 'UNetAcRes').  TEST INFRASTRUCTURE.

Follows models/unet_acresnet.py:43-101 (+ conv_conv_pool :136-184, upconv_2D :200-217) and the
skip variants models/unet_acresnet0skip.py:85,185-196 / models/unet_acresnet2skip.py:82-83.
tf.layers.conv2d defaults: bias, padding as given, no BN (commented out at :166-167,180-181).
Unnamed tf.layers.dense / conv2d get TF's default names 'dense' / 'conv2d' (SURVEY App. C).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import tfsem

SCOPE = "UNetAcRes"
Z = 150


def param_shapes(num_skip=1, embedding=False):
    s = OrderedDict()

    def conv(name, kh, kw, cin, cout):
        s["%s/%s/kernel" % (SCOPE, name)] = (kh, kw, cin, cout)
        s["%s/%s/bias" % (SCOPE, name)] = (cout,)

    conv("layer1/conv_1", 3, 3, 12, 128)
    conv("layer1/conv_2", 3, 3, 128, 128)
    conv("layer1/pool_2", 3, 3, 128, 128)
    conv("layer2/conv_1", 3, 3, 128, 133)
    conv("layer2/conv_2", 3, 3, 133, 133)
    conv("mean", 12, 16, 145, Z)
    if not embedding:
        conv("std", 12, 16, 145, Z)
    s[SCOPE + "/dense/kernel"] = (Z, 12 * 16 * 12)
    s[SCOPE + "/dense/bias"] = (12 * 16 * 12,)
    conv("conv2d", 3, 3, 12, 133)
    conv("layer4/conv_1", 3, 3, 266 if num_skip == 2 else 133, 128)
    conv("layer4/conv_2", 3, 3, 128, 128)
    conv("layer5/conv_1", 3, 3, 128, 128)
    conv("layer5/conv_2", 3, 3, 128, 128)
    s[SCOPE + "/upsample_1/kernel"] = (2, 2, 128, 128)  # [kh, kw, out, in]
    s[SCOPE + "/upsample_1/bias"] = (128,)
    conv("layer6/conv_1", 3, 3, 128 if num_skip == 0 else 256, 128)
    conv("layer6/conv_2", 3, 3, 128, 128)
    conv("layer7/conv_1", 3, 3, 128, 64)
    conv("layer7/conv_2", 3, 3, 64, 64)
    conv("final", 3, 3, 64, 12)
    return s


def init_params(seed=1239, num_skip=1, embedding=False, dtype=torch.float32, bias_std=0.0):
    """xavier_uniform kernels (explicit at :165; Glorot-uniform is also tf.layers' default), zero biases
    (bias_std > 0 randomises them so parity tests exercise the bias paths)."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, shape in param_shapes(num_skip, embedding).items():
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


_MASKS = None  # see forward(relu_masks=...)


def _relu(name, y):
    """ReLU; with a mask override the on/off pattern comes from the implementation under test, so that
    both sides differentiate the SAME piecewise-linear function (a pre-activation within fp32 rounding
    of zero otherwise picks a different, equally valid, subgradient).  Values are unaffected beyond
    that rounding."""
    if _MASKS is not None and name in _MASKS:
        return y * _MASKS[name].to(y.dtype)
    return torch.relu(y)


def _c(p, name, x, stride=1, padding="SAME", act="relu"):
    y = tfsem.conv2d(x, p["%s/%s/kernel" % (SCOPE, name)], p["%s/%s/bias" % (SCOPE, name)], stride, padding)
    if act == "relu":
        return _relu(name, y)
    return act(y) if act is not None else y


def forward(p, inputs, resnetfeature, eps, num_skip=1, embedding=False, end_points=None, relu_masks=None):
    """inputs: tiled MFCC map [N,36,48,12]; resnetfeature [N,12,16,12]; eps [N,150] stands in for
    tf.random_normal (:77).  Returns (mean, std, output) or (z, None, output) when embedding.
    relu_masks: optional {layer name: bool tensor} ReLU on/off patterns (see _relu)."""
    global _MASKS
    _MASKS = relu_masks
    try:
        return _forward(p, inputs, resnetfeature, eps, num_skip, embedding, end_points)
    finally:
        _MASKS = None


def _forward(p, inputs, resnetfeature, eps, num_skip, embedding, end_points):
    ep = end_points if end_points is not None else {}
    c = _c(p, "layer1/conv_1", inputs)
    ep["layer1/conv_1"] = c
    conv1 = _c(p, "layer1/conv_2", c)
    pool1 = _c(p, "layer1/pool_2", conv1, stride=3)  # 3x3 s3 'same' on 36x48: no padding
    c = _c(p, "layer2/conv_1", pool1)
    ep["layer2/conv_1"] = c
    conv2_0 = _c(p, "layer2/conv_2", c)
    ep.update(conv1=conv1, pool1=pool1, conv2_0=conv2_0)
    sel = _MASKS or {}     # optional prescribed argmin / argmax sets, see tfsem.minmax_norm
    conv2 = tfsem.minmax_norm(conv2_0, (1, 2, 3), sel.get("minmax/conv2_0"))
    feat = tfsem.minmax_norm(resnetfeature, (1, 2, 3), sel.get("minmax/feature"))
    conv2 = torch.cat((conv2, feat), dim=-1)
    ep["features"] = conv2
    n = inputs.shape[0]
    if embedding:
        z = _c(p, "mean", conv2, padding="VALID", act=None).reshape(n, Z)
        z = tfsem.minmax_norm(z, (1,))
        mean, std = z, None
    else:
        mean = _c(p, "mean", conv2, padding="VALID", act=None).reshape(n, Z)
        std = F.softplus(_c(p, "std", conv2, padding="VALID", act=None)).reshape(n, Z)
        z = mean + std * eps
    ep["z"] = z
    net = _relu("dense", (z @ p[SCOPE + "/dense/kernel"] + p[SCOPE + "/dense/bias"]).reshape(n, 12, 16, 12))
    ep["dense"] = net
    net = _c(p, "conv2d", net)
    ep["conv2d"] = net
    if num_skip == 2:
        net = torch.cat((net, conv2_0), dim=-1)
    c = _c(p, "layer4/conv_1", net)
    ep["layer4/conv_1"] = c
    conv4 = _c(p, "layer4/conv_2", c)
    c = _c(p, "layer5/conv_1", conv4)
    ep["layer5/conv_1"] = c
    conv5 = _c(p, "layer5/conv_2", c)
    up = tfsem.conv2d_transpose_valid(conv5, p[SCOPE + "/upsample_1/kernel"], p[SCOPE + "/upsample_1/bias"], 3)
    ep.update(conv4=conv4, conv5=conv5, upsample_1=up)
    if num_skip >= 1:
        up = torch.cat((up, conv1), dim=-1)
    c = _c(p, "layer6/conv_1", up)
    ep["layer6/conv_1"] = c
    conv6 = _c(p, "layer6/conv_2", c)
    c = _c(p, "layer7/conv_1", conv6)
    ep["layer7/conv_1"] = c
    conv7 = _c(p, "layer7/conv_2", c)
    out = _c(p, "final", conv7, act=torch.sigmoid)
    ep.update(conv6=conv6, conv7=conv7, output=out)
    return mean, std, out
