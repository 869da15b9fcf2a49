"""Oracle: the reference's modified ResNet-v1-50 image encoder.  TEST INFRASTRUCTURE.

Follows models/vision.py:45-71 (arg scope: weight decay 5e-4, BN decay .997 eps 1e-5 scale=True,
`is_training` feeds batch_norm) and models/resnet50.py:
  :75-125   bottleneck (shortcut = subsample | 1x1 conv+BN; conv1 1x1; conv2 3x3 conv2d_same stride;
            conv3 1x1 no ReLU; relu(shortcut + residual))
  :205-209  root conv 7x7/2 conv2d_same + max_pool 3x3/2, blocks, conv_map 3x4 VALID 2048->12 (+BN+ReLU)
  :229-250  stride sits in the LAST unit of a block
  :261-266  block1 stride 1 (!), block2 stride 2, block3 stride 2, block4 stride 1
Every slim conv2d here has normalizer_fn=batch_norm, hence no bias, and ReLU unless activation_fn=None.
"""
from collections import OrderedDict

import torch

from . import tfsem

SCOPE = "resnet_v1_50"
BLOCKS = (("block1", 64, 3, 1), ("block2", 128, 4, 2), ("block3", 256, 6, 2), ("block4", 512, 3, 1))
WEIGHT_DECAY = 5e-4
BN_DECAY = 0.997
BN_EPS = 1e-5
BN_SUFFIXES = ("gamma", "beta", "moving_mean", "moving_variance")


def units():
    """yields (scope, depth_in, depth, depth_bottleneck, stride) in execution order"""
    depth_in = 64
    for name, base, n, stride in BLOCKS:
        for u in range(n):
            s = stride if u == n - 1 else 1
            yield ("%s/%s/unit_%d/bottleneck_v1" % (SCOPE, name, u + 1), depth_in, base * 4, base, s)
            depth_in = base * 4


def conv_layers():
    """yields (scope, kh, kw, cin, cout) for every conv (each followed by BatchNorm)"""
    yield (SCOPE + "/conv1", 7, 7, 3, 64)
    for scope, din, d, db, s in units():
        if din != d:
            yield (scope + "/shortcut", 1, 1, din, d)
        yield (scope + "/conv1", 1, 1, din, db)
        yield (scope + "/conv2", 3, 3, db, db)
        yield (scope + "/conv3", 1, 1, db, d)
    yield (SCOPE + "/conv_map", 3, 4, 2048, 12)


def param_shapes():
    shapes = OrderedDict()
    for scope, kh, kw, cin, cout in conv_layers():
        shapes[scope + "/weights"] = (kh, kw, cin, cout)
        for sfx in BN_SUFFIXES:
            shapes[scope + "/BatchNorm/" + sfx] = (cout,)
    return shapes


def train_var_names():
    """modelimages.train_vars: slim.get_trainable_variables(scope + '/conv_map') (vision.py:70-71)"""
    s = SCOPE + "/conv_map"
    return [s + "/weights", s + "/BatchNorm/gamma", s + "/BatchNorm/beta"]


def init_params(seed=1238, dtype=torch.float32, randomize_bn=False):
    """slim defaults: variance-scaling conv weights, gamma 1, beta 0, moving mean 0 / var 1.
    randomize_bn perturbs gamma/beta/moving stats so parity tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for scope, kh, kw, cin, cout in conv_layers():
        p[scope + "/weights"] = tfsem.variance_scaling_trunc_normal(g, (kh, kw, cin, cout), kh * kw * cin, dtype)
        if randomize_bn:
            p[scope + "/BatchNorm/gamma"] = (1 + 0.2 * torch.randn(cout, generator=g, dtype=torch.float64)).to(dtype)
            p[scope + "/BatchNorm/beta"] = (0.2 * torch.randn(cout, generator=g, dtype=torch.float64)).to(dtype)
            p[scope + "/BatchNorm/moving_mean"] = (0.1 * torch.randn(cout, generator=g, dtype=torch.float64)).to(dtype)
            p[scope + "/BatchNorm/moving_variance"] = (1 + 0.5 * torch.rand(cout, generator=g, dtype=torch.float64)).to(dtype)
        else:
            p[scope + "/BatchNorm/gamma"] = torch.ones(cout, dtype=dtype)
            p[scope + "/BatchNorm/beta"] = torch.zeros(cout, dtype=dtype)
            p[scope + "/BatchNorm/moving_mean"] = torch.zeros(cout, dtype=dtype)
            p[scope + "/BatchNorm/moving_variance"] = torch.ones(cout, dtype=dtype)
    return p


def _f16_operand(t, scale):
    """the value an fp16-storage operand carries: round(t * scale) to fp16, rescaled (the product path's hi plane;
    scale = 2^-2 for activations, 2^10 for weights: exact powers of two, they only move fp16's range)"""
    return (t * scale).to(torch.float16).to(t.dtype) / scale


def _conv_bn(p, scope, x, stride, training, relu, updates, padding=None, mask=None, f16_operands=False):
    w = p[scope + "/weights"]
    if f16_operands:      # BASELINE configs[4] "fp16 storage, fp32 accumulation": both conv operands at 11 bits
        x, w = _f16_operand(x, 0.25), _f16_operand(w, 1024.0)
    if padding == "VALID":
        y = tfsem.conv2d(x, w, None, stride, "VALID")
    elif w.shape[0] == 1:
        y = tfsem.conv2d(x, w, None, stride, "SAME")       # layers.conv2d 1x1, SAME
    else:
        y = tfsem.conv2d_same_slim(x, w, stride)            # resnet_utils.conv2d_same
    b = scope + "/BatchNorm/"
    y, mm, mv, _, _ = tfsem.batch_norm(y, p[b + "gamma"], p[b + "beta"], p[b + "moving_mean"],
                                       p[b + "moving_variance"], training, BN_DECAY, BN_EPS)
    if training:
        updates[b + "moving_mean"] = mm
        updates[b + "moving_variance"] = mv
    if relu and mask is not None:
        return y * mask.to(y.dtype)   # ReLU pattern of the implementation under test (see unet_acresnet._relu)
    return torch.relu(y) if relu else y


def forward(p, images, training, end_points=None, feat_mask=None, f16_operands=False):
    """images [N,224,298,3] -> features [N,12,16,12]; returns (features, moving-stat updates).
    Gradients flow only from conv_map on (the trunk is not in var_list, mfcctrainer.py:64): the trunk
    runs under no_grad.  f16_operands: the 52 bottleneck convs see fp16-rounded activations and weights (the
    product's precision="f16" mode; the stem and conv_map keep fp32 operands there as well)."""
    updates = OrderedDict()
    ep = end_points if end_points is not None else {}
    with torch.no_grad():
        net = _conv_bn(p, SCOPE + "/conv1", images, 2, training, True, updates)
        ep[SCOPE + "/conv1"] = net
        net = tfsem.max_pool_same(net, 3, 2)
        ep[SCOPE + "/pool1"] = net
        for scope, din, d, db, s in units():
            if din == d:
                shortcut = tfsem.subsample(net, s)
            else:
                shortcut = _conv_bn(p, scope + "/shortcut", net, s, training, False, updates, f16_operands=f16_operands)
            r = _conv_bn(p, scope + "/conv1", net, 1, training, True, updates, f16_operands=f16_operands)
            r = _conv_bn(p, scope + "/conv2", r, s, training, True, updates, f16_operands=f16_operands)
            r = _conv_bn(p, scope + "/conv3", r, 1, training, False, updates, f16_operands=f16_operands)
            net = torch.relu(shortcut + r)
            ep[scope] = net
    net = _conv_bn(p, SCOPE + "/conv_map", net, 1, training, True, updates, padding="VALID", mask=feat_mask)
    ep[SCOPE + "/conv_map"] = net
    return net, updates


def regularization_loss(p):
    """sum of slim l2_regularizer(5e-4) over every conv kernel of the scope (App. B.5)"""
    tot = 0.0
    for scope, *_ in conv_layers():
        tot = tot + tfsem.l2_regularizer(p[scope + "/weights"], WEIGHT_DECAY)
    return tot
