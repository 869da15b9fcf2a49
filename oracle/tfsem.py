"""TensorFlow-1.x op semantics restated on PyTorch CPU tensors (NHWC at the interface).
TEST INFRASTRUCTURE — see oracle/__init__.py.  Rules: SURVEY.md App. B.
"""
import math

import torch
import torch.nn.functional as F


def same_pads(size, k, s):
    """TF 'SAME': out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total//2."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d(x, w, bias=None, stride=1, padding="SAME"):
    """tf.layers.conv2d / tf.nn.conv2d.  x [N,H,W,C], w HWIO.  padding 'SAME' | 'VALID' | int p
    (explicit symmetric zero pad then VALID: slim resnet_utils.conv2d_same for stride > 1)."""
    R, S = w.shape[0], w.shape[1]
    xt = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        _, pt, pb = same_pads(x.shape[1], R, stride)
        _, pl, pr = same_pads(x.shape[2], S, stride)
        xt = F.pad(xt, (pl, pr, pt, pb))
    elif padding != "VALID":
        p = int(padding)
        xt = F.pad(xt, (p, p, p, p))
    y = F.conv2d(xt, w.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_same_slim(x, w, stride):
    """slim resnet_utils.conv2d_same: SAME if stride == 1, else explicit pad (k-1)//2 each side +
    VALID (models/resnet50.py:118,207)."""
    k = w.shape[0]
    if stride == 1:
        return conv2d(x, w, None, 1, "SAME")
    total = k - 1
    beg = total // 2
    end = total - beg
    xt = F.pad(x.permute(0, 3, 1, 2), (beg, end, beg, end))
    return F.conv2d(xt, w.permute(3, 2, 0, 1), None, stride=stride).permute(0, 2, 3, 1)


def conv2d_transpose_valid(x, w, bias, stride):
    """tf.layers.conv2d_transpose, padding VALID.  w is [kh, kw, out, in]; TF output size is
    in*stride + max(k - stride, 0) (App. B.2), i.e. PyTorch's size plus output_padding = stride-k
    when k < stride."""
    op = (max(stride - w.shape[0], 0), max(stride - w.shape[1], 0))
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), bias, stride=stride,
                           output_padding=op)
    return y.permute(0, 2, 3, 1)


def max_pool_same(x, k, stride):
    """slim max_pool2d inside resnet_arg_scope: padding SAME (-inf outside)."""
    _, pt, pb = same_pads(x.shape[1], k, stride)
    _, pl, pr = same_pads(x.shape[2], k, stride)
    xt = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(xt, k, stride).permute(0, 2, 3, 1)


def subsample(x, factor):
    """slim resnet_utils.subsample = max_pool2d(1x1, stride=factor) SAME."""
    if factor == 1:
        return x
    return x[:, ::factor, ::factor, :]


def batch_norm(x, gamma, beta, moving_mean, moving_var, training, decay=0.997, eps=1e-5):
    """slim batch_norm (fused): batch statistics with biased variance when training; the moving
    variance is updated with the unbiased batch variance (App. B.4).
    Returns (y, new_moving_mean, new_moving_var, mean, var)."""
    if training:
        dims = tuple(range(x.dim() - 1))
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[-1]
        unb = var * (n / max(n - 1, 1))
        new_mm = decay * moving_mean + (1 - decay) * mean.detach()
        new_mv = decay * moving_var + (1 - decay) * unb.detach()
    else:
        mean, var = moving_mean, moving_var
        new_mm, new_mv = moving_mean, moving_var
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    return y, new_mm, new_mv, mean, var


def minmax_norm(x, dims, sel=None):
    """x - reduce_min; then / reduce_max of the shifted tensor (models/unet_acresnet.py:55-58).
    torch.amin/amax split the gradient equally among ties, as TF's reduce_min/max do (App. B.8).
    sel = (argmin set, argmax set) as bool tensors prescribes WHICH elements are the minimum / maximum (taken from
    the implementation under test, like the ReLU patterns of oracle/unet_acresnet.py:_relu): min and max are then the
    means over those sets, i.e. the same values up to fp32 rounding and the same equal split of the gradient, so
    that both sides differentiate the same piecewise-linear function when an element sits within rounding of the
    minimum (a post-ReLU zero that is 1e-7 on one side)."""
    if sel is not None:
        mnm, mxm = sel[0].to(x.dtype), sel[1].to(x.dtype)
        a = x - (x * mnm).sum(dim=dims, keepdim=True) / mnm.sum(dim=dims, keepdim=True)
        return a / ((a * mxm).sum(dim=dims, keepdim=True) / mxm.sum(dim=dims, keepdim=True))
    a = x - x.amin(dim=dims, keepdim=True)
    return a / a.amax(dim=dims, keepdim=True)


def mse_loss(labels, predictions):
    """tf.losses.mean_squared_error: mean over all elements (App. B.5)."""
    return ((predictions - labels) ** 2).mean()


def huber_loss(labels, predictions, delta=1.0):
    """tf.losses.huber_loss: 0.5 q^2 + delta (|e| - q), q = min(|e|, delta); mean over elements."""
    e = (predictions - labels).abs()
    q = torch.clamp(e, max=delta)
    return (0.5 * q * q + delta * (e - q)).mean()


def l2_regularizer(w, scale):
    """slim l2_regularizer(scale)(w) = scale * sum(w^2) / 2."""
    return scale * 0.5 * (w * w).sum()


def adam_tf1(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update for 1-based `step` (App. B.7); returns new (param, m, v)."""
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    return param - lr_t * m / (torch.sqrt(v) + eps), m, v


# ---- initialisers (distribution parity only, App. B.10) ------------------------------------------
def xavier_uniform(gen, shape, fan_in, fan_out, dtype=torch.float32):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(*shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def variance_scaling_trunc_normal(gen, shape, fan_in, dtype=torch.float32):
    """slim variance_scaling_initializer(): factor 2, FAN_IN, truncated normal sigma=sqrt(1.3*2/fan_in)."""
    std = math.sqrt(1.3 * 2.0 / fan_in)
    t = torch.randn(*shape, generator=gen, dtype=torch.float64)
    for _ in range(8):  # resample outside 2 sigma
        bad = t.abs() > 2
        if not bad.any():
            break
        t = torch.where(bad, torch.randn(*shape, generator=gen, dtype=torch.float64), t)
    t = t.clamp(-2, 2)
    return (t * std).to(dtype)
