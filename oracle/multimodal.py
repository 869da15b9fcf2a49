"""Oracle: the latent associators of models/multimodal.py (`AssociatorVideoAc` :5-73, `AssociatorAudioAc` :75-137)
and the single-associator step of trainer/trainer_proietta.py:104-146 that drives the `unet_z` decoder.
TEST INFRASTRUCTURE — see oracle/__init__.py.

An associator maps the (mean, std) of one modality's VAE to the (mean, std) of the acoustic-image latent with two
towers of tf.layers.dense (ReLU), the last layer linear; std = softplus(.).  Unnamed tf.layers.dense layers get
TF's default names dense, dense_1, ... in creation order: the mean tower first.
Step (trainer_proietta.py:117-146, l2 = 0 terms aside): the decoder `UNetAc` of models/unet_z.py reconstructs the
acoustic image from z = mean + std * eps; loss = MSE + Huber + mean_b(0.5 * sum_j(mu^2 + s^2 - log(1e-8 + s^2) - 1))
/ 1e6; Adam on the associator's variables (the fusion branch's var_list, :101; the single-encoder branch lists two
more models, one of which it never builds).  Parity unpinned at the TensorFlow boundary.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import tfsem, unet_acoustic

TOWERS = {"AssociatorVideoAc": (1024, [512, 512, 256, 256, 150, 150]),
          "AssociatorAudioAc": (256, [256, 256, 150])}


def param_shapes(scope):
    din, widths = TOWERS[scope]
    s = OrderedDict()
    idx = 0
    for tower in range(2):
        cin = din
        for wdt in widths:
            name = "dense" if idx == 0 else "dense_%d" % idx
            s["%s/%s/kernel" % (scope, name)] = (cin, wdt)
            s["%s/%s/bias" % (scope, name)] = (wdt,)
            cin = wdt
            idx += 1
    return s


def init_params(scope, seed=1243, dtype=torch.float32, bias_std=0.0):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, shape in param_shapes(scope).items():
        if name.endswith("/bias"):
            p[name] = (bias_std * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        else:
            p[name] = tfsem.xavier_uniform(g, shape, shape[0], shape[1], dtype)
    return p


def forward(p, scope, mean, std, relu_masks=None):
    """mean, std [N, din] -> (mean' [N,150], std' [N,150] = softplus(raw), raw, relu masks)"""
    _, widths = TOWERS[scope]
    n = len(widths)
    masks = OrderedDict()
    outs = []
    idx = 0
    for t in (mean, std):
        for li in range(n):
            name = "dense" if idx == 0 else "dense_%d" % idx
            t = t @ p["%s/%s/kernel" % (scope, name)] + p["%s/%s/bias" % (scope, name)]
            if li < n - 1:
                if relu_masks is not None and name in relu_masks:
                    t = t * relu_masks[name].to(t.dtype)
                else:
                    t = torch.relu(t)
                    masks[name] = t > 0
            idx += 1
        outs.append(t)
    return outs[0], F.softplus(outs[1]), outs[1], masks


def step_loss_audio(pa, pdec, spec, x, eps, relu_masks_a=None, relu_masks_d=None):
    """the same step with the CONV associator `AssociatorAudio` (models/multimodal.py:139-285: a spectrogram encoder
    in training mode, batch-norm batch statistics) in front of the decoder; tf.losses.get_total_loss() also collects
    the l2_regularizer(8e-5) terms of its conv_conv_pool kernels"""
    from . import unet_vae
    fa, new_stats = unet_vae.forward(pa, spec, None, model="AssociatorAudio", training=True, relu_masks=relu_masks_a)
    m, s = fa["mean"], fa["std"]
    fw = unet_acoustic.forward(pdec, x, eps, m, s, relu_masks=relu_masks_d)
    kl = 0.5 * (m * m + s * s - torch.log(1e-8 + s * s) - 1).sum(1)
    latent = kl.mean(0) / 1000000
    mse, hub = tfsem.mse_loss(x, fw["output"]), tfsem.huber_loss(x, fw["output"])
    reg = sum(tfsem.l2_regularizer(w, 8e-5) for n, w in pa.items() if unet_vae.regularized(n))
    return dict(loss=latent + mse + hub + reg, mse=mse, huber=hub, latent=latent, reg=reg, mean=m, std=s,
                output=fw["output"], masks_a=fa["masks"], masks_d=fw["masks"], new_stats=new_stats)


def step_loss(pa, scope, pdec, x, eps, mean_in, std_in, relu_masks_a=None, relu_masks_d=None):
    m, s, _, masks_a = forward(pa, scope, mean_in, std_in, relu_masks_a)
    fw = unet_acoustic.forward(pdec, x, eps, m, s, relu_masks=relu_masks_d)
    kl = 0.5 * (m * m + s * s - torch.log(1e-8 + s * s) - 1).sum(1)
    latent = kl.mean(0) / 1000000
    mse, hub = tfsem.mse_loss(x, fw["output"]), tfsem.huber_loss(x, fw["output"])
    return dict(loss=latent + mse + hub, mse=mse, huber=hub, latent=latent, mean=m, std=s, output=fw["output"],
                masks_a=masks_a, masks_d=fw["masks"])


# ---- joint-latent fusion MLPs (models/multimodal.py:287-465) ------------------------------------------------
JOINT_HEADS = {"Jointmvae": (("outputac", 133), ("outputvideo", 512), ("outputaudio", 128)),
               "JointTwomvae": (("outputac", 133),),
               "JointTwomvae2": (("outputac", 133), ("outputvideo", 512), ("outputaudio", 128))}


def joint_param_shapes(scope, cin):
    s = OrderedDict()
    widths = [(cin, 512), (512, 512), (512, 512)] + [(512, w) for _, w in JOINT_HEADS[scope]]
    for idx, (a, b) in enumerate(widths):
        name = "dense" if idx == 0 else "dense_%d" % idx
        s["%s/%s/kernel" % (scope, name)] = (a, b)
        s["%s/%s/bias" % (scope, name)] = (b,)
    return s


def joint_init_params(scope, cin, seed=1249, dtype=torch.float32, bias_std=0.0):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, shape in joint_param_shapes(scope, cin).items():
        if name.endswith("/bias"):
            p[name] = (bias_std * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        else:
            p[name] = tfsem.xavier_uniform(g, shape, shape[0], shape[1], dtype)
    return p


def joint_forward(p, scope, inputs, relu_masks=None):
    """inputs: feature maps [..., C_i]; tf.concat on the last axis, tf.layers.dense on the last axis (ReLU
    everywhere, heads included).  -> ({head: tensor}, relu masks by layer name)"""
    net = torch.cat(list(inputs), dim=-1)
    masks = OrderedDict()

    def dense(t, idx):
        name = "dense" if idx == 0 else "dense_%d" % idx
        t = t @ p["%s/%s/kernel" % (scope, name)] + p["%s/%s/bias" % (scope, name)]
        if relu_masks is not None and name in relu_masks:
            return t * relu_masks[name].to(t.dtype)
        t = torch.relu(t)
        masks[name] = t > 0
        return t

    for idx in range(3):
        net = dense(net, idx)
    outs = OrderedDict()
    for k, (attr, _) in enumerate(JOINT_HEADS[scope]):
        outs[attr] = dense(net, 3 + k)
    return outs, masks
