"""Oracle: the `TrainerMask` train / eval step.  TEST INFRASTRUCTURE.

Follows trainer/mfcctrainer.py:
  :38-40   mfccmap = tile(mfcc) -> [N,36,48,12]
  :42,45   modelimages._build_model(video); modelac._build_model(mfccmap, modelimages.output)
  :46-62   lossmse, l1 (Huber), latent_loss = FLAGS.latent_loss * mean_b(KL), loss = latent +
           tf.losses.get_total_loss() (= MSE + Huber + sum of slim L2 regularisers, App. B.5)
  :64      var_list = UNetAcRes/* + resnet_v1_50/conv_map/*
  :72-79   Adam(lr) under control_dependencies(UPDATE_OPS)  (BN moving averages advance)
  :343-349 one session.run per step with is_training=1 for both models
  :411-442 evaluation: is_training=0, MSE only
"""
from collections import OrderedDict

import torch

from . import resnet50, tfsem, unet_acresnet


class Oracle(object):
    def __init__(self, num_skip=1, embedding=False, learning_rate=1e-4, latent_loss=1e-6,
                 use_mse=True, use_huber=True, dtype=torch.float32, seed=1238, randomize=False, f16_operands=False):
        self.num_skip = num_skip
        self.embedding = bool(embedding)
        self.lr = learning_rate
        self.latent_w = latent_loss
        self.use_mse, self.use_huber = use_mse, use_huber
        self.dtype = dtype
        self.f16_operands = f16_operands      # trunk convs on fp16-rounded operands (the product's precision="f16")
        self.res = resnet50.init_params(seed, dtype, randomize_bn=randomize)
        self.gen = unet_acresnet.init_params(seed + 1, num_skip, self.embedding, dtype,
                                             bias_std=0.05 if randomize else 0.0)
        self.train_names = list(self.gen.keys()) + resnet50.train_var_names()
        self.m = OrderedDict((k, torch.zeros_like(self._get(k))) for k in self.train_names)
        self.v = OrderedDict((k, torch.zeros_like(self._get(k))) for k in self.train_names)
        self.step = 0

    def _get(self, name):
        return self.gen[name] if name in self.gen else self.res[name]

    def _set(self, name, val):
        if name in self.gen:
            self.gen[name] = val
        else:
            self.res[name] = val

    def state_dict(self):
        d = OrderedDict()
        d.update(self.res)
        d.update(self.gen)
        return d

    def forward(self, video, mfcc, eps, training, end_points=None, relu_masks=None):
        n = mfcc.shape[0]
        mfccmap = mfcc.reshape(n, 1, 1, 12).expand(n, 36, 48, 12).contiguous()
        rm = relu_masks or {}
        feat, updates = resnet50.forward(self.res, video, training, end_points, feat_mask=rm.get("conv_map"),
                                         f16_operands=self.f16_operands)
        mean, std, out = unet_acresnet.forward(self.gen, mfccmap, feat, eps, self.num_skip,
                                               self.embedding, end_points, relu_masks=relu_masks)
        return mean, std, out, updates

    def losses(self, acoustic, mean, std, out):
        mse = tfsem.mse_loss(acoustic, out)
        hub = tfsem.huber_loss(acoustic, out)
        reg = resnet50.regularization_loss(self.res)
        total = reg
        if self.use_mse:
            total = total + mse
        if self.use_huber:
            total = total + hub
        lat = torch.zeros((), dtype=self.dtype)
        if not self.embedding:
            kl = 0.5 * (mean ** 2 + std ** 2 - torch.log(1e-8 + std ** 2) - 1).sum(1)
            lat = self.latent_w * kl.mean(0)
            total = total + lat
        return OrderedDict(mse=mse, huber=hub, latent=lat, reg=reg, loss=total)

    def train_step(self, acoustic, mfcc, video, eps, end_points=None, keep_grads=False, relu_masks=None):
        """one optimisation step; returns dict of python floats (+ tensors when asked).
        relu_masks: {layer: bool tensor} ReLU on/off patterns taken from the implementation under test
        (parity tests only; see unet_acresnet._relu)."""
        leaves = []
        for k in self.train_names:
            t = self._get(k).detach().clone().requires_grad_(True)
            self._set(k, t)
            leaves.append(t)
        mean, std, out, updates = self.forward(video, mfcc, eps, True, end_points, relu_masks)
        L = self.losses(acoustic, mean, std, out)
        grads = torch.autograd.grad(L["loss"], leaves, allow_unused=True)
        self.step += 1
        gd = OrderedDict()
        with torch.no_grad():
            for k, t, g in zip(self.train_names, leaves, grads):
                if g is None:
                    g = torch.zeros_like(t)
                gd[k] = g
                p2, m2, v2 = tfsem.adam_tf1(t.detach(), g, self.m[k], self.v[k], self.step, self.lr)
                self._set(k, p2)
                self.m[k], self.v[k] = m2, v2
            for k, val in updates.items():
                self.res[k] = val.detach()
        res = OrderedDict((k, float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in L.items())
        res["output"] = out.detach()
        res["mean"] = mean.detach()
        res["std"] = None if std is None else std.detach()
        if keep_grads:
            res["grads"] = gd
        return res

    @torch.no_grad()
    def eval_step(self, acoustic, mfcc, video, eps):
        mean, std, out, _ = self.forward(video, mfcc, eps, False)
        res = OrderedDict(mse=float(tfsem.mse_loss(acoustic, out)))
        for i in range(4):  # per-3-channel MSEs of test(), mfcctrainer.py:105-117
            res["mse%d" % i] = float(tfsem.mse_loss(acoustic[..., 3 * i:3 * i + 3], out[..., 3 * i:3 * i + 3]))
        res["output"] = out
        return res


def synthetic_batch(n, seed=1234, dtype=torch.float32):
    """Seeded synthetic inputs at the reference's shapes: video U[0,1) [n,224,298,3];
    mfcc U[0,1) min-max normalised per vector; acoustic U[0,1) min-max normalised per image; eps N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    video = torch.rand(n, 224, 298, 3, generator=g, dtype=torch.float64)
    mfcc = torch.rand(n, 12, generator=g, dtype=torch.float64)
    mfcc = mfcc - mfcc.amin(1, keepdim=True)
    mfcc = mfcc / mfcc.amax(1, keepdim=True)
    ac = torch.rand(n, 36, 48, 12, generator=g, dtype=torch.float64)
    ac = ac - ac.amin((1, 2, 3), keepdim=True)
    ac = ac / ac.amax((1, 2, 3), keepdim=True)
    eps = torch.randn(n, unet_acresnet.Z, generator=g, dtype=torch.float64)
    return ac.to(dtype), mfcc.to(dtype), video.to(dtype), eps.to(dtype)
