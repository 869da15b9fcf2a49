"""Oracle: `DualCamHybridModel` (scope 'DualCamNet') and the clip-level classification loss of
trainer/trainer_reconstructed_class.py.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows models/dualcamnet.py:82-106 and models/base.py:4-66:
  reshape [-1,12,36,48,12] -> conv3d W[12,1,1,12,12] SAME (temporal pad 5 / 6) + b, ReLU -> reshape [-1,36,48,12]
  -> conv 5x5 12->32 SAME + b, ReLU -> max-pool 3x3 s3 VALID -> conv 5x5 32->128 SAME + b, ReLU
  -> reduce_sum over H, W -> FC 128->1000 + b, ReLU -> FC 1000->classes + b.
Loss (trainer_reconstructed_class.py:49-56): logits = mean over the clip's 12 frames;
tf.losses.softmax_cross_entropy (mean over clips); accuracy = mean(argmax == argmax labels).
Init: truncated normal sigma = 0.01, biases 0 (base.py:9-10,23-26,64-65).
Parity unpinned at the TensorFlow boundary.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import tfsem

SCOPE = "DualCamNet"


def param_shapes(num_classes=14):
    return OrderedDict([
        (SCOPE + "/conv1/weights", (12, 1, 1, 12, 12)), (SCOPE + "/conv1/biases", (12,)),
        (SCOPE + "/conv2/weights", (5, 5, 12, 32)), (SCOPE + "/conv2/biases", (32,)),
        (SCOPE + "/conv3/weights", (5, 5, 32, 128)), (SCOPE + "/conv3/biases", (128,)),
        (SCOPE + "/full1/weights", (128, 1000)), (SCOPE + "/full1/biases", (1000,)),
        (SCOPE + "/full3/weights", (1000, num_classes)), (SCOPE + "/full3/biases", (num_classes,)),
    ])


def init_params(num_classes=14, seed=1241, dtype=torch.float32, std=0.01, bias_std=0.0):
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for name, shape in param_shapes(num_classes).items():
        if name.endswith("biases"):
            p[name] = (bias_std * torch.randn(*shape, generator=g, dtype=torch.float64)).to(dtype)
        else:
            t = torch.randn(*shape, generator=g, dtype=torch.float64).clamp(-2, 2)
            p[name] = (t * std).to(dtype)
    return p


def forward(p, x, num_frames=12, relu_masks=None):
    """x [clips*12, 36, 48, 12] -> per-frame logits [clips*12, classes]"""
    masks = OrderedDict()

    def relu(name, t):
        if relu_masks is not None and name in relu_masks:
            return t * relu_masks[name].to(t.dtype).reshape(t.shape)
        y = torch.relu(t)
        masks[name] = y > 0
        return y

    NF, H, W, C = x.shape
    clips = NF // num_frames
    v = x.reshape(clips, num_frames, H, W, C).permute(0, 4, 1, 2, 3)            # N C D H W
    w1 = p[SCOPE + "/conv1/weights"].permute(4, 3, 0, 1, 2)                     # [out, in, D, 1, 1]
    v = F.pad(v, (0, 0, 0, 0, 5, 6))                                             # TF SAME, k = 12: 5 before, 6 after
    v = F.conv3d(v, w1, p[SCOPE + "/conv1/biases"])
    v = relu("conv1", v.permute(0, 2, 3, 4, 1).reshape(NF, H, W, C))
    v = relu("conv2", tfsem.conv2d(v, p[SCOPE + "/conv2/weights"], p[SCOPE + "/conv2/biases"], 1, "SAME"))
    v = F.max_pool2d(v.permute(0, 3, 1, 2), 3, 3).permute(0, 2, 3, 1)
    v = relu("conv3", tfsem.conv2d(v, p[SCOPE + "/conv3/weights"], p[SCOPE + "/conv3/biases"], 1, "SAME"))
    v = v.sum(dim=(1, 2))
    v = relu("full1", v @ p[SCOPE + "/full1/weights"] + p[SCOPE + "/full1/biases"])
    logits = v @ p[SCOPE + "/full3/weights"] + p[SCOPE + "/full3/biases"]
    return logits, masks


def loss_and_accuracy(frame_logits, labels, num_frames=12):
    """labels: int64 class per clip"""
    K = frame_logits.shape[-1]
    logits = frame_logits.reshape(-1, num_frames, K).mean(1)
    loss = F.cross_entropy(logits, labels)
    acc = (logits.argmax(1) == labels).double().mean()
    return loss, acc, logits


def train_step_grads(p, x, labels, num_frames=12, relu_masks=None):
    q = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in p.items())
    fl, masks = forward(q, x, num_frames, relu_masks)
    loss, acc, logits = loss_and_accuracy(fl, labels, num_frames)
    grads = torch.autograd.grad(loss, list(q.values()))
    return dict(loss=float(loss.detach()), accuracy=float(acc), logits=logits.detach(), frame_logits=fl.detach(),
                grads=OrderedDict(zip(q.keys(), grads)), masks=masks)
