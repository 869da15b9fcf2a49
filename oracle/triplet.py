"""Oracle: the cross-modal triplet losses of trainer/trainer_three.py — `_pairwise_distances` :551-591 (always called
with squared=True), `_get_anchor_positive_and_negative_triplet_mask` :593-624, `_get_triplet_mask` :626-642,
`mix_data_hard` :648-683, `mix_all` :685-732.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Restated op for op in torch so that autograd reproduces TensorFlow's gradients:
  * tf.maximum(x, 0.0) passes the gradient where x >= 0 (equality included)      -> `_tf_max0`
  * tf.reduce_max / reduce_min split the gradient evenly among ties               -> torch.amax / amin do the same
  * counts built from tf.greater / tf.to_float carry no gradient
The reference's distance matrix is kept as written: expand_dims(square_norm0, 0) broadcasts |e0_j|^2 along rows and
expand_dims(square_norm1, 1) broadcasts |e1_i|^2 along columns, so D[i][j] = |e0_j|^2 - 2 <e0_i, e1_j> + |e1_i|^2.
Parity unpinned at the TensorFlow boundary.
"""
import torch


def _tf_max0(x):
    return torch.where(x >= 0, x, torch.zeros_like(x))


def pairwise_distances(e0, e1):
    dot0 = e0 @ e0.t()
    dot1 = e1 @ e1.t()
    dotab = e0 @ e1.t()
    n0 = torch.diagonal(dot0)
    n1 = torch.diagonal(dot1)
    d = n0.unsqueeze(0) - 2.0 * dotab + n1.unsqueeze(1)
    return _tf_max0(d)


def _same_video(labels, scenario):
    return (labels.unsqueeze(0) == labels.unsqueeze(1)) & (scenario.unsqueeze(0) == scenario.unsqueeze(1))


def triplet_mask(labels, scenario):
    same = _same_video(labels, scenario)
    return same.unsqueeze(2) & ~same.unsqueeze(1)


def mix_all(e0, e1, labels, scenario, margin):
    d = pairwise_distances(e0, e1)
    t = d.unsqueeze(2) - d.unsqueeze(1) + margin
    mask = triplet_mask(labels, scenario).to(d.dtype)
    t = _tf_max0(mask * t)
    num_pos = (t > 1e-16).to(d.dtype).sum()
    num_valid = mask.sum()
    return t.sum() / (num_pos + 1e-16), num_pos / (num_valid + 1e-16), num_pos, num_valid


def mix_data_hard(e0, e1, labels, scenario, margin):
    d = pairwise_distances(e0, e1)
    same = _same_video(labels, scenario)
    pos = same.to(d.dtype)
    neg = (~same).to(d.dtype)
    hardest_pos = torch.amax(pos * d, dim=1, keepdim=True)
    row_max = torch.amax(d, dim=1, keepdim=True)
    an = d + row_max * (1.0 - neg)
    hardest_neg = torch.amin(an, dim=1, keepdim=True)
    tl = _tf_max0(hardest_pos - hardest_neg + margin)
    mask = triplet_mask(labels, scenario).to(d.dtype)
    num_pos = (tl > 1e-16).to(d.dtype).sum()
    num_valid = mask.sum()
    return tl.mean(), num_pos / (num_valid + 1e-16), num_pos, num_valid
