"""Synthetic stand-in for the reference's loaders (dataloader/outdoor_data_mfcc.py): same surface —
`.data` (iterable of 6-tuples), `.num_samples`, `.total_batches` (:117,214,973-976) — with seeded
synthetic tensors at the reference's shapes and value ranges (every tensor entering the
hot path is float32 in [0,1]; acoustic image and MFCC vector min-max normalised per sample)."""
import torch


class SyntheticDataLoader(object):
    def __init__(self, num_samples, batch_size, num_actions=10, num_locations=61, seed=1234, device="cpu"):
        self.num_samples = int(num_samples)
        self.batch_size = int(batch_size)
        self.total_batches = -(-self.num_samples // self.batch_size)
        self.num_actions, self.num_locations = num_actions, num_locations
        self.seed = seed
        self.device = device
        self.data = self

    def _batch(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        video = torch.rand(n, 224, 298, 3, generator=g)
        mfcc = torch.rand(n, 12, generator=g)
        mfcc = mfcc - mfcc.amin(1, keepdim=True)
        mfcc = mfcc / mfcc.amax(1, keepdim=True)
        ac = torch.rand(n, 36, 48, 12, generator=g)
        ac = ac - ac.amin((1, 2, 3), keepdim=True)
        ac = ac / ac.amax((1, 2, 3), keepdim=True)
        labels = torch.nn.functional.one_hot(torch.randint(0, self.num_actions, (n,), generator=g), self.num_actions)
        scen = torch.nn.functional.one_hot(torch.randint(0, self.num_locations, (n,), generator=g), self.num_locations)
        return ac, mfcc, video, labels.float(), scen.float(), mfcc.clone()

    def __iter__(self):
        left = self.num_samples
        i = 0
        while left > 0:
            n = min(self.batch_size, left)
            yield self._batch(n, self.seed + i)
            left -= n
            i += 1
