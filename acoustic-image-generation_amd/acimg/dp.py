"""Data-parallel gradient exchange: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the GPU node, "gloo" in the CPU tests).

The reference is single-device (no distributed code at all).  The train step shards by
batch: identical weights, local batch per rank, ONE exchange per step — a sum all-reduce of the flat
fp32 gradient buffer (10.85 M elements = 43.4 MB), divided by world size inside the Adam kernel.

MI355X shaping: xGMI is point-to-point (7 links x ~153 GB/s), ring collectives are per-link bound, so
the payload goes out as a FEW LARGE contiguous buckets (the parameters are laid out in the flat
buffer in the order their gradients become final) on a side stream, each fired by an event as soon
as the backward pass has produced its last gradient, overlapping the remaining backward kernels.
Batch-norm statistics of the frozen trunk stay per replica (no SyncBN: 53 latency-bound tiny
all-reduces per step would serialise the trunk).
"""
import torch
import torch.distributed as dist


def make_buckets(ranges, boundaries):
    """ranges: [(name, offset, numel)] in flat order; boundaries: names after which a bucket closes.
    Returns [(start, end, last_name)] contiguous, covering [0, end of last range)."""
    buckets, start = [], 0
    bset = set(boundaries)
    end = 0
    last = None
    for name, off, n in ranges:
        end = off + n
        last = name
        if name in bset:
            buckets.append((start, end, name))
            start = end
    if end > start:
        buckets.append((start, end, last))
    return buckets


class GradComm(object):
    """Bucketed, stream-overlapped all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad, buckets, group=None):
        self.flat = flat_grad
        self.buckets = [(int(a), int(b)) for a, b, *_ in buckets]
        self.group = group
        # ACIMG_DP_FORCE=1 keeps the exchange on at world size 1 (rehearses the RCCL path on a one-GPU box)
        import os
        force = os.environ.get("ACIMG_DP_FORCE") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if (self.cuda and self.enabled) else None
        self.events = [torch.cuda.Event() for _ in self.buckets] if self.stream is not None else []
        self._works = []

    def bucket_ready(self, i):
        """call (from the host, in launch order) right after the last kernel writing bucket i"""
        if not self.enabled:
            return
        a, b = self.buckets[i]
        view = self.flat[a:b]
        if self.stream is None:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return
        self.events[i].record(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.events[i])
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def hook(self, i):
        return lambda: self.bucket_ready(i)

    def wait(self):
        """make the compute stream wait for every outstanding bucket (before the optimiser)"""
        if not self.enabled or self.stream is None:
            return
        for w in self._works:
            w.wait()   # orders the CURRENT stream after the collective; no host block on NCCL/RCCL
        self._works = []
        torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def allreduce_scalars(self, t):
        """mean over ranks of a small tensor (loss logging only)"""
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t /= self.world
        return t
