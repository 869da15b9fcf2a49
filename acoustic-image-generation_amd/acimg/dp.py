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


def is_writer(group=None):
    """True on the one rank that writes checkpoints / model.txt / test_accuracy files (rank 0), and on a
    single-process run.  The reference is single-process (trainer/mfcctrainer.py:400-406): under data parallelism
    every rank holds the same weights after each step, so exactly one of them saves."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group) == 0
    return True


def average_moving_statistics(flat_state, moving_ranges, group=None):
    """Batch-norm moving statistics are per replica during training (each rank normalises with its local batch and
    advances its own moving mean / variance: SURVEY §5 'BN x DP').  At checkpoint / evaluation time every rank calls
    this (a collective): the `moving_ranges` [(offset, numel)] of the flat state buffer are replaced by their mean
    over ranks, so that all ranks — and the checkpoint rank 0 writes — hold ONE set of statistics.  gamma / beta in
    the same buffer are frozen and identical on every rank and are left untouched."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat_state
    world = dist.get_world_size(group)
    tmp = flat_state.clone()
    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
    for off, n in moving_ranges:
        flat_state[off:off + n].copy_(tmp[off:off + n] / world)
    return flat_state


def allreduce_sums(values, device=None, group=None):
    """sum over ranks of a few host numbers (validation loss sum, data-set size): every rank gets the same totals, so a
    decision taken on them (best epoch -> a checkpoint, which contains a collective) is the same on every rank.  The
    reduction is done in float64; single process: returned as given."""
    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return vals
    cuda = dist.get_backend(group) == "nccl"
    t = torch.tensor(vals, dtype=torch.float64, device=device if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()


def shards_per_rank(global_batch, shard, world):
    """Strong scaling: a FIXED global batch is cut into shards of `shard` images (the batch-norm group size: every
    shard is normalised with its own batch statistics, on whichever rank it runs); each rank takes
    global_batch / shard / world of them per step, accumulates their gradients, and the ranks exchange once.  The
    arithmetic of a step is therefore the same for every world size that divides the shard count."""
    if global_batch % shard:
        raise ValueError("global batch %d is not a multiple of the shard size %d" % (global_batch, shard))
    n = global_batch // shard
    if n % world:
        raise ValueError("%d shards of %d images do not divide over %d ranks" % (n, shard, world))
    return n // world


class GradComm(object):
    """Bucketed, stream-overlapped all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad, buckets, group=None, force=False):
        """force: keep the exchange on at world size 1 (rehearses the RCCL path on a one-GPU box)"""
        self.flat = flat_grad
        self.buckets = [(int(a), int(b)) for a, b, *_ in buckets]
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grad.device) if (self.cuda and self.enabled) else None
        self.events = [torch.cuda.Event() for _ in self.buckets] if self.stream is not None else []
        self._works = []
        # measurement only (bench.py): `timing` = list of (event, event) pairs bracketing every collective on the stream
        # it is issued from; `muted` = skip the collectives (the no-exchange leg that tells whether the exchange is
        # hidden - the ranks' weights diverge, so never outside a benchmark)
        self.timing = None
        self.muted = False

    def start_timing(self):
        self.timing = []

    def exchange_ms(self):
        """total device time between the bracketing events of the collectives timed so far (call after a synchronize)"""
        return sum(e0.elapsed_time(e1) for e0, e1 in (self.timing or []))

    def bucket_ready(self, i):
        """call (from the host, in launch order) right after the last kernel writing bucket i"""
        if not self.enabled or self.muted:
            return
        a, b = self.buckets[i]
        view = self.flat[a:b]
        if self.stream is None:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return
        self.events[i].record(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.events[i])
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append(w)
            if self.timing is not None:
                w.wait()             # orders the exchange stream behind the collective (no host block)
                e1.record(self.stream)
                self.timing.append((e0, e1))

    def hook(self, i):
        return lambda: self.bucket_ready(i)

    def allreduce_all(self):
        """the whole flat gradient in one exchange (gradient-accumulation steps: the buckets of the individual
        micro-batches are not final, so nothing can be overlapped before the last one has been added)"""
        if self.muted:
            return
        if self.world > 1 or self.enabled:
            if dist.is_available() and dist.is_initialized():
                if self.timing is not None and self.cuda:
                    cur = torch.cuda.current_stream(self.flat.device)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing is not None and self.cuda:
                    e1.record(cur)
                    self.timing.append((e0, e1))

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
