"""`Trainer`: the acoustic-image train / eval loop, MI355X-native (`TrainerMask` of the reference).

Mirrors trainer/mfcctrainer.py: ctor (:16-26), `_build_functions` (:28-82: tile the MFCC vector,
build image encoder + generator, MSE + Huber + latent_loss*mean(KL) + slim regularisers, Adam on
UNetAcRes/* + resnet_v1_50/conv_map/* with the BN moving-average updates), `train` (:249-398 epoch
loop, validation, best / every-10-epochs checkpoints, model.txt), `_evaluate` (:411-442),
`_retrieve_batch` (:444-467), `test` (:476-536 with the four per-3-channel MSEs), `_save_checkpoint`,
`_restore_model`, `_init_model`.  The body of the hot loop (:343-349, one session.run of
[lossmse, loss, summary_op, train_op]) is `train_step`.

One step = ONE recorded plan of C-ABI kernel launches on fixed device buffers:
  zero loss sums -> tile -> trunk forward (BN batch statistics, moving averages advance) ->
  generator forward -> reconstruction loss (+ its gradient) -> L2 regulariser -> loss scalars ->
  generator backward -> conv_map backward -> [bucketed RCCL all-reduce, overlapped] -> Adam (1 launch).
The eight TensorBoard image summaries the reference evaluates every step (:278-297,343-344) are not
part of the step; scalars are returned as a dict.
"""
import os
from collections import OrderedDict
from datetime import datetime

import torch

from . import _lib, dp, ops
from .flags import FLAGS
from .session import Session
from .unet_acresnet import Z

_FRAMES_PER_SECOND = 12


class _Graph(object):
    """Buffers + plans for one batch size (the reference's graph has a dynamic batch dimension)."""
    pass


class Trainer(object):

    def __init__(self, modelac, modelimages, display_freq=1,
                 learning_rate=0.0001, num_classes=14, num_epochs=1, nr_frames=12, temporal_pooling=False,
                 session=None):
        self.modelac = modelac
        self.modelimages = modelimages
        self.display_freq = display_freq
        self.learning_rate = learning_rate
        self.num_classes = num_classes
        self.num_epochs = num_epochs
        self.nr_frames = nr_frames
        self.temporal_pooling = temporal_pooling
        self.session = session
        self.global_step = 0
        self.graphs = {}
        self.comm = None
        self.noise_seed = 1237
        self.log = print
        self.hold_exchange = False     # True while micro-batch gradients are being accumulated (train_step_sharded)
        self._acc = None
        self._pipe = None              # state of the two-lane pipeline (train_step_pipelined)

    # ------------------------------------------------------------------------------------------------
    # graph construction
    # ------------------------------------------------------------------------------------------------
    def _build_functions(self, data=None, batch_size=None):
        """Builds the step for `batch_size` (default: data.batch_size or FLAGS.batch_size)."""
        if batch_size is None:
            batch_size = getattr(data, "batch_size", None) or FLAGS.batch_size
        if self.session is None:
            self.session = Session()
        g = self._build_graph(int(batch_size), self.modelimages, self.modelac)
        self.session.finalize()
        self.acoustic, self.mfcc, self.video = g.acoustic, g.mfcc, g.video
        self.mfccmap = g.mfccmap
        self.primary = g
        return g

    def _build_graph(self, N, modelimages, modelac):
        sess = self.session
        z = sess.zeros
        g = _Graph()
        g.N = N
        g.modelimages, g.modelac = modelimages, modelac
        g.acoustic = z(N, 36, 48, 12)
        g.mfcc = z(N, 12)
        g.video = z(N, 224, 298, 3)
        g.eps = z(N, Z)
        # self.mfccmap = tile(mfcc) -> [-1,36,48,12]   (trainer/mfcctrainer.py:38-40)
        g.mfccmap = z(N, 36, 48, 12)
        modelimages._build_model(g.video, session=sess)
        modelac._build_model(g.mfccmap, modelimages.output, session=sess, eps=g.eps)
        g.sums = z(4)          # [sum e^2, sum huber, sum w^2, -]
        g.chan = z(16)         # per-channel squared error (test)
        g.losses = z(8)        # mse, huber, latent, reg, total
        g.g_logit = z(N, 36, 48, 12)
        count = N * 36 * 48 * 12
        w_mse, w_hub = float(bool(FLAGS.MSE)), float(bool(FLAGS.huber_loss))
        ae = modelac.embedding
        kl = None if ae else modelac.kl

        def head(plan):
            ops.zero(plan, g.sums)
            ops.tile_mfcc(plan, g.mfcc, g.mfccmap, N, 36 * 48, 12)

        # ---- training step ----
        p = sess.new_plan()
        head(p)
        head_calls = len(p.calls)
        p.extend(modelimages.plan_train)
        p.extend(modelac.plan_fwd)
        ops.recon_loss(p, modelac.output, g.acoustic, g.g_logit, g.sums, count, w_mse, w_hub)
        modelimages.record_regularizer(p, ops.Ptr(g.sums, 2))
        from .vision import WEIGHT_DECAY
        ops.loss_finalize(p, g.sums, kl, N, count, FLAGS.latent_loss, 0.5 * WEIGHT_DECAY, w_mse, w_hub, g.losses)
        # gradient buckets: the flat buffer is laid out in backward-completion order; a host hook after
        # the last kernel of each bucket fires its all-reduce on the side stream (no-op on one GPU)
        s = modelac.scope
        cm = modelimages.scope + "/conv_map"
        self.bucket_boundaries = [cm + "/BatchNorm/beta", s + "/layer6/conv_1/bias", s + "/dense/bias",
                                  s + "/heads/bias"]
        ready = {"layer6/conv_1": 1, "dense": 2, "heads": 3, "layer1/conv_1": 4}

        def on_ready(name, p=p):
            if name in ready:
                # the bucket's weight gradients run on the plan's side lane: joined before an exchange reads them
                p.add_hook(lambda i=ready[name], join=p.make_join(): self._bucket_ready(i, join))

        modelac.record_backward(p, g.g_logit, modelimages.g_output, FLAGS.latent_loss / N if not ae else 0.0,
                                on_ready=on_ready)
        modelimages.record_backward(p)
        p.add_hook(lambda: self._bucket_ready(0))
        g.plan_train = p
        # the calls [head_calls, head_calls + frozen_calls) are the FROZEN trunk (train_step_pipelined's lane A)
        g.head_calls = head_calls
        g.frozen_calls = getattr(modelimages, "frozen_calls", None)
        g.stage_calls = getattr(modelimages, "stage_calls", None) if getattr(modelimages, "stages", 1) == 2 else None
        g.stage_read_calls = getattr(modelimages, "stage_read_calls", None)
        # ---- evaluation step (is_training = 0: moving statistics; MSE only, :411-442) ----
        e = sess.new_plan()
        head(e)
        e.extend(modelimages.plan_eval)
        e.extend(modelac.plan_fwd)
        ops.recon_loss(e, modelac.output, g.acoustic, None, g.sums, count, 1.0, 1.0)
        ops.zero(e, g.chan)
        ops.sqerr_channels(e, modelac.output, g.acoustic, N * 36 * 48, 12, g.chan)
        ops.loss_finalize(e, g.sums, None, N, count, 0.0, 0.0, 1.0, 0.0, g.losses)
        g.plan_eval = e
        g.count = count
        self.graphs[N] = g
        return g

    def _graph_for(self, N):
        if N in self.graphs:
            return self.graphs[N]
        # a different batch size (last partial batch): same variables, new buffers + plans
        # (every constructor argument that shapes the recorded plan comes from the model's own record of them, so the
        #  partial batch runs the same kernel path - two-pass rule included - as the primary graph)
        mi = type(self.modelimages)(**self.modelimages.clone_kwargs())
        ma = type(self.modelac)(input_shape=[36, 48, 12], embedding=self.modelac.embedding,
                                num_skip=self.modelac.num_skip, precision=self.modelac.precision,
                                side_lane=self.modelac.side_lane)
        mi._register = lambda store: None
        ma_heads = self.modelac.heads

        def _reuse(store, ma=ma):
            ma.heads = ma_heads
        ma._register = _reuse
        return self._build_graph(N, mi, ma)

    def _bucket_ready(self, i, join=None):
        if self.comm is not None and not self.hold_exchange:
            if join is not None:
                join()
            self.comm.bucket_ready(i)

    def enable_data_parallel(self, group=None, exchange="auto", force=False):
        """All-reduce the flat gradient across ranks (torch.distributed must be initialised).  exchange:
          "bucketed": five contiguous buckets, each fired from its hook in the recorded backward plan on an exchange stream
                      (the all-reduce overlaps the rest of the backward pass: BASELINE.json's north_star form);
          "whole":    ONE all-reduce of the whole flat gradient behind the backward pass;
          "auto":     bucketed on the one-stream entries (train_step), whole on the pipelined entry, whose trained part
                      already runs a pipeline tick behind the trunk stages (the slack budget).
        force: exchange even at world size 1 (rehearsal of the RCCL path on a one-GPU box)."""
        assert exchange in ("auto", "bucketed", "whole"), exchange
        store = self.session.store
        self.buckets = dp.make_buckets(store.train_ranges(), self.bucket_boundaries)
        assert len(self.buckets) == 5, self.buckets
        self.comm = dp.GradComm(store.grad, self.buckets, group, force=force)
        self.group = group
        self.exchange = exchange
        return self.comm

    def sync_moving_statistics(self):
        """collective: every rank ends up with the mean over ranks of the trunk's BN moving statistics
        (per-replica while training; one set for checkpoints and evaluation: SURVEY §8(e))"""
        self.flush_pipeline()
        store = self.session.store
        rng = [(v.offset, v.numel) for n, v in store.vars.items() if v.group == "state" and "/moving_" in n]
        dp.average_moving_statistics(store.flat["state"], rng, getattr(self, "group", None))
        store.version += 1

    # ------------------------------------------------------------------------------------------------
    # the step
    # ------------------------------------------------------------------------------------------------
    def _feed(self, g, batch, eps):
        acoustic, mfcc, video = batch[0], batch[1], batch[2]
        g.acoustic.copy_(acoustic.reshape(g.N, 36, 48, 12), non_blocking=True)
        g.mfcc.copy_(mfcc.reshape(g.N, 12), non_blocking=True)
        g.video.copy_(video.reshape(g.N, 224, 298, 3), non_blocking=True)
        self._noise(g, eps)

    def _noise(self, g, eps):
        """eps given: copy it in (parity tests); else sample N(0,1) on the device, a fresh Philox
        counter range every step (the reference's tf.random_normal, models/unet_acresnet.py:77)"""
        if eps is not None:
            g.eps.copy_(eps.reshape(g.N, Z), non_blocking=True)
        else:
            self._noise_calls = getattr(self, "_noise_calls", 0) + 1
            rc = _lib.load().acimg_randn(g.eps.data_ptr(), g.N * Z, self.noise_seed, self._noise_calls * 65536,
                                         ops.current_stream_handle(self.session.device))
            _lib.check(rc, "randn")

    def train_step(self, batch=None, eps=None, sync=True, probe=None):
        """One optimisation step (the body of the reference's hot loop, trainer/mfcctrainer.py:343-349).
        batch: (acoustic [N,36,48,12], mfcc [N,12], video [N,224,298,3], ...) or None to reuse the
        tensors already resident in the graph's input buffers.  Returns {mse, huber, latent, reg, loss}
        (python floats; sync=False returns the device tensor instead and does not block)."""
        g = self.primary if batch is None else self._graph_for(int(batch[1].reshape(-1, 12).shape[0]))
        self.flush_pipeline()
        if batch is not None:
            self._feed(g, batch, eps)
        else:
            self._noise(g, eps)
        whole = self.comm is not None and getattr(self, "exchange", "auto") == "whole"
        held, self.hold_exchange = self.hold_exchange, self.hold_exchange or whole
        try:
            if probe is None:
                g.plan_train.run()      # hooks inside fire the bucketed all-reduce when data-parallel
            else:
                g.plan_train.run_probed(probe[0], probe[1])
        finally:
            self.hold_exchange = held
        store = self.session.store
        scale = 1.0
        if self.comm is not None and self.comm.enabled:
            if whole:
                self.comm.allreduce_all()
            else:
                self.comm.wait()
            scale = self.comm.grad_scale
        self.global_step += 1
        lr_t = ops.adam_lr_t(self.learning_rate, self.global_step)
        st = ops.current_stream_handle(self.session.device)
        rc = _lib.load().acimg_adam_step(store.flat["train"].data_ptr(), store.grad.data_ptr(),
                                         store.adam_m.data_ptr(), store.adam_v.data_ptr(),
                                         store.train_numel(), lr_t, 0.9, 0.999, 1e-8, scale, st)
        _lib.check(rc, "adam_step")
        if not sync:
            return g.losses
        return self._scalars(g)

    # ---- software pipeline over HIP streams ------------------------------------------------------------------
    def _pipeline(self, g):
        """Stages: the FROZEN trunk (no trainable variable is read: trainer/mfcctrainer.py:64 keeps it out of var_list),
        as one stage or — `ResNet50Model(stages=2)`, the default — as two (units 1-8 | units 9-16, each with its own
        arena / statistics / tail workspace, a dedicated boundary tensor between them); then the TRAINED part: conv_map,
        the generator, the losses, the backward pass, the gradient exchange, Adam.  One HIP stream per stage (the first
        is the caller's); per call the stages run consecutive batches: trunk stage 1 of batch n, stage 2 of batch n - 1,
        the trained part of batch n - 2.  The arithmetic of every batch is the one-stream step's, bit for bit: a trunk
        stage's inputs (images or the boundary tensor, frozen weights, its own moving statistics, in batch order) do not
        depend on later stages, and every stage processes the batches in order on its own stream.  Buffers that cross a
        stage boundary are written by the producing stage's LAST call, which waits until the consuming stage of the
        previous batch has read them (boundary tensor: after stage 2's first unit; `xfinal`: after the whole trained
        part, whose last kernel — conv_map's weight gradient — reads it)."""
        if self._pipe is not None and self._pipe["g"] is g:
            return self._pipe
        mi = self.modelimages
        if g.frozen_calls is None or not mi._split:
            raise RuntimeError("train_step_pipelined: needs the split-MFMA trunk (precision f16x3 / f16): its frozen part "
                               "uses no shared workspace")
        assert self._pipe is None or not self._pipe["inflight"]
        dev = self.session.device
        full = g.plan_train
        lo, cut = g.head_calls, g.head_calls + g.frozen_calls
        assert full.calls[cut - 1][0] == "bn_add_relu_split", full.calls[cut - 1][0]     # the call that writes xfinal
        cur = torch.cuda.current_stream(dev)
        # one stream per stage beyond the first, one for the trained part and one for its side lane — only streams that
        # were MEASURED to run beside the caller's and beside each other (ops.concurrent_streams): the runtime maps
        # streams onto a few hardware queues (4 by default; RCCL takes some) and two streams on one queue serialise
        want = (2 if g.stage_calls is not None else 1) + 1
        key = (cur.cuda_stream, want)
        if getattr(self, "_lane_cache", None) is None or self._lane_cache[0] != key:
            # (lane_priority: attribute, 0 by default - the lanes beside the caller's stream as high-priority streams was
            #  measured in round 3, profiles/r03/pipeline_lane_priority_r03al.txt)
            self._lane_cache = (key, ops.concurrent_streams(dev, cur, want, priority=getattr(self, "lane_priority", 0)))
        lanes = list(self._lane_cache[1])
        side_b = lanes.pop() if len(lanes) == want else None
        while len(lanes) < want - 1:
            lanes.append(lanes[-1] if lanes else cur)      # too few queues: stages share a lane (slower, still correct)
        stages = []
        if g.stage_calls is not None:
            sc, src = g.head_calls + g.stage_calls, g.head_calls + g.stage_read_calls
            # ... the boundary tensor (a two-pass unit writes it from conv3's second pass)
            assert full.calls[sc - 1][0] in ("bn_add_relu_split", "conv2d_fwd_split3p_tail", "conv2d_fwd_split3p_tail_proj")
            stages.append(dict(parts=[(full.slice(lo, sc - 1), lo, None, None), (full.slice(sc - 1, sc), sc - 1, "x", None)],
                               stream=cur))
            stages.append(dict(parts=[(full.slice(sc, src), sc, None, "x"), (full.slice(src, cut - 1), src, None, None),
                                      (full.slice(cut - 1, cut), cut - 1, "b", None)],
                               stream=lanes[0]))
        else:
            stages.append(dict(parts=[(full.slice(lo, cut - 1), lo, None, None), (full.slice(cut - 1, cut), cut - 1, "b", None)],
                               stream=cur))
        for st in stages:
            st["done"] = torch.cuda.Event()
        pipe = dict(g=g, stages=stages, head=full.slice(0, lo), b=full.slice(cut, len(full.calls)), b_off=cut,
                    sb=lanes[-1], ev={"x": torch.cuda.Event(), "b": torch.cuda.Event()},
                    recorded=set(), inflight=[], finished=[], lanes=len(set(x.cuda_stream for x in lanes + [cur])))
        ops.set_side_lane(dev, pipe["sb"], side_b if side_b is not None else pipe["sb"])
        if pipe["sb"] != cur:
            pipe["sb"].wait_stream(cur)
        for st in stages[1:]:
            if st["stream"] != cur:
                st["stream"].wait_stream(cur)
        self._pipe = pipe
        return pipe

    def _run_trained_part(self, pipe, item, probe=None):
        """the trained part of the oldest batch on its stream (targets copied in behind the previous batch), Adam included"""
        g = pipe["g"]
        with torch.cuda.stream(pipe["sb"]):
            if item["targets"] is not None:
                ac, mf = item["targets"]
                g.acoustic.copy_(ac.reshape(g.N, 36, 48, 12), non_blocking=True)
                g.mfcc.copy_(mf.reshape(g.N, 12), non_blocking=True)
            self._noise(g, item["eps"])
            pipe["sb"].wait_event(pipe["stages"][-1]["done"])      # the trunk has written xfinal for this batch
            pipe["head"].run()
            # data parallel: ONE exchange of the whole flat gradient, issued from THIS stream after the backward pass (no
            # buckets, no separate exchange stream): the trained part has a whole tick of slack behind the trunk stages,
            # so nothing needs to overlap inside it; every collective call costs host time, and every stream beyond the
            # runtime's 4 hardware queues shares a queue with one of the lanes (RCCL's own stream is the fourth)
            bucketed = getattr(self, "exchange", "auto") == "bucketed" and item.get("accum") is None
            held, self.hold_exchange = self.hold_exchange, self.hold_exchange or not bucketed
            try:
                if probe is None:
                    pipe["b"].run()
                else:
                    pipe["b"].run_probed(probe[0], probe[1], offset=pipe["b_off"])
            finally:
                self.hold_exchange = held
            store = self.session.store
            scale = 1.0
            acc = item.get("accum")
            if acc is not None:
                # shard i of n of a strong-scaling step (train_step_sharded): gradients and losses are accumulated on this
                # stream; the exchange and the ONE Adam update follow the last shard
                i, n = acc
                L = _lib.load()
                sth = ops.current_stream_handle(self.session.device)
                if i == 0:
                    self._acc.copy_(store.grad)
                    self._acc_loss = g.losses[:5].clone()
                else:
                    _lib.check(L.acimg_axpy(1.0, store.grad.data_ptr(), self._acc.data_ptr(), store.grad.numel(), sth),
                               "axpy")
                    self._acc_loss = self._acc_loss + g.losses[:5]
                if i + 1 < n:
                    pipe["ev"]["b"].record(pipe["sb"])
                    pipe["recorded"].add("b")
                    return
                store.grad.copy_(self._acc)
                world = 1
                if self.comm is not None and self.comm.enabled:
                    self.comm.allreduce_all()
                    world = self.comm.world
                scale = 1.0 / (n * world)
                if item.get("tag") is not None:
                    pipe["finished"].append((item["tag"], self._acc_loss / n))
            elif self.comm is not None and self.comm.enabled:
                if bucketed:
                    self.comm.wait()
                else:
                    self.comm.allreduce_all()
                scale = self.comm.grad_scale
            self.global_step += 1
            lr_t = ops.adam_lr_t(self.learning_rate, self.global_step)
            rc = _lib.load().acimg_adam_step(store.flat["train"].data_ptr(), store.grad.data_ptr(),
                                             store.adam_m.data_ptr(), store.adam_v.data_ptr(),
                                             store.train_numel(), lr_t, 0.9, 0.999, 1e-8, scale,
                                             ops.current_stream_handle(self.session.device))
            _lib.check(rc, "adam_step")
            if acc is None and item.get("tag") is not None:
                # a private copy of this batch's losses (the graph's buffer is rewritten by the next batch)
                pipe["finished"].append((item["tag"], g.losses[:5].clone()))
            pipe["ev"]["b"].record(pipe["sb"])
        pipe["recorded"].add("b")

    def _run_trunk_stage(self, pipe, k, item, probe=None):
        g, st = pipe["g"], pipe["stages"][k]
        with torch.cuda.stream(st["stream"]):
            if k == 0:
                if item["video"] is not None:
                    g.video.copy_(item["video"].reshape(g.N, 224, 298, 3), non_blocking=True)
            else:
                st["stream"].wait_event(pipe["stages"][k - 1]["done"])
            for plan, off, wait, record in st["parts"]:
                if wait is not None and wait in pipe["recorded"]:
                    st["stream"].wait_event(pipe["ev"][wait])      # the consumer of the previous batch has read it
                if probe is None:
                    plan.run()
                else:
                    plan.run_probed(probe[0], probe[1], offset=off)
                if record is not None:
                    pipe["ev"][record].record(st["stream"])
                    pipe["recorded"].add(record)
            st["done"].record(st["stream"])

    def _advance(self, pipe, new_item, probe=None):
        """one tick: every batch in flight moves one stage on (oldest first, so that an event waited on in this tick was
        recorded for the batch before); returns the loss tensor if a batch finished"""
        out, n = None, len(pipe["stages"])
        flight = pipe["inflight"]
        if flight and flight[0]["stage"] == n:
            self._run_trained_part(pipe, flight.pop(0), probe)
            out = pipe["g"].losses
        for item in flight:                      # oldest first = deepest stage first
            self._run_trunk_stage(pipe, item["stage"], item, probe)
            item["stage"] += 1
        if new_item is not None:
            new_item["stage"] = 0
            self._run_trunk_stage(pipe, 0, new_item, probe)
            new_item["stage"] = 1
            flight.append(new_item)
        return out

    def pop_finished(self):
        """[(tag, {mse, huber, latent, reg, loss})] of the batches submitted with a tag that finished since the last call,
        oldest first (reads behind the trained part's stream, not behind the trunks in flight)"""
        pipe = self._pipe
        if pipe is None or not pipe["finished"]:
            return []
        done, pipe["finished"] = pipe["finished"], []
        return [(tag, self._lane_b_scalars(t)) for tag, t in done]

    def train_step_pipelined(self, batch=None, eps=None, probe=None, tag=None):
        """One call = one batch in, one optimisation step out, in steady state: the frozen trunk of THIS batch starts on
        the caller's stream while the later stages of the previous batches run on theirs (`_pipeline`).  Returns the
        device tensor of the losses of the batch that finished in this call (None while the pipeline fills);
        `flush_pipeline()` finishes the batches in flight.  batch None: reuse the images / targets resident in the
        graph's buffers (bench.py).  tag: keep this batch's losses for `pop_finished()` (the training log)."""
        g = self.primary if batch is None else self._graph_for(int(batch[1].reshape(-1, 12).shape[0]))
        out = None
        carried = []
        if self._pipe is not None and self._pipe["g"] is not g:
            out = self.flush_pipeline()      # another batch size (the last, partial batch): finish the ones in flight
            carried, self._pipe = self._pipe["finished"], None
        pipe = self._pipeline(g)
        pipe["finished"] = carried + pipe["finished"]
        # the targets are consumed two calls later, on the trained part's stream: they are staged NOW (a device copy per
        # batch in flight), so a loader may reuse its host buffers as soon as this call returns; the images are copied
        # in this call (trunk stage 1 runs in it)
        targets = None
        if batch is not None:
            targets = (self._stage_for_lane_b(pipe, batch[0]), self._stage_for_lane_b(pipe, batch[1]))
            if eps is not None:
                eps = self._stage_for_lane_b(pipe, eps)
        item = dict(video=None if batch is None else batch[2], targets=targets, eps=eps, tag=tag)
        res = self._advance(pipe, item, probe)
        return res if res is not None else out

    def _stage_for_lane_b(self, pipe, x):
        """device copy of a host / device tensor made NOW on the caller's stream and read two ticks later by the trained
        part's stream: `record_stream` tells the caching allocator about that second stream, so the block is not handed
        to a later staging copy (on the caller's stream) before the trained part's D2D copy of it has run"""
        t = torch.as_tensor(x).to(self.session.device, torch.float32, copy=True)
        if pipe["sb"] != torch.cuda.current_stream(self.session.device):
            t.record_stream(pipe["sb"])
        return t

    def _lane_b_scalars(self, losses):
        """python floats of a loss tensor produced on lane B (read behind that lane, not behind the trunk in flight)"""
        with torch.cuda.stream(self._pipe["sb"]):
            v = losses[:5].tolist()
        return OrderedDict(mse=v[0], huber=v[1], latent=v[2], reg=v[3], loss=v[4])

    def flush_pipeline(self):
        """finish the batches in flight (oldest first) and make the current stream wait for every lane; returns the
        losses of the last one (device tensor) or None"""
        pipe, out = self._pipe, None
        if pipe is None:
            return None
        while pipe["inflight"]:
            res = self._advance(pipe, None)
            out = res if res is not None else out
        cur = torch.cuda.current_stream(self.session.device)
        for st in pipe["stages"]:
            if st["stream"] != cur:
                cur.wait_stream(st["stream"])
        cur.wait_stream(pipe["sb"])
        return out

    def train_step_sharded(self, shards, eps=None, probe=None, pipelined=False, tag=None):
        """Strong-scaling step: `shards` = this rank's micro-batches (each of the primary graph's size: the BN group),
        run one after the other with their gradients accumulated; ONE exchange of the accumulated gradient, ONE Adam
        update with scale 1 / (shards * world).  With one shard per rank this is `train_step` (overlapped buckets).
        pipelined: the shards go through the lanes of `train_step_pipelined` - inside a step every shard sees the same
        weights and the frozen trunk reads none of them, so the trunk of shard i + 1 (and of the next step's first
        shards) runs beside the trained part of shard i; gradients and losses are accumulated on the trained part's
        stream in shard order: bit-identical to the one-stream form.  Returns None then (`flush_pipeline()` finishes the
        shards in flight; `pop_finished()` returns the mean losses of the steps submitted with a `tag`)."""
        if pipelined:
            g = self.primary
            if self._pipe is not None and self._pipe["g"] is not g:
                self.flush_pipeline()
                self._pipe = None
            pipe = self._pipeline(g)
            if self._acc is None:
                self._acc = torch.zeros_like(self.session.store.grad)
            for i, b in enumerate(shards):
                e = None if eps is None else self._stage_for_lane_b(pipe, eps[i])
                item = dict(video=b[2], targets=(self._stage_for_lane_b(pipe, b[0]), self._stage_for_lane_b(pipe, b[1])),
                            eps=e, tag=tag, accum=(i, len(shards)))
                self._advance(pipe, item, probe)
            return None
        if len(shards) == 1:
            return self.train_step(shards[0], eps=None if eps is None else eps[0], sync=probe is None, probe=probe)
        self.flush_pipeline()          # a batch in flight reads the input buffers this call is about to overwrite
        store = self.session.store
        g = self.primary
        if self._acc is None:
            self._acc = torch.zeros_like(store.grad)
        L = _lib.load()
        st = ops.current_stream_handle(self.session.device)
        self.hold_exchange = True
        tot = None
        try:
            for i, b in enumerate(shards):
                self._feed(g, b, None if eps is None else eps[i])
                if probe is None:
                    g.plan_train.run()
                else:
                    g.plan_train.run_probed(probe[0], probe[1])
                if i == 0:
                    self._acc.copy_(store.grad)
                else:
                    _lib.check(L.acimg_axpy(1.0, store.grad.data_ptr(), self._acc.data_ptr(), store.grad.numel(), st),
                               "axpy")
                tot = g.losses[:5].clone() if tot is None else tot + g.losses[:5]
        finally:
            self.hold_exchange = False
        store.grad.copy_(self._acc)
        world = 1
        if self.comm is not None and self.comm.enabled:
            self.comm.allreduce_all()
            world = self.comm.world
        self.global_step += 1
        lr_t = ops.adam_lr_t(self.learning_rate, self.global_step)
        rc = L.acimg_adam_step(store.flat["train"].data_ptr(), store.grad.data_ptr(), store.adam_m.data_ptr(),
                               store.adam_v.data_ptr(), store.train_numel(), lr_t, 0.9, 0.999, 1e-8,
                               1.0 / (len(shards) * world), st)
        _lib.check(rc, "adam_step")
        v = (tot / len(shards)).tolist()
        return OrderedDict(mse=v[0], huber=v[1], latent=v[2], reg=v[3], loss=v[4])

    def _scalars(self, g):
        v = g.losses[:5].tolist()
        return OrderedDict(mse=v[0], huber=v[1], latent=v[2], reg=v[3], loss=v[4])

    def eval_step(self, batch=None, eps=None):
        """forward in inference mode (BN moving statistics); returns mse + the four per-3-channel MSEs"""
        g = self.primary if batch is None else self._graph_for(int(batch[1].reshape(-1, 12).shape[0]))
        self.flush_pipeline()
        if batch is not None:
            self._feed(g, batch, eps)
        else:
            self._noise(g, eps)
        g.plan_eval.run()
        mse = float(g.losses[0])
        ch = g.chan[:12].tolist()
        res = OrderedDict(mse=mse)
        per = g.N * 36 * 48 * 3
        for i in range(4):
            res["mse%d" % i] = sum(ch[3 * i:3 * i + 3]) / per
        return res

    # ------------------------------------------------------------------------------------------------
    # reference protocol: loops, checkpoints
    # ------------------------------------------------------------------------------------------------
    def _retrieve_batch(self, next_batch):
        if FLAGS.model == 'UNet':
            mfcc = next_batch[1].reshape(-1, 12)
            images = next_batch[2].reshape(-1, 224, 298, 3)
            acoustic = next_batch[0].reshape(-1, 36, 48, 12)
            labels = next_batch[3].reshape(mfcc.shape[0], -1).argmax(1)
            scenario = next_batch[4].reshape(mfcc.shape[0], -1).argmax(1)
        else:
            raise ValueError('Unknown model type')
        return acoustic, mfcc, images, labels, scenario

    def _init_model(self, session):
        """random init, then optional partial restores (trainer/mfcctrainer.py:163-234)"""
        self.modelimages.initialize()
        self.modelac.initialize()
        if FLAGS.init_checkpoint is not None:
            self.modelac.init_model(session, FLAGS.init_checkpoint)
        elif FLAGS.acoustic_init_checkpoint is not None or FLAGS.visual_init_checkpoint is not None:
            if FLAGS.acoustic_init_checkpoint is not None:
                self.modelac.init_model(session, FLAGS.acoustic_init_checkpoint)
            if FLAGS.visual_init_checkpoint is not None:
                self.modelimages.init_model(session, FLAGS.visual_init_checkpoint)
        elif FLAGS.restore_checkpoint is not None:
            self._restore_model(session)

    def _restore_model(self, session):
        """model variables only: Adam slots and global_step start afresh (:236-247)"""
        from .vision import load_state_file
        state = load_state_file(FLAGS.restore_checkpoint)
        session.store.load_state(state, strict=False)

    def _save_checkpoint(self, session, epoch):
        """`self.saver.save(session, '<checkpoint_dir>/<exp_name>/epoch_<n>.ckpt')` of trainer/mfcctrainer.py:400-406,
        :81 (Saver(max_to_keep=11)): a TensorFlow Saver-V2 bundle (epoch_<n>.ckpt.index + .data-00000-of-00001 + the
        `checkpoint` state file) holding what the reference's Saver holds — every model variable under its TF name,
        the Adam slots `<var>/Adam`, `<var>/Adam_1`, `beta1_power`, `beta2_power` and `global_step` — so the
        reference's own `_restore_model` / `init_model` (:214-247) can read it.  Data-parallel: every rank calls this
        (the BN moving statistics are averaged over ranks first, a collective); only rank 0 writes."""
        from . import tfio
        self.flush_pipeline()
        if self.comm is not None and self.comm.enabled:
            self.sync_moving_statistics()
        if not dp.is_writer(getattr(self, "group", None)):
            return None
        checkpoint_dir = '{}/{}'.format(FLAGS.checkpoint_dir, FLAGS.exp_name)
        os.makedirs(checkpoint_dir, exist_ok=True)
        model_name = 'epoch_{}.ckpt'.format(epoch)
        self.log('{}: {} - Saving model to {}/{}'.format(datetime.now(), FLAGS.exp_name, checkpoint_dir, model_name))
        store = session.store
        import numpy as np
        tensors = OrderedDict((k, v.numpy()) for k, v in store.state_dict().items())
        train = set(n for n in store.grad_dict())
        for which, sfx in (("m", "/Adam"), ("v", "/Adam_1")):
            for k, v in store.slot_dict(which).items():
                if k in train:
                    tensors[k + sfx] = v.numpy()
        t = max(int(self.global_step), 0)
        tensors["beta1_power"] = np.float32(0.9 ** (t + 1))     # TF-1 Adam: the powers start at beta and are
        tensors["beta2_power"] = np.float32(0.999 ** (t + 1))   # multiplied once per apply_gradients
        tensors["global_step"] = np.int64(t)
        path = '{}/{}'.format(checkpoint_dir, model_name)
        tfio.write_checkpoint(path, tensors)
        self._saved = [p for p in getattr(self, "_saved", []) if p != path] + [path]
        while len(self._saved) > 11:   # Saver(max_to_keep=11), :81
            old = self._saved.pop(0)
            for f in (old + ".index", old + ".data-00000-of-00001"):
                if os.path.exists(f):
                    os.remove(f)
        with open(os.path.join(checkpoint_dir, "checkpoint"), "w") as f:   # tf.train.CheckpointState, text format
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(self._saved[-1]))
            for p in self._saved:
                f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
        return path

    def train(self, train_data=None, valid_data=None):
        assert train_data is not None
        assert valid_data is not None
        if not self.graphs:
            self._build_functions(train_data)
        session = self.session
        self._init_model(session)
        if FLAGS.checkpoint_dir is not None:
            self._save_checkpoint(session, 'random')
        start_epoch = int(self.global_step)
        best_epoch, best_loss = -1, 10000
        for epoch in range(start_epoch, start_epoch + self.num_epochs):
            step = 0
            pipelined = bool(getattr(self.modelimages, "_split", False)) and FLAGS.pipeline

            def report(i, r):
                if i % self.display_freq == 0:
                    self.log('{}: {} - Iteration: [{:3}]\t Training_mse_Loss: {:6f}\t Training_Loss: {:6f}'.format(
                        datetime.now(), FLAGS.exp_name, i, r["mse"], r["loss"]))

            for next_batch in train_data.data:
                acoustic, mfcc, images, _, _ = self._retrieve_batch(next_batch)
                if pipelined:
                    # this batch's trunk starts beside the later stages of the batches before it: the line of iteration
                    # i is written when that batch finishes, a few calls later (same numbers, same order)
                    self.train_step_pipelined((acoustic, mfcc, images),
                                              tag=step if step % self.display_freq == 0 else None)
                    for i, r in self.pop_finished():
                        report(i, r)
                else:
                    report(step, self.train_step((acoustic, mfcc, images)))
                step += 1
            if pipelined:
                self.flush_pipeline()
                for i, r in self.pop_finished():
                    report(i, r)
            if self.comm is not None and self.comm.enabled:
                # data parallel: validate with ONE set of BN moving statistics (their mean over ranks, a collective every
                # rank enters here, unconditionally); _evaluate reduces its sums over ranks, so `total_loss` - and with
                # it every branch below that contains a collective (_save_checkpoint) - is the same on all ranks
                self.sync_moving_statistics()
            total_loss = self._evaluate(session, 'validation', valid_data)
            self.log('{}: {} - Epoch: {}\t Validation_mse_Loss: {:6f}'.format(datetime.now(), FLAGS.exp_name, epoch,
                                                                              total_loss))
            if FLAGS.checkpoint_dir is not None:
                if epoch % 10 == 0:
                    self._save_checkpoint(session, epoch)
                if total_loss <= best_loss:
                    best_epoch, best_loss = epoch, total_loss
                    self._save_checkpoint(session, epoch)
                    if dp.is_writer(getattr(self, "group", None)):
                        with open('{}/{}'.format(FLAGS.checkpoint_dir, FLAGS.exp_name) + "/model.txt", "w") as outfile:
                            outfile.write('{}: {}\nBest Epoch: {}\nValidation_mse_Loss: {:6f}\n'.format(
                                datetime.now(), FLAGS.exp_name, best_epoch, best_loss))
            elif total_loss <= best_loss:
                best_epoch, best_loss = epoch, total_loss
        self.log('{}: {} - Best Epoch: {}\t Validation_mse_Loss: {:6f}'.format(datetime.now(), FLAGS.exp_name,
                                                                            best_epoch, best_loss))
        return best_loss

    def _valid(self, session, data):
        return self._evaluate(session, 'validation', data)

    def _evaluate(self, session, mod, data):
        """size-weighted mean of the per-batch MSE (trainer/mfcctrainer.py:411-442)"""
        loss_sum, data_set_size = 0.0, 0
        for next_batch in data.data:
            acoustic, mfcc, images, labels, _ = self._retrieve_batch(next_batch)
            r = self.eval_step((acoustic, mfcc, images))
            n = labels.shape[0]
            data_set_size += n
            loss_sum += r["mse"] * n
        if self.comm is not None and self.comm.enabled:     # every rank validates its own shard of `data`
            loss_sum, data_set_size = dp.allreduce_sums([loss_sum, data_set_size], self.session.device,
                                                        getattr(self, "group", None))
        return loss_sum / data_set_size

    def test(self, test_data=None):
        assert test_data is not None
        if not self.graphs:
            self._build_functions(test_data)
        session = self.session
        self.modelimages.initialize()
        self.modelac.initialize()
        if FLAGS.restore_checkpoint is not None:
            self._restore_model(session)
        sums = [0.0] * 5
        data_set_size = 0
        for next_batch in test_data.data:
            acoustic, mfcc, images, labels, _ = self._retrieve_batch(next_batch)
            r = self.eval_step((acoustic, mfcc, images))
            n = labels.shape[0]
            data_set_size += n
            for i, k in enumerate(("mse", "mse0", "mse1", "mse2", "mse3")):
                sums[i] += r[k] * n
        test_loss, l0, l1, l2, l3 = [s / data_set_size for s in sums]
        line = '{} - Testing_Loss: {:6f}\t  Testing_Loss0: {:6f}\t Testing_Loss1: {:6f}\t Testing_Loss2: {:6f}\t ' \
               'Testing_Loss3: {:6f}'.format(datetime.now(), test_loss, l0, l1, l2, l3)
        if FLAGS.restore_checkpoint is not None and dp.is_writer(getattr(self, "group", None)):
            name_folder = str.join('/', FLAGS.restore_checkpoint.split('/')[:-1])
            tag = FLAGS.restore_checkpoint.split('/')[-1].split('.')[0].split('_')[-1]
            with open('{}'.format(name_folder) + "/test_accuracy_{}.txt".format(tag), "w") as outfile:
                outfile.write(line)
        self.log(line)
        return test_loss
