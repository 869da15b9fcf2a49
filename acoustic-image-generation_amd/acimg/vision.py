"""`ResNet50Model`: the modified ResNet-v1-50 image encoder, MI355X-native.

Mirrors the reference's model protocol (models/vision.py:8-71): `scope`, `init_model(session,
checkpoint_file)`, `_build_model(visual_images)` which sets `output`, `network`, `train_vars`
(conv_map only) and `train_vars2` (the frozen trunk).  Architecture: models/resnet50.py:75-125
(bottleneck), :205-209 (7x7/2 stem, 3x3/2 pool, conv_map 3x4 VALID 2048->12 + BN + ReLU), :261-266
(block strides 1,2,2,1 placed in each block's last unit).

MI355X design (f16x3, the default): every trunk conv is one implicit-GEMM launch on PRE-SPLIT operands (two fp16
planes per tensor in LDS-tile order, "bricks": csrc/igemm_split3d_kernel.hpp) that writes the RAW fp32 conv output
once and leaves per-row-block (sum, sum^2) partials; `bn_finalize` turns them into per-channel scale/shift (and
advances the moving averages); ONE elementwise pass per tensor (`bn_relu_split`, `bn_relu_maxpool_split`,
`bn_add_relu_split`) applies scale/shift (+ shortcut) + ReLU and writes the consumer's operand planes, so a normalised
activation is written exactly once, in the format the next K loop copies into LDS.  conv3 of a bottleneck never writes
its fp32 output when its batch statistics are known beforehand (the statistics pass of round 3, the input-side Gram
statistics of round 4): BN + shortcut + ReLU + split then run in the conv's own epilogue (`*_tail`, `*_tail_proj`).
Activations live in a few reusable arenas (about 1.2 GB at batch 32), so consecutive layers hit the 256 MiB Infinity
Cache instead of streaming 6.5 GB of distinct tensors.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .params import Var, up4
from .session import get_default_session

BLOCKS = (("block1", 64, 3, 1), ("block2", 128, 4, 2), ("block3", 256, 6, 2), ("block4", 512, 3, 1))
WEIGHT_DECAY = 5e-4   # resnet_arg_scope(weight_decay=5e-4), models/vision.py:54
BN_DECAY = 0.997
BN_EPS = 1e-5
PLANE_SLACK = 2 * 15 * 2048 * 2 + 512   # a split-format plane ends on a whole 16-pixel brick: <= 15 pixels more, twice


class ResNet50Model(object):

    def __init__(self, input_shape=None, num_classes=None, precision="f16x3", stages=None, stage_cut=8, side_lane=True,
                 two_pass=True, two_pass_max_cin=256, gram=True):
        """precision: arithmetic of the frozen trunk convs — "f16x3" (split-fp16 MFMA, fp32-class
        results, default), "f32" (exact-f32 MFMA) or "f16" (fp16 OPERAND STORAGE: the same data path, but the 52 trunk
        convs fetch and multiply only the hi fp16 plane of activations and weights — one MFMA per product, fp32
        accumulation, fp32 batch-norm statistics; BASELINE configs[4]'s "fp16 with fp32 loss accumulation").  With
        "f16x3" / "f16" the stem runs as a row-run conv and conv_map as a tap GEMM on the split-MFMA kernels
        (_stem_bn, _conv_map_tap; those two layers keep the three-term product)."""
        self.scope = 'resnet_v1_50'
        assert precision in ("f16x3", "f32", "f16")
        self.precision = precision
        self._split = precision in ("f16x3", "f16")
        self._terms = 1 if precision == "f16" else 3
        # stages = 2: the frozen trunk is recorded as TWO pipeline stages (units 1-8 | units 9-16) with an activation
        # arena, a statistics buffer and a tail workspace each and a dedicated boundary tensor, so that the trainer can
        # run stage 1 of batch t + 2 beside stage 2 of batch t + 1 (Trainer.train_step_pipelined): two trunk kernels of
        # different depth fill each other's partial last rounds.  stages = 1: one arena; the four projection shortcuts
        # then run on the plan's side lane beside conv1 .. conv3 (with two stages the stage streams take that role and
        # the 4 hardware queues are spent on the pipeline's lanes).
        if stages is None:
            stages = 2 if precision in ("f16x3", "f16") else 1
        assert stages in (1, 2)
        if precision not in ("f16x3", "f16"):
            stages = 1
        self.stages = stages
        # units in stage 1: 8 = blocks 1 + 2 + the first unit of block 3 (measured: 5: 7.53, 6: 7.30, 7: 7.18, 8: 7.06,
        # 9: 7.19 ms/step; the two trunk stages and the trained part should take about the same time)
        # (constructor arguments, not environment variables: stages, stage_cut, side_lane)
        self.STAGE_CUT = int(stage_cut)
        self.side_lane = bool(side_lane) and stages == 1
        # two_pass: conv3 of the identity units runs twice - statistics only, then again with BN + shortcut + ReLU + split
        # in its epilogue (ops.conv2d_fwd_split3p_stats / _tail) - instead of conv + bn_add_relu_split: the unit's widest
        # tensor never exists in fp32 (f16x3 only; units with a subsampled shortcut and the last unit keep the pass)
        self.two_pass = bool(two_pass) and precision == "f16x3"
        # widest conv3 INPUT that takes the two passes: with the chip to itself the second K loop pays up to 128 channels; in
        # the pipelined step, where three lanes share HBM, 256 is as fast (6.617 vs 6.632 ms, three alternating runs each:
        # profiles/r03/pipeline_two_pass_rule_r03ao.txt) and moves 1.4 GB less per step; 512 is slower (6.656 ms)
        self.two_pass_max_cin = int(two_pass_max_cin)
        # gram (round 4): the statistics of those conv3 layers come from the Gram matrix of the conv's INPUT
        # (ops.gram_stats: P C^2 MACs at two MFMAs per product + two small launches) instead of a statistics pass that
        # repeats the conv's K loop (4 P C^2 MACs at three): the conv then runs ONCE, with the fused tail.  False = round
        # 3's two K loops (kept for the A/B tables and the bit-exactness tests of the pass form).
        self.gram = bool(gram) and self.two_pass
        self._ctor = dict(input_shape=list(input_shape), num_classes=num_classes, precision=precision, stages=stages,
                          stage_cut=int(stage_cut), side_lane=bool(side_lane), two_pass=bool(two_pass),
                          two_pass_max_cin=int(two_pass_max_cin), gram=bool(gram))
        self.num_classes = num_classes
        self.height = input_shape[0]
        self.width = input_shape[1]
        self.channels = input_shape[2]
        if num_classes is not None:
            raise NotImplementedError("the logits layer is not on the acoustic-image path (num_classes=None)")
        self.session = None
        self.output = None
        self.network = None

    def clone_kwargs(self):
        """constructor arguments of a second instance that records the same plan shape (Trainer._graph_for: another
        batch size over the same variables)"""
        return dict(self._ctor)

    # ---- variable inventory (TF names) --------------------------------------------------------------
    def _units(self):
        depth_in = 64
        for name, base, n, stride in BLOCKS:
            for u in range(n):
                s = stride if u == n - 1 else 1
                yield ("%s/%s/unit_%d/bottleneck_v1" % (self.scope, name, u + 1), depth_in, base * 4, base, s)
                depth_in = base * 4

    def _conv_layers(self):
        yield (self.scope + "/conv1", 7, 7, 3, 64)
        for scope, din, d, db, s in self._units():
            if din != d:
                yield (scope + "/shortcut", 1, 1, din, d)
            yield (scope + "/conv1", 1, 1, din, db)
            yield (scope + "/conv2", 3, 3, db, db)
            yield (scope + "/conv3", 1, 1, db, d)
        yield (self.scope + "/conv_map", 3, 4, 2048, 12)

    def _register(self, store):
        cm = self.scope + "/conv_map"
        # trainable first (conv_map/* is in the optimiser's var_list, trainer/mfcctrainer.py:64)
        store.add(Var(cm + "/weights", (3, 4, 2048, 12), "conv", "train"))
        store.add(Var(cm + "/BatchNorm/gamma", (12,), "vec", "train"))
        store.add(Var(cm + "/BatchNorm/beta", (12,), "vec", "train"))
        store.add(Var(cm + "/BatchNorm/moving_mean", (12,), "vec", "state"))
        store.add(Var(cm + "/BatchNorm/moving_variance", (12,), "vec", "state"))
        for scope, kh, kw, cin, cout in self._conv_layers():
            if scope == cm:
                continue
            store.add(Var(scope + "/weights", (kh, kw, cin, cout), "conv", "trunkw"))
            for sfx in ("gamma", "beta", "moving_mean", "moving_variance"):
                store.add(Var(scope + "/BatchNorm/" + sfx, (cout,), "vec", "state"))

    # ---- reference protocol ------------------------------------------------------------------------
    def init_model(self, session, checkpoint_file):
        """Initialise from a TF-named state dict, everything except logits / conv_map
        (models/vision.py:20-42).  checkpoint_file: path to .npz / torch file, or a dict."""
        state = load_state_file(checkpoint_file)
        store = (session or self.session).store
        return store.load_state(state, strict=False,
                                only=lambda n: n.startswith(self.scope + "/") and "/logits" not in n and "/conv_map" not in n)

    def initialize(self, seed=1238):
        """slim defaults: variance_scaling_initializer conv kernels, gamma 1, beta 0, moving 0/1"""
        g = torch.Generator().manual_seed(seed)
        state = OrderedDict()
        for scope, kh, kw, cin, cout in self._conv_layers():
            fan_in = kh * kw * cin
            t = torch.randn(kh, kw, cin, cout, generator=g, dtype=torch.float64).clamp_(-2, 2)
            state[scope + "/weights"] = (t * np.sqrt(1.3 * 2.0 / fan_in)).float()
            state[scope + "/BatchNorm/gamma"] = torch.ones(cout)
            state[scope + "/BatchNorm/beta"] = torch.zeros(cout)
            state[scope + "/BatchNorm/moving_mean"] = torch.zeros(cout)
            state[scope + "/BatchNorm/moving_variance"] = torch.ones(cout)
        self.session.store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def _build_model(self, visual_images, session=None):
        """visual_images: device buffer [N,224,298,3] float32 (the feed target).  Records the forward
        plans (training / inference BN mode) and the conv_map backward plan."""
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        N = visual_images.shape[0]
        assert tuple(visual_images.shape[1:]) == (self.height, self.width, self.channels)
        self.N = N
        self._alloc(sess, visual_images)
        self.plan_train = sess.new_plan()
        self.plan_eval = sess.new_plan()
        self._record_forward(self.plan_train, True)
        self._record_forward(self.plan_eval, False)
        self.wsplit = torch.zeros(max(self._sp3_bytes, 16), dtype=torch.uint8, device=sess.device)
        self.network = OrderedDict(input=visual_images, is_training=None, keep_prob=None)
        self.network[self.scope + "/conv_map"] = self.output
        cm = self.scope + "/conv_map"
        self.train_vars = [cm + "/weights", cm + "/BatchNorm/gamma", cm + "/BatchNorm/beta"]
        self.train_vars2 = [n for n, v in sess.store.vars.items()
                            if v.group == "trunkw" or (v.group == "state" and n.endswith(("gamma", "beta")))]

    # ---- buffers -------------------------------------------------------------------------------------
    def _alloc(self, sess, images):
        N, H, W = self.N, self.height, self.width
        z = sess.zeros
        self.images = images
        self.xpad = z(N, H, W, 4)
        oh1, ow1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        self.stem_hw = (oh1, ow1)
        self.raw0 = z(N, oh1, ow1, 64)
        ph, _ = ops.same_out_pad(oh1, 3, 2)
        pw, _ = ops.same_out_pad(ow1, 3, 2)
        self.pool_hw = (ph, pw)
        # arena sizes over all units
        h, w = ph, pw
        mx_io = N * h * w * 64
        mx_r1 = mx_r2 = mx_r3 = 0
        nch = 64
        for scope, din, d, db, s in self._units():
            oh, _ = ops.same_out_pad(h, 1, s)
            ow, _ = ops.same_out_pad(w, 1, s)
            mx_r1 = max(mx_r1, N * h * w * db)
            mx_r2 = max(mx_r2, N * oh * ow * db)
            mx_r3 = max(mx_r3, N * oh * ow * d)
            mx_io = max(mx_io, N * h * w * din, N * oh * ow * d)
            h, w = oh, ow
            nch += (d if din != d else 0) + 2 * db + d
        self.final_hw = (h, w)
        self.arena_a = z(mx_io)
        self.arena_b = z(mx_io)
        self.arena_r1 = z(mx_r1)
        self.arena_r2 = z(mx_r2)
        self.arena_r3 = z(mx_r3)
        self.arena_sc = z(mx_r3)
        if self._split:
            # split-format arenas (two fp16 planes per tensor = the bytes of the fp32 tensor)
            u8 = lambda n: torch.zeros(int(n), dtype=torch.uint8, device=sess.device)  # noqa: E731
            self.planes_a, self.planes_b = u8(4 * mx_io + PLANE_SLACK), u8(4 * mx_io + PLANE_SLACK)
            self.planes_1, self.planes_2 = u8(4 * mx_r1 + PLANE_SLACK), u8(4 * mx_r2 + PLANE_SLACK)
            if self.stages == 2:
                # stage 2's own arenas (sized for its units) and the tensor that crosses the stage boundary
                hh, ww = ph, pw
                io2 = r12 = r22 = r32 = xb = 0
                for k, (scope, din, d, db, s_) in enumerate(self._units()):
                    oh, _ = ops.same_out_pad(hh, 1, s_)
                    ow, _ = ops.same_out_pad(ww, 1, s_)
                    if k == self.STAGE_CUT - 1:
                        xb = N * oh * ow * d
                    if k >= self.STAGE_CUT:
                        r12, r22 = max(r12, N * hh * ww * db), max(r22, N * oh * ow * db)
                        r32, io2 = max(r32, N * oh * ow * d), max(io2, N * oh * ow * d)
                    hh, ww = oh, ow
                self.planes_x = u8(4 * xb + PLANE_SLACK)
                self.planes_a2, self.planes_b2 = u8(4 * io2 + PLANE_SLACK), u8(4 * io2 + PLANE_SLACK)
                self.planes_12, self.planes_22 = u8(4 * r12 + PLANE_SLACK), u8(4 * r22 + PLANE_SLACK)
                self.arena_r12, self.arena_r22 = z(r12), z(r22)
                self.arena_r32, self.arena_sc2 = z(r32), z(r32)
        self.xfinal = z(N, h, w, 2048)          # block4 output, kept for the conv_map weight gradient
        self.affine = z(2 * (nch + 64 + 16))    # scale/shift of every BN layer
        self._aff_off = 0
        fh, fw = h - 3 + 1, w - 4 + 1
        self.feat_hw = (fh, fw)
        self.raw_cm = z(N, fh, fw, 12)
        self.output = z(N, fh, fw, 12)
        self.save_mean = z(16)
        self.save_invstd = z(16)
        self.g_output = z(N, fh, fw, 12)        # filled by the generator's backward (min-max bwd)
        self.g_raw_cm = z(N, fh, fw, 12)
        self.stats = None
        self._stats_need = 0
        if self._split:
            # stem as a row-run conv on the split-MFMA kernel: zero-padded 4-channel frame (+ slack for the last run),
            # kernel re-laid as [7][32 = 7 pixels x 4 channels + 4 zero rows][64]
            self.stem_frame = z(N * (H + 6) * (W + 6) * 4 + 64)
            self.stem_w = z(7, 1, 32, 64)
            self._stem_d = ops.conv_desc(N, H + 6, W + 6, 32, 64, 7, 1, 2, "VALID", ldx=4, ldy=64, ldw=64)
            self._stem_d.OH, self._stem_d.OW = oh1, ow1
            # conv_map as a tap GEMM (ops.tapconv_*): packed kernel, its split image, per-input-pixel products
            self._cm_d = ops.conv_desc(N, h, w, 2048, 12, 3, 4, 1, "VALID", ldx=2048, ldy=12, ldw=12)
            self._cm_d1 = ops.conv_desc(N, h, w, 2048, 144, 1, 1, 1, "SAME", ldx=2048, ldy=144, ldw=144)
            self.cm_wt = z(2048, 144)
            self.cm_wsplit = torch.zeros(int(ops.conv2d_split3_weight_bytes(self._cm_d1)), dtype=torch.uint8,
                                         device=sess.device)
            self.cm_z = z(N * h * w, 144)
            self.cm_gz = z(N * h * w, 144)
            self.cm_dwt = z(2048, 144)
        # f16x3: split + transposed copies of the frozen kernels, refreshed when the store changes
        self._sp3 = {}
        self._sp3_bytes = 0
        self._sp3_version = -1
        self.plan_prepare = sess.new_plan()
        self.wsplit = None

    def _new_affine(self, c):
        sc = self.affine[self._aff_off:self._aff_off + c]
        sh = self.affine[self._aff_off + c:self._aff_off + 2 * c]
        self._aff_off += 2 * c
        assert self._aff_off <= self.affine.numel()
        return sc, sh

    # ---- plan recording ------------------------------------------------------------------------------
    def _conv_bn(self, plan, scope, x, hw, cin, kh, kw, cout, stride, padding, out, in_aff, training,
                 save=False):
        """raw conv + BN statistics -> (scale, shift) of this layer; returns (OH, OW, scale, shift)"""
        st = self.session.store
        P = lambda n: ops.LazyPtr(lambda n=n: st.p(n))  # noqa: E731
        d = ops.conv_desc(self.N, hw[0], hw[1], cin, cout, kh, kw, stride, padding, ldx=cin, ldy=up4(cout),
                          ldw=up4(cout))
        sp3 = False   # the f16x3 trunk goes through _conv_bn_planes; conv1 (C=3) and conv_map stay exact-f32
        rows = ops.conv2d_fwd_split3_stats_rows(d) if sp3 else ops.conv2d_stats_rows(d)
        self._stats_need = max(self._stats_need, rows * 2 * up4(cout))
        stats = ops.LazyPtr(lambda: self.stats)
        isc, ish = (in_aff if in_aff is not None else (None, None))
        if sp3:
            if scope not in self._sp3:
                off = self._sp3_bytes
                self._sp3[scope] = off
                self._sp3_bytes += -(-ops.conv2d_split3_weight_bytes(d) // 256) * 256
                ops.conv2d_split3_prepare(self.plan_prepare, d, P(scope + "/weights"),
                                       ops.LazyPtr(lambda off=off: self.wsplit[off:]))
            off = self._sp3[scope]
            ops.conv2d_fwd_split3(plan, d, x, ops.LazyPtr(lambda off=off: self.wsplit[off:]), out, isc, ish, 1,
                               stats if training else None)
        else:
            ops.conv2d_fwd(plan, d, x, P(scope + "/weights"), None, out, isc, ish, 1,
                           stats if training else None)
        if not hasattr(self, "_aff_cache"):
            self._aff_cache = {}
        if scope not in self._aff_cache:
            self._aff_cache[scope] = self._new_affine(up4(cout))
        sc, sh = self._aff_cache[scope]
        b = scope + "/BatchNorm/"
        ops.bn_finalize(plan, stats if training else None, rows if training else 0, cout, up4(cout),
                        self.N * d.OH * d.OW if training else 0, P(b + "gamma"), P(b + "beta"),
                        P(b + "moving_mean"), P(b + "moving_variance"), sc, sh, BN_DECAY, BN_EPS, training,
                        self.save_mean if save else None, self.save_invstd if save else None)
        return d.OH, d.OW, sc, sh

    def _refresh_split_weights(self):
        """host hook at the head of the forward plans: re-split the frozen kernels iff they changed"""
        st = self.session.store
        if self._sp3_bytes and self._sp3_version != st.version:
            if self._split:      # [7][7][4][64] -> [7][28 of 32][64] (parameter plumbing, load time only)
                self.stem_w.view(7, 32, 64)[:, :28].copy_(st.p(self.scope + "/conv1/weights").reshape(7, 28, 64))
            self.plan_prepare.run()
            self._sp3_version = st.version

    @staticmethod
    def _lo_off(rows, c):
        """byte offset of the lo plane of a [rows, c] split-format tensor = acimg_split_plane_bytes(rows, c): planes are
        whole 16-pixel x 32-channel bricks (include/acimg.h, pre-split activation format)"""
        return -(-rows // 16) * 16 * c * 2

    def _conv_bn_planes(self, plan, scope, xplanes, hw, cin, kh, kw, cout, stride, padding, out, training, side=False,
                        tag=""):
        """like _conv_bn, for an input in split format (pre-normalised fp16 hi/lo planes).  side: conv + finalize on the
        plan's side lane (the projection shortcut beside conv1..conv3 of its unit); tag "2": second pipeline stage.
        Every (stage, lane) pair has a statistics buffer and a tail workspace (tickets) of its own: they may run at the
        same time"""
        st = self.session.store
        P = lambda n: ops.LazyPtr(lambda n=n: st.p(n))  # noqa: E731
        d = ops.conv_desc(self.N, hw[0], hw[1], cin, cout, kh, kw, stride, padding, ldx=cin, ldy=up4(cout),
                          ldw=up4(cout))
        rows = ops.conv2d_fwd_split3p_stats_rows(d) if self._terms == 3 else ops.conv2d_fwd_split3_stats_rows(d)
        lane = ("_side" if side else "") + tag
        if lane:
            self._lane_stats_need = getattr(self, "_lane_stats_need", {})
            self._lane_stats_need[lane] = max(self._lane_stats_need.get(lane, 0), rows * 2 * up4(cout))
            stats = ops.LazyPtr(lambda lane=lane: self.lane_stats[lane])
        else:
            self._stats_need = max(self._stats_need, rows * 2 * up4(cout))
            stats = ops.LazyPtr(lambda: self.stats)
        if scope not in self._sp3:
            off = self._sp3_bytes
            self._sp3[scope] = off
            self._sp3_bytes += -(-ops.conv2d_split3_weight_bytes(d) // 256) * 256
            ops.conv2d_split3_prepare(self.plan_prepare, d, P(scope + "/weights"),
                                      ops.LazyPtr(lambda off=off: self.wsplit[off:]))
        off = self._sp3[scope]
        ws_attr = "_tail_ws" + lane
        if getattr(self, ws_attr, None) is None:
            # partial sums + tile tickets of the trunk kernel's tail split; used by nothing else, zero at start
            setattr(self, ws_attr, torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8,
                                               device=self.session.device))
        # (recording the staged trunk WITHOUT the tail split — the other stages fill the partial last rounds anyway —
        #  was measured: 7.21 -> 7.13 ms/step pipelined, but 8.96 -> 9.22 ms on one stream and 249 -> 223 TFLOP/s for
        #  the kernel with the chip to itself; the plan is shared by both entry points, so the split stays)
        tail_ws = getattr(self, ws_attr)
        ops.conv2d_fwd_split3p(plan, d, xplanes, self._lo_off(self.N * hw[0] * hw[1], cin),
                               ops.LazyPtr(lambda off=off: self.wsplit[off:]), out, stats if training else None,
                               tail_ws=tail_ws, terms=self._terms, side=side)
        if not hasattr(self, "_aff_cache"):
            self._aff_cache = {}
        if scope not in self._aff_cache:
            self._aff_cache[scope] = self._new_affine(up4(cout))
        sc, sh = self._aff_cache[scope]
        b = scope + "/BatchNorm/"
        ops.bn_finalize(plan, stats if training else None, rows if training else 0, cout, up4(cout),
                        self.N * d.OH * d.OW if training else 0, P(b + "gamma"), P(b + "beta"),
                        P(b + "moving_mean"), P(b + "moving_variance"), sc, sh, BN_DECAY, BN_EPS, training, side=side)
        return d.OH, d.OW, sc, sh

    def _conv3_two_pass(self, plan, scope, xplanes, hw, cin, cout, sc_planes, out_planes, training, tag="", proj=None):
        """conv3 of a unit without its raw output (include/acimg.h, acimg_conv2d_fwd_split3p_stats / _tail / _tail_proj):
        [statistics pass ->] bn_finalize -> the conv again with relu(BN + shortcut) + split in the epilogue.  In inference
        mode the affine comes from the moving statistics and the first pass is not needed.  proj = (raw fp32 output of the
        unit's shortcut conv, its scale, its shift) for a unit with a projection shortcut, else the identity shortcut is
        read from `sc_planes`."""
        st = self.session.store
        P = lambda n: ops.LazyPtr(lambda n=n: st.p(n))  # noqa: E731
        d = ops.conv_desc(self.N, hw[0], hw[1], cin, cout, 1, 1, 1, "SAME", ldx=cin, ldy=cout, ldw=cout)
        rows = -(-self.N * hw[0] * hw[1] // 128)
        if tag:
            self._lane_stats_need = getattr(self, "_lane_stats_need", {})
            self._lane_stats_need[tag] = max(self._lane_stats_need.get(tag, 0), rows * 2 * cout)
            stats = ops.LazyPtr(lambda lane=tag: self.lane_stats[lane])
        else:
            self._stats_need = max(self._stats_need, rows * 2 * cout)
            stats = ops.LazyPtr(lambda: self.stats)
        if scope not in self._sp3:
            off = self._sp3_bytes
            self._sp3[scope] = off
            self._sp3_bytes += -(-ops.conv2d_split3_weight_bytes(d) // 256) * 256
            ops.conv2d_split3_prepare(self.plan_prepare, d, P(scope + "/weights"),
                                      ops.LazyPtr(lambda off=off: self.wsplit[off:]))
        off = self._sp3[scope]
        wsplit = ops.LazyPtr(lambda off=off: self.wsplit[off:])
        ws_attr = "_tail_ws" + tag
        if getattr(self, ws_attr, None) is None:
            setattr(self, ws_attr, torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8,
                                               device=self.session.device))
        tail_ws = getattr(self, ws_attr)
        lo = self._lo_off(self.N * hw[0] * hw[1], cin)
        lo_out = self._lo_off(self.N * hw[0] * hw[1], cout)
        if not hasattr(self, "_aff_cache"):
            self._aff_cache = {}
        if scope not in self._aff_cache:
            self._aff_cache[scope] = self._new_affine(up4(cout))
        sc, sh = self._aff_cache[scope]
        b = scope + "/BatchNorm/"
        M = self.N * hw[0] * hw[1]
        if training and self.gram and ops.gram_stats_workspace(M, cin) > 0:
            # statistics from the conv's input (one workspace per pipeline stage: the stages run side by side)
            need = ops.gram_stats_workspace(M, cin)
            gw_attr = "_gram_ws" + tag
            if getattr(self, gw_attr, None) is None or getattr(self, gw_attr).numel() < need:
                setattr(self, gw_attr, torch.zeros(need, dtype=torch.uint8, device=self.session.device))
            gws = getattr(self, gw_attr)     # (a unit that needed a smaller one keeps the buffer it was recorded with)
            ops.gram_stats(plan, xplanes, lo, M, cin, P(scope + "/weights"), cout, cout, P(b + "gamma"), P(b + "beta"),
                           P(b + "moving_mean"), P(b + "moving_variance"), sc, sh, gws, BN_DECAY, BN_EPS)
        else:
            if training:
                ops.conv2d_fwd_split3p_stats(plan, d, xplanes, lo, wsplit, stats, tail_ws=tail_ws)
            ops.bn_finalize(plan, stats if training else None, rows if training else 0, cout, up4(cout),
                            M if training else 0, P(b + "gamma"), P(b + "beta"),
                            P(b + "moving_mean"), P(b + "moving_variance"), sc, sh, BN_DECAY, BN_EPS, training)
        if proj is not None:
            ops.conv2d_fwd_split3p_tail_proj(plan, d, xplanes, lo, wsplit, sc, sh, proj[0], proj[1], proj[2], out_planes,
                                             lo_out, tail_ws=tail_ws)
        else:
            ops.conv2d_fwd_split3p_tail(plan, d, xplanes, lo, wsplit, sc, sh, sc_planes, lo_out, out_planes, lo_out,
                                        tail_ws=tail_ws)

    def _two_pass_ok(self, hw, cin, cout):
        """measured per shape at batch 32 (profiles/r03/op_report_two_pass_all_r03u.txt): the statistics pass costs 85-95 %
        of the conv it repeats once K >= 256 (the K loop, not the output, is its time), so the second K loop only pays
        for the short-K / large-image units: 56x75 64->256 113 -> 84 us, 128->512 247 -> 192 us per unit, against
        28x38 256->1024 130 -> 144 and 14x19 512->2048 94 -> 121"""
        M = self.N * hw[0] * hw[1]
        if not (self.two_pass and self._terms == 3 and cin <= self.two_pass_max_cin and cout % 128 == 0 and
                M * cout * 4 < 2 ** 31):
            return False
        # the library's own tile choice for this shape under the current configuration (a forced experiment tile, or
        # fewer than 200 tiles, takes the unit back to conv + bn_add_relu_split)
        d = ops.conv_desc(self.N, hw[0], hw[1], cin, cout, 1, 1, 1, "SAME", ldx=cin, ldy=cout, ldw=cout)
        return tuple(ops.conv2d_fwd_split3_tiling(d)[:2]) == (128, 128)

    def _record_forward_split(self, plan, training):
        """f16x3 trunk: activations between convs live in split format.  Per bottleneck unit:
             conv1(X) -> raw1 -> [BN+ReLU+split] -> conv2 -> raw2 -> [BN+ReLU+split] -> conv3 -> raw3
             out = relu(BN(raw3) + shortcut) written straight in split format (next unit's X)."""
        N = self.N
        H, W = self.height, self.width
        plan.add_hook(self._refresh_split_weights)
        ops.pad_image(plan, self.images, self.stem_frame, N, H, W, 3, 4, H + 6, W + 6, 3, 3)
        oh, ow, sc, sh = self._stem_bn(plan, training)
        ph, pw = self.pool_hw
        _, pt = ops.same_out_pad(oh, 3, 2)
        _, pl = ops.same_out_pad(ow, 3, 2)
        cur, nxt = self.planes_a, self.planes_b
        ops.bn_relu_maxpool_split(plan, self.raw0, sc, sh, cur, self._lo_off(N * ph * pw, 64), N, oh, ow, 64, ph, pw,
                                  pt, pl)
        h, w = ph, pw
        units = list(self._units())
        two = self.stages == 2
        for i, (scope, din, d, db, s) in enumerate(units):
            last = i == len(units) - 1
            st2 = two and i >= self.STAGE_CUT
            tag = "2" if st2 else ""
            r1, r2, r3, asc = ((self.arena_r12, self.arena_r22, self.arena_r32, self.arena_sc2) if st2 else
                               (self.arena_r1, self.arena_r2, self.arena_r3, self.arena_sc))
            p1, p2 = (self.planes_12, self.planes_22) if st2 else (self.planes_1, self.planes_2)
            if two and i == self.STAGE_CUT:
                # stage 2 starts here: it reads the boundary tensor and ping-pongs inside its own arena
                if training:
                    self.stage_calls = len(plan.calls)
                cur, nxt = self.planes_x, self.planes_a2
            beside = din != d and self.side_lane
            if beside:
                # the projection shortcut reads the same planes as conv1 and meets the main branch only in the unit's
                # last pass: it runs on the plan's side lane beside conv1 .. conv3 (their tails leave slots)
                plan.fork()
                _, _, ssc, tsc = self._conv_bn_planes(plan, scope + "/shortcut", cur, (h, w), din, 1, 1, d, s, "SAME",
                                                      asc, training, side=True, tag=tag)
            oh1, ow1, s1, t1 = self._conv_bn_planes(plan, scope + "/conv1", cur, (h, w), din, 1, 1, db, 1, "SAME",
                                                    r1, training, tag=tag)
            ops.bn_relu_split(plan, r1, s1, t1, 1, p1, self._lo_off(N * h * w, db), N * h * w, db)
            oh2, ow2, s2, t2 = self._conv_bn_planes(plan, scope + "/conv2", p1, (h, w), db, 3, 3, db, s,
                                                    "SAME" if s == 1 else 1, r2, training, tag=tag)
            ops.bn_relu_split(plan, r2, s2, t2, 1, p2, self._lo_off(N * oh2 * ow2, db), N * oh2 * ow2, db)
            if two and i == self.STAGE_CUT - 1:
                nxt_out = self.planes_x          # the last unit of stage 1 writes the boundary tensor
            else:
                nxt_out = nxt
            if s == 1 and not last and self._two_pass_ok((oh2, ow2), db, d):
                # conv3 twice: the raw [N, oh, ow, d] tensor and its BN pass never exist
                proj = None
                if din != d:
                    if beside:
                        plan.join()
                    else:
                        _, _, ssc, tsc = self._conv_bn_planes(plan, scope + "/shortcut", cur, (h, w), din, 1, 1, d, s,
                                                              "SAME", asc, training, tag=tag)
                    proj = (asc, ssc, tsc)
                self._conv3_two_pass(plan, scope + "/conv3", p2, (oh2, ow2), db, d, cur, nxt_out, training, tag=tag,
                                     proj=proj)
                h, w = oh2, ow2
                if two and i == self.STAGE_CUT and training:
                    self.stage_read_calls = len(plan.calls)
                if two and i == self.STAGE_CUT:
                    cur, nxt = self.planes_a2, self.planes_b2
                else:
                    cur, nxt = nxt, cur
                continue
            oh3, ow3, s3, t3 = self._conv_bn_planes(plan, scope + "/conv3", p2, (oh2, ow2), db, 1, 1, d, 1,
                                                    "SAME", r3, training, tag=tag)
            out_planes = None if last else nxt_out
            out_lo = 0 if last else self._lo_off(N * oh3 * ow3, d)
            out32 = self.xfinal if last else None
            if din != d:
                if beside:
                    plan.join()
                else:
                    _, _, ssc, tsc = self._conv_bn_planes(plan, scope + "/shortcut", cur, (h, w), din, 1, 1, d, s,
                                                          "SAME", asc, training, tag=tag)
                ops.bn_add_relu_split(plan, r3, s3, t3, asc, ssc, tsc, None, 0, out_planes, out_lo,
                                      out32, N, oh3, ow3, d, oh3, ow3, 1)
            else:
                ops.bn_add_relu_split(plan, r3, s3, t3, None, None, None, cur, self._lo_off(N * h * w, din),
                                      out_planes, out_lo, out32, N, oh3, ow3, d, h, w, s)
            h, w = oh3, ow3
            if two and i == self.STAGE_CUT and training:
                self.stage_read_calls = len(plan.calls)      # stage 2 no longer reads the boundary tensor from here on
            if not last:
                if two and i == self.STAGE_CUT:
                    cur, nxt = self.planes_a2, self.planes_b2
                else:
                    cur, nxt = nxt, cur
        fh, fw = self.feat_hw
        if training:
            # everything recorded so far is the FROZEN trunk (no trainable variable is read): the trainer may run it
            # for the next batch beside the trained part of the current one; the last call writes `xfinal`
            self.frozen_calls = len(plan.calls)
        scm, tcm = self._conv_map_tap(plan, training)
        ops.bn_relu(plan, self.raw_cm, scm, tcm, self.output, N * fh * fw, 12, 12, 12)
        if self.stats is None or self.stats.numel() < self._stats_need:
            self.stats = self.session.zeros(self._stats_need)
        self.lane_stats = getattr(self, "lane_stats", {})
        for lane, need in getattr(self, "_lane_stats_need", {}).items():
            if lane not in self.lane_stats or self.lane_stats[lane].numel() < need:
                self.lane_stats[lane] = self.session.zeros(need)

    def _stem_bn(self, plan, training):
        """conv1 (7x7/2 after 3+3 explicit zero padding) + BN statistics as a row-run conv (include/acimg.h,
        acimg_conv2d_fwd_split3) on the split-MFMA kernel"""
        st = self.session.store
        P = lambda n: ops.LazyPtr(lambda n=n: st.p(n))  # noqa: E731
        scope = self.scope + "/conv1"
        d = self._stem_d
        rows = ops.conv2d_fwd_split3_stats_rows(d)
        self._stats_need = max(self._stats_need, rows * 2 * 64)
        stats = ops.LazyPtr(lambda: self.stats)
        if scope not in self._sp3:
            off = self._sp3_bytes
            self._sp3[scope] = off
            self._sp3_bytes += -(-ops.conv2d_split3_weight_bytes(d) // 256) * 256
            ops.conv2d_split3_prepare(self.plan_prepare, d, self.stem_w, ops.LazyPtr(lambda off=off: self.wsplit[off:]))
        off = self._sp3[scope]
        ops.conv2d_fwd_split3(plan, d, self.stem_frame, ops.LazyPtr(lambda off=off: self.wsplit[off:]), self.raw0,
                              None, None, 0, stats if training else None)
        if not hasattr(self, "_aff_cache"):
            self._aff_cache = {}
        if scope not in self._aff_cache:
            self._aff_cache[scope] = self._new_affine(64)
        sc, sh = self._aff_cache[scope]
        b = scope + "/BatchNorm/"
        ops.bn_finalize(plan, stats if training else None, rows if training else 0, 64, 64,
                        self.N * d.OH * d.OW if training else 0, P(b + "gamma"), P(b + "beta"),
                        P(b + "moving_mean"), P(b + "moving_variance"), sc, sh, BN_DECAY, BN_EPS, training)
        return d.OH, d.OW, sc, sh

    def _conv_map_tap(self, plan, training):
        """conv_map (3x4 VALID, 2048 -> 12) + BN statistics on the split-MFMA path: one GEMM over the INPUT pixels
        with 12 taps x 12 channels = 144 columns, then a 12-tap gather (ops.tapconv_*); the kernel trains, so its
        packed / split image is rebuilt in the plan (two tiny launches)"""
        st = self.session.store
        P = lambda n: ops.LazyPtr(lambda n=n: st.p(n))  # noqa: E731
        cm = self.scope + "/conv_map"
        d, d1 = self._cm_d, self._cm_d1
        fh, fw = self.feat_hw
        ops.tapconv_pack(plan, d, P(cm + "/weights"), self.cm_wt, 144)
        ops.conv2d_split3_prepare(plan, d1, self.cm_wt, self.cm_wsplit)
        ops.conv2d_fwd_split3(plan, d1, self.xfinal, self.cm_wsplit, self.cm_z)
        rows = ops.tapconv_stats_rows(d)
        # a statistics buffer of its own: conv_map is the first layer of the TRAINED part of the step, which the
        # two-lane pipeline (acimg/trainer.py) runs beside the frozen trunk of the next batch
        if getattr(self, "stats_cm", None) is None or self.stats_cm.numel() < rows * 2 * 12:
            self.stats_cm = self.session.zeros(rows * 2 * 12)
        stats = self.stats_cm
        ops.tapconv_gather(plan, d, self.cm_z, 144, self.raw_cm, stats if training else None)
        if not hasattr(self, "_aff_cache"):
            self._aff_cache = {}
        if cm not in self._aff_cache:
            self._aff_cache[cm] = self._new_affine(12)
        sc, sh = self._aff_cache[cm]
        b = cm + "/BatchNorm/"
        ops.bn_finalize(plan, stats if training else None, rows if training else 0, 12, 12,
                        self.N * fh * fw if training else 0, P(b + "gamma"), P(b + "beta"),
                        P(b + "moving_mean"), P(b + "moving_variance"), sc, sh, BN_DECAY, BN_EPS, training,
                        self.save_mean, self.save_invstd)
        return sc, sh

    def _record_forward(self, plan, training):
        if self._split:
            return self._record_forward_split(plan, training)
        N = self.N
        H, W = self.height, self.width
        plan.add_hook(self._refresh_split_weights)
        ops.pad_channels(plan, self.images, self.xpad, N * H * W, 3, 4)
        oh, ow, sc, sh = self._conv_bn(plan, self.scope + "/conv1", self.xpad, (H, W), 4, 7, 7, 64, 2, 3,
                                       self.raw0, None, training)
        ph, pw = self.pool_hw
        _, pt = ops.same_out_pad(oh, 3, 2)
        _, pl = ops.same_out_pad(ow, 3, 2)
        cur, nxt = self.arena_a, self.arena_b
        ops.bn_relu_maxpool(plan, self.raw0, sc, sh, cur, N, oh, ow, 64, ph, pw, pt, pl)
        h, w = ph, pw
        units = list(self._units())
        for i, (scope, din, d, db, s) in enumerate(units):
            last = i == len(units) - 1
            oh1, ow1, s1, t1 = self._conv_bn(plan, scope + "/conv1", cur, (h, w), din, 1, 1, db, 1, "SAME",
                                             self.arena_r1, None, training)
            oh2, ow2, s2, t2 = self._conv_bn(plan, scope + "/conv2", self.arena_r1, (h, w), db, 3, 3, db, s,
                                             "SAME" if s == 1 else 1, self.arena_r2, (s1, t1), training)
            oh3, ow3, s3, t3 = self._conv_bn(plan, scope + "/conv3", self.arena_r2, (oh2, ow2), db, 1, 1, d, 1,
                                             "SAME", self.arena_r3, (s2, t2), training)
            out = self.xfinal if last else nxt
            if din != d:
                _, _, ssc, tsc = self._conv_bn(plan, scope + "/shortcut", cur, (h, w), din, 1, 1, d, s, "SAME",
                                               self.arena_sc, None, training)
                ops.bn_add_relu(plan, self.arena_r3, s3, t3, self.arena_sc, ssc, tsc, out, N, oh3, ow3, d,
                                oh3, ow3, 1)
            else:
                ops.bn_add_relu(plan, self.arena_r3, s3, t3, cur, None, None, out, N, oh3, ow3, d, h, w, s)
            h, w = oh3, ow3
            if not last:
                cur, nxt = nxt, cur
        fh, fw = self.feat_hw
        _, _, scm, tcm = self._conv_bn(plan, self.scope + "/conv_map", self.xfinal, (h, w), 2048, 3, 4, 12, 1,
                                       "VALID", self.raw_cm, None, training, save=True)
        ops.bn_relu(plan, self.raw_cm, scm, tcm, self.output, N * fh * fw, 12, 12, 12)
        if self.stats is None or self.stats.numel() < self._stats_need:
            self.stats = self.session.zeros(self._stats_need)

    def record_backward(self, plan):
        """conv_map backward: g(output) -> BN/ReLU backward -> weight gradient (+ L2 term).  The trunk
        below conv_map receives no gradient (it is not in var_list, trainer/mfcctrainer.py:64)."""
        st = self.session.store
        cm = self.scope + "/conv_map"
        P = lambda n: ops.LazyPtr(lambda n=n: st.p(n))  # noqa: E731
        G = lambda n: ops.LazyPtr(lambda n=n: st.g(n))  # noqa: E731
        N = self.N
        fh, fw = self.feat_hw
        h, w = self.final_hw
        ops.bn_relu_bwd(plan, self.raw_cm, self.output, self.g_output, P(cm + "/BatchNorm/gamma"),
                        self.save_mean, self.save_invstd, self.g_raw_cm, G(cm + "/BatchNorm/gamma"),
                        G(cm + "/BatchNorm/beta"), N * fh * fw, 12)
        if self._split:
            # tap-GEMM form: dWt[2048][144] = xfinal^T . (tap-scattered g_raw_cm), then back to HWIO + the L2 term
            ops.tapconv_scatter(plan, self._cm_d, self.g_raw_cm, 12, self.cm_gz, 144)
            ops.conv2d_wgrad_split3(plan, self._cm_d1, self.xfinal, self.cm_gz, 144, self.cm_dwt, None)
            ops.tapconv_unpack(plan, self._cm_d, self.cm_dwt, 144, P(cm + "/weights"), WEIGHT_DECAY, G(cm + "/weights"))
            return
        d = ops.conv_desc(N, h, w, 2048, 12, 3, 4, 1, "VALID", ldx=2048, ldy=12, ldw=12)
        ops.conv2d_wgrad(plan, d, self.xfinal, self.g_raw_cm, 12, G(cm + "/weights"), None)
        ops.axpy(plan, WEIGHT_DECAY, P(cm + "/weights"), G(cm + "/weights"), 3 * 4 * 2048 * 12)

    def record_regularizer(self, plan, accum):
        """adds sum(w^2) over every conv kernel of the scope into accum[0] (slim l2_regularizer terms
        collected by tf.losses.get_total_loss, trainer/mfcctrainer.py:60).  The trunk is frozen
        (trainer/mfcctrainer.py:64), so its 23.5 M-element sum is computed when the parameters are (re)loaded and
        only added here; conv_map trains and is summed every step."""
        st = self.session.store
        if not hasattr(self, "_trunk_ss"):
            self._trunk_ss = self.session.zeros(4)
            self._trunk_ss_version = -1
            self._plan_reg = self.session.new_plan()
            ops.zero(self._plan_reg, self._trunk_ss)
            ops.sumsq(self._plan_reg, ops.LazyPtr(lambda: st.flat["trunkw"]), st._sizes["trunkw"], self._trunk_ss)
        plan.add_hook(self._refresh_trunk_sumsq)
        ops.axpy(plan, 1.0, self._trunk_ss, accum, 1)
        ops.sumsq(plan, ops.LazyPtr(lambda: st.p(self.scope + "/conv_map/weights")), 3 * 4 * 2048 * 12, accum)

    def _refresh_trunk_sumsq(self):
        st = self.session.store
        if self._trunk_ss_version != st.version:
            self._plan_reg.run()
            self._trunk_ss_version = st.version


def load_state_file(f):
    """{TF variable name: array}: a dict, a TensorFlow checkpoint prefix, an .npz, or a torch-saved dict (possibly
    under 'model')."""
    if isinstance(f, dict):
        return f.get("model", f)
    if os.path.exists(str(f) + ".index"):
        # a TensorFlow Saver-V2 bundle prefix (the reference's own checkpoints: trainer/mfcctrainer.py:214-247)
        from . import tfio
        return tfio.read_checkpoint(str(f))
    if str(f).endswith(".npz"):
        return dict(np.load(f))
    obj = torch.load(f, map_location="cpu", weights_only=False)
    return obj.get("model", obj) if isinstance(obj, dict) else obj
