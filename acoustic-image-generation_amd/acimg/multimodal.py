"""The latent associators `AssociatorVideoAc` / `AssociatorAudioAc` (models/multimodal.py:5-137), MI355X-native:
two towers of dense layers mapping one modality's VAE statistics (mean, std) to the acoustic-image latent's
(mean, std = softplus(.)).  Model protocol as in the reference: `scope`, `init_model`, `_build_model(mean, std)`
setting `mean`, `std`, `network`, `train_vars`.  Every layer is one GEMM launch (bias + ReLU in the epilogue); the
outputs are written as the halves of ONE [N, 300] buffer, which is exactly what `UNetAcZ` takes as its external
latent statistics; `record_backward(plan, g_ext)` consumes the gradient `UNetAcZ.record_backward` leaves there.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU
from .params import Var, up4
from .session import get_default_session
from .vision import load_state_file
from .unet_vae import AssociatorAudio          # noqa: F401  (models/multimodal.py:139-285: the conv associator)

Z = 150


class _Associator(object):
    SCOPE, DIN, WIDTHS = None, None, None

    def __init__(self, input_shape=None, num_frames=12, embedding=True):
        self.scope = self.SCOPE
        self.num_frames = num_frames
        self.height = input_shape
        self.session = None

    def _names(self):
        """[(tf layer name, cin, cout, tower, last)] in creation order (mean tower, then std tower)"""
        out, idx = [], 0
        for tower in range(2):
            cin = self.DIN
            for i, w in enumerate(self.WIDTHS):
                out.append(("dense" if idx == 0 else "dense_%d" % idx, cin, w, tower, i == len(self.WIDTHS) - 1))
                cin = w
                idx += 1
        return out

    def _register(self, store):
        for name, cin, cout, _, _ in reversed(self._names()):       # backward-completion order
            store.add(Var("%s/%s/kernel" % (self.scope, name), (cin, cout), "dense", "train"))
            store.add(Var("%s/%s/bias" % (self.scope, name), (cout,), "vec", "train"))

    def init_model(self, session, checkpoint_file):
        state = load_state_file(checkpoint_file)
        store = (session or self.session).store
        return store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def initialize(self, seed=1243, state=None):
        if state is None:
            g = torch.Generator().manual_seed(seed)
            state = OrderedDict()
            for name, cin, cout, _, _ in self._names():
                lim = np.sqrt(6.0 / (cin + cout))
                state["%s/%s/kernel" % (self.scope, name)] = (
                    (torch.rand(cin, cout, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()
                state["%s/%s/bias" % (self.scope, name)] = torch.zeros(cout)
        self.session.store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def _P(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.p(self.scope + "/" + name))

    def _G(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.g(self.scope + "/" + name))

    def _build_model(self, mean, std, session=None):
        """mean, std: the halves of ONE device buffer [N, 2*DIN] = [mean | std] (an encoder's fused head output)"""
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        src = mean._base if mean._base is not None else mean
        N = src.shape[0]
        assert src.shape == (N, 2 * self.DIN), "mean / std must be the halves of one [N, %d] buffer" % (2 * self.DIN)
        self.N, self.src = N, src
        z = sess.zeros
        self.ext = z(N, 2 * Z)              # [mean' | std']: UNetAcZ's external statistics
        self.raw_std = z(N, up4(Z))
        self.layers = []
        p = sess.new_plan()
        for tower in range(2):
            x, ldx, off = src, 2 * self.DIN, tower * self.DIN
            for name, cin, cout, tw, last in self._names():
                if tw != tower:
                    continue
                if last:
                    y, ldy = (self.ext, 2 * Z) if tower == 0 else (self.raw_std, up4(Z))
                else:
                    y, ldy = z(N, up4(cout)), up4(cout)
                d = ops.conv_desc(N, 1, 1, up4(cin), cout, 1, 1, 1, "VALID", ldx=ldx, ldy=ldy, ldw=up4(cout),
                                  act=ACT_NONE if last else ACT_RELU)
                ops.conv2d_fwd(p, d, ops.Ptr(x, off), self._P(name + "/kernel"), self._P(name + "/bias"), y)
                self.layers.append((name, d, x, off, ldx, y, ldy, tower, last))
                x, ldx, off = y, ldy, 0
        ops.softplus_fwd(p, self.raw_std, up4(Z), ops.Ptr(self.ext, Z), 2 * Z, N, Z)
        self.plan_fwd = p
        self.mean, self.std = self.ext[:, :Z], self.ext[:, Z:]
        self.network = OrderedDict(input=mean, input2=std)
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/")]

    def record_backward(self, plan, g_ext):
        """g_ext [N, 300]: d loss / d [mean' | std'] (e.g. UNetAcZ.g_ext)"""
        N = self.N
        z = self.session.zeros
        g_raw = z(N, up4(Z))
        ops.softplus_bwd(plan, self.raw_std, up4(Z), ops.Ptr(g_ext, Z), 2 * Z, g_raw, up4(Z), N, Z)
        for tower in range(2):
            gy, ldgy = (g_ext, 2 * Z) if tower == 0 else (g_raw, up4(Z))
            tl = [L for L in self.layers if L[7] == tower]
            for i in range(len(tl) - 1, -1, -1):
                name, d, x, off, ldx, y, ldy, _, last = tl[i]
                ops.conv2d_wgrad(plan, d, ops.Ptr(x, off), gy, ldgy, self._G(name + "/kernel"), self._G(name + "/bias"))
                if i > 0:
                    gx = z(N, ldx)
                    # the layer below has a ReLU: mask from its output (= this layer's input)
                    ops.conv2d_dgrad(plan, d, gy, ldgy, self._P(name + "/kernel"), gx, None, 0, ops.Ptr(x, off), ldx)
                    gy, ldgy = gx, ldx


class AssociatorVideoAc(_Associator):
    SCOPE, DIN, WIDTHS = "AssociatorVideoAc", 1024, [512, 512, 256, 256, 150, 150]


class AssociatorAudioAc(_Associator):
    SCOPE, DIN, WIDTHS = "AssociatorAudioAc", 256, [256, 256, 150]


class _JointMLP(object):
    """The joint-latent fusion MLPs (models/multimodal.py:287-465): tf.layers.dense acts on the LAST axis, so on
    [N, 12, 16, C] feature maps these are per-pixel MLPs = 1x1 convolutions.  concat(inputs) -> 3 x dense 512 (ReLU)
    -> one ReLU dense head per output modality.  The inputs must be the consecutive channel slices of ONE device
    buffer [..., Ctot] (the encoders write their feature maps there: the tf.concat costs nothing);
    `record_backward` takes the gradients w.r.t. the head outputs and leaves d loss / d concat(inputs) in
    `g_input`."""
    SCOPE, NIN, HEADS, HIDDEN = None, 3, (), 512

    def __init__(self, input_shape=None):
        self.scope = self.SCOPE
        self.session = None

    def _names(self):
        out, idx, cin = [], 0, self.cin
        for _ in range(3):
            out.append(("dense" if idx == 0 else "dense_%d" % idx, cin, self.HIDDEN, None))
            cin = self.HIDDEN
            idx += 1
        for attr, w in self.HEADS:
            out.append(("dense_%d" % idx, self.HIDDEN, w, attr))
            idx += 1
        return out

    def _register(self, store):
        for name, cin, cout, _ in reversed(self._names()):
            store.add(Var("%s/%s/kernel" % (self.scope, name), (cin, cout), "dense", "train"))
            store.add(Var("%s/%s/bias" % (self.scope, name), (cout,), "vec", "train"))

    init_model = _Associator.init_model
    _P = _Associator._P
    _G = _Associator._G

    def initialize(self, seed=1249, state=None):
        if state is None:
            g = torch.Generator().manual_seed(seed)
            state = OrderedDict()
            for name, cin, cout, _ in self._names():
                lim = np.sqrt(6.0 / (cin + cout))
                state["%s/%s/kernel" % (self.scope, name)] = (
                    (torch.rand(cin, cout, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()
                state["%s/%s/bias" % (self.scope, name)] = torch.zeros(cout)
        self.session.store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def _build_model(self, *inputs, **kw):
        sess = kw.get("session") or get_default_session()
        self.session = sess
        assert len(inputs) == self.NIN, "%s takes %d feature maps" % (self.scope, self.NIN)
        src = inputs[0]._base
        assert src is not None and all(t._base is src for t in inputs), "inputs must be slices of one buffer"
        ctot = src.shape[-1]
        off = 0
        for t in inputs:                                     # consecutive channel slices, in order
            assert t.shape[:-1] == src.shape[:-1] and t.stride() == src.stride()
            assert t.storage_offset() == src.storage_offset() + off, "inputs must be consecutive channel slices"
            off += t.shape[-1]
        # (the joint step's 133 + 512 + 128 = 773 channels sit in a 776-wide buffer: the three pad channels are zeros and
        #  meet zero rows of the padded first kernel)
        assert up4(off) == ctot, "the slices must cover the buffer up to its padding to a multiple of 4 channels"
        self.cin, self.src = off, src
        self._register(sess.store)
        rows = src.numel() // ctot
        self.rows = rows
        lead = tuple(src.shape[:-1])
        z = sess.zeros
        p = sess.new_plan()
        self.layers = []
        x, ldx = src, ctot
        self.heads = OrderedDict()
        for name, cin, cout, attr in self._names():
            if attr is not None:
                x, ldx = self.net, self.HIDDEN
            y = z(rows, up4(cout))
            d = ops.conv_desc(rows, 1, 1, up4(cin), cout, 1, 1, 1, "VALID", ldx=ldx, ldy=up4(cout), ldw=up4(cout),
                              act=ACT_RELU)
            ops.conv2d_fwd(p, d, x, self._P(name + "/kernel"), self._P(name + "/bias"), y)
            self.layers.append((name, d, x, ldx, y, attr))
            if attr is None:
                x, ldx = y, up4(cout)
                self.net = y
            else:
                self.heads[attr] = y
                setattr(self, attr, y.view(*(lead + (up4(cout),)))[..., :cout])
        self.plan_fwd = p
        self.network = OrderedDict(("input%s" % ("" if i == 0 else str(i + 1)), t) for i, t in enumerate(inputs))
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/")]

    def record_backward(self, plan, g_heads, need_input_grad=True):
        """g_heads: {head attribute: (gradient tensor / Ptr, row stride)} for the heads that carry a loss"""
        rows = self.rows
        z = self.session.zeros
        g_net, first = z(rows, self.HIDDEN), True
        for name, d, x, ldx, y, attr in self.layers:
            if attr is None or attr not in g_heads:
                continue
            g, ldg = g_heads[attr]
            gm = z(rows, d.ldy)
            ops.grad_slice(plan, g, int(ldg), gm, d.ldy, y, d.ldy, rows, d.K)         # ReLU of the head
            ops.conv2d_wgrad(plan, d, x, gm, d.ldy, self._G(name + "/kernel"), self._G(name + "/bias"))
            ops.conv2d_dgrad(plan, d, gm, d.ldy, self._P(name + "/kernel"), g_net, None if first else g_net,
                             0 if first else self.HIDDEN, self.net, self.HIDDEN)
            first = False
        assert not first, "no head gradient given"
        gy, ldgy = g_net, self.HIDDEN
        trunk = [L for L in self.layers if L[5] is None]
        self.g_input = None
        for i in (2, 1, 0):
            name, d, x, ldx, y, _ = trunk[i]
            ops.conv2d_wgrad(plan, d, x, gy, ldgy, self._G(name + "/kernel"), self._G(name + "/bias"))
            if i > 0:
                gx = z(rows, ldx)
                ops.conv2d_dgrad(plan, d, gy, ldgy, self._P(name + "/kernel"), gx, None, 0, x, ldx)
                gy, ldgy = gx, ldx
            elif need_input_grad:
                self.g_input = z(rows, ldx)
                ops.conv2d_dgrad(plan, d, gy, ldgy, self._P(name + "/kernel"), self.g_input)


class Jointmvae(_JointMLP):
    """models/multimodal.py:287-347: (acoustic 128, video 512, audio 128 channels at 12x16) -> 133 / 512 / 128"""
    SCOPE, NIN, HEADS = "Jointmvae", 3, (("outputac", 133), ("outputvideo", 512), ("outputaudio", 128))


class JointTwomvae(_JointMLP):
    """models/multimodal.py:349-404: (video, audio) -> acoustic features only"""
    SCOPE, NIN, HEADS = "JointTwomvae", 2, (("outputac", 133),)


class JointTwomvae2(_JointMLP):
    """models/multimodal.py:406-465: (video, audio) -> all three modalities"""
    SCOPE, NIN, HEADS = "JointTwomvae2", 2, (("outputac", 133), ("outputvideo", 512), ("outputaudio", 128))
