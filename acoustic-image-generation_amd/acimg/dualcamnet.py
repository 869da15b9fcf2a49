"""`DualCamHybridModel` (scope 'DualCamNet'): the classifier that trainer/trainer_reconstructed_class.py trains on
GENERATED acoustic images (models/dualcamnet.py:13-158), MI355X-native.

Model protocol as in the reference: `scope`, `init_model`, `_build_model(acoustic_images)` setting `output`
(per-frame logits) and `network`.  Graph (dualcamnet.py:82-106): temporal conv3d 12x1x1 + ReLU, conv 5x5 12->32 +
ReLU, max-pool 3x3/3, conv 5x5 32->128 + ReLU, reduce_sum over the map, FC 128->1000 + ReLU, FC 1000->classes.

MI355X mapping: the conv3d is a 12x1 conv over the [clips, 12, 36*48, 12] view of the frame batch (frames are
the H axis, TF SAME pads 5 / 6); every layer is one implicit-GEMM launch with bias + ReLU in its epilogue; the
backward kernels emit the pre-activation gradient of the layer below (ReLU masks from the saved activations,
the pool's argmax routing and the reduce_sum broadcast fused with those masks).
"""
from collections import OrderedDict

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU
from .params import Var, up4
from .session import get_default_session
from .unet_acresnet import Act
from .vision import load_state_file


class DualCamHybridModel(object):

    def __init__(self, input_shape=None, num_classes=10, num_frames=12, embedding=True):
        self.scope = 'DualCamNet'
        self.num_classes = num_classes
        self.num_frames = num_frames
        self.height, self.width, self.channels = input_shape
        self.embedding = embedding
        self.session = None

    # ---- variables (TF names: models/base.py:11-12,27-28,64-65) ---------------------------------------------
    def _register(self, store):
        s = self.scope
        K = self.num_classes
        # registration = backward-completion order
        store.add(Var(s + "/full3/weights", (1000, K), "dense", "train"))
        store.add(Var(s + "/full3/biases", (K,), "vec", "train"))
        store.add(Var(s + "/full1/weights", (128, 1000), "dense", "train"))
        store.add(Var(s + "/full1/biases", (1000,), "vec", "train"))
        store.add(Var(s + "/conv3/weights", (5, 5, 32, 128), "conv", "train"))
        store.add(Var(s + "/conv3/biases", (128,), "vec", "train"))
        store.add(Var(s + "/conv2/weights", (5, 5, 12, 32), "conv", "train"))
        store.add(Var(s + "/conv2/biases", (32,), "vec", "train"))
        # tf.nn.conv3d kernel [kD, kH, kW, in, out] = [12, 1, 1, 12, 12]: stored as the 12x1 HWIO kernel it is
        store.add(Var(s + "/conv1/weights", (12, 1, 12, 12), "conv", "train"))
        store.add(Var(s + "/conv1/biases", (12,), "vec", "train"))

    def _to_internal(self, state):
        k = self.scope + "/conv1/weights"
        if k in state and len(state[k].shape) == 5:
            state = dict(state)
            state[k] = torch.as_tensor(state[k]).reshape(12, 1, 12, 12)
        return state

    def state_dict_tf(self):
        """TF-shaped variables of this scope (conv1 back to its 5-D conv3d shape)"""
        sd = OrderedDict((k, v) for k, v in self.session.store.state_dict().items() if k.startswith(self.scope + "/"))
        sd[self.scope + "/conv1/weights"] = sd[self.scope + "/conv1/weights"].reshape(12, 1, 1, 12, 12)
        return sd

    def init_model(self, session, checkpoint_file):
        state = self._to_internal(load_state_file(checkpoint_file))
        store = (session or self.session).store
        return store.load_state(state, strict=False, only=lambda n: n.startswith(self.scope + "/"))

    def initialize(self, seed=1241, state=None):
        """truncated normal sigma=0.01 weights, zero biases (models/base.py:9-10)"""
        if state is None:
            g = torch.Generator().manual_seed(seed)
            state = OrderedDict()
            for name, v in self.session.store.vars.items():
                if not name.startswith(self.scope + "/"):
                    continue
                if name.endswith("biases"):
                    state[name] = torch.zeros(v.tf_shape)
                else:
                    state[name] = (torch.randn(*v.tf_shape, generator=g, dtype=torch.float64).clamp(-2, 2) * 0.01).float()
        self.session.store.load_state(self._to_internal(state), strict=False,
                                      only=lambda n: n.startswith(self.scope + "/"))

    def _P(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.p(self.scope + "/" + name))

    def _G(self, name):
        st = self.session.store
        return ops.LazyPtr(lambda: st.g(self.scope + "/" + name))

    # ---- graph ------------------------------------------------------------------------------------------------
    def _build_model(self, acoustic_images, session=None):
        """acoustic_images: device buffer [clips*12, 36, 48, 12] (trainer_reconstructed_class.py:44 reshapes the
        generator output to [-1, 12, 36, 48, 12]: consecutive frames form a clip)"""
        sess = session or get_default_session()
        self.session = sess
        self._register(sess.store)
        NF = acoustic_images.shape[0]
        F_, H, W, C = self.num_frames, self.height, self.width, self.channels
        assert NF % F_ == 0 and tuple(acoustic_images.shape[1:]) == (H, W, C) and C % 4 == 0
        self.NF, self.clips = NF, NF // F_
        z = sess.zeros
        K = self.num_classes
        kp = up4(K)
        self.x = acoustic_images
        self.relu1 = Act(z(NF, H, W, C), NF, H, W, C)
        self.relu2 = Act(z(NF, H, W, 32), NF, H, W, 32)
        ph, pw = H // 3, W // 3
        self.pool2 = Act(z(NF, ph, pw, 32), NF, ph, pw, 32)
        self.relu3 = Act(z(NF, ph, pw, 128), NF, ph, pw, 128)
        self.pool3 = z(NF, 128)
        self.relu4 = z(NF, 1000)
        self.logits = z(NF, kp)
        # conv3d 12x1x1 SAME as a 12x1 conv over [clips, 12, H*W, C]
        self.d1 = ops.conv_desc(self.clips, F_, H * W, C, C, F_, 1, 1, "SAME", ldx=C, ldy=C, ldw=C, act=ACT_RELU)
        self.d2 = ops.conv_desc(NF, H, W, C, 32, 5, 5, 1, "SAME", ldx=C, ldy=32, ldw=32, act=ACT_RELU)
        self.d3 = ops.conv_desc(NF, ph, pw, 32, 128, 5, 5, 1, "SAME", ldx=32, ldy=128, ldw=128, act=ACT_RELU)
        self.d4 = ops.conv_desc(NF, 1, 1, 128, 1000, 1, 1, 1, "VALID", ldx=128, ldy=1000, ldw=1000, act=ACT_RELU)
        self.d5 = ops.conv_desc(NF, 1, 1, 1000, K, 1, 1, 1, "VALID", ldx=1000, ldy=kp, ldw=kp, act=ACT_NONE)
        p = sess.new_plan()
        ops.conv2d_fwd(p, self.d1, self.x, self._P("conv1/weights"), self._P("conv1/biases"), self.relu1.t)
        ops.conv2d_fwd(p, self.d2, self.relu1.t, self._P("conv2/weights"), self._P("conv2/biases"), self.relu2.t)
        ops.maxpool_fwd(p, self.relu2.t, 32, self.pool2.t, 32, NF, H, W, 32, 3)
        ops.conv2d_fwd(p, self.d3, self.pool2.t, self._P("conv3/weights"), self._P("conv3/biases"), self.relu3.t)
        ops.spatial_sum(p, self.relu3.t, 128, self.pool3, NF, ph * pw, 128)
        ops.conv2d_fwd(p, self.d4, self.pool3, self._P("full1/weights"), self._P("full1/biases"), self.relu4)
        ops.conv2d_fwd(p, self.d5, self.relu4, self._P("full3/weights"), self._P("full3/biases"), self.logits)
        self.plan_fwd = p
        self.output = self.logits
        self.network = OrderedDict([("input", acoustic_images), ("is_training", None), ("keep_prob", None),
                                    (2, self.relu1.t), (4, self.relu2.t), (5, self.pool2.t), (7, self.relu3.t),
                                    (8, self.pool3), (10, self.relu4), (13, self.logits)])
        self.train_vars = [n for n in sess.store.tf_names() if n.startswith(self.scope + "/")]

    def record_backward(self, plan, g_logits):
        """g_logits [clips*12, up4(classes)]: d loss / d per-frame logits (acimg_clip_softmax_ce)"""
        NF = self.NF
        z = self.session.zeros
        H, W = self.height, self.width
        ph, pw = self.pool2.H, self.pool2.W
        kp = up4(self.num_classes)
        g4 = z(NF, 1000)
        ops.conv2d_wgrad(plan, self.d5, self.relu4, g_logits, kp, self._G("full3/weights"), self._G("full3/biases"))
        ops.conv2d_dgrad(plan, self.d5, g_logits, kp, self._P("full3/weights"), g4, None, 0, self.relu4, 1000)
        g_pool3 = z(NF, 128)
        ops.conv2d_wgrad(plan, self.d4, self.pool3, g4, 1000, self._G("full1/weights"), self._G("full1/biases"))
        ops.conv2d_dgrad(plan, self.d4, g4, 1000, self._P("full1/weights"), g_pool3)
        g3 = z(NF, ph, pw, 128)
        ops.spatial_sum_relu_bwd(plan, self.relu3.t, 128, g_pool3, g3, 128, NF, ph * pw, 128)
        g_pool2 = z(NF, ph, pw, 32)
        ops.conv2d_wgrad(plan, self.d3, self.pool2.t, g3, 128, self._G("conv3/weights"), self._G("conv3/biases"))
        ops.conv2d_dgrad(plan, self.d3, g3, 128, self._P("conv3/weights"), g_pool2)
        g2 = z(NF, H, W, 32)
        ops.maxpool_relu_bwd(plan, self.relu2.t, 32, g_pool2, 32, g2, 32, NF, H, W, 32, 3)
        g1 = z(NF, H, W, self.channels)
        ops.conv2d_wgrad(plan, self.d2, self.relu1.t, g2, 32, self._G("conv2/weights"), self._G("conv2/biases"))
        ops.conv2d_dgrad(plan, self.d2, g2, 32, self._P("conv2/weights"), g1, None, 0, self.relu1.t, self.channels)
        ops.conv2d_wgrad(plan, self.d1, self.x, g1, self.channels, self._G("conv1/weights"), self._G("conv1/biases"))
        self._grad_bufs = dict(g1=g1, g2=g2, g3=g3, g4=g4)
