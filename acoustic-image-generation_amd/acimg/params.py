"""Parameter storage in HBM.

Variables keep the reference's TensorFlow names and layouts at the boundary (state dicts are keyed
by TF variable name, SURVEY App. C: a TF-checkpoint importer can feed them directly) while living on
the device in *padded internal layouts* inside a few flat fp32 buffers:

  * every channel dimension is rounded up to a multiple of 4 so the GEMM kernels use 16-byte
    accesses; pad entries are zero and stay zero (their gradients are exactly zero);
  * all trainable variables share ONE flat buffer (plus flat grad / Adam-m / Adam-v twins): the
    optimiser is a single kernel launch and the data-parallel gradient exchange is a few large
    contiguous all-reduces (xGMI likes few, large messages);
  * frozen trunk conv kernels share another flat buffer (one pass computes the L2 regulariser).
"""
from collections import OrderedDict

import numpy as np
import torch


def up4(v):
    return (int(v) + 3) & ~3


class Var(object):
    """One TF variable: TF-side shape, internal (padded) shape, packing rules."""

    def __init__(self, name, tf_shape, kind, group):
        self.name = name
        self.tf_shape = tuple(int(s) for s in tf_shape)
        self.kind = kind          # conv | deconv | dense | vec
        self.group = group        # train | trunkw | state
        s = self.tf_shape
        if kind == "conv":        # HWIO
            self.shape = (s[0], s[1], up4(s[2]), up4(s[3]))
        elif kind == "deconv":    # [kh, kw, out, in]
            assert s[2] % 4 == 0
            self.shape = (s[0], s[1], s[2], up4(s[3]))
        elif kind == "dense":
            self.shape = (up4(s[0]), up4(s[1]))
        elif kind == "vec":
            self.shape = (up4(s[0]),)
        else:
            raise ValueError(kind)
        self.numel = int(np.prod(self.shape))
        self.offset = None

    def pack(self, arr):
        """TF-layout array -> padded internal array (float32, CPU)"""
        t = torch.as_tensor(np.asarray(arr), dtype=torch.float32)
        if tuple(t.shape) != self.tf_shape:
            raise ValueError("%s: expected shape %s, got %s" % (self.name, self.tf_shape, tuple(t.shape)))
        out = torch.zeros(self.shape, dtype=torch.float32)
        out[tuple(slice(0, n) for n in self.tf_shape)] = t
        return out

    def unpack(self, t):
        return t[tuple(slice(0, n) for n in self.tf_shape)].contiguous()


class FusedHeads(object):
    """`mean` and `std` 12x16-VALID convs (models/unet_acresnet.py:73-76) stored as ONE GEMM weight
    [12*16*148, 300] (columns 0..149 = mean, 150..299 = std) so both heads are a single pass over
    the 145-channel feature map.  In auto-encoder mode (:64) only `mean` exists: [.., 152]."""

    def __init__(self, scope, cin, z, with_std, hw=(12, 16), names=("mean", "std")):
        """hw: the VALID kernel size = the whole feature map; names: TF layer names of the two heads
        (('mean', 'variance') with a 14x18 / 6x16 map in models/unet_architecture.py:62-65, unet_sound.py:65-68)"""
        self.scope = scope
        self.hh, self.hw = hw
        self.n0, self.n1 = names
        self.cin, self.cp, self.z, self.with_std = cin, up4(cin), z, with_std
        self.ncols = 2 * z if with_std else up4(z)
        self.kernel = Var(scope + "/heads/kernel", (self.hh * self.hw * self.cp, self.ncols), "dense", "train")
        self.bias = Var(scope + "/heads/bias", (self.ncols,), "vec", "train")
        self.tf_names = ["%s/%s/kernel" % (scope, self.n0), "%s/%s/bias" % (scope, self.n0)]
        if with_std:
            self.tf_names += ["%s/%s/kernel" % (scope, self.n1), "%s/%s/bias" % (scope, self.n1)]

    def pack(self, tf):
        f32 = lambda n: torch.as_tensor(np.asarray(tf[n]), dtype=torch.float32)  # noqa: E731
        k = torch.zeros(self.hh, self.hw, self.cp, self.ncols)
        b = torch.zeros(self.ncols)
        k[:, :, :self.cin, :self.z] = f32("%s/%s/kernel" % (self.scope, self.n0))
        b[:self.z] = f32("%s/%s/bias" % (self.scope, self.n0))
        if self.with_std:
            k[:, :, :self.cin, self.z:2 * self.z] = f32("%s/%s/kernel" % (self.scope, self.n1))
            b[self.z:2 * self.z] = f32("%s/%s/bias" % (self.scope, self.n1))
        return k.reshape(self.hh * self.hw * self.cp, self.ncols), b

    def unpack(self, k, b):
        k = k.reshape(self.hh, self.hw, self.cp, self.ncols)
        out = OrderedDict()
        out["%s/%s/kernel" % (self.scope, self.n0)] = k[:, :, :self.cin, :self.z].contiguous()
        out["%s/%s/bias" % (self.scope, self.n0)] = b[:self.z].contiguous()
        if self.with_std:
            out["%s/%s/kernel" % (self.scope, self.n1)] = k[:, :, :self.cin, self.z:2 * self.z].contiguous()
            out["%s/%s/bias" % (self.scope, self.n1)] = b[self.z:2 * self.z].contiguous()
        return out


class ParamStore(object):
    """Flat device buffers + name -> view map."""

    ALIGN = 64  # floats (256 B)

    def __init__(self, device):
        self.device = device
        self.vars = OrderedDict()       # internal name -> Var
        self.fused = []                 # FusedHeads objects
        self.flat = {}                  # group -> tensor
        self.grad = None
        self.adam_m = None
        self.adam_v = None
        self._sizes = {"train": 0, "trunkw": 0, "state": 0}
        self._final = False
        self.version = 0                # bumped whenever variables are (re)loaded from the host

    # ---- registration ---------------------------------------------------------------------------
    def add(self, var):
        assert not self._final and var.name not in self.vars
        var.offset = self._sizes[var.group]
        self._sizes[var.group] += -(-var.numel // self.ALIGN) * self.ALIGN
        self.vars[var.name] = var
        return var

    def add_fused(self, heads):
        self.fused.append(heads)
        self.add(heads.kernel)
        self.add(heads.bias)

    def finalize(self):
        for g, n in self._sizes.items():
            self.flat[g] = torch.zeros(max(n, self.ALIGN), dtype=torch.float32, device=self.device)
        n = self.flat["train"].numel()
        self.grad = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._final = True

    # ---- views ------------------------------------------------------------------------------------
    def _view(self, buf, var):
        return buf[var.offset:var.offset + var.numel].view(var.shape)

    def p(self, name):
        v = self.vars[name]
        return self._view(self.flat[v.group], v)

    def g(self, name):
        v = self.vars[name]
        assert v.group == "train"
        return self._view(self.grad, v)

    def slot(self, name, which):
        v = self.vars[name]
        return self._view(self.adam_m if which == "m" else self.adam_v, v)

    def train_numel(self):
        return self.flat["train"].numel()

    def train_ranges(self):
        """[(name, offset, numel)] of the trainable flat buffer, in registration order"""
        return [(v.name, v.offset, v.numel) for v in self.vars.values() if v.group == "train"]

    # ---- TF-named state I/O --------------------------------------------------------------------------
    def tf_names(self):
        names = []
        fused_internal = set()
        for h in self.fused:
            names += h.tf_names
            fused_internal.update((h.kernel.name, h.bias.name))
        names += [n for n in self.vars if n not in fused_internal]
        return names

    def load_state(self, tf_state, strict=True, only=None):
        """tf_state: {TF variable name: array in TF layout}.  `only(name)->bool` filters names."""
        self.version += 1
        loaded = []
        fused_internal = set()
        for h in self.fused:
            fused_internal.update((h.kernel.name, h.bias.name))
            have = [n in tf_state for n in h.tf_names]
            if all(have) and (only is None or all(only(n) for n in h.tf_names)):
                k, b = h.pack(tf_state)
                self.p(h.kernel.name).copy_(k.to(self.device))
                self.p(h.bias.name).copy_(b.to(self.device))
                loaded += h.tf_names
            elif strict and only is None:
                raise KeyError("missing %s" % [n for n, ok in zip(h.tf_names, have) if not ok])
        for name, v in self.vars.items():
            if name in fused_internal or (only is not None and not only(name)):
                continue
            if name not in tf_state:
                if strict and only is None:
                    raise KeyError("missing variable %s" % name)
                continue
            self.p(name).copy_(v.pack(tf_state[name]).to(self.device))
            loaded.append(name)
        return loaded

    def load_slots(self, m_state, v_state):
        """Adam slot variables from TF-named dicts (checkpoint resume / parity tests)"""
        for which, state, buf in (("m", m_state, self.adam_m), ("v", v_state, self.adam_v)):
            fused_internal = set()
            for h in self.fused:
                fused_internal.update((h.kernel.name, h.bias.name))
                if all(n in state for n in h.tf_names):
                    k, b = h.pack(state)
                    self._view(buf, h.kernel).copy_(k.to(self.device))
                    self._view(buf, h.bias).copy_(b.to(self.device))
            for name, v in self.vars.items():
                if v.group == "train" and name not in fused_internal and name in state:
                    self._view(buf, v).copy_(v.pack(state[name]).to(self.device))

    def _export(self, getter):
        out = OrderedDict()
        fused_internal = set()
        for h in self.fused:
            fused_internal.update((h.kernel.name, h.bias.name))
            out.update(h.unpack(getter(h.kernel.name).detach().cpu(), getter(h.bias.name).detach().cpu()))
        for name, v in self.vars.items():
            if name in fused_internal:
                continue
            out[name] = v.unpack(getter(name).detach().cpu())
        return out

    def state_dict(self):
        """{TF variable name: CPU tensor in TF layout} for every variable"""
        return self._export(self.p)

    def grad_dict(self):
        names = set(n for n, _, _ in self.train_ranges())
        full = self._export(lambda n: self.g(n) if n in names else self.p(n))
        keep = set()
        for h in self.fused:
            keep.update(h.tf_names)
        keep.update(n for n in names)
        return OrderedDict((k, v) for k, v in full.items() if k in keep)

    def slot_dict(self, which):
        names = set(n for n, _, _ in self.train_ranges())
        full = self._export(lambda n: self.slot(n, which) if n in names else self.p(n))
        keep = set()
        for h in self.fused:
            keep.update(h.tf_names)
        keep.update(names)
        return OrderedDict((k, v) for k, v in full.items() if k in keep)
