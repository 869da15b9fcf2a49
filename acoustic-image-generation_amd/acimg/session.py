"""`Session`: what stands where the reference has a tf.Session — the device, its stream, the
parameter store and the kernel workspace shared by every model built in it."""
import torch

from . import _lib, ops
from .params import ParamStore


class Session(object):
    def __init__(self, device=None):
        _lib.load()  # fail loudly, now, if the HIP extension is missing
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("acimg needs an MI355X (no CPU fallback exists for the hot path)")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.store = ParamStore(self.device)
        self.ws = ops.Workspace(self.device)
        self._finalized = False

    def new_plan(self):
        return ops.Plan(self.device, eager=False, ws=self.ws)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def finalize(self):
        if not self._finalized:
            self.store.finalize()
            self._finalized = True


_default = None


def get_default_session():
    global _default
    if _default is None:
        _default = Session()
    return _default


def set_default_session(s):
    global _default
    _default = s
