"""Host-side op layer: builds *plans* (lists of C-ABI calls on device buffers) for the kernels in
libacimg.so.  PyTorch is only the allocator / stream provider here; every arithmetic op of the hot
path is a call through ``acimg._lib`` (no torch math, no CPU fallback).

A :class:`Plan` can run eagerly (tests, one-off calls) or be recorded once and replayed every
step with fixed buffers, which keeps Python overhead to one ctypes call per kernel launch.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvDesc  # noqa: F401


def up4(v):
    return (int(v) + 3) & ~3


class Ptr(object):
    """A device address = tensor base + offset (in floats); keeps the tensor alive."""

    __slots__ = ("t", "off")

    def __init__(self, t, off=0):
        self.t = t
        self.off = int(off)

    def addr(self):
        return self.t.data_ptr() + 4 * self.off


class LazyPtr(object):
    """Device address known only once the session's flat buffers exist (resolved at Plan.finalize)."""

    __slots__ = ("fn", "off")

    def __init__(self, fn, off=0):
        self.fn = fn
        self.off = int(off)

    def addr(self):
        return self.fn().data_ptr() + 4 * self.off


class Workspace(object):
    """Caller-owned scratch shared by all calls of a plan (the library allocates nothing)."""

    def __init__(self, device):
        self.device = device
        self.need = 256
        self.buf = None
        self.version = 0
        self._tickets = None
        self._side = None

    @property
    def side(self):
        """the workspace of the calls this workspace's plans put on their SIDE lane (a second stream): a child of its
        own, so that two sessions / trainers in one process - whose plans run on different main streams at the same
        time - never share a split-K slab through a per-device singleton"""
        if self._side is None:
            self._side = Workspace(self.device)
        return self._side

    @property
    def tickets(self):
        """ACIMG_TICKET_WORDS zeroed ints for the in-kernel split-K combine (include/acimg.h): owned by this
        workspace, i.e. by the one stream its plans run on; every launch leaves them at zero"""
        if self._tickets is None and torch.device(self.device).type == "cuda":
            self._tickets = torch.zeros(TICKET_WORDS, dtype=torch.int32, device=self.device)
        return self._tickets

    def require(self, nbytes):
        self.need = max(self.need, int(nbytes))

    def allocate(self):
        if self.buf is None or self.buf.numel() < self.need:
            self.buf = torch.empty(self.need, dtype=torch.uint8, device=self.device)
            self.version += 1   # plans resolved against the old buffer must re-resolve
        return self

    @property
    def ptr(self):
        return self.buf.data_ptr()

    @property
    def nbytes(self):
        return self.buf.numel()


TICKET_WORDS = 4096   # ACIMG_TICKET_WORDS


class _Tickets(object):
    """resolves to the workspace's ticket block (None without a GPU, or when `off`)"""

    def __init__(self, ws, off=False):
        self.ws, self.off = ws, off


class _WsPtr(object):
    def __init__(self, ws):
        self.ws = ws


class _WsBytes(object):
    def __init__(self, ws):
        self.ws = ws


def _resolve(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    if isinstance(a, (Ptr, LazyPtr)):
        return a.addr()
    if isinstance(a, _WsPtr):
        return a.ws.ptr
    if isinstance(a, _WsBytes):
        return a.ws.nbytes
    if isinstance(a, _Tickets):
        t = None if a.off else a.ws.tickets
        return None if t is None else t.data_ptr()
    if isinstance(a, _JobsArg):
        return a.value()
    return a


class PrepareJobs(object):
    """Kernels whose split images one acimg_conv2d_split3_prepare_multi launch rebuilds: (descriptor, fp32 kernel,
    image buffer, mode 0 forward / 1 data gradient).  The ctypes tables are built when the plan is finalised, so
    jobs may still be added after the launch has been recorded."""

    def __init__(self):
        self.jobs = []

    def add(self, d, w, out, mode):
        self.jobs.append((d, w, out, int(mode)))

    def build(self):
        n = len(self.jobs)
        assert n <= 16, "conv2d_split3_prepare_multi takes at most 16 jobs (%d)" % n
        self._d = (C.POINTER(ConvDesc) * n)(*[C.pointer(j[0]) for j in self.jobs])
        self._w = (C.c_void_p * n)(*[_resolve(j[1]) for j in self.jobs])
        self._o = (C.c_void_p * n)(*[_resolve(j[2]) for j in self.jobs])
        self._m = (C.c_int * n)(*[j[3] for j in self.jobs])
        # several plans may replay this launch (each resolves its own copy of the arguments): every generation of
        # the tables stays alive
        self._keep = getattr(self, "_keep", [])
        self._keep.append((self._d, self._w, self._o, self._m))
        return n


class _JobsArg(object):
    __slots__ = ("jobs", "which")

    def __init__(self, jobs, which):
        self.jobs, self.which = jobs, which

    def value(self):
        if self.which == "n":
            return self.jobs.build()
        return C.addressof(getattr(self.jobs, "_" + self.which))


def current_stream_handle(device=None):
    return torch.cuda.current_stream(device).cuda_stream


_SIDE = {}        # (device, main stream) -> side stream


def side_lane(device):
    """The SIDE LANE of the current stream of a device: a second HIP stream, one per main stream, so that two plans
    running on two streams do not couple through it (their side WORKSPACE belongs to the plan's own workspace:
    `Workspace.side`).  Plans put the weight gradient of a layer there (`Plan.add(..., side=True)` between
    `Plan.fork()` and `Plan.join()`), so that it runs beside the data gradient of the same layer on the main stream:
    the two read the same gradient, write disjoint buffers and each leaves part of the chip idle (slab tails, split-K
    hand-offs, 48-tile layers)."""
    skey = (str(device), torch.cuda.current_stream(device).cuda_stream)
    if skey not in _SIDE:
        _SIDE[skey] = torch.cuda.Stream(device=device)
    return _SIDE[skey]


def _runs_beside(a, b, work):
    """True if a small kernel on stream b completes while stream a is still busy (two streams that share a hardware queue
    serialise: the runtime maps streams onto a few queues in an order the program cannot see).  The load on `a` is the
    library's own timed spin (`acimg_spin`: one wave reading the shader clock for ~2 ms), the probe on `b` a 32-byte
    `acimg_zero`: no vendor kernel runs inside the product package."""
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    L = _L()
    torch.cuda.synchronize(work.device)
    with torch.cuda.stream(a):
        _lib.check(L.acimg_spin(4000000, work.data_ptr(), a.cuda_stream), "spin")      # ~2 ms at ~2 GHz
        ea.record(a)
    with torch.cuda.stream(b):
        _lib.check(L.acimg_zero(work.data_ptr() + 64, 32, b.cuda_stream), "zero")
        eb.record(b)
    while not eb.query() and not ea.query():
        pass
    beside = eb.query() and not ea.query()
    torch.cuda.synchronize(work.device)
    return beside


def concurrent_streams(device, base, want, pool=8, priority=0):
    """up to `want` new streams that really run beside `base` and beside each other (measured, a few ms each); fewer if
    the runtime has fewer free hardware queues — the caller then shares lanes, which costs time, never correctness.
    priority: HIP stream priority of the new streams (0 normal, -1 high): a scheduling hint only"""
    work = torch.zeros(64, device=device)
    chosen, cands = [base], [torch.cuda.Stream(device=device, priority=int(priority)) for _ in range(pool)]
    for c in cands:
        if len(chosen) > want:
            break
        if all(_runs_beside(x, c, work) and _runs_beside(c, x, work) for x in chosen):
            chosen.append(c)
    return chosen[1:]


def set_side_lane(device, main_stream, side_stream):
    """use `side_stream` as the side lane of plans run on `main_stream` (side_stream = main_stream: no second lane)"""
    _SIDE[(str(device), main_stream.cuda_stream)] = side_stream


class Plan(object):
    """Ordered list of C-ABI calls.  eager=True runs each call as it is added (everything on the current stream)."""

    def __init__(self, device=None, eager=False, ws=None):
        self.device = device
        self.eager = eager
        self.ws = ws if ws is not None else Workspace(device)
        self.calls = []      # (name, fn, raw args); args None => host hook
        self.side = set()    # indices of `calls` launched on the side lane's stream
        self._resolved = None
        self._ws_version = -1
        self._cuda = device is not None and torch.device(device).type == "cuda"

    @property
    def side_ws(self):
        """workspace of the calls added with side=True (the main workspace when the plan is eager / not on a GPU)"""
        return self.ws if (self.eager or not self._cuda) else self.ws.side

    def add_hook(self, pyfn, name="hook"):
        """a host callback run in order between kernel launches (e.g. fire a gradient all-reduce)"""
        if self.eager:
            pyfn()
        else:
            self.calls.append((name, pyfn, None))
            self._resolved = None

    def add(self, name, fn, *args, side=False):
        if self.eager:
            self.ws.allocate()
            rc = fn(*[_resolve(a) for a in args], current_stream_handle(self.device))
            _lib.check(rc, name)
        else:
            if side and self._cuda:
                self.side.add(len(self.calls))
            self.calls.append((name, fn, args))
            self._resolved = None

    def fork(self):
        """the side lane waits for everything issued on the main stream so far"""
        if self.eager or not self._cuda:
            return
        ev, dev = torch.cuda.Event(), self.device

        def _fork():
            ev.record(torch.cuda.current_stream(dev))
            side_lane(dev).wait_event(ev)
        self.add_hook(_fork, "fork")

    def make_join(self):
        """a callable that makes the main stream wait for everything issued on the side lane so far (for hooks that
        decide at run time whether they consume side-lane results)"""
        if self.eager or not self._cuda:
            return lambda: None
        ev, dev = torch.cuda.Event(), self.device

        def _join():
            ev.record(side_lane(dev))
            torch.cuda.current_stream(dev).wait_event(ev)
        return _join

    def join(self):
        """the main stream waits for everything issued on the side lane so far"""
        if self.eager or not self._cuda:
            return
        self.add_hook(self.make_join(), "join")

    def shared_calls(self):
        """indices of the calls (either lane) recorded between a fork and the next join: they may share the chip"""
        out, open_ = set(), False
        for i, (name, fn, args) in enumerate(self.calls):
            if args is None:
                open_ = True if name == "fork" else (False if name == "join" else open_)
            elif open_ or i in self.side:
                out.add(i)
        return out

    def slice(self, lo, hi):
        """calls [lo, hi) as a plan of their own (same workspace, same lanes)"""
        p = Plan(self.device, ws=self.ws)
        p.calls = self.calls[lo:hi]
        p.side = set(i - lo for i in self.side if lo <= i < hi)
        return p

    def extend(self, other):
        base = len(self.calls)
        self.calls.extend(other.calls)
        self.side.update(base + i for i in other.side)
        self._resolved = None

    def _versions(self):
        return (self.ws.version, self.side_ws.version)

    def finalize(self):
        self.ws.allocate()
        self.side_ws.allocate()
        self._ws_version = self._versions()
        self._resolved = [(name, fn, None if args is None else tuple(_resolve(a) for a in args))
                          for name, fn, args in self.calls]
        return self

    def _side_handle(self):
        return side_lane(self.device).cuda_stream if self.side else None

    def run(self, stream=None):
        if self._resolved is None or self._ws_version != self._versions():
            self.finalize()
        st = stream if stream is not None else current_stream_handle(self.device)
        side_st, side = self._side_handle(), self.side
        for i, (name, fn, args) in enumerate(self._resolved):
            if args is None:
                fn()
                continue
            rc = fn(*args, side_st if i in side else st)
            if rc:
                _lib.check(rc, name)

    def run_probed(self, indices, out, stream=None, offset=0):
        """run(), bracketing the calls whose index (+ offset: position of this slice in the plan the indices refer to)
        is in `indices` with events on the launch stream; appends (index, start_event, end_event) to `out` (bench.py's
        live per-kernel timing)."""
        if self._resolved is None or self._ws_version != self._versions():
            self.finalize()
        st = stream if stream is not None else current_stream_handle(self.device)
        ts = torch.cuda.current_stream(self.device)
        side_st, side = self._side_handle(), self.side
        for i, (name, fn, args) in enumerate(self._resolved):
            if args is None:
                fn()
                continue
            on_side = i in side
            if i + offset in indices:
                lane = side_lane(self.device) if on_side else ts
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(lane)
                rc = fn(*args, side_st if on_side else st)
                e1.record(lane)
                out.append((i + offset, e0, e1))
            else:
                rc = fn(*args, side_st if on_side else st)
            if rc:
                _lib.check(rc, name)

    def __len__(self):
        return len(self.calls)


# ------------------------------------------------------------------------------------------------
# geometry helpers (TensorFlow padding rules)
# ------------------------------------------------------------------------------------------------
def same_out_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_desc(N, H, W, C, K, R, S, stride=1, padding="SAME", ldx=None, ldy=None, ldw=None,
              act=ACT_NONE):
    """Descriptor for tf.layers.conv2d / slim conv2d.  padding: 'SAME', 'VALID' or an int p for
    explicit symmetric padding followed by VALID (slim conv2d_same)."""
    d = ConvDesc()
    if padding == "SAME":
        OH, pt = same_out_pad(H, R, stride)
        OW, pl = same_out_pad(W, S, stride)
    elif padding == "VALID":
        OH, OW, pt, pl = (H - R) // stride + 1, (W - S) // stride + 1, 0, 0
    else:
        p = int(padding)
        OH, OW, pt, pl = (H + 2 * p - R) // stride + 1, (W + 2 * p - S) // stride + 1, p, p
    d.N, d.H, d.W, d.C = N, H, W, C
    d.ldx = C if ldx is None else ldx
    d.K = K
    d.ldy = up4(K) if ldy is None else ldy
    d.OH, d.OW, d.R, d.S, d.stride, d.pad_t, d.pad_l = OH, OW, R, S, stride, pt, pl
    d.ldw = up4(K) if ldw is None else ldw
    d.act = act
    return d


def deconv_desc(N, H, W, C, K, R, S, stride, ldx=None, ldy=None, ldw=None):
    """Descriptor for tf.layers.conv2d_transpose VALID: out = in*stride + max(kernel - stride, 0)."""
    d = ConvDesc()
    d.N, d.H, d.W, d.C = N, H, W, C
    d.ldx = C if ldx is None else ldx
    d.K = K
    d.ldy = K if ldy is None else ldy
    d.OH, d.OW = H * stride + max(R - stride, 0), W * stride + max(S - stride, 0)
    d.R, d.S, d.stride, d.pad_t, d.pad_l = R, S, stride, 0, 0
    d.ldw = C if ldw is None else ldw
    d.act = ACT_NONE
    return d


# ------------------------------------------------------------------------------------------------
# op wrappers: each appends one (or a few) C-ABI calls to `plan`
# ------------------------------------------------------------------------------------------------
def _L():
    return _lib.load()


def conv2d_stats_rows(d):
    return _L().acimg_conv2d_stats_rows(C.byref(d))


def conv2d_fwd_tiling(d):
    """(BM, BN, splits) the forward kernel launch for `d` uses"""
    out = (C.c_int * 3)()
    _lib.check(_L().acimg_conv2d_fwd_tiling(C.byref(d), out), "conv2d_fwd_tiling")
    return out[0], out[1], out[2]


def _check_stats_rows(stats, rows, what):
    """the C side writes `rows` rows of 2 x ldw floats and cannot see the buffer: a tensor sized from ANOTHER descriptor (the
    row count depends on the kernel the shape takes, activation included) would be overrun silently"""
    if isinstance(stats, torch.Tensor) and stats.dim() >= 1 and stats.shape[0] < rows:
        raise ValueError("%s: statistics buffer has %d rows, this descriptor writes %d" % (what, stats.shape[0], rows))


def conv2d_fwd(plan, d, x, w, bias, y, in_scale=None, in_shift=None, in_relu=0, stats=None):
    L = _L()
    _check_stats_rows(stats, L.acimg_conv2d_stats_rows(C.byref(d)), "conv2d_fwd")
    plan.ws.require(L.acimg_conv2d_fwd_workspace(C.byref(d)))
    plan.add("conv2d_fwd", L.acimg_conv2d_fwd, C.byref(d), x, w, bias, y, in_scale, in_shift,
             int(in_relu), stats, _WsPtr(plan.ws), _WsBytes(plan.ws), _Tickets(plan.ws))


def conv2d_split3_weight_bytes(d):
    return _L().acimg_conv2d_split3_weight_bytes(C.byref(d))


def conv2d_fwd_split3_stats_rows(d):
    return _L().acimg_conv2d_fwd_split3_stats_rows(C.byref(d))


def conv2d_fwd_split3p_stats_rows(d):
    """statistics rows of conv2d_fwd_split3p(terms=3) for this shape under the CURRENT configuration"""
    return _L().acimg_conv2d_fwd_split3p_stats_rows(C.byref(d))


def conv2d_fwd_split3_tiling(d):
    """(BM, BN, persistent) of the pre-split forward conv launch for `d`"""
    out = (C.c_int * 3)()
    _lib.check(_L().acimg_conv2d_fwd_split3_tiling(C.byref(d), out), "conv2d_fwd_split3_tiling")
    return out[0], out[1], out[2]


def conv2d_split3_prepare(plan, d, w, wsplit, bf16=False):
    """bf16: the forward image of the bf16-operand convs (acimg_conv2d_bf16_prepare)"""
    if bf16:
        plan.add("conv2d_bf16_prepare", _L().acimg_conv2d_bf16_prepare, C.byref(d), w, wsplit)
    else:
        plan.add("conv2d_split3_prepare", _L().acimg_conv2d_split3_prepare, C.byref(d), w, wsplit)


def conv2d_split3_prepare_multi(plan, jobs):
    """one launch for all of `jobs` (a PrepareJobs; the first argument resolved builds the tables)"""
    plan.add("conv2d_split3_prepare_multi", _L().acimg_conv2d_split3_prepare_multi, _JobsArg(jobs, "n"),
             _JobsArg(jobs, "d"), _JobsArg(jobs, "w"), _JobsArg(jobs, "o"), _JobsArg(jobs, "m"))


def conv2d_fwd_split3(plan, d, x, wsplit, y, in_scale=None, in_shift=None, in_relu=0, stats=None, bias=None, bf16=False):
    """bf16: both operands rounded to bf16, one MFMA per product (acimg_conv2d_fwd_bf16; wsplit from the bf16 prepare)"""
    fn = _L().acimg_conv2d_fwd_bf16 if bf16 else _L().acimg_conv2d_fwd_split3
    _check_stats_rows(stats, _L().acimg_conv2d_fwd_split3_stats_rows(C.byref(d)), "conv2d_fwd_split3")
    plan.add("conv2d_fwd_bf16" if bf16 else "conv2d_fwd_split3", fn, C.byref(d), x, wsplit, bias, y, in_scale, in_shift,
             int(in_relu), stats)


def conv2d_split3_dgrad_weight_bytes(d):
    return _L().acimg_conv2d_split3_dgrad_weight_bytes(C.byref(d))


def conv2d_split3_prepare_dgrad(plan, d, w, wsplit):
    plan.add("conv2d_split3_prepare_dgrad", _L().acimg_conv2d_split3_prepare_dgrad, C.byref(d), w, wsplit)


def conv2d_dgrad_split3(plan, d, gy, ldgy, wsplit_t, dx, residual=None, ldres=0, mask=None, ldmask=0, lddx=0,
                        bf16=False):
    fn = _L().acimg_conv2d_dgrad_bf16 if bf16 else _L().acimg_conv2d_dgrad_split3
    plan.add("conv2d_dgrad_bf16" if bf16 else "conv2d_dgrad_split3", fn, C.byref(d), gy, int(ldgy), wsplit_t, dx,
             int(lddx), residual, int(ldres), mask, int(ldmask))


def conv2d_fwd_split3p_workspace(d):
    return int(_L().acimg_conv2d_fwd_split3p_workspace(C.byref(d)))


def conv2d_fwd_split3p(plan, d, x_planes, x_lo_off, wsplit, y, stats=None, tail_ws=None, terms=3, side=False):
    """tail_ws: a uint8 buffer DEDICATED to split3p calls ON ONE LANE (tickets in its first 4 KiB must start, and stay,
    zero; a call on the side lane brings its own); terms = 1: fp16 operand storage (acimg_conv2d_fwd_split1p: hi
    planes only)"""
    nbytes = 0 if tail_ws is None else (tail_ws.numel() * tail_ws.element_size() if hasattr(tail_ws, "numel") else 0)
    fn = _L().acimg_conv2d_fwd_split3p if terms == 3 else _L().acimg_conv2d_fwd_split1p
    plan.add("conv2d_fwd_split3p" if terms == 3 else "conv2d_fwd_split1p", fn, C.byref(d), x_planes, int(x_lo_off),
             wsplit, y, stats, tail_ws, int(nbytes), side=side)


def conv2d_fwd_split3p_stats(plan, d, x_planes, x_lo_off, wsplit, stats, tail_ws=None, side=False):
    """first pass of a two-pass conv (acimg_conv2d_fwd_split3p_stats): batch-norm partials, no output"""
    nbytes = 0 if tail_ws is None else (tail_ws.numel() * tail_ws.element_size() if hasattr(tail_ws, "numel") else 0)
    plan.add("conv2d_fwd_split3p_stats", _L().acimg_conv2d_fwd_split3p_stats, C.byref(d), x_planes, int(x_lo_off), wsplit,
             stats, tail_ws, int(nbytes), side=side)


def gram_stats_workspace(rows, Cn):
    return int(_L().acimg_gram_stats_workspace(int(rows), int(Cn)))


def gram_stats(plan, x_planes, x_lo_off, rows, Cn, w, ldw, K, gamma, beta, moving_mean, moving_var, scale, shift, ws,
               decay=0.997, eps=1e-5, side=False):
    """batch-norm statistics of y = x w (a 1x1 conv) from the Gram matrix of its INPUT, finalised into scale / shift and the
    moving averages (acimg_gram_stats: replaces conv2d_fwd_split3p_stats + bn_finalize in front of a fused tail).  ws: a
    uint8 buffer of gram_stats_workspace(rows, Cn) bytes, dedicated to the lane the call runs on."""
    plan.add("gram_stats", _L().acimg_gram_stats, x_planes, int(x_lo_off), int(rows), int(Cn), w, int(ldw), int(K), gamma,
             beta, moving_mean, moving_var, float(decay), float(eps), scale, shift, ws,
             int(ws.numel() * ws.element_size()), side=side)


def conv2d_fwd_split3p_tail(plan, d, x_planes, x_lo_off, wsplit, scale, shift, sc_planes, sc_lo_off, out_planes,
                            out_lo_off, tail_ws=None, side=False):
    """second pass (acimg_conv2d_fwd_split3p_tail): relu(conv * scale + shift + shortcut) straight into split planes"""
    nbytes = 0 if tail_ws is None else (tail_ws.numel() * tail_ws.element_size() if hasattr(tail_ws, "numel") else 0)
    plan.add("conv2d_fwd_split3p_tail", _L().acimg_conv2d_fwd_split3p_tail, C.byref(d), x_planes, int(x_lo_off), wsplit,
             scale, shift, sc_planes, int(sc_lo_off), out_planes, int(out_lo_off), tail_ws, int(nbytes), side=side)


def conv2d_fwd_split3p_tail_proj(plan, d, x_planes, x_lo_off, wsplit, scale, shift, sc32, sc_scale, sc_shift, out_planes,
                                 out_lo_off, tail_ws=None, side=False):
    """second pass of a unit with a projection shortcut (acimg_conv2d_fwd_split3p_tail_proj)"""
    nbytes = 0 if tail_ws is None else (tail_ws.numel() * tail_ws.element_size() if hasattr(tail_ws, "numel") else 0)
    plan.add("conv2d_fwd_split3p_tail_proj", _L().acimg_conv2d_fwd_split3p_tail_proj, C.byref(d), x_planes, int(x_lo_off),
             wsplit, scale, shift, sc32, sc_scale, sc_shift, out_planes, int(out_lo_off), tail_ws, int(nbytes), side=side)


def bn_relu_split(plan, x, scale, shift, relu, out, lo_off, rows, Cn):
    plan.add("bn_relu_split", _L().acimg_bn_relu_split, x, scale, shift, int(relu), out, int(lo_off), int(rows), Cn)


def bn_add_relu_split(plan, a, sa, ta, b32, sb, tb, b_planes, b_lo_off, out_planes, out_lo_off, out32, N, OH, OW,
                      Cn, BH, BW, bstride):
    plan.add("bn_add_relu_split", _L().acimg_bn_add_relu_split, a, sa, ta, b32, sb, tb, b_planes, int(b_lo_off),
             out_planes, int(out_lo_off), out32, N, OH, OW, Cn, BH, BW, bstride)


def bn_relu_maxpool_split(plan, x, scale, shift, out, lo_off, N, H, W, Cn, OH, OW, pad_t, pad_l):
    plan.add("bn_relu_maxpool_split", _L().acimg_bn_relu_maxpool_split, x, scale, shift, out, int(lo_off), N, H, W,
             Cn, OH, OW, pad_t, pad_l)


def conv2d_dgrad(plan, d, gy, ldgy, w, dx, residual=None, ldres=0, mask=None, ldmask=0, lddx=0):
    L = _L()
    plan.ws.require(L.acimg_conv2d_dgrad_workspace(C.byref(d)))
    plan.add("conv2d_dgrad", L.acimg_conv2d_dgrad, C.byref(d), gy, int(ldgy), w, dx, int(lddx), residual,
             int(ldres), mask, int(ldmask), _WsPtr(plan.ws), _WsBytes(plan.ws), _Tickets(plan.ws))


def conv2d_wgrad(plan, d, x, gy, ldgy, dw, db=None, side=False):
    """side: launch on the plan's side lane (Plan.fork / Plan.join around it and the work it runs beside)"""
    L = _L()
    ws = plan.side_ws if side else plan.ws
    ws.require(L.acimg_conv2d_wgrad_workspace(C.byref(d)))
    plan.add("conv2d_wgrad", L.acimg_conv2d_wgrad, C.byref(d), x, gy, int(ldgy), dw, db, _WsPtr(ws), _WsBytes(ws),
             side=side)


def conv2d_wgrad_split3(plan, d, x, gy, ldgy, dw, db=None, bf16=False, side=False):
    L = _L()
    ws = plan.side_ws if side else plan.ws
    ws.require(L.acimg_conv2d_wgrad_workspace(C.byref(d)))
    plan.add("conv2d_wgrad_bf16" if bf16 else "conv2d_wgrad_split3",
             L.acimg_conv2d_wgrad_bf16 if bf16 else L.acimg_conv2d_wgrad_split3, C.byref(d), x, gy, int(ldgy), dw, db,
             _WsPtr(ws), _WsBytes(ws), side=side)


def conv2d_affine_input_ok(d, precision):
    """precision 0 / 1 / 2 = fp32-class / split3 / bf16 entries: do the forward AND the weight gradient of this layer apply a
    producer's batch-norm affine + ReLU while they stage their input (the halo kernels)?"""
    return bool(_L().acimg_conv2d_affine_input_ok(C.byref(d), int(precision)))


def conv2d_wgrad_affine(plan, d, precision, x, in_scale, in_shift, gy, ldgy, dw, db=None, in_relu=True):
    """the weight gradient with x' = relu(x * in_scale + in_shift) formed on load (see include/acimg.h)"""
    L = _L()
    plan.ws.require(L.acimg_conv2d_wgrad_workspace(C.byref(d)))
    plan.add("conv2d_wgrad_affine", L.acimg_conv2d_wgrad_affine, C.byref(d), int(precision), x, in_scale, in_shift,
             int(bool(in_relu)), gy, int(ldgy), dw, db, _WsPtr(plan.ws), _WsBytes(plan.ws))


def deconv_fwd(plan, d, x, w, bias, y):
    L = _L()
    plan.ws.require(L.acimg_deconv_workspace(C.byref(d)))
    plan.add("deconv_fwd", L.acimg_deconv_fwd, C.byref(d), x, w, bias, y, _WsPtr(plan.ws),
             _WsBytes(plan.ws), _Tickets(plan.ws))


def deconv_dgrad(plan, d, gy, ldgy, w, dx, mask=None, ldmask=0):
    L = _L()
    plan.ws.require(L.acimg_deconv_workspace(C.byref(d)))
    plan.add("deconv_dgrad", L.acimg_deconv_dgrad, C.byref(d), gy, int(ldgy), w, dx, mask,
             int(ldmask), _WsPtr(plan.ws), _WsBytes(plan.ws), _Tickets(plan.ws))


def deconv_wgrad(plan, d, x, gy, ldgy, dw, db=None, side=False):
    L = _L()
    ws = plan.side_ws if side else plan.ws
    ws.require(L.acimg_deconv_workspace(C.byref(d)))
    plan.add("deconv_wgrad", L.acimg_deconv_wgrad, C.byref(d), x, gy, int(ldgy), dw, db, _WsPtr(ws), _WsBytes(ws),
             side=side)


def bn_finalize(plan, stats, rows, Cn, ldstats, count, gamma, beta, moving_mean, moving_var, scale,
                shift, decay=0.997, eps=1e-5, training=True, save_mean=None, save_invstd=None, side=False):
    plan.add("bn_finalize", _L().acimg_bn_finalize, stats, int(rows), int(Cn), int(ldstats),
             float(count), gamma, beta, moving_mean, moving_var, float(decay), float(eps),
             int(bool(training)), scale, shift, save_mean, save_invstd, side=side)


def bn_add_relu(plan, a, sa, ta, b, sb, tb, out, N, OH, OW, Cn, BH, BW, bstride):
    plan.add("bn_add_relu", _L().acimg_bn_add_relu, a, sa, ta, b, sb, tb, out, N, OH, OW, Cn, BH, BW,
             bstride)


def bn_relu_maxpool(plan, x, scale, shift, out, N, H, W, Cn, OH, OW, pad_t, pad_l):
    plan.add("bn_relu_maxpool", _L().acimg_bn_relu_maxpool, x, scale, shift, out, N, H, W, Cn, OH, OW,
             pad_t, pad_l)


def bn_relu(plan, x, scale, shift, y, rows, Cn, ldx, ldy):
    plan.add("bn_relu", _L().acimg_bn_relu, x, scale, shift, y, int(rows), Cn, ldx, ldy)


def bn_relu_bwd(plan, x, y, gy, gamma, save_mean, save_invstd, gx, dgamma, dbeta, rows, Cn):
    plan.add("bn_relu_bwd", _L().acimg_bn_relu_bwd, x, y, gy, gamma, save_mean, save_invstd, gx,
             dgamma, dbeta, int(rows), Cn)


def pad_channels(plan, x, y, pixels, Cn, Cp):
    plan.add("pad_channels", _L().acimg_pad_channels, x, y, int(pixels), Cn, Cp)


def pad_image(plan, x, y, N, H, W, Cn, Cp, Hp, Wp, pad_t, pad_l):
    plan.add("pad_image", _L().acimg_pad_image, x, y, int(N), int(H), int(W), int(Cn), int(Cp), int(Hp), int(Wp),
             int(pad_t), int(pad_l))


def tile_mfcc(plan, mfcc, out, N, HW, Cn):
    plan.add("tile_mfcc", _L().acimg_tile_mfcc, mfcc, out, N, HW, Cn)


def minmax_fwd(plan, x, ldx, out, ldo, mm, N, P, Cn):
    L = _L()
    plan.ws.require(L.acimg_minmax_workspace(int(N), int(P), int(Cn)))
    plan.add("minmax_fwd", L.acimg_minmax_fwd, x, ldx, out, ldo, mm, N, P, Cn, _WsPtr(plan.ws), _WsBytes(plan.ws))


def minmax_bwd(plan, x, ldx, go, ldgo, mm, gx, ldgx, N, P, Cn, accumulate=False, mask_relu=False):
    L = _L()
    plan.ws.require(L.acimg_minmax_workspace(int(N), int(P), int(Cn)))
    plan.add("minmax_bwd", L.acimg_minmax_bwd, x, ldx, go, ldgo, mm, gx, ldgx, N, P, Cn,
             int(accumulate), int(mask_relu), _WsPtr(plan.ws), _WsBytes(plan.ws))


def latent_fwd(plan, heads, eps, z, ldz, sigma, kl, N, Z):
    plan.add("latent_fwd", _L().acimg_latent_fwd, heads, eps, z, ldz, sigma, kl, N, Z)


def latent_bwd(plan, heads, eps, sigma, gz, ldgz, kl_weight, g_heads, N, Z):
    plan.add("latent_bwd", _L().acimg_latent_bwd, heads, eps, sigma, gz, ldgz, float(kl_weight),
             g_heads, N, Z)


def softplus_fwd(plan, x, ldx, y, ldy, rows, Cn):
    plan.add("softplus_fwd", _L().acimg_softplus_fwd, x, int(ldx), y, int(ldy), int(rows), int(Cn))


def softplus_bwd(plan, x, ldx, gy, ldgy, gx, ldgx, rows, Cn):
    plan.add("softplus_bwd", _L().acimg_softplus_bwd, x, int(ldx), gy, int(ldgy), gx, int(ldgx), int(rows), int(Cn))


def tapconv_stats_rows(d):
    return int(_L().acimg_tapconv_stats_rows(C.byref(d)))


def tapconv_pack(plan, d, w, wt, ldwt):
    plan.add("tapconv_pack", _L().acimg_tapconv_pack, C.byref(d), w, wt, int(ldwt))


def tapconv_unpack(plan, d, dwt, ldwt, w, decay, dw):
    plan.add("tapconv_unpack", _L().acimg_tapconv_unpack, C.byref(d), dwt, int(ldwt), w, float(decay), dw)


def tapconv_gather(plan, d, z, ldz, y, stats=None):
    plan.add("tapconv_gather", _L().acimg_tapconv_gather, C.byref(d), z, int(ldz), y, stats)


def tapconv_scatter(plan, d, gy, ldgy, gz, ldgz):
    plan.add("tapconv_scatter", _L().acimg_tapconv_scatter, C.byref(d), gy, int(ldgy), gz, int(ldgz))


def triplet_loss_workspace(B):
    return int(_L().acimg_triplet_loss_workspace(int(B)))


def triplet_loss_fwd(plan, e0, lde0, e1, lde1, labels, scenario, B, D, margin, hard, ws, out):
    """ws: a uint8 / float buffer of triplet_loss_workspace(B) bytes kept for the matching backward"""
    plan.add("triplet_loss_fwd", _L().acimg_triplet_loss_fwd, e0, int(lde0), e1, int(lde1), labels, scenario, int(B),
             int(D), float(margin), int(hard), ws, int(ws.numel() * ws.element_size()), out)


def triplet_loss_bwd(plan, e0, lde0, e1, lde1, B, D, weight, ws, g0, ldg0, g1, ldg1, accumulate=False):
    plan.add("triplet_loss_bwd", _L().acimg_triplet_loss_bwd, e0, int(lde0), e1, int(lde1), int(B), int(D),
             float(weight), ws, int(ws.numel() * ws.element_size()), g0, int(ldg0), g1, int(ldg1), int(accumulate))


def latent_linear_fwd(plan, heads, eps, z, ldz, kl, N, Z):
    plan.add("latent_linear_fwd", _L().acimg_latent_linear_fwd, heads, eps, z, int(ldz), kl, N, Z)


def latent_linear_bwd(plan, heads, eps, gz, ldgz, kl_weight, g_heads, N, Z):
    plan.add("latent_linear_bwd", _L().acimg_latent_linear_bwd, heads, eps, gz, int(ldgz), float(kl_weight), g_heads,
             N, Z)


def bn_bwd(plan, x, ldx, gy, ldgy, scale, shift, save_mean, save_invstd, gamma, rows, Cn, gx, ldgx, dgamma, dbeta):
    L = _L()
    plan.ws.require(L.acimg_bn_bwd_workspace(int(rows), int(Cn)))
    plan.add("bn_bwd", L.acimg_bn_bwd, x, int(ldx), gy, int(ldgy), scale, shift, save_mean, save_invstd, gamma,
             int(rows), int(Cn), gx, int(ldgx), dgamma, dbeta, _WsPtr(plan.ws), _WsBytes(plan.ws))


def maxpool_fwd(plan, x, ldx, y, ldy, N, H, W, Cn, k):
    plan.add("maxpool_fwd", _L().acimg_maxpool_fwd, x, int(ldx), y, int(ldy), N, H, W, Cn, k)


def maxpool_relu_bwd(plan, x, ldx, gy, ldgy, gx, ldgx, N, H, W, Cn, k):
    plan.add("maxpool_relu_bwd", _L().acimg_maxpool_relu_bwd, x, int(ldx), gy, int(ldgy), gx, int(ldgx), N, H, W, Cn, k)


def spatial_sum(plan, x, ldx, y, N, P, Cn):
    plan.add("spatial_sum", _L().acimg_spatial_sum, x, int(ldx), y, N, P, Cn)


def spatial_sum_relu_bwd(plan, x, ldx, gy, gx, ldgx, N, P, Cn):
    plan.add("spatial_sum_relu_bwd", _L().acimg_spatial_sum_relu_bwd, x, int(ldx), gy, gx, int(ldgx), N, P, Cn)


def clip_softmax_ce(plan, logits, ldl, clips, F, K, labels, out, g_logits, ldg):
    plan.add("clip_softmax_ce", _L().acimg_clip_softmax_ce, logits, int(ldl), clips, F, K, labels, out, g_logits,
             int(ldg))


_LOSS_SCRATCH = {}

def loss_scratch(device):
    """the dedicated, zero-initialised hand-off buffer that makes acimg_recon_loss / acimg_sumsq bit-reproducible (one
    per device: these launches are stream-ordered, the last workgroup of each leaves its ticket at zero)"""
    key = str(device)
    if key not in _LOSS_SCRATCH:
        _LOSS_SCRATCH[key] = torch.zeros(int(_L().acimg_loss_scratch_bytes()), dtype=torch.uint8, device=device)
    return _LOSS_SCRATCH[key]


def recon_loss(plan, yhat, target, g_logit, sums, count, w_mse=1.0, w_huber=1.0):
    sc = loss_scratch(plan.device)
    plan.add("recon_loss", _L().acimg_recon_loss, yhat, target, g_logit, sums, int(count),
             float(w_mse), float(w_huber), sc, int(sc.numel()))


def grad_slice(plan, src, ldsrc, dst, lddst, mask, ldmask, pixels, Cn, accumulate=False):
    plan.add("grad_slice", _L().acimg_grad_slice, src, ldsrc, dst, lddst, mask, ldmask, int(pixels),
             Cn, int(accumulate))


def loss_finalize(plan, sums, kl, N, count, latent_w, half_wd, w_mse, w_huber, out):
    plan.add("loss_finalize", _L().acimg_loss_finalize, sums, kl, N, float(count), float(latent_w),
             float(half_wd), float(w_mse), float(w_huber), out)


def zero(plan, t, nbytes=None):
    n = nbytes if nbytes is not None else t.numel() * t.element_size()
    plan.add("zero", _L().acimg_zero, t, int(n))


def randn(plan, out, n, seed, offset=0):
    plan.add("randn", _L().acimg_randn, out, int(n), int(seed), int(offset))


def sqerr_channels(plan, a, b, pixels, Cn, out):
    plan.add("sqerr_channels", _L().acimg_sqerr_channels, a, b, int(pixels), int(Cn), out)


def sumsq(plan, x, n, out):
    sc = loss_scratch(plan.device)
    plan.add("sumsq", _L().acimg_sumsq, x, int(n), out, sc, int(sc.numel()))


def axpy(plan, a, x, y, n):
    plan.add("axpy", _L().acimg_axpy, float(a), x, y, int(n))


def adam_step(plan, p, g, m, v, n, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    plan.add("adam_step", _L().acimg_adam_step, p, g, m, v, int(n), float(lr_t), float(beta1),
             float(beta2), float(eps), float(grad_scale))


def mfcc_frontend(plan, frames, window, melfb, dctl, out, nframes, normalize=False):
    fn = _L().acimg_mfcc_frontend if frames.dtype == torch.int32 else _L().acimg_mfcc_frontend_f32
    plan.add("mfcc_frontend", fn, frames, window, melfb, dctl, out, int(nframes), int(bool(normalize)))


def find_logen(plan, mfcc_img, idct, out, pixels):
    plan.add("find_logen", _L().acimg_find_logen, mfcc_img, idct, out, int(pixels))


def mask_iou(plan, map_a, map_b, N, P, iou):
    plan.add("mask_iou", _L().acimg_mask_iou, map_a, map_b, int(N), int(P), iou)


def adam_lr_t(lr, step, beta1=0.9, beta2=0.999):
    """TF-1 Adam effective step size for 1-based step t."""
    return lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)


def stft_mag(plan, wav, norm, window, twiddle, out, clips, nsamples, frame_len, step, fft_len=512):
    plan.add("stft_mag", _L().acimg_stft_mag, wav, norm, window, twiddle, out, int(clips), int(nsamples), int(frame_len),
             int(step), int(fft_len))


def absmax(plan, x, rows, n, out):
    plan.add("absmax", _L().acimg_absmax, x, int(rows), int(n), out)


def resize_bilinear(plan, x, y, N, H, W, Cn, OH, OW):
    plan.add("resize_bilinear", _L().acimg_resize_bilinear, x, y, int(N), int(H), int(W), int(Cn), int(OH), int(OW))


def filtfilt(plan, x, rows, n, ba, zi, out):
    plan.ws.require(_L().acimg_filtfilt_workspace(int(rows), int(n)))
    plan.add("filtfilt", _L().acimg_filtfilt, x, int(x.dtype == torch.int32), int(rows), int(n), ba, zi, out,
             _WsPtr(plan.ws), _WsBytes(plan.ws))
