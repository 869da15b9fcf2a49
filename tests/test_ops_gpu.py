"""GPU parity of each HIP op (through the C ABI) against a plain PyTorch CPU fp64 statement of the
same TensorFlow op.  Tolerance: the kernels compute in fp32 (exact-f32 MFMA, fp32 accumulate), so
results must agree with the fp64 reference to 2e-5 relative to the tensor's max magnitude.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def up4(v):
    return (v + 3) & ~3


def close(got, ref, tol=RTOL, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-12)
    err = (got - ref).abs().max().item() / scale
    assert np.isfinite(err) and err <= tol, "%s: rel err %.3e > %.1e" % (what, err, tol)


def tf_conv_ref(x, w, stride, pads, bias=None):
    """x NHWC fp64, w HWIO; pads = (pt, pb, pl, pr)"""
    xt = x.permute(0, 3, 1, 2)
    xt = F.pad(xt, (pads[2], pads[3], pads[0], pads[1]))
    y = F.conv2d(xt, w.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def same_pads(size, k, s):
    out = -(-size // s)
    tot = max((out - 1) * s + k - size, 0)
    return out, tot // 2, tot - tot // 2


def rnd(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float64)


def dev(t, device):
    return t.float().contiguous().to(device)


def pad_last(t, n):
    if t.shape[-1] == n:
        return t
    out = torch.zeros(*t.shape[:-1], n, dtype=t.dtype)
    out[..., : t.shape[-1]] = t
    return out


def pad_w(w, cp, kp):
    """HWIO -> zero padded [R,S,cp,kp]"""
    R, S, Cc, K = w.shape
    out = torch.zeros(R, S, cp, kp, dtype=w.dtype)
    out[:, :, :Cc, :K] = w
    return out


CONV_CASES = [
    # N, H, W, C, K, R, S, stride, padding
    (2, 9, 11, 8, 20, 3, 3, 1, "SAME"),
    (2, 12, 16, 12, 133, 3, 3, 1, "SAME"),
    (3, 36, 48, 16, 32, 3, 3, 3, "SAME"),
    (2, 13, 10, 64, 32, 1, 1, 1, "SAME"),
    (2, 30, 37, 3, 64, 7, 7, 2, 3),
    (2, 14, 19, 64, 12, 3, 4, 1, "VALID"),
    (4, 1, 1, 2816, 300, 1, 1, 1, "VALID"),
    (2, 17, 23, 133, 128, 3, 3, 1, "SAME"),
    (1, 20, 15, 32, 256, 3, 3, 2, 1),
    (2, 8, 9, 256, 128, 3, 3, 1, "SAME"),
]


def _conv_geom(H, W, R, S, stride, padding):
    if padding == "SAME":
        OH, pt, pb = same_pads(H, R, stride)
        OW, pl, pr = same_pads(W, S, stride)
    elif padding == "VALID":
        OH, OW = (H - R) // stride + 1, (W - S) // stride + 1
        pt = pb = pl = pr = 0
    else:
        p = padding
        OH, OW = (H + 2 * p - R) // stride + 1, (W + 2 * p - S) // stride + 1
        pt = pb = pl = pr = p
    return OH, OW, (pt, pb, pl, pr)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd(device, case):
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)))
    x = rnd(g, N, H, W, Cc)
    w = rnd(g, R, S, Cc, K) * 0.1
    b = rnd(g, K)
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    ref = torch.relu(tf_conv_ref(x, w, stride, pads, b))
    cp, kp = up4(Cc), up4(K)
    xd = dev(pad_last(x, cp), device)
    wd = dev(pad_w(w, cp, kp), device)
    bd = dev(pad_last(b, kp), device)
    y = torch.full((N, OH, OW, kp), 7.0, device=device)
    d = ops.conv_desc(N, H, W, cp, K, R, S, stride, padding, act=ops.ACT_RELU)
    assert (d.OH, d.OW) == (OH, OW)
    plan = ops.Plan(device, eager=True)
    ops.conv2d_fwd(plan, d, xd, wd, bd, y)
    torch.cuda.synchronize()
    close(y[..., :K], ref, what="conv fwd %s" % (case,))
    if kp != K:  # pad channels untouched (store guarded by K)
        assert (y[..., K:] == 7.0).all()


def test_conv2d_fwd_affine_stats_slice(device):
    """deferred BN (scale/shift/relu on load, zero padding after the affine), raw-output statistics,
    and writing into a channel slice of a wider buffer"""
    from acimg import ops

    g = torch.Generator().manual_seed(5)
    N, H, W, Cc, K = 3, 21, 17, 64, 128
    x = rnd(g, N, H, W, Cc)
    sc = rnd(g, Cc).abs() + 0.5
    sh = rnd(g, Cc)
    w = rnd(g, 3, 3, Cc, K) * 0.05
    xa = torch.relu(x * sc + sh)
    ref = tf_conv_ref(xa, w, 1, (1, 1, 1, 1))
    xd, wd = dev(x, device), dev(w, device)
    ybuf = torch.zeros(N, H, W, 256, device=device)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME", ldy=256)
    rows = ops.conv2d_stats_rows(d)
    stats = torch.zeros(rows, 2, K, device=device)
    plan = ops.Plan(device, eager=True)
    ops.conv2d_fwd(plan, d, xd, wd, None, ops.Ptr(ybuf, 128), dev(sc, device), dev(sh, device), 1, stats)
    torch.cuda.synchronize()
    close(ybuf[..., 128:], ref, what="affine conv")
    assert (ybuf[..., :128] == 0).all()
    flat = ref.reshape(-1, K)
    close(stats[:, 0].sum(0), flat.sum(0), tol=1e-4, what="stats sum")
    close(stats[:, 1].sum(0), (flat * flat).sum(0), tol=1e-4, what="stats sumsq")
    # bn_finalize on those partials
    gamma, beta = rnd(g, K).abs() + 0.5, rnd(g, K)
    mm, mv = rnd(g, K), rnd(g, K).abs()
    mmd, mvd = dev(mm, device), dev(mv, device)
    scale = torch.empty(K, device=device)
    shift = torch.empty(K, device=device)
    smean = torch.empty(K, device=device)
    sinv = torch.empty(K, device=device)
    cnt = flat.shape[0]
    ops.bn_finalize(plan, stats, rows, K, K, cnt, dev(gamma, device), dev(beta, device), mmd, mvd, scale,
                    shift, 0.997, 1e-5, True, smean, sinv)
    torch.cuda.synchronize()
    mean = flat.mean(0)
    var = flat.var(0, unbiased=False)
    rs = gamma / torch.sqrt(var + 1e-5)
    close(scale, rs, tol=1e-4, what="bn scale")
    close(shift, beta - mean * rs, tol=1e-4, what="bn shift")
    close(mmd, 0.997 * mm + 0.003 * mean, tol=1e-5, what="moving mean")
    close(mvd, 0.997 * mv + 0.003 * flat.var(0, unbiased=True), tol=1e-5, what="moving var")
    close(smean, mean, tol=1e-4, what="save mean")
    # inference mode uses the moving statistics
    ops.bn_finalize(plan, None, 0, K, K, 0, dev(gamma, device), dev(beta, device), mmd, mvd, scale, shift,
                    0.997, 1e-5, False)
    torch.cuda.synchronize()
    rs2 = gamma / torch.sqrt(mvd.cpu().double() + 1e-5)
    close(scale, rs2, tol=1e-5, what="bn scale (eval)")


SPLIT3_CASES = [
    # N, H, W, C, K, R, S, stride, padding   (C % 32 == 0)
    (2, 19, 23, 64, 64, 3, 3, 1, "SAME"),
    (2, 19, 23, 256, 128, 1, 1, 1, "SAME"),
    (3, 30, 28, 128, 128, 3, 3, 2, 1),
    (2, 14, 19, 512, 512, 3, 3, 1, "SAME"),
    (1, 56, 75, 64, 256, 1, 1, 1, "SAME"),
    (5, 28, 38, 256, 1024, 1, 1, 1, "SAME"),
]


@pytest.mark.parametrize("case", SPLIT3_CASES)
def test_conv2d_fwd_f16x3(device, case):
    """split-fp16 (hi/lo) MFMA forward conv with deferred BN on load and statistics partials vs fp64.
    Tolerance 2e-6 of the output's max magnitude: three fp16 MFMAs per product keep 22 mantissa bits of
    each operand, i.e. the same class as the exact-f32 kernel (a bf16 split measured ~5e-6)."""
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 7)
    x = rnd(g, N, H, W, Cc)
    sc, sh = rnd(g, Cc).abs() + 0.5, rnd(g, Cc)
    w = rnd(g, R, S, Cc, K) * (2.0 / (R * S * Cc)) ** 0.5
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    ref = tf_conv_ref(torch.relu(x * sc + sh), w, stride, pads)
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    y = torch.full((N, OH, OW, K), 7.0, device=device)
    rows = ops.conv2d_fwd_split3_stats_rows(d)
    stats = torch.zeros(rows, 2, K, device=device)
    plan = ops.Plan(device, eager=True)
    ops.conv2d_split3_prepare(plan, d, dev(w, device), wsplit)
    ops.conv2d_fwd_split3(plan, d, dev(x, device), wsplit, y, dev(sc, device), dev(sh, device), 1, stats)
    torch.cuda.synchronize()
    close(y, ref, tol=2e-6, what="f16x3 conv %s" % (case,))
    flat = ref.reshape(-1, K)
    close(stats[:, 0].sum(0), flat.sum(0), tol=2e-4, what="f16x3 stats sum")
    close(stats[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="f16x3 stats sumsq")
    # without the affine (plain input), no statistics
    ops.conv2d_fwd_split3(plan, d, dev(x, device), wsplit, y)
    torch.cuda.synchronize()
    close(y, tf_conv_ref(x, w, stride, pads), tol=2e-6, what="f16x3 conv plain %s" % (case,))


@pytest.mark.parametrize("case", [(2, 36, 48, 128, 128, 3, 3, 1, "SAME"), (2, 36, 48, 256, 128, 3, 3, 1, "SAME"),
                                  (3, 18, 20, 128, 64, 3, 3, 1, "SAME"), (2, 17, 9, 64, 64, 3, 3, 1, "SAME")])
def test_split3_generator_convs(device, case):
    """generator convs on the split-MFMA kernels: forward f16x3 with bias + ReLU into a concat slice, data
    gradient bf16x3 (gradients are outside fp16's range) with fan-in residual + ReLU mask"""
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 3)
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, R, S, Cc, K) * (2.0 / (R * S * Cc)) ** 0.5).requires_grad_(True)
    b = rnd(g, K).requires_grad_(True)
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    pre = tf_conv_ref(x, w, stride, pads, b)
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding, ldy=2 * K, act=ops.ACT_RELU)
    plan = ops.Plan(device, eager=True)
    wd = dev(w.detach(), device)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, wd, wsplit)
    ybuf = torch.zeros(N, OH, OW, 2 * K, device=device)
    ops.conv2d_fwd_split3(plan, d, dev(x.detach(), device), wsplit, ops.Ptr(ybuf, K), bias=dev(b.detach(), device))
    torch.cuda.synchronize()
    close(ybuf[..., K:], torch.relu(pre), tol=2e-6, what="f16x3 fwd+bias+relu %s" % (case,))
    assert (ybuf[..., :K] == 0).all()
    # data gradient with tiny gradient magnitudes (as in the real backward pass)
    gy = rnd(g, N, OH, OW, K) * 1e-7
    pre.backward(gy)
    res = rnd(g, N, H, W, Cc) * 1e-7
    maskt = rnd(g, N, H, W, Cc)
    wt = torch.zeros(ops.conv2d_split3_dgrad_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare_dgrad(plan, d, wd, wt)
    dx = torch.full((N, H, W, Cc), 3.0, device=device)
    ops.conv2d_dgrad_split3(plan, d, dev(gy, device), K, wt, dx, dev(res, device), Cc, dev(maskt, device), Cc)
    torch.cuda.synchronize()
    close(dx, (x.grad + res) * (maskt > 0), tol=3e-5, what="bf16x3 dgrad %s" % (case,))
    # weight + bias gradient (bf16x3, transposing LDS reads)
    dw = torch.full((R, S, Cc, K), 9.0, device=device)
    db = torch.zeros(K, device=device)
    ops.conv2d_wgrad_split3(plan, d, dev(x.detach(), device), dev(gy, device), K, dw, db)
    torch.cuda.synchronize()
    close(dw, w.grad, tol=3e-5, what="bf16x3 wgrad %s" % (case,))
    close(db, b.grad, tol=3e-5, what="bf16x3 bgrad %s" % (case,))


@pytest.mark.parametrize("case", [(2, 36, 48, 128, 128, 3, 3, 3, "SAME"), (3, 12, 16, 136, 128, 3, 3, 1, "SAME"),
                                  (2, 36, 48, 12, 128, 3, 3, 1, "SAME"), (5, 9, 7, 64, 64, 3, 3, 1, "SAME"),
                                  (4, 14, 19, 2048, 144, 1, 1, 1, "SAME")])      # 144 columns: workspace sized for either tile rule
def test_wgrad_split3_geometries(device, case):
    """strided (pool_2), padded-channel, tiny-C and ragged-M weight gradients on the bf16x3 kernel; gy read
    as a channel slice of a wider buffer"""
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 5)
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, R, S, Cc, K) * 0.05).requires_grad_(True)
    b = rnd(g, K).requires_grad_(True)
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    y = tf_conv_ref(x, w, stride, pads, b)
    gywide = rnd(g, N, OH, OW, 2 * K) * 1e-6
    y.backward(gywide[..., K:])
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding)
    plan = ops.Plan(device, eager=True)
    dw = torch.full((R, S, Cc, K), 9.0, device=device)
    db = torch.zeros(K, device=device)
    ops.conv2d_wgrad_split3(plan, d, dev(x.detach(), device), ops.Ptr(dev(gywide, device), K), 2 * K, dw, db)
    torch.cuda.synchronize()
    close(dw, w.grad, tol=3e-5, what="bf16x3 wgrad %s" % (case,))
    close(db, b.grad, tol=3e-5, what="bf16x3 bgrad %s" % (case,))


def plane_bytes(rows, Cc):
    """bytes of one split-format plane: whole 16-pixel x 32-channel bricks (include/acimg.h)"""
    return -(-rows // 16) * 16 * Cc * 2


def unsplit(planes, lo_off, rows, Cc):
    """two fp16 planes in LDS-tile order (uint8 buffer) -> float64 [rows, C] (undoing the 2^-2 scale): brick (row >> 4,
    c >> 5) of 1 KiB, inside it row & 15 at 64 bytes and 16-byte group (c >> 3) & 3 at group ((c >> 3) ^ -(row >> 2)) & 3"""
    r = torch.arange(rows).view(-1, 1)
    c = torch.arange(Cc).view(1, -1)
    off = ((r >> 4) * (Cc // 32) + (c >> 5)) * 1024 + (r & 15) * 64 + ((((c >> 3) ^ (-(r >> 2))) & 3) << 4) + (c & 7) * 2
    idx = (off // 2).reshape(-1).to(planes.device)
    n = plane_bytes(rows, Cc)
    hi = planes[:n].view(torch.float16)[idx].double()
    lo = planes[lo_off: lo_off + n].view(torch.float16)[idx].double()
    return ((hi + lo) * 4.0).reshape(rows, Cc).cpu()


@pytest.mark.parametrize("tail", [False, True])
@pytest.mark.parametrize("case", SPLIT3_CASES[:4] + [(8, 56, 75, 128, 128, 3, 3, 1, "SAME")])
def test_presplit_activation_path(device, case, tail):
    """pre-split activation format: bn_relu_split producer + split3p conv (with statistics) vs fp64, and
    the planes themselves vs the values they encode (22 mantissa bits: 5e-7 relative to max)"""
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 9)
    x = rnd(g, N, H, W, Cc)
    sc, sh = rnd(g, Cc).abs() + 0.5, rnd(g, Cc)
    w = rnd(g, R, S, Cc, K) * (2.0 / (R * S * Cc)) ** 0.5
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    xa = torch.relu(x * sc + sh)
    ref = tf_conv_ref(xa, w, stride, pads)
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding)
    rows = N * H * W
    lo_off = plane_bytes(rows, Cc)
    planes = torch.zeros(lo_off * 2, dtype=torch.uint8, device=device)
    plan = ops.Plan(device, eager=True)
    ops.bn_relu_split(plan, dev(x, device), dev(sc, device), dev(sh, device), 1, planes, lo_off, rows, Cc)
    torch.cuda.synchronize()
    close(unsplit(planes, lo_off, rows, Cc), xa.reshape(rows, Cc), tol=5e-7, what="split planes")
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, dev(w, device), wsplit)
    y = torch.full((N, OH, OW, K), 7.0, device=device)
    srows = ops.conv2d_fwd_split3_stats_rows(d)
    stats = torch.zeros(srows, 2, K, device=device)
    # tail=True: the tiles of the last partial round of workgroups are cut into K ranges (partials + tickets in a
    # dedicated workspace); run twice — the tickets must be back at zero, and the result must not change at all
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=device) if tail else None
    ops.conv2d_fwd_split3p(plan, d, planes, lo_off, wsplit, y, stats, tail_ws=tws)
    torch.cuda.synchronize()
    close(y, ref, tol=2e-6, what="split3p conv %s" % (case,))
    if tail:
        assert int(tws[:4096].view(torch.int32).abs().sum()) == 0
        y2 = torch.full_like(y, 3.0)
        ops.conv2d_fwd_split3p(plan, d, planes, lo_off, wsplit, y2, stats, tail_ws=tws)
        torch.cuda.synchronize()
        assert torch.equal(y, y2), "tail split must be deterministic"
    flat = ref.reshape(-1, K)
    close(stats[:, 0].sum(0), flat.sum(0), tol=2e-4, what="split3p stats sum")
    close(stats[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="split3p stats sumsq")


def test_presplit_unit_output_and_pool(device):
    from acimg import ops

    g = torch.Generator().manual_seed(21)
    plan = ops.Plan(device, eager=True)
    N, OH, OW, Cc = 2, 7, 9, 64
    rows = N * OH * OW
    lo = plane_bytes(rows, Cc)
    a, sa, ta = rnd(g, N, OH, OW, Cc), rnd(g, Cc), rnd(g, Cc)
    # projection shortcut (raw fp32 + affine), planes + fp32 copy out
    b, sb, tb = rnd(g, N, OH, OW, Cc), rnd(g, Cc), rnd(g, Cc)
    out = torch.zeros(2 * lo, dtype=torch.uint8, device=device)
    out32 = torch.empty(N, OH, OW, Cc, device=device)
    ops.bn_add_relu_split(plan, dev(a, device), dev(sa, device), dev(ta, device), dev(b, device), dev(sb, device),
                          dev(tb, device), None, 0, out, lo, out32, N, OH, OW, Cc, OH, OW, 1)
    ref = torch.relu(a * sa + ta + b * sb + tb)
    torch.cuda.synchronize()
    close(out32, ref, what="unit out fp32")
    close(unsplit(out, lo, rows, Cc), ref.reshape(rows, Cc), tol=5e-7, what="unit out planes")
    # identity shortcut read back from split planes, with stride-2 subsampling
    BH, BW = 2 * OH - 1, 2 * OW
    prev = torch.relu(rnd(g, N, BH, BW, Cc))
    prow = N * BH * BW
    plo = plane_bytes(prow, Cc)
    pplanes = torch.zeros(2 * plo, dtype=torch.uint8, device=device)
    ops.bn_relu_split(plan, dev(prev, device), None, None, 0, pplanes, plo, prow, Cc)
    ops.bn_add_relu_split(plan, dev(a, device), dev(sa, device), dev(ta, device), None, None, None, pplanes, plo, out,
                          lo, None, N, OH, OW, Cc, BH, BW, 2)
    torch.cuda.synchronize()
    close(unsplit(out, lo, rows, Cc), torch.relu(a * sa + ta + prev[:, ::2, ::2]).reshape(rows, Cc), tol=1e-6,
          what="unit out (identity from planes)")
    # pool1 writing planes
    N, H, W, Cc = 2, 112, 149, 64
    x, sc, sh = rnd(g, N, H, W, Cc), rnd(g, Cc), rnd(g, Cc)
    OHp, pt, pb = same_pads(H, 3, 2)
    OWp, pl, pr = same_pads(W, 3, 2)
    xa = torch.relu(x * sc + sh).permute(0, 3, 1, 2)
    ref = F.max_pool2d(F.pad(xa, (pl, pr, pt, pb), value=-1e30), 3, 2).permute(0, 2, 3, 1)
    rows = N * OHp * OWp
    lo = plane_bytes(rows, Cc)
    out = torch.zeros(2 * lo, dtype=torch.uint8, device=device)
    ops.bn_relu_maxpool_split(plan, dev(x, device), dev(sc, device), dev(sh, device), out, lo, N, H, W, Cc, OHp, OWp,
                              pt, pl)
    torch.cuda.synchronize()
    close(unsplit(out, lo, rows, Cc), ref.reshape(rows, Cc), tol=5e-7, what="pool planes")


DGRAD_CASES = [
    (2, 9, 11, 8, 20, 3, 3, 1, "SAME"),
    (2, 12, 16, 12, 133, 3, 3, 1, "SAME"),
    (2, 12, 16, 133, 128, 3, 3, 1, "SAME"),
    (2, 7, 6, 64, 12, 3, 3, 1, "SAME"),
    (3, 1, 1, 152, 2304, 1, 1, 1, "VALID"),
    (2, 36, 48, 16, 32, 3, 3, 3, "SAME"),
    (2, 18, 24, 256, 128, 3, 3, 1, "SAME"),
    (8, 12, 16, 128, 133, 3, 3, 1, "SAME"),     # 133 columns: 64-column weight-gradient tiles (3 x 64 instead of 2 x 128)
    # general stride (the strided-conv "pool" layers of the RGB / spectrogram U-Nets)
    (2, 17, 23, 8, 8, 3, 3, 2, "SAME"),
    (2, 12, 15, 32, 32, 2, 3, 2, "VALID"),
    (2, 15, 18, 8, 16, 3, 2, 2, "VALID"),
    (1, 16, 20, 64, 64, 3, 3, 2, "SAME"),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv2d_dgrad_wgrad(device, case):
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 1)
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, R, S, Cc, K) * 0.1).requires_grad_(True)
    b = rnd(g, K).requires_grad_(True)
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    y = tf_conv_ref(x, w, stride, pads, b)
    gy = rnd(g, N, OH, OW, K)
    y.backward(gy)
    res = rnd(g, N, H, W, Cc)
    maskt = rnd(g, N, H, W, Cc)
    ref_dx = (x.grad + res) * (maskt > 0)
    cp, kp = up4(Cc), up4(K)
    d = ops.conv_desc(N, H, W, cp, K, R, S, stride, padding)
    wd = dev(pad_w(w.detach(), cp, kp), device)
    gyd = dev(pad_last(gy, kp), device)
    dx = torch.full((N, H, W, cp), 3.0, device=device)
    plan = ops.Plan(device, eager=True)
    ops.conv2d_dgrad(plan, d, gyd, kp, wd, dx, dev(pad_last(res, cp), device), cp,
                     dev(pad_last(maskt, cp), device), cp)
    torch.cuda.synchronize()
    close(dx[..., :Cc], ref_dx, what="dgrad %s" % (case,))
    # weight / bias gradient
    dw = torch.full((R, S, cp, kp), 9.0, device=device)
    db = torch.zeros(kp, device=device)
    ops.conv2d_wgrad(plan, d, dev(pad_last(x.detach(), cp), device), gyd, kp, dw, db)
    torch.cuda.synchronize()
    close(dw[:, :, :Cc, :K], w.grad, what="wgrad %s" % (case,))
    close(db[:K], b.grad, what="bgrad %s" % (case,))
    if cp != Cc:
        assert (dw[:, :, Cc:, :] == 0).all()
    if kp != K:
        assert (dw[:, :, :, K:] == 0).all()


def test_dgrad_slice_input(device):
    """gy given as a channel slice of a wider gradient buffer (tf.concat backward)"""
    from acimg import ops

    g = torch.Generator().manual_seed(11)
    N, H, W, Cc, K = 2, 10, 12, 32, 128
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, 3, 3, Cc, K) * 0.1).requires_grad_(True)
    y = tf_conv_ref(x, w, 1, (1, 1, 1, 1))
    gywide = rnd(g, N, H, W, 256)
    y.backward(gywide[..., 128:])
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME")
    gyd = dev(gywide, device)
    dx = torch.empty(N, H, W, Cc, device=device)
    plan = ops.Plan(device, eager=True)
    ops.conv2d_dgrad(plan, d, ops.Ptr(gyd, 128), 256, dev(w.detach(), device), dx)
    dw = torch.empty(3, 3, Cc, K, device=device)
    ops.conv2d_wgrad(plan, d, dev(x.detach(), device), ops.Ptr(gyd, 128), 256, dw, None)
    torch.cuda.synchronize()
    close(dx, x.grad, what="dgrad slice")
    close(dw, w.grad, what="wgrad slice")


@pytest.mark.parametrize("case", [(2, 12, 16, 128, 128, 2, 2, 3), (3, 5, 4, 32, 64, 2, 2, 2), (2, 6, 7, 64, 32, 3, 3, 3),
                                  # kernel > stride (overlapping patches): unet_architecture.py upsample_6/8 [2,3],
                                  # unet_sound.py upsample_8 [3,2] and upsample_9 [3,3]
                                  (2, 7, 9, 64, 32, 2, 3, 2), (2, 6, 8, 32, 8, 3, 3, 2), (2, 5, 6, 8, 8, 3, 2, 2)])
def test_deconv(device, case):
    from acimg import ops

    N, H, W, Cc, K, R, S, s = case
    g = torch.Generator().manual_seed(77)
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, R, S, K, Cc) * 0.1).requires_grad_(True)  # TF layout [kh, kw, out, in]
    b = rnd(g, K).requires_grad_(True)
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, stride=s,
                           output_padding=(max(s - R, 0), max(s - S, 0))).permute(0, 2, 3, 1)
    OH, OW = H * s + max(R - s, 0), W * s + max(S - s, 0)
    assert y.shape[1] == OH and y.shape[2] == OW
    gy = rnd(g, *y.shape)
    y.backward(gy)
    d = ops.deconv_desc(N, H, W, Cc, K, R, S, s, ldy=2 * K)
    assert (d.OH, d.OW) == (OH, OW)
    ybuf = torch.zeros(N, OH, OW, 2 * K, device=device)
    plan = ops.Plan(device, eager=True)
    wd = dev(w.detach(), device)
    ops.deconv_fwd(plan, d, dev(x.detach(), device), wd, dev(b.detach(), device), ybuf)
    torch.cuda.synchronize()
    close(ybuf[..., :K], y, what="deconv fwd")
    assert (ybuf[..., K:] == 0).all()
    gywide = torch.zeros(N, OH, OW, 2 * K, dtype=torch.float64)
    gywide[..., :K] = gy
    gyd = dev(gywide, device)
    maskt = rnd(g, N, H, W, Cc)
    dx = torch.empty(N, H, W, Cc, device=device)
    ops.deconv_dgrad(plan, d, gyd, 2 * K, wd, dx, dev(maskt, device), Cc)
    dw = torch.empty(R, S, K, Cc, device=device)
    db = torch.empty(K, device=device)
    ops.deconv_wgrad(plan, d, dev(x.detach(), device), gyd, 2 * K, dw, db)
    torch.cuda.synchronize()
    close(dx, x.grad * (maskt > 0), what="deconv dgrad")
    close(dw, w.grad, what="deconv wgrad")
    close(db, b.grad, what="deconv bgrad")


def test_trunk_elementwise(device):
    from acimg import ops

    g = torch.Generator().manual_seed(3)
    plan = ops.Plan(device, eager=True)
    # bn_add_relu with identity / subsampled / projected shortcut
    N, OH, OW, Cc = 2, 7, 9, 64
    a, sa, ta = rnd(g, N, OH, OW, Cc), rnd(g, Cc), rnd(g, Cc)
    b1 = rnd(g, N, OH, OW, Cc)
    out = torch.empty(N, OH, OW, Cc, device=device)
    ops.bn_add_relu(plan, dev(a, device), dev(sa, device), dev(ta, device), dev(b1, device), None, None, out,
                    N, OH, OW, Cc, OH, OW, 1)
    close(out, torch.relu(a * sa + ta + b1), what="bn_add_relu id")
    b2 = rnd(g, N, 2 * OH - 1, 2 * OW, Cc)
    ops.bn_add_relu(plan, dev(a, device), dev(sa, device), dev(ta, device), dev(b2, device), None, None, out,
                    N, OH, OW, Cc, 2 * OH - 1, 2 * OW, 2)
    close(out, torch.relu(a * sa + ta + b2[:, ::2, ::2]), what="bn_add_relu subsample")
    sb, tb = rnd(g, Cc), rnd(g, Cc)
    ops.bn_add_relu(plan, dev(a, device), dev(sa, device), dev(ta, device), dev(b1, device), dev(sb, device),
                    dev(tb, device), out, N, OH, OW, Cc, OH, OW, 1)
    close(out, torch.relu(a * sa + ta + b1 * sb + tb), what="bn_add_relu proj")
    # pool1: 3x3 s2 SAME on relu(bn(x)), odd sizes (H: pad 0/1, W: pad 1/1)
    N, H, W, Cc = 2, 112, 149, 8
    x, sc, sh = rnd(g, N, H, W, Cc), rnd(g, Cc), rnd(g, Cc)
    OH, pt, pb = same_pads(H, 3, 2)
    OW, pl, pr = same_pads(W, 3, 2)
    xa = torch.relu(x * sc + sh).permute(0, 3, 1, 2)
    ref = F.max_pool2d(F.pad(xa, (pl, pr, pt, pb), value=-1e30), 3, 2).permute(0, 2, 3, 1)
    out = torch.empty(N, OH, OW, Cc, device=device)
    ops.bn_relu_maxpool(plan, dev(x, device), dev(sc, device), dev(sh, device), out, N, H, W, Cc, OH, OW, pt, pl)
    close(out, ref, what="maxpool")
    # pad channels
    x3 = rnd(g, 5, 3)
    y4 = torch.empty(5, 4, device=device)
    ops.pad_channels(plan, dev(x3, device), y4, 5, 3, 4)
    close(y4[:, :3], x3, what="pad")
    assert (y4[:, 3] == 0).all()
    # bn_relu + its backward in batch-statistics mode
    rows, Cc = 384, 12
    x = rnd(g, rows, Cc).requires_grad_(True)
    gamma = (rnd(g, Cc).abs() + 0.5).requires_grad_(True)
    beta = rnd(g, Cc).requires_grad_(True)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    y = torch.relu((x - mean) * invstd * gamma + beta)
    gy = rnd(g, rows, Cc)
    y.backward(gy)
    scale = (gamma * invstd).detach()
    shift = (beta - mean * gamma * invstd).detach()
    yd = torch.empty(rows, Cc, device=device)
    ops.bn_relu(plan, dev(x.detach(), device), dev(scale, device), dev(shift, device), yd, rows, Cc, Cc, Cc)
    close(yd, y, what="bn_relu")
    gx = torch.empty(rows, Cc, device=device)
    dg = torch.empty(Cc, device=device)
    dbt = torch.empty(Cc, device=device)
    ops.bn_relu_bwd(plan, dev(x.detach(), device), yd, dev(gy, device), dev(gamma.detach(), device),
                    dev(mean.detach(), device), dev(invstd.detach(), device), gx, dg, dbt, rows, Cc)
    torch.cuda.synchronize()
    close(gx, x.grad, tol=1e-4, what="bn bwd gx")
    close(dg, gamma.grad, tol=1e-4, what="bn bwd dgamma")
    close(dbt, beta.grad, tol=1e-4, what="bn bwd dbeta")


def test_generator_elementwise(device):
    from acimg import ops

    g = torch.Generator().manual_seed(4)
    plan = ops.Plan(device, eager=True)
    # tile
    N = 3
    mf = rnd(g, N, 12)
    out = torch.empty(N, 36, 48, 12, device=device)
    ops.tile_mfcc(plan, dev(mf, device), out, N, 36 * 48, 12)
    close(out, mf.view(N, 1, 1, 12).expand(N, 36, 48, 12), tol=1e-7, what="tile")
    # min-max with ties at the minimum (post-ReLU zeros) written into a concat slice
    P, Cc, ld, ldo = 12 * 16, 133, 136, 148
    x = torch.relu(rnd(g, N, P, Cc)).requires_grad_(True)
    mn = x.amin(dim=(1, 2), keepdim=True)
    a = x - mn
    o = a / a.amax(dim=(1, 2), keepdim=True)
    go = rnd(g, N, P, Cc)
    o.backward(go)
    xd = dev(pad_last(x.detach(), ld), device)
    cat = torch.zeros(N, P, ldo, device=device)
    mm = torch.empty(N, 4, device=device)
    ops.minmax_fwd(plan, xd, ld, cat, ldo, mm, N, P, Cc)
    close(cat[..., :Cc], o, what="minmax fwd")
    assert (cat[..., Cc:] == 0).all()
    gow = torch.zeros(N, P, ldo, dtype=torch.float64)
    gow[..., :Cc] = go
    gx = torch.zeros(N, P, ld, device=device)
    ops.minmax_bwd(plan, xd, ld, dev(gow, device), ldo, mm, gx, ld, N, P, Cc, False, True)
    torch.cuda.synchronize()
    close(gx[..., :Cc], x.grad * (x.detach() > 0), tol=1e-4, what="minmax bwd (relu masked)")
    # no mask + accumulate, strictly positive input (single min, single max)
    x2 = (rnd(g, N, 7, 12).abs() + 0.1).requires_grad_(True)
    a2 = x2 - x2.amin(dim=(1, 2), keepdim=True)
    o2 = a2 / a2.amax(dim=(1, 2), keepdim=True)
    go2 = rnd(g, N, 7, 12)
    o2.backward(go2)
    base = rnd(g, N, 7, 12)
    o2d = torch.empty(N, 7, 12, device=device)
    ops.minmax_fwd(plan, dev(x2.detach(), device), 12, o2d, 12, mm, N, 7, 12)
    gx2 = dev(base, device)
    ops.minmax_bwd(plan, dev(x2.detach(), device), 12, dev(go2, device), 12, mm, gx2, 12, N, 7, 12, True, False)
    torch.cuda.synchronize()
    close(o2d, o2, what="minmax fwd 2")
    close(gx2, x2.grad + base, tol=1e-4, what="minmax bwd accumulate")
    # latent
    Z = 150
    heads = rnd(g, N, 2 * Z).requires_grad_(True)
    eps = rnd(g, N, Z)
    mu, sg = heads[:, :Z], F.softplus(heads[:, Z:])
    z = mu + sg * eps
    kl = 0.5 * (mu ** 2 + sg ** 2 - torch.log(1e-8 + sg ** 2) - 1).sum(1)
    gz = rnd(g, N, Z)
    klw = 1e-3
    ((z * gz).sum() + klw * kl.sum()).backward()
    hd, ed = dev(heads.detach(), device), dev(eps, device)
    zd = torch.zeros(N, 152, device=device)
    sd = torch.empty(N, Z, device=device)
    kld = torch.empty(N, device=device)
    ops.latent_fwd(plan, hd, ed, zd, 152, sd, kld, N, Z)
    close(zd[:, :Z], z, what="z")
    close(kld, kl, what="kl")
    gh = torch.empty(N, 2 * Z, device=device)
    ops.latent_bwd(plan, hd, ed, sd, dev(pad_last(gz, 152), device), 152, klw, gh, N, Z)
    torch.cuda.synchronize()
    close(gh, heads.grad, what="latent bwd")
    # reconstruction loss
    cnt = N * 36 * 48 * 12
    logit = rnd(g, cnt).requires_grad_(True)
    tgt = torch.rand(cnt, generator=g, dtype=torch.float64) * 4 - 1.5  # exercises |e| > 1
    yh = torch.sigmoid(logit)
    e = yh - tgt
    q = e.abs().clamp(max=1.0)
    mse, hub = (e * e).mean(), (0.5 * q * q + (e.abs() - q)).mean()
    (mse + hub).backward()
    sums = torch.zeros(4, device=device)
    gl = torch.empty(cnt, device=device)
    ops.recon_loss(plan, dev(yh.detach(), device), dev(tgt, device), gl, sums, cnt, 1.0, 1.0)
    outv = torch.empty(5, device=device)
    ops.sumsq(plan, dev(tgt, device), cnt, ops.Ptr(sums, 2))
    ops.loss_finalize(plan, sums, kld, N, cnt, 1e-6, 2.5e-4, 1.0, 1.0, outv)
    torch.cuda.synchronize()
    close(gl, logit.grad, what="recon grad")
    reg = 2.5e-4 * (tgt * tgt).sum()
    lat = 1e-6 * kl.detach().mean()
    close(outv, torch.stack([mse.detach(), hub.detach(), lat, reg, lat + mse.detach() + hub.detach() + reg]),
          what="loss scalars")
    # grad_slice
    src = rnd(g, 20, 24)
    dst0 = rnd(g, 20, 8)
    msk = rnd(g, 20, 8)
    dstd = dev(dst0, device)
    ops.grad_slice(plan, ops.Ptr(dev(src, device), 5), 24, dstd, 8, dev(msk, device), 8, 20, 7, True)
    torch.cuda.synchronize()
    ref = dst0.clone()
    ref[:, :7] = (src[:, 5:12] + dst0[:, :7]) * (msk[:, :7] > 0)
    close(dstd, ref, what="grad_slice")
    # adam (TF-1 form) + axpy + zero
    n = 1003
    p, gr, m, v = rnd(g, n), rnd(g, n), rnd(g, n) * 0.1, rnd(g, n).abs() * 0.01
    lr_t = ops.adam_lr_t(1e-3, 3)
    m2 = 0.9 * m + 0.1 * gr
    v2 = 0.999 * v + 0.001 * gr * gr
    p2 = p - lr_t * m2 / (torch.sqrt(v2) + 1e-8)
    pd, md, vd = dev(p, device), dev(m, device), dev(v, device)
    ops.adam_step(plan, pd, dev(gr, device), md, vd, n, lr_t)
    torch.cuda.synchronize()
    close(pd, p2, what="adam p")
    close(md, m2, what="adam m")
    close(vd, v2, what="adam v")
    yv = dev(p, device)
    ops.axpy(plan, 0.5, dev(gr, device), yv, n)
    close(yv, p + 0.5 * gr, what="axpy")
    ops.zero(plan, yv)
    torch.cuda.synchronize()
    assert (yv == 0).all()


def test_error_reporting(device):
    """bad descriptors come back as error codes with text, never as a crash"""
    from acimg import _lib, ops

    d = ops.conv_desc(1, 4, 4, 6, 8, 3, 3)  # C not a multiple of 4
    x = torch.zeros(1, 4, 4, 6, device=device)
    w = torch.zeros(3, 3, 6, 8, device=device)
    y = torch.zeros(1, 4, 4, 8, device=device)
    with pytest.raises(_lib.AcimgError) as ei:
        ops.conv2d_fwd(ops.Plan(device, eager=True), d, x, w, None, y)
    assert "multiples of 4" in str(ei.value)
    L = _lib.load()
    import ctypes as C
    st = ops.current_stream_handle(device)
    # a workspace that is too small is refused, not overrun (general-stride data gradient needs the dilated copy)
    d2 = ops.conv_desc(1, 17, 23, 8, 8, 3, 3, 2, "SAME")
    gy = torch.zeros(1, d2.OH, d2.OW, 8, device=device)
    w2 = torch.zeros(3, 3, 8, 8, device=device)
    dx = torch.zeros(1, 17, 23, 8, device=device)
    tiny = torch.zeros(64, device=device)
    rc = L.acimg_conv2d_dgrad(C.byref(d2), gy.data_ptr(), 8, w2.data_ptr(), dx.data_ptr(), 8, None, 0, None, 0,
                              tiny.data_ptr(), 256, None, st)
    assert rc == -2 and "workspace" in _lib.last_error()
    # transposed conv: kernel > stride needs the matching TF output size
    d3 = ops.deconv_desc(1, 5, 6, 8, 8, 3, 3, 2)
    d3.OH = 10
    rc = L.acimg_deconv_fwd(C.byref(d3), dx.data_ptr(), w2.data_ptr(), None, dx.data_ptr(), tiny.data_ptr(), 256, None,
                            st)
    assert rc == -1 and "OH" in _lib.last_error()
    # batch-norm backward: channel count not a multiple of 4; clip cross-entropy: too many classes
    rc = L.acimg_bn_bwd(dx.data_ptr(), 6, dx.data_ptr(), 6, None, None, None, None, None, 10, 6, dx.data_ptr(), 6,
                        None, None, tiny.data_ptr(), 256, st)
    assert rc == -1
    rc = L.acimg_clip_softmax_ce(dx.data_ptr(), 100, 1, 12, 100, None, None, None, 0, st)
    assert rc == -1 and "classes" in _lib.last_error()
    rc = L.acimg_maxpool_fwd(dx.data_ptr(), 8, dx.data_ptr(), 8, 1, 2, 2, 8, 3, st)
    assert rc == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(2, 200, 180, 8, 8, 3, 3, 1, "SAME"), (2, 190, 200, 4, 8, 3, 3, 1, "SAME"),
                                  (2, 200, 190, 16, 8, 3, 3, 1, "SAME"), (3, 224, 298, 8, 8, 3, 3, 2, "SAME"),
                                  (2, 150, 240, 8, 32, 2, 3, 2, "VALID"), (2, 180, 200, 8, 16, 3, 3, 1, "SAME"),
                                  (2, 225, 299, 8, 8, 3, 3, 2, "SAME"), (2, 200, 190, 8, 16, 3, 3, 2, "SAME")])
def test_few_channel_direct_conv(device, case):
    """the direct few-channel path (>= 65536 output pixels, C*K <= 512): forward with bias + ReLU + BN statistics,
    and the data gradient with a residual, vs fp64"""
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)) + 5)
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, R, S, Cc, K) * 0.2).requires_grad_(True)
    b = rnd(g, K)
    OH, OW, pads = _conv_geom(H, W, R, S, stride, padding)
    raw = tf_conv_ref(x, w, stride, pads, b)
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding, act=1)
    plan = ops.Plan(device, eager=True)
    y = torch.zeros(N, OH, OW, K, device=device)
    srows = ops.conv2d_stats_rows(d)
    stats = torch.zeros(srows, 2, K, device=device)
    wd = dev(w.detach(), device)
    ops.conv2d_fwd(plan, d, dev(x.detach(), device), wd, dev(b, device), y, stats=stats)
    torch.cuda.synchronize()
    close(y, torch.relu(raw.detach()), what="direct fwd %s" % (case,))
    # (statistics are those of the stored tensor = conv + bias with the activation applied here; BN layers use act NONE)
    d0 = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding, act=0)
    stats = torch.zeros(ops.conv2d_stats_rows(d0), 2, K, device=device)      # (the MFMA form has its own row count)
    ops.conv2d_fwd(plan, d0, dev(x.detach(), device), wd, dev(b, device), y, stats=stats)
    torch.cuda.synchronize()
    flat = raw.detach().reshape(-1, K)
    close(stats[:, 0].sum(0), flat.sum(0), tol=2e-4, what="direct stats sum")
    close(stats[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="direct stats sumsq")
    gy = rnd(g, N, OH, OW, K)
    raw.backward(gy)
    res = rnd(g, N, H, W, Cc)
    dx = torch.full((N, H, W, Cc), 3.0, device=device)
    if Cc % 8 == 0:
        ops.conv2d_dgrad(plan, d0, dev(gy, device), K, wd, dx, dev(res, device), Cc)
        torch.cuda.synchronize()
        # (the 3x3 cases take the bf16x3 MFMA form from 65536 pixels on - 16 mantissa bits per operand; with stride 2 the
        # zero-inserted view of gy is formed while the tile is staged: even and odd sizes = leading padding 0 and 1)
        close(dx, x.grad + res, tol=3e-5, what="direct dgrad %s" % (case,))


def test_tapconv_matches_valid_conv(device):
    """tap-GEMM form of a stride-1 VALID conv with few output channels (conv_map 3x4, C -> 12): forward incl. the
    batch-norm partials, and the weight gradient incl. the L2 term, vs torch in fp64"""
    from acimg import ops
    g = torch.Generator().manual_seed(11)
    N, H, W, Cin, K, R, S = 3, 7, 9, 64, 12, 3, 4
    OH, OW, TK = H - R + 1, W - S + 1, R * S * K
    x = rnd(g, N, H, W, Cin)
    w = rnd(g, R, S, Cin, K) * 0.1
    gy = rnd(g, N, OH, OW, K)
    d = ops.conv_desc(N, H, W, Cin, K, R, S, 1, "VALID", ldx=Cin, ldy=K, ldw=K)
    d1 = ops.conv_desc(N, H, W, Cin, TK, 1, 1, 1, "SAME", ldx=Cin, ldy=TK, ldw=TK)
    plan = ops.Plan(device, eager=True)
    xd, wd, gyd = dev(x, device), dev(w, device), dev(gy, device)
    wt = torch.zeros(Cin, TK, device=device)
    ops.tapconv_pack(plan, d, wd, wt, TK)
    assert torch.equal(wt.cpu(), w.float().permute(2, 0, 1, 3).reshape(Cin, TK))
    z = torch.zeros(N * H * W, TK, device=device)
    ops.conv2d_fwd(plan, d1, xd, wt, None, z)                        # exact-f32 GEMM: isolates the gather
    y = torch.zeros(N, OH, OW, K, device=device)
    rows = ops.tapconv_stats_rows(d)
    stats = torch.zeros(rows, 2, K, device=device)
    ops.tapconv_gather(plan, d, z, TK, y, stats)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1)).permute(0, 2, 3, 1)
    close(y, ref, tol=1e-5, what="tapconv forward")
    flat = ref.reshape(-1, K)
    close(stats[:, 0].sum(0), flat.sum(0), tol=1e-5, what="tapconv stats sum")
    close(stats[:, 1].sum(0), (flat * flat).sum(0), tol=1e-5, what="tapconv stats sumsq")
    gz = torch.zeros(N * H * W, TK, device=device)
    ops.tapconv_scatter(plan, d, gyd, K, gz, TK)
    dwt = torch.zeros(Cin, TK, device=device)
    ops.conv2d_wgrad(plan, d1, xd, gz, TK, dwt, None)
    dw = torch.zeros(R, S, Cin, K, device=device)
    ops.tapconv_unpack(plan, d, dwt, TK, wd, 0.25, dw)
    torch.cuda.synchronize()
    wr = w.clone().requires_grad_(True)
    out = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1)).permute(0, 2, 3, 1)
    (out * gy).sum().backward()
    close(dw, wr.grad + 0.25 * w, tol=1e-5, what="tapconv weight gradient + decay")
    # split-MFMA GEMM in the middle (what the trunk records)
    wsplit = torch.zeros(int(ops.conv2d_split3_weight_bytes(d1)), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d1, wt, wsplit)
    ops.conv2d_fwd_split3(plan, d1, xd, wsplit, z)
    ops.tapconv_gather(plan, d, z, TK, y, None)
    close(y, ref, tol=1e-4, what="tapconv forward (split MFMA)")
    ops.conv2d_wgrad_split3(plan, d1, xd, gz, TK, dwt, None)
    ops.tapconv_unpack(plan, d, dwt, TK, None, 0.0, dw)
    torch.cuda.synchronize()
    close(dw, wr.grad, tol=1e-4, what="tapconv weight gradient (split MFMA)")
    L = __import__("acimg._lib", fromlist=["load"]).load()
    bad = ops.conv_desc(N, H, W, Cin, K, R, S, 2, "VALID", ldx=Cin, ldy=K, ldw=K)
    assert L.acimg_tapconv_pack(__import__("ctypes").byref(bad), wd.data_ptr(), wt.data_ptr(), TK, None) != 0


def test_stem_row_run_conv_matches_7x7_stride2(device):
    """the stem as a row-run conv (include/acimg.h, acimg_conv2d_fwd_split3): 7x7/2 after 3+3 zero padding on a
    4-channel frame == R=7, S=1, C=32 with pixel pitch 4, incl. the batch-norm partials"""
    from acimg import ops
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 30, 42
    x = rnd(g, N, H, W, 3).abs()
    w = rnd(g, 7, 7, 3, 64) * 0.1
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=2, padding=3).permute(0, 2, 3, 1)
    plan = ops.Plan(device, eager=True)
    Hp, Wp = H + 6, W + 6
    frame = torch.zeros(N * Hp * Wp * 4 + 64, device=device)
    ops.pad_image(plan, dev(x, device), frame, N, H, W, 3, 4, Hp, Wp, 3, 3)
    fr = frame[:N * Hp * Wp * 4].view(N, Hp, Wp, 4).cpu()
    assert torch.equal(fr[:, 3:3 + H, 3:3 + W, :3], x.float()) and float(fr[..., 3].abs().max()) == 0.0
    assert float(fr[:, :3].abs().max()) == 0.0 and float(fr[:, :, Wp - 3:].abs().max()) == 0.0
    wr = torch.zeros(7, 1, 32, 64)
    wr.view(7, 32, 64)[:, :28] = torch.nn.functional.pad(w.float(), (0, 0, 0, 1)).reshape(7, 28, 64)
    d = ops.conv_desc(N, Hp, Wp, 32, 64, 7, 1, 2, "VALID", ldx=4, ldy=64, ldw=64)
    d.OH, d.OW = OH, OW
    wsplit = torch.zeros(int(ops.conv2d_split3_weight_bytes(d)), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, dev(wr, device), wsplit)
    y = torch.zeros(N, OH, OW, 64, device=device)
    rows = ops.conv2d_fwd_split3_stats_rows(d)
    stats = torch.zeros(rows, 2, 64, device=device)
    ops.conv2d_fwd_split3(plan, d, frame, wsplit, y, None, None, 0, stats)
    torch.cuda.synchronize()
    close(y, ref, tol=1e-4, what="row-run stem")
    flat = ref.reshape(-1, 64)
    close(stats[:, 0].sum(0), flat.sum(0), tol=1e-4, what="row-run stem stats sum")
    close(stats[:, 1].sum(0), (flat * flat).sum(0), tol=1e-4, what="row-run stem stats sumsq")
    bad = ops.conv_desc(N, Hp, Wp, 32, 64, 7, 3, 2, "VALID", ldx=4, ldy=64, ldw=64)     # S != 1: not a row run
    L = __import__("acimg._lib", fromlist=["load"]).load()
    assert L.acimg_conv2d_split3_prepare(__import__("ctypes").byref(bad), wr.data_ptr(), wsplit.data_ptr(), None) != 0


def test_split3_prepare_multi_equals_single_launches(device):
    from acimg import ops
    g = torch.Generator().manual_seed(22)
    plan = ops.Plan(device, eager=True)
    jobs = ops.PrepareJobs()
    singles = []
    for (C_, K_, R_, mode) in ((64, 64, 3, 0), (128, 64, 3, 1), (32, 96, 1, 0), (64, 128, 3, 1)):
        d = ops.conv_desc(2, 12, 16, C_, K_, R_, R_, 1, "SAME")
        w = dev(rnd(g, R_, R_, C_, K_) * 0.1, device)
        nbytes = int(ops.conv2d_split3_dgrad_weight_bytes(d) if mode else ops.conv2d_split3_weight_bytes(d))
        a = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        b = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        (ops.conv2d_split3_prepare_dgrad if mode else ops.conv2d_split3_prepare)(plan, d, w, a)
        jobs.add(d, w, b, mode)
        # the multi launch serves the on-the-fly kernels (trainable generator kernels): the row-major planes only, not the
        # LDS-tile-order image the single launch appends for the pre-split trunk kernels
        rowmajor = nbytes if mode else 2 * d.ldw * R_ * R_ * C_ * 2
        singles.append((a, b, rowmajor))
    ops.conv2d_split3_prepare_multi(plan, jobs)
    torch.cuda.synchronize()
    for a, b, rowmajor in singles:
        assert torch.equal(a[:rowmajor], b[:rowmajor])
        assert not b[rowmajor:].any()
    empty = ops.PrepareJobs()
    ops.conv2d_split3_prepare_multi(plan, empty)            # nothing to do is not an error


def test_splitk_handoff_equals_reduce_launch(device):
    """split-K combined inside the kernel (the explicit `tickets` argument: ticket per tile, last arriver adds the K
    ranges in range order) is bit-identical to the separate reduce launch (tickets = NULL, or the splitk_handoff switch
    of acimg_configure off), forward (bias + ReLU epilogue) and data gradient (residual + mask epilogue), and leaves
    the caller's ticket words at zero"""
    import ctypes as C

    from acimg import _lib, ops
    L = _lib.load()
    g = torch.Generator().manual_seed(31)
    N, H, W, Cin, K = 4, 12, 16, 128, 128
    d = ops.conv_desc(N, H, W, Cin, K, 3, 3, 1, "SAME", act=1)
    assert ops.conv2d_fwd_tiling(d)[2] > 1
    x, w, b = dev(rnd(g, N, H, W, Cin), device), dev(rnd(g, 3, 3, Cin, K) * 0.05, device), dev(rnd(g, K), device)
    gy = dev(rnd(g, N, H, W, K), device)
    res = dev(rnd(g, N, H, W, Cin), device)
    st = ops.current_stream_handle(device)
    ws = torch.zeros(int(max(L.acimg_conv2d_fwd_workspace(C.byref(d)), L.acimg_conv2d_dgrad_workspace(C.byref(d)))),
                     dtype=torch.uint8, device=device)
    tick = torch.zeros(ops.TICKET_WORDS, dtype=torch.int32, device=device)
    outs = []
    for mode in ("handoff", "null", "configured-off"):
        if mode == "configured-off":
            _lib.configure(splitk_handoff=0)
        t = None if mode == "null" else tick.data_ptr()
        y = torch.zeros(N, H, W, K, device=device)
        dx = torch.zeros(N, H, W, Cin, device=device)
        _lib.check(L.acimg_conv2d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, 0,
                                      None, ws.data_ptr(), ws.numel(), t, st), "conv2d_fwd")
        _lib.check(L.acimg_conv2d_dgrad(C.byref(d), gy.data_ptr(), K, w.data_ptr(), dx.data_ptr(), Cin, res.data_ptr(),
                                        Cin, x.data_ptr(), Cin, ws.data_ptr(), ws.numel(), t, st), "conv2d_dgrad")
        torch.cuda.synchronize()
        outs.append((y.clone(), dx.clone()))
        assert int(tick.abs().max()) == 0
        _lib.configure()                                  # back to the defaults
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])
    ref = torch.relu(torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().permute(3, 2, 0, 1).cpu(),
                                                 b.double().cpu(), padding=1)).permute(0, 2, 3, 1)
    close(outs[0][0], ref, tol=1e-5, what="split-K forward")
    # the plans' own path: the workspace owns a ticket block and hands it to every split-K call
    plan = ops.Plan(device, eager=True)
    y = torch.zeros(N, H, W, K, device=device)
    ops.conv2d_fwd(plan, d, x, w, b, y)
    torch.cuda.synchronize()
    assert torch.equal(y, outs[0][0]) and int(plan.ws.tickets.abs().max()) == 0
    # misaligned ticket words are refused
    rc = L.acimg_conv2d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, 0, None,
                            ws.data_ptr(), ws.numel(), tick.data_ptr() + 4, st)
    assert rc == -1 and "ticket" in _lib.last_error()
    # acimg_configure validates
    with pytest.raises(_lib.AcimgError):
        _lib.configure(split3_tile_bm=96, split3_tile_bn=96)
    with pytest.raises(_lib.AcimgError):
        _lib.configure(splitk_target=0)


def test_deconv_dgrad_few_channels_direct(device):
    """unet_architecture.py upsample_9 (32 -> 8, 2x2 / 2) at a size that takes the pointwise MFMA kernel (round 4,
    patch2_32x8_kernel: one 32 x 32 product per input pixel, operands loaded from global memory in MFMA layout): forward with
    bias into a wider pixel stride, data gradient with and without a ReLU mask and from a wider gy stride, weight gradient
    (patch2_wgrad_32x8_kernel: the pixels are the MFMA's K axis, one slab per workgroup) and bias gradient; f16x3 forward,
    bf16x3 gradients; two runs, the same bits"""
    from acimg import ops

    N, H, W, Cc, K, R, S, s = 5, 112, 149, 32, 8, 2, 2, 2
    g = torch.Generator().manual_seed(78)
    x = rnd(g, N, H, W, Cc).requires_grad_(True)
    w = (rnd(g, R, S, K, Cc) * 0.1).requires_grad_(True)
    b = rnd(g, K)
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, stride=s).permute(0, 2, 3, 1)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    maskt = rnd(g, N, H, W, Cc)
    d = ops.deconv_desc(N, H, W, Cc, K, R, S, s, ldy=2 * K)
    plan = ops.Plan(device, eager=True)
    wd, xd, bd = dev(w.detach(), device), dev(x.detach(), device), dev(b, device)
    gywide = torch.zeros(N, 2 * H, 2 * W, 2 * K, dtype=torch.float64)
    gywide[..., :K] = gy
    gyd, md = dev(gywide, device), dev(maskt, device)
    outs = []
    for _ in range(2):
        ybuf = torch.zeros(N, 2 * H, 2 * W, 2 * K, device=device)
        ops.deconv_fwd(plan, d, xd, wd, bd, ops.Ptr(ybuf, K))
        dx = torch.empty(N, H, W, Cc, device=device)
        ops.deconv_dgrad(plan, d, gyd, 2 * K, wd, dx)
        dxm = torch.empty(N, H, W, Cc, device=device)
        ops.deconv_dgrad(plan, d, gyd, 2 * K, wd, dxm, md, Cc)
        dw = torch.empty(R, S, K, Cc, device=device)
        db = torch.empty(K, device=device)
        ops.deconv_wgrad(plan, d, xd, gyd, 2 * K, dw, db)
        torch.cuda.synchronize()
        outs.append((ybuf.cpu(), dx.cpu(), dxm.cpu(), dw.cpu(), db.cpu()))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    ybuf, dx, dxm, dw, db = outs[0]
    close(dw, w.grad, tol=2e-5, what="deconv wgrad (pointwise MFMA, pixels along K)")
    close(db, gy.sum((0, 1, 2)), tol=2e-5, what="deconv bias gradient")
    close(ybuf[..., K:], y, tol=2e-6, what="deconv fwd (pointwise MFMA)")
    assert float(ybuf[..., :K].abs().max()) == 0.0
    close(dx, x.grad, tol=3e-5, what="deconv dgrad (pointwise MFMA)")
    close(dxm, x.grad * (maskt > 0), tol=3e-5, what="deconv dgrad with mask (pointwise MFMA)")


@pytest.mark.parametrize("case", [(32, 28, 38, 256, 1024, 1, 1, 1, "SAME"),     # 2128 tiles of 8 K steps: 4.16 rounds
                                  (8, 56, 75, 128, 128, 3, 3, 1, "SAME"),       # 263 tiles < resident slots: every tile is split
                                  (5, 28, 38, 512, 256, 1, 1, 1, "SAME"),       # M = 5320: a row tail inside the last row tile
                                  (32, 56, 75, 64, 256, 1, 1, 1, "SAME"),       # 2 K steps per tile: left whole
                                  (32, 28, 38, 256, 256, 3, 3, 1, "SAME"),      # 532 tiles of 72 K steps: 1.04 rounds
                                  (32, 14, 19, 2048, 512, 1, 1, 1, "SAME")])    # 268 tiles of 64 K steps
def test_trunk_kernel_variants_agree(device, case):
    """The trunk forward conv in its forms — one tile per workgroup, persistent (a workgroup walks a tile list; the
    next tile's first operand stage is requested under the current tile's last K step, output stores drain under the
    next tile's MFMAs: counted vmcnt), each with and without the tail split, persistent with 128-byte operand rows —
    and the ring kernel (software-pipelined K loop, 256- and 128-row tiles, tail split) —
    against fp64 and against each other: the BK = 32 forms give the same bits (same K ranges, same MFMA order, partials
    added in range order whoever arrives last); tickets back at zero; statistics equal to rounding."""
    from acimg import _lib, ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(5 + Cc + K)
    x = torch.rand(N, H, W, Cc, generator=g)
    w = torch.randn(R, S, Cc, K, generator=g) * (2.0 / (R * S * Cc)) ** 0.5
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding)
    rows = N * H * W
    lo_off = plane_bytes(rows, Cc)
    planes = torch.zeros(lo_off * 2, dtype=torch.uint8, device=device)
    plan = ops.Plan(device, eager=True)
    one, zero = torch.ones(Cc, device=device), torch.zeros(Cc, device=device)
    ops.bn_relu_split(plan, x.to(device), one, zero, 1, planes, lo_off, rows, Cc)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, w.to(device), wsplit)
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=device)
    outs = {}
    try:
        for name, cfg in (("one-tile whole", dict(trunk_persistent=0, tail_split=0, trunk_ring=0)),
                          ("persistent whole", dict(trunk_persistent=2, tail_split=0, trunk_ring=0)),
                          ("one-tile", dict(trunk_persistent=0, trunk_ring=0)),
                          ("persistent", dict(trunk_persistent=2, trunk_ring=0)),
                          ("staggered", dict(trunk_persistent=2, trunk_stagger=50, trunk_ring=0)),
                          ("spread", dict(trunk_persistent=2, trunk_dma_pos=1, trunk_ring=0)),
                          ("auto no ring", dict(trunk_ring=0)), ("auto", dict()),
                          ("ring256 whole", dict(trunk_ring=2, trunk_ring_bm=256, tail_split=0)),
                          ("ring128 whole", dict(trunk_ring=2, trunk_ring_bm=128, tail_split=0)),
                          ("ring256", dict(trunk_ring=2, trunk_ring_bm=256)),
                          ("ring128", dict(trunk_ring=2, trunk_ring_bm=128)),
                          ("ring256 s3", dict(trunk_ring=2, trunk_ring_bm=256, tail_s=3)),
                          ("ring", dict(trunk_ring=2)),
                          # the halo kernel takes the 3x3 cases only (the others stay on the shipped choice)
                          ("halo whole", dict(trunk_halo=2, tail_split=0)), ("halo", dict(trunk_halo=2)),
                          ("halo s3", dict(trunk_halo=2, tail_s=3))):
            _lib.configure(**cfg)
            srows = ops.conv2d_fwd_split3p_stats_rows(d)     # depends on the kernel the configuration picks
            y = torch.full((N, d.OH, d.OW, K), float("nan"), device=device)
            st = torch.full((srows, 2, K), float("nan"), device=device)
            for _ in range(2):                        # twice: tickets and stage state must be reusable
                ops.conv2d_fwd_split3p(plan, d, planes, lo_off, wsplit, y, st, tail_ws=tws)
            torch.cuda.synchronize()
            assert int(tws[:4096].view(torch.int32).abs().sum()) == 0, name
            outs[name] = (y, st)
    finally:
        _lib.configure()
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1),
                                     padding=(R // 2, S // 2)).permute(0, 2, 3, 1)
    for name, (y, st) in outs.items():
        close(y, ref, tol=2e-6, what="trunk conv %s %s" % (name, case))
        flat = ref.reshape(-1, K)
        close(st[:, 0].sum(0), flat.sum(0), tol=2e-4, what="stats sum " + name)
        close(st[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="stats sumsq " + name)
    # whole tiles: the same K order in every kernel -> the same bits (the ring kernel keeps the per-accumulator order
    # lo*hi, hi*lo, hi*hi of a K step)
    assert torch.equal(outs["persistent whole"][0], outs["one-tile whole"][0])
    for name in ("ring256 whole", "ring128 whole"):
        assert torch.equal(outs[name][0], outs["one-tile whole"][0]), name
    # the same plan, different timing -> the same bits (partials are added in range order, whoever arrives last)
    # (the shipped choice, "auto", may put a shape on the ring kernel: other tail ranges, not the same bits)
    for name in ("persistent", "staggered", "spread", "auto no ring"):
        assert torch.equal(outs[name][0], outs["one-tile"][0]), name
    for name in ("staggered", "spread"):
        assert torch.equal(outs[name][1], outs["persistent"][1]), name
    # other K ranges in the tail (ring kernel under "auto", forced range counts): equal to rounding
    y0 = outs["one-tile whole"][0]
    for name in outs:
        assert float((outs[name][0] - y0).abs().max()) <= 4e-6 * float(y0.abs().max()), name


@pytest.mark.parametrize("case", [(32, 28, 38, 256, 1024, 0), (8, 56, 75, 64, 256, 0), (7, 28, 38, 128, 512, 0),
                                  (32, 14, 19, 512, 2048, 0), (8, 56, 75, 64, 256, 1), (7, 28, 38, 128, 512, 1)])
def test_two_pass_conv3_equals_conv_then_bn_pass(device, case):
    """conv3 of an identity unit in two passes (acimg_conv2d_fwd_split3p_stats: K loop + batch-norm partials, no output;
    acimg_conv2d_fwd_split3p_tail: the same tiles with relu(acc * scale + shift + shortcut) + split in the epilogue)
    against conv -> raw fp32 -> acimg_bn_add_relu_split on the same kernel form: the same partials and the same plane
    bytes (same K order, same fma / add / max / split sequence); a row tail inside the last row tile; tickets at zero.
    Last case field 1: a PROJECTION shortcut (acimg_conv2d_fwd_split3p_tail_proj: raw fp32 shortcut conv output with its own
    scale / shift) against the projection form of acimg_bn_add_relu_split"""
    from acimg import _lib, ops

    N, H, W, Cc, K, proj = case
    g = torch.Generator().manual_seed(3 + Cc + K)
    x = torch.rand(N, H, W, Cc, generator=g)
    w = torch.randn(1, 1, Cc, K, generator=g) * (2.0 / Cc) ** 0.5
    short = torch.rand(N, H, W, K, generator=g) * 2.0
    scale = (torch.rand(K, generator=g) + 0.5).to(device)
    shift = (torch.rand(K, generator=g) - 0.7).to(device)
    d = ops.conv_desc(N, H, W, Cc, K, 1, 1, 1, "SAME")
    rows = N * H * W
    lo_x, lo_y = plane_bytes(rows, Cc), plane_bytes(rows, K)
    plan = ops.Plan(device, eager=True)
    xp = torch.zeros(lo_x * 2, dtype=torch.uint8, device=device)
    ops.bn_relu_split(plan, x.to(device), torch.ones(Cc, device=device), torch.zeros(Cc, device=device), 1, xp, lo_x, rows, Cc)
    sp = torch.zeros(lo_y * 2, dtype=torch.uint8, device=device)
    ops.bn_relu_split(plan, short.to(device), torch.ones(K, device=device), torch.zeros(K, device=device), 1, sp, lo_y, rows, K)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, w.to(device), wsplit)
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=device)
    try:
        _lib.configure(trunk_persistent=2)          # the reference path on the same kernel form as the two passes
        srows = ops.conv2d_fwd_split3p_stats_rows(d)
        assert srows == -(-rows // 128)
        y = torch.full((N, H, W, K), float("nan"), device=device)
        st_ref = torch.full((srows, 2, K), float("nan"), device=device)
        ops.conv2d_fwd_split3p(plan, d, xp, lo_x, wsplit, y, st_ref, tail_ws=tws)
        ref = torch.zeros(lo_y * 2, dtype=torch.uint8, device=device)
        sc32 = (short - 1.0).to(device)
        sb, tb = (torch.rand(K, generator=g) + 0.5).to(device), (torch.rand(K, generator=g) - 0.5).to(device)
        if proj:
            ops.bn_add_relu_split(plan, y, scale, shift, sc32, sb, tb, None, 0, ref, lo_y, None, N, H, W, K, H, W, 1)
        else:
            ops.bn_add_relu_split(plan, y, scale, shift, None, None, None, sp, lo_y, ref, lo_y, None, N, H, W, K, H, W, 1)
        st = torch.full((srows, 2, K), float("nan"), device=device)
        out = torch.zeros(lo_y * 2, dtype=torch.uint8, device=device)
        for _ in range(2):
            ops.conv2d_fwd_split3p_stats(plan, d, xp, lo_x, wsplit, st, tail_ws=tws)
            if proj:
                ops.conv2d_fwd_split3p_tail_proj(plan, d, xp, lo_x, wsplit, scale, shift, sc32, sb, tb, out, lo_y, tail_ws=tws)
            else:
                ops.conv2d_fwd_split3p_tail(plan, d, xp, lo_x, wsplit, scale, shift, sp, lo_y, out, lo_y, tail_ws=tws)
        torch.cuda.synchronize()
    finally:
        _lib.configure()
    assert int(tws[:4096].view(torch.int32).abs().sum()) == 0
    assert torch.equal(st, st_ref)
    # compare as values first (a clearer message than a byte mismatch), then the bytes
    full = -(-rows // 16) * 16

    def values(buf):
        h = buf[:full * K * 2].view(torch.float16).float()
        lo = buf[lo_y:lo_y + full * K * 2].view(torch.float16).float()
        return (h + lo) * 4.0
    va, vb = values(out), values(ref)
    assert float((va - vb).abs().max()) <= 1e-6 * float(vb.abs().max()), float((va - vb).abs().max())
    assert torch.equal(out[:full * K * 2], ref[:full * K * 2]) and torch.equal(out[lo_y:], ref[lo_y:])
    # and against fp64
    shortcut = ((short.double() - 1.0) * sb.cpu().double() + tb.cpu().double()) if proj else short.double()
    r64 = torch.relu(torch.einsum("nhwc,ck->nhwk", x.double(), w[0, 0].double()) * scale.cpu().double() + shift.cpu().double()
                     + shortcut)
    got = unsplit(out, lo_y, rows, K)
    close(got, r64.reshape(rows, K), tol=2e-6, what="two-pass conv3 vs fp64 %s" % (case,))


@pytest.mark.parametrize("case", [(4, 112, 149, 32), (5, 112, 149, 64), (9, 85, 90, 64), (17, 61, 67, 32), (40, 56, 74, 64)])
@pytest.mark.parametrize("bf16", [True, False])
def test_conv_halo16_fwd_dgrad_match_fp64(device, case, bf16):
    """The halo form of the forward conv and of the data gradient of the 3x3 / stride-1 / SAME layers with 32 or 64 input and
    32 output channels (round 4, conv_halo16_kernel behind acimg_conv2d_fwd_split3 / _bf16 and acimg_conv2d_dgrad_split3 /
    _bf16 from 65536 pixels on): the input tile with its halo staged once as 16-bit planes, the whole weight image in LDS.
    Forward: bias, raw output into a channel slice of a wider buffer, batch-norm partials (one row per workgroup) of conv +
    bias; data gradient: residual + ReLU mask, dx into a slice.  bf16 = one MFMA per product on ROUNDED operands (against
    fp64 of the rounded operands), else f16x3 forward / bf16x3 backward (against fp64 of the unrounded ones).  Heights and
    widths off the 8 x 32 (4 x 32) tile grid; two runs, the same bits."""
    from acimg import ops

    N, H, W, Cc = case
    K = 32
    g = torch.Generator().manual_seed(9 + N + Cc + int(bf16))
    x = torch.randn(N, H, W, Cc, generator=g).requires_grad_(True)
    w = (torch.randn(3, 3, Cc, K, generator=g) * (2.0 / (9 * Cc)) ** 0.5).requires_grad_(True)
    b = (torch.randn(K, generator=g) * 0.1).requires_grad_(True)
    gy = torch.randn(N, H, W, K, generator=g) * 1e-3
    res = torch.randn(N, H, W, Cc, generator=g) * 1e-3
    maskt = torch.randn(N, H, W, Cc, generator=g)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME", ldy=K + 32)
    rows = ops.conv2d_fwd_split3_stats_rows(d)
    assert rows == 256                                   # the halo form's statistics rows: one per workgroup
    plan = ops.Plan(device, eager=True)
    wd = w.detach().to(device)
    wimg = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, wd, wimg, bf16=bf16)
    wt = torch.zeros(ops.conv2d_split3_dgrad_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare_dgrad(plan, d, wd, wt)
    xd, gyd, bd = x.detach().to(device), gy.to(device), b.detach().to(device)
    outs = []
    for _ in range(2):
        ybuf = torch.zeros(N, H, W, K + 32, device=device)
        st = torch.full((rows, 2, K), float("nan"), device=device)
        ops.conv2d_fwd_split3(plan, d, xd, wimg, ops.Ptr(ybuf, 32), stats=st, bias=bd, bf16=bf16)
        dxbuf = torch.zeros(N, H, W, Cc + 16, device=device)
        ops.conv2d_dgrad_split3(plan, d, gyd, K, wt, ops.Ptr(dxbuf, 16), res.to(device), Cc, maskt.to(device), Cc, bf16=bf16,
                                lddx=Cc + 16)
        torch.cuda.synchronize()
        outs.append((ybuf.cpu(), st.cpu(), dxbuf.cpu()))
    for a, bb in zip(outs[0], outs[1]):
        assert torch.equal(a, bb)
    ybuf, st, dxbuf = outs[0]
    assert float(ybuf[..., :32].abs().max()) == 0.0 and float(dxbuf[..., :16].abs().max()) == 0.0
    rnd_ = (lambda t: t.float().to(torch.bfloat16).double()) if bf16 else (lambda t: t.double())
    xr = rnd_(x.detach()).permute(0, 3, 1, 2).requires_grad_(True)
    wr = rnd_(w.detach()).permute(3, 2, 0, 1)
    yr = torch.nn.functional.conv2d(xr, wr, b.detach().double(), padding=1)
    gr = rnd_(gy).permute(0, 3, 1, 2)
    (gx,) = torch.autograd.grad(yr, (xr,), gr)
    yref = yr.detach().permute(0, 2, 3, 1)
    close(ybuf[..., 32:], yref, tol=2e-5 if bf16 else 2e-6, what="halo16 forward %s bf16=%s" % (case, bf16))
    flat = yref.reshape(-1, K)
    close(st[:, 0].sum(0), flat.sum(0), tol=2e-4, what="halo16 stats sum")
    close(st[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="halo16 stats sumsq")
    dref = (gx.permute(0, 2, 3, 1) + res.double()) * (maskt > 0).double()
    close(dxbuf[..., 16:], dref, tol=2e-5 if bf16 else 3e-5, what="halo16 data gradient %s bf16=%s" % (case, bf16))


@pytest.mark.parametrize("case", [(4, 112, 149, 32, 0), (5, 112, 149, 64, 0), (9, 85, 90, 64, 32), (17, 61, 67, 32, 8),
                                  (40, 56, 74, 64, 0)])
@pytest.mark.parametrize("bf16", [True, False])
def test_wgrad_halo16_matches_fp64(device, case, bf16):
    """The halo form of the weight gradient of the 3x3 / stride-1 layers with 32 or 64 input and 32 output channels (round 4,
    wgrad_halo16_kernel behind acimg_conv2d_wgrad_bf16 / _split3 from 65536 pixels on): x and gy tiles staged once as bf16
    planes, nine taps formed from LDS.  bf16 = one MFMA per product on ROUNDED operands (compared with the fp64 gradient of
    the rounded operands), else the three-term bf16 split (fp32-class, against fp64 of the unrounded operands); image
    heights / widths that are not multiples of the 8 x 32 (4 x 32) tile, x as a channel slice of a wider concat buffer,
    gy with a wider pixel stride; weight AND bias gradient; deterministic (two runs, the same bits)"""
    from acimg import ops

    N, H, W, Cc, xoff = case
    K = 32
    g = torch.Generator().manual_seed(5 + N + Cc + int(bf16))
    ldx = Cc + xoff + (8 if xoff else 0)
    xw = torch.randn(N, H, W, ldx, generator=g)
    gyw = torch.randn(N, H, W, K + 16, generator=g) * 1e-2
    x, gy = xw[..., xoff:xoff + Cc], gyw[..., :K]
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME", ldx=ldx)
    plan = ops.Plan(device, eager=True)
    xd, gyd = xw.to(device), gyw.to(device)
    outs = []
    for _ in range(2):
        dw = torch.full((3, 3, Cc, K), float("nan"), device=device)
        db = torch.full((K,), float("nan"), device=device)
        ops.conv2d_wgrad_split3(plan, d, ops.Ptr(xd, xoff), gyd, K + 16, dw, db, bf16=bf16)
        torch.cuda.synchronize()
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    rnd_ = (lambda t: t.float().to(torch.bfloat16).double()) if bf16 else (lambda t: t.double())
    xr = rnd_(x).permute(0, 3, 1, 2)
    gr = rnd_(gy).permute(0, 3, 1, 2)
    wz = torch.zeros(K, Cc, 3, 3, dtype=torch.float64, requires_grad=True)
    yr = torch.nn.functional.conv2d(xr, wz, padding=1)
    (gw,) = torch.autograd.grad(yr, (wz,), gr)
    close(outs[0][0], gw.permute(2, 3, 1, 0), tol=2e-5 if bf16 else 4e-5, what="halo16 wgrad %s bf16=%s" % (case, bf16))
    close(outs[0][1], gr.sum((0, 2, 3)), tol=2e-5 if bf16 else 4e-5, what="halo16 bias gradient %s" % (case,))


@pytest.mark.parametrize("case", [(3, 150, 160, 8, 8), (3, 150, 161, 16, 8), (3, 147, 160, 4, 8), (3, 150, 160, 8, 32),
                                  (3, 150, 160, 16, 16)])
def test_few_channel_wgrad_on_the_halo16_kernel(device, case):
    """The FEW-CHANNEL 3x3 weight gradients (fewer than 32 input channels: the full-resolution layers of the RGB / spectrogram
    U-Nets) reach the fp32 entry acimg_conv2d_wgrad and, from 65536 pixels on, run on the bf16x3 halo kernel over a
    zero-padded 32 x 32 channel tile (round 4): fp32-class results against fp64, bias gradient, pad rows / columns of dw
    untouched, two runs the same bits"""
    from acimg import ops

    N, H, W, Cc, K = case
    g = torch.Generator().manual_seed(3 + Cc + K)
    x = torch.randn(N, H, W, Cc, generator=g)
    gy = torch.randn(N, H, W, K, generator=g) * 1e-2
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME")
    plan = ops.Plan(device, eager=True)
    outs = []
    for _ in range(2):
        dw = torch.full((3, 3, Cc, K), float("nan"), device=device)
        db = torch.full((K,), float("nan"), device=device)
        ops.conv2d_wgrad(plan, d, x.to(device), gy.to(device), K, dw, db)
        torch.cuda.synchronize()
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    wz = torch.zeros(K, Cc, 3, 3, dtype=torch.float64, requires_grad=True)
    yr = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wz, padding=1)
    (gw,) = torch.autograd.grad(yr, (wz,), gy.double().permute(0, 3, 1, 2))
    close(outs[0][0], gw.permute(2, 3, 1, 0), tol=2e-5, what="few-channel halo16 wgrad %s" % (case,))
    close(outs[0][1], gy.double().sum((0, 1, 2)), tol=2e-5, what="few-channel halo16 bias gradient %s" % (case,))


@pytest.mark.parametrize("case", [(3, 150, 160, 8, 8), (3, 147, 161, 8, 32), (3, 150, 161, 16, 8), (3, 151, 170, 8, 16),
                                  (1, 224, 298, 8, 8), (1, 224, 298, 16, 16), (3, 147, 160, 4, 8), (2, 190, 181, 4, 16)])
def test_few_channel_mfma_conv_matches_fp64(device, case):
    """The MFMA form of the few-channel 3x3 / stride-1 / SAME layers (round 4, conv_few16_kernel behind acimg_conv2d_fwd and
    acimg_conv2d_dgrad from 65536 pixels on, 8 or 16 channels convolved, up to 32 written): taps along the GEMM's K axis, the
    tile with its halo staged once as hi / lo planes, 3-term split product (f16 hi/lo forward, bf16 hi/lo backward);
    4 input channels ride the 8-channel image with a zero upper half; the data gradient of the 8 -> 32 layer (32-channel gy,
    8 channels written) takes the 16-row instance of the halo kernel.
    Forward: bias, raw output into a channel slice of a wider buffer, the input a channel slice of a wider buffer,
    batch-norm partials (one row per workgroup) of conv + bias; data gradient: residual, dx into a slice, gy with a wider
    pixel stride.  Heights / widths off the 16 x 32 tile grid; two runs, the same bits."""
    from acimg import ops

    N, H, W, Cc, K = case
    g = torch.Generator().manual_seed(31 + N + Cc + K)
    x = torch.randn(N, H, W, Cc, generator=g)
    w = torch.randn(3, 3, Cc, K, generator=g) * (2.0 / (9 * Cc)) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    gy = torch.randn(N, H, W, K, generator=g) * 1e-3
    res = torch.randn(N, H, W, Cc, generator=g) * 1e-3
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME", ldx=Cc + 8, ldy=K + 4)
    rows = ops.conv2d_stats_rows(d)
    assert rows == 512                                   # one statistics row per workgroup of the MFMA form
    plan = ops.Plan(device, eager=True)
    wd, bd = w.to(device), b.to(device)
    xbuf = torch.full((N, H, W, Cc + 8), 7.0, device=device)
    xbuf[..., 8:] = x.to(device)
    gybuf = torch.full((N, H, W, K + 4), -3.0, device=device)
    gybuf[..., :K] = gy.to(device)
    outs = []
    for _ in range(2):
        ybuf = torch.zeros(N, H, W, K + 4, device=device)
        st = torch.full((rows, 2, K), float("nan"), device=device)
        ops.conv2d_fwd(plan, d, ops.Ptr(xbuf, 8), wd, bd, ops.Ptr(ybuf, 4), stats=st)
        dxbuf = torch.zeros(N, H, W, Cc + 16, device=device)
        ops.conv2d_dgrad(plan, d, gybuf, K + 4, wd, ops.Ptr(dxbuf, 16), res.to(device), Cc, None, 0, lddx=Cc + 16)
        torch.cuda.synchronize()
        outs.append((ybuf.cpu(), st.cpu(), dxbuf.cpu()))
    for a, bb in zip(outs[0], outs[1]):
        assert torch.equal(a, bb)
    ybuf, st, dxbuf = outs[0]
    assert float(ybuf[..., :4].abs().max()) == 0.0 and float(dxbuf[..., :16].abs().max()) == 0.0
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, w.double().permute(3, 2, 0, 1), b.double(), padding=1)
    (gx,) = torch.autograd.grad(yr, (xr,), gy.double().permute(0, 3, 1, 2))
    yref = yr.detach().permute(0, 2, 3, 1)
    close(ybuf[..., 4:], yref, tol=2e-6, what="few-channel MFMA forward %s" % (case,))
    flat = yref.reshape(-1, K)
    close(st[:, 0].sum(0), flat.sum(0), tol=2e-4, what="few-channel MFMA stats sum")
    close(st[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="few-channel MFMA stats sumsq")
    close(dxbuf[..., 16:], gx.permute(0, 2, 3, 1) + res.double(), tol=3e-5, what="few-channel MFMA data gradient %s" % (case,))


@pytest.mark.parametrize("case", [(3, 150, 160, 8, 8, 0), (3, 147, 161, 4, 8, 0), (2, 190, 181, 16, 8, 0), (4, 112, 149, 32, 32, 1),
                                  (4, 112, 149, 32, 32, 2), (4, 113, 150, 64, 32, 1), (4, 113, 150, 64, 32, 2)])
def test_batch_norm_affine_applied_while_staging(device, case):
    """A batch norm + ReLU between two convs without a pass of its own (round 4, include/acimg.h `acimg_conv2d_affine_input_ok`):
    the consumer's forward (few-channel MFMA kernel / halo kernel) and its weight gradient (halo kernel) read the producer's
    RAW output and apply relu(x * scale + shift) while they stage their tiles - zero padding after the affine.  Against the
    same entry points fed the materialised tensor; precision 0 / 1 / 2 = fp32-class / split3 / bf16 entries."""
    from acimg import _lib, ops

    N, H, W, Cc, K, prec = case
    g = torch.Generator().manual_seed(77 + N + Cc + prec)
    raw = torch.randn(N, H, W, Cc, generator=g)
    scale = torch.rand(Cc, generator=g) + 0.5
    shift = torch.randn(Cc, generator=g) * 0.5           # (a shift: the padding ring must stay 0, not relu(shift))
    w = torch.randn(3, 3, Cc, K, generator=g) * (2.0 / (9 * Cc)) ** 0.5
    b = torch.randn(K, generator=g) * 0.1
    gy = torch.randn(N, H, W, K, generator=g) * 1e-3
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME")
    assert ops.conv2d_affine_input_ok(d, prec)
    assert not ops.conv2d_affine_input_ok(ops.conv_desc(2, 60, 60, Cc, K, 3, 3, 1, "SAME"), prec)        # below the size rule
    assert not ops.conv2d_affine_input_ok(ops.conv_desc(N, H, W, Cc, K, 3, 3, 2, "SAME"), prec)          # strided
    plan = ops.Plan(device, eager=True)
    rawd, scd, shd, wd, bd, gyd = (t.to(device) for t in (raw, scale, shift, w, b, gy))
    xmat = torch.relu(rawd * scd + shd)
    bf16 = prec == 2
    rows = ops.conv2d_stats_rows(d) if prec == 0 else ops.conv2d_fwd_split3_stats_rows(d)
    if prec:
        wimg = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
        ops.conv2d_split3_prepare(plan, d, wd, wimg, bf16=bf16)
    outs = []
    for aff in (False, True):
        y = torch.zeros(N, H, W, K, device=device)
        st = torch.zeros(rows, 2, K, device=device)
        dw = torch.zeros(3, 3, Cc, K, device=device)
        db = torch.zeros(K, device=device)
        x = rawd if aff else xmat
        kw = dict(in_scale=scd, in_shift=shd, in_relu=1) if aff else {}
        if prec:
            ops.conv2d_fwd_split3(plan, d, x, wimg, y, stats=st, bias=bd, bf16=bf16, **kw)
        else:
            ops.conv2d_fwd(plan, d, x, wd, bd, y, stats=st, **kw)
        if aff:
            ops.conv2d_wgrad_affine(plan, d, prec, x, scd, shd, gyd, K, dw, db)
        elif prec:
            ops.conv2d_wgrad_split3(plan, d, x, gyd, K, dw, db, bf16=bf16)
        else:
            ops.conv2d_wgrad(plan, d, x, gyd, K, dw, db)
        torch.cuda.synchronize()
        outs.append((y.cpu(), st.sum(0).cpu(), dw.cpu(), db.cpu()))
    tol = 4e-3 if bf16 else 2e-5       # bf16: an fma-vs-two-roundings difference of one ulp flips a bf16 rounding of the operand
    for a, bb, what in zip(outs[0], outs[1], ("forward", "statistics", "weight gradient", "bias gradient")):
        close(bb, a, tol=tol if what != "statistics" else max(tol, 2e-4), what="affine on load, %s %s" % (what, case))
    with pytest.raises(_lib.AcimgError):      # a shape off the halo kernels has no way to apply it: refused, not ignored
        d2 = ops.conv_desc(2, 36, 48, 128, 128, 3, 3, 1, "SAME")
        ops.conv2d_wgrad_affine(plan, d2, 1, torch.zeros(2, 36, 48, 128, device=device), torch.ones(128, device=device),
                                torch.zeros(128, device=device), torch.zeros(2, 36, 48, 128, device=device), 128,
                                torch.zeros(3, 3, 128, 128, device=device), None)


@pytest.mark.parametrize("case", [(134400, 64, 256), (34048, 256, 1024), (20011, 128, 512), (8512, 512, 2048), (37, 64, 100),
                                  (4099, 128, 136), (50, 256, 256)])
def test_gram_statistics_match_fp64(device, case):
    """acimg_gram_stats (round 4, csrc/gram.hip): the batch-norm statistics of a 1x1 conv's output from the column sums and
    the Gram matrix of its INPUT (split planes in brick order), finalised into scale / shift and the moving averages -
    against fp64 statistics of y = x w over the values the planes actually hold, against acimg_bn_finalize fed with those
    fp64 sums (the same finalisation arithmetic), and against the statistics pass it replaces; row counts that are not
    multiples of 16 / 32, one row, K not a multiple of 16; deterministic (two runs, the same bits)"""
    from acimg import ops

    rows, Cc, K = case
    g = torch.Generator().manual_seed(11 + rows % 97 + Cc)
    x = torch.relu(torch.randn(rows, Cc, generator=g) + 0.3) * (0.5 + torch.rand(Cc, generator=g))
    ldw = -(-K // 4) * 4
    w = torch.zeros(Cc, ldw)
    w[:, :K] = torch.randn(Cc, K, generator=g) * (2.6 / Cc) ** 0.5
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.rand(K, generator=g) - 0.5
    mm0, mv0 = torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5
    lo = plane_bytes(rows, Cc)
    plan = ops.Plan(device, eager=True)
    xp = torch.zeros(lo * 2, dtype=torch.uint8, device=device)
    ops.bn_relu_split(plan, x.to(device), torch.ones(Cc, device=device), torch.zeros(Cc, device=device), 1, xp, lo, rows, Cc)
    need = ops.gram_stats_workspace(rows, Cc)
    assert need > 0
    outs = []
    for rep in range(2):
        ws = torch.full((need,), 0xff if rep else 0, dtype=torch.uint8, device=device)      # no dependence on its contents
        sc = torch.full((K,), float("nan"), device=device)
        sh = torch.full((K,), float("nan"), device=device)
        mm, mv = mm0.to(device), mv0.to(device)
        ops.gram_stats(plan, xp, lo, rows, Cc, w.to(device), ldw, K, gamma.to(device), beta.to(device), mm, mv, sc, sh, ws,
                       decay=0.997, eps=1e-5)
        torch.cuda.synchronize()
        outs.append((sc.cpu(), sh.cpu(), mm.cpu(), mv.cpu()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    sc, sh, mm, mv = outs[0]
    # fp64 reference on the values the planes hold (hi + lo: x to 2^-22)
    xv = unsplit(xp, lo, rows, Cc).double().cpu()
    y = xv @ w[:, :K].double()
    mean = y.mean(0)
    var = (y * y).mean(0) - mean * mean
    inv = 1.0 / torch.sqrt(var.float() + 1e-5).double()
    sc_ref = gamma.double() * inv
    sh_ref = beta.double() - mean.float().double() * sc_ref
    # statistics: mean to 1e-6 of the output's scale, variance to 2e-6 relative
    mean_got = (beta.double() - sh.double()) / sc.double()
    var_got = (gamma.double() / sc.double()) ** 2 - 1e-5
    scale_y = float(y.abs().max()) if rows > 1 else 1.0
    assert float((mean_got - mean).abs().max()) <= 2e-6 * scale_y, float((mean_got - mean).abs().max())
    if rows > 1:
        assert float(((var_got - var).abs() / var.clamp_min(1e-12)).max()) <= 4e-6, float(((var_got - var).abs() / var).max())
    close(sc, sc_ref, tol=3e-6, what="gram scale %s" % (case,))
    close(sh, sh_ref, tol=3e-6, what="gram shift %s" % (case,))
    unb = var * (rows / (rows - 1.0)) if rows > 1 else var
    close(mm, 0.997 * mm0.double() + 0.003 * mean, tol=2e-6, what="gram moving mean")
    close(mv, 0.997 * mv0.double() + 0.003 * unb, tol=2e-6, what="gram moving variance")
    # the statistics pass this replaces (same planes, split weights), finalised by acimg_bn_finalize
    if K % 128 == 0 and rows >= 128 * 200:
        d = ops.conv_desc(1, 1, rows, Cc, K, 1, 1, 1, "SAME")
        wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
        ops.conv2d_split3_prepare(plan, d, w[:, :K].reshape(1, 1, Cc, K).contiguous().to(device), wsplit)
        tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=device)
        srows = ops.conv2d_fwd_split3p_stats_rows(d)
        st = torch.zeros(srows, 2, K, device=device)
        ops.conv2d_fwd_split3p_stats(plan, d, xp, lo, wsplit, st, tail_ws=tws)
        sc2, sh2 = torch.zeros(K, device=device), torch.zeros(K, device=device)
        mm2, mv2 = mm0.to(device), mv0.to(device)
        ops.bn_finalize(plan, st, srows, K, K, rows, gamma.to(device), beta.to(device), mm2, mv2, sc2, sh2, 0.997, 1e-5, True)
        torch.cuda.synchronize()
        close(sc, sc2.cpu(), tol=3e-6, what="gram scale vs statistics pass")
        close(sh, sh2.cpu(), tol=3e-6, what="gram shift vs statistics pass")


@pytest.mark.parametrize("case", [(3, 14, 19, 64, 128), (2, 28, 38, 32, 160), (1, 56, 75, 32, 128), (5, 9, 79, 32, 128),
                                  (37, 5, 3, 32, 128), (200, 3, 1, 64, 128), (1, 1, 1, 32, 128), (2, 75, 56, 96, 256)])
def test_halo_kernel_edges(device, case):
    """The halo form of the 3x3 trunk conv (one patch of the planes per channel chunk, taps formed in LDS, zero row for
    taps that fall off an image, permuted fragment rows) at the sizes where its masks matter: several images inside
    one 128-pixel tile, one-pixel-wide and one-pixel images, the widest row an 18-brick patch holds, a row tail, a
    column tail, K ranges; against fp64 and against the per-tap kernel."""
    from acimg import _lib, ops

    N, H, W, Cc, K = case
    g = torch.Generator().manual_seed(11 + H * W + Cc)
    x = torch.rand(N, H, W, Cc, generator=g) - 0.3
    w = torch.randn(3, 3, Cc, K, generator=g) * (2.0 / (9 * Cc)) ** 0.5
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, 1, "SAME")
    rows = N * H * W
    lo_off = plane_bytes(rows, Cc)
    planes = torch.zeros(lo_off * 2, dtype=torch.uint8, device=device)
    plan = ops.Plan(device, eager=True)
    ops.bn_relu_split(plan, x.to(device), torch.ones(Cc, device=device), torch.zeros(Cc, device=device), 0, planes, lo_off,
                      rows, Cc)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, w.to(device), wsplit)
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=device)
    outs = {}
    try:
        for name, cfg in (("per-tap", dict(split3_tile_bm=128, split3_tile_bn=128, trunk_ring=0)),
                          ("halo", dict(split3_tile_bm=128, split3_tile_bn=128, trunk_halo=2)),
                          ("halo whole", dict(split3_tile_bm=128, split3_tile_bn=128, trunk_halo=2, tail_split=0)),
                          ("halo s2", dict(split3_tile_bm=128, split3_tile_bn=128, trunk_halo=2, tail_s=2)),
                          ("halo s4", dict(split3_tile_bm=128, split3_tile_bn=128, trunk_halo=2, tail_s=4))):
            _lib.configure(**cfg)
            tiling = ops.conv2d_fwd_split3_tiling(d)
            assert tiling[2] == (3 if name.startswith("halo") else tiling[2]), (name, tiling)
            srows = ops.conv2d_fwd_split3p_stats_rows(d)
            y = torch.full((N, H, W, K), float("nan"), device=device)
            st = torch.full((srows, 2, K), float("nan"), device=device)
            for _ in range(2):
                ops.conv2d_fwd_split3p(plan, d, planes, lo_off, wsplit, y, st, tail_ws=tws)
            torch.cuda.synchronize()
            assert int(tws[:4096].view(torch.int32).abs().sum()) == 0, name
            outs[name] = (y, st)
    finally:
        _lib.configure()
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1),
                                     padding=1).permute(0, 2, 3, 1)
    flat = ref.reshape(-1, K)
    for name, (y, st) in outs.items():
        close(y, ref, tol=2e-6, what="halo conv %s %s" % (name, case))
        close(st[:, 0].sum(0), flat.sum(0), tol=2e-4, what="stats sum " + name)
        close(st[:, 1].sum(0), (flat * flat).sum(0), tol=2e-4, what="stats sumsq " + name)
    y0 = outs["per-tap"][0]
    for name in outs:       # (chunk, tap) order against (tap, chunk) order: equal to fp32 rounding
        assert float((outs[name][0] - y0).abs().max()) <= 4e-6 * float(y0.abs().max()), name


@pytest.mark.parametrize("case", [(32, 28, 38, 256, 1024, 1, 1, 1, "SAME"), (8, 56, 75, 128, 128, 3, 3, 1, "SAME"),
                                  (6, 56, 75, 256, 64, 1, 1, 1, "SAME")])
def test_fp16_operand_storage_conv(device, case):
    """`acimg_conv2d_fwd_split1p` (BASELINE configs[4]: fp16 operand storage, fp32 accumulation): exactly the
    convolution of the fp16-ROUNDED operands — hi plane of the activations (f16(x * 2^-2)), hi plane of the weights
    (f16(w * 2^10)) — accumulated in fp32; persistent, one-tile and 128x64-tile kernels"""
    from acimg import ops

    N, H, W, Cc, K, R, S, stride, padding = case
    g = torch.Generator().manual_seed(77 + Cc + K)
    x = torch.rand(N, H, W, Cc, generator=g)
    w = torch.randn(R, S, Cc, K, generator=g) * (2.0 / (R * S * Cc)) ** 0.5
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride, padding)
    rows = N * H * W
    lo_off = plane_bytes(rows, Cc)
    planes = torch.zeros(lo_off * 2, dtype=torch.uint8, device=device)
    plan = ops.Plan(device, eager=True)
    ops.bn_relu_split(plan, x.to(device), torch.ones(Cc, device=device), torch.zeros(Cc, device=device), 1, planes,
                      lo_off, rows, Cc)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=device)
    ops.conv2d_split3_prepare(plan, d, w.to(device), wsplit)
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=device)
    y1 = torch.full((N, d.OH, d.OW, K), float("nan"), device=device)
    y3 = torch.full_like(y1, float("nan"))
    st = torch.zeros(ops.conv2d_fwd_split3_stats_rows(d), 2, K, device=device)
    ops.conv2d_fwd_split3p(plan, d, planes, lo_off, wsplit, y1, st, tail_ws=tws, terms=1)
    ops.conv2d_fwd_split3p(plan, d, planes, lo_off, wsplit, y3, None, tail_ws=tws, terms=3)
    torch.cuda.synchronize()
    xq = (x * 0.25).to(torch.float16).double() * 4.0
    wq = (w * 1024.0).to(torch.float16).double() / 1024.0
    conv = lambda a, b: torch.nn.functional.conv2d(a.permute(0, 3, 1, 2), b.permute(3, 2, 0, 1),  # noqa: E731
                                                   padding=(R // 2, S // 2)).permute(0, 2, 3, 1)
    ref1, ref3 = conv(xq, wq), conv(x.double(), w.double())
    close(y1, ref1, tol=2e-6, what="fp16-operand conv vs conv of rounded operands %s" % (case,))
    close(y3, ref3, tol=2e-6, what="split conv %s" % (case,))
    cost = float((ref1 - ref3).abs().max() / ref3.abs().max())
    assert 1e-5 < cost < 5e-3, cost                  # what 11-bit operands cost on one layer
    close(st[:, 0].sum(0), ref1.reshape(-1, K).sum(0), tol=2e-4, what="statistics of the fp16-operand conv")


def test_pipeline_lanes_are_measured(device):
    """`ops.concurrent_streams`: the streams handed to the pipeline really run beside the caller's stream and beside
    each other (a small kernel on one completes while the other is busy) — the runtime maps streams onto a few hardware
    queues in an order the program cannot see, and two streams on one queue serialise"""
    from acimg import ops

    cur = torch.cuda.current_stream(device)
    lanes = ops.concurrent_streams(device, cur, 3)
    assert 1 <= len(lanes) <= 3
    assert len(set(s.cuda_stream for s in lanes + [cur])) == len(lanes) + 1
    work = torch.zeros(64, device=device)        # scratch of the library's own spin / zero kernels (no vendor GEMM)
    for s in lanes:
        assert ops._runs_beside(cur, s, work) and ops._runs_beside(s, cur, work)
    # a stream never runs beside itself
    assert not ops._runs_beside(cur, cur, work)
    ops.set_side_lane(device, lanes[0], lanes[0])          # "no second lane" is a valid side lane
    with torch.cuda.stream(lanes[0]):
        assert ops.side_lane(device) == lanes[0]


@pytest.mark.parametrize("case", [(32, 28416, 300), (5, 4096, 152), (64, 8192, 300), (17, 4100, 16)])
def test_skinny_dense_gradients(device, case):
    """the dense layers over a few batch rows (the 28 416 -> 300 VAE heads, models/unet_acresnet.py:73-76) take the
    skinny kernels inside acimg_conv2d_fwd / _dgrad / _wgrad (csrc/skinny_kernel.hpp: the weight matrix read / written
    once, exact-f32 MFMA): forward with bias (K slabs combined in slab order), data gradient with residual and ReLU mask,
    weight and bias gradient vs fp64"""
    from acimg import ops

    M, Cc, K = case
    g = torch.Generator().manual_seed(9 + M + K)
    x = torch.rand(M, Cc, generator=g) - 0.3
    w = torch.randn(Cc, K, generator=g) * 0.05
    gy = torch.randn(M, K, generator=g)
    res = torch.randn(M, Cc, generator=g)
    d = ops.conv_desc(M, 1, 1, Cc, K, 1, 1, 1, "VALID", ldx=Cc, ldy=K, ldw=K)
    plan = ops.Plan(device, eager=True)
    dx = torch.full((M, Cc), float("nan"), device=device)
    ops.conv2d_dgrad(plan, d, gy.to(device), K, w.to(device), dx, res.to(device), Cc, x.to(device), Cc)
    dw = torch.full((Cc, K), float("nan"), device=device)
    db = torch.full((K,), float("nan"), device=device)
    ops.conv2d_wgrad(plan, d, x.to(device), gy.to(device), K, dw, db)
    dx2 = torch.full((M, Cc), float("nan"), device=device)
    ops.conv2d_dgrad(plan, d, gy.to(device), K, w.to(device), dx2)          # no residual, no mask
    bias = torch.randn(K, generator=g)
    y = torch.full((M, K), float("nan"), device=device)
    ops.conv2d_fwd(plan, d, x.to(device), w.to(device), bias.to(device), y)
    torch.cuda.synchronize()
    close(y, x.double() @ w.double() + bias.double(), tol=2e-6, what="skinny forward %s" % (case,))
    ref_dx = gy.double() @ w.double().t()
    close(dx2, ref_dx, tol=2e-6, what="skinny dgrad %s" % (case,))
    close(dx, (ref_dx + res.double()) * (x > 0).double(), tol=2e-6, what="skinny dgrad + residual + mask %s" % (case,))
    close(dw, x.double().t() @ gy.double(), tol=2e-6, what="skinny wgrad %s" % (case,))
    close(db, gy.double().sum(0), tol=2e-6, what="skinny bias gradient %s" % (case,))
    # a bias-gradient buffer that is 4- but not 16-byte aligned (C ABI: any float*): the dispatch must fall back to
    # the general kernel instead of issuing misaligned 16-byte stores
    dbb = torch.full((K + 1,), float("nan"), device=device)
    dw3 = torch.full((Cc, K), float("nan"), device=device)
    ops.conv2d_wgrad(plan, d, x.to(device), gy.to(device), K, dw3, dbb[1:])
    torch.cuda.synchronize()
    close(dw3, x.double().t() @ gy.double(), tol=2e-6, what="wgrad, offset db %s" % (case,))
    close(dbb[1:], gy.double().sum(0), tol=2e-6, what="bias gradient at a 4-byte offset %s" % (case,))
