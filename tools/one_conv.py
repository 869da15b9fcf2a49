"""One trunk conv shape under one configuration, a few launches: the target of rocprofv3 --pmc passes.
    python tools/one_conv.py H,W,C,K,R,stride [field:val,...] [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch  # noqa: E402

from acimg import _lib, ops  # noqa: E402


def main():
    H, W, C, K, R, s = (int(v) for v in sys.argv[1].split(","))
    cfg = dict((k, int(v)) for k, v in (x.split(":") for x in sys.argv[2].split(",") if x)) if len(sys.argv) > 2 else {}
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    N = int(os.environ.get("TRUNK_BATCH", "32"))
    dev = torch.device("cuda:0")
    _lib.load()
    g = torch.Generator(device="cpu").manual_seed(1)
    d = ops.conv_desc(N, H, W, C, K, R, R, s, "SAME" if s == 1 else (1 if R == 3 else "SAME"))
    rows = N * H * W
    lo = -(-rows // 16) * 16 * C * 2     # acimg_split_plane_bytes: whole 16-pixel bricks
    x = torch.rand(rows, C, generator=g).to(dev)
    planes = torch.zeros(2 * lo, dtype=torch.uint8, device=dev)
    plan = ops.Plan(dev, eager=True)
    ops.bn_relu_split(plan, x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1, planes, lo, rows, C)
    w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
    wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
    ops.conv2d_split3_prepare(plan, d, w, wsplit)
    y = torch.empty(N, d.OH, d.OW, K, device=dev)
    st = torch.zeros(-(-rows // 64) * 2 * K, device=dev)
    tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=dev)
    _lib.configure(**cfg)
    for _ in range(n):
        ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
    torch.cuda.synchronize()
    _lib.configure()
    print("ok", float(y.abs().max()))


if __name__ == "__main__":
    main()
