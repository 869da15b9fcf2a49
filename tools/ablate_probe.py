"""What bounds a K step of the persistent trunk kernel: the same launch with parts of the work switched off
(diagnostic library tools/debug/libacimg_ablate.so from tools/build_stamp.sh; results are garbage, only time is read).

    python tools/ablate_probe.py [H,W,C,K,R ...]

    python tools/ablate_probe.py ring[=128|256] [H,W,C,K,R ...]      the ring kernel (round 3) instead

bits: 1 no output stores | 2 no activation-tile requests | 4 no weight-tile requests | 8 only the hi x hi MFMA sweep
(1/3 of the matrix work, all of the operand traffic) | 16 (ring kernel only) no fragment reads.  If the K loop is bound by operand fetch, dropping requests
shortens it in proportion and dropping MFMAs does not; if by the MFMA pipe, the other way round."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch  # noqa: E402

from acimg import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "debug", "libacimg_ablate.so")
from acimg import ops  # noqa: E402

SHAPES = [(28, 38, 256, 256, 3), (56, 75, 128, 128, 3), (28, 38, 1024, 256, 1), (28, 38, 256, 1024, 1), (56, 75, 128, 512, 1)]
RING_CASES = [(0, "full"), (1, "no stores"), (6, "no requests"), (16, "no fragment reads"), (8, "1/3 MFMA"),
              (22, "no requests, no fragment reads"), (14, "no requests, 1/3 MFMA"), (24, "no fragment reads, 1/3 MFMA"),
              (23, "MFMA + barriers + scalar work only (no requests, reads, stores)"), (31, "barriers + 1/3 MFMA only")]
CASES = [(0, "full"), (1, "no stores"), (2, "no A requests"), (4, "no B requests"), (6, "no requests"), (8, "1/3 MFMA"),
         (14, "no requests, 1/3 MFMA"), (9, "1/3 MFMA, no stores"), (15, "barriers + frag reads + 1/3 MFMA only")]


def main():
    dev = torch.device("cuda:0")
    L = _lib.load()
    L.acimg_debug_no_output_stores.restype = C.c_int
    L.acimg_debug_no_output_stores.argtypes = [C.c_int]
    N = int(os.environ.get("TRUNK_BATCH", "32"))
    g = torch.Generator(device="cpu").manual_seed(1)
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if a.count(",") == 4] or SHAPES
    ring = [a for a in sys.argv[1:] if a.startswith("ring")]
    cases = RING_CASES if ring else CASES
    cfg = dict(trunk_persistent=2)
    if ring:
        cfg = dict(trunk_ring=2, trunk_ring_bm=int(ring[0].split("=")[1]) if "=" in ring[0] else 0)
    rounds = 9
    for (H, W, Cc, K, R) in shapes:
        d = ops.conv_desc(N, H, W, Cc, K, R, R, 1, "SAME")
        rows = N * H * W
        lo = -(-rows // 16) * 16 * Cc * 2     # acimg_split_plane_bytes: whole 16-pixel bricks
        x = torch.rand(rows, Cc, generator=g).to(dev)
        planes = torch.zeros(2 * lo, dtype=torch.uint8, device=dev)
        plan = ops.Plan(dev, eager=True)
        ops.bn_relu_split(plan, x, torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), 1, planes, lo, rows, Cc)
        w = (torch.randn(R, R, Cc, K, generator=g) * 0.05).to(dev)
        wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
        ops.conv2d_split3_prepare(plan, d, w, wsplit)
        y = torch.empty(N, d.OH, d.OW, K, device=dev)
        st = torch.zeros(-(-rows // 64) * 2 * K, device=dev)
        tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=dev)
        _lib.configure(**cfg)
        times = {b: [] for b, _ in cases}
        for r in range(rounds + 1):
            for bits, _ in cases:            # interleaved: every case sees the same clock / thermal state
                L.acimg_debug_no_output_stores(bits)
                ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[bits].append(e0.elapsed_time(e1) * 1e3 / 3)
        ksteps = R * R * Cc // 32
        tiles = -(-rows // 128) * -(-K // 128)
        full = sorted(times[0])[len(times[0]) // 4]
        print("%dx%d %d->%d %dx%d: %d tiles x %d K steps" % (H, W, Cc, K, R, R, tiles, ksteps))
        for bits, name in cases:
            t = sorted(times[bits])[len(times[bits]) // 4]
            print("    %-40s %7.1f us  %5.2f x full" % (name, t, t / full))
    L.acimg_debug_no_output_stores(0)
    _lib.configure()


if __name__ == "__main__":
    main()
