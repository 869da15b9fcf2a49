"""Where the persistent trunk kernel's waves spend their cycles (diagnostic library with in-kernel s_memtime stamps:
tools/build_stamp.sh -> tools/debug/libacimg_stamp.so).  Per shape and K-step depth: shader cycles per wave, split
into: waiting for the own DMA pieces | step barrier | DMA issue | fragment reads (incl. their latency) | MFMA block
issue | end-of-tile barrier | epilogue | other.     python tools/stamp_probe.py
    python tools/stamp_probe.py ring[=128|256] [H,W,C,K,R,s ...]   the ring kernel (round 3): wait dma = end-of-step vmcnt wait,
    barrier = step barrier, mfma issue = the whole issue block of a step incl. the completion of its fragment reads,
    tile barrier = K-range hand-off, epi rest = epilogue"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch  # noqa: E402

from acimg import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "debug", "libacimg_stamp.so")
from acimg import ops  # noqa: E402

SHAPES = [(28, 38, 256, 1024, 1, 1), (28, 38, 256, 256, 3, 1), (56, 75, 128, 512, 1, 1)]
NAMES = ["wait dma", "barrier", "dma issue", "frag reads", "mfma issue", "tile barrier", "epi rest", "other", "epi: lds wr issue",
         "wr done", "barrier", "row rd + stores", "stats rd", "barrier"]


def main():
    ring = [a for a in sys.argv[1:] if a.startswith("ring")]
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if a.count(",") == 5] or SHAPES
    dev = torch.device("cuda:0")
    L = _lib.load()
    L.acimg_debug_stamp_buffer.restype = C.c_int
    L.acimg_debug_stamp_buffer.argtypes = [C.c_void_p]
    N = int(os.environ.get("TRUNK_BATCH", "32"))
    g = torch.Generator(device="cpu").manual_seed(1)
    dbg = torch.zeros(512 * 8 * 16, dtype=torch.int32, device=dev)
    L.acimg_debug_stamp_buffer(dbg.data_ptr())
    for (H, W, Cc, K, R, s) in shapes:
        d = ops.conv_desc(N, H, W, Cc, K, R, R, s, "SAME")
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
        extra = [(32, int(b)) for b in os.environ.get("STAMP_BITS", "").split(",") if b]
        for bk, nostore in [(32, 0), (32, 1)] + extra:
            L.acimg_debug_no_output_stores(nostore)
            if ring:
                rbm = int(ring[0].split("=")[1]) if "=" in ring[0] else 256
                _lib.configure(trunk_ring=2, trunk_ring_bm=rbm, tail_split=0)
            else:
                _lib.configure(trunk_persistent=2, trunk_bk=bk, tail_split=0)
            for _ in range(3):
                ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
            torch.cuda.synchronize()
            dbg.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
            e1.record()
            torch.cuda.synchronize()
            v = dbg.view(-1, 16).cpu().numpy().astype("int64") & 0xFFFFFFFF
            v = v[v[:, 14] > 0]
            tot = v[:, 14].mean()
            ksteps = R * R * Cc // bk
            tiles = -(-rows // (rbm if ring else 128)) * -(-K // 128)
            nwg = v.shape[0] // 8
            per_wave_steps = ksteps * tiles / nwg
            print("%s%dx%d %d->%d %dx%d  BK=%d  %d workgroups, %.1f tiles each, %d K steps/tile: kernel %.1f us, %.0f cycles/wave "
                  "(%.2f GHz), %.0f cycles per K step" % (("[ablation bits %d] " % nostore) if nostore else "", H, W, Cc, K, R, R, bk, nwg, tiles / nwg, ksteps,
                                                           e0.elapsed_time(e1) * 1e3, tot,
                                                           tot / (e0.elapsed_time(e1) * 1e-3) / 1e9 * 1e-0 / 1e0 / 1e0 * 1e0 / 1e0 if False else tot / (e0.elapsed_time(e1) * 1e3) / 1e3,
                                                           tot / per_wave_steps))
            mf = (24 if bk == 32 else 48) * 16
            if ring:
                mf = (48 if rbm == 256 else 24) * 16
            print("    " + "  ".join("%s %4.1f%%" % (n, 100.0 * v[:, i].mean() / tot) for i, n in enumerate(NAMES)) +
                  "   | MFMA pipe floor per step (2 or 4 waves/SIMD share it): %d cycles per wave" % mf)
    _lib.configure()


if __name__ == "__main__":
    main()
