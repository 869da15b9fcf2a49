"""Per-shape table of the 22 distinct trunk convolutions (batch 32, random operands) under kernel variants, timed in ONE
process in interleaved rounds (cdna_hip_programming.md rule 24), each against max(x3-MFMA bound, HBM bound at 6.3 TB/s):

    python tools/trunk_shapes.py [rounds] [variant=field:val,field:val ...]

variants are AcimgConfig overrides (default: one tile per workgroup, always persistent, and the shipped per-layer choice);
`lib:<path>` in a variant runs it on another build of the library (tools/build_ref_lib.sh), same process, same rounds.
Also checks that every variant produces the same output (max |dy| / max |y|) and statistics as the first one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch  # noqa: E402

from acimg import _lib, ops  # noqa: E402

SHAPES = [(56, 75, 64, 64, 1, 1, 3), (56, 75, 64, 64, 3, 1, 3), (56, 75, 64, 256, 1, 1, 4), (56, 75, 256, 64, 1, 1, 2),
          (56, 75, 256, 128, 1, 1, 1), (56, 75, 128, 128, 3, 1, 3), (56, 75, 128, 512, 1, 1, 4), (56, 75, 256, 512, 1, 1, 1),
          (56, 75, 512, 128, 1, 1, 3), (56, 75, 128, 128, 3, 2, 1), (28, 38, 128, 512, 1, 1, 1), (28, 38, 512, 256, 1, 1, 1),
          (28, 38, 256, 256, 3, 1, 5), (28, 38, 256, 1024, 1, 1, 6), (28, 38, 512, 1024, 1, 1, 1), (28, 38, 1024, 256, 1, 1, 5),
          (28, 38, 256, 256, 3, 2, 1), (14, 19, 256, 1024, 1, 1, 1), (14, 19, 1024, 512, 1, 1, 1), (14, 19, 512, 512, 3, 1, 3),
          (14, 19, 512, 2048, 1, 1, 3), (14, 19, 1024, 2048, 1, 1, 1), (14, 19, 2048, 512, 1, 1, 2)]
PEAK_X3 = 2500e12 / 3          # algorithmic FLOP/s when every product costs three fp16 MFMAs
HBM = 6.3e12                   # measured copy bandwidth (MI355X_MICROARCH.md)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    specs = [a for a in sys.argv[1:] if "=" in a] or ["one-tile=trunk_persistent:0", "persistent=trunk_persistent:2",
                                                      "auto=trunk_persistent:1"]
    variants = []
    for sp in specs:
        name, kv = sp.split("=", 1)
        kvs = dict(x.split(":", 1) for x in kv.split(",") if x)
        libpath = kvs.pop("lib", None)
        variants.append((name, {k: int(v) for k, v in kvs.items()}, libpath))
    dev = torch.device("cuda:0")
    default_lib = _lib.load()
    handles = {None: default_lib}
    for _, _, path in variants:
        if path not in handles:
            _lib._lib, _lib.LIB_PATH = None, os.path.join(ROOT, path)
            newer = {k: _lib.PROTOTYPES.pop(k) for k in ("acimg_split_plane_bytes",)}    # entry points an older build lacks
            handles[path] = _lib.load()
            _lib.PROTOTYPES.update(newer)
    _lib._lib = default_lib

    def use(cfg, path):
        _lib._lib = handles[path]
        _lib.configure(**cfg)
    N = int(os.environ.get("TRUNK_BATCH", "32"))
    g = torch.Generator(device="cpu").manual_seed(1)
    rows_out = []
    tot = {n: 0.0 for n, _, _ in variants}
    tot_bound = 0.0
    print("%-26s %2s %5s" % ("shape", "n", "tiles") + "".join(" %11s" % n for n, _, _ in variants) +
          "   bound us (mfma / hbm)   best/bound   max|dy|/max|y|")
    for (H, W, C, K, R, s, cnt) in SHAPES:
        d = ops.conv_desc(N, H, W, C, K, R, R, s, "SAME" if s == 1 else (1 if R == 3 else "SAME"))
        rows = N * H * W
        lo = -(-rows // 16) * 16 * C * 2     # acimg_split_plane_bytes: whole 16-pixel bricks
        x = torch.rand(rows, C, generator=g).to(dev)
        planes = torch.zeros(2 * lo, dtype=torch.uint8, device=dev)
        one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        plan = ops.Plan(dev, eager=True)
        ops.bn_relu_split(plan, x, one, zero, 1, planes, lo, rows, C)
        w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
        wsplit = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
        ops.conv2d_split3_prepare(plan, d, w, wsplit)
        nrow = -(-rows // 64)       # enough statistics rows for every variant's row tile
        tws = torch.zeros(ops.conv2d_fwd_split3p_workspace(d), dtype=torch.uint8, device=dev)
        outs, times = [], {n: [] for n, _, _ in variants}
        for name, cfg, path in variants:
            use(cfg, path)
            y = torch.full((N, d.OH, d.OW, K), float("nan"), device=dev)
            st = torch.zeros(nrow * 2 * K, device=dev)      # rows a variant does not write stay zero
            for _ in range(2):
                ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
            torch.cuda.synchronize()
            outs.append((y, st))
        for _ in range(rounds):
            for (name, cfg, path), (y, st) in zip(variants, outs):
                use(cfg, path)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.conv2d_fwd_split3p(plan, d, planes, lo, wsplit, y, st, tail_ws=tws)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
        for path in handles:
            use({}, path)
        _lib._lib = default_lib
        assert int(tws[:4096].view(torch.int32).abs().max()) == 0, "tickets not back at zero"
        y0, s0 = outs[0]
        err = max(float((y - y0).abs().max() / y0.abs().max()) for y, _ in outs[1:]) if len(outs) > 1 else 0.0
        tot_s = [s_.view(nrow, 2, K).sum(0) for _, s_ in outs]       # row tiles differ between kernels: column totals
        serr = max(float((s_ - tot_s[0]).abs().max() / tot_s[0].abs().max()) for s_ in tot_s[1:]) if len(outs) > 1 else 0.0
        assert not torch.isnan(y0).any()
        fl = 2.0 * N * d.OH * d.OW * K * R * R * C
        byt = 4.0 * (N * H * W * C + K * R * R * C + N * d.OH * d.OW * K)
        b_m, b_h = fl / PEAK_X3 * 1e6, byt / HBM * 1e6
        med = {n: sorted(times[n])[len(times[n]) // 4] for n, _, _ in variants}      # lower quartile of the rounds
        best = min(med.values())
        tiles = -(-N * d.OH * d.OW // 128) * -(-K // 128)
        print("%-26s %2d %5d" % ("%dx%d %d->%d %dx%d/%d" % (H, W, C, K, R, R, s), cnt, tiles) +
              "".join(" %11.1f" % med[n] for n, _, _ in variants) +
              "   %6.1f (%5.1f / %5.1f)   %9.2f   %.1e (stats %.1e)" % (max(b_m, b_h), b_m, b_h, best / max(b_m, b_h), err, serr))
        for n, _, _ in variants:
            tot[n] += cnt * med[n]
        tot_bound += cnt * max(b_m, b_h)
        rows_out.append(dict(shape=[H, W, C, K, R, s], count=cnt, us=med, bound_mfma=b_m, bound_hbm=b_h, err=err))
    print("%-35s" % "trunk total (us, weighted by count)" + "".join(" %11.1f" % tot[n] for n, _, _ in variants) +
          "   %6.1f" % tot_bound)
    print(json.dumps({"rows": rows_out, "total": tot, "bound": tot_bound}), file=sys.stderr)


if __name__ == "__main__":
    main()
