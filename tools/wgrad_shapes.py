"""Per-shape table of the generator's split-MFMA weight / data gradient and forward convolutions (batch 32), timed in ONE
process in interleaved rounds; variants are builds of the library (tools/build_ref_lib.sh) and/or AcimgConfig overrides:

    python tools/wgrad_shapes.py [rounds] name=[lib:<path>][,field:val ...] ...

Checks every variant against the first one (max |d| / max |ref|)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "acoustic-image-generation_amd"))
import torch  # noqa: E402

from acimg import _lib, ops  # noqa: E402

# (H, W, C, K, R, launches per step) of conv2d_wgrad_split3 / conv2d_dgrad_split3 / conv2d_fwd_split3 in the TrainerMask step
SHAPES = [(36, 48, 256, 128, 3, 1), (36, 48, 128, 128, 3, 2), (36, 48, 128, 64, 3, 1), (36, 48, 64, 64, 3, 1),
          (12, 16, 128, 128, 3, 3), (14, 19, 2048, 144, 1, 1)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 7
    specs = [a for a in sys.argv[1:] if "=" in a] or ["shipped="]
    variants = []
    for sp in specs:
        name, kv = sp.split("=", 1)
        kvs = dict(x.split(":", 1) for x in kv.split(",") if x)
        path = kvs.pop("lib", None)
        variants.append((name, {k: int(v) for k, v in kvs.items()}, path))
    dev = torch.device("cuda:0")
    default_lib = _lib.load()
    handles = {None: default_lib}
    for _, _, path in variants:
        if path not in handles:
            _lib._lib, _lib.LIB_PATH = None, os.path.join(ROOT, path)
            handles[path] = _lib.load()
    _lib._lib = default_lib

    def use(cfg, path):
        _lib._lib = handles[path]
        _lib.configure(**cfg)

    N = 32
    g = torch.Generator().manual_seed(3)
    tot = {(op, n): 0.0 for op in ("wgrad", "dgrad", "fwd") for n, _, _ in variants}
    print("%-24s %2s %-6s" % ("shape", "n", "op") + "".join(" %11s" % n for n, _, _ in variants) + "   bound us (x3 MFMA)   max|d|/max|ref|")
    for (H, W, C, K, R, cnt) in SHAPES:
        d = ops.conv_desc(N, H, W, C, K, R, R, 1, "SAME")
        x = torch.randn(N, H, W, C, generator=g).to(dev)
        gy = (torch.randn(N, H, W, K, generator=g) * 1e-3).to(dev)
        w = (torch.randn(R, R, C, K, generator=g) * 0.05).to(dev)
        fl = 2.0 * N * H * W * K * R * R * C
        bound = fl / (2500e12 / 3) * 1e6
        plan = ops.Plan(dev, eager=True)
        for op in ("wgrad", "dgrad", "fwd"):
            if op != "wgrad" and (K % 32 or C % 32):
                continue
            outs, times = [], {n: [] for n, _, _ in variants}

            def run(out):
                if op == "wgrad":
                    ops.conv2d_wgrad_split3(plan, d, x, gy, K, out[0], out[1])
                elif op == "dgrad":
                    ops.conv2d_dgrad_split3(plan, d, gy, K, out[2], out[0])
                else:
                    ops.conv2d_fwd_split3(plan, d, x, out[2], out[0], stats=out[1])

            for name, cfg, path in variants:
                use(cfg, path)
                if op == "wgrad":
                    out = (torch.zeros(R, R, C, K, device=dev), torch.zeros(K, device=dev), None)
                elif op == "dgrad":
                    wt = torch.zeros(ops.conv2d_split3_dgrad_weight_bytes(d), dtype=torch.uint8, device=dev)
                    ops.conv2d_split3_prepare_dgrad(plan, d, w, wt)
                    out = (torch.zeros(N, H, W, C, device=dev), None, wt)
                else:
                    wi = torch.zeros(ops.conv2d_split3_weight_bytes(d), dtype=torch.uint8, device=dev)
                    ops.conv2d_split3_prepare(plan, d, w, wi)
                    out = (torch.zeros(N, H, W, K, device=dev),
                           torch.zeros(ops.conv2d_fwd_split3_stats_rows(d), 2, K, device=dev), wi)
                run(out)
                run(out)
                torch.cuda.synchronize()
                outs.append(out)
            for _ in range(rounds):
                for (name, cfg, path), out in zip(variants, outs):
                    use(cfg, path)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        run(out)
                    e1.record()
                    torch.cuda.synchronize()
                    times[name].append(e0.elapsed_time(e1) / 4 * 1e3)
            ref = outs[0][0]
            err = max(float((o[0] - ref).abs().max() / ref.abs().max()) for o in outs[1:]) if len(outs) > 1 else 0.0
            med = {n: sorted(times[n])[len(times[n]) // 4] for n, _, _ in variants}
            print("%-24s %2d %-6s" % ("%dx%d %d->%d %dx%d" % (H, W, C, K, R, R), cnt, op) +
                  "".join(" %11.1f" % med[n] for n, _, _ in variants) + "   %8.1f   %.1e" % (bound, err))
            for n, _, _ in variants:
                tot[(op, n)] += cnt * med[n]
    for path in handles:
        use({}, path)
    _lib._lib = default_lib
    for op in ("wgrad", "dgrad", "fwd"):
        print("%-34s" % ("total %s (us, weighted)" % op) + "".join(" %11.1f" % tot[(op, n)] for n, _, _ in variants))


if __name__ == "__main__":
    main()
