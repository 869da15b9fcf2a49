"""HBM traffic per kernel from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; KB per
dispatch).  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-B requests of
16-B/lane streams are tallied at 64 B).
usage: python tools/pmc_summary.py <dir with fetch/ and write/> [steps the profiled command ran, warm-up included] [commit]
       [description of the profiled command]"""
import csv, os, sys, collections
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def load(sub, counter):
    f = os.path.join(root, sub, "run_counter_collection.csv")
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("acimg::", "").split("(")[0]
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def head_commit():
    import subprocess
    try:
        return subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return os.environ.get("ACIMG_COMMIT", "unknown")


rd, wr = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
rows = []
for k, (n, kb) in rd.items():
    wn, wkb = wr.get(k, [n, 0.0])
    rmb = 2.0 * kb / n / 1e3            # KB -> MB, gfx950 correction x2
    wmb = wkb / max(wn, 1) / 1e3
    rows.append((k, n / steps, rmb, wmb, (rmb + wmb) * n / steps / 1e3))
rows.sort(key=lambda r: -r[4])
print("commit %s   (tree the PMC passes ran on; bench.py reads this file: load_traffic_profile)" % (sys.argv[3] if len(sys.argv) > 3 else head_commit()))
desc = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
print("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- %s (%d steps)" % (desc, steps))
print("FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of 16-B/lane streams at 64 B); "
      "MB per launch, averaged")
print("%-62s %10s %14s %15s %12s" % ("kernel", "calls/step", "read MB/launch", "write MB/launch", "GB/step"))
tot = setup = 0.0
# dispatches that are not part of a step: torch's fills (the arenas' zeros at construction; nothing of torch's runs inside
# a recorded step on one GPU) and the one-time LDS-tile-order / split images of the frozen trunk kernels
ONE_TIME = ("at::native::", "split3_brick_kernel", "split3_prepare_kernel")
for k, c, r, w, g in rows:
    tot += g
    if k.startswith(ONE_TIME):
        setup += g
    if g >= 0.01:
        print("%-62s %10.1f %14.1f %15.1f %12.2f%s" % (k[:62], c, r, w, g, "   (set-up)" if k.startswith(ONE_TIME) else ""))
print("total %.2f GB/step" % tot)
print("of which set-up dispatches of the profiled process (divided by the steps like everything else) %.2f GB; "
      "steady-state step %.2f GB" % (setup, tot - setup))
