"""Counters of the trunk kernel per launch from the rocprofv3 --pmc passes of tools/r04_round_h.sh (one directory per
(shape, pass)); prints one block per shape: raw counters (averaged over the launches of the dominant kernel) and the derived
per-MFMA figures.  usage: python tools/pmc_one_conv_summary.py <dir>"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
shapes = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(root, "*_p[0-9]"))):
    shape = os.path.basename(d).rsplit("_p", 1)[0]
    files = glob.glob(os.path.join(d, "**", "run_counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = shapes.setdefault(shape, collections.OrderedDict())
    per = collections.defaultdict(lambda: [0, 0.0])
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        if "igemm_split3" not in k:
            continue
        a = per[(k.split("(")[0].replace("void acimg::", ""), r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (k, c), (n, v) in per.items():
        acc[(k, c)] = v / n
    tr = glob.glob(os.path.join(d, "**", "run_kernel_trace.csv"), recursive=True)
    if tr:
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if "igemm_split3" in r["Kernel_Name"]]
        if ds:
            acc[("duration_ns_under_pmc", os.path.basename(d))] = sum(ds) / len(ds)
for shape, acc in shapes.items():
    print("== shape %s (H,W,C,K,R,stride; batch 32)" % shape.replace("_", ","))
    vals = {}
    for (k, c), v in acc.items():
        print("   %-44s %-34s %16.0f" % (k[:44], c, v))
        vals[c] = v
    try:
        mf = vals["SQ_INSTS_MFMA"]
        per_simd = mf / (256 * 4)
        busy = vals["SQ_BUSY_CYCLES"]          # per rocprofv3: summed over XCDs / SEs, see DESIGN for the unit
        print("   MFMA / SIMD %.0f; non-MFMA instructions per MFMA: SALU %.2f VALU %.2f LDS %.2f VMEM %.3f" % (
            per_simd, vals["SQ_INSTS_SALU"] / mf, vals["SQ_INSTS_VALU"] / mf, vals["SQ_INSTS_LDS"] / mf, vals["SQ_INSTS_VMEM"] / mf))
        wc = vals["SQ_WAVE_CYCLES"]
        print("   wave cycles: parked (s_waitcnt / barrier) %.0f %%, issue stall %.0f %%, issuing %.0f %%; MFMA pipe busy %.0f %% of wave-quad-cycles/waves" % (
            100 * vals["SQ_WAIT_ANY"] / wc, 100 * vals["SQ_WAIT_INST_ANY"] / wc, 100 * vals["SQ_ACTIVE_INST_ANY"] / wc,
            100 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc)))
    except KeyError as e:
        print("   (missing counter %s)" % e)
