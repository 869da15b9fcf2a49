"""Timeline of the LAST train step in a rocprofv3 kernel trace of bench.py: every launch in order with its
duration and the idle gap before it, then totals per kernel.  usage: python tools/step_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pad_channels_kernel" in r["Kernel_Name"]]
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
b = starts[-1]; e = [x for x in ends if x > b][0]
step = rows[b:e + 1]
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0
agg = collections.OrderedDict(); gap_tot = 0
for i, r in enumerate(step):
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("acimg::", "").replace("void ", "")
    short = name.split("(")[0][:58]
    gap = (s - prev_end) / 1e3
    gap_tot += max(gap, 0)
    if "-v" in sys.argv:
        print("%4d %9.1f %7.1f us  gap %5.1f  %s" % (i, (s - t0) / 1e3, (en - s) / 1e3, gap, short))
    a = agg.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += (en - s) / 1e3
    prev_end = max(prev_end, en)
tot = (prev_end - t0) / 1e3
print("step: %d launches, %.1f us wall, %.1f us idle gaps" % (len(step), tot, gap_tot))
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s %4d %9.1f us %5.1f%%" % (k, n, us, 100 * us / tot))
