# round 4, call c: per-kernel times of the Gram statistics path (rocprofv3 kernel trace of tools/gram_probe.py)
R=$GRAFT_REPO_ROOT
TAG=${1:-r04c}
cd $R
python tools/gram_probe.py 20 > gpurun_out/${TAG}_gram_probe.txt 2>&1
cat gpurun_out/${TAG}_gram_probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o run --output-format csv -- python3 $R/tools/gram_probe.py 20 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/${TAG}_prof/**/run_kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if "gram" in r["Name"] or "bn_relu_split" in r["Name"]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/${TAG}_prof/**/run_kernel_trace.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "gram" in r["Kernel_Name"]]
# per shape (calls come in groups of 23 x 3 launches per shape)
import collections
seq=collections.OrderedDict()
for i,r in enumerate(rows):
    shape=i//(23*3)
    k=(shape,r["Kernel_Name"].split("(")[0][:40])
    seq.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in seq.items():
    v=sorted(v); print(k, "n=%d median %.1f us"%(len(v), v[len(v)//2]/1e3))
PY
echo done
