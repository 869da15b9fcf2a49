# round 3: the pipelined step under stage cuts x two-pass rules (HBM is shared by three lanes there: fewer bytes may pay
# where the extra K loop lost on one stream)
R=$GRAFT_REPO_ROOT
cd $R
for cut in 8 7 9; do for tp in 128 0 512; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --stage-cut $cut --two-pass-cin $tp > gpurun_out/r03aa_c${cut}_t${tp}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r03aa_c${cut}_t${tp}.json"))
print("cut ${cut} two-pass-cin ${tp}: %.3f ms  %.0f img/s  kernel %.1f TF" % (d["ms_per_step"], d["value"], d["roofline"]["achieved"]))
PY
done; done
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r03aa_again.json 2>/dev/null; cut -c1-160 gpurun_out/r03aa_again.json
