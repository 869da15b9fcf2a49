# round 3: the two-pass rule in the pipelined step, alternating repetitions: K <= 128 (shipped), <= 256, <= 512
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do for tp in 128 256 512; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --two-pass-cin $tp > gpurun_out/r03ao_t${tp}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ao_t${tp}_$rep.json"))
print("two-pass-cin ${tp} (rep $rep): %.3f ms  %.0f img/s" % (d["ms_per_step"], d["value"]))
PY
done; done
