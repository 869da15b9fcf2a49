# round 3: more hardware queues for the runtime (GPU_MAX_HW_QUEUES, default 4) under the same three lanes + side lane
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r03an_q${q}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r03an_q${q}_$rep.json"))
print("GPU_MAX_HW_QUEUES ${q} (rep $rep): %.3f ms  %.0f img/s  lanes: %s" % (d["ms_per_step"], d["value"], d["config"]["lanes"][:40]))
PY
done; done
