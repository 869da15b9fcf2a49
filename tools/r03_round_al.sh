# round 3: the pipeline's extra lanes as high-priority HIP streams (stage 1 of the trunk keeps the caller's normal stream)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do for pr in 0 1; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --lane-priority $pr > gpurun_out/r03al_p${pr}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r03al_p${pr}_$rep.json"))
print("lane priority ${pr} (rep $rep): %.3f ms  %.0f img/s" % (d["ms_per_step"], d["value"]))
PY
done; done
