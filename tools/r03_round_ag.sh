# round 3: soak - 600 pipelined steps (finite, falling loss; stable step time) and the bit-exact two-pass tests 8 times
# in one process each (the store hazard was timing dependent)
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 600 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r03ag_soak.json 2> gpurun_out/r03ag_soak.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03ag_soak.json"))
print("600 steps: %.3f ms/step, %.0f img/s, final loss %.5f mse %.5f" % (d["ms_per_step"], d["value"], d["final_loss"], d["final_mse"]))
PY
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 120 python -m pytest tests/test_ops_gpu.py -x -q -k "two_pass" 2>&1 | tail -1; done
