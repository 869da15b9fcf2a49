# round 4, call l: pipeline stage cut after the Gram statistics (the trunk stages lost ~0.3 ms between them); RCCL world-1 line
R=$GRAFT_REPO_ROOT
TAG=${1:-r04l}
cd $R
for rep in 1 2; do
  for cut in 7 8 9 10; do
    python bench.py --no-cpu-baseline --no-secondary --stage-cut $cut > gpurun_out/${TAG}_cut${cut}_${rep}.json 2>/dev/null
    echo "stage cut $cut rep $rep: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_cut${cut}_${rep}.json | head -1)"
  done
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 20 --warmup 5 --dp-force --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_dpforce_world1.json 2> gpurun_out/${TAG}_bench_dpforce_world1.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_dpforce_world1.json"))
print("dp-force world 1:", d["ms_per_step"], d["config"]["dp"], d["config"]["exchange"])
PY
echo done
