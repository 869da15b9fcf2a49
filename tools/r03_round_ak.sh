# round 3: N = 1 records at HEAD for the scaling series: weak (batch 32) and strong (global batch 256 through the lanes), same box
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r03ak_weak_n1.json 2>/dev/null; cut -c1-220 gpurun_out/r03ak_weak_n1.json
python bench.py --steps 8 --warmup 3 --scaling strong --global-batch 256 --no-cpu-baseline --no-secondary > gpurun_out/r03ak_strong256_n1.json 2>/dev/null; cut -c1-220 gpurun_out/r03ak_strong256_n1.json
