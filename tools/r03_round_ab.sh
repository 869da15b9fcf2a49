# round 3: the one configuration of the sweep that printed nothing (stage cut 9, every identity unit in two passes)
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --stage-cut 9 --two-pass-cin 512 > gpurun_out/r03ab.json 2> gpurun_out/r03ab.err; echo rc=$?
tail -12 gpurun_out/r03ab.err; cut -c1-200 gpurun_out/r03ab.json
