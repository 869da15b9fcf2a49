# round 3: full GPU suite + default bench with the ring kernel's auto rule (14x19 layers)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03j_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03j_gputests.log
tail -5 gpurun_out/r03j_gputests.log
python bench.py > gpurun_out/r03j_bench_default.json 2> gpurun_out/r03j_bench_default.err || tail -5 gpurun_out/r03j_bench_default.err
cut -c1-300 gpurun_out/r03j_bench_default.json
ACIMG_NO_RING=1 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r03j_bench_noring.json 2>/dev/null
cut -c1-200 gpurun_out/r03j_bench_noring.json
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r03j_bench_ring.json 2>/dev/null
cut -c1-200 gpurun_out/r03j_bench_ring.json
