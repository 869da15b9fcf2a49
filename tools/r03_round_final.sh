# final evidence of round 3 at HEAD: full GPU suite (durations), default bench line (pipelined, CPU baseline, secondaries),
# rocprof kernel stats of the default and of the one-stream command, per-op report.  (PMC traffic: tools/r03_round_w.sh ran
# it on the same kernel tree - profiles/r03/hbm_traffic_r03a_9674477.txt.)
R=$GRAFT_REPO_ROOT
TAG=${1:-r03z}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -22 gpurun_out/${TAG}_gputests.log
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; }
cut -c1-260 gpurun_out/${TAG}_bench_default.json
python tools/op_report.py 32 > gpurun_out/${TAG}_op_report.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof1 -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof1_bench.json 2>/dev/null
echo done
