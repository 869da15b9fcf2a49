# round 3: checkpoint at HEAD (two-pass conv3): full GPU suite, default bench, PMC traffic on one stream, kernel stats
R=$GRAFT_REPO_ROOT
TAG=${1:-r03w}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -18 gpurun_out/${TAG}_gputests.log
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; }
cut -c1-260 gpurun_out/${TAG}_bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof1 -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof1_bench.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc/fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc/write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_pmc 3 ${2:-unknown} > gpurun_out/${TAG}_hbm_traffic.txt 2>&1
head -30 gpurun_out/${TAG}_hbm_traffic.txt
