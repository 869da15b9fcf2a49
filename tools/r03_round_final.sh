# final evidence of round 3 at HEAD: full GPU suite (durations), default bench line, rocprof kernel stats of the default and
# the one-stream command, PMC traffic on one stream (per-dispatch counters need the kernel alone), per-op report
R=$GRAFT_REPO_ROOT
TAG=${1:-r03z}
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -22 gpurun_out/${TAG}_gputests.log
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; }
cut -c1-260 gpurun_out/${TAG}_bench_default.json
python bench.py --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_onestream.json 2>/dev/null
cut -c1-200 gpurun_out/${TAG}_bench_onestream.json
python tools/op_report.py 32 > gpurun_out/${TAG}_op_report.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof1 -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof1_bench.json 2>/dev/null
echo prof done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc/fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc/write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > /dev/null 2>&1
ls $R/gpurun_out/${TAG}_prof $R/gpurun_out/${TAG}_pmc/fetch | head
echo done
