# final evidence of round 2 at HEAD (three-stage pipeline): default bench line; the same with fewer lanes; rocprof kernel
# stats of the default and the one-stream command; PMC traffic on one stream (per-dispatch counters need the kernel alone)
set -e
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r02p_bench_default.json 2> gpurun_out/r02p_bench_default.err || { tail -5 gpurun_out/r02p_bench_default.err; exit 1; }
cut -c1-200 gpurun_out/r02p_bench_default.json
ACIMG_TRUNK_STAGES=1 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r02p_bench_twolane.json 2>/dev/null
cut -c1-200 gpurun_out/r02p_bench_twolane.json
python bench.py --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/r02p_bench_nopipeline.json 2>/dev/null
cut -c1-200 gpurun_out/r02p_bench_nopipeline.json
ACIMG_NO_SIDE_LANE=1 python bench.py --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/r02p_bench_onestream.json 2>/dev/null
cut -c1-200 gpurun_out/r02p_bench_onestream.json
python tools/op_report.py 32 > gpurun_out/r02p_op_report.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02p_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/r02p_prof_bench.json 2>/dev/null
ACIMG_NO_SIDE_LANE=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02p_prof1 -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-pipeline --no-cpu-baseline --no-secondary > $R/gpurun_out/r02p_prof1_bench.json 2>/dev/null
echo prof done
ACIMG_NO_SIDE_LANE=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_traffic/fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary > /dev/null 2>&1
ACIMG_NO_SIDE_LANE=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_traffic/write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary > /dev/null 2>&1
echo done
