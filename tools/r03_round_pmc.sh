# HBM traffic by counters at HEAD: separate FETCH_SIZE / WRITE_SIZE passes of the one-stream bench (per-dispatch counters
# need the kernel alone), summary -> gpurun_out/<tag>_hbm_traffic.txt
R=$GRAFT_REPO_ROOT
TAG=${1:-r03p}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc/fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc/write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_pmc 3 ${2:-unknown} > gpurun_out/${TAG}_hbm_traffic.txt 2>&1
head -14 gpurun_out/${TAG}_hbm_traffic.txt; tail -3 gpurun_out/${TAG}_hbm_traffic.txt
