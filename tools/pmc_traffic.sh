# HBM traffic of a bench.py command from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, kernel trace only)
# usage: tools/pmc_traffic.sh [subdir under gpurun_out/pmc_traffic] [bench.py arguments ...]   (default: the headline workload)
set -e
R=$GRAFT_REPO_ROOT
SUB=${1:-main}
shift || true
ARGS="$@"
[ -z "$ARGS" ] && ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_traffic/$SUB/fetch -o run --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_traffic/$SUB/write -o run --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2>&1
echo ok
