# GPU-box call: full GPU suite, bench x2, per-shape table, rocprof kernel stats
set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r02b_tests.log 2>&1 || { tail -40 gpurun_out/r02b_tests.log; exit 1; }
tail -3 gpurun_out/r02b_tests.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r02b_bench1.json 2> gpurun_out/r02b_bench.err
ACIMG_NO_PERSISTENT=1 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r02b_bench_onetile.json 2>> gpurun_out/r02b_bench.err
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r02b_bench2.json 2>> gpurun_out/r02b_bench.err
ACIMG_NO_PERSISTENT=1 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r02b_bench_onetile2.json 2>> gpurun_out/r02b_bench.err
for f in r02b_bench1 r02b_bench_onetile r02b_bench2 r02b_bench_onetile2; do python -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(d['value']), round(d['ms_per_step'],3), 'dominant', r['kernel'], round(r['achieved'],1), 'TF', round(r['avg_launch_ms']*1e3,1), 'us x', r['launches_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02b_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/r02b_prof_bench.json 2>/dev/null
echo done
