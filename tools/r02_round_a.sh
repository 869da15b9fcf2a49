# one GPU-box call: GPU test suite, bench (default + strong), rocprof kernel stats, PMC traffic passes
set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -x -v -m gpu > gpurun_out/r02a_tests.log 2>&1 || { tail -40 gpurun_out/r02a_tests.log; exit 1; }
tail -3 gpurun_out/r02a_tests.log
python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err || { tail -20 gpurun_out/r02a_bench.err; exit 1; }
echo bench done
python bench.py --scaling strong --global-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r02a_bench_strong.json 2> gpurun_out/r02a_bench_strong.err || { tail -20 gpurun_out/r02a_bench_strong.err; exit 1; }
echo strong done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02a_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/r02a_prof_bench.json 2>/dev/null
echo prof done
bash $R/tools/pmc_traffic.sh
echo done
