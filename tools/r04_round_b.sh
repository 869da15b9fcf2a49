# round 4, call b: the Gram statistics path - op test, the train-step parity tests that go through it, op report
R=$GRAFT_REPO_ROOT
TAG=${1:-r04b}
cd $R
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gram or two_pass or skinny" > gpurun_out/${TAG}_ops.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_ops.log
tail -15 gpurun_out/${TAG}_ops.log
grep -q "rc=0" gpurun_out/${TAG}_ops.log || exit 1
timeout -k 10 900 python -m pytest tests/test_trainstep_gpu.py -x -q -m gpu -k "matches_oracle or bench_configuration or full_size or pipelined_schedule" --durations=10 > gpurun_out/${TAG}_trainstep.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_trainstep.log
tail -25 gpurun_out/${TAG}_trainstep.log
python tools/op_report.py 32 > gpurun_out/${TAG}_op_report.txt 2>&1
grep -A12 "^sum" gpurun_out/${TAG}_op_report.txt
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || tail -5 gpurun_out/${TAG}_bench.err
cut -c1-200 gpurun_out/${TAG}_bench.json
echo done
