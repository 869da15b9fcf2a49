# round 4, call a: the GPU suite with the oracle's thread count set (durations), then the default bench line
R=$GRAFT_REPO_ROOT
TAG=${1:-r04a}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -32 gpurun_out/${TAG}_gputests.log
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; }
cut -c1-300 gpurun_out/${TAG}_bench_default.json
echo done
