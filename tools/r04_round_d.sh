# round 4, call d: Gram statistics after the ring-depth / reduce / quadfin changes: op test + per-kernel times
R=$GRAFT_REPO_ROOT
TAG=${1:-r04d}
cd $R
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gram" > gpurun_out/${TAG}_ops.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_ops.log
tail -5 gpurun_out/${TAG}_ops.log
grep -q "rc=0" gpurun_out/${TAG}_ops.log || exit 1
bash tools/r04_round_c.sh ${TAG}
