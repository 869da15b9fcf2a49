# round 3: skinny kernels for the VAE heads' data / weight gradients: op parity, model parity, per-op report, bench
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "skinny" > gpurun_out/r03ai_ops.log 2>&1 || { tail -30 gpurun_out/r03ai_ops.log; exit 1; }
tail -2 gpurun_out/r03ai_ops.log
timeout -k 10 700 python -m pytest tests/test_trainstep_gpu.py -x -q -k "matches_oracle or free_running or pipelined_steps_equal" > gpurun_out/r03ai_models.log 2>&1 || { tail -40 gpurun_out/r03ai_models.log; exit 1; }
tail -2 gpurun_out/r03ai_models.log
python tools/op_report.py 32 > gpurun_out/r03ai_op_report.txt 2>&1 || true
grep -E "28416|^sum" gpurun_out/r03ai_op_report.txt
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r03ai_bench.json 2>/dev/null; cut -c1-200 gpurun_out/r03ai_bench.json
