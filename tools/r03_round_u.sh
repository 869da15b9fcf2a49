# round 3: two-pass conv3 (statistics pass + fused BN/shortcut/ReLU/split tail): op parity, model parity, bench
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "two_pass or trunk_kernel_variants_agree" > gpurun_out/r03u_ops.log 2>&1 || { tail -40 gpurun_out/r03u_ops.log; exit 1; }
tail -3 gpurun_out/r03u_ops.log
timeout -k 10 600 python -m pytest tests/test_trainstep_gpu.py -x -q > gpurun_out/r03u_trainstep.log 2>&1 || { tail -40 gpurun_out/r03u_trainstep.log; exit 1; }
tail -3 gpurun_out/r03u_trainstep.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err || { tail -5 gpurun_out/r03u_bench.err; exit 1; }
cut -c1-200 gpurun_out/r03u_bench.json
python tools/op_report.py 32 > gpurun_out/r03u_op_report.txt 2>&1 || true
tail -40 gpurun_out/r03u_op_report.txt
