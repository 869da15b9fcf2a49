# round 3: the lo-plane store with its offset in the vector operand (hazard-safe form), identity and projection tails:
# op parity (bit-equal), model parity, bench, per-op report
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "two_pass" > gpurun_out/r03ae_ops.log 2>&1 || { tail -30 gpurun_out/r03ae_ops.log; exit 1; }
tail -2 gpurun_out/r03ae_ops.log
timeout -k 10 700 python -m pytest tests/test_trainstep_gpu.py -x -q > gpurun_out/r03ae_trainstep.log 2>&1 || { tail -40 gpurun_out/r03ae_trainstep.log; exit 1; }
tail -2 gpurun_out/r03ae_trainstep.log
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary > gpurun_out/r03ae_bench.json 2> gpurun_out/r03ae_bench.err || { tail -5 gpurun_out/r03ae_bench.err; exit 1; }
cut -c1-200 gpurun_out/r03ae_bench.json
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary --two-pass-cin 0 > gpurun_out/r03ae_bench_off.json 2>/dev/null; cut -c1-200 gpurun_out/r03ae_bench_off.json
python tools/op_report.py 32 > gpurun_out/r03ae_op_report.txt 2>&1 || true
grep -E "stats |_tail|^sum|  conv2d_fwd_split3p|  bn_" gpurun_out/r03ae_op_report.txt
