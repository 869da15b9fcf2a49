set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_dualcamnet_gpu.py tests/test_ops_gpu.py -x -q -m gpu -s > gpurun_out/r02c_tests.log 2>&1 || { tail -40 gpurun_out/r02c_tests.log; exit 1; }
grep "fp16 operand storage" gpurun_out/r02c_tests.log || true
tail -2 gpurun_out/r02c_tests.log
python bench.py --workload classifier --batch 60 --precision f16 --steps 20 --warmup 5 > gpurun_out/r02c_bench_cls_f16.json 2> gpurun_out/r02c_err.txt
python bench.py --workload classifier --batch 60 --steps 20 --warmup 5 > gpurun_out/r02c_bench_cls_f16x3.json 2>> gpurun_out/r02c_err.txt
cut -c1-260 gpurun_out/r02c_bench_cls_f16.json; cut -c1-260 gpurun_out/r02c_bench_cls_f16x3.json
bash tools/pmc_traffic.sh
echo done
