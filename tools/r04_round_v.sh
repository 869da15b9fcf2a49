# round 4, call v: weight-gradient slabs to XCDs (wgrad_split3_kernel): op tests, train-step tests, op report and bench of the main workload
R=$GRAFT_REPO_ROOT
TAG=${1:-r04v}
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_trainstep_gpu.py tests/test_unet_vae_gpu.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -5 gpurun_out/${TAG}_tests.log
grep -q "pytest rc=0" gpurun_out/${TAG}_tests.log || exit 1
python tools/op_report.py 32 > gpurun_out/${TAG}_op_report.txt 2>&1
grep -A12 "^sum" gpurun_out/${TAG}_op_report.txt
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench.json
python bench.py --no-cpu-baseline --no-secondary --no-pipeline --no-side-lane > gpurun_out/${TAG}_bench_onestream.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_onestream.json
python bench.py --workload unet_rgb --unet-precision bf16 > gpurun_out/${TAG}_bench_unet_rgb.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb.json
echo done
