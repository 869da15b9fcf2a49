# round 4, call z: wide slab reduce with 8 outputs per workgroup for small gradients, 16 part groups in the narrow column-sum finish
R=$GRAFT_REPO_ROOT
TAG=${1:-r04z2}
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_trainstep_gpu.py tests/test_unet_vae_gpu.py tests/test_associator_gpu.py tests/test_joint_gpu.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -4 gpurun_out/${TAG}_tests.log
grep -q "pytest rc=0" gpurun_out/${TAG}_tests.log || exit 1
python tools/op_report.py 32 0 unet_rgb bf16 > gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt 2>&1
grep -A8 "^sum" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt
python bench.py --workload unet_rgb --unet-precision bf16 > gpurun_out/${TAG}_bench_unet_rgb.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb.json
python bench.py --workload unet_rgb --unet-precision split > gpurun_out/${TAG}_bench_unet_rgb_split.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb_split.json
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench.json
echo done
