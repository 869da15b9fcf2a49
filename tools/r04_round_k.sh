# round 4, call k: halo form of the 32/64 -> 32 channel weight gradients: op test, U-Net VAE tests, op report + bench of configs[1]
R=$GRAFT_REPO_ROOT
TAG=${1:-r04k}
cd $R
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo16 or wgrad" > gpurun_out/${TAG}_ops.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_ops.log
tail -12 gpurun_out/${TAG}_ops.log
grep -q "rc=0" gpurun_out/${TAG}_ops.log || exit 1
timeout -k 10 600 python -m pytest tests/test_unet_vae_gpu.py tests/test_associator_gpu.py tests/test_joint_gpu.py -x -q -m gpu > gpurun_out/${TAG}_vae.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_vae.log
tail -6 gpurun_out/${TAG}_vae.log
python tools/op_report.py 32 0 unet_rgb bf16 > gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt 2>&1
grep -A10 "^sum" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt
grep "wgrad_bf16" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt | head -12
python bench.py --workload unet_rgb --unet-precision bf16 > gpurun_out/${TAG}_bench_unet_rgb.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb.json
python bench.py --workload unet_rgb --unet-precision split > gpurun_out/${TAG}_bench_unet_rgb_split.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb_split.json
echo done
