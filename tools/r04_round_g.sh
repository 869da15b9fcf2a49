# round 4, call g: sub-pixel form of the stride-2 transposed convs / data gradients: op tests, U-Net VAE tests, op report + bench of configs[1]
R=$GRAFT_REPO_ROOT
TAG=${1:-r04g}
cd $R
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "deconv or dgrad" > gpurun_out/${TAG}_ops.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_ops.log
tail -12 gpurun_out/${TAG}_ops.log
grep -q "rc=0" gpurun_out/${TAG}_ops.log || exit 1
timeout -k 10 600 python -m pytest tests/test_unet_vae_gpu.py tests/test_associator_gpu.py tests/test_joint_gpu.py tests/test_unet_acoustic_gpu.py -x -q -m gpu > gpurun_out/${TAG}_vae.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_vae.log
tail -6 gpurun_out/${TAG}_vae.log
python tools/op_report.py 32 0 unet_rgb bf16 > gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt 2>&1
grep -A14 "^sum" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt
python bench.py --workload unet_rgb > gpurun_out/${TAG}_bench_unet_rgb.json 2>/dev/null; cut -c1-250 gpurun_out/${TAG}_bench_unet_rgb.json
echo done
