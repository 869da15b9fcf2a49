# round 4, call n: bn_bwd reduce with 1024 partial workgroups: U-Net VAE tests + op report of configs[1]
R=$GRAFT_REPO_ROOT
TAG=${1:-r04n}
cd $R
timeout -k 10 600 python -m pytest tests/test_unet_vae_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "bn or vae or unet" > gpurun_out/${TAG}_vae.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_vae.log
tail -5 gpurun_out/${TAG}_vae.log
python tools/op_report.py 32 0 unet_rgb bf16 > gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt 2>&1
grep -A8 "^sum" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt
python bench.py --workload unet_rgb --unet-precision bf16 > gpurun_out/${TAG}_bench_unet_rgb.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb.json
echo done
