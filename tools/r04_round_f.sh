# round 4, call f: full GPU suite (durations) after the bench.py / worker-thread fixes; op report of configs[1] (UNet RGB VAE, bf16 and split)
R=$GRAFT_REPO_ROOT
TAG=${1:-r04f}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -34 gpurun_out/${TAG}_gputests.log
python tools/op_report.py 32 0 unet_rgb bf16 > gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt 2>&1
grep -A25 "^sum" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt
echo done
