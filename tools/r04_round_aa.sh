# round 4, call aa: bn_bwd reduce with four rows in flight per thread
R=$GRAFT_REPO_ROOT
TAG=${1:-r04aa}
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_vae_gpu.py tests/test_associator_gpu.py tests/test_joint_gpu.py tests/test_unet_acoustic_gpu.py tests/test_dualcamnet_gpu.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -4 gpurun_out/${TAG}_tests.log
grep -q "pytest rc=0" gpurun_out/${TAG}_tests.log || exit 1
python bench.py --workload unet_rgb --unet-precision bf16 > gpurun_out/${TAG}_bench_unet_rgb.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb.json
python bench.py --workload unet_rgb --unet-precision split > gpurun_out/${TAG}_bench_unet_rgb_split.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb_split.json
python bench.py --workload unet_sound --batch 4 > gpurun_out/${TAG}_bench_unet_sound_b4.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_sound_b4.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_unet -o run --output-format csv -- python3 $R/bench.py --workload unet_rgb --unet-precision bf16 --steps 5 --warmup 2 > /dev/null 2>&1
head -12 $R/gpurun_out/${TAG}_prof_unet/run_kernel_stats.csv | cut -c1-60,150-260
echo done
