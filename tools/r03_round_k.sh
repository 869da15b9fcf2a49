# round 3: the new host-side tests + the strong-scaling N=1 record (pipelined shards) next to the weak line
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_trainstep_gpu.py -x -q -m gpu --durations=12 > gpurun_out/r03k_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03k_tests.log
tail -25 gpurun_out/r03k_tests.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r03k_bench_weak.json 2> gpurun_out/r03k_bench_weak.err || tail -5 gpurun_out/r03k_bench_weak.err
cut -c1-220 gpurun_out/r03k_bench_weak.json
python bench.py --scaling strong --global-batch 256 --no-cpu-baseline --no-secondary > gpurun_out/r03k_bench_strong256.json 2> gpurun_out/r03k_bench_strong256.err || tail -5 gpurun_out/r03k_bench_strong256.err
cut -c1-220 gpurun_out/r03k_bench_strong256.json
python bench.py --scaling strong --global-batch 256 --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/r03k_bench_strong256_onestream.json 2>/dev/null
cut -c1-220 gpurun_out/r03k_bench_strong256_onestream.json
