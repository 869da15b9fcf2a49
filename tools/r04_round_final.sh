# evidence of round 4 at HEAD: full GPU suite (durations), default bench line (pipelined, CPU baseline, secondaries), the other
# workload lines, rocprof kernel stats of the default and of the one-stream command, per-op reports, PMC traffic of the three
# benched configurations (one stream: per-dispatch counters need the kernel alone on the chip).
# usage: tools/r04_round_final.sh <tag> <commit>
R=$GRAFT_REPO_ROOT
TAG=${1:-r04z}
COMMIT=${2:-unknown}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -22 gpurun_out/${TAG}_gputests.log
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; }
cut -c1-260 gpurun_out/${TAG}_bench_default.json
python bench.py --workload unet_rgb --unet-precision bf16 > gpurun_out/${TAG}_bench_unet_rgb_bf16.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_rgb_bf16.json
python bench.py --workload classifier --batch 60 > gpurun_out/${TAG}_bench_classifier60.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_classifier60.json
python bench.py --workload classifier --batch 60 --precision f16 > gpurun_out/${TAG}_bench_classifier60_f16.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_classifier60_f16.json
python bench.py --workload unet_sound --batch 4 > gpurun_out/${TAG}_bench_unet_sound_b4.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_unet_sound_b4.json
python bench.py --scaling strong --global-batch 256 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_strong256_n1.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_strong256_n1.json
python tools/op_report.py 32 > gpurun_out/${TAG}_op_report.txt 2>&1
grep -A16 "^sum" gpurun_out/${TAG}_op_report.txt
python tools/op_report.py 32 0 unet_rgb bf16 > gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt 2>&1
grep -A6 "^sum" gpurun_out/${TAG}_op_report_unet_rgb_bf16.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof1 -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > $R/gpurun_out/${TAG}_prof1_bench.json 2>/dev/null
cd $R
bash tools/pmc_traffic.sh ${TAG}_main --steps 6 --warmup 2 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary
python tools/pmc_summary.py gpurun_out/pmc_traffic/${TAG}_main 8 $COMMIT "python3 bench.py --steps 6 --warmup 2 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary, batch 32" > gpurun_out/${TAG}_hbm_traffic_main.txt 2>&1
head -12 gpurun_out/${TAG}_hbm_traffic_main.txt; tail -2 gpurun_out/${TAG}_hbm_traffic_main.txt
bash tools/pmc_traffic.sh ${TAG}_b64 --num-skip 2 --batch 64 --steps 3 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary
python tools/pmc_summary.py gpurun_out/pmc_traffic/${TAG}_b64 4 $COMMIT "python3 bench.py --num-skip 2 --batch 64 --steps 3 --warmup 1 --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary (BASELINE configs[2])" > gpurun_out/${TAG}_hbm_traffic_b64.txt 2>&1
head -8 gpurun_out/${TAG}_hbm_traffic_b64.txt; tail -2 gpurun_out/${TAG}_hbm_traffic_b64.txt
bash tools/pmc_traffic.sh ${TAG}_unet_rgb --workload unet_rgb --unet-precision bf16 --steps 4 --warmup 2
python tools/pmc_summary.py gpurun_out/pmc_traffic/${TAG}_unet_rgb 6 $COMMIT "python3 bench.py --workload unet_rgb --unet-precision bf16 --steps 4 --warmup 2, batch 32 (BASELINE configs[1])" > gpurun_out/${TAG}_hbm_traffic_unet_rgb.txt 2>&1
head -10 gpurun_out/${TAG}_hbm_traffic_unet_rgb.txt; tail -2 gpurun_out/${TAG}_hbm_traffic_unet_rgb.txt
echo done
