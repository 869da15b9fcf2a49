# final evidence of round 2 at HEAD: default bench line, one-lane bench line, rocprof kernel stats of both, PMC traffic
# (one lane: per-dispatch counters need the kernel alone on the chip), per-op report, per-shape trunk table
set -e
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r02m_bench_default.json 2> gpurun_out/r02m_bench_default.err || { tail -5 gpurun_out/r02m_bench_default.err; exit 1; }
cut -c1-200 gpurun_out/r02m_bench_default.json
python bench.py --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/r02m_bench_onelane.json 2> gpurun_out/r02m_bench_onelane.err
cut -c1-200 gpurun_out/r02m_bench_onelane.json
ACIMG_NO_SIDE_LANE=1 python bench.py --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/r02m_bench_onestream.json 2> gpurun_out/r02m_bench_onestream.err
cut -c1-200 gpurun_out/r02m_bench_onestream.json
for w in unet_rgb unet_sound classifier; do
  python bench.py --workload $w --steps 10 --warmup 3 > gpurun_out/r02m_$w.json 2> gpurun_out/r02m_$w.err || { tail -5 gpurun_out/r02m_$w.err; exit 1; }
  cut -c1-150 gpurun_out/r02m_$w.json
done
python tools/op_report.py 32 > gpurun_out/r02m_op_report.txt 2>&1
python tools/trunk_shapes.py 12 one-tile=trunk_persistent:0 persistent=trunk_persistent:2 shipped= > gpurun_out/r02m_trunk_shapes.txt 2> gpurun_out/r02m_trunk_shapes.err
tail -2 gpurun_out/r02m_trunk_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02m_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/r02m_prof_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02m_prof1 -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-pipeline --no-cpu-baseline --no-secondary > $R/gpurun_out/r02m_prof1_bench.json 2>/dev/null
echo prof done
ACIMG_NO_SIDE_LANE=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_traffic/fetch -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary > /dev/null 2>&1
ACIMG_NO_SIDE_LANE=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_traffic/write -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary > /dev/null 2>&1
echo done
