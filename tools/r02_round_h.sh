# final evidence at HEAD: other workloads, rocprof kernel stats, PMC traffic
set -e
R=$GRAFT_REPO_ROOT
cd $R
for w in unet_rgb unet_sound classifier; do
  python bench.py --workload $w --steps 10 --warmup 3 > gpurun_out/r02h_$w.json 2> gpurun_out/r02h_$w.err || { tail -5 gpurun_out/r02h_$w.err; exit 1; }
  wc -l gpurun_out/r02h_$w.json | tr '\n' ' '; cut -c1-150 gpurun_out/r02h_$w.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02h_prof -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/r02h_prof_bench.json 2>/dev/null
echo prof done
bash $R/tools/pmc_traffic.sh
echo done
