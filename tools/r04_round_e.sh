# round 4, call e: full GPU suite (durations), default bench line, op report, A/B of the statistics source (--no-gram)
R=$GRAFT_REPO_ROOT
TAG=${1:-r04e}
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputests.log
tail -34 gpurun_out/${TAG}_gputests.log
python tools/op_report.py 32 > gpurun_out/${TAG}_op_report.txt 2>&1
grep -A14 "^sum" gpurun_out/${TAG}_op_report.txt
for rep in 1 2; do
  for v in gram nogram; do
    extra=""; [ $v = nogram ] && extra="--no-gram"
    python bench.py --no-cpu-baseline --no-secondary $extra > gpurun_out/${TAG}_bench_${v}_${rep}.json 2> gpurun_out/${TAG}_bench_${v}_${rep}.err || tail -5 gpurun_out/${TAG}_bench_${v}_${rep}.err
    echo "$v $rep: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_${v}_${rep}.json | head -1) $(grep -o '"achieved": [0-9.]*' gpurun_out/${TAG}_bench_${v}_${rep}.json | head -1)"
  done
done
python bench.py --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_onestream_gram.json 2>/dev/null
python bench.py --no-pipeline --no-side-lane --no-cpu-baseline --no-secondary --no-gram > gpurun_out/${TAG}_bench_onestream_nogram.json 2>/dev/null
for v in gram nogram; do echo "one stream $v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_onestream_${v}.json | head -1)"; done
echo done
