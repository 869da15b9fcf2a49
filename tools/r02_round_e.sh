set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r02e_tests.log 2>&1 || { tail -40 gpurun_out/r02e_tests.log; exit 1; }
tail -2 gpurun_out/r02e_tests.log
( time python bench.py > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err ) 2>&1 | tail -3
wc -l gpurun_out/r02e_bench.json
python -c "import __graft_entry__ as g; g.smoke()"
echo done
