# round 3: last check at HEAD - smoke(), the skinny / two-pass op tests, one oracle-parity train step, default-length bench
set -e
R=$GRAFT_REPO_ROOT
cd $R
python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "skinny or two_pass" 2>&1 | tail -1
timeout -k 10 400 python -m pytest tests/test_trainstep_gpu.py -x -q -k "matches_oracle and 1-False-f16x3" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-330
