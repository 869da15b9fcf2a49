# round 3: loader glue test
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_trainer_surface_gpu.py -x -q -m gpu > gpurun_out/r03m_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03m_tests.log
tail -25 gpurun_out/r03m_tests.log
