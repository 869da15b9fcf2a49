# round 3: conv associator + refactored U-Net VAE recorder
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_associator_gpu.py tests/test_unet_vae_gpu.py tests/test_unet_acoustic_gpu.py -x -q -m gpu > gpurun_out/r03n_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03n_tests.log
tail -30 gpurun_out/r03n_tests.log
