# round 3: the joint-latent step (trainermulti.py) against its oracle
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 800 python -m pytest tests/test_joint_gpu.py tests/test_unet_acoustic_gpu.py tests/test_associator_gpu.py -x -q > gpurun_out/r03x_joint.log 2>&1; echo "rc=$?" >> gpurun_out/r03x_joint.log
tail -40 gpurun_out/r03x_joint.log
