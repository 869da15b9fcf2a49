# round 4, call i: the other branches of the joint trainer (fusion, onlyaudiovideo, moddrop) against the oracle
R=$GRAFT_REPO_ROOT
TAG=${1:-r04i}
cd $R
timeout -k 10 900 python -m pytest tests/test_joint_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "joint or deconv or dgrad" > gpurun_out/${TAG}_joint.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_joint.log
tail -25 gpurun_out/${TAG}_joint.log
