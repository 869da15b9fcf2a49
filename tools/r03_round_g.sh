# round 3: stamps of the ring kernel's skeleton (ablation bits 31: no requests / reads / stores, 1/3 MFMA; 23: full MFMA)
set -e
R=$GRAFT_REPO_ROOT
cd $R
TRUNK_BATCH=30 STAMP_BITS=23,31,6,16 timeout -k 10 200 python tools/stamp_probe.py ring=256 28,38,256,256,3,1 > gpurun_out/r03g_stamps_skeleton.txt 2>&1 || { tail -20 gpurun_out/r03g_stamps_skeleton.txt; exit 1; }
cat gpurun_out/r03g_stamps_skeleton.txt
