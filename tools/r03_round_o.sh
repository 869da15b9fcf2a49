# round 3: matrix-pipe micro-probe (what a K-step-shaped MFMA stream gets from the pipe with nothing else going on)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 120 tools/debug/mfma_probe > gpurun_out/r03o_mfma_probe.txt 2>&1; echo "rc=$?" >> gpurun_out/r03o_mfma_probe.txt
cat gpurun_out/r03o_mfma_probe.txt
