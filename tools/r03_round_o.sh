# round 3: matrix-pipe micro-probe (what a K-step-shaped MFMA stream gets from the pipe with nothing else going on)
R=$GRAFT_REPO_ROOT
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_probe.hip -o gpurun_out/mfma_probe   # built from the tracked source (no binary in git)
timeout -k 10 120 gpurun_out/mfma_probe > gpurun_out/r03o_mfma_probe.txt 2>&1; echo "rc=$?" >> gpurun_out/r03o_mfma_probe.txt
cat gpurun_out/r03o_mfma_probe.txt
