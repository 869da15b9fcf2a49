# round 3: ring kernel with the steady-state pair loop (fast steps folded, single register-set parity)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py -x -q -k "trunk_kernel_variants_agree" > gpurun_out/r03c_variants.log 2>&1 || { tail -30 gpurun_out/r03c_variants.log; exit 1; }
tail -3 gpurun_out/r03c_variants.log
timeout -k 10 300 python tools/trunk_shapes.py 6 "shipped=trunk_persistent:1" "ring256=trunk_ring:2,trunk_ring_bm:256" "ring128=trunk_ring:2,trunk_ring_bm:128" > gpurun_out/r03c_shapes.txt 2> gpurun_out/r03c_shapes.json || { tail -20 gpurun_out/r03c_shapes.json; exit 1; }
cat gpurun_out/r03c_shapes.txt
timeout -k 10 300 python tools/ablate_probe.py ring=256 28,38,256,256,3 56,75,128,512,1 28,38,1024,256,1 > gpurun_out/r03c_ablate_ring256.txt 2>&1 || { tail -20 gpurun_out/r03c_ablate_ring256.txt; exit 1; }
cat gpurun_out/r03c_ablate_ring256.txt
