# round 3: ring kernel: al' reads earlier; wave offset as a switch; MFMA-only ablation on a clean single-round shape
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py -x -q -k "trunk_kernel_variants_agree" > gpurun_out/r03f_variants.log 2>&1 || { tail -30 gpurun_out/r03f_variants.log; exit 1; }
tail -3 gpurun_out/r03f_variants.log
TRUNK_BATCH=30 timeout -k 10 300 python tools/trunk_shapes.py 5 "shipped=trunk_persistent:1" "ring256=trunk_ring:2,trunk_ring_bm:256" "ring256off=trunk_ring:2,trunk_ring_bm:256,trunk_stagger:50" > gpurun_out/r03f_shapes_b30.txt 2> gpurun_out/r03f_shapes_b30.json || { tail -20 gpurun_out/r03f_shapes_b30.json; exit 1; }
cat gpurun_out/r03f_shapes_b30.txt
TRUNK_BATCH=30 timeout -k 10 300 python tools/ablate_probe.py ring=256 28,38,256,256,3 28,38,1024,256,1 > gpurun_out/r03f_ablate_ring256_b30.txt 2>&1 || { tail -20 gpurun_out/r03f_ablate_ring256_b30.txt; exit 1; }
cat gpurun_out/r03f_ablate_ring256_b30.txt
