# round 3, first GPU contact of the ring kernel: bit-agreement test, then the per-shape table against the shipped kernels
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py -x -q -k "trunk_kernel_variants_agree" > gpurun_out/r03a_variants.log 2>&1 || { tail -30 gpurun_out/r03a_variants.log; exit 1; }
tail -3 gpurun_out/r03a_variants.log
timeout -k 10 400 python tools/trunk_shapes.py 6 "shipped=trunk_persistent:1" "ring256=trunk_ring:2,trunk_ring_bm:256" "ring128=trunk_ring:2,trunk_ring_bm:128" "ring=trunk_ring:2" > gpurun_out/r03a_shapes.txt 2> gpurun_out/r03a_shapes.json || { tail -20 gpurun_out/r03a_shapes.json; exit 1; }
cat gpurun_out/r03a_shapes.txt
