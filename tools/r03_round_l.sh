# round 3: two-slot ring kernel (ring2): agreement + per-shape table at batch 32 (+ ablation on two shapes)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py -x -q -k "trunk_kernel_variants_agree" > gpurun_out/r03l_variants.log 2>&1 || { tail -30 gpurun_out/r03l_variants.log; exit 1; }
tail -3 gpurun_out/r03l_variants.log
timeout -k 10 400 python tools/trunk_shapes.py 6 "shipped=trunk_ring:0" "ring2=trunk_ring:3" "ring2w=trunk_ring:3,tail_split:0" > gpurun_out/r03l_shapes.txt 2> gpurun_out/r03l_shapes.json || { tail -20 gpurun_out/r03l_shapes.json; exit 1; }
cat gpurun_out/r03l_shapes.txt
