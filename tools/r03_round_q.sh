# round 3: ring tail after the reduce fix: forced K ranges at batch 32, long-K shapes
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python -m pytest tests/test_ops_gpu.py -x -q -k "trunk_kernel_variants_agree or lanes_are_measured" > gpurun_out/r03q_variants.log 2>&1 || { tail -30 gpurun_out/r03q_variants.log; exit 1; }
tail -3 gpurun_out/r03q_variants.log
timeout -k 10 500 python tools/trunk_shapes.py 4 "shipped=trunk_ring:0" "r256s2=trunk_ring:2,trunk_ring_bm:256,tail_s:2" "r256s4=trunk_ring:2,trunk_ring_bm:256,tail_s:4" "r256s6=trunk_ring:2,trunk_ring_bm:256,tail_s:6" "r256s8=trunk_ring:2,trunk_ring_bm:256,tail_s:8" "r256s12=trunk_ring:2,trunk_ring_bm:256,tail_s:12" "r128s8=trunk_ring:2,trunk_ring_bm:128,tail_s:8" "r128s16=trunk_ring:2,trunk_ring_bm:128,tail_s:16" > gpurun_out/r03q_shapes_tail.txt 2> gpurun_out/r03q_shapes_tail.json || { tail -20 gpurun_out/r03q_shapes_tail.json; exit 1; }
cat gpurun_out/r03q_shapes_tail.txt
