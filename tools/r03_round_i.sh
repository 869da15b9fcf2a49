# round 3: ring kernel tail split: forced numbers of K ranges on the batch-32 table
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python tools/trunk_shapes.py 4 "shipped=trunk_persistent:1" "r256s4=trunk_ring:2,trunk_ring_bm:256,tail_s:4" "r256s6=trunk_ring:2,trunk_ring_bm:256,tail_s:6" "r256s8=trunk_ring:2,trunk_ring_bm:256,tail_s:8" "r128s4=trunk_ring:2,trunk_ring_bm:128,tail_s:4" "r128s8=trunk_ring:2,trunk_ring_bm:128,tail_s:8" "r256w=trunk_ring:2,trunk_ring_bm:256,tail_split:0" > gpurun_out/r03i_shapes_tail.txt 2> gpurun_out/r03i_shapes_tail.json || { tail -20 gpurun_out/r03i_shapes_tail.json; exit 1; }
cat gpurun_out/r03i_shapes_tail.txt
