# round 3: 64x128 tiles on the layers with few 128x128 tiles (one workgroup per CU there)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python tools/trunk_shapes.py 4 "shipped=trunk_ring:1" "t64x128=split3_tile_bm:64,split3_tile_bn:128,trunk_ring:0" "t128x64=split3_tile_bm:128,split3_tile_bn:64,trunk_ring:0" > gpurun_out/r03y_shapes.txt 2> gpurun_out/r03y_shapes.json || { tail -20 gpurun_out/r03y_shapes.json; exit 1; }
cat gpurun_out/r03y_shapes.txt
