# round 3: the persistent kernel's timing switches re-measured on the brick layout (request placement, start stagger)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python tools/trunk_shapes.py 4 "shipped=trunk_ring:1" "spread=trunk_ring:1,trunk_dma_pos:1" "stag25=trunk_ring:1,trunk_stagger:25" "stag50=trunk_ring:1,trunk_stagger:50" "spread_stag=trunk_ring:1,trunk_dma_pos:1,trunk_stagger:25" > gpurun_out/r03af_shapes.txt 2> gpurun_out/r03af_shapes.json || { tail -20 gpurun_out/r03af_shapes.json; exit 1; }
cat gpurun_out/r03af_shapes.txt
