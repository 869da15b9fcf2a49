# round 3: one tile per workgroup vs persistent on the brick layout, per shape (the tiles >= 512 rule dates from round 2)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python tools/trunk_shapes.py 4 "shipped=trunk_ring:1" "one_tile=trunk_ring:1,trunk_persistent:0" "persistent=trunk_ring:1,trunk_persistent:2" > gpurun_out/r03aj_shapes.txt 2> gpurun_out/r03aj_shapes.json || { tail -20 gpurun_out/r03aj_shapes.json; exit 1; }
cat gpurun_out/r03aj_shapes.txt
