# round 3: halo kernel (3x3 trunk convs from one staged patch per channel chunk): parity, then the per-shape table
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -k "halo_kernel_edges or trunk_kernel_variants_agree or prepare_multi" > gpurun_out/r03t_tests.log 2>&1 || { tail -40 gpurun_out/r03t_tests.log; exit 1; }
tail -3 gpurun_out/r03t_tests.log
timeout -k 10 400 python tools/trunk_shapes.py 4 "shipped=trunk_ring:1" "halo=trunk_ring:1,trunk_halo:2" "halo_whole=trunk_ring:1,trunk_halo:2,tail_split:0" > gpurun_out/r03t_shapes.txt 2> gpurun_out/r03t_shapes.json || { tail -20 gpurun_out/r03t_shapes.json; exit 1; }
cat gpurun_out/r03t_shapes.txt
