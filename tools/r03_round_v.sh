# (historical: trunk_dma_pos 2 was measured with this script and then removed from the library - DESIGN 7d (3); configure now refuses it)
# round 3: operand requests two steps ahead (trunk_dma_pos 2): parity, per-shape table, stats / tail passes
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -k "two_pass or trunk_kernel_variants_agree" > gpurun_out/r03v_ops.log 2>&1 || { tail -40 gpurun_out/r03v_ops.log; exit 1; }
tail -3 gpurun_out/r03v_ops.log
timeout -k 10 400 python tools/trunk_shapes.py 4 "shipped=trunk_ring:1" "deep=trunk_ring:1,trunk_dma_pos:2" "deep_all=trunk_ring:1,trunk_dma_pos:2,trunk_persistent:2" > gpurun_out/r03v_shapes.txt 2> gpurun_out/r03v_shapes.json || { tail -20 gpurun_out/r03v_shapes.json; exit 1; }
cat gpurun_out/r03v_shapes.txt
ACIMG_TRUNK_DMA_POS=2 python tools/op_report.py 32 > gpurun_out/r03v_op_report.txt 2>&1 || true
grep -E "stats |_tail|^sum|  conv2d_fwd_split3p|  bn_" gpurun_out/r03v_op_report.txt
