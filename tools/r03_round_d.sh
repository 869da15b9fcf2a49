# round 3: where the ring kernel's step goes: stamps (clean single-round shape, tail shape, short-K shape), SQ counters of
# ring vs shipped on one long-K shape, and the table at batch 30 (28x38: one full round for both kernels)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 200 python tools/stamp_probe.py ring=256 14,19,512,512,3,1 28,38,256,256,3,1 56,75,128,512,1,1 > gpurun_out/r03d_stamps_ring256.txt 2>&1 || { tail -20 gpurun_out/r03d_stamps_ring256.txt; exit 1; }
cat gpurun_out/r03d_stamps_ring256.txt
TRUNK_BATCH=30 timeout -k 10 300 python tools/trunk_shapes.py 5 "shipped=trunk_persistent:1" "ring256=trunk_ring:2,trunk_ring_bm:256" "ring128=trunk_ring:2,trunk_ring_bm:128" > gpurun_out/r03d_shapes_b30.txt 2> gpurun_out/r03d_shapes_b30.json || { tail -20 gpurun_out/r03d_shapes_b30.json; exit 1; }
cat gpurun_out/r03d_shapes_b30.txt
cd /tmp && export TMPDIR=/tmp
for v in "ring trunk_ring:2,trunk_ring_bm:256,tail_split:0" "shipped trunk_persistent:2,tail_split:0"; do
  set -- $v
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
    TRUNK_BATCH=30 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/r03d_pmc/$1_$tag -o run --output-format csv -- python3 $R/tools/one_conv.py 28,38,256,256,3,1 $2 4 > /dev/null 2>&1 || echo "pmc pass failed: $1 $tag"
  done
done
ls -R $R/gpurun_out/r03d_pmc | head -30
