set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ACIMG_SPLIT3_DMA=2
for shape in "28 38 256 1024 1 1" "56 75 128 128 3 1"; do
tag=$(echo $shape | tr ' ' '_')
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_d/$tag/sq -o run --output-format csv -- python3 $R/tools/one_conv.py $shape 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $R/gpurun_out/pmc_d/$tag/tcc -o run --output-format csv -- python3 $R/tools/one_conv.py $shape 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_d/$tag/tcp -o run --output-format csv -- python3 $R/tools/one_conv.py $shape 3 > /dev/null 2>&1 || echo tcp-pass-failed
done
echo ok
