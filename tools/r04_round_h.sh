# round 4, call h: counter list of the box; SQ / TA / TCP counters of the dominant kernel on three trunk shapes (one_conv)
R=$GRAFT_REPO_ROOT
TAG=${1:-r04h}
cd $R
rocprofv3 -L > gpurun_out/${TAG}_counters_list.txt 2>&1
grep -c . gpurun_out/${TAG}_counters_list.txt
grep -o "\bTA_[A-Z_a-z]*\|\bTCP_[A-Z_a-z]*\|\bSQ_INST_CYCLES[A-Z_]*\|\bSQ_BUSY[A-Z_]*\|\bSQ_ACTIVE_INST[A-Z_]*\|\bSQ_WAIT[A-Z_]*\|\bGRBM_[A-Z_]*" gpurun_out/${TAG}_counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000
echo
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM"
P2="SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
P3="GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
for shape in 56,75,128,128,3,1 56,75,128,512,1,1 28,38,256,256,3,1; do
  s=$(echo $shape | tr ',' '_')
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/${TAG}_pmc/${s}_p$i -o run --output-format csv -- python3 $R/tools/one_conv.py $shape "" 6 > $R/gpurun_out/${TAG}_pmc/${s}_p$i.log 2>&1 || echo "pass $i of $shape failed"
  done
done
python3 $R/tools/pmc_one_conv_summary.py $R/gpurun_out/${TAG}_pmc > $R/gpurun_out/${TAG}_kloop_counters.txt 2>&1
cat $R/gpurun_out/${TAG}_kloop_counters.txt
echo done
