for v in "192 512" "320 768" "320 1024" "640 1024" "640 1536"; do
  set -- $v
  ACIMG_SPLITK_CUT=$1 ACIMG_SPLITK_TARGET=$2 timeout -k 10 200 python tools/op_report.py 32 > gpurun_out/opr_$1_$2.txt 2>&1 || exit 1
  echo "cut=$1 target=$2: $(grep '^sum' gpurun_out/opr_$1_$2.txt) fwd=$(grep '  conv2d_fwd  ' gpurun_out/opr_$1_$2.txt) dgrad=$(grep '  conv2d_dgrad  ' gpurun_out/opr_$1_$2.txt)"
done
