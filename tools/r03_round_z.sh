# round 3: generator layers (36x48, 432 tiles of 128x128) under other split-MFMA tiles: per-op reports
R=$GRAFT_REPO_ROOT
cd $R
python tools/op_report.py 32 > gpurun_out/r03z_op_default.txt 2>&1
ACIMG_SPLIT3_TILE=64x128 python tools/op_report.py 32 > gpurun_out/r03z_op_64x128.txt 2>&1
ACIMG_SPLIT3_TILE=128x64 python tools/op_report.py 32 > gpurun_out/r03z_op_128x64.txt 2>&1
for f in default 64x128 128x64; do echo == $f; grep -E "^ *(16[0-9]|17[0-9]|18[0-9]|19[0-9]|2[0-3][0-9]) .*split3" gpurun_out/r03z_op_$f.txt | grep -v split3p; done
