# diagnostic library with in-kernel cycle stamps in the persistent trunk kernel: tools/debug/libacimg_stamp.so
set -e
cd "$(dirname "$0")/../acoustic-image-generation_amd/csrc"
OUT=../../tools/debug
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed -DACIMG_STAMP"
for f in igemm elementwise frontend hostutil triplet records; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/stamp_$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/stamp_*.o -lz -o $OUT/libacimg_stamp.so
rm -f $OUT/stamp_*.o
echo built $OUT/libacimg_stamp.so
