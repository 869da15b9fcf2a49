# diagnostic libraries of the persistent trunk kernel (never shipped, never loaded by the product):
#   tools/debug/libacimg_stamp.so   in-kernel cycle stamps (+ the ablation switches)   -> tools/stamp_probe.py
#   tools/debug/libacimg_ablate.so  the ablation switches alone, undisturbed timing    -> tools/ablate_probe.py
set -e
cd "$(dirname "$0")/../acoustic-image-generation_amd/csrc"
OUT=../../tools/debug
mkdir -p $OUT
for kind in stamp ablate; do
  DEF=$([ $kind = stamp ] && echo -DACIMG_STAMP || echo -DACIMG_ABLATE)
  FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize $DEF"
  for f in igemm elementwise frontend hostutil triplet records; do
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/${kind}_$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/${kind}_*.o -lz -o $OUT/libacimg_$kind.so
  rm -f $OUT/${kind}_*.o
  echo built $OUT/libacimg_$kind.so
done
