# build libacimg.so of another commit for same-process A/B timing (tools/trunk_shapes.py variant `lib:<path>`):
#   tools/build_ref_lib.sh <git-ref> [name]   ->   tools/debug/libacimg_<name>.so
set -e
REF=${1:?git ref}
NAME=${2:-ref}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d /tmp/acimg_ref.XXXXXX)
git -C "$ROOT" archive "$REF" acoustic-image-generation_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/acoustic-image-generation_amd/csrc" -j8 >/dev/null
mkdir -p "$ROOT/tools/debug"
cp "$TMP/acoustic-image-generation_amd/csrc/libacimg.so" "$ROOT/tools/debug/libacimg_$NAME.so" 2>/dev/null ||
  cp "$(find "$TMP" -name libacimg.so | head -1)" "$ROOT/tools/debug/libacimg_$NAME.so"
rm -rf "$TMP"
echo built tools/debug/libacimg_$NAME.so from $REF
