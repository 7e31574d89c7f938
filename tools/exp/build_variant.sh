#!/bin/bash
# Build an experiment library from a patched COPY of 2048_amd/csrc (the tracked sources and lib2048_hip.so are not touched):
#     tools/exp/build_variant.sh NAME edit.py        ->  tools/exp/build/lib_NAME.so
# edit.py is run with the scratch copy of g2048.hip as argv[1] (it rewrites the file in place).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; EDIT=$2
S=/tmp/g2048_variant_$NAME
rm -rf "$S"; mkdir -p "$S/2048_amd" "$S/include" "$ROOT/tools/exp/build"
cp -r "$ROOT/2048_amd/csrc" "$S/2048_amd/csrc"; cp "$ROOT/include/g2048.h" "$S/include/"
python3 "$EDIT" "$S/2048_amd/csrc/g2048.hip"
(cd "$S/2048_amd/csrc" && diff -u "$ROOT/2048_amd/csrc/g2048.hip" g2048.hip > "$ROOT/tools/exp/build/$NAME.diff" || true)
(cd "$S/2048_amd/csrc" && touch g2048.hip && make -s ../lib2048_hip.so 2>&1 | grep -E "error|warning: v" || true)
cp "$S/2048_amd/lib2048_hip.so" "$ROOT/tools/exp/build/lib_$NAME.so"
ls -la "$ROOT/tools/exp/build/lib_$NAME.so"
