#!/bin/bash
# AddressSanitizer + UBSan over the host-side native code (splitter, de-dup, text generator) on
# adversarial inputs: invalid / truncated UTF-8, empty and tiny texts, arbitrary document cuts.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/libhost_asan.so
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -shared -fPIC -pthread \
    -I"$ROOT/include" -I"$ROOT/minbpe_amd/csrc" \
    "$ROOT/minbpe_amd/csrc/split.cpp" "$ROOT/minbpe_amd/csrc/dedup.cpp" "$ROOT/minbpe_amd/csrc/synth.cpp" -o "$OUT"
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python "$ROOT/tools/asan_drive.py" "$OUT"
