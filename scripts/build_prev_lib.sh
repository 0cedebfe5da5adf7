#!/bin/bash
# Dev: build the library of an EARLIER COMMIT next to the current one, for same-box A/Bs:
#   bash scripts/build_prev_lib.sh <commit> [name]   ->  kornia-rs_amd/lib/libkornia_hip_<name>.so   (default name: prev)
#   KORNIA_HIP_LIB=kornia-rs_amd/lib/libkornia_hip_prev.so python bench.py --workload ...
set -eu
C=$1; NAME=${2:-prev}; REPO=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
git -C "$REPO" archive "$C" kornia-rs_amd/csrc kornia-rs_amd/Makefile kornia-rs_amd/diag include | tar -x -C "$T"
make -C "$T/kornia-rs_amd" -j8 "$T/kornia-rs_amd/lib/libkornia_hip.so" > "$T/build.log" 2>&1 || { tail -20 "$T/build.log"; exit 1; }
cp "$T/kornia-rs_amd/lib/libkornia_hip.so" "$REPO/kornia-rs_amd/lib/libkornia_hip_$NAME.so"
rm -rf "$T"; ls -la "$REPO/kornia-rs_amd/lib/libkornia_hip_$NAME.so"
