#!/bin/bash
# Round 6 diagnostic: a build of the whole library with EVERY streaming store turned into an ordinary write-back store (kAuxStream = 0),
# to see which kernels pay for partially written lines (profiles/r06zr_misaligned_rows.txt, r06zu_warp_gray_tiles.txt).  Not shipped:
# the output lands in scripts/ubench/bin/ (git-ignored) and is loaded through KORNIA_HIP_LIB by the ab_*_plain_* scripts.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); T=$(mktemp -d)
cp -r "$ROOT/kornia-rs_amd/csrc" "$T/csrc"; mkdir "$T/obj"
sed -i 's/constexpr int kAuxStream = kAuxSc0 | kAuxSc1 | kAuxNt;/constexpr int kAuxStream = 0;/' "$T/csrc/kh_common.h"
grep -q "kAuxStream = 0;" "$T/csrc/kh_common.h"
ls "$T"/csrc/*.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -Wno-unused-function -Wno-pass-failed -I$ROOT/include -I$T/csrc -c {} -o $T/obj/\$(basename {} .hip).o"
mkdir -p "$ROOT/scripts/ubench/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scripts/ubench/bin/libkornia_hip_plain.so" "$T"/obj/*.o
rm -rf "$T"; ls -la "$ROOT/scripts/ubench/bin/libkornia_hip_plain.so"
