#!/bin/bash
# Build experiment variants of the library into tools/ab_libs/ (for tools/ab_builds.py run).
#   tools/build_variants.sh name1:"-DX=1" name2:"-DY=2 -DZ=1" ...     ("base" = no defines)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab_libs
rm -f tools/ab_libs/*.so
k=0
for spec in "$@"; do
    name="${spec%%:*}"; defs="${spec#*:}"
    [ "$defs" = "$spec" ] && defs=""
    D2D_BUILD_DEFINES="$defs" python -m gym_d2d_amd.build --force > /dev/null
    cp gym_d2d_amd/lib/libd2d_hip.so "tools/ab_libs/$(printf %02d $k)_${name}.so"
    echo "built $name ($defs)"
    k=$((k+1))
done
python -m gym_d2d_amd.build --force > /dev/null      # leave the shipped build in place
