#!/bin/bash
# Same-box A/B of persistent-Cholesky build variants (potrf.hip recompiled with extra -D flags, the other objects shared):
#   tools/potrf_variants.sh build NAME "-DFLAG=.. ..."      (here: cross-compiles to tools/tmp/libpotrf_NAME.so)
#   gpurun -- 'bash tools/potrf_variants.sh run NAME1 NAME2 ...'   (on the GPU box: tools/nll_sizes.py per variant, three rounds)
set -e
C=approxposterior_amd/csrc
if [ "$1" = build ]; then
    mkdir -p tools/tmp
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c $C/potrf.hip -o /tmp/pv_$2.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/libpotrf_$2.so $C/gram.o $C/linalg.o $C/sweep.o $C/grad.o /tmp/pv_$2.o $C/ensemble.o
    exit 0
fi
shift
for rep in 1 2; do
    for v in "$@"; do
        if [ $v = ship ]; then timeout 200 python tools/nll_sizes.py 2>&1 | tail -1; else timeout 200 python tools/nll_sizes.py tools/tmp/libpotrf_$v.so 2>&1 | tail -1; fi
    done
done
