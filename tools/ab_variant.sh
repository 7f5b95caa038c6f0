#!/bin/bash
# Same-box A/B of sweep-kernel variants against the in-tree libapgp.so (boxes differ by +-1 %, so
# variants are only ever compared inside one gpurun call).  Build the variant libraries first:
#   tools/ab_variant.sh build NAME /path/to/variant_sweep.hip ["-DFLAG ..."]   (here, cross-compiles)
# then on the GPU box:   gpurun -- 'bash tools/ab_variant.sh run NAME1 NAME2 ...'
set -e
C=approxposterior_amd/csrc
if [ "$1" = build ]; then
    mkdir -p tools/tmp
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$C -Iinclude $4 -c "$3" -o /tmp/ab_$2.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp/lib$2.so $C/gram.o $C/linalg.o /tmp/ab_$2.o $C/grad.o $C/potrf.o $C/ensemble.o
    exit 0
fi
shift
cp $C/libapgp.so /tmp/ab_keep.so
trap 'cp /tmp/ab_keep.so '$C'/libapgp.so' EXIT
for rep in 1 2 3; do
    for v in "$@" ship; do
        if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
        echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
    done
done
