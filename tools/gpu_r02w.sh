#!/bin/bash
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/pytest_r02w.txt
bash tools/gpu_r02u.sh | grep -v potrf
cp approxposterior_amd/csrc/libapgp.so /tmp/keep.so
cp tools/tmp/libsyrkl2.so approxposterior_amd/csrc/libapgp.so
echo "--- syrk with every chunk re-reading the first (L2-resident operands)"
bash tools/gpu_r02u.sh | grep syrk
cp /tmp/keep.so approxposterior_amd/csrc/libapgp.so
