#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
for cfg in "stamps4096 4096 8" "stamps1152 1152 8"; do
    set -- $cfg
    cp tools/tmp/lib$1.so $C/libapgp.so
    timeout 300 python tools/tmp/read_stamps.py $2 $3 > gpurun_out/stamps_$2_$3.txt 2>&1
done
cp /tmp/ship.so $C/libapgp.so
head -60 gpurun_out/stamps_4096_8.txt; echo; cat gpurun_out/stamps_1152_8.txt | head -120
