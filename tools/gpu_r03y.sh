#!/bin/bash
# elimination builds (results wrong on purpose): what bounds the half-work tiles of a partial last row block
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
O=gpurun_out/${OUT:-ab_r03y.txt}
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ${VARIANTS:-ship e_nopark e_img4 e_nodma e_nopredmfma ship}; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --elim --inverse-only 2>&1 | grep -E "N=" >> $O
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-100 $O
