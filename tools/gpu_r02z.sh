#!/bin/bash
cp approxposterior_amd/csrc/libapgp.so /tmp/keep.so
for v in ship upd2 ship upd2; do
  if [ $v = ship ]; then cp /tmp/keep.so approxposterior_amd/csrc/libapgp.so; else cp tools/tmp/lib$v.so approxposterior_amd/csrc/libapgp.so; fi
  echo "--- $v"; bash tools/gpu_r02u.sh | grep potrf
done
cp /tmp/keep.so approxposterior_amd/csrc/libapgp.so
