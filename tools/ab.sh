#!/bin/bash
# A/B timing of prebuilt libksolve variants (ab/*.so) on ONE box: tools/ab.sh [pods]
cp karpenter_core_amd/libksolve.so /tmp/libksolve_keep.so
for rep in 1 2; do
  for v in ab/t_*.so; do
    cp "$v" karpenter_core_amd/libksolve.so
    echo -n "$v  "; python tools/phase_profile.py "$@" 2>&1 | grep kernel_ms
  done
done
cp /tmp/libksolve_keep.so karpenter_core_amd/libksolve.so
