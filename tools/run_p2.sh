#!/bin/bash
# On the GPU box: run the probe builds found in ab/ (tools/mkvariant.sh; slot meanings in tools/diag_variants.sh) on config #3, one log each.
cp karpenter_core_amd/libksolve.so /tmp/keep.so
for v in x_p2 x_cut; do [ -f ab/$v.so ] || continue; cp ab/$v.so karpenter_core_amd/libksolve.so; python tools/p2_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/u1_$v.log; done
if [ -f ab/x_probes.so ]; then cp ab/x_probes.so karpenter_core_amd/libksolve.so; python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/u1_x_probes.log; fi
cp /tmp/keep.so karpenter_core_amd/libksolve.so
cat gpurun_out/u1_x_p2.log gpurun_out/u1_x_probes.log
