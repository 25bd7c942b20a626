#!/bin/bash
# Diagnostics of the pack kernel on config #3 (run on the GPU box): the shipped library's time, then the probe builds in ab/
#   ab/x_probes.so  -DKS_PROBES               per-phase cycle counters (tools/phase_profile.py prints them)
#   ab/x_cut.so     -DKS_PROBES -DKS_CUTSTATS why speculation rounds end: cyc_stage = topology read-after-record, cyc_scan = nothing in the window
#                                             accepts, cyc_evalout = order / closed-node rules, cyc_full = rounds that placed every planned pod
#   ab/x_p2.so      -DKS_PROBES -DKS_P2PROBES resolver: cyc_pop = setup cycles, cyc_stage = loop cycles, cyc_evalout = iterations
cp karpenter_core_amd/libksolve.so /tmp/keep.so
python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids | head -3
for v in x_probes x_cut x_p2; do [ -f ab/$v.so ] || continue; cp ab/$v.so karpenter_core_amd/libksolve.so; echo "== $v"; if [ $v = x_probes ]; then python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids; else python tools/p2_probe.py 2>&1 | grep -v amdgpu.ids; fi; done
cp /tmp/keep.so karpenter_core_amd/libksolve.so
