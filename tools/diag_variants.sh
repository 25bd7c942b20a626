#!/bin/bash
# Diagnostics of the pack kernel on config #3 (run on the GPU box): the shipped library's time, then the probe builds in ab/
#   ab/x_probes.so  -DKS_PROBES               per-phase cycle counters (tools/phase_profile.py prints them)
#   ab/x_cut.so     -DKS_PROBES -DKS_CUTSTATS why speculation rounds end: cyc_stage = topology read-after-record, cyc_scan = nothing in the window
#                                             accepts, cyc_evalout = order / closed-node rules, cyc_full = rounds that placed every planned pod
#   ab/x_p2.so      -DKS_PROBES -DKS_P2PROBES resolver / filter phase (raw slots, tools/p2_probe.py): cyc_evalout = resolver iterations; cycles of its
#                                             sections: cyc_stage = top of an iteration, scan_chunks - 9081 = dynamic-spread block, cyc_order = recheck of moved
#                                             candidates, cyc_new = setup + arg-min + checks, p26 = sweep / climb / bookkeeping; cyc_full = order moves;
#                                             filter phase: cyc_pop = worker 1 in total, cyc_commit = ... up to the totals, p20 = totals + lower bounds,
#                                             cyc_scan = the leader's own share (plan of the next step)
# build them with tools/mkvariant.sh <name> <flags>; tools/run_p2.sh runs whatever of them exists and writes gpurun_out/u1_<name>.log
cp karpenter_core_amd/libksolve.so /tmp/keep.so
python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids | head -3
for v in x_probes x_cut x_p2; do [ -f ab/$v.so ] || continue; cp ab/$v.so karpenter_core_amd/libksolve.so; echo "== $v"; if [ $v = x_probes ]; then python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids; else python tools/p2_probe.py 2>&1 | grep -v amdgpu.ids; fi; done
cp /tmp/keep.so karpenter_core_amd/libksolve.so
