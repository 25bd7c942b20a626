cp karpenter_core_amd/libksolve.so /tmp/keep.so
python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids | head -3
for v in x_probes x_cut x_p2; do cp ab/$v.so karpenter_core_amd/libksolve.so; echo "== $v"; if [ $v = x_probes ]; then python tools/phase_profile.py 2>&1 | grep -v amdgpu.ids; else python tools/p2_probe.py 2>&1 | grep -v amdgpu.ids; fi; done
cp /tmp/keep.so karpenter_core_amd/libksolve.so
