#!/bin/bash
# Build an experiment variant of libksolve.so into karpenter_core_amd/_variants/<name>/ (git-ignored; travels to the GPU box):  tools/mkvariant_dir.sh <name> [-DFLAG ...]
# tools that take KS_VARIANT=<name> (tools/phase_profile_rr.py, tools/win_profile.py) load the libraries from there instead of the product's.
d=karpenter_core_amd/_variants/$1; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "${@:2}" -o $d/libksolve.so karpenter_core_amd/csrc/ksolve.hip || exit 1
cp karpenter_core_amd/libkshost.so $d/
