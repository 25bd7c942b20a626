#!/bin/bash
# Build a libksolve variant into ab/<name>.so:   tools/mkvariant.sh <name> [-DFLAG ...]
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "${@:2}" -o ab/$1.so karpenter_core_amd/csrc/ksolve.hip
