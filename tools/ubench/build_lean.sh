#!/bin/bash
# Build the variants of tools/ubench/lean_loop.hip (and wave_costs) HERE, in the build container: the binaries travel to the GPU box with the snapshot
# (git-ignored, not gpurun-ignored); tools/gpu_calls/r5_ubench.sh runs them there in one call of about half a GPU-minute.
cd "$(dirname "$0")"
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w"
rm -f lean_loop_*
b() { out=$1; shift; $H "$@" -o lean_loop_$out lean_loop.hip || exit 1; }
$H -o wave_costs wave_costs.hip || exit 1
b 00 -DNPL=4                                                '-DVARIANT=", base (ladder commit, counter items)"'
b 01 -DNPL=4 -DCOMMIT_SELECT                                '-DVARIANT=", commit without control flow"'
b 02 -DNPL=4 -DITEM_MASKS                                   '-DVARIANT=", hostname items as per-counter slot masks"'
b 03 -DNPL=4 -DITEM_MASKS -DCOMMIT_SELECT                   '-DVARIANT=", masks + select commit"'
b 04 -DNPL=4 -DITEM_MASKS -DCOMMIT_SELECT -DR32             '-DVARIANT=", masks + select commit + 32-bit resources"'
b 05 -DNPL=3 -DITEM_MASKS -DCOMMIT_SELECT -DR32             '-DVARIANT=", masks + select commit + 32-bit resources"'
b 06 -DNPL=2 -DITEM_MASKS -DCOMMIT_SELECT -DR32             '-DVARIANT=", masks + select commit + 32-bit resources"'
b 07 -DNPL=8 -DITEM_MASKS -DCOMMIT_SELECT -DR32             '-DVARIANT=", masks + select commit + 32-bit resources"'
b 08 -DNPL=4 -DITEM_MASKS -DCOMMIT_SELECT -DR32 -DNW=4      '-DVARIANT=", masks + select commit + 32-bit resources"'
b 09 -DNPL=4 -DITEM_MASKS -DCOMMIT_SELECT -DR32 -DCONST_POD '-DVARIANT=", masks + select commit + 32-bit, loop-invariant pod"'
b 10 -DNPL=4 -DITEM_MASKS -DCOMMIT_SELECT -DR32 -DNO_XCHG   '-DVARIANT=", masks + select commit + 32-bit, no exchange"'
b 11 -DNPL=4 -DNO_COMMIT                                    '-DVARIANT=", no commit"'
ls -la lean_loop_* wave_costs | awk '{print $5, $9}'
