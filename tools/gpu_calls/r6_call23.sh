#!/bin/bash
# round 6, call 23: what the no_T fix (and the barrier in ks_pack's failure path) cost the kernel -- the build before it (prefix) against the product, interleaved
bash tools/gpu_calls/r6_ab.sh prefix product prefix product prefix product 2>&1 | cut -c1-120
