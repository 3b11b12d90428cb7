#!/bin/bash
# usage: tools/asm.sh k_warp  -> /tmp/k_warp.s + register summary
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o /tmp/$1.s $(dirname $0)/../instant-nvr_amd/csrc/$1.hip -I $(dirname $0)/../include 2>&1 | grep -v "warning\|^$" | head
grep "^_Z.*:\|NumVgprs\|NumAgprs\|ScratchSize\|Occupancy\|LDSByteSize" /tmp/$1.s | sed 's/ *; @.*//'
