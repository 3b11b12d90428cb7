#!/bin/bash
# tools/isa_dump.sh <file.hip> <out.s>: the gfx950 assembly of one kernel source with the product's flags (instant-nvr_amd/build.py),
# for before / after comparisons of a source change (diff <(grep -v '^\s*[;.]' a.s) <(grep -v '^\s*[;.]' b.s)).
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -S --cuda-device-only "$1" -o "$2"
