#!/bin/bash
# An experiment build of the library beside the product one:  tools/build_variant.sh <name> -DEXP_...   -> variants/libinvr_<name>.so
# (git-ignored, travels to the GPU box with gpurun; A/B against the product build: bash tools/gpu.sh ab <tag> instant-nvr_amd/libinvr.so variants/libinvr_<name>.so 2)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
O=$R/variants/obj_$N; rm -rf $O; mkdir -p $O
SRC=${SRC:-$R/instant-nvr_amd/csrc}          # (SRC=<dir>: another source tree, e.g. `git archive HEAD instant-nvr_amd/csrc include | tar -x -C /tmp/base` for the committed baseline)
for f in $SRC/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function "$@" -c $f -o $O/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o $R/variants/libinvr_$N.so
echo $R/variants/libinvr_$N.so
