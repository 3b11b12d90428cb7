#!/bin/bash
# Build ablated variants of the part-MLP kernel (MLP_DBG = 1 no activations, 2 no MFMA) into scratch/ and time the frame with each.
R=${GRAFT_REPO_ROOT:-$(dirname $0)/..}; cd $R
for V in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DMLP_DBG=$V -c instant-nvr_amd/csrc/k_mlp.hip -o scratch/k_mlp_$V.o || exit 1
  OBJS=$(ls instant-nvr_amd/build/*.o | grep -v k_mlp.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS scratch/k_mlp_$V.o -o scratch/libinvr_$V.so || exit 1
done
