#!/bin/bash
# Build a second library with the phase timers of k_knn_pairs compiled in (-DKNN_PROF) — run HERE (no GPU needed):
#   bash tools/knn_phase_prof.sh build
# then on the GPU box:   gpurun -- 'python tools/knn_phase_prof.py [--shard-of W]'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p $R/tools/_prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DKNN_PROF -c $R/instant-nvr_amd/csrc/k_knn.hip -o $R/tools/_prof/k_knn_prof.o
OBJS=$(ls $R/instant-nvr_amd/build/*.o | grep -v k_knn.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/tools/_prof/k_knn_prof.o -o $R/tools/_prof/libinvr_knnprof.so
echo built $R/tools/_prof/libinvr_knnprof.so
