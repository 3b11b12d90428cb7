#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
{ bash $R/tools/ab.sh instant-nvr_amd/libinvr.so scratch/libinvr_cull_occ4.so 2
  bash $R/tools/ab.sh instant-nvr_amd/libinvr.so scratch/libinvr_cull_occ4.so 1 --shard-of 8; } > $O/out.txt 2>&1
cat $O/out.txt
