#!/bin/bash
# round 4, first GPU call: the three prepared A/Bs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
bash $R/tools/ab.sh instant-nvr_amd/libinvr.so scratch/libinvr_knn_b128.so 2 > $O/ab_knn_b128.txt 2>&1
bash $R/tools/ab.sh instant-nvr_amd/libinvr.so scratch/libinvr_enc_rcpnorm.so 2 > $O/ab_enc_rcpnorm.txt 2>&1
bash $R/tools/ab.sh instant-nvr_amd/libinvr.so scratch/libinvr_knn_tsize.so 2 --shard-of 8 > $O/ab_knn_tsize.txt 2>&1
cat $O/*.txt
