#!/bin/bash
# grid-size sweep of the persistent kernels on a 1/8 shard (env knobs INVR_G_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants --shard-of ${W:-8}"
one() { echo -n "[$*] "; env "$@" $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"; }
{
one A=1
one INVR_G_DEFORM=256; one INVR_G_DEFORM=128; one INVR_G_DEFORM=64
one INVR_G_WARP=512; one INVR_G_WARP=256
one INVR_G_ENC=1024; one INVR_G_ENC=512
one INVR_G_OCC=512; one INVR_G_OCC=256
one INVR_G_RGB=512; one INVR_G_RGB=256
one INVR_G_KNN=192; one INVR_G_KNN=128
one A=1
} > $O/out${W:-8}.txt 2>&1
cat $O/out${W:-8}.txt
