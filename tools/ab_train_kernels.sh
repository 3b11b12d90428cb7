#!/bin/bash
# gpurun -- 'bash tools/ab_train_kernels.sh <tag> libA libB ...': the training iteration (tools/train_bench.py) under rocprofv3 with each library
# build on ONE box: iteration time + the average / maximum duration of the backward's kernels (boxes differ by 10-30 % on these
# latency-bound kernels: only same-box comparisons count)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift; OUT=$R/gpurun_out/$T; mkdir -p $OUT
for rep in 1 2; do for L in "$@"; do
  N=$(basename $L .so)_$rep
  rm -rf /tmp/trk_$N
  INVR_LIB_PATH=$R/$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trk_$N -o tr -- python $R/tools/train_bench.py --iters 30 > $OUT/$N.log 2>&1
  echo "== $L (run $rep): $(grep iteration $OUT/$N.log | cut -c1-60)"
  python - <<PY
import csv, glob
f = glob.glob('/tmp/trk_$N/**/tr_kernel_stats.csv', recursive=True)[0]
want = ('k_part_encode_bwd', 'k_wgrad', 'k_part_mlp_bwd', 'k_deform_bwd', 'k_deform_slice_bwd', 'k_adam(')
for r in csv.DictReader(open(f)):
    if any(w in r['Name'] for w in want):
        print('   %-44s %5s calls  avg %8.1f  max %8.1f us' % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  F=$(find /tmp/trk_$N -name "*kernel_trace.csv" | head -1); python $R/tools/trace_timeline.py $F k_adam 12 > $OUT/timeline_$N.txt; head -1 $OUT/timeline_$N.txt
  rm -rf /tmp/trk_$N
done; done 2>&1 | tee $OUT/ab.log
