#!/bin/bash
# same-box A/B of the training iteration under two environment settings:  bash tools/ab_train.sh "<envA>" "<envB>" <rounds>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in $(seq $3); do
  for E in "$1" "$2"; do
    echo "[$E] $(env $E python $R/tools/train_bench.py --iters 60 2>&1 | grep -E 'iteration|synchronised' | tr '\n' ' ')"
  done
done
