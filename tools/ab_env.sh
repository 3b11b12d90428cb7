cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6ab1
for i in 1 2; do for E in "" "INVR_ENC_NOLDS=1"; do
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$E] %.4f ms' % d['ms_per_step'], d['replay_bit_exact'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"
done; done 2>&1 | tee gpurun_out/r6ab1/ab.log
bash tools/gpu.sh trace r6trace2
