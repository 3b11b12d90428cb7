cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2c; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o tr -- python $R/tools/train_bench.py --iters 20 > $OUT/tr.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/tr/tr_kernel_stats.csv')))
for r in rows[:30]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%9.1f'%(float(r['AverageNs'])/1e3), '%9.1f'%(float(r['MaxNs'])/1e3), r['Percentage'])
PY
find $OUT/tr -name "*kernel_trace.csv" | head -1 | xargs -I{} python - {} <<PY
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].split('(')[0]
    if 'k_part_encode_bwd' in n or 'k_wgrad' in n or 'k_part_mlp_bwd' in n: d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(k, [round(x) for x in v[-15:]])
PY
find $OUT/tr -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for v in "INVR_ENCB_NOATOM=1" "INVR_ENCB_TP=64" "INVR_ENCB_TP=32"; do echo == $v; env $v python $R/tools/train_bench.py --iters 20 2>&1 | grep -E "iteration|synchron"; done
