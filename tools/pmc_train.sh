#!/bin/bash
# gpurun -- 'bash tools/pmc_train.sh <tag>': PMC passes over the training iteration (tools/train_bench.py), per-kernel averages of the backward's kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; OUT=$R/gpurun_out/$T; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -o sq -- python $R/tools/train_bench.py --iters 6 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/mf -o mf -- python $R/tools/train_bench.py --iters 6 > $OUT/mf.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/*_counter_collection.csv') + glob.glob('$OUT/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].replace('void ', '').split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
want = ('k_part_mlp_bwd', 'k_deform_bwd', 'k_part_encode_bwd', 'k_wgrad', 'k_deform_slice_bwd', 'k_knn_pairs', 'k_part_encode_rows_all', 'k_pair_term_fwd', 'k_train_terms')
keys = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'GRBM_GUI_ACTIVE']
print('%-40s %s' % ('kernel (max over launches)', ' '.join('%12s' % k[-12:] for k in keys)))
for k in sorted(agg):
    if any(k.startswith(w) for w in want):
        print('%-40s %s' % (k, ' '.join('%12.4g' % (max(agg[k][c]) if c in agg[k] else float('nan')) for c in keys)))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*_counter_collection.csv" -size +20M -delete
