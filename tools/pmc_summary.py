"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one directory per pass)."""
import csv, sys, glob, os, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '*', '*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
keys = sorted({c for k in agg for c in agg[k]})
sel = [k for k in agg if k.startswith('k_')]
print('| kernel | dispatches | ' + ' | '.join(keys) + ' |')
print('|---|---|' + '---|' * len(keys))
for k in sorted(sel):
    n = max(len(v) for v in agg[k].values())
    print('| %s | %d | ' % (k, n) + ' | '.join(('%.4g' % (sum(agg[k][c]) / len(agg[k][c]))) if c in agg[k] else '-' for c in keys) + ' |')
