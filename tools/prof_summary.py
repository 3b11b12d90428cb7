"""Markdown summary of one tools/prof_all.sh result directory (gpurun_out/<tag>): bench line, rocprofv3
--kernel-trace --stats table, per-kernel PMC averages and derived utilisations; also writes
profiles/hbm_traffic_per_launch.json (FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, bytes per launch).
usage: python tools/prof_summary.py gpurun_out/r1f profiles/r1f_kernel_stats_and_pmc.md "<title>" """
import collections
import csv
import glob
import json
import os
import sys

R, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
out = ['# ' + title + '\n']
out.append('Commands (MI355X box, `bash tools/prof_all.sh <tag>`): `python bench.py --steps 20 --warmup 5` (default flags, hipGraph replay), then\n'
           '`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-graph --steps 10 --warmup 3`, then one\n'
           '`rocprofv3 --kernel-trace --pmc <group> --output-format csv` run per counter group (`--steps 2 --warmup 1`; eager launches).\n')
out.append('## bench.py (default flags) JSON line\n```\n' + open(R + '/bench.json').read().strip() + '\n```\n')
rows = list(csv.DictReader(open(R + '/trace/trace_kernel_stats.csv')))
short = lambda n: n.replace('void ', '').split('(')[0][:80]
out.append('## kernel-trace stats (16 renders = 3 warm-up + 10 timed + 3 stage-profile frames)\n| kernel | calls | total_ms | avg_us | min_us | max_us | pct |\n|---|---|---|---|---|---|---|')
for r in rows[:24]:
    out.append('| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
                                                              float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in [x for d in ('fetch', 'write', 'sq', 'mfma') for x in glob.glob(R + '/' + d + '/*_counter_collection.csv')]:
    for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
keys = sorted({c for k in agg for c in agg[k]})
sel = sorted(k for k in agg if k.startswith('k_'))
out.append('\n## PMC, average per dispatch (FETCH_SIZE/WRITE_SIZE in KiB as reported; GRBM_GUI_ACTIVE is summed over the 8 XCDs)\n| kernel | dispatches | '
           + ' | '.join(keys) + ' |\n|---|---|' + '---|' * len(keys))
for k in sel:
    n = max(len(v) for v in agg[k].values())
    out.append('| %s | %d | ' % (k, n) + ' | '.join(('%.4g' % (sum(agg[k][c]) / len(agg[k][c]))) if c in agg[k] else '-' for c in keys) + ' |')
avg = lambda k, c: sum(agg[k][c]) / len(agg[k][c]) if c in agg[k] else float('nan')
dur = {short(r['Name']): float(r['AverageNs']) / 1e3 for r in rows}
out.append('\n## Derived (per launch)\nVALU busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles) / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8).\n\n'
           '| kernel | avg_us | HBM/Infinity-Cache read GB (FETCH_SIZE x2, guide gfx950 correction) | write GB | L2 hit % | VALU busy % | MFMA busy % | WAIT_INST_ANY % of wave cycles | VALU instr. per wave |\n|---|---|---|---|---|---|---|---|---|')
traffic, counters = {}, {}
for k in sel:
    fetch = avg(k, 'FETCH_SIZE') * 1024 * 2 / 1e9
    wr = avg(k, 'WRITE_SIZE') * 1024 / 1e9
    hit, miss = avg(k, 'TCC_HIT_sum'), avg(k, 'TCC_MISS_sum')
    gui = avg(k, 'GRBM_GUI_ACTIVE') / 8
    valu = avg(k, 'SQ_ACTIVE_INST_VALU') * 4 / 1024 / gui * 100 if gui == gui and gui > 0 else float('nan')
    mfma = avg(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / gui * 100 if gui == gui and gui > 0 else float('nan')
    wait = avg(k, 'SQ_WAIT_INST_ANY') / avg(k, 'SQ_WAVE_CYCLES') * 100
    out.append('| %s | %.1f | %.3f | %.3f | %.0f | %.0f | %.0f | %.0f | %.0f |' % (k, dur.get(k, float('nan')), fetch, wr, 100 * hit / (hit + miss) if hit + miss > 0 else float('nan'),
                                                                         valu, mfma, wait, avg(k, 'SQ_INSTS_VALU') / avg(k, 'SQ_WAVES')))
    t = (avg(k, 'FETCH_SIZE') * 2 + avg(k, 'WRITE_SIZE')) * 1024
    if t == t:
        traffic[k] = int(t)
    rnd = lambda x, n=3: round(x, n) if x == x else None
    counters[k] = {'avg_us': rnd(dur.get(k, float('nan')), 1), 'read_GB': rnd(fetch), 'write_GB': rnd(wr),
                   'l2_hit': rnd(hit / (hit + miss)) if hit + miss > 0 else None, 'valu_busy': rnd(valu / 100), 'mfma_busy': rnd(mfma / 100),
                   'wait_inst_any_of_wave_cycles': rnd(wait / 100), 'valu_instr_per_wave': rnd(avg(k, 'SQ_INSTS_VALU') / avg(k, 'SQ_WAVES'), 0)}
# ---- the 64-byte-row encoder (bench.py --full-rows: what a training forward and eval_row_sums False read) ----
if os.path.exists(R + '/fr_trace/trace_kernel_stats.csv'):
    fr_rows = list(csv.DictReader(open(R + '/fr_trace/trace_kernel_stats.csv')))
    fr_dur = {short(r['Name']): float(r['AverageNs']) / 1e3 for r in fr_rows}
    fr = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(R + '/fr_fetch/*_counter_collection.csv') + glob.glob(R + '/fr_write/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            fr[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    favg = lambda k, c: sum(fr[k][c]) / len(fr[k][c]) if c in fr[k] else float('nan')
    out.append('\n## `--full-rows` frame (64-byte table rows: k_part_encode_rows_all / k_part_encode), per launch\n'
               '| kernel | avg_us | read GB (FETCH_SIZE x2) | write GB | L2 hit % | VALU busy % | read GB/s |\n|---|---|---|---|---|---|---|')
    for k in sorted(fr):
        if 'encode' not in k:
            continue
        fetch = favg(k, 'FETCH_SIZE') * 1024 * 2 / 1e9
        wr = favg(k, 'WRITE_SIZE') * 1024 / 1e9
        hit, miss = favg(k, 'TCC_HIT_sum'), favg(k, 'TCC_MISS_sum')
        gui = favg(k, 'GRBM_GUI_ACTIVE') / 8
        valu = favg(k, 'SQ_ACTIVE_INST_VALU') * 4 / 1024 / gui * 100 if gui == gui and gui > 0 else float('nan')
        d = fr_dur.get(k, float('nan'))
        out.append('| %s | %.1f | %.3f | %.3f | %.0f | %.0f | %.0f |' % (k, d, fetch, wr, 100 * hit / (hit + miss) if hit + miss > 0 else float('nan'), valu, fetch / (d * 1e-6) if d == d else float('nan')))
        counters['full_rows:' + k] = {'avg_us': round(d, 1) if d == d else None, 'read_GB': round(fetch, 3) if fetch == fetch else None,
                                     'write_GB': round(wr, 3) if wr == wr else None}
open(dst, 'w').write('\n'.join(out) + '\n')
dg = open(R + '/csrc_digest.txt').read().strip() if os.path.exists(R + '/csrc_digest.txt') else None
counters['_meta'] = {'csrc_digest': dg, 'source': R}          # bench.py flags counters measured on other kernel sources (counters_stale)
json.dump(traffic, open(os.path.join(os.path.dirname(dst), 'hbm_traffic_per_launch.json'), 'w'), indent=1)
json.dump(counters, open(os.path.join(os.path.dirname(dst), 'kernel_counters.json'), 'w'), indent=1)
print('\n'.join(out[-20:]))
