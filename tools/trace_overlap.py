"""Timeline summary of a rocprofv3 --kernel-trace CSV: per-kernel average, the gaps between consecutive kernels and how much kernels overlap.
usage: python tools/trace_overlap.py <*_kernel_trace.csv> [skip_first_n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in rows), key=lambda e: e[0])
ev = [e for e in ev if e[2].startswith('k_') or e[2].startswith('void k_') or 'fillBuffer' in e[2] or 'nccl' in e[2].lower()]
ev = ev[len(ev) // 3:]            # steady state (after warm-up / capture)
busy = sum(e[1] - e[0] for e in ev)
# union of intervals
u, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: u += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
u += cur_e - cur_s
span = ev[-1][1] - ev[0][0]
print('kernels %d  span %.3f ms  sum of durations %.3f ms  union %.3f ms  idle %.1f %%  overlap factor %.3f' % (len(ev), span / 1e6, busy / 1e6, u / 1e6, 100.0 * (span - u) / span, busy / u))
per = {}
for s, e, n in ev: per.setdefault(n, []).append(e - s)
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])): print('  %-42s n=%5d avg %8.1f us' % (n, len(v), sum(v) / len(v) / 1e3))
# one steady-state frame as a timeline (start / end in us relative to the frame's first kernel)
fills = [i for i, e in enumerate(ev) if 'fillBuffer' in e[2]]
if len(fills) > 4:
    a, b = fills[len(fills) // 2], fills[len(fills) // 2 + 1]
    t0 = ev[a][0]
    print('timeline of one frame:')
    for s, e, n in ev[a:b + 1]: print('  %8.1f %8.1f  %6.1f  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
