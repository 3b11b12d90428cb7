"""One steady-state unit of work of a rocprofv3 --kernel-trace CSV as a timeline: the kernels between two consecutive launches of a
delimiter kernel (start / end / duration in us relative to the first, queue / stream ids), plus idle time and the overlap factor.
usage: python tools/trace_timeline.py <*_kernel_trace.csv> <delimiter substring, e.g. k_adam> [which occurrence, default the middle]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
delim = sys.argv[2]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if e[2] == delim or e[2].startswith(delim + '<')]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) // 2
a, b = marks[k] + 1, marks[k + 1] + 1
win = ev[a:b]
t0 = win[0][0]
busy = sum(e[1] - e[0] for e in win)
u, cs, ce = 0, None, None
for s, e, *_ in win:
    if ce is None or s > ce:
        if ce is not None: u += ce - cs
        cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
span = win[-1][1] - win[0][0]
print('unit %d: %d kernels, span %.1f us, union %.1f us (idle %.1f us = %.1f %%), sum of durations %.1f us (overlap factor %.2f)' % (k, len(win), span / 1e3, u / 1e3, (span - u) / 1e3, 100.0 * (span - u) / span, busy / 1e3, busy / u))
prev_end = None
for s, e, n, q, st in win:
    gap = '' if prev_end is None else ('%+7.1f' % ((s - prev_end) / 1e3))
    print('  %9.1f %9.1f %8.1f  gap %8s  q%s s%s  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, q, st, n))
    prev_end = e if prev_end is None else max(prev_end, e)
