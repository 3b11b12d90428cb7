"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) as a kernel stats table (markdown/CSV)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else 'kernel_name'
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("| kernel | calls | total_ms | avg_us | min_us | max_us | pct |")
print("|---|---|---|---|---|---|---|")
for n,c,s,a,mn,mx in rows[:40]:
    print(f"| {n[:90]} | {c} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.1f} |")
