"""Per-kernel summary (count / total / avg / min / max / %) of a rocprofv3 rocpd SQLite trace -- the same table
`rocprofv3 --stats` prints.  Usage: python tools/rocpd_stats.py trace.db > profiles/xxx_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    a = agg.setdefault(name, [0, 0, 10 ** 18, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (name.replace('"', "'"), a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))
