"""Timeline / per-kernel summary of the LAST evaluation in a rocprofv3 rocpd trace of bench.py.
usage: python tools/trace_eval.py trace.db [first_row last_row]"""
import sqlite3, sys, collections
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_gram' in r[0]]
ev = rows[idx[-1]:]
t0 = ev[0][1]
print("kernels in eval:", len(ev), "span ms", (max(r[2] for r in ev) - t0) / 1e6)
cls = collections.defaultdict(list)
for r in ev:
    n = r[0].split('(')[0].replace('void ', '').replace('mogp::', '')
    cls[n].append(((r[2] - r[1]) / 1e3, r[3] // max(r[4], 1)))
for n, v in cls.items():
    d = np.array([x[0] for x in v])
    print("%-28s %4d tot %8.1f us mean %7.1f" % (n, len(v), d.sum(), d.mean()))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 0)
for r in ev[lo:hi]:
    n = r[0].split('(')[0].replace('void ', '').replace('mogp::', '')
    print("%8.1f %8.1f  q%s %-24s grid %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[5], n, r[3] // max(r[4], 1)))
