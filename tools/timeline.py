"""Timeline of the LAST evaluation in a rocprofv3 --kernel-trace CSV of bench.py (one evaluation = from a k_gram launch to the next):
per hardware queue busy time, outer-block periods of the fused schedule (k_wkk launches), and the critical queue's kernel list.
usage: python tools/timeline.py <dir or kernel_trace.csv> [--list]"""
import csv, glob, os, sys, collections
import numpy as np

src = sys.argv[1]
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(src)):
    rows.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort(key=lambda r: r[1])
idx, last_gram = [], -10 ** 18
for i, r in enumerate(rows):                      # an evaluation starts at its first Gram launch (the strip kernel and the general one follow each other)
    if r[0].startswith("k_gram"):
        if r[1] - last_gram > 1000000: idx.append(i)
        last_gram = r[1]
ev = rows[idx[-2]:idx[-1]] if len(idx) >= 2 else rows[idx[-1]:]
t0 = ev[0][1]
span = (max(r[2] for r in ev) - t0) / 1e3
print("kernels in eval: %d, span %.1f us" % (len(ev), span))
cls = collections.defaultdict(list)
for r in ev:
    cls[r[0]].append((r[2] - r[1]) / 1e3)
for n, v in sorted(cls.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s %4d  total %8.1f us  mean %7.1f" % (n, len(v), sum(v), np.mean(v)))
busy = collections.defaultdict(float); last = collections.defaultdict(float)
for r in ev:
    busy[r[3]] += (r[2] - r[1]) / 1e3; last[r[3]] = max(last[r[3]], (r[2] - t0) / 1e3)
print("busy us per queue", {q: round(b) for q, b in busy.items()}, "| last end", {q: round(b) for q, b in last.items()})
mark = "k_chain" if any(r[0].startswith("k_chain") for r in ev) else "k_wkk"      # one per outer block either way
wk = [(r[1] - t0) / 1e3 for r in ev if r[0].startswith(mark)]
if wk:
    print("outer-block (%s) start times us:" % mark, [round(x) for x in wk])
    print("periods us:", [round(b - a) for a, b in zip(wk, wk[1:])])
ints = sorted((r[1], r[2]) for r in ev if r[0].startswith("k_gemm"))
union = 0; cur_s, cur_e = ints[0]
for s, e in ints[1:]:
    if s > cur_e: union += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
union += cur_e - cur_s
print("k_gemm: sum %.0f us, union %.0f us" % (sum(e - s for s, e in ints) / 1e3, union / 1e3))
if "--list" in sys.argv:
    q = max(busy, key=lambda k: sum(1 for r in ev if r[3] == k and (r[0].startswith("k_leaf") or r[0].startswith("k_chain"))))
    for r in ev:
        if r[3] == q: print("%9.1f %8.1f  %-26s grid %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0], r[4]))
