"""Kernel timeline of the LAST step in a rocprofv3 --kernel-trace CSV of bench.py, any configuration: per hardware queue the kernels in time order
(start, duration in us, name, workgroups), consecutive launches of one kernel on one queue folded into one line.  A step starts at the first
k_phase_table launch after a gap.   usage: python tools/eval_timeline.py <dir or kernel_trace.csv> [min_us_to_list=20]"""
import csv, glob, os, sys, collections
src = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(src)):
    rows.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort(key=lambda r: r[1])
starts, last_end = [], -10 ** 18
for i, r in enumerate(rows):
    if r[0].startswith("k_phase_table") and r[1] - last_end > 200000:
        starts.append(i)
    last_end = max(last_end, r[2])
ev = rows[starts[-2]:starts[-1]] if len(starts) >= 2 else rows[starts[-1]:]
t0 = ev[0][1]
print("step: %d kernels, span %.1f us, sum of durations %.1f us" % (len(ev), (max(r[2] for r in ev) - t0) / 1e3, sum(r[2] - r[1] for r in ev) / 1e3))
qs = collections.defaultdict(list)
for r in ev:
    qs[r[3]].append(r)
for q, lst in sorted(qs.items(), key=lambda kv: kv[1][0][1]):
    print("\nqueue %d: %d kernels, busy %.1f us, first start %.1f, last end %.1f" % (q, len(lst), sum(r[2] - r[1] for r in lst) / 1e3, (lst[0][1] - t0) / 1e3, (max(r[2] for r in lst) - t0) / 1e3))
    i = 0
    while i < len(lst):
        j = i
        while j + 1 < len(lst) and lst[j + 1][0] == lst[i][0] and lst[j + 1][4] == lst[i][4]:
            j += 1
        dur = sum(r[2] - r[1] for r in lst[i:j + 1]) / 1e3
        if dur >= thr or j > i:
            print("  %9.1f .. %9.1f  busy %8.1f  x%-3d %-34s wgs %d" % ((lst[i][1] - t0) / 1e3, (lst[j][2] - t0) / 1e3, dur, j - i + 1, lst[i][0], lst[i][4]))
        i = j + 1
