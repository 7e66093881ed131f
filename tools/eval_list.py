"""Every launch of the last evaluation in a rocprofv3 kernel trace: start, duration, queue, kernel.  usage: python tools/eval_list.py <dir> <first kernel substring>"""
import csv, glob, sys
d, first = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
ev = [i for k, i in enumerate(starts) if k == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[starts[k - 1]]["End_Timestamp"]) > 200_000]
seg = rows[ev[-2]:ev[-1]]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%8.1f %7.1f q%-2s %s" % (s, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "").replace("mogp::", "").split("(")[0][:40]))
