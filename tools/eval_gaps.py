"""Where one evaluation's time goes when it is NOT GEMM-bound: from a rocprofv3 kernel trace, the last evaluation's kernels grouped into
phases of similar kernels, with the busy time per phase and the number of launches -- to find chains of small dependent launches.
usage: python tools/eval_gaps.py <trace dir> <first kernel of an evaluation (substring)>"""
import csv, glob, sys
d, first = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# an evaluation starts at a `first` kernel that follows a gap of > 1 ms
ev = [i for k, i in enumerate(starts) if k == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[starts[k - 1]]["End_Timestamp"]) > 1_000_000]
a, b = ev[-2], ev[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in seg)
print("evaluation: %d launches, span %.2f ms" % (len(seg), (end - t0) / 1e6))
def short(n):
    return n.replace("void ", "").replace("mogp::", "").split("(")[0][:34]
cur, out = None, []
for r in seg:
    n = short(r["Kernel_Name"])
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    big = (e - s) > 150.0
    key = n if big else "small"
    if cur and cur["key"] == key and (not big):
        cur["n"] += 1; cur["busy"] += e - s; cur["end"] = max(cur["end"], e); cur["names"].add(n)
    else:
        cur = dict(key=key, n=1, busy=e - s, start=s, end=e, names={n})
        out.append(cur)
for c in out:
    print("%9.1f -> %9.1f us  %-8s launches %4d  busy %8.1f us  wall %8.1f us  %s" % (c["start"], c["end"], "BIG" if c["key"] != "small" else "small", c["n"], c["busy"], c["end"] - c["start"],
                                                                                  ",".join(sorted(c["names"]))[:90]))
