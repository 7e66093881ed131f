"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (Calls / average / min / max / total), the table `rocprofv3 --stats` prints.
usage: python tools/ktrace.py <dir or kernel_trace.csv> [--csv out.csv]
Collect with:  cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d <dir> -o p -- python bench.py ...
(rocprofv3 of this image can crash in its own exit handler AFTER the trace is written; the CSV is complete.)"""
import csv, glob, collections, os, sys

src = sys.argv[1]
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0, 10 ** 18, 0, []])
for r in csv.DictReader(open(src)):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"]
    if "k_flow" in name and "Grid_Size_X" in r:      # two instances per evaluation (bulk CUs / the private stream's after the last chain kernel): keep them apart
        name += " [grid %s]" % r["Grid_Size_X"]
    a = agg[name]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d); a[4].append(d)
tot = sum(a[1] for a in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
if "--csv" in sys.argv:
    with open(sys.argv[sys.argv.index("--csv") + 1], "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage","MedianNs"\n')
        for k, a in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f,%d\n' % (k.replace('"', "'"), a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot, sorted(a[4])[len(a[4]) // 2]))
for k, a in rows[:int(os.environ.get("KTRACE_TOP", "24"))]:
    print("%-66s %6d avg %9.1f us  min %8.1f  max %8.1f  %5.1f%%" % (k.split("(")[0].replace("void ", "").replace("mogp::", "")[:66], a[0], a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
