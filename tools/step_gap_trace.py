"""The seam between two training steps of the headline configuration in a rocprofv3 --kernel-trace CSV of bench.py: every kernel (all queues, start order) from the
LAST moment kernel of a step to the dataflow kernel's launch of the next -- what the device does while the host turns moments into the next term table.
usage: python tools/step_gap_trace.py <dir or csv> [which step from the end = 2]"""
import csv, glob, os, sys
src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(src)):
    rows.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort(key=lambda r: r[1])
moms = [i for i, r in enumerate(rows) if r[0].startswith("k_moments")]
i0 = moms[-back]
t0 = rows[i0][1]
flows = [r for r in rows if r[0].startswith("k_flow") and r[4] > 100]
gaps = []
for a in moms[2:-1]:
    nxt = [f for f in flows if f[1] > rows[a][2]]
    if nxt: gaps.append((nxt[0][1] - rows[a][2]) / 1e3)
print("end of the moment kernel -> start of the next step's dataflow kernel, per step (us):", " ".join("%.0f" % g for g in gaps))
for r in rows[i0:]:
    print("%9.1f %8.1f  q%-2d %-44s wgs %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:44], r[4]))
    if r[0].startswith("k_flow") and r[4] > 100 and r[1] > t0: break
