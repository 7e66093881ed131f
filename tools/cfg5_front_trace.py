"""Every kernel (all queues, in start order) of the FRONT of the last configs[4] step in a rocprofv3 --kernel-trace CSV: from the step's first kernel to the first
leaf of the substitution v = L^-1 K_uf -- the K_uu / K_uf Gram builds and the K_uu factorisation.   usage: python tools/cfg5_front_trace.py <dir or csv>"""
import csv, glob, os, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(src)):
    rows.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort(key=lambda r: r[1])
leafs = [i for i, r in enumerate(rows) if r[0].startswith("k_trsm_leaf_r<false")]
# first leaf of the last group of 16
first = leafs[-16]
T = rows[first][1]
lo = T - 6_000_000
ev = [r for r in rows if lo <= r[1] <= T + 100_000]
# the step's start: the last k_phase_table before T preceded by a kernel-free gap
ph = [r for r in ev if r[0].startswith("k_phase_table")]
t0 = ph[0][1] if ph else ev[0][1]
for r in ev:
    if r[1] < t0: continue
    print("%9.1f %8.1f  q%-2d %-40s wgs %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:40], r[4]))
