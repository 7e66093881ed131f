"""GEMM throughput over time in the LAST evaluation of a rocprofv3 kernel trace of bench.py (cfg2): estimated flops per 500 us window
(128x128x512 per big tile; triangular-K and small-tile launches weighted 0.4) and the list of big launches with their solo-equivalent rate.
usage: python tools/gemm_rate.py <dir or csv> [--list]"""
import csv, glob, os, sys
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
def tile_shape(name):
    """k_gemm<AKM, BKM, WTM, WTN[, NWJ, NWI]> -> workgroup tile (rows, columns)"""
    a = [int(v) for v in name[name.index("<") + 1:name.index(">")].split(",")]
    nwj, nwi = (a[4] if len(a) > 4 else 2), (a[5] if len(a) > 5 else 2)
    return 16 * nwi * a[2], 16 * nwj * a[3]


W = 500.0
nb = int((max(r[2] for r in ev) - t0) / 1e3 / W) + 1
fl = np.zeros(nb)
for r in ev:
    if not r[0].startswith("k_gemm"): continue
    tm, tn = tile_shape(r[0])
    big = tm == 128 and tn == 128
    f = r[4] * tm * tn * 512 * 2 * (1.0 if big else 0.4)
    s = (r[1] - t0) / 1e3; e = (r[2] - t0) / 1e3
    for b in range(int(s // W), int(e // W) + 1):
        lo = max(s, b * W); hi = min(e, (b + 1) * W)
        if hi > lo: fl[b] += f * (hi - lo) / (e - s)
print("window start (us): TF  ", "  ".join("%d:%.0f" % (b * W, fl[b] / W / 1e6) for b in range(nb)))
if "--list" in sys.argv:
    for r in ev:
        if r[0].startswith("k_gemm") and tile_shape(r[0]) == (128, 128):
            d = (r[2] - r[1]) / 1e3
            print("%8.1f %7.1f q%d grid %5d  %5.1f TF %s" % ((r[1] - t0) / 1e3, d, r[3], r[4], r[4] * 128 * 128 * 512 * 2 / d / 1e6, r[0]))
