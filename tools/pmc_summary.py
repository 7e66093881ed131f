"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, csv output).
usage: python tools/pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv n_evals > profiles/xxx_pmc.csv
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide
coalesced reads -> doubled here; WRITE_SIZE is taken as is.  Calibration inside this very run: k_gram writes the lower triangle of
the N x N Gram matrix once (4 N (N+1) bytes) and k_moments reads the lower triangle of K^-1 once -- see the two 'calib' columns."""
import csv, collections, json, sys

def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mogp::", "")
            a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024.0
    return agg

f, w, nev = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
print('"kernel","launches_per_eval","fetch_bytes_per_eval(corrected x2)","write_bytes_per_eval","bytes_per_launch"')
tot = collections.defaultdict(float)
for n in sorted(f, key=lambda k: -f[k][1]):
    calls = f[n][0] / nev
    fb, wb = 2.0 * f[n][1] / nev, w.get(n, [0, 0.0])[1] / nev
    print('"%s",%.1f,%.4g,%.4g,%.4g' % (n, calls, fb, wb, (fb + wb) / max(calls, 1e-9)))
    if n.startswith("k_gemm"):
        tot["launches"] += calls; tot["fetch"] += fb; tot["write"] += wb
print('"k_gemm (all variants)",%.1f,%.4g,%.4g,%.4g' % (tot["launches"], tot["fetch"], tot["write"], (tot["fetch"] + tot["write"]) / tot["launches"]))
json.dump({"kernel": "k_gemm (all variants)", "launches_per_eval": tot["launches"], "fetch_bytes_per_eval": tot["fetch"],
           "write_bytes_per_eval": tot["write"], "bytes_per_launch": (tot["fetch"] + tot["write"]) / tot["launches"],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 per the gfx950 correction"},
          open(sys.argv[4], "w"), indent=1) if len(sys.argv) > 4 else None
