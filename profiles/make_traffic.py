#!/usr/bin/env python3
"""HBM traffic per kernel LAUNCH from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains).
usage: make_traffic.py <fetch_dir> <write_dir> <out.json> <streams> <blocks> <types>
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of the bytes read (MI355X_MICROARCH.md, HBM
section), for every access pattern of this project -- calibrated with known-byte kernels, profiles/r02_hbm_calibration.json:
16 / 4 / 2 B per lane coalesced: x2.00, lane-per-row 16 B loads (the biquad passes): x1.90 (5 % over-fetch); WRITE_SIZE is
exact for coalesced stores of any width and reports 2.76x the bytes of lane-per-row 16 B stores (partial-line writes:
real extra traffic).  total_hbm_bytes_per_batch = sum over kernels of hbm_bytes x launches per batch (one batch = one
frontend launch)."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    acc = collections.defaultdict(list)
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tfrec::", "")
        names[r["Dispatch_Id"]] = nm if nm.startswith("spec_biquad_kernel") else nm.split("<")[0]  # the six biquad passes apart
    for d_id, v in per_dispatch.items():
        acc[names[d_id]].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"note": __doc__.strip().split("\n", 2)[2], "streams": int(sys.argv[4]), "blocks": int(sys.argv[5]),
       "types": int(sys.argv[6]), "kernels": {}}
for k in sorted(fetch):
    if k.startswith("__"):
        continue
    out["kernels"][k] = {"fetch_size_kb_raw": round(fetch[k], 1), "write_size_kb_raw": round(write.get(k, 0.0), 1),
                         "hbm_bytes": int((2 * fetch[k] + write.get(k, 0.0)) * 1024), "dispatches_profiled": n[k]}
nb = max(1, n.get("frontend_kernel", 1))
total = 0
for k, v in out["kernels"].items():
    v["launches_per_batch"] = round(v["dispatches_profiled"] / nb, 2)
    total += v["hbm_bytes"] * v["dispatches_profiled"] / nb
out["total_hbm_bytes_per_batch"] = int(total)
out["algorithmic_bytes_per_batch"] = 2 * int(sys.argv[4]) * int(sys.argv[5]) * 32768
out["traffic_ratio"] = round(total / out["algorithmic_bytes_per_batch"], 3)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
