#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (rocprofv3, KB) of profiles/ubench/hbm_calib's known-byte kernels -> factor = true bytes / counted
bytes per access pattern.  usage: make_calibration.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys

KNOWN = 2 << 30


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    per = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    acc = collections.defaultdict(list)
    for k, v in per.items():
        acc[name[k]].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"known_bytes_per_kernel": KNOWN, "patterns": {}}
for k in ("rd16", "rd4", "rd2", "rdrow"):
    out["patterns"][k] = {"fetch_size_kb": round(fetch.get(k, 0), 1), "fetch_factor": round(KNOWN / 1024 / fetch[k], 4) if fetch.get(k) else None}
for k in ("wr16", "wr4", "wr2", "wrrow"):
    out["patterns"][k] = {"write_size_kb": round(write.get(k, 0), 1), "write_factor": round(KNOWN / 1024 / write[k], 4) if write.get(k) else None,
                          "fetch_size_kb_of_a_pure_write": round(fetch.get(k, 0), 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
