#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel, averaged over dispatches."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    d = max(len(n[k]), 1)
    print("%-44s dispatches=%d  " % (k[:44], d) + "  ".join("%s=%.4g" % (c, x / d) for c, x in sorted(v.items())))
