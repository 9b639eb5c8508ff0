#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter values per kernel: pmc_summary.py <counter_collection.csv> [...] -> markdown table."""
import collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import short
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
print("| kernel | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for k in sorted(agg):
    print("| %s | " % k + " | ".join("%.4g" % (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else "-" for c in names) + " |")
