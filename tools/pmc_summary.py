#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs by kernel: mean per dispatch of each counter.
usage: python tools/pmc_summary.py dir [dir ...]"""
import csv, glob, sys, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("scm::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in sorted(agg.items()):
    if not any(k in name for k in ("gemm", "wgrad", "favor", "ln_", "colsum")):
        continue
    print(name[:70])
    line = []
    for c, vals in sorted(ctrs.items()):
        line.append(f"{c}={sum(vals)/len(vals):.4g}")
    print("   n=%d  " % len(next(iter(ctrs.values()))) + "  ".join(line))
