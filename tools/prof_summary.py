#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats CSV directory.
usage: python tools/prof_summary.py gpurun_out/profN [steps]"""
import csv, glob, sys, re
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms ({tot/1e6/steps:.2f} ms/step over {steps:g} steps)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("scm::", "")
    print(f"{name[:60]:60s} calls {int(r['Calls']):5d} ms/step {float(r['TotalDurationNs'])/1e6/steps:7.3f} "
          f"avg_us {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
