#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 kernel trace of the bench (graph replay): for the longest run of
back-to-back micro-steps, wall span against the sum of kernel durations, and a histogram of the gaps.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o s -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline
       python tools/trace_gaps.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
masks = [i for i, r in enumerate(rows) if 's1_mask' in r[2]]
# the timed region: the last 40 micro-steps before the 3 traced eager ones
lo, hi = masks[-44], masks[-4]
seg = rows[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
pos = [g for g in gaps if g > 0]
print(f"{len(seg)} kernels over 40 micro-steps: span {span / 40e3:.1f} us/step, kernel time {busy / 40e3:.1f} us/step, "
      f"idle {sum(pos) / 40e3:.1f} us/step in {len(pos) / 40:.0f} gaps/step (overlapping starts: {len(gaps) - len(pos)})")
for a, b in ((0, 500), (500, 1000), (1000, 2000), (2000, 5000), (5000, 20000), (20000, 10**9)):
    sel = [g for g in pos if a <= g < b]
    print(f"  gaps {a / 1e3:5.1f}-{b / 1e3 if b < 10**9 else float('inf'):6.1f} us: {len(sel) / 40:6.1f}/step, {sum(sel) / 40e3:7.1f} us/step")
# the gaps by the pair of kernels around them
from collections import defaultdict
pairs = defaultdict(lambda: [0, 0])
for i in range(len(seg) - 1):
    g = seg[i + 1][0] - seg[i][1]
    if g > 2000:
        k = (seg[i][2].split('(')[0][-45:], seg[i + 1][2].split('(')[0][-45:])
        pairs[k][0] += 1; pairs[k][1] += g
for (a, b), (n, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {g / 40e3:7.1f} us/step in {n / 40:5.2f} gaps/step (mean {g / n / 1e3:6.1f} us)  {a} -> {b}")
