#!/usr/bin/env python3
"""Inter-kernel gaps of the headline step, eager against HIP-graph replay, from a rocprofv3 --kernel-trace CSV
(`rocprofv3 --kernel-trace -d DIR -o run --output-format csv -- python bench.py [--graph] --no-extras --no-cpu-baseline --no-other-kind`).
A step = the kernels from one sort_runs_kernel to the next; prints per kernel its mean duration and the mean idle time in front of it
(start - end of the previous kernel on the device), and the step's span.   python tools/graph_gaps.py run_kernel_trace.csv [first_kernel_substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "sort_runs_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0][-40:]
steps, cur = [], None
for r in rows:
    name = r["Kernel_Name"]
    if first in name:
        if cur:
            steps.append(cur)
        cur = []
    if cur is not None:
        cur.append((short(name), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
steps = [s for s in steps if len(s) == max(set(len(t) for t in steps), key=[len(t) for t in steps].count)]
steps = steps[len(steps) // 4:]                       # the warm part
if not steps:
    sys.exit("no steps found")
dur, gap, queues = defaultdict(list), defaultdict(list), defaultdict(set)
spans = []
for s in steps:
    for i, (n, t0, t1, q) in enumerate(s):
        dur[(i, n)].append(t1 - t0)
        queues[(i, n)].add(q)
        if i:
            gap[(i, n)].append(t0 - s[i - 1][2])
    spans.append(s[-1][2] - s[0][1])
print("%d steps of %d kernels; span first start -> last end: mean %.1f us" % (len(steps), len(steps[0]), sum(spans) / len(spans) / 1e3))
for k in sorted(dur):
    g = gap.get(k)
    print("  %-40s dur %7.1f us   idle in front %6.1f us   queue(s) %s" % (k[1], sum(dur[k]) / len(dur[k]) / 1e3, (sum(g) / len(g) / 1e3) if g else 0.0, sorted(queues[k])))
print("  sum of kernels %.1f us, sum of idle %.1f us" % (sum(sum(v) / len(v) for v in dur.values()) / 1e3, sum(sum(v) / len(v) for v in gap.values()) / 1e3))
