#!/usr/bin/env python3
"""Kernel timeline of a few steps from a rocprofv3 --kernel-trace run of bench.py (scripts/trace_linked.sh):
what runs between two pushes, in microseconds from the start of each push.
usage: step_timeline.py <dir with kt_kernel_trace.csv> [steps to print] [steps to skip from the end]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skip_end = int(sys.argv[3]) if len(sys.argv) > 3 else 10
path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    m = re.search(r"(k_[a-z_]+(?:<[^>]*>)?|rccl\w+|ncclDevKernel\w+)", name)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else name[:40]))
rows.sort()
push = [i for i, r in enumerate(rows) if r[2].startswith("k_push_pair")]
steps = [(push[k], push[k + 1]) for k in range(len(push) - 1)]
steps = steps[:-skip_end] if skip_end else steps
per = [rows[b][0] - rows[a][0] for a, b in steps[-200:]]
per.sort()
print(f"{len(steps)} steps in the trace; step period over the last {len(per)}: median {per[len(per) // 2] / 1e3:.1f} us, "
      f"min {per[0] / 1e3:.1f}, p90 {per[int(0.9 * len(per))] / 1e3:.1f}")
for a, b in steps[-nshow:]:
    t0 = rows[a][0]
    print(f"  step of {(rows[b][0] - t0) / 1e3:.1f} us")
    for i in range(a, b):
        s, e, n = rows[i]
        print(f"    {(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
# gaps between the launches of a step, over the last 400 steps: where the GPU waits for its next kernel
import collections
gaps = collections.defaultdict(list)
for idx, (a, b) in enumerate(steps[-400:]):
    for i in range(a, b):
        nxt = rows[i + 1]
        gaps[f"{rows[i][2][:24]} -> {nxt[2][:24]}"].append(((nxt[0] - rows[i][1]) / 1e3, idx))
print("  gaps between consecutive kernels over the last 400 steps (us): mean, median, p90, share of steps with a gap > 2 us; mean gap by step index mod 4")
for k, v in sorted(gaps.items()):
    g = sorted(x for x, _ in v)
    by = [0.0] * 4
    cnt = [0] * 4
    for x, i in v:
        by[i % 4] += x
        cnt[i % 4] += 1
    print(f"    {k:52s} {sum(g) / len(g):6.2f} {g[len(g) // 2]:6.2f} {g[int(0.9 * len(g))]:6.2f}  {sum(1 for x in g if x > 2.0) / len(g):5.2f}   "
          + " ".join(f"{by[m] / max(cnt[m], 1):5.2f}" for m in range(4)))
