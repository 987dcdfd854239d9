#!/usr/bin/env python3
"""Per-launch durations of one training step from a rocprofv3 kernel trace (csv): the step's launches in order, averaged over
the steps of the run.  usage: tools/train_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1] for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
start = [int(r["Start_Timestamp"]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "dw_adam_kernel"]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seq = list(range(a + 1, b + 1))
    if len(seq) <= 7:
        steps.append(seq)
steps = steps[len(steps) // 2:]
n = len(steps[0])
steps = [s for s in steps if len(s) == n]
print("%d steps" % len(steps))
for j in range(n):
    d = sum(dur[s[j]] for s in steps) / len(steps)
    gap = sum(start[s[j]] - (start[s[j] - 1] + dur[s[j] - 1]) for s in steps) / len(steps)
    print("%-22s %7.2f us   (gap before %5.2f us)" % (names[steps[0][j]], d / 1e3, gap / 1e3))
print("step %.2f us" % (sum((start[s[-1]] + dur[s[-1]]) - (start[s[0] - 1] + dur[s[0] - 1]) for s in steps) / len(steps) / 1e3))
