"""One train step as the ordered list of its kernel dispatches, from a rocprofv3 --kernel-trace CSV of tools/prof_step.py:
    python3 tools/step_trace.py <dir or *_kernel_trace.csv> [which_step_from_the_end=1] [min_us=0]
Prints index, start offset, duration and the idle gap before each dispatch; the step is delimited by the last adam_kernel of the previous step."""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|ampnet::|void |at::native::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:90]


adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
# the last adam launch of every step: one launch per step (FusedAdam.step_together) or two back to back (one per optimiser)
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 4]
lo, hi = ends[-1 - back] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = int(rows[lo - 1]["End_Timestamp"])
busy = gaps = 0.0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d, g = (e - s) / 1e3, (s - prev_end) / 1e3
    busy += d
    gaps += max(g, 0.0)
    if d >= min_us:
        print(f"{i:4d} {(s - t0) / 1e3:10.1f} us  {d:8.1f} us  gap {g:6.1f}  grid {r.get('Grid_Size', '?'):>8} wg {r.get('Workgroup_Size', '?'):>5}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
wall = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
print(f"{len(step)} dispatches, wall {wall:.1f} us, kernel time {busy:.1f} us, idle gaps {gaps:.1f} us")
