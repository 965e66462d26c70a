"""Per-step view of a rocprofv3 kernel_stats.csv of tools/prof_step.py: launches per step, share of the < 100 us kernels."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
def short(n):
    n = re.sub(r'\(anonymous namespace\)::|ampnet::|void ', '', n)
    return n[:100]
steps = int([r for r in rows if 'reg_finalize' in r['Name']][0]['Calls'])            # one launch per step
tot = sum(float(r['TotalDurationNs']) for r in rows)
agg = sorted(((float(r['TotalDurationNs']) / steps / 1e3, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, short(r['Name'])) for r in rows), reverse=True)
small = sum(a[0] for a in agg if a[2] < 100)
print(f"steps {steps:.0f}  kernel time {tot / steps / 1e6:.3f} ms/step  launches/step {sum(a[1] for a in agg):.1f}  "
      f"kernels < 100 us: {sum(a[1] for a in agg if a[2] < 100):.1f} launches, {small:.0f} us/step = {small / (tot / steps / 1e3):.3f} of the step")
for a in agg[:top]:
    print(f"{a[0]:9.1f} us/step  {a[1]:6.1f} calls  {a[2]:8.1f} us avg  {a[3]}")
