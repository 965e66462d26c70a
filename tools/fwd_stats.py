"""Per-forward view of a rocprofv3 kernel_stats.csv of tools/prof_fwd.py N: python3 tools/fwd_stats.py <csv> N [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
short = lambda n: re.sub(r'\(anonymous namespace\)::|ampnet::|void ', '', n)[:100]
tot = sum(float(r['TotalDurationNs']) for r in rows)
agg = sorted(((float(r['TotalDurationNs']) / steps / 1e3, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, short(r['Name'])) for r in rows), reverse=True)
print(f"forwards {steps:.0f}  kernel time {tot / steps / 1e6:.3f} ms/forward  launches/forward {sum(a[1] for a in agg):.1f}")
for a in agg[:top]:
    print(f"{a[0]:9.1f} us/fwd  {a[1]:6.1f} calls  {a[2]:8.1f} us avg  {a[3]}")
