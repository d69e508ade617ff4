"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (us per step): python tools/diff_stats.py a.csv b.csv <steps>"""
import csv
import sys


def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}


a, b, steps = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    rows.append(((tb - ta) / steps / 1e3, k, ca, cb, ta / steps / 1e3, tb / steps / 1e3))
rows.sort(key=lambda r: -abs(r[0]))
print(f"total us/step: a {sum(v[1] for v in a.values()) / steps / 1e3:.1f}  b {sum(v[1] for v in b.values()) / steps / 1e3:.1f}")
for d, k, ca, cb, ta, tb in rows[:25]:
    print(f"{d:+9.1f} us/step  {k[:70]:70s} calls {ca:5d} -> {cb:5d}   {ta:8.1f} -> {tb:8.1f}")
