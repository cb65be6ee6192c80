"""Ordered kernel timeline of ONE steady-state iteration out of a rocprofv3 --kernel-trace database (rocpd sqlite).

    python tools/kernel_timeline.py <results.db> <timeline.csv> [marker-kernel-substring, default adam_clamp]

An iteration is what lies between two consecutive launches of the marker kernel (the last kernel of `train()`).  Per launch:
start offset, duration, and the GAP to the end of the previous launch (device idle at a kernel boundary on the one stream the
path runs on).  The footer sums durations and gaps, and counts launches by duration class."""
import csv
import re
import sqlite3
import sys

db, dst = sys.argv[1:3]
marker = sys.argv[3] if len(sys.argv) > 3 else "adam_clamp"
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
if len(marks) < 3:
    sys.exit(f"fewer than three '{marker}' launches in the trace")
lo, hi = marks[-2] + 1, marks[-1] + 1          # the last complete iteration


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_114gemmfit_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E", n)
    if m:
        wgm, wgn, fm, fn, ks, conv = map(int, m.groups())
        return f"gemmfit<{16 * fm * wgm}x{16 * fn * wgn} ks{ks}{' conv' if conv else ''}>"
    return n.split("(")[0][:70]


prev_end = rows[lo - 1][2]
t0 = rows[lo][1]
tot_d = tot_g = 0
classes = {"<=6us": [0, 0.0], "6-12us": [0, 0.0], "12-30us": [0, 0.0], ">30us": [0, 0.0]}
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["#", "kernel", "start_us", "dur_us", "gap_before_us"])
    for k, (name, s, e) in enumerate(rows[lo:hi]):
        d, g = (e - s) / 1e3, (s - prev_end) / 1e3
        w.writerow([k, short(name), f"{(s - t0) / 1e3:.2f}", f"{d:.2f}", f"{g:.2f}"])
        tot_d += d
        tot_g += max(g, 0.0)
        key = "<=6us" if d <= 6 else "6-12us" if d <= 12 else "12-30us" if d <= 30 else ">30us"
        classes[key][0] += 1
        classes[key][1] += d
        prev_end = max(prev_end, e)
    n = hi - lo
    wall = (rows[hi - 1][2] - rows[lo - 1][2]) / 1e3
    w.writerow([])
    w.writerow(["# launches", n, "sum of durations us", f"{tot_d:.1f}", f"sum of gaps us {tot_g:.1f}"])
    w.writerow(["# wall (end of previous marker to end of this one) us", f"{wall:.1f}"])
    for k, (cnt, t) in classes.items():
        w.writerow([f"# {k}", cnt, f"{t:.1f} us"])
print(open(dst).read()[-600:])
