"""Per-iteration view of a rocprofv3 kernel-statistics table (tools/rocpd_to_csv.py stats output).

    python tools/kernel_stats_summary.py <raw_stats.csv> <iterations in the trace> <out.csv>
"""
import csv
import sys

src, iters, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(src)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary; {iters} iterations in the trace; total kernel time {total / 1e6:.2f} ms = "
            f"{total / 1e6 / iters:.3f} ms/iteration\n")
    f.write("kernel,calls,calls_per_iter,total_ms,ms_per_iter,avg_us,pct\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        t = float(r["TotalDurationNs"])
        name = r["Name"].replace('"', "'")
        f.write(f"\"{name[:160]}\",{r['Calls']},{int(r['Calls']) / iters:.1f},{t / 1e6:.3f},{t / 1e6 / iters:.4f},{float(r['AverageNs']) / 1e3:.2f},"
                f"{100 * t / total:.2f}\n")
print(open(dst).read()[:6000])
