"""Condense a rocprofv3 `*_kernel_stats.csv` into a short per-kernel table (name shortened), for profiles/."""
import csv, re, sys
src, dst, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = list(csv.DictReader(open(src)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = n.replace("bool _Accum", "bf16")
    if "distribution_elementwise_grid_stride_kernel" in n: n = "at::native normal_ (torch.randn noise)"
    return n[:110]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary; {iters} iterations in the trace; total kernel time {tot/1e6:.2f} ms "
            f"= {tot/1e6/iters:.3f} ms/iteration\n")
    f.write("kernel,calls,calls_per_iter,total_ms,ms_per_iter,avg_us,pct\n")
    for r in rows:
        t = float(r["TotalDurationNs"])
        f.write(f"\"{short(r['Name'])}\",{r['Calls']},{int(r['Calls'])/iters:.1f},{t/1e6:.3f},{t/1e6/iters:.4f},{float(r['AverageNs'])/1e3:.2f},{100*t/tot:.2f}\n")
print("wrote", dst)
