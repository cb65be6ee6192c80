"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into HBM bytes per
kernel family, per iteration and per launch.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <iterations> <out.csv> [<out.json>]

The optional JSON ({"bytes_per_launch", ...} of the GEMM engine) is what bench.py quotes as `roofline.traffic_pmc_profile`
when it is committed as profiles/r02_<config>_hbm_traffic.json.

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are reported in KiB;
FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so the corrected fetch doubles it."""
import csv
import re
import sys
from collections import OrderedDict

fetch_csv, write_csv, iters, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]

FAMILIES = [
    ("GEMM engine (gemmfit_kernel / gemm_glds_kernel / gemm8p_kernel / gemmrow_kernel / gemmrowconv_kernel / gemm_kernel / splitk_reduce, all shapes)",
     r"gemmfit_kernel|gemm_glds_kernel|gemm8p_kernel|gemmrow_kernel|gemmrowconv_kernel|gemm_kernel|gemm_f32_kernel|splitk_reduce"),
    ("GroupNorm kernels", r"gn_stats|gn_apply"),
    ("LayerNorm", r"ln_fwd|ln_bwd"),
    ("ViT attention", r"mha_"),
    ("cutout kernels", r"warp_|pool_|patchify|reduce_planes|minmax|colminmax"),
    ("other", r"."),
]


def family(name):
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "other"


def load(path, counter):
    agg = OrderedDict((f, [0, 0.0]) for f, _ in FAMILIES)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[family(r["Kernel_Name"])]
        if "splitk_reduce" not in r["Kernel_Name"]:      # the reduce pass belongs to its GEMM launch: bytes yes, launch no
            a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


f = load(fetch_csv, "FETCH_SIZE")
w = load(write_csv, "WRITE_SIZE")
with open(dst, "w") as out:
    out.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of the same bench.py command), {iters} iterations in each trace\n")
    out.write("# units: rocprofv3 reports KiB; FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> 'fetch_corrected' doubles it\n")
    out.write("kernel_family,launches,fetch_KiB_raw,fetch_MB_corrected_per_iter,write_MB_per_iter,launches_per_iter,hbm_MB_per_launch(corrected fetch + write)\n")
    for fam, _ in FAMILIES:
        n, fk = f[fam]
        _, wk = w[fam]
        if n == 0:
            continue
        fmb = 2.0 * fk * 1024 / 1e6 / iters
        wmb = wk * 1024 / 1e6 / iters
        out.write(f"\"{fam}\",{n},{fk:.0f},{fmb:.1f},{wmb:.1f},{n / iters:.1f},{(fmb + wmb) / (n / iters):.2f}\n")
# per-kernel table (top 45 by bytes): which launches re-read their operands
per = {}
for path, counter, slot in ((fetch_csv, "FETCH_SIZE", 0), (write_csv, "WRITE_SIZE", 1)):
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        e = per.setdefault(r["Kernel_Name"], [0, 0.0, 0.0])
        if slot == 0:
            e[0] += 1
        e[1 + slot] += float(r["Counter_Value"])
with open(dst, "a") as out:
    out.write("# per kernel: name,launches_per_iter,fetch_MB_corrected_per_launch,write_MB_per_launch\n")
    for name, (n, fk, wk) in sorted(per.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2]))[:45]:
        if n:
            out.write(f"\"{name[:150]}\",{n / iters:.1f},{2.0 * fk * 1024 / 1e6 / n:.2f},{wk * 1024 / 1e6 / n:.2f}\n")
print(open(dst).read())
if len(sys.argv) > 5:
    import json
    fam = FAMILIES[0][0]
    n, fk = f[fam]
    _, wk = w[fam]
    if n:
        json.dump({"kernel_family": fam, "iterations_in_trace": iters, "launches_per_iter": n / iters,
                   "fetch_bytes_per_iter_corrected": 2.0 * fk * 1024 / iters, "write_bytes_per_iter": wk * 1024 / iters,
                   "bytes_per_launch": (2.0 * fk + wk) * 1024 / n,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the bench.py command; KiB units; "
                             "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md"}, open(sys.argv[5], "w"), indent=1)
