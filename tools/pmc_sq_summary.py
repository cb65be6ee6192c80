"""Stall breakdown per kernel from one rocprofv3 --pmc pass of SQ counters (counter_collection.csv) -> profiles/.

    python tools/pmc_sq_summary.py <counter_collection.csv> <iterations> <out.csv>

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(MI355X_MICROARCH.md, per-instruction constants).  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import collections
import csv
import re
import sys

src, iters, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
agg = collections.OrderedDict()
for r in csv.DictReader(open(src)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = n.replace("bool _Accum", "bf16")[:100]
    a = agg.setdefault(n, collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    a["_n_" + r["Counter_Name"]] += 1
cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --pmc {' '.join(cols)} (one pass), {iters} iterations in the trace\n")
    f.write("# fractions are of SQ_WAVE_CYCLES (wave-resident quad-cycles): wait_any = parked on s_waitcnt/barrier, wait_inst = issue stall,\n")
    f.write("# active = issuing; mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)\n")
    f.write("kernel,launches_per_iter,wave_cycles_per_launch,wait_any,wait_inst_any,active_inst_any,wait_inst_lds,lds_bank_conflict,mfma_busy_per_wave_cycle\n")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
        wc = a["SQ_WAVE_CYCLES"]
        if wc <= 0:
            continue
        launches = a["_n_SQ_WAVE_CYCLES"]
        f.write(f"\"{n}\",{launches / iters:.1f},{wc / launches:.0f},{a['SQ_WAIT_ANY'] / wc:.3f},{a['SQ_WAIT_INST_ANY'] / wc:.3f},"
                f"{a['SQ_ACTIVE_INST_ANY'] / wc:.3f},{a['SQ_WAIT_INST_LDS'] / wc:.3f},{a['SQ_LDS_BANK_CONFLICT'] / wc:.4f},"
                f"{a['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc):.4f}\n")
print("wrote", dst)
