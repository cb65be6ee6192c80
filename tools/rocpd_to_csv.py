"""rocprofv3's default output here is a rocpd sqlite database; turn it into the two CSV shapes the summarisers read.

    python tools/rocpd_to_csv.py stats    <results.db> <kernel_stats.csv>
    python tools/rocpd_to_csv.py counters <results.db> <counter_collection.csv>
"""
import csv
import sqlite3
import sys

mode, db, dst = sys.argv[1:4]
c = sqlite3.connect(db)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    if mode == "stats":
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
        rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc")
        w.writerows(rows)
    else:
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerows(c.execute("select kernel_name, counter_name, value from counters_collection"))
print("wrote", dst)
