"""Per launch shape (kernel, grid, block) summary of a rocprofv3 kernel trace (CSV or rocpd .db)."""
import csv
import glob
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1),
                     int(r["Workgroup_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for f in glob.glob(root + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for r in c.execute("select name, grid_x, grid_y, workgroup_x, workgroup_y, end-start from kernels"):
        rows.append((r[0], r[1] // max(r[3], 1), r[2] // max(r[4], 1), r[3], r[5]))
agg = defaultdict(list)
for name, gx, gy, wg, d in rows:
    agg[(name.replace("void smoltts::", "").replace("smoltts::", "")[:60], gx, gy, wg)].append(d)
tot = sum(sum(v) for v in agg.values())
print(f"# total kernel time {tot / 1e6:.2f} ms over {len(rows)} launches")
for (name, gx, gy, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name:60s} wg={wg:5d} blocks=({gx},{gy}) n={len(v):6d} avg={sum(v) / len(v) / 1e3:8.2f} min={min(v) / 1e3:7.2f} total_ms={sum(v) / 1e6:8.2f} {100 * sum(v) / tot:5.1f}%")
