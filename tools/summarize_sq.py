"""Per launch shape: SQ counter ratios from a rocprofv3 --pmc pass (SQ_* counters listed on the command line)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].replace("void smoltts::", "").replace("smoltts::", "")[:44], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[key] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print(f"{'kernel':44s} {'grid':>9s} {'n':>3s} {'wave_cyc/launch':>15s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'wait_lds':>8s} {'mfma_busy':>9s} {'lds_conf/act':>12s}")
for (name, grid, wg), c in rows[:24]:
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    n = max(cnt[(name, grid, wg)], 1)
    print(f"{name:44s} {grid:9d} {n:3d} {wc / n:15.3e} {c.get('SQ_WAIT_ANY', 0) / wc:8.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:9.2f} "
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:7.2f} {c.get('SQ_WAIT_INST_LDS', 0) / wc:8.2f} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * wc):9.2f} "
          f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):12.2f}")
