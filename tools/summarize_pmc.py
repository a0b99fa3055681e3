"""FETCH_SIZE / WRITE_SIZE per launch shape from two rocprofv3 --pmc passes (tools/collect_pmc.sh).

gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte-per-lane stores.
Counter values are in KB per dispatch."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(root, counter):
    agg = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (r["Kernel_Name"].replace("void smoltts::", "").replace("smoltts::", "")[:70], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            agg[key].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for key, v in fetch.items():
    w = write.get(key, [0.0])
    f_kb, w_kb = sum(v) / len(v), sum(w) / len(w)
    rows.append((sum(v), key, len(v), f_kb, w_kb))
rows.sort(reverse=True)
with open(sys.argv[3] + ".txt", "w") as out:
    out.write("# kernel | grid | wg | dispatches | FETCH_SIZE avg KB | corrected read MB (2x) | WRITE_SIZE avg KB | HBM MB per launch\n")
    for _, (name, grid, wg), n, f_kb, w_kb in rows[:40]:
        out.write(f"{name} | {grid} | {wg} | {n} | {f_kb:.1f} | {2 * f_kb * 1024 / 1e6:.2f} | {w_kb:.1f} | {(2 * f_kb + w_kb) * 1024 / 1e6:.2f}\n")
# the bench's roofline kernel: the SwiGLU GEMM of the 150m model at 32 rows (N = 6144 -> grid 128 x 2 x 512 threads); since round 3
# it has two instantiations (weights streamed past the caches for the slow layers, cached for the depth layers): weighted by dispatches
sel = [(name, n, f_kb, w_kb) for _, (name, grid, wg), n, f_kb, w_kb in rows if name.startswith("gemm3_kernel<1, 3, 3, 2") and grid == 128 * 2 * 512]
if sel:
    n_all = sum(x[1] for x in sel)
    f_kb = sum(x[1] * x[2] for x in sel) / n_all
    w_kb = sum(x[1] * x[3] for x in sel) / n_all
    json.dump({"kernel": " + ".join(x[0] for x in sel) + " (w1|w3 GEMM + SwiGLU, 150m, B=32)", "dispatches": n_all, "FETCH_SIZE_avg_KB": f_kb, "WRITE_SIZE_avg_KB": w_kb,
               "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024),
               "per_instantiation": [{"kernel": x[0], "dispatches": x[1], "hbm_bytes_per_launch": int((2 * x[2] + x[3]) * 1024)} for x in sel],
               "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950), write = WRITE_SIZE x 1024",
               "source": sys.argv[4] if len(sys.argv) > 4 else "tools/collect_pmc.sh"}, open(sys.argv[3] + "_w13.json", "w"), indent=1)
print(open(sys.argv[3] + ".txt").read()[:3000])
