"""Copy the summaries that tools/collect_profile.sh, collect_pmc.sh, profile_mimi.sh and pmc_mimi.sh left under gpurun_out/ into
profiles/ under the round's tag, with the header lines that say what each file is.  argv: [tag] (default r02).
Run in the repo root after the GPU call has merged gpurun_out/ back."""
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
G, P = ROOT / "gpurun_out", ROOT / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"

shutil.copy(G / f"prof_{tag}_kernel_stats.csv", P / f"{tag}_bench_kernel_stats.csv")
log = (G / f"prof_{tag}.log").read_text(errors="replace").splitlines()
bench_lines = [ln for ln in log if ln.startswith("[bench") and ("timed" in ln or "in-situ" in ln)] + [ln for ln in log if ln.startswith("{")]
head = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --cpu-frames 0 --no-latency --no-overlap-mimi   (tools/collect_profile.sh "
        f"{tag}; the default bench command minus the CPU sample and the latency probe, Mimi decode on the frame graphs' stream)",
        "# smoltts_byte_150m, B=32, chunk 32, 9 steps incl. warm-up + the duplicated-launch timing frames; per launch shape; durations in us; MI355X"]
head += ["# bench (profiler attached): " + ln for ln in bench_lines]
(P / f"{tag}_bench_kernel_breakdown.txt").write_text("\n".join(head) + "\n" + (G / f"prof_{tag}_breakdown.txt").read_text())

shutil.copy(G / "pmc_summary.txt", P / f"{tag}_pmc_fetch_write.txt")
shutil.copy(G / "pmc_summary_w13.json", P / f"{tag}_pmc_w13.json")
for k in ("fetch", "write"):
    keep = [ln for ln in (G / f"pmc_{k}.log").read_text(errors="replace").splitlines() if not re.match(r"^[EWI]\d{8} ", ln) or "rocprofv3" in ln]
    (P / f"{tag}_pmc_{k}.log").write_text("\n".join(keep[-40:]) + "\n")

mimi = (G / f"mimi_prof_{tag}.txt").read_text()
t = [ln for ln in (G / "prof_mimi.log").read_text(errors="replace").splitlines() if ln.startswith("Mimi chunk decode")]
(P / f"{tag}_mimi_chunk_kernel_breakdown.txt").write_text(
    "# rocprofv3 --kernel-trace -- python3 tools/time_mimi.py   (tools/profile_mimi.sh): 21 chunk decodes of 32 slots x 32 frames; durations in us; MI355X\n"
    + "".join("# (profiler attached) " + ln + "\n" for ln in t) + mimi)
shutil.copy(G / f"pmc_mimi_{tag}.txt", P / f"{tag}_pmc_mimi_chunk.txt")
print("installed:", *sorted(p.name for p in P.glob(f"{tag}_*")))
