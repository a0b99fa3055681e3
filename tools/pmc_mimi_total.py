"""Per-chunk HBM traffic of the Mimi decoder from a summarize_pmc.py table of tools/time_mimi.py runs.
argv: summary.txt  chunk_decodes_in_the_run.  Kernels are grouped into SEANet (convs + resnet blocks) and the decoder transformer."""
import sys

chunks = int(sys.argv[2])
tot = {"seanet": [0.0, 0.0], "transformer": [0.0, 0.0], "other": [0.0, 0.0]}
lines = []
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        continue
    name, grid, wg, n, f_kb, rd_mb, w_kb, mb = [x.strip() for x in ln.split("|")]
    n, rd, wr = int(n), float(rd_mb) * int(n) / chunks, float(w_kb) * 1024 / 1e6 * int(n) / chunks
    if n % chunks:  # not once (or k times) per chunk: set-up work of the script
        continue
    if "conv_xs_kernel<" in name and ", 2, 32, " in name:  # conv_xs.hip in Linear mode: the transformer's wqkv / wo
        grp = "transformer"
    elif "resblock" in name or "gemm_b3_kernel<4, 4, 0>" in name or "gemm_b3_kernel<4, 4, 1>" in name or "gemm_b3_kernel<2, 2, 0>" in name \
            or "gemm_b3_kernel<2, 2, 1>" in name or "seanet" in name or "conv_xs" in name or "conv_ks" in name or "halo" in name or "rvq" in name:
        grp = "seanet"
    elif "gemm_b3" in name or "attn" in name or "layernorm" in name or "splitk" in name or "mimi_rows" in name:
        grp = "transformer"
    else:
        grp = "other"
    tot[grp][0] += rd
    tot[grp][1] += wr
    lines.append(f"{grp:11s} {name[:58]:58s} grid {grid:>9s} x{n // chunks:3d}/chunk  read {rd:8.1f} MB  write {wr:8.1f} MB")
print("# per 1024-frame chunk (32 slots x 32 frames); read = 2 x FETCH_SIZE (gfx950), write = WRITE_SIZE")
print("\n".join(lines))
for g, (r, w) in tot.items():
    print(f"== {g}: read {r:.0f} MB + write {w:.0f} MB = {r + w:.0f} MB per chunk")
