"""The Mimi many-row GEMM shapes (one 1024-frame chunk: 32 slots x 32 frames) on the fp32-MFMA kernel vs the bf16x3-split one."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import os  # noqa: E402

from smoltts_amd import engine as E, ops  # noqa: E402

if any(k in os.environ for k in ("SMOLTTS_B3_TILE", "SMOLTTS_B3_TAPS", "SMOLTTS_B3_XCD", "SMOLTTS_CONV_XS", "SMOLTTS_LINEAR_XS")):
    from smoltts_amd.build import build_library  # the experiment switches exist only in the `knobs` variant

    os.environ["SMOLTTS_LIB"] = str(build_library(variant="knobs"))
E.load_library()
shapes = [("conv0", 2048, 1024, 3584), ("convT1", 2048, 4096, 2048), ("res1.c3", 16384, 256, 1536), ("res1.c1", 16384, 512, 256),
          ("convT2", 16384, 1536, 1024), ("res2.c3", 98304, 128, 768), ("res2.c1", 98304, 256, 128), ("convT3", 98304, 640, 512),
          ("res3.c3", 491520, 64, 384), ("res3.c1", 491520, 128, 64), ("convT4", 491520, 256, 256), ("res4.c1", 1966080, 64, 32),
          ("qkv", 2048, 1536, 512), ("wo", 2048, 512, 512), ("fc1", 2048, 2048, 512), ("fc2", 2048, 512, 2048)]
tot = [0.0, 0.0]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K) / K ** 0.5
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    out = torch.empty(M, N, device="cuda")
    ts = []
    ws = torch.empty(4 * M * N, device="cuda") if M * N <= 1 << 21 else None
    for kw in ({}, {"w3": w3, "splitk_ws": ws}):
        for _ in range(2):
            ops.linear(x, w32, N, w_fp32=True, out=out, elu_out=True, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.linear(x, w32, N, w_fp32=True, out=out, elu_out=True, **kw)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 5 * 1e3)
    gf = 2.0 * M * N * K / 1e9
    tot[0] += ts[0]; tot[1] += ts[1]
    print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d} {gf:6.1f} GF: fp32 MFMA {ts[0]:8.1f} us ({gf / ts[0] * 1e3:6.1f} TF/s)   bf16x3 {ts[1]:8.1f} us ({gf / ts[1] * 1e3:6.1f} TF/s)   x{ts[0] / ts[1]:.2f}")
print(f"sum: {tot[0]:.0f} us -> {tot[1]:.0f} us")
